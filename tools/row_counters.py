"""SQ counters of one shape's kernels (development evidence: occupancy and LDS pressure of the long rows).

    python3 tools/row_counters.py 32768 complex64 8192 [more "N dtype batch" triples ...]

Runs `rocprofv3 --pmc <pair> --kernel-trace -- python3 tools/quick_bench.py one N dtype batch` once per counter pair (counters in
their own runs, the program directly after `--`) and prints the per-dispatch average of every counter for every mifft kernel."""
import csv
import glob
import os
import shutil
import subprocess
import sys
from collections import defaultdict

PAIRS = [["SQ_WAVES", "SQ_WAVE_CYCLES"], ["SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU"], ["SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT"],
         ["SQ_INSTS_LDS", "SQ_WAIT_INST_LDS"], ["SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"], ["SQ_INSTS_VALU", "SQ_INSTS_VMEM"]]


def main():
    args = sys.argv[1:]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    os.environ.setdefault("TMPDIR", "/tmp")
    for i in range(0, len(args) - 2, 3):
        n, dt, batch = args[i:i + 3]
        acc = defaultdict(lambda: [0.0, set()])
        for pair in PAIRS:
            d = os.path.join(root, "gpurun_out", "row_counters", "%s_%s" % (n, "_".join(pair)))
            shutil.rmtree(d, ignore_errors=True)
            cmd = ["rocprofv3", "--pmc"] + pair + ["--kernel-trace", "--output-format", "csv", "-d", d, "--",
                   "python3", os.path.join(root, "tools", "quick_bench.py"), "one", n, dt, batch]
            r = subprocess.run(cmd, cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
            line = [l for l in r.stdout.splitlines() if "GFLOPS" in l]
            for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
                for row in csv.DictReader(open(fn)):
                    if "mifft" not in row["Kernel_Name"]:
                        continue
                    k = (row["Kernel_Name"].split("(")[0][:90], row["Counter_Name"])
                    acc[k][0] += float(row["Counter_Value"])
                    acc[k][1].add(row["Dispatch_Id"])
        print("== %s %s batch %s   %s" % (n, dt, batch, line[-1].strip() if line else "(no timing line)"))
        for (kn, cn), (v, disp) in sorted(acc.items()):
            print("  %-92s %-22s %.4e per dispatch (%d dispatches)" % (kn, cn, v / max(1, len(disp)), len(disp)))


if __name__ == "__main__":
    main()
