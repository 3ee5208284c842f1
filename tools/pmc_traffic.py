#!/usr/bin/env python3
"""Regenerate profiles/traffic_<config>.json (bench.py's roofline.traffic) and the kernel-stats summary for one or more
bench configurations.  Run ON THE GPU BOX from the repo root:

    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && python3 tools/pmc_traffic.py [--tag r02] c2 c3 c4 c4s c5

For each config three runs of `python3 bench.py --config C --plain --steps 3 --warmup 1` (the program directly after
`--`, as the pool requires): rocprofv3 --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE (the two counters do
not fit one pass: MI355X_MICROARCH.md, PMC slots).  FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 B:
calibrated on the chain strategy in round 1, profiles/r01_b_c2_pmc_summary.json); both are KiB.  The counters sit
between L2 and the fabric, so Infinity-Cache hits are included.  Writes, under profiles/:
    traffic_<config>.json                        what bench.py reports as roofline.traffic
    <tag>_<config>_kernel_stats.csv              per-kernel calls / average duration (rocprofv3 --stats)
This script itself never touches the GPU (it only starts rocprofv3 as a child process).
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STEPS, WARMUP = 3, 1
# (the number of executes of a run -- parity gate, clock spin-up, warm-up, timed steps -- is what bench.py itself counted:
# "executes_in_this_process" of its line; rounds 1-5 assumed 1 + WARMUP + STEPS, which the spin-up of round 6 made wrong)
STATS_STEPS = 12                         # the kernel-stats run: enough warm calls that the two cold ones (parity gate, warm-up)
                                         # move the average by < 2 %; the csv's MinNs column is the warm figure


def run(kind, cfg, outdir):
    shutil.rmtree(outdir, ignore_errors=True)
    opts = ["--kernel-trace", "--stats"] if kind == "stats" else ["--pmc", kind, "--kernel-trace"]
    steps = STATS_STEPS if kind == "stats" else STEPS
    cmd = ["rocprofv3"] + opts + ["--output-format", "csv", "-d", outdir, "--", "python3", os.path.join(ROOT, "bench.py"),
           "--config", cfg, "--plain", "--steps", str(steps), "--warmup", str(WARMUP)] + (["--chunk-only"] if cfg == "c5" else [])
    p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    if p.returncode != 0 or not line:
        raise SystemExit("%s failed (rc %d): %s" % (" ".join(cmd), p.returncode, p.stderr[-1500:]))
    return json.loads(line[-1])


def counter_sum(outdir, name):
    total, rows = 0.0, 0
    for fn in glob.glob(outdir + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] == name and "mifft::" in r["Kernel_Name"]:
                total += float(r["Counter_Value"])
                rows += 1
    return total, rows


def library_digest():
    import hashlib
    with open(os.path.join(ROOT, "pyfft_amd", "libmifft.so"), "rb") as f:
        return hashlib.sha256(f.read()).hexdigest()[:16]


def tree_commit():
    try:
        return open(os.path.join(ROOT, ".tree_commit")).read().strip() or None
    except OSError:
        return None


def main():
    args = sys.argv[1:]
    tag = "r03"
    if args and args[0] == "--tag":
        tag, args = args[1], args[2:]
    scratch = os.path.join(ROOT, "gpurun_out", "pmc_traffic")
    for cfg in args or ["c2"]:
        line = run("stats", cfg, os.path.join(scratch, cfg, "stats"))
        for fn in glob.glob(os.path.join(scratch, cfg, "stats") + "/**/*kernel_stats.csv", recursive=True):
            shutil.copy(fn, os.path.join(ROOT, "profiles", "%s_%s_kernel_stats.csv" % (tag, cfg)))
        ef = run("FETCH_SIZE", cfg, os.path.join(scratch, cfg, "fetch"))["executes_in_this_process"]
        ew = run("WRITE_SIZE", cfg, os.path.join(scratch, cfg, "write"))["executes_in_this_process"]
        fetch_kib, nf = counter_sum(os.path.join(scratch, cfg, "fetch"), "FETCH_SIZE")
        write_kib, nw = counter_sum(os.path.join(scratch, cfg, "write"), "WRITE_SIZE")
        alg = line["roofline"]["algorithmic_bytes_per_step"]
        traffic = (2.0 * fetch_kib / ef + write_kib / ew) * 1024.0
        out = {"hbm_bytes_per_step": traffic,
               "traffic_over_algorithmic": traffic / alg,
               "strategy": line["config"]["strategy"],
               "algorithmic_bytes_per_step": alg,
               "fetch_KiB_raw_per_step": fetch_kib / ef, "write_KiB_per_step": write_kib / ew,
               "dispatches_counted": [nf, nw], "executes_per_run": [ef, ew],
               "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --config %s --plain --steps %d --warmup %d%s" % (cfg, STEPS, WARMUP, " --chunk-only" if cfg == "c5" else ""),
               "method": "tools/pmc_traffic.py: separate passes per counter, every mifft:: dispatch of the run summed and divided by the executes bench.py counted in that run; FETCH_SIZE x2 (gfx950 counts 128-B requests as 64 B); L2<->fabric bytes, Infinity-Cache hits included",
               "bench_frac_under_profiler": line["roofline"]["frac"],
               # what was measured: the library by content, the tree by the commit gpurun shipped (.tree_commit, written by the caller:
               # `git rev-parse HEAD > .tree_commit` -- the GPU box has no .git)
               "libmifft_sha256_16": library_digest(), "commit": tree_commit()}
        json.dump(out, open(os.path.join(ROOT, "profiles", "traffic_%s.json" % cfg), "w"), indent=1)
        print("%s: strategy %s  traffic %.3f x algorithmic  (fetch x2 %.2f GiB + write %.2f GiB per step)" % (
            cfg, out["strategy"], out["traffic_over_algorithmic"], 2 * fetch_kib / ef / 2**20, write_kib / ew / 2**20), flush=True)


if __name__ == "__main__":
    main()
