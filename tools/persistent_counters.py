#!/usr/bin/env python3
"""SQ / TCP / TCC counters of the persistent and pass-pair kernels, side by side (round 6, review item 2).

Run ON THE GPU BOX from the repo root:

    cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && python3 tools/persistent_counters.py [--out FILE] [case ...]

A case is a name of CASES below (default: all of them).  For every case and every counter group one run of
`rocprofv3 --pmc <group> --kernel-trace -- python3 bench.py <case args> --plain` (counters in their own runs, the program
directly after `--`; a group that the part rejects is retried one counter at a time and the rejected names are listed).
The table printed at the end has one column per case, one row per counter: the SUM over the kernel's dispatches divided
by the number of dispatches, for the dominant mifft kernel of the case (the one with the largest SQ_WAVE_CYCLES or, when
that group failed, the most dispatches).  Derived rows follow the guide's identities (MI355X_MICROARCH.md, PMC slots):
WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES.  This script itself never touches the GPU.
"""
import argparse
import csv
import glob
import os
import shutil
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    # the 512-thread persistent kernel of configuration 5 (one 256-transform chunk, out of place)
    "c5": ["--config", "c5", "--chunk-only", "--steps", "3", "--warmup", "1"],
    # the 256-thread persistent kernel of configuration 2 at the same bytes per step (512 x 8 MiB = 256 x 16 MiB ... 4 GiB per side)
    "c2": ["--config", "c2", "--batch", "512", "--steps", "3", "--warmup", "1"],
    # configuration 4: PairXY + PairYZ
    "c4": ["--config", "c4", "--steps", "3", "--warmup", "1"],
    # configuration 3: the 2-D persistent kernel
    "c3": ["--config", "c3", "--steps", "3", "--warmup", "1"],
}

GROUPS = [
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY",
     "SQ_ACTIVE_INST_VALU"],
    ["SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC", "SQ_INSTS_VALU", "SQ_INSTS_LDS",
     "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR"],
    ["SQ_INSTS_SALU", "SQ_INSTS_SMEM", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT", "SQ_WAVE32_INSTS",
     "SQ_INST_CYCLES_VMEM", "SQ_WAIT_INST_ANY"],
    ["TCP_PENDING_STALL_CYCLES_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum", "TCP_TCC_ATOMIC_WITH_RET_REQ_sum"],
    ["TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_TCC_NC_READ_REQ_sum", "TCP_TCC_UC_READ_REQ_sum", "TCP_TCC_CC_READ_REQ_sum"],
    ["TCP_TA_TCP_STATE_READ_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", "TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum",
     "TCP_TCR_TCP_STALL_CYCLES_sum"],
    ["TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_READ_sum"],
    ["TCC_WRITE_sum", "TCC_ATOMIC_sum", "TCC_STREAMING_REQ_sum", "TCC_NC_REQ_sum"],
    ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"],
    ["TCC_EA0_WRREQ_STALL_sum", "TCC_EA0_RD_UNCACHED_32B_sum", "TCC_TAG_STALL_sum", "TCC_TOO_MANY_EA_WRREQS_STALL_sum"],
    ["TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_WRREQ_LEVEL_sum", "TCC_BUSY_sum", "TCC_CYCLE_sum"],
    ["TCC_WRITEBACK_sum", "TCC_NORMAL_WRITEBACK_sum", "TCC_NORMAL_EVICT_sum", "TCC_ALL_TC_OP_WB_WRITEBACK_sum"],
    ["GRBM_GUI_ACTIVE", "GRBM_COUNT"],
]


def available():
    try:
        p = subprocess.run(["rocprofv3", "-L"], cwd="/tmp", stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    except Exception as e:                                    # noqa: BLE001 - evidence tool: report and go on
        print("# rocprofv3 -L failed: %s" % e)
        return None
    names = set()
    for tok in p.stdout.replace(",", " ").replace(":", " ").split():
        if tok[:3] in ("SQ_", "TCP", "TCC", "GRB", "TA_", "TD_"):
            names.add(tok.strip())
    return names or None


def run(case, group, tag):
    d = os.path.join(ROOT, "gpurun_out", "persistent_counters", "%s_%s" % (case, tag))
    shutil.rmtree(d, ignore_errors=True)
    cmd = ["rocprofv3", "--pmc"] + group + ["--kernel-trace", "--output-format", "csv", "-d", d, "--", "python3",
           os.path.join(ROOT, "bench.py")] + CASES[case] + ["--plain"]
    p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    rows = []
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if "mifft" in r["Kernel_Name"]:
                rows.append((r["Kernel_Name"].split("(")[0], r["Counter_Name"], float(r["Counter_Value"]), r["Dispatch_Id"]))
    shutil.rmtree(d, ignore_errors=True)
    return (p.returncode == 0 and bool(line) and bool(rows)), rows, (p.stderr[-600:] if p.returncode else "")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("cases", nargs="*", default=[])
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    cases = a.cases or list(CASES)
    os.environ.setdefault("TMPDIR", "/tmp")
    out = open(a.out, "w") if a.out else sys.stdout

    def say(s=""):
        print(s, file=out)
        out.flush()

    names = available()
    groups = GROUPS
    if names:
        groups = [[c for c in g if c in names] for g in GROUPS]
        dropped = [c for g in GROUPS for c in g if c not in names]
        groups = [g for g in groups if g]
        say("# counters the part does not list (dropped): %s" % (", ".join(dropped) or "none"))
    # acc[case][kernel][counter] = [sum, set(dispatch ids)]
    acc = {c: defaultdict(lambda: defaultdict(lambda: [0.0, set()])) for c in cases}
    rejected = []
    for case in cases:
        for gi, g in enumerate(groups):
            ok, rows, err = run(case, g, "g%d" % gi)
            todo = []
            if not ok:
                say("# %s: group %s failed as a group, retrying singly  %s" % (case, g, err.replace("\n", " ")[-200:]))
                todo = [[c] for c in g]
            for ci, single in enumerate(todo):
                ok1, rows1, err1 = run(case, single, "g%d_%d" % (gi, ci))
                if ok1:
                    rows += rows1
                else:
                    rejected.append((case, single[0]))
            seen = set()
            for kn, cn, v, disp in rows:
                if (kn, cn, disp) in seen:                   # a counter listed in two groups: keep the first
                    continue
                seen.add((kn, cn, disp))
                if disp in acc[case][kn][cn][1]:
                    continue
                acc[case][kn][cn][0] += v
                acc[case][kn][cn][1].add(disp)
            say("# %s group %d done (%d rows)" % (case, gi, len(rows)))
    if rejected:
        say("# rejected singly: %s" % rejected)

    # the dominant kernel(s) of each case: everything with >= 20 % of the case's largest SQ_WAVE_CYCLES (c4 has two)
    cols = []
    for case in cases:
        ks = acc[case]
        weight = {k: (v["SQ_WAVE_CYCLES"][0] if "SQ_WAVE_CYCLES" in v else sum(len(x[1]) for x in v.values())) for k, v in ks.items()}
        if not weight:
            continue
        top = max(weight.values())
        for k in sorted(ks, key=lambda k: -weight[k]):
            if weight[k] >= 0.2 * top:
                cols.append((case, k))
    say()
    for i, (case, k) in enumerate(cols):
        say("# column %d: %s  %s" % (i, case, k))
    allc = []
    for g in groups:
        for c in g:
            if c not in allc:
                allc.append(c)

    def per(case, k, c):
        v = acc[case][k].get(c)
        return None if not v or not v[1] else v[0] / len(v[1])

    say("%-42s" % "counter (per dispatch)" + "".join("%16s" % ("col%d" % i) for i in range(len(cols))))
    for c in allc:
        vals = [per(case, k, c) for case, k in cols]
        if all(v is None for v in vals):
            continue
        say("%-42s" % c + "".join("%16s" % ("-" if v is None else "%.4e" % v) for v in vals))
    say()
    derived = [
        ("WAIT_ANY / WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES"),
        ("WAIT_INST_ANY / WAVE_CYCLES", "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"),
        ("WAIT_INST_LDS / WAVE_CYCLES", "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES"),
        ("ACTIVE_INST_ANY / WAVE_CYCLES", "SQ_ACTIVE_INST_ANY", "SQ_WAVE_CYCLES"),
        ("ACTIVE_INST_VALU / WAVE_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"),
        ("ACTIVE_INST_LDS / WAVE_CYCLES", "SQ_ACTIVE_INST_LDS", "SQ_WAVE_CYCLES"),
        ("ACTIVE_INST_VMEM / WAVE_CYCLES", "SQ_ACTIVE_INST_VMEM", "SQ_WAVE_CYCLES"),
        ("WAVE_CYCLES / BUSY_CYCLES (waves per SQ)", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES"),
        ("LDS_BANK_CONFLICT / LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"),
        ("TCC_HIT / TCC_REQ", "TCC_HIT_sum", "TCC_REQ_sum"),
        ("TCC_MISS / TCC_REQ", "TCC_MISS_sum", "TCC_REQ_sum"),
        ("EA0_WRREQ_STALL / EA0_WRREQ", "TCC_EA0_WRREQ_STALL_sum", "TCC_EA0_WRREQ_sum"),
        ("EA0_RDREQ_LEVEL / EA0_RDREQ (read latency, TCC cycles)", "TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_RDREQ_sum"),
        ("EA0_WRREQ_LEVEL / EA0_WRREQ (write latency, TCC cycles)", "TCC_EA0_WRREQ_LEVEL_sum", "TCC_EA0_WRREQ_sum"),
        ("TCP_PENDING_STALL / TCP_TCC_READ_REQ", "TCP_PENDING_STALL_CYCLES_sum", "TCP_TCC_READ_REQ_sum"),
        ("INSTS_VALU per wave", "SQ_INSTS_VALU", "SQ_WAVES"),
        ("INSTS_LDS per wave", "SQ_INSTS_LDS", "SQ_WAVES"),
        ("INSTS_VMEM_RD per wave", "SQ_INSTS_VMEM_RD", "SQ_WAVES"),
        ("INSTS_VMEM_WR per wave", "SQ_INSTS_VMEM_WR", "SQ_WAVES"),
    ]
    say("%-58s" % "derived" + "".join("%12s" % ("col%d" % i) for i in range(len(cols))))
    for name, num, den in derived:
        vals = []
        for case, k in cols:
            n, d = per(case, k, num), per(case, k, den)
            vals.append(None if n is None or not d else n / d)
        if all(v is None for v in vals):
            continue
        say("%-58s" % name + "".join("%12s" % ("-" if v is None else "%.4g" % v) for v in vals))


if __name__ == "__main__":
    main()
