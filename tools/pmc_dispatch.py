"""Per-dispatch sums of rocprofv3 --pmc counters (one directory per counter run), in dispatch order:
    python3 tools/pmc_dispatch.py <dir FETCH_SIZE run> <dir WRITE_SIZE run>
FETCH_SIZE is doubled (MI355X_MICROARCH.md: it tallies 128-byte requests at 64 B on gfx950); both are printed in MiB
(the counters are in KiB)."""
import csv
import glob
import sys
from collections import OrderedDict


def load(d):
    out = OrderedDict()
    for fn in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
        for row in csv.DictReader(open(fn)):
            k = int(row["Dispatch_Id"])
            name, val = out.get(k, (row["Kernel_Name"], {}))
            val[row["Counter_Name"]] = val.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            out[k] = (name, val)
    return out


runs = [load(d) for d in sys.argv[1:]]
ids = sorted(set().union(*[set(r) for r in runs]))
for i in ids:
    name = next(r[i][0] for r in runs if i in r)
    vals = {}
    for r in runs:
        if i in r:
            vals.update(r[i][1])
    parts = []
    for c, v in sorted(vals.items()):
        mib = v / 1024.0 * (2.0 if c == "FETCH_SIZE" else 1.0)
        parts.append("%s%s %.1f MiB" % (c, "x2" if c == "FETCH_SIZE" else "", mib))
    print("%4d %-60s %s" % (i, name[:60], "  ".join(parts)))
