"""Sum a rocprofv3 --pmc counter per kernel name from the *_counter_collection.csv files under a directory."""
import csv
import glob
import sys
from collections import defaultdict
tot = defaultdict(lambda: [0.0, 0])
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(fn)):
        k = (row["Kernel_Name"][:60], row["Counter_Name"])
        tot[k][0] += float(row["Counter_Value"])
        tot[k][1] += 1
seen = {}
for (kn, cn), (v, n) in sorted(tot.items()):
    # rows per dispatch = one per XCD/instance dimension; report the per-dispatch sum
    disp = len({1})
    print("%-62s %-12s total %.4e over %d rows" % (kn, cn, v, n))
