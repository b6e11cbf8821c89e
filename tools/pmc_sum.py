"""Average a rocprofv3 --pmc counter per (kernel, grid) from the *_counter_collection.csv files under a directory.

    python tools/pmc_sum.py gpurun_out/r02/pmc_fetch [name-filter]
"""
import collections
import csv
import glob
import re
import sys

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.OrderedDict()
for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"])[:64]
        if flt and flt not in name:
            continue
        key = (r["Counter_Name"], name, r.get("Grid_Size", "?"))
        a = agg.setdefault(key, [0, 0.0])
        a[0] += 1
        a[1] += float(r["Counter_Value"])
for (ctr, name, grid), (n, tot) in agg.items():
    print(f"{ctr:12s} {name:64s} grid {grid:>9s} launches {n:4d} avg {tot / n:14.1f}")
