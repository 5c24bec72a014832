#!/usr/bin/env python3
"""Per-kernel mean of rocprofv3 --pmc counters (r_counter_collection.csv files).
usage: summarize_pmc.py dir [dir ...]   -> markdown table"""
import csv
import collections
import glob
import os
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for fn in glob.glob(os.path.join(d, "*counter_collection.csv")):
        for row in csv.DictReader(open(fn)):
            k = row["Kernel_Name"].split("(")[0].replace("void cddp_dev::", "").replace("cddp_dev::", "")
            if not k.startswith("void kload"):      # (the calibration micro-benchmarks are told apart by their template arguments)
                k = k.split("<")[0]
            acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
names = sorted({c for k in acc for c in acc[k]})
print("| kernel | dispatches | " + " | ".join(names) + " |")
print("|---|---|" + "---|" * len(names))
for k in sorted(acc):
    n = max(len(v) for v in acc[k].values())
    cells = []
    for c in names:
        v = acc[k].get(c)
        cells.append("%.4g" % (sum(v) / len(v)) if v else "")
    print("| `%s` | %d | %s |" % (k, n, " | ".join(cells)))
