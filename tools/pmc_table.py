#!/usr/bin/env python
"""Aggregate rocprofv3 `--pmc … --output-format csv` counter_collection.csv files into one
kernel x counter table (mean per dispatch, last N dispatches of every kernel)."""
import csv, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            name = r["Kernel_Name"][:60] + "|" + r["Grid_Size"]
            rows[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
counters = sorted({c for k in rows.values() for c in k})
print("kernel|grid," + ",".join(counters))
for k, d in rows.items():
    print('"%s",' % k + ",".join("%.4g" % (sum(d[c][-3:]) / len(d[c][-3:])) if d.get(c) else "" for c in counters))
