#!/usr/bin/env python3
"""Average the per-dispatch PMC values of rocprofv3 `*_counter_collection.csv` files per (kernel, counter).
usage: pmc_summary.py DIR [kernel-substring]"""
import collections
import csv
import glob
import sys

acc = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        if len(sys.argv) > 2 and sys.argv[2] not in k:
            continue
        a = acc[(k, r["Counter_Name"])]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
for (k, c), (s, n) in sorted(acc.items()):
    print(f"{k:40s} {c:28s} calls={n:5d} avg={s / n:.6g}")
