#!/usr/bin/env python3
"""Print the last N kernels of a rocprofv3 --kernel-trace csv as a timeline (start offset, duration, queue, name)."""
import csv
import sys

path, n = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40
skip = sys.argv[3].split(",") if len(sys.argv) > 3 else []
rows = [r for r in csv.DictReader(open(path)) if not any(s in r["Kernel_Name"] for s in skip)]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"{(s - t0) / 1e3:9.1f} us  {(e - s) / 1e3:7.1f} us  q{r.get('Queue_Id', '?'):3s} {r['Kernel_Name'][:70]}")
