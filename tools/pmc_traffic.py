#!/usr/bin/env python3
"""HBM traffic of one step-kernel launch from two rocprofv3 PMC passes -> profiles/traffic_latest.json (read by bench.py).
usage: pmc_traffic.py DIR_FETCH DIR_WRITE ENVS SOURCE_LABEL
FETCH_SIZE / WRITE_SIZE are in KiB-like units of 1024 B; on gfx950 FETCH_SIZE under-reports wide coalesced reads by 2x
(MI355X_MICROARCH.md, HBM): the read figure is doubled, which makes it an UPPER bound for this kernel's 4-B/lane accesses."""
import collections
import csv
import glob
import json
import os
import sys


def avg(d, counter, kernel="vnl_step_kernel"):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == counter:
                acc[counter][0] += float(r["Counter_Value"])
                acc[counter][1] += 1
    s, n = acc[counter]
    return s / max(n, 1)


rd, wr, envs, label = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
fetch_kb, write_kb = avg(rd, "FETCH_SIZE"), avg(wr, "WRITE_SIZE")
out = dict(envs=envs, fetch_kb=fetch_kb, write_kb=write_kb, bytes_per_launch=(2.0 * fetch_kb + write_kb) * 1024.0,
           source=label, note="FETCH_SIZE x2 (gfx950 read correction, upper bound for 4-B/lane accesses) + WRITE_SIZE, per launch")
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# fingerprint of the kernel sources the figure was taken with: bench.py drops the figure when the sources have changed since
import hashlib  # noqa: E402

h = hashlib.sha256()
for f in ("vnl_body.h", "vnl_lib.hip", "vnl_types.h"):
    h.update(open(os.path.join(root, "vnl-brax-imitation_amd", "csrc", f), "rb").read())
out["kernel_sources_sha256"] = h.hexdigest()
# gpurun_out/ travels back from the GPU box; profiles/ is what bench.py reads and what gets committed: write both
for d in ("gpurun_out", "profiles"):
    os.makedirs(os.path.join(root, d), exist_ok=True)
    json.dump(out, open(os.path.join(root, d, "traffic_latest.json"), "w"), indent=1)
print(json.dumps(out))
