#!/usr/bin/env python3
"""Print the kernels of ONE step from a rocprofv3 --kernel-trace CSV as a timeline (start, duration).
usage: tools/trace_timeline.py <kernel_trace.csv> [anchor-kernel-substring] [count]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
anchor = sys.argv[2] if len(sys.argv) > 2 else "gather"
count = int(sys.argv[3]) if len(sys.argv) > 3 else 30
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# start from the LAST-but-two occurrence group of the anchor so that warm-up is over
idx = [i for i, r in enumerate(rows) if anchor in r["Kernel_Name"]]
i0 = idx[len(idx) * 2 // 3] if idx else 0
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:i0 + count]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%9.1f us  +%8.1f us  %s" % ((s - t0) / 1e3, (e - s) / 1e3, r["Kernel_Name"][:70]))
