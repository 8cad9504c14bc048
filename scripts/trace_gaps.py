#!/usr/bin/env python3
"""kernel_trace.csv of rocprofv3 -> what the loop's time is made of: per kernel name the average duration and the average idle
gap in FRONT of it (previous kernel's end to this one's start), over the last `tail` dispatches.
   python scripts/trace_gaps.py <dir with *_kernel_trace.csv> [tail]"""
import collections
import csv
import glob
import os
import re
import sys

d, tail = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 80
f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-tail:]
short = lambda n: (re.search(r"(\w+_kernel(?:_w\d)?(?:<[^>]*>)?)", n) or re.search(r"(\w+)", n)).group(1)
dur, gap = collections.defaultdict(list), collections.defaultdict(list)
prev_end = None
for r in rows:
    s, e, k = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]) + " g%s" % r["Grid_Size_X"]
    dur[k].append((e - s) / 1e3)
    if prev_end is not None:
        gap[k].append((s - prev_end) / 1e3)
    prev_end = e
span = (int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])) / 1e3
print("%d dispatches over %.1f us" % (len(rows), span))
for k in dur:
    g = gap.get(k, [0.0])
    print("%-60s n=%3d  avg %9.2f us   gap in front: avg %7.2f  min %7.2f  max %7.2f" % (k, len(dur[k]), sum(dur[k]) / len(dur[k]), sum(g) / len(g), min(g), max(g)))
print("sum of kernel time %.1f us, idle %.1f us" % (sum(sum(v) for v in dur.values()), span - sum(sum(v) for v in dur.values())))
