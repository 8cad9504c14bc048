#!/usr/bin/env python3
"""kernel_trace.csv of rocprofv3 -> a timeline of the last `tail` dispatches: queue, start and end relative to the first one, duration,
and how much of it ran beside the dispatch before it (overlapped launches on two streams).
   python scripts/trace_timeline.py <dir with *_kernel_trace.csv> [tail]"""
import csv
import glob
import os
import re
import sys

d, tail = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 24
f = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))[-tail:]
short = lambda n: (re.search(r"(\w+_kernel(?:_w\d)?)", n) or re.search(r"(\w+)", n)).group(1)
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
print("%-34s %6s %10s %10s %9s %9s" % ("kernel", "queue", "start us", "end us", "dur us", "beside prev"))
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ov = max(0, min(e, prev_end) - s) / 1e3 if prev_end is not None else 0.0
    print("%-34s %6s %10.1f %10.1f %9.1f %9.1f" % (short(r["Kernel_Name"])[:34], r.get("Queue_Id", "?"), (s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, ov))
    prev_end = e if prev_end is None else max(prev_end, e)
print("span %.1f us for %d dispatches" % ((max(int(r["End_Timestamp"]) for r in rows) - t0) / 1e3, len(rows)))
