#!/usr/bin/env python3
"""Prints a per-kernel timeline (start offset, duration, gap to previous kernel) from a rocprofv3
--kernel-trace CSV; used to see where a tick's time goes between the solver kernels and the RCCL gather.
usage: trace_timeline.py <kernel_trace.csv> [first_row] [rows]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else max(0, len(rows) - 40)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 40
t0 = int(rows[first]["Start_Timestamp"])
prev_end = None
for r in rows[first:first + n]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f} us  gap {gap:8.1f} us  q{r.get('Queue_Id', '?'):>3}  {r['Kernel_Name'][:70]}")
    prev_end = max(prev_end or 0, e)
