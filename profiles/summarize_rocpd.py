#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / average / share.
Usage: python profiles/summarize_rocpd.py <results.db> <out.csv> [--all]
By default torch's data-generation kernels (at::native::*, Cijk_*) are folded into one line."""
import csv
import sqlite3
import sys

db, out = sys.argv[1], sys.argv[2]
keep_all = "--all" in sys.argv
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc"))
other = [0, 0.0, 0.0]
with open(out, "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["kernel", "calls", "total_us", "avg_us", "percent_of_gpu_time"])
    for name, calls, tot, avg, pct in rows:
        short = name.split("(")[0]
        if not keep_all and (short.startswith("void at::native") or short.startswith("Cijk_") or "at::native" in short):
            other[0] += calls; other[1] += tot; other[2] += pct
            continue
        w.writerow([short[:120], calls, round(tot, 1), round(avg, 3), round(pct, 3)])
    if other[0]:
        w.writerow(["(torch kernels: synthetic data generation, not part of the path)", other[0], round(other[1], 1), round(other[1] / other[0], 3), round(other[2], 3)])
print(open(out).read())
