#!/usr/bin/env python3
"""Kernel statistics CSV (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs) from a rocprofv3
rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- python3 ...` writes NAME_results.db on
ROCm 7.2), in the column layout of rocprofv3's own kernel_stats.csv.  usage: rocpd_stats.py in.db out.csv"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else "kernel_name"
rows = db.execute(f"select {name_col}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
                  f"from kernels group by {name_col} order by sum(end - start) desc").fetchall()
total = sum(r[2] for r in rows) or 1
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for n, c, t, a, mn, mx in rows:
        w.writerow([n, c, t, round(a, 1), round(100.0 * t / total, 4), mn, mx])
