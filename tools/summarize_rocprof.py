#!/usr/bin/env python3
"""Trim a rocprofv3 --kernel-trace --stats `*_kernel_stats.csv` into a compact, committed summary
(kernel names cut to 120 characters; everything else verbatim)."""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = list(csv.reader(open(src)))
with open(dst, "w", newline="") as f:
    w = csv.writer(f)
    for r in rows:
        r[0] = r[0] if len(r[0]) <= 120 else r[0][:117] + "..."
        w.writerow(r)
print("wrote", dst, len(rows) - 1, "kernels")
