#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc counter_collection.csv: mean of every counter per kernel (name substring filter).
usage: pmc_kernel.py <counter_collection.csv> <kernel-substring>"""
import csv
import collections
import sys

acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print("%-32s %.4g  (n=%d)" % (k, sum(acc[k]) / len(acc[k]), len(acc[k])))
