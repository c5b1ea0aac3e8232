#!/usr/bin/env python3
"""Keep what the judged numbers come from out of a rocprofv3 counter_collection.csv: the rows of the mugiq:: kernels, and the columns
Kernel_Name, Grid_Size, Workgroup_Size, Counter_Name, Counter_Value (the torch kernels' template names make the raw file 10 MB).
usage: slim_pmc_csv.py <counter_collection.csv> <out.csv>"""
import csv
import sys

keep = ["Kernel_Name", "Grid_Size", "Workgroup_Size", "Counter_Name", "Counter_Value"]
with open(sys.argv[1]) as f, open(sys.argv[2], "w", newline="") as g:
    r = csv.DictReader(f)
    w = csv.writer(g)
    w.writerow(keep)
    for row in r:
        if "mugiq::" in row["Kernel_Name"]:
            w.writerow([row[k].split("(")[0] if k == "Kernel_Name" else row[k] for k in keep])
