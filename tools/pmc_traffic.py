#!/usr/bin/env python3
"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate passes, TCC slots do not fit both) into
HBM bytes per launch, corrected as /opt/skills/guides/MI355X_MICROARCH.md (section HBM) prescribes for gfx950:
FETCH_SIZE and WRITE_SIZE are in KiB; FETCH_SIZE reports exactly 1/2 of the bytes of a wide (16 B/lane)
coalesced streaming read -> doubled; WRITE_SIZE is exact for 16-B-per-lane streaming stores.

usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel-substring> <workload> <out.json> [grid-size]
(grid-size: only launches of that many work-items, when the run holds the kernel at several sizes)
"""
import csv
import hashlib
import re
import json
import os
import sys


def _normalised(path):
    """source text without comments and with whitespace collapsed: a comment edit does not invalidate a PMC measurement"""
    t = open(path, "r", errors="replace").read()
    t = re.sub(r"/\*.*?\*/", " ", t, flags=re.S)
    t = re.sub(r"//[^\n]*", " ", t)
    return re.sub(r"\s+", " ", t).encode()


def source_fingerprint():
    """the same fingerprint bench.py computes: sha1 over the sources of the headline kernel"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    h = hashlib.sha1()
    for f in ("contract.hip", "internal.h"):
        h.update(_normalised(os.path.join(root, "mugiq_amd", "csrc", f)))
    return h.hexdigest()[:12]


GRID = sys.argv[6] if len(sys.argv) > 6 else None


def mean_counter(path, kernel, name):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if kernel in r["Kernel_Name"] and r["Counter_Name"] == name and (GRID is None or r["Grid_Size"] == GRID)]
    return sum(vals) / len(vals), len(vals)


fetch_csv, write_csv, kernel, workload, out = sys.argv[1:6]
f, nf = mean_counter(fetch_csv, kernel, "FETCH_SIZE")
w, nw = mean_counter(write_csv, kernel, "WRITE_SIZE")
res = {
    "workload": workload, "kernel": kernel, "launches_averaged": [nf, nw], "source_fingerprint": source_fingerprint(),
    "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB_raw": w,
    "fetch_bytes_corrected_x2": 2.0 * f * 1024.0, "write_bytes": w * 1024.0,
    "hbm_bytes_per_launch": 2.0 * f * 1024.0 + w * 1024.0,
    "correction": "gfx950: FETCH_SIZE counts 128-B requests at 64 B -> x2 for 16 B/lane coalesced streams (MI355X_MICROARCH.md, HBM)",
}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
