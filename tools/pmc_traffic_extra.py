#!/usr/bin/env python3
"""HBM bytes per launch of every mugiq kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes) of
   rocprofv3 --kernel-trace --pmc <counter> --output-format csv -- python3 bench.py --steps 3 --warmup 1 --extra displaced,mg --no-cpu-baseline
corrected as MI355X_MICROARCH.md (HBM section) prescribes for gfx950 (KiB units; FETCH_SIZE x 2 for 16-B-per-lane streams).
bench.py attaches these figures to the roofline blocks of its extra legs when the library sources are unchanged.

usage: pmc_traffic_extra.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [displaced-nev]
(displaced-nev: the --displaced-nev of the profiled bench.py run, default 400; the figures of the displaced leg are attached only
to a run with the same number of eigenvectors)
"""
import collections
import csv
import hashlib
import re
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _normalised(path):
    """source text without comments and with whitespace collapsed: a comment edit does not invalidate a PMC measurement"""
    t = open(path, "r", errors="replace").read()
    t = re.sub(r"/\*.*?\*/", " ", t, flags=re.S)
    t = re.sub(r"//[^\n]*", " ", t)
    return re.sub(r"\s+", " ", t).encode()


def library_fingerprint():
    """sha1 over the kernel sources of libmugiq_hip.so (the same function lives in bench.py)"""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "mugiq_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith(".hip") or f == "internal.h":       # the kernels (the host-side driver does not change a kernel's traffic)
            h.update(_normalised(os.path.join(d, f)))
    return h.hexdigest()[:12]


def load(path, name):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name and "mugiq::" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0], int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


if __name__ == "__main__":
    f = load(sys.argv[1], "FETCH_SIZE")
    w = load(sys.argv[2], "WRITE_SIZE")
    kernels = []
    for (name, grid), (fv, n) in f.items():
        wv = w.get((name, grid), (0.0, 0))[0]
        kernels.append({"name": name, "grid": grid, "launches_averaged": n, "fetch_bytes_corrected_x2": 2.0 * fv * 1024.0,
                        "write_bytes": wv * 1024.0, "hbm_bytes_per_launch": 2.0 * fv * 1024.0 + wv * 1024.0})
    json.dump({"library_fingerprint": library_fingerprint(), "displaced_nev": int(sys.argv[4]) if len(sys.argv) > 4 else 400,
               "correction": "gfx950: FETCH_SIZE / WRITE_SIZE in KiB; FETCH_SIZE x 2 for 16 B/lane coalesced streams (MI355X_MICROARCH.md, HBM)",
               "kernels": kernels}, open(sys.argv[3], "w"), indent=1)
    print("wrote %s (%d kernels, library %s)" % (sys.argv[3], len(kernels), library_fingerprint()))
