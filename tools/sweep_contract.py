#!/usr/bin/env python3
"""Sweep the launch variants of loop_contract_kernel (MUGIQ_HIP_CONTRACT_TUNE = "block,depth,nt") on the bench
workload, interleaved rounds in ONE process (cdna_hip_programming.md rule 24), median + min per variant."""
import argparse
import itertools
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mugiq_amd as hip  # noqa: E402
from bench import make_evecs  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lattice", type=int, nargs=4, default=[32, 32, 32, 32])
ap.add_argument("--nev", type=int, default=200)
ap.add_argument("--precision", type=int, default=8)
ap.add_argument("--order", type=int, default=2)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--blocks", default="128,256,512")
ap.add_argument("--depths", default="1,2,3")
ap.add_argument("--nts", default="0,1")
ap.add_argument("--pad", type=int, default=0)
ap.add_argument("--swz", default="0")
a = ap.parse_args()
X = tuple(a.lattice)
V = int(np.prod(X))
big, fields = make_evecs(hip, X, a.nev, a.precision, a.order, torch.device("cuda"), 777, a.pad)
sig = 0.01 + 0.002 * np.arange(a.nev)
loop = torch.zeros(16 * V, dtype=torch.complex128 if a.precision == 8 else torch.complex64, device="cuda")
variants = ["%s,%s,%s,%s" % v for v in itertools.product(a.blocks.split(","), a.depths.split(","), a.nts.split(","), a.swz.split(","))]
times = {v: [] for v in variants}
alg = V * (a.nev * 24 * a.precision + 32 * a.precision)
# launches queued back to back per round (the host runs ahead of the device): with a synchronisation after every launch the time Python
# spends building the field descriptors of a call would sit between its two events
for r in range(a.rounds + 1):
    evs = []
    for v in variants:
        os.environ["MUGIQ_HIP_CONTRACT_TUNE"] = v
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        loop.zero_()
        e0.record()
        hip.performLoopContractionBatched(loop, fields, fields, sig)
        e1.record()
        evs.append((v, e0, e1))
    torch.cuda.synchronize()
    if r > 0:
        for v, e0, e1 in evs:
            times[v].append(e0.elapsed_time(e1))
# calibration: achievable streaming-read bandwidth of this device on the same 40 GB
for nt in (0, 1):
    ts = []
    for r in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.probeReadBandwidth(big, nt)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(json.dumps({"probe_read_nt": nt, "median_ms": float(np.median(ts[1:])), "GBps": big.numel() * big.element_size() / np.median(ts[1:]) / 1e6}))
res = []
for v in variants:
    t = np.array(times[v])
    res.append({"tune": v, "median_ms": float(np.median(t)), "min_ms": float(t.min()), "GBps_median": alg / np.median(t) / 1e6})
res.sort(key=lambda x: x["median_ms"])
for x in res:
    print(json.dumps(x))
