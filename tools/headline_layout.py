#!/usr/bin/env python3
"""Placement sensitivity of the headline kernel (loop_contract_kernel, 32^4 fp64 N_ev = 200): ONE configuration per process, so
that every run gets its own allocations / physical placement.  Prints one JSON line: per-launch event times (min / median / max)
and, for single-allocation modes, the non-temporal read probe over the same buffer.

  --mode one       all eigenvectors in ONE allocation, vector n at n * per (bench.py's layout)
  --mode separate  one allocation per eigenvector (what separately created QUDA fields look like)
  --mode stagger   one allocation, vector n at n * (per + stagger) -- bases move off the 192 MiB grid
  --pad            QUDA pad (plane stride = volumeCB + pad complex)
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mugiq_amd as hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lattice", type=int, nargs=4, default=[32, 32, 32, 32])
ap.add_argument("--nev", type=int, default=200)
ap.add_argument("--mode", default="one", choices=["one", "separate", "stagger"])
ap.add_argument("--stagger-bytes", type=int, default=4096)
ap.add_argument("--pad", type=int, default=0)
ap.add_argument("--steps", type=int, default=30)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--tune", default="")
ap.add_argument("--label", default="")
a = ap.parse_args()
if a.tune:
    os.environ["MUGIQ_HIP_CONTRACT_TUNE"] = a.tune
X = tuple(a.lattice)
V = int(np.prod(X))
vcb = V // 2
per = 2 * 12 * (vcb + a.pad)
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(777)
big = None
if a.mode == "separate":
    bufs = [torch.empty(per, dtype=torch.complex128, device=dev) for _ in range(a.nev)]
else:
    st = (a.stagger_bytes // 16) if a.mode == "stagger" else 0
    big = torch.empty(a.nev * (per + st), dtype=torch.complex128, device=dev)
    bufs = [big[n * (per + st):n * (per + st) + per] for n in range(a.nev)]
fields = []
for b in bufs:
    b.copy_(torch.complex(torch.randn(per, dtype=torch.float64, device=dev, generator=g),
                          torch.randn(per, dtype=torch.float64, device=dev, generator=g)) / np.sqrt(24.0 * V))
    fields.append(hip.SpinorField(X, 8, 2, pad=a.pad, data=b))
sig = 0.01 + 0.002 * np.arange(a.nev)
loop = torch.zeros(16 * V, dtype=torch.complex128, device=dev)
# launches queued back to back like bench.py's timed loop (the host runs ahead of the device: with a synchronisation per launch the
# ~0.3 ms Python spends building 200 descriptors would sit between the two events of every launch)
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(a.warmup + a.steps)]
for e0, e1 in ev:
    loop.zero_()
    e0.record()
    hip.performLoopContractionBatched(loop, fields, fields, sig)
    e1.record()
torch.cuda.synchronize()
ms = [e0.elapsed_time(e1) for e0, e1 in ev[a.warmup:]]
ms = np.array(ms)
alg = V * (a.nev * 192 + 256)
out = {"label": a.label, "mode": a.mode, "pad": a.pad, "stagger_bytes": a.stagger_bytes if a.mode == "stagger" else 0, "tune": a.tune,
       "min_ms": float(ms.min()), "median_ms": float(np.median(ms)), "max_ms": float(ms.max()), "GBps_median": alg / np.median(ms) / 1e6,
       "base_mod_2MiB": [int(b.data_ptr() % (2 << 20)) for b in bufs[:3]], "base_ptr": hex(bufs[0].data_ptr()), "loop_ptr": hex(loop.data_ptr())}
if big is not None:
    pm = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.probeReadBandwidth(big, 1)
        e1.record()
        torch.cuda.synchronize()
        pm.append(e0.elapsed_time(e1))
    out["probe_nt_GBps"] = big.numel() * 16 / np.median(pm[1:]) / 1e6
print(json.dumps(out), flush=True)
