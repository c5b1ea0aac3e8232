#!/usr/bin/env python3
"""Time the fused reorder + momentum projection (mugiq_hip_convert_and_project) on a configs[2]-shaped loop buffer:
48.48.24.24, 25 slots, momenta p^2 <= 9; HIP events around the call; optional MUGIQ_HIP_EO_TILES_PER_WG sweep."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mugiq_amd as hip  # noqa: E402

X = tuple(int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (48, 48, 24, 24)))
nLoop = int(sys.argv[5]) if len(sys.argv) > 5 else 25
prec = int(sys.argv[6]) if len(sys.argv) > 6 else 8
V = int(np.prod(X))
nData = 16 * nLoop
cdt = torch.complex128 if prec == 8 else torch.complex64
pos = torch.randn(nData * V, dtype=cdt, device="cuda")
r = 3
moms = [(x, y, z) for x in range(-r, r + 1) for y in range(-r, r + 1) for z in range(-r, r + 1) if x * x + y * y + z * z <= 9]
out = torch.zeros(X[3] * nData * len(moms), dtype=cdt, device="cuda")
res = {}
for knob in (os.environ.get("SWEEP", "default").split(",")):
    if knob != "default":
        os.environ["MUGIQ_HIP_EO_TILES_PER_WG"] = knob
    ts = []
    for i in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.convertAndProject(out, pos, nData, nLoop, moms, -1, X, X)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts[1:]))
    res[knob] = {"ms": ms, "read_once_GBps": nData * V * 2 * prec / ms / 1e6}
print(json.dumps({"lattice": X, "nLoop": nLoop, "precision": prec, "n_mom": len(moms), "results": res}))
