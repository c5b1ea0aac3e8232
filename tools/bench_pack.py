#!/usr/bin/env python3
"""How fast pack_layers_kernel packs the eigenvector halo of configs[2] on its own (48.48.24.24 fp64, N_ev eigenvectors, 3 layers):
bytes read + written over the kernel time, per axis."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mugiq_amd as hip  # noqa: E402
from bench import make_evecs  # noqa: E402

X, nev, layers = (48, 48, 24, 24), int(sys.argv[1]) if len(sys.argv) > 1 else 400, 3
dev = torch.device("cuda", 0)
_, f = make_evecs(hip, X, nev, 8, 2, dev, seed=1)
out = {}
for dim in (2, 3):
    face = f[0].face_cb(dim)
    buf = torch.empty(nev * layers * 24 * face, dtype=torch.complex128, device=dev)
    ms = []
    for r in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.packFaceLayers(buf, f, dim, 0, layers)
        e1.record()
        torch.cuda.synchronize()
        ms.append(e0.elapsed_time(e1))
    b = buf.numel() * 16
    out["xyzt"[dim]] = {"ms": min(ms[1:]), "packed_GB": b / 1e9, "read_plus_written_GBps": 2 * b / (min(ms[1:]) * 1e-3) / 1e9}
print(json.dumps(out))
