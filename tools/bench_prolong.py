#!/usr/bin/env python3
"""Time prolongateEvecs (coarse -> fine for all eigenvectors) at configs[4] size: 32^4 fp64, 4^4 aggregates, n_vec 24, N_ev 200.
usage: bench_prolong.py [nev]   (MUGIQ_HIP_PROLONG_MFMA=0 selects the vector kernel)"""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mugiq_amd as hip  # noqa: E402

nev = int(sys.argv[1]) if len(sys.argv) > 1 else 200
X, bs, nvec = (32, 32, 32, 32), (4, 4, 4, 4), 24
V = int(np.prod(X))
T = hip.Transfer(X, nvec, bs, 2, 8)
g = torch.Generator(device="cuda").manual_seed(4)
T.V.copy_(torch.complex(torch.randn(T.V.numel(), dtype=torch.float64, device="cuda", generator=g),
                        torch.randn(T.V.numel(), dtype=torch.float64, device="cuda", generator=g)) / np.sqrt(24.0 * nvec))
cf = []
for n in range(nev):
    c = hip.CoarseField(T.Xc, nvec, 8)
    c.data.copy_(torch.complex(torch.randn(c.data.numel(), dtype=torch.float64, device="cuda", generator=g),
                               torch.randn(c.data.numel(), dtype=torch.float64, device="cuda", generator=g)))
    cf.append(c)
big = torch.empty(nev * 24 * (V // 2), dtype=torch.complex128, device="cuda")
ff = [hip.SpinorField(X, 8, 2, data=big[n * 24 * (V // 2):(n + 1) * 24 * (V // 2)]) for n in range(nev)]
ms = []
for r in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    hip.prolongateEvecs(ff, cf, T)
    e1.record()
    torch.cuda.synchronize()
    ms.append(e0.elapsed_time(e1))
best = min(ms[1:])
flops = 8.0 * 12 * nvec * V * nev
print(json.dumps({"workload": "prolongateEvecs 32^4 fp64 n_vec 24 4^4 aggregates N_ev %d" % nev, "mfma": os.environ.get("MUGIQ_HIP_PROLONG_MFMA", "1"),
                  "ms": best, "all_ms": ms, "TFLOPs": flops / best / 1e9, "write_GBps": nev * V * 192 / best / 1e6}))
