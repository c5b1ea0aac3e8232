#!/usr/bin/env python3
"""Secondary benchmark: MG coarse path (BASELINE.json configs[4]): 32^4 fp64, 4^4 aggregates, n_vec = 24, N_ev
coarse eigenvectors -> (a) batched prolongation to fine vectors, (b) fused prolong + ultra-local contraction."""
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
ap.add_argument("--nvec", type=int, default=24)
ap.add_argument("--precision", type=int, default=8)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--write-fine", action="store_true")
a = ap.parse_args()
X = tuple(a.lattice)
V = int(np.prod(X))
cdt = torch.complex128 if a.precision == 8 else torch.complex64
T = hip.Transfer(X, a.nvec, (4, 4, 4, 4), 2, a.precision)
T.V.copy_(torch.complex(torch.randn(T.V.numel(), dtype=torch.float64, device="cuda"), torch.randn(T.V.numel(), dtype=torch.float64, device="cuda")).to(cdt) / np.sqrt(24.0 * a.nvec))
cf = []
for n in range(a.nev):
    c = hip.CoarseField(T.Xc, a.nvec, a.precision)
    c.data.copy_(torch.complex(torch.randn(c.data.numel(), dtype=torch.float64, device="cuda"), torch.randn(c.data.numel(), dtype=torch.float64, device="cuda")).to(cdt))
    cf.append(c)
sig = 0.01 + 0.002 * np.arange(a.nev)
loop = torch.zeros(16 * V, dtype=cdt, device="cuda")
res = {}


def timeit(fn):
    ts = []
    for r in range(a.reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[1:]))


flops = 8.0 * 12 * a.nvec * V * a.nev                     # complex MACs of P
ms = timeit(lambda: (loop.zero_(), hip.prolongateContractBatched(loop, cf, sig, T)))
res["fused_prolong_contract"] = {"ms": ms, "sites_per_s": V / ms * 1e3, "plan": os.environ.get("MUGIQ_HIP_MG_PLAN", "coarse"),
                                 # flops of the per-eigenvector algorithm divided by the time: comparable across plans, not a hardware rate
                                 "per_eigenvector_algorithm_TFLOPs_equivalent": flops / ms / 1e9}
if a.write_fine:
    big = torch.empty(a.nev * 24 * (V // 2), dtype=cdt, device="cuda")
    ff = [hip.SpinorField(X, a.precision, 2, data=big[n * 24 * (V // 2):(n + 1) * 24 * (V // 2)]) for n in range(a.nev)]
    ms = timeit(lambda: hip.prolongateEvecs(ff, cf, T))
    res["prolongate_to_fine"] = {"ms": ms, "prolong_TFLOPs": flops / ms / 1e9, "write_GBps": a.nev * 24 * V * a.precision / ms / 1e6}
    ms2 = timeit(lambda: (loop.zero_(), hip.performLoopContractionBatched(loop, ff, ff, sig)))
    res["then_contract"] = {"ms": ms2}
print(json.dumps({"lattice": X, "nev": a.nev, "nvec": a.nvec, "precision": a.precision, "results": res}))
