#!/usr/bin/env python3
"""Momentum-projection path (reorder + skinny complex GEMM) at BASELINE config sizes."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mugiq_amd as hip  # noqa: E402


def timeit(fn, reps=3):
    ts = []
    for r in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts[1:]))


for name, X, nLoop, Nmom in [("cfg2 32^4 nLoop=1 p2<=9", (32, 32, 32, 32), 1, 123), ("cfg3 48x48x24x24 nLoop=25 p2<=9", (48, 48, 24, 24), 25, 123),
                             ("cfg3 nLoop=25 p2<=2", (48, 48, 24, 24), 25, 19)]:
    V = int(np.prod(X))
    nData = 16 * nLoop
    locV3 = X[0] * X[1] * X[2]
    pos = torch.randn(nData * V, dtype=torch.complex128, device="cuda")
    mp = torch.empty_like(pos)
    moms = [(x, y, z) for x in range(-3, 4) for y in range(-3, 4) for z in range(-3, 4) if x * x + y * y + z * z <= 9][:Nmom]
    ph = torch.empty(locV3 * Nmom, dtype=torch.complex128, device="cuda")
    hip.createPhaseMatrixGPU(ph, moms, locV3, Nmom, 1, X, X)
    mom = torch.empty(X[3] * nData * Nmom, dtype=torch.complex128, device="cuda")
    t_conv = timeit(lambda: hip.convertIdxOrder_mapGamma(mp, pos, nData, nLoop, 2, V // 2, X))
    t_gemm = timeit(lambda: hip.momentumProjection(mom, mp, ph, X[3], nData, locV3, Nmom))
    t_sep = timeit(lambda: hip.momentumProjectionSeparable(mom, mp, moms, 1, X, X, X[3], nData))
    M, K, N = X[3] * nData, locV3, Nmom
    print(json.dumps({"case": name, "M": M, "K": K, "N": N, "convert_ms": t_conv, "convert_GBps": 2 * 16 * nData * V / t_conv / 1e6,
                      "gemm_ms": t_gemm, "gemm_TFLOPs": 8.0 * M * K * N / t_gemm / 1e9, "gemm_A_GBps_once": 16.0 * M * K / t_gemm / 1e6,
                      "separable_ms": t_sep, "separable_A_GBps_once": 16.0 * M * K / t_sep / 1e6}))
    del pos, mp, ph, mom
