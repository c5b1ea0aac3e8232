#!/usr/bin/env python3
"""Is it the PLACEMENT?  One process, the headline kernel timed on the same 40 GB of data allocated again and again: between two
rounds the buffer is freed (empty_cache: back to the driver) and a spacer of a different size is allocated first, so that the next
buffer lands elsewhere.  Clocks, box, library and process stay the same; only where the bytes live changes."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mugiq_amd as hip  # noqa: E402

X = (32, 32, 32, 32)
V = int(np.prod(X))
nev = 200
per = 24 * (V // 2)
dev = torch.device("cuda")
sig = 0.01 + 0.002 * np.arange(nev)
loop = torch.zeros(16 * V, dtype=torch.complex128, device=dev)
spacers = []
for rnd in range(int(sys.argv[1]) if len(sys.argv) > 1 else 8):
    big = torch.empty(nev * per, dtype=torch.complex128, device=dev)
    big.view(torch.float64).normal_(generator=torch.Generator(device=dev).manual_seed(1))
    fields = [hip.SpinorField(X, 8, 2, data=big[n * per:(n + 1) * per]) for n in range(nev)]
    pm = []
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(15)]
    for e0, e1 in ev:                       # queued back to back: the host-side descriptor build stays off the event pairs
        loop.zero_()
        e0.record()
        hip.performLoopContractionBatched(loop, fields, fields, sig)
        e1.record()
    torch.cuda.synchronize()
    ms = [e0.elapsed_time(e1) for e0, e1 in ev[3:]]
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.probeReadBandwidth(big, 1)
        e1.record()
        torch.cuda.synchronize()
        pm.append(e0.elapsed_time(e1))
    gbs = V * (nev * 192 + 256) / np.median(ms) / 1e6
    pg = big.numel() * 16 / np.median(pm[1:]) / 1e6
    print(json.dumps({"round": rnd, "base_mod_1GiB_MiB": (big.data_ptr() % (1 << 30)) >> 20, "kernel_median_ms": float(np.median(ms)), "GBps": gbs,
                      "probe_GBps": pg, "frac_of_probe": gbs / pg}), flush=True)
    del fields, big
    torch.cuda.empty_cache()
    spacers.append(torch.empty(int(np.random.default_rng(rnd).integers(1, 64)) << 26, dtype=torch.uint8, device=dev))   # 64 MiB .. 4 GiB, kept
