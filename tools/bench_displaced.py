#!/usr/bin/env python3
"""Secondary benchmark: displaced loops (BASELINE.json configs[2] per-GPU shape: 48x48x24x24 local, +-4 dirs,
lengths 1..3) through the C++ driver, fused (OPT) vs reference-sequence (BASIC) plans.  Single GPU, periodic."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mugiq_amd as hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lattice", type=int, nargs=4, default=[48, 48, 24, 24])
ap.add_argument("--nev", type=int, default=100)
ap.add_argument("--precision", type=int, default=8)
ap.add_argument("--order", type=int, default=2)
ap.add_argument("--loop-precision", type=int, default=0, help="8 with --precision 4: fp32 eigenvectors, fp64 loops")
ap.add_argument("--entries", default="+x:1,3;-x:1,3;+y:1,3;-y:1,3;+z:1,3;-z:1,3;+t:1,3;-t:1,3")
ap.add_argument("--plans", default="opt,basic")
ap.add_argument("--reps", type=int, default=2)
ap.add_argument("--ab-tile", action="store_true", help="alternate MUGIQ_HIP_FUSED_TILE=0/1 in one process (interleaved rounds)")
ap.add_argument("--momproj", type=int, default=-1, help="also run the momentum projection for all p^2 <= this (whole pipeline)")
ap.add_argument("--ab-env", default=None, help="NAME=v1,v2,...: alternate an environment knob of the library in one process")
a = ap.parse_args()

X = tuple(a.lattice)
V = int(np.prod(X))
vcb = V // 2
per = 24 * vcb
cdt = torch.complex128 if a.precision == 8 else torch.complex64
big = torch.empty(a.nev * per, dtype=cdt, device="cuda")
fields = []
for n in range(a.nev):
    v = big[n * per:(n + 1) * per]
    w = torch.complex(torch.randn(per, dtype=torch.float64, device="cuda"), torch.randn(per, dtype=torch.float64, device="cuda"))
    w /= torch.linalg.vector_norm(w)
    v.copy_(w.to(cdt))
    fields.append(hip.SpinorField(X, a.precision, a.order, data=v))
# random SU(3) links: QR of Gaussian matrices on the GPU (det phase not fixed: irrelevant for throughput, U(3) is unitary)
g = hip.GaugeField(X, (0, 0, 0, 0), a.precision)
m = torch.complex(torch.randn(4 * 2 * vcb, 3, 3, dtype=torch.float64, device="cuda"), torch.randn(4 * 2 * vcb, 3, 3, dtype=torch.float64, device="cuda"))
def _gs(m):            # Gram-Schmidt on rows, vectorised (torch.linalg.qr cannot batch 10^7 matrices)
    r0 = m[:, 0] / torch.linalg.vector_norm(m[:, 0], dim=-1, keepdim=True)
    r1 = m[:, 1] - (r0.conj() * m[:, 1]).sum(-1, keepdim=True) * r0
    r1 = r1 / torch.linalg.vector_norm(r1, dim=-1, keepdim=True)
    r2 = m[:, 2] - (r0.conj() * m[:, 2]).sum(-1, keepdim=True) * r0
    r2 = r2 - (r1.conj() * r2).sum(-1, keepdim=True) * r1
    r2 = r2 / torch.linalg.vector_norm(r2, dim=-1, keepdim=True)
    return torch.stack([r0, r1, r2], dim=1)


q = _gs(m)
q = q.reshape(4, 2, vcb, 9).permute(1, 0, 3, 2).contiguous()          # [parity][dir][row*3+col][x_cb]
g.data.copy_(q.reshape(-1).to(cdt))
sig = 0.01 + 0.002 * np.arange(a.nev)
B = a.precision
res = {}
if a.ab_env:
    name, vals = a.ab_env.split("=")
    vals = vals.split(",")
    prm = hip.MugiqLoopParam(gauge=g, calcType=hip.LOOP_CALC_TYPE_OPT_KERNEL)
    if a.loop_precision:
        prm.loopPrecision = a.loop_precision
    prm.set_displace_entry_string(a.entries)
    loop = hip.Loop_Mugiq(prm, fields, sig)
    ts = {v: [] for v in vals}
    for r in range(a.reps + 1):
        for v in vals:
            os.environ[name] = v
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loop.computeCoarseLoop()
            torch.cuda.synchronize()
            if r > 0:
                ts[v].append(time.perf_counter() - t0)
    print(json.dumps({"lattice": X, "nev": a.nev, "precision": a.precision, "order": a.order, "entries": a.entries, "knob": name,
                      "median_s": {v: float(np.median(ts[v])) for v in vals}, "min_s": {v: min(ts[v]) for v in vals}}))
    sys.exit(0)
if a.ab_tile:
    prm = hip.MugiqLoopParam(gauge=g, calcType=hip.LOOP_CALC_TYPE_OPT_KERNEL)
    prm.set_displace_entry_string(a.entries)
    loop = hip.Loop_Mugiq(prm, fields, sig)
    ts = {"0": [], "1": [], "2": []}
    for r in range(a.reps + 1):
        for t in ("0", "1", "2"):
            os.environ["MUGIQ_HIP_FUSED_TILE"] = t
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loop.computeCoarseLoop()
            torch.cuda.synchronize()
            if r > 0:
                ts[t].append(time.perf_counter() - t0)
    print(json.dumps({"lattice": X, "nev": a.nev, "entries": a.entries,
                      "streaming_median_s": float(np.median(ts["0"])), "tiled_median_s": float(np.median(ts["1"])),
                      "column_tile_only_median_s": float(np.median(ts["2"])),
                      "streaming_min_s": min(ts["0"]), "tiled_min_s": min(ts["1"]), "column_tile_only_min_s": min(ts["2"]),
                      "streaming_all": ts["0"], "tiled_all": ts["1"], "column_tile_only_all": ts["2"]}))
    sys.exit(0)
for plan in a.plans.split(","):
    prm = hip.MugiqLoopParam(gauge=g, calcType=hip.LOOP_CALC_TYPE_OPT_KERNEL if plan == "opt" else hip.LOOP_CALC_TYPE_BASIC_KERNEL)
    prm.set_displace_entry_string(a.entries)
    if a.momproj >= 0:
        r = int(np.sqrt(a.momproj)) + 1
        moms = [[x, y, z] for x in range(-r, r + 1) for y in range(-r, r + 1) for z in range(-r, r + 1) if x * x + y * y + z * z <= a.momproj]
        prm.doMomProj, prm.momMatrix, prm.Nmom, prm.FTSign = True, moms, len(moms), -1
    loop = hip.Loop_Mugiq(prm, fields, sig)
    times = []
    for r in range(a.reps):
        if a.momproj >= 0 and r > 0:          # performMomentumProjection may run once per Loop_Mugiq (as in the reference)
            loop.close()
            loop = hip.Loop_Mugiq(prm, fields, sig)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop.computeCoarseLoop()
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    nslots = loop.nLoop - 1
    # algorithmic bytes of the whole job in the fused formulation: ultra-local (N_ev*24B + 32B) per site, plus per
    # displaced slot N_ev*24B (shifted vector) + per entry N_ev*24B (v(x), shared by the entry's slots) + 32B out
    alg = V * (a.nev * 24 * B + 32 * B) + V * (nslots * (a.nev * 24 * B + 32 * B + 18 * B) + loop.nDispEntries * a.nev * 24 * B)
    t = min(times)
    res[plan] = {"seconds": t, "sites_per_s_all_slots": V / t, "nLoop": loop.nLoop, "algorithmic_GB": alg / 1e9,
                 "effective_GBps_vs_fused_algorithmic": alg / t / 1e9}
    loop.close()
print(json.dumps({"lattice": X, "nev": a.nev, "entries": a.entries, "results": res}))
