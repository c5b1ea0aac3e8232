#!/usr/bin/env python3
"""Phase table of one driver job (mugiq_hip_loop_set_profiling): where the time of a SMALL per-rank job goes -- what a rank of the
strong-scaling leg runs at 8 GPUs is configs[2]'s local lattice with 48 eigenvectors, and fixed costs (path-link fields, axial
gauges, reflections, projection) weigh 8 x more there than at N_ev 400.  One GPU; --force z,t runs the partitioned plan."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402  (the synthetic inputs of bench.py)
import mugiq_amd as hip  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--lattice", type=int, nargs=4, default=[48, 48, 24, 24])
ap.add_argument("--nev", type=int, default=48)
ap.add_argument("--entries", default="+x:1,3;-x:1,3;+y:1,3;-y:1,3;+z:1,3;-z:1,3;+t:1,3;-t:1,3")
ap.add_argument("--force", type=int, nargs=4, default=[0, 0, 0, 0])
ap.add_argument("--p2max", type=int, default=9)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
X = tuple(a.lattice)
dev = torch.device("cuda", 0)
comm = None
if any(a.force):
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29655")
    dist.init_process_group("gloo", rank=0, world_size=1)
    comm = hip.GridComm((1, 1, 1, 1), device=dev, force_partitioned=tuple(a.force))
big, fields = bench.make_evecs(hip, X, a.nev, 8, 2, dev, 11)
sig = 0.01 + 0.002 * np.arange(a.nev)
gauge = bench.make_gauge(hip, X, 8, dev, 12, comm=comm)
torch.cuda.empty_cache()
r = int(np.sqrt(a.p2max)) + 1
moms = [[x, y, z] for x in range(-r, r + 1) for y in range(-r, r + 1) for z in range(-r, r + 1) if x * x + y * y + z * z <= a.p2max]
best = None
for rep in range(a.reps):
    prm = hip.MugiqLoopParam(gauge=gauge, calcType=hip.LOOP_CALC_TYPE_OPT_KERNEL, doMomProj=True, momMatrix=moms, Nmom=len(moms), FTSign=-1)
    prm.set_displace_entry_string(a.entries)
    loop = hip.Loop_Mugiq(prm, fields, sig, comm) if comm else hip.Loop_Mugiq(prm, fields, sig)
    loop.setProfiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop.computeCoarseLoop()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ph = loop.phases()
    loop.close()
    if rep > 0 and (best is None or el < best[0]):
        best = (el, ph)
el, ph = best
table = {}
for p in ph:
    key = p["kind"] + ("" if p["entry"] < 0 else "[%d]" % p["entry"])
    table[key] = table.get(key, 0.0) + p["ms"]
print(json.dumps({"lattice": X, "nev": a.nev, "force": a.force, "seconds": el, "phase_ms": {k: round(v, 3) for k, v in table.items()}}))
