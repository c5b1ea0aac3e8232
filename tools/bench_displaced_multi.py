#!/usr/bin/env python3
"""Multi-GPU displaced-loop benchmark (BASELINE.json configs[2]: 48^3 x 96 on a 1x1x2x4 process grid, +-4 directions,
lengths 1..3) through the C++ driver with halos over torch.distributed (nccl = RCCL over xGMI).

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_displaced_multi.py \\
         --grid 1 1 2 4 --local-lattice 48 48 24 24 --nev 400
MUGIQ_BENCH_BACKEND=gloo rehearses the path on fewer GPUs (device buffers staged through the host; ranks share GPUs).
Synthetic inputs: per-rank random unit-norm eigenvectors and random U(3) links (borders filled from the neighbours by
the driver's setup path); throughput only -- parity of the decomposed run is covered by tests/test_gpu_driver.py.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

ap = argparse.ArgumentParser()
ap.add_argument("--grid", type=int, nargs=4, default=[1, 1, 2, 4])
ap.add_argument("--local-lattice", type=int, nargs=4, default=[48, 48, 24, 24])
ap.add_argument("--nev", type=int, default=100)
ap.add_argument("--precision", type=int, default=8)
ap.add_argument("--entries", default="+x:1,3;-x:1,3;+y:1,3;-y:1,3;+z:1,3;-z:1,3;+t:1,3;-t:1,3")
ap.add_argument("--nmom", type=int, default=19)
ap.add_argument("--reps", type=int, default=2)
a = ap.parse_args()

backend = os.environ.get("MUGIQ_BENCH_BACKEND", "nccl")
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
local_rank = int(os.environ.get("LOCAL_RANK", "0"))
dev = local_rank % torch.cuda.device_count()
torch.cuda.set_device(dev)
device = torch.device("cuda", dev)
if backend == "nccl":
    dist.init_process_group("nccl", device_id=device)
else:
    dist.init_process_group(backend)
import mugiq_amd as hip  # noqa: E402

comm = hip.GridComm(a.grid, device=device)
X = tuple(a.local_lattice)
V = int(np.prod(X))
vcb = V // 2
cdt = torch.complex128 if a.precision == 8 else torch.complex64
per = 24 * vcb
big = torch.empty(a.nev * per, dtype=cdt, device=device)
fields = []
for n in range(a.nev):
    w = torch.complex(torch.randn(per, dtype=torch.float64, device=device), torch.randn(per, dtype=torch.float64, device=device))
    w /= torch.linalg.vector_norm(w)
    big[n * per:(n + 1) * per] = w.to(cdt)
    fields.append(hip.SpinorField(X, a.precision, 2, data=big[n * per:(n + 1) * per]))
    del w
# local U(3) links in QDP host order -> extended device field with neighbour-filled borders (Displace's setup path)
rng = np.random.default_rng(100 + rank)
m = rng.standard_normal((4, V, 3, 3)) + 1j * rng.standard_normal((4, V, 3, 3))
q, _ = np.linalg.qr(m)
qdp = [np.ascontiguousarray(q[d].reshape(V, 9)).view(np.float64).reshape(-1).copy() for d in range(4)]
R = [2 * comm.comm_dim_partitioned(d) for d in range(4)]
gauge = hip.GaugeField(X, R, a.precision).set_from_qdp_host(qdp, comm)
moms = [(x, y, z) for x in range(-2, 3) for y in range(-2, 3) for z in range(-2, 3) if x * x + y * y + z * z <= 2][:a.nmom]
sig = 0.01 + 0.002 * np.arange(a.nev)
times = []
for r in range(a.reps):
    prm = hip.MugiqLoopParam(gauge=gauge, doMomProj=True, momMatrix=[list(p) for p in moms], Nmom=len(moms), FTSign=-1)
    prm.set_displace_entry_string(a.entries)
    loop = hip.Loop_Mugiq(prm, fields, sig, comm)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop.computeCoarseLoop()
    torch.cuda.synchronize()
    dist.barrier()
    times.append(time.perf_counter() - t0)
    nLoop = loop.nLoop
    loop.close()
t = torch.tensor([min(times)], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
if rank == 0:
    tot_sites = V * world
    print(json.dumps({"grid": a.grid, "local": list(X), "nev": a.nev, "nLoop": nLoop, "backend": backend, "n_ranks": world,
                      "seconds": float(t[0]), "sites_per_s_all_slots": tot_sites / float(t[0]),
                      "site_evec_slots_per_s": tot_sites * a.nev * nLoop / float(t[0])}))
dist.destroy_process_group()
