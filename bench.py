#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native MuGiq loop engine.

Metric (BASELINE.json): loop-trace sites/s, one "site" = all 16 gamma traces summed over all N_ev low modes
at one lattice site, plus achieved HBM GB/s against the 8 TB/s roofline.

Workload at N=1 (BASELINE.json configs[1]): 32^4 Wilson-clover fp64, N_ev=200, ultra-local 16-gamma loop.
A "step" is one pass of the hot path over the whole local lattice: zero the loop slot
(lib/loop_mugiq.cpp:476) + the eigenvector-batched contraction (lib/loop_mugiq.cpp:478-503 folded into one
launch of loop_contract_kernel).  Inputs are synthetic (seeded Gaussian unit-norm eigenvectors, sigma_n =
0.01 + 0.002 n) and resident in HBM before the timed region.

N>1, headline: the lattice is block-partitioned over ranks (T first, then Z); the ultra-local contraction has no
inter-site coupling, so ranks run independently on their local 32^4 block (weak scaling, no data-path
collective); value = total sites / max-over-ranks time.  The line then also carries, at its top level, what the
PARTITIONED configs[2] job did on the same ranks (`partitioned`: sites/s over all 25 slots, halo GB/s per rank, the halo wait
the overlap did not hide), the process grid and the number of ranks RCCL saw (`nccl_ranks`).
`python bench.py --gpus N` without a launcher starts its N ranks itself (children of this process, started before anything
touches the GPU) and relays rank 0's line.

At EVERY N one extra leg runs the SAME global problem (48^3 x 96, all 25 slots, N_ev = --strong-nev), unpartitioned at N = 1 and on the
T-then-Z process grid otherwise, and reports it at the top level (`strong_scaling`): the speedup of the partitioned path from 1 to N GPUs.

After the timed region (not part of `value`; skipped with --no-extra) the other legs of the path are measured too and
reported under `also_measured`, each with its own roofline block, timed by HIP events inside the driver
(mugiq_hip_loop_set_profiling):
  N = 1 : the configs[2] job on its per-GPU lattice at the metric's N_ev = 400 (48.48.24.24, 8 entries x lengths 1..3,
          momentum projection p^2 <= 9; 102 GB of eigenvectors), the same job with the partitioned code path FORCED on z and t
          (configs[2]'s 1x1x2x4 grid seen from one rank, the rank being its own neighbour: packed halos, interior / boundary
          tiles, halo buffers in HBM -- no xGMI), the MG coarse loop of configs[4] (32^4, n_vec 24, N_ev 200), and the
          configs[3] per-GPU ultra-local loop (64.64.32.16, fp32 storage / fp64 loops, N_ev 600).
  N > 1 : the configs[2] job PARTITIONED over the ranks (T first, then Z: 1x1x1x2, 1x1x1x4, 1x1x2x4) through
          Loop_Mugiq + GridComm on nccl (= RCCL over xGMI): eigenvector halos posted ahead on a halo stream, interior tiles
          before the wait, boundary tiles after; halo bytes, transfer time and the interior / boundary split are reported.
          N = 4 and 8 run the metric's own 48^3 x 96, N_ev = 400 (204 / 102 GB of eigenvectors per GPU).

Usage: python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run)
"""
import argparse
import hashlib
import re
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
CLOCK_NOTE = ("flops = the reference's arithmetic per site, slot and eigenvector (W_k psi: 36 complex FMAs + colour-traced outer product: 48 = 672 flop); "
              "peak = 78.6 TFLOP/s at 2.4 GHz.  The kernel itself (csrc/fused_mfma.hip) rotates the eigenvectors into an axial gauge once per staged "
              "position instead of applying W_k per slot and runs the outer products on the fp64 matrix pipe: profiles/r04_mfma_tile.txt")
FP64_VECTOR_PEAK_TFLOPS = 78.6   # same guide: FP32 vector 157.3 TFLOP/s; fp64 FMA (vector and MFMA alike) runs at half of it
ENTRIES_CFG2 = "+x:1,3;-x:1,3;+y:1,3;-y:1,3;+z:1,3;-z:1,3;+t:1,3;-t:1,3"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--lattice", type=int, nargs=4, default=[32, 32, 32, 32], help="LOCAL lattice per GPU (x y z t)")
    ap.add_argument("--nev", type=int, default=200)
    ap.add_argument("--precision", type=int, default=8, choices=[4, 8])
    ap.add_argument("--order", type=int, default=2, choices=[2, 4])
    ap.add_argument("--loop-precision", type=int, default=0, choices=[0, 4, 8],
                    help="8 with --precision 4 = mixed precision (fp32 storage, fp64 accumulation; configs[3])")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="CPU-baseline budget (rank 0, N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true",
                    help="only the headline (use this under `rocprofv3 --stats`: the extra legs launch the headline kernel with "
                         "other shapes and would blur its per-kernel average)")
    ap.add_argument("--extra", default="", help="comma list restricting the extra legs: displaced,forced,strong,mg,cfg3 (N=1) / strong,partitioned,native (N>1)")
    ap.add_argument("--displaced-nev", type=int, default=400, help="eigenvectors of the displaced extra legs (configs[2]: 400 = 102 GB; 100 for a quick run)")
    ap.add_argument("--strong-nev", type=int, default=48, help="eigenvectors of the strong-scaling leg (48^3 x 96 on every N; 48 fit one GPU "
                                                              "next to all 25 position-space slots with room to spare: 98 + 68 GB; 64 is the most that fits)")
    ap.add_argument("--emulate-link-GBps", type=float, default=75.0,
                    help="forced-partition leg, second pass: hold the halo stream busy for bytes / this rate per (axis, direction) link, so that "
                         "the overlap schedule is exercised at about the pace of one xGMI link per neighbour (0: skip the pass)")
    ap.add_argument("--extra-timeout", type=float, default=420.0, help="watchdog for the extra legs (s); the headline line is printed anyway")
    # overrides of the partitioned leg (rehearsals on a one-GPU box: MUGIQ_BENCH_BACKEND=gloo and a small lattice)
    ap.add_argument("--part-lattice", type=int, nargs=4, default=None, help="LOCAL lattice of the partitioned leg")
    ap.add_argument("--part-nev", type=int, default=0)
    ap.add_argument("--part-grid", type=int, nargs=4, default=None)
    ap.add_argument("--strong-lattice", type=int, nargs=4, default=None, help="GLOBAL lattice of the strong-scaling leg (default 48 48 48 96)")
    ap.add_argument("--strong-grid", type=int, nargs=4, default=None)
    ap.add_argument("--detail-file", default="", help="where the full record (every leg, per-phase times, per-kernel rooflines) goes; "
                                                      "default gpurun_out/bench_detail_n<N>.json")
    return ap.parse_args()


# ---- synthetic inputs -------------------------------------------------------------------------
def global_site_index(X, grid, coord, device):
    """[2][volumeCB] int64: the lexicographic index in the GLOBAL lattice (X[d] * grid[d] per axis) of every local even-odd site
    (parity, x_cb) of the rank at `coord` (QUDA's checkerboard: full index i = x + X0 (y + X1 (z + X2 t)), x_cb = i / 2, parity =
    (x + y + z + t) & 1; the local extents are even, so the local parity is the global one)."""
    X0, X1, X2, X3 = X
    vcb = X0 * X1 * X2 * X3 // 2
    i2 = 2 * torch.arange(vcb, dtype=torch.int64, device=device)
    t = i2 // (X0 * X1 * X2)
    z = (i2 // (X0 * X1)) % X2
    y = (i2 // X0) % X1
    x0 = i2 % X0
    G = [X[d] * grid[d] for d in range(4)]
    out = []
    for par in range(2):
        x = x0 + ((y + z + t + par) & 1)
        gx, gy, gz, gt = x + coord[0] * X0, y + coord[1] * X1, z + coord[2] * X2, t + coord[3] * X3
        out.append(gx + G[0] * (gy + G[1] * (gz + G[2] * gt)))
    return torch.stack(out)


def _i64(x):
    """a Python integer wrapped to the signed 64-bit range (what int64 tensor arithmetic does on the device)"""
    x &= (1 << 64) - 1
    return x - (1 << 64) if x >= (1 << 63) else x


def _mix64(h):
    """splitmix64-style avalanche on int64 tensors (two's-complement wrap-around; the arithmetic shift only smears sign bits
    into positions that the following multiply scrambles again)"""
    h = (h ^ (h >> 30)) * (-4658895280553007687)        # 0xBF58476D1CE4E5B9
    h = (h ^ (h >> 27)) * (-7723592293110705685)        # 0x94D049BB133111EB
    return h ^ (h >> 31)


def _hash_complex(key):
    """complex128 with real and imaginary parts uniform in [-0.5, 0.5): a pure function of the int64 `key`"""
    h = _mix64(key)
    re = ((h >> 8) & 0xFFFFF).to(torch.float64) * (1.0 / 1048576.0) - 0.5
    im = ((h >> 32) & 0xFFFFF).to(torch.float64) * (1.0 / 1048576.0) - 0.5
    return torch.complex(re, im)


def make_evecs(hip, X, nev, prec, order, device, seed, pad=0, gidx=None, gvol=None):
    """N_ev synthetic eigenvectors in one HBM allocation (native layout, Stride() = volumeCB + pad), each unit-norm
    (the pad sites carry random numbers too; they are never addressed).
    gidx given (global_site_index): every component is a pure function of (seed, eigenvector, GLOBAL site, spin-colour), scaled
    by 1 / sqrt(12 * gvol) -- the same global field on any process grid, which is what lets a partitioned run be compared with
    the one-GPU run of the same job (FLOAT2, no pad)."""
    vcb = int(np.prod(X)) // 2
    per = 2 * 12 * (vcb + pad)
    cdt = torch.complex128 if prec == 8 else torch.complex64
    big = torch.empty(nev * per, dtype=cdt, device=device)
    if gidx is not None:
        assert order == 2 and pad == 0
        base = (gidx.view(2, 1, vcb) * 12 + torch.arange(12, dtype=torch.int64, device=device).view(1, 12, 1)).reshape(-1)   # native FLOAT2: [parity][k][x_cb]
        scale = float(np.sqrt(0.5 / gvol))                 # |uniform(-.5,.5) + i uniform(-.5,.5)|^2 averages 1/6; 12 * gvol components: global norm ~ 1
        fields = []
        for n in range(nev):
            v = big[n * per:(n + 1) * per]
            v.copy_((_hash_complex(base + _i64((seed * 1000003 + n) * 6364136223846793005)) * scale).to(cdt))
            fields.append(hip.SpinorField(X, prec, order, data=v))
        del base
        return big, fields
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    fields = []
    for n in range(nev):
        v = big[n * per:(n + 1) * per]
        re = torch.randn(per, dtype=torch.float64, device=device, generator=g)
        im = torch.randn(per, dtype=torch.float64, device=device, generator=g)
        w = torch.complex(re, im)
        w /= torch.linalg.vector_norm(w)
        v.copy_(w.to(cdt))
        fields.append(hip.SpinorField(X, prec, order, pad=pad, data=v))
        del re, im, w
    return big, fields


def random_su3_eo(X, device, seed, gidx=None):
    """Random SU(3) links of the local lattice, [4 dir][2 parity * volumeCB][3][3] complex128 on the device
    (Gram-Schmidt on rows of random matrices, third row = conj cross product => det 1).  gidx given: the matrix is a pure function
    of (seed, direction, GLOBAL site) -- the same global gauge field on any process grid."""
    vcb = int(np.prod(X)) // 2
    if gidx is not None:
        key = ((torch.arange(4, dtype=torch.int64, device=device).view(4, 1) * (1 << 40) + gidx.reshape(1, -1)).reshape(-1, 1) * 6
               + torch.arange(6, dtype=torch.int64, device=device).view(1, 6)) + _i64(seed * 6364136223846793005)
        m = _hash_complex(key).reshape(4 * 2 * vcb, 2, 3)
    else:
        gen = torch.Generator(device=device).manual_seed(seed)
        m = torch.complex(torch.randn(4 * 2 * vcb, 2, 3, dtype=torch.float64, device=device, generator=gen),
                          torch.randn(4 * 2 * vcb, 2, 3, dtype=torch.float64, device=device, generator=gen))
    r0 = m[:, 0] / torch.linalg.vector_norm(m[:, 0], dim=-1, keepdim=True)
    r1 = m[:, 1] - (r0.conj() * m[:, 1]).sum(-1, keepdim=True) * r0
    r1 = r1 / torch.linalg.vector_norm(r1, dim=-1, keepdim=True)
    r2 = torch.linalg.cross(r0.conj(), r1.conj())
    return torch.stack([r0, r1, r2], dim=1).reshape(4, 2 * vcb, 3, 3)


def make_gauge(hip, X, prec, device, seed, comm=None, gidx=None):
    """The border-extended device gauge field.  One process: written in place (periodic, no border).  With a process
    grid: through Displace's setup path -- host QDP links of the local lattice -> mugiq_hip_create_extended_gauge,
    borders from the neighbours (lib/displace.cpp:104-134)."""
    vcb = int(np.prod(X)) // 2
    u = random_su3_eo(X, device, seed, gidx)
    if comm is None:
        g = hip.GaugeField(X, (0, 0, 0, 0), prec)
        q = u.reshape(4, 2, vcb, 9).permute(1, 0, 3, 2).contiguous()          # [parity][dir][row*3+col][x_cb]
        g.data.copy_(q.reshape(-1).to(g.data.dtype))
        return g
    R = [2 * comm.comm_dim_partitioned(d) for d in range(4)]                   # lib/displace.cpp:16
    qdp = [np.ascontiguousarray(u[d].reshape(-1).cpu().numpy()).view(np.float64) for d in range(4)]
    del u
    return hip.GaugeField(X, R, prec).set_from_qdp_host(qdp, comm)


def momenta_p2_le(n):
    r = int(np.floor(np.sqrt(n)))
    return [[x, y, z] for x in range(-r, r + 1) for y in range(-r, r + 1) for z in range(-r, r + 1) if x * x + y * y + z * z <= n]


# ---- CPU baseline ----------------------------------------------------------------------------------------------------
def cpu_baseline(fields, sigmas, X, prec, order, budget_s):
    """Time the plain-C restatement of the reference kernel (oracle/mugiq_oracle.c, `port`) on the host cores,
    on a bounded sample of the same workload: the first S even + S odd checkerboard sites of every eigenvector, S sized so
    that every host thread gets >= 4096 sites; the sample is copied into place by the threads that read it (first touch)."""
    from oracle import c_oracle
    nev = len(fields)
    vcb = fields[0].volumeCB
    if "OMP_NUM_THREADS" not in os.environ:
        c_oracle.set_num_threads(c_oracle.usable_cpus())   # affinity mask capped by the cgroup CPU quota, not every hardware thread
    threads = c_oracle.num_threads()
    S = min(vcb, max(32768, 2048 * threads))      # sample: 2*S sites x all N_ev eigenvectors
    cdt = np.complex128 if prec == 8 else np.complex64
    bufs = []
    for f in fields:
        d = f.data.view(2, -1)              # [parity][12*stride] complex (parity_offset = 12*stride)
        if order == 2:
            h = d.view(2, 12, f.stride)[:, :, :S].contiguous().cpu().numpy()              # planes of S sites
        else:
            h = d.view(2, 6, f.stride, 2)[:, :, :S, :].contiguous().cpu().numpy()
        h = np.ascontiguousarray(h.reshape(-1)).astype(cdt, copy=False)
        bufs.append(c_oracle.first_touch_copy(h, 12 if order == 2 else 6, S))
        del h
    loop = np.zeros(16 * 2 * S, dtype=cdt)
    c_oracle.loop_contract_native(loop, bufs, bufs, sigmas, S, S, 12 * S, order)        # warm-up pass (first touch of `loop`)
    t0 = time.perf_counter()
    passes = 0
    while True:
        loop[:] = 0
        c_oracle.loop_contract_native(loop, bufs, bufs, sigmas, S, S, 12 * S, order)
        passes += 1
        el = time.perf_counter() - t0
        if el >= budget_s or passes >= 1000:
            break
    sites_per_s = passes * 2 * S / el
    return {"value": sites_per_s, "unit": "sites/s", "cores": threads, "kind": "port",
            "host_GBps": sites_per_s * nev * 24 * prec / 1e9,
            "sample_short": "%d sites x %d eigenvectors of the bench fields, %d passes in %.1f s, oracle/mugiq_oracle.c + OpenMP" % (2 * S, nev, passes, el),
            "sample": "%d sites x %d eigenvectors (first %d checkerboard sites of each parity of the bench fields = %d sites per "
                      "thread, pages first-touched by the reading threads), %d passes in %.1f s, oracle/mugiq_oracle.c with OpenMP (threads = the "
                      "CPUs this job may use: affinity mask capped by the cgroup quota; the host has %d hardware threads), chunks of "
                      "256 sites x all eigenvectors, blocks of 8 sites" % (2 * S, nev, S, 2 * S // threads, passes, el, os.cpu_count() or 0)}, loop, S


# ---- roofline helpers --------------------------------------------------------------------------------------------------
def roof(bound, kernel, ms, alg_bytes=None, flops=None, note=None):
    """One roofline block: algorithmic bytes (read-once minimum) and/or flops of the mathematics over the device time."""
    out = {"bound": bound, "kernel": kernel, "kernel_ms": ms}
    if alg_bytes is not None:
        gbs = alg_bytes / (ms * 1e-3) / 1e9
        out.update({"algorithmic_bytes": alg_bytes, "achieved_GBps": gbs, "hbm_frac": gbs / HBM_PEAK_GBS})
    if flops is not None:
        tf = flops / (ms * 1e-3) / 1e12
        out.update({"flops": flops, "achieved_TFLOPs": tf, "fp64_vector_frac": tf / FP64_VECTOR_PEAK_TFLOPS})
    if bound == "hbm":
        out.update({"achieved": out["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": out["hbm_frac"]})
    else:
        out.update({"achieved": out["achieved_TFLOPs"], "peak": FP64_VECTOR_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": out["fp64_vector_frac"]})
    out["traffic"] = None                     # PMC counters are not collected in-run; see profiles/ for rocprofv3 --pmc passes
    if note:
        out["note"] = note
    return out


def _normalised(path):
    """source text without comments and with whitespace collapsed: a comment edit does not invalidate a PMC measurement"""
    t = open(path, "r", errors="replace").read()
    t = re.sub(r"/\*.*?\*/", " ", t, flags=re.S)
    t = re.sub(r"//[^\n]*", " ", t)
    return re.sub(r"\s+", " ", t).encode()


def source_fingerprint():
    """sha1 over the sources of the headline kernel: a committed PMC measurement is attached to a bench line only if it was
    taken with the same kernel (tools/pmc_traffic.py records the same fingerprint)."""
    h = hashlib.sha1()
    for f in ("contract.hip", "internal.h"):
        h.update(_normalised(os.path.join(ROOT, "mugiq_amd", "csrc", f)))
    return h.hexdigest()[:12]


def library_fingerprint():
    """sha1 over the kernel sources of libmugiq_hip.so (tools/pmc_traffic_extra.py records the same)"""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "mugiq_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith(".hip") or f == "internal.h":       # the kernels (the host-side driver does not change a kernel's traffic)
            h.update(_normalised(os.path.join(d, f)))
    return h.hexdigest()[:12]


_EXTRA_TRAFFIC = None


def extra_traffic_nev():
    """N_ev of the displaced leg the committed PMC passes ran with (files written before that was recorded: 100)"""
    attach_traffic({}, ["\0"])
    return _EXTRA_TRAFFIC.get("displaced_nev", 100) if _EXTRA_TRAFFIC else None


def attach_traffic(block, substrings, grid=None):
    """HBM bytes per launch of a kernel of the extra legs from the committed PMC passes (profiles/traffic_extra_latest.json),
    attached only when they were taken with the library sources of this run; otherwise the block keeps traffic = null."""
    global _EXTRA_TRAFFIC
    if _EXTRA_TRAFFIC is None:
        _EXTRA_TRAFFIC = {}
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic_extra_latest.json")))
            if tj.get("library_fingerprint") == library_fingerprint():
                _EXTRA_TRAFFIC = tj
        except (OSError, ValueError):
            pass
    for k in _EXTRA_TRAFFIC.get("kernels", []):
        if all(x in k["name"] for x in substrings) and (grid is None or k["grid"] == grid):
            block["traffic"] = k["hbm_bytes_per_launch"]
            block["traffic_source"] = "NOT measured in this run: profiles/traffic_extra_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of " \
                                      "this workload, library sources %s = the ones running here)" % _EXTRA_TRAFFIC["library_fingerprint"]
            break
    return block


def phase_sum(phases, kind, entry=None):
    return sum(p["ms"] for p in phases if p["kind"] == kind and (entry is None or p["entry"] == entry))


# ---- extra legs ----------------------------------------------------------------------------------------------------------
def displaced_job(hip, device, X, nev, prec, comm, world, reps=2, p2max=9, backend="nccl", fields=None, gauge=None, hashed=False):
    """The configs[2] job through the driver's OPT plan: ultra-local + 8 entries x lengths 1..3 (25 slots), then the
    momentum projection onto p^2 <= p2max.  Returns the record with per-phase device times (best repetition) and, under "_mom",
    the gathered momentum-space loops [Nmom][nLoop][16][totT] of that repetition (popped by the callers before anything is printed).
    hashed: eigenvectors and links are functions of the GLOBAL site (make_evecs / random_su3_eo with gidx), so that the result
    does not depend on the process grid."""
    V = int(np.prod(X))
    B = prec
    multi = comm is not None and world > 1
    gidx = None
    if hashed:
        grid, coord = (comm.grid, comm.coord) if comm is not None else ((1, 1, 1, 1), (0, 0, 0, 0))
        gidx = global_site_index(X, grid, coord, device)
        gvol = V * int(np.prod(grid))
    if fields is None:
        if hashed:
            _, fields = make_evecs(hip, X, nev, prec, 2, device, seed=4242, gidx=gidx, gvol=gvol)
        else:
            _, fields = make_evecs(hip, X, nev, prec, 2, device, seed=4242 + (comm.rank if comm else 0))
    if gauge is None:
        gauge = make_gauge(hip, X, prec, device, 20240501 + (0 if hashed else (comm.rank if comm else 0)), comm, gidx)
    del gidx
    torch.cuda.empty_cache()       # the generators' temporaries go back to the device: the driver's own hipMalloc calls crawl when HBM is nearly full
    moms = momenta_p2_le(p2max)
    sig = 0.01 + 0.002 * np.arange(nev)
    best = None
    for r in range(reps + 1):                 # the first repetition warms the scratch pool up (hipMalloc of ~GB buffers)
        prm = hip.MugiqLoopParam(gauge=gauge, calcType=hip.LOOP_CALC_TYPE_OPT_KERNEL, doMomProj=True, momMatrix=moms,
                                 Nmom=len(moms), FTSign=-1).set_displace_entry_string(ENTRIES_CFG2)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        loop = hip.Loop_Mugiq(prm, fields, sig, comm).setProfiling()
        torch.cuda.synchronize()
        create_s = time.perf_counter() - tc       # buffers (loop slots, the halo buffers of the plan), phase matrix: the constructor's work
        if multi:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop.computeCoarseLoop()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        el = time.perf_counter() - t0
        if multi:
            t = torch.tensor([el], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t[0])
        free_b, total_b = torch.cuda.mem_get_info(device)
        rec = {"seconds": el, "phases": loop.phases(), "nLoop": loop.nLoop, "carrier": loop.ultraLocalCarrier(), "packed_in_entry": loop.halosPackedInEntry(),
               "derived": [loop.derivedFrom(i) for i in range(loop.nDispEntries)],
               "entries": [loop.entry(i) for i in range(loop.nDispEntries)],
               "device_bytes_in_use": int(total_b - free_b), "create_seconds": create_s,      # eigenvectors + links + loop buffers + the driver's scratch / halo pool
               "mom": loop.dataMom_global()}
        loop.close()
        if r > 0 and (best is None or el < best["seconds"]):
            best = rec
    ph = best["phases"]
    nslot = 3
    pmc_workload = comm is None and tuple(X) == (48, 48, 24, 24) and nev == extra_traffic_nev() and prec == 8   # what the committed PMC passes ran
    ent_bytes = V * (nev * 24 * B + nslot * (24 * B + 32 * B))                 # eigenvectors once + W_k once + slots written once
    ent_flops = V * nev * nslot * (36 + 48) * 8.0                               # SU(3) x spinor + colour-traced outer product, complex FMAs
    out = {"seconds": best["seconds"], "sites_per_s_all_slots": world * V / best["seconds"], "n_loop_slots": best["nLoop"],
           "entries_reflected": sum(1 for d in best["derived"] if d >= 0), "device_bytes_in_use": best["device_bytes_in_use"],
           "create_seconds": best["create_seconds"], "_mom": best["mom"],
           "roofline": {}}
    names = ["x", "y", "z", "t"]
    for i, e in enumerate(best["entries"]):
        if best["derived"][i] >= 0:
            continue
        tag = "entry_%s%s" % ("+" if e[1] == 1 else "-", names[e[0]])
        ms_f = phase_sum(ph, "entry_fused", i)
        if ms_f > 0:
            carries = best["carrier"] == i   # this entry's pass also produced the ultra-local loop: + 48 complex FMAs per site and eigenvector, + its slot
            mfma = prec == 8 and e[2] == 1 and e[3] <= 3 and os.environ.get("MUGIQ_HIP_TILE_MFMA", "1") != "0"     # what csrc/fused.hip hands to csrc/fused_mfma.hip
            out["roofline"][tag] = roof("fp64_vector", (("mfma_tile_displaced_contract_kernel (axial gauge + fp64 matrix pipe; " + ("whole x rows" if e[0] == 0 else "column tile")) if mfma else
                                                        ("tile16_displaced_contract_kernel<DIR=0> (row tile, 16-line items" if e[0] == 0 else "tile_displaced_contract_kernel (column tile"))
                                        + (") + the ultra-local loop as a fourth slot" if carries else ")"),
                                        ms_f, ent_bytes + (V * 32 * B if carries else 0), ent_flops + (V * nev * 48 * 8.0 if carries else 0), note=CLOCK_NOTE)
            if pmc_workload:
                attach_traffic(out["roofline"][tag], ["mfma_tile_displaced_contract_kernel<%d, %d," % (e[0], e[1])] if mfma else
                               ["displaced_contract_kernel<double, double, 2, %d, %d," % (e[0], e[1])])
        else:
            ms_i, ms_b = phase_sum(ph, "entry_interior", i), phase_sum(ph, "entry_boundary", i)
            out["roofline"][tag] = roof("fp64_vector", "displaced contraction, interior + boundary tiles (partitioned axis)",
                                        ms_i + ms_b, ent_bytes, ent_flops)
            out["roofline"][tag].update({"interior_ms": ms_i, "boundary_ms": ms_b, "halo_wait_ms": phase_sum(ph, "halo_wait", i)})
    ms_u = phase_sum(ph, "ultra_local")
    if ms_u > 0:
        out["roofline"]["ultra_local"] = roof("hbm", "loop_contract_kernel", ms_u, V * (nev * 24 * B + 32 * B))
        if pmc_workload:
            attach_traffic(out["roofline"]["ultra_local"], ["loop_contract_kernel"], V)
    else:
        out["ultra_local"] = "carried by a displaced entry as a fourth slot of the tiled kernel (no pass of its own)"
    ms_r = phase_sum(ph, "entry_reflected")
    nref = sum(3 for d in best["derived"] if d >= 0)
    if nref and ms_r > 0:
        out["roofline"]["reflected_slots"] = roof("hbm", "reflect_kernel (%d slots)" % nref, ms_r, nref * V * 2 * 32 * B)
    elif nref:
        out["reflected_slots"] = "derived in momentum space on the gathered array (host, %.3f ms); never formed in position space" % phase_sum(ph, "momentum_reflect")
    ms_m = phase_sum(ph, "momentum_projection")
    nData = 16 * (best["nLoop"] - (nref if nref and ms_r == 0 else 0))     # slots that went through the reorder + Fourier kernels
    npx = len(set(m[0] for m in moms))
    out["roofline"]["momentum_projection"] = roof("hbm", "eo_dft_x (reorder + x sum) + partial_dft_kernel (y, z)", ms_m,
                                                  V * nData * 2 * B * (1 + npx / X[0]),
                                                  note="algorithmic bytes = the position-space buffer read once + the x-summed array (%d distinct p_x of "
                                                       "%d momenta, 1/%d of the input per p_x) written once; the y and z steps work on arrays Lx and Lx*Ly "
                                                       "times smaller and are counted in the time only" % (npx, len(moms), X[0]))
    if pmc_workload and "momentum_projection" in out["roofline"]:
        attach_traffic(out["roofline"]["momentum_projection"], ["eo_dft_x"])
        if out["roofline"]["momentum_projection"].get("traffic"):
            out["roofline"]["momentum_projection"]["traffic_note"] = "eo_dft_x only (the y and z steps move 1.3 GB more)"
    out["momentum_copy_ms"] = phase_sum(ph, "momentum_copy")
    out["momentum_reduce_host_ms"] = phase_sum(ph, "momentum_reduce")
    halo = [p for p in ph if p["kind"] == "halo_transfer"]
    if halo:
        hb, hms = sum(p["bytes"] for p in halo), sum(p["ms"] for p in halo)
        pms = phase_sum(ph, "halo_prepare")
        out["halo"] = {"bytes_sent_per_rank": hb, "transfer_ms": hms, "GBps_per_rank": hb / (hms * 1e-3) / 1e9 if hms > 0 else None,
                       "prepare_ms": pms, "pack_GBps": 2 * hb / (pms * 1e-3) / 1e9 if pms > 0 else None,
                       "wait_ms_not_hidden": phase_sum(ph, "halo_wait"), "messages": len(halo),
                       "halos_packed_by_first_entry": best.get("packed_in_entry", 0),
                       "note": "pack_GBps counts the face layers read + written by pack_layers_kernel over the prepare phase "
                               "(which also builds the path-link fields of the entry)"}
        if best.get("packed_in_entry", 0) > 0:
            out["halo"]["pack_GBps"] = None
            out["halo"]["note"] = ("the face layers of %d halos were written by the entry that runs first, on its way through the eigenvectors "
                                   "(csrc/fused_mfma.hip row tile); the prepare phase ends when that entry does and says nothing about a pack rate"
                                   % best["packed_in_entry"])
    out["phase_ms"] = {k: phase_sum(ph, k) for k in sorted(set(p["kind"] for p in ph))}
    return out


_CFG2_INPUTS = {}


def cfg2_inputs(hip, device, nev):
    """The eigenvectors of the configs[2] per-GPU lattice, generated once and shared by the unpartitioned and the
    forced-partition leg (102 GB at N_ev = 400)."""
    if _CFG2_INPUTS.get("nev") != nev:
        _CFG2_INPUTS.clear()
        big, fields = make_evecs(hip, (48, 48, 24, 24), nev, 8, 2, device, seed=4242)
        _CFG2_INPUTS.update({"nev": nev, "big": big, "fields": fields})
    return _CFG2_INPUTS["fields"]


def max_rel_diff(a, b):
    """max |a - b| / max |b| over the gathered momentum-space loops of two runs of the same job"""
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))


PARITY_TOL = 1e-12        # north_star: loop traces within 1e-12 in fp64


def extra_displaced(hip, device, nev=400):
    X = (48, 48, 24, 24)
    out = displaced_job(hip, device, X, nev, 8, None, 1, fields=cfg2_inputs(hip, device, nev))
    _CFG2_INPUTS["mom_unpartitioned"] = out.pop("_mom")          # 19 MB on the host: what the forced-partition leg must reproduce
    out["workload"] = "48x48x24x24 fp64 N_ev=%d (configs[2] per-GPU lattice%s), entries %s, momentum projection p^2<=9, driver OPT plan" % (
        nev, "" if nev == 400 else ", N_ev reduced from 400", ENTRIES_CFG2)
    return out


def extra_forced(hip, device, nev=400, emulate_GBps=0.0):
    """configs[2] as ONE rank of its 1x1x2x4 grid sees it: the partitioned code path forced on z and t with the rank as its own
    forward and backward neighbour (MugiqHipComm.partitioned = QUDA's comm_dim_partitioned_set).  Everything the 8-GPU run does
    on a rank happens here at full size -- gauge borders R = 2 through sendrecv, `stop` face layers of all eigenvectors packed
    and posted ahead on the halo stream, interior tiles before the halo event, boundary tiles after, halos of the reflected
    slots -- except that the message is a device copy instead of an xGMI transfer."""
    X = (48, 48, 24, 24)
    fields = cfg2_inputs(hip, device, nev)
    comm = hip.GridComm((1, 1, 1, 1), device=device, force_partitioned=(0, 0, 1, 1))
    gauge = make_gauge(hip, X, 8, device, 20240501, comm)
    out = displaced_job(hip, device, X, nev, 8, comm, 1, reps=1, fields=fields, gauge=gauge)
    mom = out.pop("_mom")
    ref = _CFG2_INPUTS.get("mom_unpartitioned")
    if ref is not None:
        # same eigenvectors, same links (seed), same momenta: the partitioned code path must give the unpartitioned numbers
        out["max_rel_diff_forced_vs_unpartitioned"] = max_rel_diff(mom, ref)
        out["parity_ok"] = bool(out["max_rel_diff_forced_vs_unpartitioned"] < PARITY_TOL)
    out["workload"] = "48x48x24x24 fp64 N_ev=%d, z and t FORCED-partitioned on one rank (self-neighbour: face layers written straight into the ghost buffers by the first entry, no xGMI), " \
                      "entries %s, momentum projection p^2<=9, driver OPT plan, halos posted ahead" % (nev, ENTRIES_CFG2)
    out["forced_partition"] = [0, 0, 1, 1]
    if emulate_GBps > 0:
        # the same job once more with the self-messages slowed down to the pace of an xGMI link (see GridComm): what the schedule
        # hides and what it does not when the halo takes as long as it would between GPUs
        comm2 = hip.GridComm((1, 1, 1, 1), device=device, force_partitioned=(0, 0, 1, 1), emulate_link_GBps=emulate_GBps)
        os.environ["MUGIQ_HIP_SELF_HALO_COPY"] = "1"      # the messages to self are copies again (default: packed in place), so that they can be paced
        try:
            emu = displaced_job(hip, device, X, nev, 8, comm2, 1, reps=1, fields=fields, gauge=gauge)
        finally:
            os.environ.pop("MUGIQ_HIP_SELF_HALO_COPY", None)
        emom = emu.pop("_mom")
        if ref is not None:
            out["max_rel_diff_emulated_vs_unpartitioned"] = max_rel_diff(emom, ref)
            out["parity_ok"] = bool(out.get("parity_ok", True) and out["max_rel_diff_emulated_vs_unpartitioned"] < PARITY_TOL)
        h = emu.get("halo", {})
        out["emulated_link"] = {"GBps_per_link": emulate_GBps, "seconds": emu["seconds"], "halo_transfer_ms": h.get("transfer_ms"),
                                "halo_wait_ms_not_hidden": h.get("wait_ms_not_hidden"), "phase_ms": emu["phase_ms"],
                                "note": "EMULATION on one GPU: the self-neighbour copies are followed by a spin kernel on the halo stream until bytes / rate "
                                        "have passed; the z and t messages of the transfer group count as two links side by side.  No xGMI, no RCCL: it "
                                        "shows the overlap schedule (what waits for what), not a transport"}
    return out


def extra_mg(hip, device):
    """configs[4]: 32^4 fp64, 4^4 aggregates, n_vec 24, N_ev 200 coarse eigenvectors, ultra-local loop through the driver."""
    X, nvec, nev = (32, 32, 32, 32), 24, 200
    V = int(np.prod(X))
    T = hip.Transfer(X, nvec, (4, 4, 4, 4), 2, 8)
    g = torch.Generator(device=device).manual_seed(99)
    T.V.copy_(torch.complex(torch.randn(T.V.numel(), dtype=torch.float64, device=device, generator=g),
                            torch.randn(T.V.numel(), dtype=torch.float64, device=device, generator=g)) / np.sqrt(24.0 * nvec))
    cf = []
    for n in range(nev):
        c = hip.CoarseField(T.Xc, nvec, 8)
        c.data.copy_(torch.complex(torch.randn(c.data.numel(), dtype=torch.float64, device=device, generator=g),
                                   torch.randn(c.data.numel(), dtype=torch.float64, device=device, generator=g)))
        cf.append(c)
    sig = 0.01 + 0.002 * np.arange(nev)
    best = None
    for r in range(3):
        loop = hip.Loop_Mugiq(hip.MugiqLoopParam(), cf, sig, transfer=T).setProfiling()
        loop.computeCoarseLoop()
        ms = phase_sum(loop.phases(), "ultra_local")
        loop.close()
        if r > 0 and (best is None or ms < best):
            best = ms
    NC = 2 * nvec
    # coarse-grid plan: C(X) = sum_n phi phi^dag / sigma (volc * nev * NC^2 complex FMAs), then per fine site the congruence
    # V C V^dag: Y(s,c; chi',j') = sum_j V(s,c,j) C[(chi(s),j),(chi',j')] (12 * NC * n_vec) and the colour-traced 4x4 spin
    # matrix sum_{c,j'} Y(s,c; chi(s'),j') conj V(s',c,j') (16 * 3 * n_vec); 8 flops per complex FMA
    volc = V // 256
    flops = 8.0 * (volc * nev * NC * NC + V * (12 * NC * nvec + 48 * nvec))
    byts = V * 12 * nvec * 16 + nev * volc * NC * 16 + V * 32 * 8
    res = {"workload": "32x32x32x32 fp64 MG coarse path: n_vec=24, 4^4 aggregates, N_ev=200 coarse eigenvectors, ultra-local loop (configs[4])",
            "kernel_ms": best, "sites_per_s": V / (best * 1e-3),
            "roofline": roof("fp64_vector", "coarse_outer_kernel + fine_congruence_kernel", best, byts, flops,
                             note="flops of the coarse-grid plan actually executed (outer product on the coarse grid + V C V^dag per fine "
                                  "site), not of the per-eigenvector prolongation it replaces (8*12*n_vec*V*N_ev = %.0f GFLOP)" % (8.0 * 12 * nvec * V * nev / 1e9))}
    # prolongateEvecs: coarse -> fine for all eigenvectors, the first step of every DISPLACED MG loop (lib/loop_mugiq.cpp:482)
    try:
        big = torch.empty(nev * 24 * (V // 2), dtype=torch.complex128, device=device)
        ff = [hip.SpinorField(X, 8, 2, data=big[n * 24 * (V // 2):(n + 1) * 24 * (V // 2)]) for n in range(nev)]
        # queued back to back (the host runs ahead of the device): with a synchronisation per call the time Python spends building
        # the 400 field descriptors would sit between the two events
        evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(4)]
        for e0, e1 in evs:
            e0.record()
            hip.prolongateEvecs(ff, cf, T)
            e1.record()
        torch.cuda.synchronize()
        pbest = min(e0.elapsed_time(e1) for e0, e1 in evs[1:])
        res["prolongate_to_fine"] = roof("fp64_vector", "prolong_mfma_kernel<24> x passes + coarse_pack_kernel (prolongateEvecs, all %d eigenvectors)" % nev,
                                         pbest, V * nev * 192 + V * 12 * nvec * 16, 8.0 * 12 * nvec * V * nev,
                                         note="algorithmic bytes = the fine eigenvectors written once + V read once; fp64 MFMA shares the vector peak")
        del ff, big
    except Exception as e:                                            # (never take the headline down with an extra)
        res["prolongate_to_fine"] = {"error": repr(e)}
    a1, a2 = attach_traffic({}, ["coarse_outer_kernel"]), attach_traffic({}, ["fine_congruence"])
    if a1.get("traffic") and a2.get("traffic"):
        res["roofline"]["traffic"] = a1["traffic"] + a2["traffic"]
        res["roofline"]["traffic_source"] = a2["traffic_source"]
    return res


def extra_cfg3(hip, device):
    """configs[3] per-GPU share through the driver: 64.64.32.16, fp32 FLOAT4 eigenvectors, fp64 loop accumulation, N_ev = 600
    (121 GB), ultra-local loop + momentum projection (FT over sites, p^2 <= 9) -- "mixed-precision contraction + momentum-projected
    loop" of BASELINE.json."""
    X, nev = (64, 64, 32, 16), 600
    V = int(np.prod(X))
    _, fields = make_evecs(hip, X, nev, 4, 4, device, seed=31337)
    sig = 0.01 + 0.002 * np.arange(nev)
    moms = momenta_p2_le(9)
    best = None
    for r in range(3):
        prm = hip.MugiqLoopParam(doMomProj=True, momMatrix=moms, Nmom=len(moms), FTSign=-1, loopPrecision=8)
        loop = hip.Loop_Mugiq(prm, fields, sig).setProfiling()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop.computeCoarseLoop()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        ph = loop.phases()
        loop.close()
        if r > 0 and (best is None or el < best[0]):
            best = (el, phase_sum(ph, "ultra_local"), phase_sum(ph, "momentum_projection"), phase_sum(ph, "momentum_copy"))
    el, ms_u, ms_m, ms_c = best
    npx = len(set(m[0] for m in moms))
    return {"workload": "64x64x32x16 fp32-storage/fp64-accumulate N_ev=600 ultra-local loop + momentum projection onto %d momenta (configs[3] per-GPU "
                        "share, FLOAT4), through the driver" % len(moms),
            "seconds": el, "kernel_ms": ms_u, "sites_per_s": V / el, "momentum_projection_ms": ms_m, "momentum_copy_ms": ms_c,
            "roofline": roof("hbm", "loop_contract_kernel<float,double,4,...>", ms_u, V * (nev * 24 * 4 + 32 * 8)),
            "roofline_momentum_projection": roof("hbm", "eo_dft_x + partial_dft_kernel (1 slot, fp64)", ms_m, V * 16 * 16 * (1 + npx / X[0]))}


def extra_partitioned(hip, device, a, world, rank, backend):
    """configs[2] partitioned over the ranks through Loop_Mugiq + GridComm."""
    grid = {1: (1, 1, 1, 1), 2: (1, 1, 1, 2), 4: (1, 1, 1, 4), 8: (1, 1, 2, 4)}.get(world)
    if a.part_grid:
        grid = tuple(a.part_grid)
    if grid is None or int(np.prod(grid)) != world:
        return {"error": "no process grid for %d ranks" % world}
    G = (48, 48, 48, 96)
    X = tuple(a.part_lattice) if a.part_lattice else tuple(G[d] // grid[d] for d in range(4))
    # N_ev: the metric's 400 where 48^3 x 96 x 400 fits the ranks' HBM (N >= 4), else what fits
    nev = a.part_nev or (400 if world >= 4 else 100)
    if world == 4 and not a.part_lattice:
        # 204 GB of eigenvectors per GPU: no room for the 2 x 25.5 GB halo buffers posted ahead -> halo in blocks of <= 4 GiB
        os.environ.setdefault("MUGIQ_HIP_HALO_AHEAD", "0")
    comm = hip.GridComm(grid, device=device)
    out = displaced_job(hip, device, X, nev, 8, comm, world, reps=1, backend=backend)
    out.pop("_mom", None)
    out["workload"] = "%dx%dx%dx%d global (local %dx%dx%dx%d on a %dx%dx%dx%d process grid) fp64 N_ev=%d, entries %s, momentum projection p^2<=9, " \
                      "driver OPT plan, halos over %s" % (tuple(X[d] * grid[d] for d in range(4)) + X + grid + (nev, ENTRIES_CFG2, backend))
    out["grid"], out["local_lattice"], out["n_ev"] = list(grid), list(X), nev
    out["halo_ahead"] = os.environ.get("MUGIQ_HIP_HALO_AHEAD", "1")
    return out


def extra_strong(hip, device, a, world, rank, backend):
    """Strong scaling of the PARTITIONED path: the SAME global problem on every N -- configs[2]'s 48^3 x 96 lattice, all 25 slots,
    momentum projection -- with as many eigenvectors as fit ONE GPU next to the position-space slots (N_ev = --strong-nev; halo
    bytes and arithmetic both scale with N_ev, so the ratio that decides the scaling is that of the N_ev = 400 job).  Eigenvectors
    and links are functions of the GLOBAL site, so every process grid works on the same global fields.
    N = 1 runs it unpartitioned.  N > 1: rank 0 ALONE first runs the unpartitioned job (the others wait at a barrier), keeps its
    seconds and its momentum-space loops; then all ranks run it on 1x1x1x2, 1x1x1x4, 1x1x2x4 through GridComm (T first, then Z).
    The line carries seconds_1gpu, seconds, speedup = seconds_1gpu / seconds -- the number north_star asks about -- and the
    largest relative difference between the partitioned and the one-GPU loops."""
    grid = {1: (1, 1, 1, 1), 2: (1, 1, 1, 2), 4: (1, 1, 1, 4), 8: (1, 1, 2, 4)}.get(world)
    if a.strong_grid:
        grid = tuple(a.strong_grid)
    if grid is None or int(np.prod(grid)) != world:
        return {"error": "no process grid for %d ranks" % world}
    G = tuple(a.strong_lattice) if a.strong_lattice else (48, 48, 48, 96)
    X = tuple(G[d] // grid[d] for d in range(4))
    nev = a.strong_nev
    one = None
    if world > 1:
        import torch.distributed as dist
        if rank == 0:
            one = displaced_job(hip, device, G, nev, 8, None, 1, reps=1, backend=backend, hashed=True)
            torch.cuda.empty_cache()
        dist.barrier()
    comm = hip.GridComm(grid, device=device) if world > 1 else None
    out = displaced_job(hip, device, X, nev, 8, comm, world, reps=1, backend=backend, hashed=True)
    mom = out.pop("_mom")
    out["workload"] = "%dx%dx%dx%d GLOBAL lattice on %d GPU(s) (process grid %dx%dx%dx%d, local %dx%dx%dx%d) fp64 N_ev=%d, entries %s, momentum " \
                      "projection p^2<=9, driver OPT plan%s" % (G + (world,) + grid + X + (nev, ENTRIES_CFG2, (", halos over " + backend) if world > 1 else ""))
    out["grid"], out["local_lattice"], out["n_ev"] = list(grid), list(X), nev
    out["global_sites"] = int(np.prod(G))
    out["seconds_1gpu"] = out["seconds"] if world == 1 else None
    if world > 1:
        one = dist_bcast_reference(one, mom, device, backend, rank)
        _STRONG_REF.update({"mom": one["_mom"], "seconds": one["seconds"], "grid": grid, "G": G, "X": X, "nev": nev})
    if one is not None:
        out["seconds_1gpu"] = one["seconds"]
        out["speedup"] = one["seconds"] / out["seconds"]
        out["max_rel_diff_vs_1gpu"] = max_rel_diff(mom, one.pop("_mom"))
        out["parity_ok"] = bool(out["max_rel_diff_vs_1gpu"] < PARITY_TOL)
        out["one_gpu_phase_ms"] = one["phase_ms"]
    return out


# ---- the printed line ------------------------------------------------------------------------------------------------------
_ROOF_KEYS = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "kernel_ms", "probe_GBps", "frac_of_probe")
LINE_LIMIT = 3000          # characters; the driver reads the last ~8 KB of stdout


def _r(x, digits=6):
    """numbers to `digits` significant figures (the full-precision values are in the detail file)"""
    if isinstance(x, (float, np.floating)):
        return float("%.*g" % (digits, x)) if np.isfinite(x) else None      # strict JSON has no NaN / Infinity
    if isinstance(x, (np.integer, np.bool_)):
        return x.item()
    if isinstance(x, dict):
        return {k: _r(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, digits) for v in x]
    return x


def _leg_summary(rec):
    """one short object per extra leg: its time, its slowest roofline fraction and what bounds that kernel"""
    if "error" in rec:
        return {"error": str(rec["error"])[:120]}
    o = {}
    if "seconds" in rec:
        o["s"] = rec["seconds"]
    elif "kernel_ms" in rec:
        o["ms"] = rec["kernel_ms"]
    rf = rec.get("roofline", {})
    blocks = [rf] if "frac" in rf else [b for b in rf.values() if isinstance(b, dict) and "frac" in b]
    if blocks:
        heavy = max(blocks, key=lambda b: b["kernel_ms"])        # the block that takes the most time
        o.update({"top_kernel_ms": heavy["kernel_ms"], "top_frac": heavy["frac"], "top_bound": heavy["bound"]})
    if "skipped" in rec:
        return {"skipped": str(rec["skipped"])[:80]}
    for k in ("max_rel_diff_forced_vs_unpartitioned", "max_rel_diff_emulated_vs_unpartitioned", "max_rel_diff_vs_1gpu", "parity_ok"):
        if k in rec:
            o[k] = rec[k]
    return o


def compact_line(out, detail_path):
    """The ONE line the driver parses: the contract's keys, the headline roofline, a trimmed cpu_baseline, the strong-scaling and
    partitioned summaries and one short object per extra leg -- nothing else.  Everything measured is in `detail_path`."""
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data") if k in out}
    c = out.get("config", {})
    line["config"] = {k: c[k] for k in ("workload", "local_lattice", "n_ev", "partition") if k in c}
    line["value_is"] = "ultra-local contraction over independent site blocks (no halo on this path); the partitioned job is under strong_scaling"
    line["roofline"] = {k: out["roofline"].get(k) for k in _ROOF_KEYS if k in out["roofline"]}
    cb = out.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample_short", "max_rel_err_gpu_vs_cpu_on_sample") if k in cb}
        line["cpu_baseline"]["sample"] = line["cpu_baseline"].pop("sample_short", str(cb.get("sample", ""))[:160])
    for k in ("backend", "nccl_ranks", "process_grid"):
        if k in out:
            line[k] = out[k]
    ss = out.get("strong_scaling")
    if ss:
        line["strong_scaling"] = {k: ss[k] for k in ("global_lattice", "n_ev", "grid", "seconds_1gpu", "seconds", "speedup",
                                                     "global_sites_per_s_all_slots", "halo_wait_ms_not_hidden", "halo_GBps_per_rank",
                                                     "max_rel_diff_vs_1gpu") if ss.get(k) is not None}
    pt = out.get("partitioned")
    if pt:
        line["partitioned"] = {k: pt[k] for k in ("local_lattice", "n_ev", "seconds", "sites_per_s_all_slots", "halo_bytes_sent_per_rank",
                                                  "halo_GBps_per_rank", "wait_ms_not_hidden") if pt.get(k) is not None}
    nt = (out.get("also_measured") or {}).get("native_rccl_transport")
    if isinstance(nt, dict) and "seconds" in nt:
        line["native_rccl"] = {k: nt[k] for k in ("seconds", "speedup", "max_rel_diff_vs_1gpu") if nt.get(k) is not None}
        if isinstance(nt.get("multipath"), dict):
            line["native_rccl"].update({"multipath_" + k: nt["multipath"][k] for k in ("seconds", "speedup", "max_rel_diff_vs_1gpu") if nt["multipath"].get(k) is not None})
    also = out.get("also_measured")
    if also:
        line["legs"] = {k: (_leg_summary(v) if isinstance(v, dict) else str(v)[:160]) for k, v in also.items()}
    if "parity_ok" in out:
        line["parity_ok"] = out["parity_ok"]
    line["detail_file"] = detail_path
    line = _r(line)
    text = json.dumps(line, allow_nan=False)
    if len(text) > LINE_LIMIT:                   # never again a line the driver cannot read: drop the per-leg summaries first
        line.pop("legs", None)
        line["legs_dropped"] = "line over %d characters; see detail_file" % LINE_LIMIT
        text = json.dumps(line, allow_nan=False)
    assert "\n" not in text
    return text


_STRONG_REF = {}


def dist_bcast_reference(one, mom, device, backend, rank):
    """rank 0's one-GPU result (seconds, momentum-space loops) to every rank, so that each can check what it holds"""
    import torch.distributed as dist
    wire = device if backend == "nccl" else "cpu"
    t = torch.from_numpy(np.ascontiguousarray(one["_mom"] if rank == 0 else np.zeros_like(mom))).view(torch.float64).to(wire)
    dist.broadcast(t, src=0)
    sec = torch.tensor([one["seconds"] if rank == 0 else 0.0], dtype=torch.float64, device=wire)
    dist.broadcast(sec, src=0)
    if rank != 0:
        one = {"seconds": float(sec[0]), "phase_ms": {}}
    one["_mom"] = t.cpu().numpy().view(np.complex128).reshape(mom.shape)
    return one


def extra_native(hip, device, a, world, rank, backend):
    """The strong-scaling job once more with the halos, the gauge borders and the momentum-space reductions on the library's OWN
    transport (csrc/comm_rccl.cpp: ncclSend / ncclRecv groups on the halo stream, ncclReduce / AllGather / Broadcast over
    ncclCommSplit sub-communicators; no Python callback in the data path) instead of torch.distributed.  Runs last: it has been on
    hardware with one rank only, and a failure here must not cost the other legs."""
    if backend != "nccl":
        return {"skipped": "the native transport is RCCL: one device per rank (backend %s here)" % backend}
    if not _STRONG_REF:
        return {"skipped": "no strong-scaling reference in this run"}
    r = _STRONG_REF
    out = None
    for multipath in (False, True):          # one message per neighbour, then every message over 1 + 6 xGMI paths (two hops through the other GPUs)
        if multipath and world <= 2:
            break
        comm = hip.RcclComm(r["grid"], device=device, multipath=multipath)
        rec = displaced_job(hip, device, r["X"], r["nev"], 8, comm, world, reps=1, backend=backend, hashed=True)
        mom = rec.pop("_mom")
        comm.close()
        rec["seconds_1gpu"] = r["seconds"]
        rec["speedup"] = r["seconds"] / rec["seconds"]
        rec["max_rel_diff_vs_1gpu"] = max_rel_diff(mom, r["mom"])
        rec["parity_ok"] = bool(rec["max_rel_diff_vs_1gpu"] < PARITY_TOL)
        if not multipath:
            out = rec
        else:
            out["multipath"] = {k: rec[k] for k in ("seconds", "speedup", "max_rel_diff_vs_1gpu", "parity_ok", "phase_ms")}
            out["multipath"]["halo"] = rec.get("halo")
            out["parity_ok"] = bool(out["parity_ok"] and rec["parity_ok"])
    out["workload"] = "the strong-scaling job (48x48x48x96 global, N_ev=%d) on mugiq_hip_rccl_comm_create's transport; `multipath`: the same with " \
                      "every halo message cut over 1 + 6 xGMI paths" % r["nev"]
    out["grid"], out["local_lattice"], out["n_ev"] = list(r["grid"]), list(r["X"]), r["nev"]
    return out


# ---- main --------------------------------------------------------------------------------------------------------------
def self_launch(a):
    """`python bench.py --gpus N` typed without a launcher: start the N ranks as CHILDREN (python -m torch.distributed.run)
    before this process has touched the GPU, relay rank 0's JSON line, exit with the launcher's status.  (Never exec: a process
    that has initialised the GPU must not be replaced.)"""
    backend = os.environ.get("MUGIQ_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()          # counting devices does not initialise the GPU
    if backend == "nccl" and ndev < a.gpus:
        raise SystemExit("bench.py --gpus %d: only %d HIP device(s) visible (RCCL needs one device per rank; MUGIQ_BENCH_BACKEND=gloo "
                         "rehearses N ranks on fewer devices)" % (a.gpus, ndev))
    sk = socket.socket()
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
    sk.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    for line in child.stdout:                 # rank 0's line goes to stdout, anything else the ranks print to stderr
        (sys.stdout if line.startswith('{"metric"') else sys.stderr).write(line)
        sys.stdout.flush()
    raise SystemExit(child.wait())


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(a)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE = %d" % (a.gpus, world))
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py needs an MI355X: no HIP device is visible (there is no CPU fallback)")
    backend = os.environ.get("MUGIQ_BENCH_BACKEND", "nccl")      # "gloo" only to rehearse N>1 on a one-GPU box
    dev_index = local_rank if backend == "nccl" else local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)     # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    import mugiq_amd as hip            # raises if libmugiq_hip.so is missing: there is no fallback

    X = tuple(a.lattice)
    V = int(np.prod(X))
    prec, order, nev = a.precision, a.order, a.nev
    B = prec
    sig = 0.01 + 0.002 * np.arange(nev)
    big, fields = make_evecs(hip, X, nev, prec, order, device, seed=777 + rank)
    lprec = a.loop_precision or prec
    cdt = torch.complex128 if lprec == 8 else torch.complex64
    loop = torch.zeros(16 * V, dtype=cdt, device=device)

    def step():
        loop.zero_()                                                     # cudaMemset, lib/loop_mugiq.cpp:476
        hip.performLoopContractionBatched(loop, fields, fields, sig)     # lib/loop_mugiq.cpp:478-503

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    barrier()
    # ---- timed region: exactly K steps ---------------------------------------------------------------
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(a.steps)]
    t0 = time.perf_counter()
    for i in range(a.steps):
        loop.zero_()
        ev0[i].record()                                                  # same (current) stream as the kernel
        hip.performLoopContractionBatched(loop, fields, fields, sig)
        ev1[i].record()
    barrier()
    elapsed = time.perf_counter() - t0
    # ---------------------------------------------------------------------------------------------------
    step_ms = np.array([ev0[i].elapsed_time(ev1[i]) for i in range(a.steps)])
    kern_ms = float(np.median(step_ms))                                  # median of the per-launch event times
    t = torch.tensor([elapsed, kern_ms], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed, kern_ms = float(t[0]), float(t[1])
    ms_per_step = elapsed * 1e3 / a.steps
    value = world * V / (ms_per_step * 1e-3)

    # what this device streams right now: a pure 16-B/lane non-temporal read of the SAME eigenvector buffer, same process,
    # straight after the timed region (tells a slow kernel from slow memory when a box runs off the usual numbers)
    probe_ms = []
    for r in range(6):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        hip.probeReadBandwidth(big, 1)
        e1.record()
        torch.cuda.synchronize()
        probe_ms.append(e0.elapsed_time(e1))
    probe_gbs = big.numel() * big.element_size() / (float(np.median(probe_ms[1:])) * 1e-3) / 1e9

    alg_bytes = V * (nev * 24 * B + 32 * lprec)      # per site N_ev*24*B read + 32*B written
    achieved = alg_bytes / (kern_ms * 1e-3) / 1e9
    workload = "%dx%dx%dx%d %s N_ev=%d ultra-local 16-gamma loop (order FLOAT%d)" % (
        X + ("fp64" if prec == 8 else ("fp32" if lprec == 4 else "fp32-storage/fp64-accumulate"), nev, order))
    # HBM traffic from the PMC counters cannot be collected inside this process; a committed rocprofv3 --pmc measurement
    # (tools/pmc_traffic.py -> profiles/traffic_latest.json) is attached only when it was taken for this workload with
    # exactly these kernel sources, and says where it came from
    traffic, traffic_source = None, None
    tfile = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            if tj.get("workload") == workload and tj.get("source_fingerprint") == source_fingerprint():
                traffic = tj.get("hbm_bytes_per_launch")
                traffic_source = "NOT measured in this run: profiles/traffic_latest.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of " \
                                 "this workload, kernel sources %s = the ones running here)" % tj.get("source_fingerprint")
        except Exception:
            traffic = None

    out = {
        "metric": "loop_trace_sites_per_sec", "value": value, "unit": "sites/s", "n_gpus": world, "steps": a.steps,
        "warmup": a.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64" if lprec == 8 else "f32", "data": "synthetic",
        "config": {"workload": workload, "local_lattice": list(X), "n_ev": nev, "n_gamma": 16,
                   "site_evecs_per_s": value * nev, "partition": "independent site blocks, one per rank"},
        "roofline": {"bound": "hbm", "kernel": "loop_contract_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                     "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": kern_ms,
                     "kernel_ms_stats": {"min": float(step_ms.min()), "median": float(np.median(step_ms)), "max": float(step_ms.max()),
                                         "mean": float(step_ms.mean()), "n": int(a.steps),
                                         "note": "per-launch HIP event times of the timed steps on rank 0; kernel_ms = their median "
                                                 "(max over ranks)"},
                     "probe_GBps": probe_gbs, "frac_of_probe": achieved / probe_gbs,
                     "probe_note": "probe = read_probe_kernel, 16-B non-temporal loads over the same eigenvector buffer, same process, "
                                   "right after the timed region (median of 5)"},
    }
    if world > 1:
        out["backend"] = backend
        out["nccl_ranks"] = dist.get_world_size() if backend == "nccl" else 0   # ranks RCCL saw (0: rehearsal over another backend)
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        base, cpu_loop, S = cpu_baseline(fields, sig, X, prec, order, a.cpu_seconds)
        # the same sample on the GPU result: the checker agrees with what was just timed
        g = loop.view(16, 2, V // 2)[:, :, :S].reshape(-1).cpu().numpy()
        err = float(np.max(np.abs(g - cpu_loop)) / np.max(np.abs(cpu_loop)))
        base["max_rel_err_gpu_vs_cpu_on_sample"] = err
        out["cpu_baseline"] = base

    # ---- extra legs: never part of `value`; a failure or a hang there must not cost the headline line ----------------------
    lock = threading.Lock()
    printed = threading.Event()
    running = {"leg": None}

    detail_path = a.detail_file or os.path.join("gpurun_out", "bench_detail_n%d.json" % world)

    def emit():
        with lock:
            if not printed.is_set():
                printed.set()
                if rank == 0:
                    try:
                        os.makedirs(os.path.dirname(os.path.abspath(detail_path)), exist_ok=True)
                        with open(detail_path, "w") as f:
                            json.dump(out, f, indent=1)
                        where = detail_path
                    except OSError as e:
                        where = "not written: %s" % e
                    sys.stdout.write(compact_line(out, where) + "\n")       # the LAST stdout line, and the only one starting with {"metric"
                    sys.stdout.flush()

    if not a.no_extra:
        del fields, big, loop
        torch.cuda.empty_cache()
        want = [w for w in a.extra.split(",") if w]
        legs = ([("displaced_loops", "displaced", lambda: extra_displaced(hip, device, a.displaced_nev)),
                 ("forced_partition_displaced_loops", "forced", lambda: extra_forced(hip, device, a.displaced_nev, a.emulate_link_GBps)),
                 ("strong_scaling_displaced_loops", "strong", lambda: extra_strong(hip, device, a, world, rank, backend)),
                 ("mg_coarse_loop", "mg", lambda: extra_mg(hip, device)),
                 ("cfg3_mixed_precision_ultra_local", "cfg3", lambda: extra_cfg3(hip, device))] if world == 1 else
                [("strong_scaling_displaced_loops", "strong", lambda: extra_strong(hip, device, a, world, rank, backend)),     # the speedup first
                 ("partitioned_displaced_loops", "partitioned", lambda: extra_partitioned(hip, device, a, world, rank, backend)),
                 ("native_rccl_transport", "native", lambda: extra_native(hip, device, a, world, rank, backend))])
        also = {}
        out["also_measured"] = also

        def on_timeout():
            # a leg hung (or is far slower than it should be): the headline line is still printed, but the process reports
            # the failure -- status 3 and the name of the leg that was running -- so that the hang is seen and can be traced
            # (the native-transport leg comes last and repeats a job that has already been measured on torch's transport: if IT
            # hangs, every contract number is in the line already and the process reports the leg, not a failure)
            code = 0 if running["leg"] == "native_rccl_transport" else 3
            with lock:
                also["error"] = "extra leg '%s' did not finish within %.0f s; headline unaffected; exit status %d" % (running["leg"], a.extra_timeout, code)
            emit()
            os._exit(code)

        dog = threading.Timer(a.extra_timeout, on_timeout)
        dog.daemon = True
        dog.start()
        for key, short, fn in legs:
            if want and short not in want:
                continue
            running["leg"] = key
            if short not in ("displaced", "forced") and _CFG2_INPUTS:
                _CFG2_INPUTS.clear()                      # the 102 GB of configs[2] eigenvectors are not needed any more
                torch.cuda.empty_cache()
            try:
                res = fn()
            except Exception as e:                        # reported, not raised: the headline above stands on its own
                res = {"error": "%s: %s" % (type(e).__name__, e)}
            with lock:
                also[key] = res
            if "error" in res and world > 1:
                break                                     # the ranks may be out of step: no further collective work
            torch.cuda.empty_cache()
        _CFG2_INPUTS.clear()
        dog.cancel()
        sd = also.get("strong_scaling_displaced_loops")
        if sd and "error" not in sd:
            # the same global problem at every N: seconds_1gpu / seconds is the speedup of the partitioned path
            h = sd.get("halo", {})
            G = [sd["grid"][d] * X_ for d, X_ in enumerate(sd["local_lattice"])]
            out["strong_scaling"] = {"workload": sd["workload"], "global_lattice": G, "n_ev": a.strong_nev, "grid": sd["grid"],
                                     "seconds_1gpu": sd.get("seconds_1gpu"), "seconds": sd["seconds"], "speedup": sd.get("speedup", 1.0 if world == 1 else None),
                                     "global_sites_per_s_all_slots": sd["sites_per_s_all_slots"],
                                     "halo_wait_ms_not_hidden": h.get("wait_ms_not_hidden"), "halo_GBps_per_rank": h.get("GBps_per_rank"),
                                     "max_rel_diff_vs_1gpu": sd.get("max_rel_diff_vs_1gpu")}
        pd = also.get("partitioned_displaced_loops")
        if world > 1 and pd and "error" not in pd:
            # the partitioned configs[2] job at the top level: what a scaling curve of THIS path would be drawn from
            h = pd.get("halo", {})
            out["process_grid"] = pd.get("grid")
            out["partitioned"] = {"workload": pd["workload"], "local_lattice": pd.get("local_lattice"), "n_ev": pd.get("n_ev"),
                                  "seconds": pd["seconds"], "sites_per_s_all_slots": pd["sites_per_s_all_slots"],
                                  "halo_bytes_sent_per_rank": h.get("bytes_sent_per_rank"), "halo_GBps_per_rank": h.get("GBps_per_rank"),
                                  "wait_ms_not_hidden": h.get("wait_ms_not_hidden")}
        checks = [v["parity_ok"] for v in also.values() if isinstance(v, dict) and "parity_ok" in v]
        if checks:
            out["parity_ok"] = bool(all(checks))
    emit()
    if dist is not None:
        try:
            dist.destroy_process_group()
        except Exception:
            pass
    if out.get("parity_ok") is False:
        # a partitioned run that does not reproduce the unpartitioned numbers is a failure of the product, not a footnote
        sys.stderr.write("bench.py: a partitioned leg differs from its unpartitioned reference by more than %g (see %s)\n" % (PARITY_TOL, detail_path))
        raise SystemExit(4)


if __name__ == "__main__":
    main()
