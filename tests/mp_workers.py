"""Worker functions for the multi-process tests (spawned with torch.multiprocessing; one process per rank).

`cpu_worker`  : world_size-2 gloo run on CPU -- exercises mugiq_amd.comm.GridComm (topology, face exchange,
                space-reduce / time-gather / broadcast) with the oracle doing the per-rank arithmetic.
`gpu_worker`  : the C++ driver (HIP kernels) on each rank, gloo transport with device buffers staged through
                the host, all ranks sharing cuda:0 of the one-GPU box.
Both compare against the single-domain oracle computed from the same seeded global fields.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def _global_problem(G, nev, seed):
    from util import random_gauge_lex, random_spinor_lex, sigmas
    rng = np.random.default_rng(seed)
    ev_lex = [random_spinor_lex(rng, G) for _ in range(nev)]
    U_lex = random_gauge_lex(rng, G)
    return ev_lex, U_lex, sigmas(nev)


def _single_domain_reference(orc, G, ev_lex, U_lex, sg, disp, moms, FTSign):
    cprm = orc.LoopComputeParam(*disp) if disp else orc.LoopComputeParam(doNonLocal=False)
    Uo = orc.extended_gauge_from_global(U_lex, (0, 0, 0, 0), (1, 1, 1, 1), (0, 0, 0, 0))
    pos = orc.compute_loop_position_space([orc.lex_to_eo(v, G) for v in ev_lex], sg, cprm, Uo, G)
    V = int(np.prod(G))
    locV3 = G[0] * G[1] * G[2]
    mp_ = orc.convert_idx_order_map_gamma(pos, cprm.nData, cprm.nLoop, 2, V // 2, G)
    mom = orc.momentum_projection_local(mp_, orc.phase_matrix(moms, locV3, FTSign, G, G), G[3], cprm.nData, locV3, len(moms))
    return cprm, pos, mom.reshape(len(moms), cprm.nLoop, 16, G[3])


def _init(rank, world, port, backend="gloo"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend, rank=rank, world_size=world)
    return dist


def _check_pos(orc, rank_coord, grid, G, l, cprm, pos_local, pos_global, tol):
    from util import rel_err
    Vg, Vl = int(np.prod(G)), int(np.prod(l))
    for idata in range(cprm.nData):
        glob = orc.eo_to_lex(pos_global[Vg * idata:Vg * (idata + 1)].reshape(2, Vg // 2), G)
        loc = orc.eo_to_lex(pos_local[Vl * idata:Vl * (idata + 1)].reshape(2, Vl // 2), l)
        e = rel_err(loc, orc.local_block(glob, rank_coord, grid))
        assert e < tol, ("dataPos", idata, e)


def cpu_worker(rank, world, port, grid, G=(4, 4, 4, 8), force=(0, 0, 0, 0)):
    """N>1 path on CPU: GridComm over gloo + oracle arithmetic == single-domain oracle.
    force: axes of extent 1 on which the partitioned path is forced (the rank is its own neighbour)."""
    import torch
    from util import orc, momenta_p2_le, rel_err
    dist = _init(rank, world, port)
    from mugiq_amd.comm import GridComm
    disp = (["+t", "-t", "+z", "-x"], [1, 1, 2, 1], [2, 1, 2, 1])
    moms = momenta_p2_le(2)
    FTSign = -1
    ev_lex, U_lex, sg = _global_problem(G, 2, 99)
    cprm, pos_g, mom_g = _single_domain_reference(orc, G, ev_lex, U_lex, sg, disp, moms, FTSign)

    comm = GridComm(grid, force_partitioned=force)
    assert comm.rank_of(comm.coord) == rank and comm.coords_of(rank) == comm.coord
    l = [G[d] // grid[d] for d in range(4)]
    commDim = [comm.comm_dim_partitioned(d) for d in range(4)]
    brd = [2 * c for c in commDim]
    U = orc.extended_gauge_from_global(U_lex, comm.coord, grid, brd)
    ev = [orc.lex_to_eo(orc.local_block(v, comm.coord, grid), l) for v in ev_lex]

    def ghost_exchange(v):
        """exchangeGhostVec: all partitioned dims, both directions, through GridComm.sendrecv"""
        gh = [[None, None] for _ in range(4)]
        for d in range(4):
            if not commDim[d]:
                continue
            for high in (0, 1):
                face = torch.from_numpy(np.ascontiguousarray(orc.pack_face(v, l, d, high)))
                recv = torch.empty_like(face)
                comm.sendrecv(face, recv, d, +1 if high else -1)       # low face -> backward nbr; high -> forward
                gh[d][1 - high] = recv.numpy()
        return gh

    pos = orc.compute_loop_position_space(ev, sg, cprm, U, l, commDim, brd, ghost_exchange)
    _check_pos(orc, comm.coord, grid, G, l, cprm, pos, pos_g, 1e-13)

    # momentum projection with the COMM_SPACE reduce / COMM_TIME gather / world bcast
    Vl = int(np.prod(l))
    locV3 = l[0] * l[1] * l[2]
    ph = orc.phase_matrix(moms, locV3, FTSign, l, G, comm.coord)
    mom_loc = orc.momentum_projection_local(orc.convert_idx_order_map_gamma(pos, cprm.nData, cprm.nLoop, 2, Vl // 2, l),
                                            ph, l[3], cprm.nData, locV3, len(moms))
    send = torch.from_numpy(mom_loc.view(np.float64).copy())
    red = torch.zeros_like(send)
    comm.reduce_space(send, red)
    full = torch.zeros(send.numel() * grid[3], dtype=send.dtype)
    comm.gather_time(red, full)
    comm.bcast(full)
    got = full.numpy().view(np.complex128).reshape(grid[3], len(moms), cprm.nLoop, 16, l[3])
    got = got.transpose(1, 2, 3, 0, 4).reshape(len(moms), cprm.nLoop, 16, G[3])
    assert rel_err(got, mom_g) < 1e-12
    dist.barrier()
    dist.destroy_process_group()


def gpu_worker(rank, world, port, grid, prec, order, calc_type, G=(4, 4, 8, 8), seed=None, force=(0, 0, 0, 0), case=None):
    """The C++ driver on every rank (all on cuda:0), halos and FT reduction through the comm callbacks.
    force: axes of extent 1 on which the partitioned path is forced (MugiqHipComm.partitioned; self-neighbour).
    case "pool_tie": stop * N_ev == the local extent of the partitioned axes, so the multi-layer halo buffers have exactly the size
    of a path-link field, and the entry that runs before the halos are posted has the larger stop (more link fields than any
    posted entry) -- the scratch pool must not hand that entry's fields to the pack stream while its kernels still run.
    case "pack": an entry along x first and four halos along z / t behind it: on a lattice the row tile of csrc/fused_mfma.hip takes
    (X0 = 8, X1 = 16) the first entry writes their face layers itself; MUGIQ_TEST_EXPECT_PACKED says how many the driver must report.
    case "long": entries of lengths 1 .. 8 (the matrix-pipe tile takes them as launches of three lengths over one axial gauge)."""
    import torch
    from util import orc, momenta_p2_le, rel_err
    dist = _init(rank, world, port)
    torch.cuda.set_device(0)
    import mugiq_amd as hip
    # the last entry is longer than the local t extent on a t-partitioned grid: the OPT plan hands it to the
    # step-by-step sequence (one halo per step) instead of the multi-layer halo
    # ... and "+t:5" after it has a computed opposite-sign source of the same length, which the OPT plan must NOT
    # reflect on a t-partitioned grid (length > local extent) but may on the others
    disp = (["+t", "-t", "+z", "-z", "+x", "-y", "-t", "+y", "+t"], [1, 2, 1, 1, 1, 2, 5, 1, 5], [3, 2, 2, 1, 1, 2, 5, 3, 5])
    if seed is not None:                                                      # seeded random entry list (same on every rank)
        r = np.random.default_rng(seed)
        n = int(r.integers(2, 7))
        names, lo, hi = [], [], []
        for _ in range(n):
            names.append("+-"[int(r.integers(2))] + "xyzt"[int(r.integers(4))])
            a_, b_ = sorted(int(v) for v in r.integers(1, 6, size=2))
            lo.append(a_)
            hi.append(b_)
        disp = (names, lo, hi)
    moms = momenta_p2_le(2)
    FTSign = 1
    nev = 3
    if case == "pool_tie":
        nev = 4
        disp = (["+x", "+t", "-z", "+y", "-t"], [1, 1, 1, 2, 2], [3, 2, 2, 2, 2])
        assert all(2 * nev == G[d] // grid[d] for d in (2, 3))
    if case == "pack":                                                        # "+x" runs first and writes the face layers of the four halos
        disp = (["+x", "+t", "-z", "+z", "-t", "+y"], [1, 1, 1, 2, 2, 1], [2, 3, 2, 3, 4, 1])   # (no entry derivable from another)
    if case == "long":                                                        # the reference's own example: lengths 1 .. 8
        nev = 2
        disp = (["+x", "-x", "+y", "-z", "+t", "-t", "+z"], [1, 1, 1, 1, 1, 1, 1], [8, 7, 8, 5, 8, 4, 8])
    ev_lex, U_lex, sg = _global_problem(G, nev, 1234)
    cdt = np.complex128 if prec == 8 else np.complex64
    ev_lex = [v.astype(cdt).astype(np.complex128) for v in ev_lex]           # the inputs the GPU sees
    U_lex = U_lex.astype(cdt).astype(np.complex128)
    cprm, pos_g, mom_g = _single_domain_reference(orc, G, ev_lex, U_lex, sg, disp, moms, FTSign)

    comm = hip.GridComm(grid, device="cuda:0", force_partitioned=force)
    l = [G[d] // grid[d] for d in range(4)]
    brd = [2 * comm.comm_dim_partitioned(d) for d in range(4)]
    # Displace's setup: host QDP links of the LOCAL lattice -> extended device field, borders from the neighbours
    U_loc = np.stack([orc.lex_to_eo(orc.local_block(U_lex[mu], comm.coord, grid), l) for mu in range(4)])
    gauge = hip.GaugeField(l, brd, prec).set_from_qdp_host(orc.gauge_to_qdp_host(U_loc), comm)
    exp_ext = orc.extended_gauge_from_global(U_lex, comm.coord, grid, brd).astype(cdt)
    assert np.array_equal(gauge.get_logical(), exp_ext), "extended gauge (borders, edges, corners) differs from the global field"
    f = [hip.SpinorField(l, prec, order).set_logical(orc.lex_to_eo(orc.local_block(v, comm.coord, grid), l)) for v in ev_lex]
    prm = hip.MugiqLoopParam(Nmom=len(moms), momMatrix=[list(m) for m in moms], FTSign=FTSign, calcType=calc_type,
                             doMomProj=True, doNonLocal=True, disp_entry=[], disp_str=disp[0], disp_start=disp[1],
                             disp_stop=disp[2], gauge=gauge)
    loop = hip.Loop_Mugiq(prm, f, sg, comm)
    assert loop.nLoop == cprm.nLoop and loop.totT == G[3] and loop.locT == l[3]
    loop.computeCoarseLoop()
    tol = 1e-12 if prec == 8 else 1e-5
    _check_pos(orc, comm.coord, grid, G, l, cprm, loop.dataPos_d.cpu().numpy().astype(np.complex128), pos_g, tol)
    e = rel_err(loop.dataMom_global(), mom_g)
    assert e < tol, ("dataMom", e)
    if case == "pack":
        assert loop.halosPackedInEntry() == int(os.environ["MUGIQ_TEST_EXPECT_PACKED"]), loop.halosPackedInEntry()
    if seed is None and calc_type == hip.LOOP_CALC_TYPE_BASIC_KERNEL:
        # the reference's own nest for two entries, call for call through the Displace mirror (its exchangeGhostVec goes through
        # `comm`, its extended gauge is built from the host QDP links): same slots as the driver's
        from types import SimpleNamespace
        got = loop.dataPos_d.clone()
        displace = hip.Displace(SimpleNamespace(gauge=None, gauge_qdp=orc.gauge_to_qdp_host(U_loc)), f[0], prec, comm)
        per = 16 * int(np.prod(l))
        R = hip.SpinorField(l, prec, order)
        for idx in (0, 3):                                                  # "+t": 1..3 and "-z": 1
            displace.setupDisplacement(disp[0][idx])
            mine = torch.zeros(per * cprm.nLoopPerEntry[idx], dtype=got.dtype, device=got.device)
            for n in range(nev):
                R.data.copy_(f[n].data)
                cnt = 0
                for idisp in range(1, cprm.dispStop[idx] + 1):
                    displace.doVectorDisplacement(hip.DISPLACE_TYPE_COVARIANT, R, idisp)
                    if idisp >= cprm.dispStart[idx]:
                        hip.performLoopContraction(mine[per * cnt:per * (cnt + 1)], f[n], R, sg[n])
                        cnt += 1
            ref_slots = got[per * cprm.nLoopOffset[idx]:per * (cprm.nLoopOffset[idx] + cprm.nLoopPerEntry[idx])]
            e = rel_err(mine.cpu().numpy(), ref_slots.cpu().numpy())
            assert e < tol, ("Displace class nest", idx, e)
    loop.close()
    dist.barrier()
    dist.destroy_process_group()


def nccl_world1_worker(rank, world, port, loopback):
    """ONE rank on the nccl (= RCCL) backend: every GridComm operation with the buffers and on the streams the driver
    uses them with -- host payloads of the FT reduction through _wire (device round trip + RCCL reduce / all_gather /
    broadcast), device halos through the C callbacks on a side stream (the ExternalStream branch), a transfer group, and
    a whole driver run with a comm handed in.  loopback: also send the halo to self through RCCL's isend / irecv."""
    import ctypes
    import torch
    from util import orc, momenta_p2_le, rel_err
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["MUGIQ_HIP_SELF_HALO_COPY"] = "1"      # the halo to self goes through the transport (default: packed in place)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    import mugiq_amd as hip
    comm = hip.GridComm((1, 1, 1, 1), device=dev)
    assert comm.backend == "nccl" and comm.is_time_process and comm.space_root == 0
    for dt in (torch.float64, torch.float32):                       # reduce_space / gather_time / bcast: lib/loop_mugiq.cpp:406-424
        send = torch.arange(1, 4097, dtype=dt) / 7
        red = torch.zeros_like(send)
        comm.reduce_space(send, red)
        assert torch.equal(red, send)
        full = torch.zeros_like(send)
        comm.gather_time(red, full)
        assert torch.equal(full, send)
        b = full.clone()
        comm.bcast(b)
        assert torch.equal(b, send)
    # device halos through the C callbacks, on a stream that is NOT torch's current one
    c = comm.c_struct()
    side = torch.cuda.Stream()
    n = 1 << 20
    a = torch.randn(n, dtype=torch.float64, device=dev)
    r = torch.zeros_like(a)
    torch.cuda.synchronize()
    comm.loopback_through_transport = bool(loopback)
    with torch.cuda.stream(side):
        a2 = a * 2                                                   # producer on the side stream; the exchange must order after it
    assert c.sendrecv(None, a2.data_ptr(), r.data_ptr(), n * 8, 3, +1, side.cuda_stream) == 0
    side.synchronize()
    assert torch.equal(r, a * 2)
    r.zero_()
    r2 = torch.zeros_like(a)
    assert c.group_begin(None) == 0                                  # two halos of different axes in one transfer group
    assert c.sendrecv(None, a.data_ptr(), r.data_ptr(), n * 8, 3, -1, side.cuda_stream) == 0
    assert c.sendrecv(None, a2.data_ptr(), r2.data_ptr(), n * 8, 2, +1, side.cuda_stream) == 0
    assert c.group_end(None, side.cuda_stream) == 0
    side.synchronize()
    assert torch.equal(r, a) and torch.equal(r2, a * 2)
    comm.loopback_through_transport = False
    # the driver with a communicator on a 1x1x1x1 grid == the driver without one == the oracle
    X = (4, 4, 4, 8)
    ev_lex, U_lex, sg = _global_problem(X, 3, 5)
    moms = momenta_p2_le(2)
    disp = (["+t", "-t", "+z"], [1, 1, 2], [2, 2, 2])
    cprm, pos_g, mom_g = _single_domain_reference(orc, X, ev_lex, U_lex, sg, disp, moms, -1)
    U_loc = np.stack([orc.lex_to_eo(U_lex[mu], X) for mu in range(4)])
    gauge = hip.GaugeField(X, (0, 0, 0, 0), 8).set_from_qdp_host(orc.gauge_to_qdp_host(U_loc), comm)
    f = [hip.SpinorField(X, 8, 2).set_logical(orc.lex_to_eo(v, X)) for v in ev_lex]
    prm = hip.MugiqLoopParam(Nmom=len(moms), momMatrix=[list(m) for m in moms], FTSign=-1, doMomProj=True, doNonLocal=True,
                             disp_str=disp[0], disp_start=disp[1], disp_stop=disp[2], gauge=gauge)
    loop = hip.Loop_Mugiq(prm, f, sg, comm).setProfiling()
    loop.computeCoarseLoop()
    assert rel_err(loop.dataPos_d.cpu().numpy(), pos_g) < 1e-12
    assert rel_err(loop.dataMom_global(), mom_g) < 1e-12
    kinds = [p["kind"] for p in loop.phases()]
    # (the ultra-local loop has a phase of its own unless a displaced entry carried it along as a fourth slot)
    assert ("ultra_local" in kinds or "entry_fused" in kinds) and "momentum_projection" in kinds and kinds[-1] == "total_wall", kinds
    loop.close()
    dist.barrier()
    dist.destroy_process_group()


def forced_full_size_worker(rank, world, port, X, nev, force, backend):
    """BASELINE.json configs[2]'s per-GPU lattice on ONE rank with the partitioned path forced on `force` axes (the rank is
    its own neighbour): packed face layers, ghost-layer reads, interior / boundary tiles, halos of reflected slots and gauge
    borders through sendrecv at the real size, against an independent path -- the unpartitioned run of the same job, which
    wraps around inside the kernels.  The oracle is far too slow at this size; small lattices of the same construction are
    checked against it by test_forced_partitioning_on_one_rank."""
    import torch
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    import mugiq_amd as hip
    from bench import make_evecs, make_gauge, momenta_p2_le, ENTRIES_CFG2
    _, f = make_evecs(hip, X, nev, 8, 2, dev, seed=11)
    sg = 0.01 + 0.002 * np.arange(nev)
    moms = momenta_p2_le(2)
    comm = hip.GridComm((1, 1, 1, 1), device=dev, force_partitioned=force)
    g_plain = make_gauge(hip, X, 8, dev, 4321, None)            # written in place, no border: shifts wrap inside the kernels
    g_part = make_gauge(hip, X, 8, dev, 4321, comm)             # host QDP links -> borders R = 2 through sendrecv (self)
    assert list(g_part.R) == [2 * int(bool(x)) for x in force]

    def run(gauge, cm, calc, entries, n):
        prm = hip.MugiqLoopParam(gauge=gauge, calcType=calc, doMomProj=True, momMatrix=moms, Nmom=len(moms), FTSign=-1)
        prm.set_displace_entry_string(entries)
        loop = hip.Loop_Mugiq(prm, f[:n], sg[:n], cm).setProfiling()
        loop.computeCoarseLoop()
        pos, mom = loop.dataPos_d.clone(), np.array(loop.dataMom_global())
        kinds = set(p["kind"] for p in loop.phases())
        der = [loop.derivedFrom(i) for i in range(loop.nDispEntries)]
        loop.close()
        return pos, mom, kinds, der

    ref_pos, ref_mom, _, ref_der = run(g_plain, None, hip.LOOP_CALC_TYPE_OPT_KERNEL, ENTRIES_CFG2, nev)
    scale, mscale = float(ref_pos.abs().max()), float(np.abs(ref_mom).max())
    for ahead, copy in (("1", "0"), ("0", "0"), ("1", "1"), ("0", "1")):   # copy 0: face layers packed straight into the ghost buffer
        os.environ["MUGIQ_HIP_HALO_AHEAD"] = ahead
        os.environ["MUGIQ_HIP_SELF_HALO_COPY"] = copy
        pos, mom, kinds, der = run(g_part, comm, hip.LOOP_CALC_TYPE_OPT_KERNEL, ENTRIES_CFG2, nev)
        assert der == ref_der
        assert {"halo_transfer", "entry_interior", "entry_boundary"} <= kinds, kinds
        e = float((pos - ref_pos).abs().max()) / scale
        assert e < 1e-13, ("forced partition, position space, halo ahead " + ahead, e)
        em = float(np.abs(mom - ref_mom).max()) / mscale
        assert em < 1e-12, ("forced partition, momentum space", em)
        del pos
    os.environ.pop("MUGIQ_HIP_HALO_AHEAD", None)
    os.environ.pop("MUGIQ_HIP_SELF_HALO_COPY", None)
    # a second compute on the SAME loop object (position space only): pooled halo buffers, per-block events and link fields are reused
    prm = hip.MugiqLoopParam(gauge=g_part, calcType=hip.LOOP_CALC_TYPE_OPT_KERNEL).set_displace_entry_string(ENTRIES_CFG2)
    loop = hip.Loop_Mugiq(prm, f, sg, comm)
    loop.computeCoarseLoop()
    first = loop.dataPos_d.clone()
    loop.computeCoarseLoop()
    assert torch.equal(loop.dataPos_d, first), "second compute on the same loop object differs"
    e = float((first - ref_pos).abs().max()) / scale
    assert e < 1e-13, ("forced partition, no projection", e)
    loop.close()
    del first
    # the reference's own sequence (BASIC: one face exchange per step and eigenvector) on a subset
    sub = "+z:1,2;-t:1;+x:1"
    b_pos, b_mom, _, _ = run(g_part, comm, hip.LOOP_CALC_TYPE_BASIC_KERNEL, sub, 2)
    o_pos, o_mom, _, _ = run(g_plain, None, hip.LOOP_CALC_TYPE_OPT_KERNEL, sub, 2)
    e = float((b_pos - o_pos).abs().max()) / float(o_pos.abs().max())
    assert e < 1e-13, ("forced partition, BASIC", e)
    assert float(np.abs(b_mom - o_mom).max()) / float(np.abs(o_mom).max()) < 1e-12
    dist.barrier()
    dist.destroy_process_group()


def native_rccl_worker(rank, world, port):
    """one rank, the library's own RCCL transport, forced partitioning on z and t; against the unpartitioned run and GridComm"""
    import torch
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    os.environ["MUGIQ_HIP_SELF_HALO_COPY"] = "1"      # the halo to self goes through the transport (default: packed in place)
    import mugiq_amd as hip
    from bench import make_evecs, make_gauge, momenta_p2_le
    X, nev = (8, 8, 8, 8), 4
    _, f = make_evecs(hip, X, nev, 8, 2, dev, seed=5)
    sg = 0.01 + 0.002 * np.arange(nev)
    moms = momenta_p2_le(2)
    entries = "+z:1,3;-z:1,3;+t:1,2;-t:2,3;+x:1,2;-y:1"

    def run(gauge, comm):
        prm = hip.MugiqLoopParam(gauge=gauge, doMomProj=True, momMatrix=moms, Nmom=len(moms), FTSign=-1).set_displace_entry_string(entries)
        loop = hip.Loop_Mugiq(prm, f, sg, comm).setProfiling()
        loop.computeCoarseLoop()
        out = loop.dataPos_d.clone(), np.array(loop.dataMom_global()), set(p["kind"] for p in loop.phases())
        loop.close()
        return out

    ref_pos, ref_mom, _ = run(make_gauge(hip, X, 8, dev, 321, None), None)
    native = hip.RcclComm((1, 1, 1, 1), device=dev, force_partitioned=(0, 0, 1, 1))
    assert native.comm_dim_partitioned(3) == 1 and native.coord == (0, 0, 0, 0) and native.c_struct().sendrecv
    g_native = make_gauge(hip, X, 8, dev, 321, native)            # borders R = 2 through the native sendrecv
    pos, mom, kinds = run(g_native, native)
    assert {"halo_transfer", "entry_interior", "entry_boundary"} <= kinds, kinds
    scale = float(ref_pos.abs().max())
    assert float((pos - ref_pos).abs().max()) < 1e-13 * scale
    assert float(np.abs(mom - ref_mom).max()) < 1e-12 * float(np.abs(ref_mom).max())
    grid = hip.GridComm((1, 1, 1, 1), device=dev, force_partitioned=(0, 0, 1, 1))
    g_grid = make_gauge(hip, X, 8, dev, 321, grid)
    assert torch.equal(g_grid.data, g_native.data), "gauge borders differ between the two transports"
    pos2, mom2, _ = run(g_grid, grid)
    assert torch.equal(pos2, pos), "loops differ between the native and the torch transport"
    native.close()
