"""GPU tests of the C++ driver (Loop_Mugiq / Displace mirror): slot bookkeeping, both execution plans
(BASIC = the reference's launch sequence, OPT = batched + fused), momentum projection, and the
domain-decomposed run (2 ranks sharing the one GPU of the box, gloo transport through the comm callbacks)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

import mp_workers
from test_multi_rank_cpu import free_port
from util import orc, random_gauge_lex, random_spinor_lex, sigmas, momenta_p2_le, rel_err

pytestmark = pytest.mark.gpu


def _setup(hip, X, nev, prec, order, seed, pad=0, gpad=0):
    rng = np.random.default_rng(seed)
    cdt = np.complex128 if prec == 8 else np.complex64
    ev = [orc.lex_to_eo(random_spinor_lex(rng, X), X).astype(cdt).astype(np.complex128) for _ in range(nev)]
    Uo = orc.extended_gauge_from_global(random_gauge_lex(rng, X), (0, 0, 0, 0), (1, 1, 1, 1), (0, 0, 0, 0))
    Uo = Uo.astype(cdt).astype(np.complex128)
    f = [hip.SpinorField(X, prec, order, pad=pad).set_logical(v) for v in ev]
    U = hip.GaugeField(X, (0, 0, 0, 0), prec, pad=gpad).set_logical(Uo)
    return ev, Uo, f, U


@pytest.mark.parametrize("prec,order", [(8, 2), (4, 4)])
def test_reference_loop_nest_through_displace_class(hip, prec, order):
    """The loop nest of Loop_Mugiq::computeCoarseLoop (lib/loop_mugiq.cpp:455-509) written down call for call with the
    mirrors of the reference's own objects -- Displace::setupDisplacement / doVectorDisplacement (lib/displace.cpp:55-67,
    206-223) and performLoopContraction (lib/contract_wrappers.cu:88-115) -- must give what the driver gives (both plans)
    and what the oracle gives."""
    X, nev = (4, 6, 4, 8), 3
    ev, Uo, f, U = _setup(hip, X, nev, prec, order, 555)
    sg = sigmas(nev)
    entry = "+z:1,2;-x:2;+t:3,1;-y:1,3"
    prm = hip.MugiqLoopParam(gauge=U).set_displace_entry_string(entry)
    _, s, a, b = orc.parse_disp_entry_string(entry)
    cprm = orc.LoopComputeParam(s, a, b)
    V = int(np.prod(X))
    cdt = torch.complex128 if prec == 8 else torch.complex64
    dataPos_d = torch.zeros(16 * V * cprm.nLoop, dtype=cdt, device="cuda")
    nElemPosLocPerLoop = 16 * V
    displace = hip.Displace(prm, f[0], prec)
    fineEvecL, fineEvecR = hip.SpinorField(X, prec, order), hip.SpinorField(X, prec, order)
    for idx in range(-1, cprm.nDispEntries):
        if idx != -1:
            displace.setupDisplacement(cprm.dispString[idx])
        bufOffset = 0 if idx == -1 else nElemPosLocPerLoop * cprm.nLoopOffset[idx]
        for n in range(nev):
            fineEvecL.data.copy_(f[n].data)
            fineEvecR.data.copy_(fineEvecL.data)
            if idx == -1:
                hip.performLoopContraction(dataPos_d, fineEvecL, fineEvecR, sg[n])
                continue
            dispCount = 0
            for idisp in range(1, cprm.dispStop[idx] + 1):
                displace.doVectorDisplacement(hip.DISPLACE_TYPE_COVARIANT, fineEvecR, idisp)
                if cprm.dispStart[idx] <= idisp <= cprm.dispStop[idx]:
                    off = bufOffset + nElemPosLocPerLoop * dispCount
                    hip.performLoopContraction(dataPos_d[off:off + nElemPosLocPerLoop], fineEvecL, fineEvecR, sg[n])
                    dispCount += 1
    ref = orc.compute_loop_position_space(ev, np.float32(sg).astype(np.float64) if prec == 4 else sg, cprm, Uo, X)
    tol = 1e-12 if prec == 8 else 1e-5
    got = dataPos_d.cpu().numpy()
    assert rel_err(got, ref) < tol
    for calc in (hip.LOOP_CALC_TYPE_BASIC_KERNEL, hip.LOOP_CALC_TYPE_OPT_KERNEL):
        loop = hip.Loop_Mugiq(hip.MugiqLoopParam(gauge=U, calcType=calc).set_displace_entry_string(entry), f, sg)
        loop.computeCoarseLoop()
        assert rel_err(loop.dataPos_d.cpu().numpy(), got) < tol
        loop.close()
    with pytest.raises(hip.MugiqHipError):
        displace.setupDisplacement("+w")
    with pytest.raises(hip.MugiqHipError):
        displace.doVectorDisplacement(1, fineEvecR, 1)


@pytest.mark.parametrize("prec,order", [(8, 2), (8, 4), (4, 2), (4, 4)])
@pytest.mark.parametrize("calc", ["basic", "opt"])
def test_driver_single_process_vs_oracle(hip, prec, order, calc):
    X = (4, 6, 4, 8)
    nev = 5
    ev, Uo, f, U = _setup(hip, X, nev, prec, order, 321)
    sg = sigmas(nev)
    entry = "+z:1,2;-x:2;+t:3,1;-y:1;-t:1,5"               # includes start > stop (swapped) and 5 slots (> 4 per launch)
    prm = hip.MugiqLoopParam(FTSign=-1, doMomProj=True, gauge=U,
                             calcType=hip.LOOP_CALC_TYPE_BASIC_KERNEL if calc == "basic" else hip.LOOP_CALC_TYPE_OPT_KERNEL)
    prm.set_displace_entry_string(entry)
    moms = momenta_p2_le(3)
    prm.momMatrix, prm.Nmom = [list(m) for m in moms], len(moms)
    loop = hip.Loop_Mugiq(prm, f, sg)
    _, s, a, b = orc.parse_disp_entry_string(entry)
    cprm = orc.LoopComputeParam(s, a, b)
    assert (loop.nLoop, loop.nData, loop.nDispEntries) == (cprm.nLoop, cprm.nData, cprm.nDispEntries)
    for i in range(cprm.nDispEntries):
        d, sgn = orc.parse_displacement(cprm.dispString[i])
        assert loop.entry(i) == (d, sgn, cprm.dispStart[i], cprm.dispStop[i], cprm.nLoopPerEntry[i], cprm.nLoopOffset[i])
    V = int(np.prod(X))
    assert loop.nElemPosLoc == 16 * V * cprm.nLoop and loop.nElemMomTot == 16 * len(moms) * X[3] * cprm.nLoop
    lines = []
    loop.printLoopComputeParams(lines.append)                       # lib/loop_mugiq.cpp:233-273
    assert "Precision is %s" % ("double" if prec == 8 else "single") in lines and "Will NOT use Multigrid" in lines
    assert "  0: +z with lengths from 1 to 2, #loops = 2, loop-offset = 1" in lines
    assert "  1: -x with length 2, #loops = 1, loop-offset = 3" in lines
    assert "  2: +t with lengths from 1 to 3, #loops = 3, loop-offset = 4" in lines      # start > stop was swapped
    assert "Total number of Loop Traces to perform: %d" % cprm.nLoop in lines and "Local  3d volume: %d" % (X[0] * X[1] * X[2]) in lines
    loop.computeCoarseLoop()
    ref_pos = orc.compute_loop_position_space(ev, sg, cprm, Uo, X)
    tol = 1e-12 if prec == 8 else 1e-5
    assert rel_err(loop.dataPos_d.cpu().numpy(), ref_pos) < tol
    locV3 = X[0] * X[1] * X[2]
    ref_mom = orc.momentum_projection_local(orc.convert_idx_order_map_gamma(ref_pos, cprm.nData, cprm.nLoop, 2, V // 2, X),
                                            orc.phase_matrix(moms, locV3, -1, X, X), X[3], cprm.nData, locV3, len(moms))
    assert rel_err(loop.dataMom_bcast, ref_mom) < tol
    loop.close()


def test_driver_ultralocal_only_and_errors(hip):
    X = (4, 4, 4, 4)
    ev, Uo, f, U = _setup(hip, X, 3, 8, 2, 5)
    sg = sigmas(3)
    loop = hip.Loop_Mugiq(hip.MugiqLoopParam(), f, sg)          # no mom-proj, no displacements
    assert loop.nLoop == 1 and loop.dataMom_bcast is None
    loop.computeCoarseLoop()
    ref = orc.compute_loop_position_space(ev, sg, orc.LoopComputeParam(doNonLocal=False))
    assert rel_err(loop.dataPos_d.cpu().numpy(), ref) < 1e-13
    loop.close()
    with pytest.raises(hip.MugiqHipError):                       # displacements without a gauge field
        hip.Loop_Mugiq(hip.MugiqLoopParam().set_displace_entry_string("+x:1"), f, sg)
    with pytest.raises(hip.MugiqHipError):                       # unparsable displacement string
        hip.Loop_Mugiq(hip.MugiqLoopParam(doNonLocal=True, disp_str=["+w"], disp_start=[1], disp_stop=[1], gauge=U), f, sg)
    with pytest.raises(hip.MugiqHipError):                       # mismatched limits
        hip.Loop_Mugiq(hip.MugiqLoopParam(doNonLocal=True, disp_str=["+x"], disp_start=[1, 2], disp_stop=[1], gauge=U), f, sg)
    with pytest.raises(hip.MugiqHipError):                       # momentum projection without momenta
        hip.Loop_Mugiq(hip.MugiqLoopParam(doMomProj=True), f, sg)


@pytest.mark.parametrize("order", [2, 4])
def test_fused_operator_with_ghost_layers(hip, order):
    """Operator-level check of the fused kernel across a domain boundary (2 domains emulated on one GPU):
    path links from E_k = D^k E_0 per domain, 3 ghost layers packed by pack_face_layers."""
    G = (4, 4, 4, 8)
    grid = (1, 1, 1, 2)
    l = (4, 4, 4, 4)
    comm = (0, 0, 0, 1)
    brd = (0, 0, 0, 2)
    rng = np.random.default_rng(8)
    nev = 3
    ev_lex = [random_spinor_lex(rng, G) for _ in range(nev)]
    U_lex = random_gauge_lex(rng, G)
    sg = sigmas(nev)
    ranks = [(0, 0, 0, 0), (0, 0, 0, 1)]
    for dispstr in ("+t", "-t"):
        dirn, sign = orc.parse_displacement(dispstr)
        cprm = orc.LoopComputeParam([dispstr], [1], [3])
        ref = orc.compute_loop_position_space([orc.lex_to_eo(v, G) for v in ev_lex], sg, cprm,
                                              orc.extended_gauge_from_global(U_lex, (0, 0, 0, 0), (1, 1, 1, 1), (0, 0, 0, 0)), G)
        f = {r: [hip.SpinorField(l, 8, order).set_logical(orc.lex_to_eo(orc.local_block(v, r, grid), l)) for v in ev_lex] for r in ranks}
        Ue = {r: hip.GaugeField(l, brd, 8).set_logical(orc.extended_gauge_from_global(U_lex, r, grid, brd)) for r in ranks}
        high = 0 if sign == hip.DispSignPlus else 1
        # path links per rank (needs the depth-1 face of E_{k-1} from the neighbour at every step)
        E = {r: [hip.SpinorField(l, 8, 2) for _ in range(4)] for r in ranks}
        ident = np.zeros((2, 128, 4, 3), dtype=np.complex128)
        for s in range(3):
            ident[:, :, s, s] = 1.0
        for r in ranks:
            E[r][0].set_logical(ident)
        for k in range(1, 4):
            faces = {}
            for r in ranks:
                faces[r] = torch.zeros(24 * E[r][k - 1].face_cb(3), dtype=torch.complex128, device="cuda")
                hip.packFace(faces[r], E[r][k - 1], 3, high)
            for i, r in enumerate(ranks):
                E[r][k - 1].ghost[3][1 - high] = faces[ranks[1 - i]]
                hip.performCovariantDisplacementVector(E[r][k], E[r][k - 1], Ue[r], dirn, sign, comm)
        layers = {}
        for r in ranks:
            layers[r] = torch.zeros(nev * 3 * 24 * f[r][0].face_cb(3), dtype=torch.complex128, device="cuda")
            hip.packFaceLayers(layers[r], f[r], 3, high, 3)
        Vl, Vg = 256, 512
        for i, r in enumerate(ranks):
            out = torch.zeros(3 * 16 * Vl, dtype=torch.complex128, device="cuda")
            hip.displacedLoopContractionFused(out, f[r], sg, E[r][1:], [1, 2, 3], dirn, sign, comm, layers[ranks[1 - i]], 3)
            got = out.cpu().numpy()
            for k in range(3):
                for ig in range(16):
                    gl = orc.eo_to_lex(ref[Vg * (16 * (1 + k) + ig):Vg * (16 * (1 + k) + ig + 1)].reshape(2, Vg // 2), G)
                    lo = orc.eo_to_lex(got[Vl * (16 * k + ig):Vl * (16 * k + ig + 1)].reshape(2, Vl // 2), l)
                    assert rel_err(lo, orc.local_block(gl, r, grid)) < 1e-13, (dispstr, r, k, ig)


@pytest.mark.parametrize("prec,order,lengths", [(8, 2, [1, 2, 3]), (8, 2, [2]), (8, 4, [1, 2]), (4, 4, [1, 3]), (4, 2, [1, 2, 3]), (8, 4, [1, 2, 3])])
@pytest.mark.parametrize("dispstr", ["+y", "-t", "+x"])
def test_fused_operator_carries_the_ultra_local_loop(hip, prec, order, lengths, dispstr):
    """mugiq_hip_displaced_loop_contraction_fused_carry: where the tiled kernel has room (a free slot of its 12-wave forms, the
    fourth slot of the 16-wave fp64 FLOAT2 form) the ultra-local loop comes out of the same pass and equals the loop of
    performLoopContractionBatched; where it has not, `carried` is False and the slot is left alone.  The displaced slots are the
    same either way."""
    X, nev = (8, 8, 4, 8), 5
    ev, Uo, f, U = _setup(hip, X, nev, prec, order, 99)
    sg = sigmas(nev)
    dirn, sign = orc.parse_displacement(dispstr)
    V = int(np.prod(X))
    cdt = torch.complex128 if prec == 8 else torch.complex64
    E = [hip.SpinorField(X, prec, 2) for _ in range(max(lengths) + 1)]
    ident = np.zeros((2, V // 2, 4, 3), dtype=np.complex128)
    for s_ in range(3):
        ident[:, :, s_, s_] = 1.0
    E[0].set_logical(ident)
    for k in range(1, len(E)):
        hip.performCovariantDisplacementVector(E[k], E[k - 1], U, dirn, sign)
    links = [E[k] for k in lengths]
    plain = torch.zeros(len(lengths) * 16 * V, dtype=cdt, device="cuda")
    hip.displacedLoopContractionFused(plain, f, sg, links, lengths, dirn, sign)
    both = torch.zeros_like(plain)
    ultra = torch.full((16 * V,), 7.0, dtype=cdt, device="cuda")       # (accumulated into: the 7 must survive underneath)
    carried = hip.displacedLoopContractionFused(both, f, sg, links, lengths, dirn, sign, ultraLocalSlot_d=ultra)
    assert torch.equal(both, plain)
    if carried:
        ref = torch.full((16 * V,), 7.0, dtype=cdt, device="cuda")
        hip.performLoopContractionBatched(ref, f, f, sg)
        tol = 1e-13 if prec == 8 else 1e-5
        assert rel_err(ultra.cpu().numpy(), ref.cpu().numpy()) < tol
    else:
        assert torch.all(ultra == 7.0)
    if prec == 8 and order == 2 and dirn >= 1:
        assert carried                                     # the forms this was built for: fp64 FLOAT2 column tiles, up to three lengths


@pytest.mark.parametrize("grid,prec,order,calc", [((1, 1, 1, 2), 8, 2, 1), ((1, 1, 2, 1), 8, 4, 1), ((1, 1, 1, 2), 8, 2, 2),
                                                  ((1, 1, 2, 1), 4, 4, 2), ((2, 1, 1, 1), 4, 2, 1), ((1, 2, 1, 1), 8, 2, 2)])
def test_two_rank_driver_on_one_gpu(grid, prec, order, calc):
    mp.spawn(mp_workers.gpu_worker, args=(2, free_port(), grid, prec, order, calc), nprocs=2, join=True)


@pytest.mark.parametrize("grid,G,prec,order", [((1, 1, 1, 2), (4, 4, 8, 16), 8, 2), ((1, 1, 2, 1), (4, 8, 16, 4), 8, 4),
                                               ((1, 2, 1, 1), (4, 16, 4, 4), 4, 4)])
def test_two_rank_driver_interior_boundary_tiles(grid, G, prec, order):
    """Local extent 8 along the partitioned axis: the tiled kernel runs its INTERIOR tiles while the halo is in flight and
    its BOUNDARY tiles afterwards (two tiles along the axis)."""
    mp.spawn(mp_workers.gpu_worker, args=(2, free_port(), grid, prec, order, 2, G), nprocs=2, join=True)


@pytest.mark.parametrize("seed", range(int(os.environ.get("MUGIQ_TEST_MP_SEEDS", 4))))
def test_two_rank_driver_random_entries(seed):
    """Seeded random entry lists (both signs, lengths 1..5 against a local extent of 4: reflected entries with and without
    halo, lengths past the neighbour) on a randomly chosen partitioned axis; OPT plan, every third case BASIC."""
    axis = (seed * 7 + 3) % 4
    grid = tuple(2 if d == axis else 1 for d in range(4))
    prec, order = [(8, 2), (4, 4), (8, 4), (4, 2)][seed % 4]
    calc = 2 if seed % 3 == 2 else 1                                  # every third case through the BASIC plan
    mp.spawn(mp_workers.gpu_worker, args=(2, free_port(), grid, prec, order, calc, (8, 8, 8, 8), 9000 + seed), nprocs=2, join=True)


def test_four_rank_driver_z_and_t_partitioned_on_one_gpu():
    """configs[2]'s kind of grid (z and t partitioned): the halos of the two axes are posted together at the start of
    the compute and travel in one transfer group (GridComm.group_begin / group_end)."""
    mp.spawn(mp_workers.gpu_worker, args=(4, free_port(), (1, 1, 2, 2), 8, 2, 1), nprocs=4, join=True)


@pytest.mark.parametrize("grid,G,prec,order,calc,ahead", [((1, 1, 1, 4), (4, 4, 4, 16), 8, 2, 1, "1"), ((1, 1, 1, 4), (4, 4, 4, 16), 8, 2, 1, "0"),
                                                          ((1, 1, 1, 4), (4, 4, 4, 16), 4, 4, 2, "1"), ((1, 1, 4, 1), (4, 4, 16, 4), 8, 4, 1, "1"),
                                                          ((4, 1, 1, 1), (16, 4, 4, 4), 4, 2, 1, "0")])
def test_four_rank_driver_extent_four_grid_on_one_gpu(grid, G, prec, order, calc, ahead, monkeypatch):
    """A process grid with extent 4 along one axis (configs[2] is 1x1x2x4): the forward and the backward neighbour are
    DIFFERENT ranks, so a swapped send direction in exchange_face / send_halo / entry_reflected / the extended-gauge
    setup would fail here (with extent 2 both neighbours are the same rank).  Local extent 4: lengths 1..3 go through the
    multi-layer halo (OPT; posted ahead or not), '-' entries are reflected from '+' ones, length 5 takes the step-by-step
    sequence; calc 2 = BASIC (the reference's sequence: one face per step)."""
    monkeypatch.setenv("MUGIQ_HIP_HALO_AHEAD", ahead)
    mp.spawn(mp_workers.gpu_worker, args=(4, free_port(), grid, prec, order, calc, G), nprocs=4, join=True)


def test_driver_writes_reference_hdf5_tree(hip, tmp_path):
    """computeLoop -> writeLoopsHDF5 (lib/interface_mugiq.cpp:158-172): the file holds, group by group, the numbers the ORACLE gets for
    the same job (position-space loops -> reorder + gamma5 map -> projection), and they are bit for bit dataMom_bcast."""
    import h5read
    try:
        h5 = h5read.H5()
    except ImportError:
        pytest.skip("libhdf5 not available")
    X = (4, 4, 4, 8)
    ev, Uo, f, U = _setup(hip, X, 3, 8, 2, 77)
    moms = momenta_p2_le(1)
    fn = str(tmp_path / "loops.h5")
    prm = hip.MugiqLoopParam(FTSign=1, doMomProj=True, gauge=U, momMatrix=[list(m) for m in moms], Nmom=len(moms),
                             writeMomSpaceHDF5=True, fname_mom_h5=fn).set_displace_entry_string("-t:1,2;+y:2")
    loop = hip.Loop_Mugiq(prm, f, sigmas(3))
    with pytest.raises(hip.MugiqHipError):
        loop.writeLoopsHDF5()                                 # before computeCoarseLoop
    loop.computeCoarseLoop()
    loop.writeLoopsHDF5()
    mom = loop.dataMom_global()                               # [Nmom][nLoop][16][totT]
    # the oracle's momentum-space loops of the same job
    _, ds, da, db = orc.parse_disp_entry_string("-t:1,2;+y:2")
    cprm = orc.LoopComputeParam(ds, da, db)
    pos = orc.compute_loop_position_space(ev, sigmas(3), cprm, Uo, X)
    V, locV3 = int(np.prod(X)), X[0] * X[1] * X[2]
    mp_ = orc.convert_idx_order_map_gamma(pos, cprm.nData, cprm.nLoop, 2, V // 2, X)
    ref = orc.momentum_projection_local(mp_, orc.phase_matrix(moms, locV3, 1, X, X), X[3], cprm.nData, locV3, len(moms))
    ref = np.asarray(ref).reshape(len(moms), cprm.nLoop, 16, X[3])
    scale = np.max(np.abs(ref))
    fid = h5.open(fn)
    names = ["disp_0", "disp_-t_1", "disp_-t_2", "disp_+y_2"]
    for im, p in enumerate(moms):
        for iL, dn in enumerate(names):
            for ig in range(16):
                got = h5.read(fid, "/mom_%+d_%+d_%+d/%s/%s/loop" % (p + (dn, hip.GammaName(ig))))
                assert np.array_equal(got[:, 0] + 1j * got[:, 1], mom[im, iL, ig])
                assert np.max(np.abs(got[:, 0] + 1j * got[:, 1] - ref[im, iL, ig])) < 1e-12 * scale, (p, dn, ig)
    h5.close(fid)
    loop.close()
    # position-space output is "Not supported yet!" in the reference too
    l2 = hip.Loop_Mugiq(hip.MugiqLoopParam(writePosSpaceHDF5=True, fname_pos_h5=str(tmp_path / "p.h5")), f, sigmas(3))
    l2.computeCoarseLoop()
    with pytest.raises(hip.MugiqHipError):
        l2.writeLoopsHDF5()
    l2.close()


@pytest.mark.parametrize("prec", [8, 4])
def test_extended_gauge_from_qdp_host_single_process(hip, prec):
    """lib/displace.cpp:70-134 on one process: unpartitioned dims wrap periodically, any even total border."""
    X = (4, 6, 4, 2)
    rng = np.random.default_rng(2)
    U_lex = random_gauge_lex(rng, X)
    U_loc = np.stack([orc.lex_to_eo(U_lex[mu], X) for mu in range(4)])
    cdt = np.complex128 if prec == 8 else np.complex64
    for R in [(0, 0, 0, 0), (2, 0, 0, 2), (1, 1, 0, 0), (2, 2, 2, 2)]:
        g = hip.GaugeField(X, R, prec).set_from_qdp_host(orc.gauge_to_qdp_host(U_loc))
        exp = orc.extended_gauge_from_global(U_lex, (0, 0, 0, 0), (1, 1, 1, 1), R).astype(cdt)
        assert np.array_equal(g.get_logical(), exp), R


@pytest.mark.parametrize("order,calc", [(4, "opt"), (2, "basic")])
def test_driver_mixed_precision(hip, order, calc):
    """fp32 eigenvectors + links, fp64 loop buffers / FT (MugiqLoopParam.loopPrecision = 8)."""
    X = (4, 4, 4, 8)
    nev = 4
    ev, Uo, f, U = _setup(hip, X, nev, 4, order, 91)
    sg = sigmas(nev)
    moms = momenta_p2_le(2)
    prm = hip.MugiqLoopParam(FTSign=1, doMomProj=True, gauge=U, momMatrix=[list(m) for m in moms], Nmom=len(moms), loopPrecision=8,
                             calcType=hip.LOOP_CALC_TYPE_BASIC_KERNEL if calc == "basic" else hip.LOOP_CALC_TYPE_OPT_KERNEL)
    prm.set_displace_entry_string("+x:1,2;-t:2")
    loop = hip.Loop_Mugiq(prm, f, sg)
    assert loop.precision == 4 and loop.loopPrecision == 8
    loop.computeCoarseLoop()
    pos = loop.dataPos_d
    assert pos.dtype == torch.complex128
    cprm = orc.LoopComputeParam(["+x", "-t"], [1, 2], [2, 2])
    sg32 = np.float32(sg).astype(np.float64)
    ref_pos = orc.compute_loop_position_space(ev, sg32, cprm, Uo, X)
    V = int(np.prod(X))
    # ultra-local slot: only the inputs are fp32 -> fp64-exact; displaced slots carry the fp32 link products
    assert rel_err(pos[:16 * V].cpu().numpy(), ref_pos[:16 * V]) < 1e-13
    assert rel_err(pos.cpu().numpy(), ref_pos) < 2e-6
    locV3 = X[0] * X[1] * X[2]
    ref_mom = orc.momentum_projection_local(orc.convert_idx_order_map_gamma(ref_pos, cprm.nData, cprm.nLoop, 2, V // 2, X),
                                            orc.phase_matrix(moms, locV3, 1, X, X), X[3], cprm.nData, locV3, len(moms))
    assert loop.dataMom_bcast.dtype == np.complex128 and rel_err(loop.dataMom_bcast, ref_mom) < 2e-6
    loop.close()
    with pytest.raises(hip.MugiqHipError):
        ev8, _, f8, U8 = _setup(hip, X, 1, 8, 2, 1)
        hip.Loop_Mugiq(hip.MugiqLoopParam(loopPrecision=4), f8, sg[:1])        # fp32 loops over fp64 eigenvectors


@pytest.mark.parametrize("calc,entries", [("opt", None), ("opt", "+z:1,2;-t:1"), ("basic", "+x:1")])
def test_driver_mg_coarse_path(hip, calc, entries):
    """configs[4]: coarse eigenvectors + transfer operator -> prolong -> loops (fused prolong-contract when no
    displacement entries are requested), vs oracle prolongate + compute_loop_position_space."""
    X, bs, nvec, nev = (8, 4, 4, 8), (4, 2, 2, 4), 6, 5
    rng = np.random.default_rng(808)
    vcb = int(np.prod(X)) // 2
    Xc = [X[d] // bs[d] for d in range(4)]
    vcbc = int(np.prod(Xc)) // 2
    Vn = (rng.standard_normal((2, vcb, 4, 3, nvec)) + 1j * rng.standard_normal((2, vcb, 4, 3, nvec))) / np.sqrt(12.0 * nvec)
    phis = [rng.standard_normal((2, vcbc, 2, nvec)) + 1j * rng.standard_normal((2, vcbc, 2, nvec)) for _ in range(nev)]
    Uo = orc.extended_gauge_from_global(random_gauge_lex(rng, X), (0, 0, 0, 0), (1, 1, 1, 1), (0, 0, 0, 0))
    sg = sigmas(nev)
    T = hip.Transfer(X, nvec, bs, 2, 8).set_logical(Vn)
    cf = [hip.CoarseField(Xc, nvec, 8).set_logical(p) for p in phis]
    U = hip.GaugeField(X, (0, 0, 0, 0), 8).set_logical(Uo)
    moms = momenta_p2_le(1)
    prm = hip.MugiqLoopParam(FTSign=-1, doMomProj=True, momMatrix=[list(m) for m in moms], Nmom=len(moms), gauge=U,
                             calcType=hip.LOOP_CALC_TYPE_BASIC_KERNEL if calc == "basic" else hip.LOOP_CALC_TYPE_OPT_KERNEL)
    if entries:
        prm.set_displace_entry_string(entries)
        _, s, a, b = orc.parse_disp_entry_string(entries)
        cprm = orc.LoopComputeParam(s, a, b)
    else:
        cprm = orc.LoopComputeParam(doNonLocal=False)
    loop = hip.Loop_Mugiq(prm, cf, sg, transfer=T)
    loop.computeCoarseLoop()
    fine = [orc.prolongate(p, Vn, X, bs) for p in phis]
    ref_pos = orc.compute_loop_position_space(fine, sg, cprm, Uo, X)
    assert rel_err(loop.dataPos_d.cpu().numpy(), ref_pos) < 1e-12
    V = int(np.prod(X))
    locV3 = X[0] * X[1] * X[2]
    ref_mom = orc.momentum_projection_local(orc.convert_idx_order_map_gamma(ref_pos, cprm.nData, cprm.nLoop, 2, V // 2, X),
                                            orc.phase_matrix(moms, locV3, -1, X, X), X[3], cprm.nData, locV3, len(moms))
    assert rel_err(loop.dataMom_bcast, ref_mom) < 1e-12
    loop.close()


@pytest.mark.parametrize("levels,entries,calc,prec", [(2, None, "opt", 8), (2, "+z:1,2;-t:1", "opt", 8), (3, None, "opt", 8),
                                                      (3, "-x:1;+x:1", "opt", 4), (2, "+y:1", "basic", 8)])
def test_driver_mg_multilevel_hierarchy(hip, levels, entries, calc, prec):
    """mg_env.nCoarseLevels = 2 and 3 (lib/loop_mugiq.cpp:296-314; include/mg_mugiq.h:20,30): the eigenvectors live on the
    COARSEST level and go through transfer[nCoarseLevels-1] ... transfer[1] (coarse -> coarse, nSpin 2 -> 2) before the
    finest transfer.  8^3x16 -> 4^3x8 -> 2^3x4 (-> 2^3x2), n_vec 8 / 6 / 4, vs the oracle's level-by-level prolongation."""
    rng = np.random.default_rng(4400 + levels)
    X0 = (8, 8, 8, 16)
    bss = [(2, 2, 2, 2), (2, 2, 2, 2), (1, 1, 1, 2)][:levels]
    nvecs = [8, 6, 4][:levels]
    cdt = np.complex128 if prec == 8 else np.complex64
    Xs, Vs, Ts = [X0], [], []
    for l in range(levels):
        X = Xs[l]
        vcb = int(np.prod(X)) // 2
        ns, nc = (4, 3) if l == 0 else (2, nvecs[l - 1])
        V = ((rng.standard_normal((2, vcb, ns, nc, nvecs[l])) + 1j * rng.standard_normal((2, vcb, ns, nc, nvecs[l]))) / np.sqrt(ns * nc * nvecs[l])).astype(cdt)
        Vs.append(V.astype(np.complex128))
        Ts.append(hip.Transfer(X, nvecs[l], bss[l], 2 if l == 0 else 1, prec, fine_spin=ns, fine_color=nc).set_logical(V))
        Xs.append(tuple(X[d] // bss[l][d] for d in range(4)))
    Xc = Xs[-1]
    nev = 5
    vcbc = int(np.prod(Xc)) // 2
    phis = [(rng.standard_normal((2, vcbc, 2, nvecs[-1])) + 1j * rng.standard_normal((2, vcbc, 2, nvecs[-1]))).astype(cdt) for _ in range(nev)]
    cf = [hip.CoarseField(Xc, nvecs[-1], prec).set_logical(p) for p in phis]
    Uo = orc.extended_gauge_from_global(random_gauge_lex(rng, X0), (0, 0, 0, 0), (1, 1, 1, 1), (0, 0, 0, 0)).astype(cdt).astype(np.complex128)
    U = hip.GaugeField(X0, (0, 0, 0, 0), prec).set_logical(Uo)
    sg = sigmas(nev)
    moms = momenta_p2_le(1)
    prm = hip.MugiqLoopParam(FTSign=-1, doMomProj=True, momMatrix=[list(m) for m in moms], Nmom=len(moms), gauge=U,
                             calcType=hip.LOOP_CALC_TYPE_BASIC_KERNEL if calc == "basic" else hip.LOOP_CALC_TYPE_OPT_KERNEL)
    if entries:
        prm.set_displace_entry_string(entries)
        _, s, a, b = orc.parse_disp_entry_string(entries)
        cprm = orc.LoopComputeParam(s, a, b)
    else:
        cprm = orc.LoopComputeParam(doNonLocal=False)
    loop = hip.Loop_Mugiq(prm, cf, sg, transfer=Ts)
    loop.computeCoarseLoop()
    fine = []
    for p in phis:
        v = p.astype(np.complex128)
        for l in range(levels - 1, 0, -1):                      # the engine stores every intermediate level in the field precision
            v = orc.prolongate(v, Vs[l], Xs[l], bss[l], 1).astype(cdt).astype(np.complex128)
        v = orc.prolongate(v, Vs[0], Xs[0], bss[0], 2)
        fine.append(v.astype(cdt).astype(np.complex128) if (prec == 4 and (entries or calc == "basic")) else v)
    ref_pos = orc.compute_loop_position_space(fine, np.float32(sg).astype(np.float64) if prec == 4 else sg, cprm, Uo, X0)
    tol = 1e-12 if prec == 8 else 1e-5
    assert rel_err(loop.dataPos_d.cpu().numpy(), ref_pos) < tol
    loop.close()
    # a hierarchy whose levels do not fit together is refused
    bad = list(Ts)
    if levels >= 2:
        bad[1] = hip.Transfer(Xs[0], nvecs[1], bss[1], 1, prec, fine_spin=2, fine_color=nvecs[0])      # wrong lattice for level 1
        with pytest.raises(hip.MugiqHipError):
            hip.Loop_Mugiq(prm, cf, sg, transfer=bad)


@pytest.mark.parametrize("seed", range(int(os.environ.get("MUGIQ_TEST_SEEDS", 8))))
def test_driver_mg_coarse_path_random(hip, seed, record_max):
    """Seeded random MG set-ups through the driver: aggregate shapes, n_vec (coarse-grid plan for 8/12/16/24/32, the
    per-eigenvector kernel otherwise), precision incl. mixed, with and without displacement entries."""
    rng = np.random.default_rng(8100 + seed)
    X = tuple(int(v) for v in rng.choice([4, 8, 12], size=4))
    while np.prod(X) > 2048:
        X = tuple(int(v) for v in rng.choice([4, 8, 12], size=4))
    bs = tuple(int(rng.choice([b for b in (1, 2, 3, 4, 6) if X[d] % b == 0 and (X[d] // b) % 2 == 0])) for d in range(4))
    nvec = int(rng.choice([2, 6, 8, 12, 16, 24]))
    nev = int(rng.integers(1, 6))
    prec = int(rng.choice([8, 4]))
    lprec = 8 if (prec == 4 and rng.integers(2)) else prec
    cdt = np.complex128 if prec == 8 else np.complex64
    vcb = int(np.prod(X)) // 2
    Xc = [X[d] // bs[d] for d in range(4)]
    vcbc = int(np.prod(Xc)) // 2
    Vn = ((rng.standard_normal((2, vcb, 4, 3, nvec)) + 1j * rng.standard_normal((2, vcb, 4, 3, nvec))) / np.sqrt(12.0 * nvec)).astype(cdt)
    phis = [(rng.standard_normal((2, vcbc, 2, nvec)) + 1j * rng.standard_normal((2, vcbc, 2, nvec))).astype(cdt) for _ in range(nev)]
    Uo = orc.extended_gauge_from_global(random_gauge_lex(rng, X), (0, 0, 0, 0), (1, 1, 1, 1), (0, 0, 0, 0)).astype(cdt).astype(np.complex128)
    sg = sigmas(nev)
    T = hip.Transfer(X, nvec, bs, 2, prec).set_logical(Vn)
    cf = [hip.CoarseField(Xc, nvec, prec).set_logical(p) for p in phis]
    U = hip.GaugeField(X, (0, 0, 0, 0), prec).set_logical(Uo)
    entries = [None, "+z:1,2;-t:1", "-x:2;+x:2;+y:1,3"][int(rng.integers(3))]
    prm = hip.MugiqLoopParam(gauge=U, loopPrecision=lprec if lprec != prec else 0)
    if entries:
        prm.set_displace_entry_string(entries)
        _, s, a, b = orc.parse_disp_entry_string(entries)
        cprm = orc.LoopComputeParam(s, a, b)
    else:
        cprm = orc.LoopComputeParam(doNonLocal=False)
    loop = hip.Loop_Mugiq(prm, cf, sg, transfer=T)
    loop.computeCoarseLoop()
    fine = [orc.prolongate(p.astype(np.complex128), Vn.astype(np.complex128), X, bs) for p in phis]
    if prec == 4 and entries:       # the engine stores the prolonged vectors in fp32 before displacing them
        fine = [f.astype(np.complex64).astype(np.complex128) for f in fine]
    ref = orc.compute_loop_position_space(fine, np.float32(sg).astype(np.float64) if prec == 4 else sg, cprm, Uo, X)
    tol = 1e-12 if prec == 8 else 1e-5                                  # north_star: 1e-12 fp64 / 1e-5 fp32
    err = rel_err(loop.dataPos_d.cpu().numpy(), ref)
    record_max("mg_driver_sweep_%s" % ("fp64" if prec == 8 else ("fp32" if lprec == 4 else "mixed")), err)
    assert err < tol, (X, bs, nvec, nev, prec, lprec, entries, err)
    loop.close()


@pytest.mark.parametrize("X", [(4, 8, 4, 8), (8, 4, 4, 8), (6, 4, 12, 4), (16, 4, 4, 4), (12, 8, 4, 4)])
@pytest.mark.parametrize("tile", ["0", "1", "cols16", "cols32", "regs", "nocarry"])
def test_fused_plans_agree_tiled_and_streaming(hip, tile, X, monkeypatch):
    """The LDS-tiled kernels (csrc/fused_tile.hip: 32-line positions; csrc/fused_tile16.hip: 16-line items, by default only
    for x rows that do not fill 32-line positions) and the first-generation streaming kernel are implementations of the same
    entry point: each must match the oracle.  cols16 / cols32 force one tiled generation for everything it can take; regs
    stages the fp64 column tiles through registers instead of global -> LDS transfers (MUGIQ_HIP_TILE_GLDS=0); nocarry keeps the
    ultra-local loop in a pass of its own."""
    if tile.startswith("cols"):
        monkeypatch.setenv("MUGIQ_HIP_TILE_COLS", tile[4:])
    elif tile == "regs":
        monkeypatch.setenv("MUGIQ_HIP_TILE_GLDS", "0")
    elif tile == "nocarry":                        # the ultra-local loop in its own pass (by default it rides along with an entry)
        monkeypatch.setenv("MUGIQ_HIP_CARRY_ULTRALOCAL", "0")
    else:
        monkeypatch.setenv("MUGIQ_HIP_FUSED_TILE", tile)
    nev = 3
    ev, Uo, f, U = _setup(hip, X, nev, 8, 2, 4242)
    sg = sigmas(nev)
    entry = "+y:1,3;-y:2,3;+z:1,2;-z:1;+t:1,3;-t:1,3;+x:1,2;-x:1,3;+t:5,6"
    prm = hip.MugiqLoopParam(gauge=U).set_displace_entry_string(entry)
    loop = hip.Loop_Mugiq(prm, f, sg)
    loop.computeCoarseLoop()
    _, s, a, b = orc.parse_disp_entry_string(entry)
    ref = orc.compute_loop_position_space(ev, sg, orc.LoopComputeParam(s, a, b), Uo, X)
    assert rel_err(loop.dataPos_d.cpu().numpy(), ref) < 1e-12
    loop.close()


def _random_case(seed):
    rng = np.random.default_rng(seed)
    X = tuple(int(v) for v in rng.choice([2, 4, 6, 8, 12], size=4))
    while np.prod(X) > 4096 or np.prod(X) < 64:
        X = tuple(int(v) for v in rng.choice([2, 4, 6, 8, 12], size=4))
    prec, order = [(8, 2), (8, 4), (4, 2), (4, 4)][int(rng.integers(4))]
    nev = int(rng.integers(1, 6))
    ents = []
    for _ in range(int(rng.integers(1, 6))):
        d = "xyzt"[int(rng.integers(4))]
        a, b = int(rng.integers(1, 8)), int(rng.integers(1, 8))        # lengths may exceed the extent (wraps) and start > stop
        ents.append("%s%s:%d,%d" % ("+-"[int(rng.integers(2))], d, a, b) if rng.integers(3) else "%s%s:%d" % ("+-"[int(rng.integers(2))], d, a))
    pad, gpad = int(rng.choice([0, 0, 6, 32])), int(rng.choice([0, 0, 10]))      # stride = volumeCB + pad (QUDA's pad)
    return X, prec, order, nev, ";".join(ents), pad, gpad


@pytest.mark.parametrize("seed", range(int(os.environ.get("MUGIQ_TEST_SEEDS", 48))))   # MUGIQ_TEST_SEEDS=N widens the sweep
def test_driver_random_shapes_both_fused_plans(hip, seed, monkeypatch, record_max):
    """Seeded random lattice shapes (extents from 2 to 12), storage types, eigenvector counts and displacement entries
    (lengths past the extent, start > stop) through the OPT plan with the tiled and the streaming kernels."""
    X, prec, order, nev, entry, pad, gpad = _random_case(1000 + seed)
    ev, Uo, f, U = _setup(hip, X, nev, prec, order, 77 + seed, pad, gpad)
    sg = sigmas(nev)
    _, s, a, b = orc.parse_disp_entry_string(entry)
    ref = orc.compute_loop_position_space(ev, np.float32(sg).astype(np.float64) if prec == 4 else sg, orc.LoopComputeParam(s, a, b), Uo, X)
    cprm = orc.LoopComputeParam(s, a, b)
    moms = momenta_p2_le(2)
    FTSign = 1 if seed % 2 else -1
    V, locV3 = int(np.prod(X)), X[0] * X[1] * X[2]
    ref_mom = orc.momentum_projection_local(orc.convert_idx_order_map_gamma(ref, cprm.nData, cprm.nLoop, 2, V // 2, X),
                                            orc.phase_matrix(moms, locV3, FTSign, X, X), X[3], cprm.nData, locV3, len(moms))
    tol = 1e-12 if prec == 8 else 1e-5                                  # north_star: 1e-12 fp64 / 1e-5 fp32
    tag = "fp64" if prec == 8 else "fp32"
    for tile in ("1", "0", "cols16") + (("regs",) if prec == 8 and order == 2 else ()):
        monkeypatch.delenv("MUGIQ_HIP_TILE_GLDS", raising=False)
        if tile == "cols16":
            monkeypatch.setenv("MUGIQ_HIP_FUSED_TILE", "1")
            monkeypatch.setenv("MUGIQ_HIP_TILE_COLS", "16")
        elif tile == "regs":
            monkeypatch.delenv("MUGIQ_HIP_TILE_COLS", raising=False)
            monkeypatch.setenv("MUGIQ_HIP_FUSED_TILE", "1")
            monkeypatch.setenv("MUGIQ_HIP_TILE_GLDS", "0")
        else:
            monkeypatch.delenv("MUGIQ_HIP_TILE_COLS", raising=False)
            monkeypatch.setenv("MUGIQ_HIP_FUSED_TILE", tile)
        prm = hip.MugiqLoopParam(gauge=U, FTSign=FTSign, doMomProj=True, momMatrix=[list(m) for m in moms], Nmom=len(moms))
        loop = hip.Loop_Mugiq(prm.set_displace_entry_string(entry), f, sg)
        loop.computeCoarseLoop()
        err = rel_err(loop.dataPos_d.cpu().numpy(), ref)
        err_mom = rel_err(loop.dataMom_bcast, ref_mom)
        loop.close()
        record_max("driver_sweep_pos_%s" % tag, err)
        record_max("driver_sweep_mom_%s" % tag, err_mom)
        assert err < tol and err_mom < tol, (X, prec, order, nev, entry, pad, gpad, tile, err, err_mom)
    if os.environ.get("MUGIQ_TEST_BASIC") or seed % 8 == 0:       # the reference's launch sequence on a subset (all with MUGIQ_TEST_BASIC=1)
        prm = hip.MugiqLoopParam(gauge=U, FTSign=FTSign, doMomProj=True, momMatrix=[list(m) for m in moms], Nmom=len(moms),
                                 calcType=hip.LOOP_CALC_TYPE_BASIC_KERNEL)
        loop = hip.Loop_Mugiq(prm.set_displace_entry_string(entry), f, sg)
        loop.computeCoarseLoop()
        err = rel_err(loop.dataPos_d.cpu().numpy(), ref)
        err_mom = rel_err(loop.dataMom_bcast, ref_mom)
        loop.close()
        record_max("driver_sweep_basic_pos_%s" % tag, err)
        record_max("driver_sweep_basic_mom_%s" % tag, err_mom)
        assert err < tol and err_mom < tol, (X, prec, order, nev, entry, pad, gpad, "basic", err, err_mom)


@pytest.mark.parametrize("force,G,prec,order,calc,ahead", [((0, 0, 0, 1), (4, 4, 8, 8), 8, 2, 1, "1"), ((0, 0, 1, 1), (4, 4, 8, 8), 8, 2, 1, "1"),
                                                           ((0, 0, 1, 1), (4, 4, 8, 8), 8, 2, 1, "0"), ((0, 0, 1, 1), (4, 4, 8, 8), 4, 4, 2, "1"),
                                                           ((1, 1, 0, 0), (8, 8, 4, 4), 8, 4, 1, "1"), ((1, 1, 1, 1), (4, 4, 4, 4), 4, 2, 1, "1"),
                                                           ((0, 0, 0, 1), (8, 4, 4, 4), 8, 2, 1, "1")])
def test_forced_partitioning_on_one_rank(force, G, prec, order, calc, ahead, monkeypatch):
    """MugiqHipComm.partitioned (QUDA's comm_dim_partitioned_set): ONE rank runs the partitioned code path on axes of extent 1
    as its own neighbour -- extended gauge borders through sendrecv, packed multi-layer halos (posted ahead or not), interior /
    boundary tiles, reflected slots with halo, BASIC's face per step, and (t extent 4) the step-by-step sequence for a length
    past the local extent -- and must reproduce the single-domain oracle."""
    monkeypatch.setenv("MUGIQ_HIP_HALO_AHEAD", ahead)
    mp.spawn(mp_workers.gpu_worker, args=(1, free_port(), (1, 1, 1, 1), prec, order, calc, G, None, force), nprocs=1, join=True)


def test_forced_partitioning_scratch_pool_size_tie(monkeypatch):
    """halo buffers of the posted entries exactly as large as a link field, and the entry that runs before the halos are posted
    holds more link fields than any of them (mp_workers.gpu_worker, case "pool_tie"): the pool must not hand that entry's fields to
    the pack stream while its kernels still read them.  Once with the default halo blocks, once with one message per entry."""
    for blocks in (None, "1"):
        if blocks:
            monkeypatch.setenv("MUGIQ_HIP_HALO_BLOCKS", blocks)
        mp.spawn(mp_workers.gpu_worker, args=(1, free_port(), (1, 1, 1, 1), 8, 2, 1, (4, 4, 8, 8), None, (0, 0, 1, 1), "pool_tie"), nprocs=1, join=True)


@pytest.mark.parametrize("world,grid,G,force,env,expect", [
    (1, (1, 1, 1, 1), (8, 16, 8, 8), (0, 0, 1, 1), {}, 4),                                     # own neighbour, packed in place
    (1, (1, 1, 1, 1), (8, 16, 8, 8), (0, 0, 1, 1), {"MUGIQ_HIP_PACK_IN_ENTRY": "0"}, 0),       # ... by the pack kernels
    (1, (1, 1, 1, 1), (8, 16, 8, 8), (0, 0, 1, 1), {"MUGIQ_HIP_SELF_HALO_COPY": "1", "MUGIQ_HIP_HALO_BLOCKS": "3"}, 4),  # messages to self
    (1, (1, 1, 1, 1), (8, 16, 8, 8), (0, 0, 1, 1), {"MUGIQ_HIP_SELF_HALO_COPY": "1"}, 0),      # one block per halo: its kernel packs it
    (1, (1, 1, 1, 1), (8, 16, 8, 8), (0, 0, 1, 1), {"MUGIQ_HIP_MFMA_ROW_WAVES": "16"}, 4),     # 16 rows per workgroup (default: 8)
    (1, (1, 1, 1, 1), (8, 8, 8, 8), (0, 0, 1, 1), {}, 4),                                      # 8 rows = one (z, t) per workgroup
    (1, (1, 1, 1, 1), (8, 8, 8, 8), (0, 0, 1, 1), {"MUGIQ_HIP_MFMA_ROW_WAVES": "16"}, 0),      # 16 rows per workgroup straddle z: not taken
    (2, (1, 1, 1, 2), (8, 16, 8, 16), (0, 0, 1, 0), {"MUGIQ_HIP_HALO_BLOCKS": "2"}, 4),        # t travels (block 0 ahead), z to self
    (4, (1, 1, 2, 2), (8, 16, 16, 16), (0, 0, 0, 0), {"MUGIQ_HIP_HALO_BLOCKS": "3"}, 4)])
def test_first_entry_writes_the_face_layers(world, grid, G, force, env, expect, monkeypatch):
    """OPT plan, z / t partitioned, a mu = x entry first: the row tile of csrc/fused_mfma.hip writes the face layers of the posted halos
    on its way through the eigenvectors (loop.halosPackedInEntry() of them) instead of pack kernels reading the eigenvectors once
    more beside it; the first block of a halo that travels is still packed by its own kernel and goes out before the entry starts.
    Every variant against the single-domain oracle, position and momentum space."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("MUGIQ_TEST_EXPECT_PACKED", str(expect))
    mp.spawn(mp_workers.gpu_worker, args=(world, free_port(), grid, 8, 2, 1, G, None, force, "pack"), nprocs=world, join=True)


@pytest.mark.parametrize("world,grid,force,tj", [(1, (1, 1, 1, 1), (0, 0, 0, 0), None), (1, (1, 1, 1, 1), (0, 0, 0, 0), "4"),
                                                 (1, (1, 1, 1, 1), (0, 0, 1, 1), None), (2, (1, 1, 1, 2), (0, 0, 0, 0), None),
                                                 (1, (1, 1, 1, 1), (0, 1, 0, 1), None)])
def test_lengths_one_to_eight(world, grid, force, tj, monkeypatch):
    """"+z:1,8" is the reference's own example of an entry (its --displace-entry-string help): lengths 1 .. 8 in every direction on a
    16.8.8.16 lattice -- whole, with z and t (y and t) forced-partitioned (8 ghost layers = the whole local extent), and on two
    ranks along t (local extent 8).  csrc/fused_mfma.hip takes such an entry as launches of three lengths over ONE axial gauge
    continued 8 positions past the line; with MUGIQ_HIP_MFMA_TJ=4 the 4 x 32 tile cannot (4 + 8 > 8 positions) and the vector
    tiles run.  Against the single-domain oracle, position and momentum space."""
    monkeypatch.setenv("MUGIQ_HIP_REFLECT", "0")
    if tj:
        monkeypatch.setenv("MUGIQ_HIP_MFMA_TJ", tj)
    mp.spawn(mp_workers.gpu_worker, args=(world, free_port(), grid, 8, 2, 1, (16, 8, 8, 16), None, force, "long"), nprocs=world, join=True)


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_full_size_forced_partition_equals_unpartitioned(backend):
    """configs[2] per-GPU lattice 48.48.24.24 with z and t forced-partitioned (its 1x1x2x4 grid seen from one rank): all 8
    entries x lengths 1..3, OPT with halos posted ahead and not, BASIC on a subset == the unpartitioned run to 1e-13."""
    mp.spawn(mp_workers.forced_full_size_worker, args=(1, free_port(), (48, 48, 24, 24), 4, (0, 0, 1, 1), backend), nprocs=1, join=True)


def test_configs2_full_size_nev400_forced_partition(hip, monkeypatch, record_max):
    """BASELINE.json configs[2] AS WRITTEN for one rank of its 1x1x2x4 grid: 48.48.24.24, N_ev = 400 (102 GB of eigenvectors), all 8
    entries x lengths 1..3, momentum projection p^2 <= 9, z and t partitioned (the rank is its own neighbour).  At this N_ev the
    partitioned plan runs what the few-eigenvector tests never reach: 2 x 25.5 GB of halo buffers from the pool, six blocks of
    eigenvectors per halo, messages split below 2 GiB, `soff` / ghost_vec_stride past 2^31 bytes.
      (1) the partitioned momentum-space loops equal the unpartitioned ones (an independent path: shifts wrap inside the kernels)
          to 1e-12 -- with reflection OFF on the partitioned side, so that all 8 entries go through the halo path;
      (2) on the partitioned position-space loops, Gamma = 1: the lattice sum of the loop displaced by -k mu is the complex
          conjugate of the one displaced by +k mu (two independent computations, both through halos on z and t)."""
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import make_evecs, make_gauge, ENTRIES_CFG2
    X, nev = (48, 48, 24, 24), 400
    V = int(np.prod(X))
    dev = torch.device("cuda", 0)
    free_b, _ = torch.cuda.mem_get_info(dev)
    if free_b < 230e9:
        pytest.skip("needs ~215 GB of free HBM (%.0f GB free)" % (free_b / 1e9))
    big, f = make_evecs(hip, X, nev, 8, 2, dev, seed=2024)
    sg = 0.01 + 0.002 * np.arange(nev)
    moms = momenta_p2_le(9)

    def run(gauge, comm):
        prm = hip.MugiqLoopParam(gauge=gauge, doMomProj=True, momMatrix=moms, Nmom=len(moms), FTSign=-1).set_displace_entry_string(ENTRIES_CFG2)
        loop = hip.Loop_Mugiq(prm, f, sg, comm).setProfiling()
        loop.computeCoarseLoop()
        return loop

    monkeypatch.delenv("MUGIQ_HIP_REFLECT", raising=False)
    loop = run(make_gauge(hip, X, 8, dev, 77, None), None)
    assert sum(loop.derivedFrom(i) >= 0 for i in range(8)) == 4
    ref_mom = np.array(loop.dataMom_global())
    loop.close()
    del loop
    torch.cuda.empty_cache()

    monkeypatch.setenv("MUGIQ_HIP_REFLECT", "0")
    comm = hip.GridComm((1, 1, 1, 1), device=dev, force_partitioned=(0, 0, 1, 1))
    loop = run(make_gauge(hip, X, 8, dev, 77, comm), comm)
    assert all(loop.derivedFrom(i) < 0 for i in range(8))
    ph = loop.phases()
    kinds = set(p["kind"] for p in ph)
    assert {"halo_transfer", "entry_interior", "entry_boundary"} <= kinds, kinds
    halo_bytes = sum(p["bytes"] for p in ph if p["kind"] == "halo_transfer")
    # z and t entries, both signs, 3 layers of all 400 eigenvectors: 4 x 3 x 400 x (V / 24) x 192 B
    assert halo_bytes == 4 * 3 * nev * (V // 24) * 192, halo_bytes
    mom = np.array(loop.dataMom_global())
    e = float(np.max(np.abs(mom - ref_mom)) / np.max(np.abs(ref_mom)))
    record_max("configs2_nev400_forced_vs_unpartitioned_mom", e)
    assert e < 1e-12, e
    pos = loop.dataPos_d.view(loop.nLoop, 16, V)
    for axis in range(4):
        for k in range(3):
            plus = pos[1 + 6 * axis + k, 0].sum().item()
            minus = pos[1 + 6 * axis + 3 + k, 0].sum().item()
            assert abs(plus - np.conj(minus)) < 1e-11 * max(abs(plus), 1e-3), (axis, k, plus, minus)
    loop.close()
    del pos, loop, f, big
    torch.cuda.empty_cache()


def test_full_size_reflected_entries_agree_with_computed_ones(hip, monkeypatch):
    """BASELINE.json configs[2] per-GPU lattice (48.48.24.24), few eigenvectors: the OPT plan with reflected entries, without
    them, and with the streaming kernel give the same 25 loop slots (a size-independent property; the oracle is too
    slow at this size)."""
    X = (48, 48, 24, 24)
    V = int(np.prod(X))
    nev = 3
    gen = torch.Generator(device="cuda").manual_seed(5)
    f = []
    for n in range(nev):
        sp = hip.SpinorField(X, 8, 2)
        sp.data.copy_(torch.complex(torch.randn(sp.data.numel(), dtype=torch.float64, device="cuda", generator=gen),
                                    torch.randn(sp.data.numel(), dtype=torch.float64, device="cuda", generator=gen)) / np.sqrt(24.0 * V))
        f.append(sp)
    vcb = V // 2
    m = torch.complex(torch.randn(8 * vcb, 3, 3, dtype=torch.float64, device="cuda", generator=gen),
                      torch.randn(8 * vcb, 3, 3, dtype=torch.float64, device="cuda", generator=gen))
    r0 = m[:, 0] / torch.linalg.vector_norm(m[:, 0], dim=-1, keepdim=True)
    r1 = m[:, 1] - (r0.conj() * m[:, 1]).sum(-1, keepdim=True) * r0
    r1 = r1 / torch.linalg.vector_norm(r1, dim=-1, keepdim=True)
    r2 = torch.linalg.cross(r0.conj(), r1.conj())
    U = hip.GaugeField(X, (0, 0, 0, 0), 8)
    U.data.copy_(torch.stack([r0, r1, r2], dim=1).reshape(4, 2, vcb, 9).permute(1, 0, 3, 2).contiguous().reshape(-1))
    entry = "+x:1,3;-x:1,3;+y:1,3;-y:1,3;+z:1,3;-z:1,3;+t:1,3;-t:1,3"
    sg = sigmas(nev)
    res = {}
    for mode, env in (("reflect", {}), ("direct", {"MUGIQ_HIP_REFLECT": "0"}), ("streaming", {"MUGIQ_HIP_REFLECT": "0", "MUGIQ_HIP_FUSED_TILE": "0"})):
        for k in ("MUGIQ_HIP_REFLECT", "MUGIQ_HIP_FUSED_TILE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        loop = hip.Loop_Mugiq(hip.MugiqLoopParam(gauge=U).set_displace_entry_string(entry), f, sg)
        loop.computeCoarseLoop()
        assert sum(loop.derivedFrom(i) >= 0 for i in range(8)) == (4 if mode == "reflect" else 0)
        res[mode] = loop.dataPos_d.clone()
        loop.close()
    scale = float(res["direct"].abs().max())
    assert float((res["reflect"] - res["direct"]).abs().max()) < 1e-12 * scale
    assert float((res["streaming"] - res["direct"]).abs().max()) < 1e-12 * scale


def test_axial_gauge_matrix_pipe_tile_geometries(hip, monkeypatch):
    """csrc/fused_mfma.hip: every tile geometry of the axial-gauge matrix-pipe kernel (4 x 32, 8 x 16, 12 x 16 sites for mu = y, z, t;
    whole x rows for mu = x), both signs, one to three lengths, with and without the ultra-local loop riding along, the axial gauge
    straight from the gauge field (default on an unpartitioned lattice) or from path-link fields (MUGIQ_HIP_GAUGE_FROM_LINKS=0), against the
    oracle -- and against the vector tiles of csrc/fused_tile.hip (MUGIQ_HIP_TILE_MFMA=0), which apply W_k per slot instead of
    rotating the eigenvectors into the axial gauge once.  The lattice has a t extent every geometry divides."""
    X, nev = (8, 8, 4, 24), 3
    ev, Uo, f, U = _setup(hip, X, nev, 8, 2, 4242)
    sg = sigmas(nev)
    entry = "+t:1,3;-t:1,3;+x:1,3;-x:1,2;+y:1;-z:1,2"
    _, s, a, b = orc.parse_disp_entry_string(entry)
    ref = orc.compute_loop_position_space(ev, sg, orc.LoopComputeParam(s, a, b), Uo, X)
    monkeypatch.setenv("MUGIQ_HIP_REFLECT", "0")            # every entry from the eigenvectors: both signs go through the kernels
    settings = [{"MUGIQ_HIP_MFMA_TJ": "4"}, {"MUGIQ_HIP_MFMA_TJ": "8"}, {"MUGIQ_HIP_MFMA_TJ": "12"}, {"MUGIQ_HIP_MFMA_ROW": "0"},
                {"MUGIQ_HIP_CARRY_ULTRALOCAL": "0"}, {"MUGIQ_HIP_GAUGE_FROM_LINKS": "0"}, {"MUGIQ_HIP_TILE_MFMA": "0"}]
    got = {}
    for env in settings:
        for k in ("MUGIQ_HIP_MFMA_TJ", "MUGIQ_HIP_MFMA_ROW", "MUGIQ_HIP_CARRY_ULTRALOCAL", "MUGIQ_HIP_TILE_MFMA", "MUGIQ_HIP_GAUGE_FROM_LINKS"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        loop = hip.Loop_Mugiq(hip.MugiqLoopParam(gauge=U).set_displace_entry_string(entry), f, sg)
        loop.computeCoarseLoop()
        tag = "%s=%s" % next(iter(env.items()))
        got[tag] = loop.dataPos_d.cpu().numpy()
        assert rel_err(got[tag], ref) < 1e-12, tag
        if "CARRY" not in tag and "TILE_MFMA" not in tag:
            assert loop.ultraLocalCarrier() >= 0, tag          # an unpartitioned column entry took the ultra-local loop along
        loop.close()
    assert rel_err(got["MUGIQ_HIP_MFMA_TJ=8"], got["MUGIQ_HIP_TILE_MFMA=0"]) < 1e-13


def test_axial_gauge_tile_lengths_not_starting_at_one(hip, monkeypatch):
    """"-x:3" (the second entry of the reference's example string): the links of the fused call are W_3 only, the axial gauge needs
    W_1 .. W_3 -- the driver, which holds them, builds the gauge and hands it to the matrix-pipe tile.  Entries starting at 2 .. 4
    in every direction, unpartitioned, against the oracle and against the vector tiles."""
    X, nev = (8, 8, 8, 16), 3
    ev, Uo, f, U = _setup(hip, X, nev, 8, 2, 777)
    sg = sigmas(nev)
    entry = "+z:2,5;-x:3;+y:4;-t:2,3;+x:2,4;-z:3,8"
    _, s, a, b = orc.parse_disp_entry_string(entry)
    ref = orc.compute_loop_position_space(ev, sg, orc.LoopComputeParam(s, a, b), Uo, X)
    monkeypatch.setenv("MUGIQ_HIP_REFLECT", "0")
    got = {}
    for mfma in ("1", "0"):
        monkeypatch.setenv("MUGIQ_HIP_TILE_MFMA", mfma)
        loop = hip.Loop_Mugiq(hip.MugiqLoopParam(gauge=U).set_displace_entry_string(entry), f, sg)
        loop.computeCoarseLoop()
        got[mfma] = loop.dataPos_d.cpu().numpy()
        assert rel_err(got[mfma], ref) < 1e-12, mfma
        loop.close()
    assert rel_err(got["1"], got["0"]) < 1e-13
    assert not np.array_equal(got["1"], got["0"])              # (two different kernels did run)


@pytest.mark.parametrize("X", [(2, 2, 4, 8), (2, 8, 8, 8)])
def test_row_tile_on_the_smallest_x_extent(hip, X):
    """X0 = 2 (one checkerboard entry per x-row): found by the 2500-seed sweep -- the row tile computed zero tiles along x
    and launched nothing."""
    ev, Uo, f, U = _setup(hip, X, 2, 8, 2, 5)
    sg = sigmas(2)
    for entry in ("-x:1", "+x:1"):
        _, s, a, b = orc.parse_disp_entry_string(entry)
        ref = orc.compute_loop_position_space(ev, sg, orc.LoopComputeParam(s, a, b), Uo, X)
        loop = hip.Loop_Mugiq(hip.MugiqLoopParam(gauge=U).set_displace_entry_string(entry), f, sg)
        loop.computeCoarseLoop()
        assert rel_err(loop.dataPos_d.cpu().numpy(), ref) < 1e-12, entry
        loop.close()


def test_driver_reflects_opposite_sign_entries(hip, monkeypatch):
    """OPT plan: an entry whose direction and lengths were already computed with the opposite sign is derived from it
    (csrc/reflect.hip) instead of going through the eigenvectors again; MUGIQ_HIP_REFLECT=0 and the BASIC plan compute
    every entry.  All of them agree with the oracle."""
    X = (4, 8, 4, 8)
    nev = 3
    ev, Uo, f, U = _setup(hip, X, nev, 8, 2, 555)
    sg = sigmas(nev)
    entry = "+t:1,3;-t:1,3;-y:2,4;+y:3;+x:1,2;-x:1,3;-z:1;+z:1"
    _, s, a, b = orc.parse_disp_entry_string(entry)
    ref = orc.compute_loop_position_space(ev, sg, orc.LoopComputeParam(s, a, b), Uo, X)
    expect = {"opt": [-1, 0, -1, 2, -1, -1, -1, 6], "opt_noreflect": [-1] * 8, "basic": [-1] * 8}
    for mode in ("opt", "opt_noreflect", "basic"):
        if mode == "opt_noreflect":
            monkeypatch.setenv("MUGIQ_HIP_REFLECT", "0")
        else:
            monkeypatch.delenv("MUGIQ_HIP_REFLECT", raising=False)
        prm = hip.MugiqLoopParam(gauge=U, calcType=hip.LOOP_CALC_TYPE_BASIC_KERNEL if mode == "basic" else hip.LOOP_CALC_TYPE_OPT_KERNEL)
        loop = hip.Loop_Mugiq(prm.set_displace_entry_string(entry), f, sg)
        loop.computeCoarseLoop()
        assert [loop.derivedFrom(i) for i in range(8)] == expect[mode], mode     # "-x:1,3" is not covered by "+x:1,2"
        assert rel_err(loop.dataPos_d.cpu().numpy(), ref) < 1e-12, mode
        loop.close()


@pytest.mark.parametrize("prec,order,FTSign", [(8, 2, -1), (8, 4, 1), (4, 4, -1)])
def test_driver_reflects_in_momentum_space(hip, prec, order, FTSign, monkeypatch):
    """OPT plan with momentum projection: reflected entries are derived on the momentum-space array (csrc/reflect_mom.cpp) and
    left out of position space, of the reorder and of the Fourier kernels -- when the momentum list holds -p for every p.  The
    momentum-space result equals the oracle's either way; dataPos is produced on request; a momentum list that is not closed
    under p -> -p (and MUGIQ_HIP_REFLECT_MOM=0) keep the position-space reflection inside the compute."""
    X = (4, 8, 4, 8)
    nev = 3
    ev, Uo, f, U = _setup(hip, X, nev, prec, order, 556)
    sg = sigmas(nev)
    entry = "+t:1,3;-t:1,3;-y:2,4;+y:3;+x:1,2;-x:1,3;-z:1;+z:1"
    _, s, a, b = orc.parse_disp_entry_string(entry)
    cprm = orc.LoopComputeParam(s, a, b)
    pos = orc.compute_loop_position_space(ev, sg, cprm, Uo, X)
    V, locV3 = int(np.prod(X)), X[0] * X[1] * X[2]
    mp = orc.convert_idx_order_map_gamma(pos, cprm.nData, cprm.nLoop, 2, V // 2, X)
    tol = 1e-12 if prec == 8 else 1e-5
    closed = momenta_p2_le(3)
    open_list = [m for m in closed if tuple(m) != (-1, -1, -1)]                 # (1, 1, 1) has lost its partner
    for moms, env, in_mom_space in ((closed, None, True), (closed, "0", False), (open_list, None, False)):
        if env is None:
            monkeypatch.delenv("MUGIQ_HIP_REFLECT_MOM", raising=False)
        else:
            monkeypatch.setenv("MUGIQ_HIP_REFLECT_MOM", env)
        ref = orc.momentum_projection_local(mp, orc.phase_matrix(moms, locV3, FTSign, X, X), X[3], cprm.nData, locV3, len(moms))
        prm = hip.MugiqLoopParam(gauge=U, doMomProj=True, momMatrix=[list(m) for m in moms], Nmom=len(moms), FTSign=FTSign)
        loop = hip.Loop_Mugiq(prm.set_displace_entry_string(entry), f, sg).setProfiling()
        loop.computeCoarseLoop()
        assert [loop.derivedFrom(i) for i in range(8)] == [-1, 0, -1, 2, -1, -1, -1, 6]
        kinds = [p["kind"] for p in loop.phases()]
        assert ("momentum_reflect" in kinds) == in_mom_space and ("entry_reflected" in kinds) == (not in_mom_space), kinds
        assert rel_err(loop.dataMom_global(), ref.reshape(len(moms), cprm.nLoop, 16, X[3])) < tol
        assert rel_err(loop.dataPos_d.cpu().numpy(), pos) < tol                 # materialised on request when it was left out
        assert rel_err(loop.dataPos, pos) < tol
        loop.close()


@pytest.mark.parametrize("what", ["mfma_geometries", "mfma_long", "mfma_start", "random0", "random1", "random2", "random3", "random4", "random5",
                                  "mg0", "mg1", "mg2", "mixed", "single"])
def test_kernels_do_not_read_unwritten_lds(hip, what, monkeypatch, record_max):
    """LDS is not cleared between kernels: a kernel that reads a cell it never wrote -- the padding of a tile image, an operand lane that
    "multiplies zero anyway" -- computes with what the previous kernel left there, which is finite nearly always.  With
    MUGIQ_HIP_DEBUG_POISON_LDS=1 every compute entry point first fills the LDS of all CUs with NaN patterns; the cases of the tests
    named here must still match the oracle (a NaN anywhere fails them)."""
    monkeypatch.setenv("MUGIQ_HIP_DEBUG_POISON_LDS", "1")
    if what == "mfma_geometries":
        test_axial_gauge_matrix_pipe_tile_geometries(hip, monkeypatch)
    elif what == "mfma_long":
        test_lengths_one_to_eight(1, (1, 1, 1, 1), (0, 0, 1, 1), None, monkeypatch)
        test_lengths_one_to_eight(1, (1, 1, 1, 1), (0, 0, 0, 0), None, monkeypatch)
    elif what == "mfma_start":
        test_axial_gauge_tile_lengths_not_starting_at_one(hip, monkeypatch)
    elif what.startswith("random"):
        test_driver_random_shapes_both_fused_plans(hip, 7 * int(what[6:]) + 3, monkeypatch, record_max)
    elif what.startswith("mg"):
        test_driver_mg_coarse_path_random(hip, int(what[2:]), record_max)
    elif what == "mixed":
        test_driver_mixed_precision(hip, 4, "opt")
    else:
        test_driver_single_process_vs_oracle(hip, 8, 2, "opt")
        test_driver_single_process_vs_oracle(hip, 4, 4, "opt")
