"""GPU parity tests: every operator of the C ABI against the CPU oracle on the same seeded inputs.

Tolerances (north_star): complex loop traces 1e-12 (fp64) / 1e-5 (fp32), relative to the largest element;
displacement INDEXING bit-exact (checked with integer-valued fields and unit links, where every output is
an exact copy of one input element).
"""
import os

import numpy as np
import pytest
import torch

from util import orc, random_gauge_lex, random_spinor_lex, unit_gauge_lex, sigmas, momenta_p2_le, rel_err

pytestmark = pytest.mark.gpu

TOL = {8: 1e-12, 4: 1e-5}
CASES = [(8, 2), (8, 4), (4, 2), (4, 4)]          # (precision, field order): the reference's four instantiations


def _np_c(prec):
    return np.complex128 if prec == 8 else np.complex64


def _field(hip, v, X, prec, order, pad=0):
    return hip.SpinorField(X, prec, order, pad=pad).set_logical(v.astype(_np_c(prec)))


def _rounded(v, prec):
    """The inputs the GPU actually sees (fp32 cases are checked against the fp64 oracle on rounded inputs)."""
    return v.astype(_np_c(prec)).astype(np.complex128)


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec,order", CASES)
@pytest.mark.parametrize("X", [(4, 4, 4, 4), (8, 4, 6, 2), (8, 8, 8, 8)])
def test_loop_contraction_single_and_batched(hip, prec, order, X):
    rng = np.random.default_rng(101)
    nev = 5
    V = int(np.prod(X))
    evL = [orc.lex_to_eo(random_spinor_lex(rng, X), X) for _ in range(nev)]
    evR = [orc.lex_to_eo(random_spinor_lex(rng, X), X) for _ in range(nev)]
    sg = sigmas(nev)
    fL = [_field(hip, v, X, prec, order) for v in evL]
    fR = [_field(hip, v, X, prec, order) for v in evR]
    cdt = torch.complex128 if prec == 8 else torch.complex64

    # oracle in fp64 on the rounded inputs
    ref_lr = np.zeros(16 * V, dtype=np.complex128)
    ref_ll = np.zeros(16 * V, dtype=np.complex128)
    for n in range(nev):
        orc.loop_contract(ref_lr, _rounded(evL[n], prec), _rounded(evR[n], prec), sg[n])
        orc.loop_contract(ref_ll, _rounded(evL[n], prec), _rounded(evL[n], prec), sg[n])

    # one call per eigenvector, accumulating in place like the reference (lib/loop_mugiq.cpp:493,502)
    loop = torch.zeros(16 * V, dtype=cdt, device="cuda")
    for n in range(nev):
        hip.performLoopContraction(loop, fL[n], fR[n], sg[n])
    assert rel_err(loop.cpu().numpy(), ref_lr) < TOL[prec]

    # batched, L != R
    loop_b = torch.zeros(16 * V, dtype=cdt, device="cuda")
    hip.performLoopContractionBatched(loop_b, fL, fR, sg)
    assert rel_err(loop_b.cpu().numpy(), ref_lr) < TOL[prec]

    # batched ultra-local (L == R, Hermitian kernel) with odd and even batch sizes, accumulating on top
    loop_u = torch.zeros(16 * V, dtype=cdt, device="cuda")
    hip.performLoopContractionBatched(loop_u, fL[:3], fL[:3], sg[:3])
    hip.performLoopContractionBatched(loop_u, fL[3:], fL[3:], sg[3:])
    assert rel_err(loop_u.cpu().numpy(), ref_ll) < TOL[prec]
    # single-vector ultra-local path agrees too
    loop_s = torch.zeros(16 * V, dtype=cdt, device="cuda")
    for n in range(nev):
        hip.performLoopContraction(loop_s, fL[n], fL[n], sg[n])
    assert rel_err(loop_s.cpu().numpy(), ref_ll) < TOL[prec]


def test_loop_contraction_padded_stride_and_ragged_volume(hip):
    """pad != 0 (stride > volumeCB) and a volume that is not a multiple of the workgroup size."""
    X = (6, 4, 2, 2)          # V = 96 sites: one partially filled workgroup
    rng = np.random.default_rng(7)
    v = orc.lex_to_eo(random_spinor_lex(rng, X), X)
    w = orc.lex_to_eo(random_spinor_lex(rng, X), X)
    V = int(np.prod(X))
    ref = np.zeros(16 * V, dtype=np.complex128)
    orc.loop_contract(ref, v, w, 0.37)
    for order in (2, 4):
        a = hip.SpinorField(X, 8, order, pad=10).set_logical(v)
        b = hip.SpinorField(X, 8, order, pad=10).set_logical(w)
        loop = torch.zeros(16 * V, dtype=torch.complex128, device="cuda")
        hip.performLoopContraction(loop, a, b, 0.37)
        assert rel_err(loop.cpu().numpy(), ref) < 1e-13


def test_gamma_unit_slot_property_full_precision(hip):
    """Size-independent property: sum_x loop[x, G=1] = sum_n ||v_n||^2 / sigma_n; G=1 slot is real."""
    X = (8, 8, 8, 16)
    rng = np.random.default_rng(3)
    nev = 6
    V = int(np.prod(X))
    f = []
    for _ in range(nev):
        v = torch.randn(2 * 12 * (V // 2), dtype=torch.complex128, device="cuda")
        v /= torch.linalg.vector_norm(v)
        f.append(hip.SpinorField(X, 8, 2, data=v))
    sg = sigmas(nev)
    loop = torch.zeros(16 * V, dtype=torch.complex128, device="cuda")
    hip.performLoopContractionBatched(loop, f, f, sg)
    one = loop[:V]
    assert abs(one.sum().item() - np.sum(1.0 / sg)) < 1e-11 * np.sum(1.0 / sg)
    assert torch.max(torch.abs(one.imag)).item() == 0.0


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec,order", CASES)
@pytest.mark.parametrize("X", [(4, 4, 4, 4), (4, 6, 8, 2)])
def test_displacement_indexing_is_bit_exact(hip, prec, order, X):
    """Unit links + integer-valued spinor: every output element must be an exact copy of the right input."""
    V = int(np.prod(X))
    lexid = np.arange(V).reshape(X[3], X[2], X[1], X[0])
    psi_lex = (lexid[..., None, None] * 12 + np.arange(12).reshape(4, 3)).astype(np.float64)
    psi_lex = psi_lex + 1j * (psi_lex + 0.5)              # all exactly representable in fp32 (< 2^24) and fp64
    psi = orc.lex_to_eo(psi_lex, X)
    U = hip.GaugeField(X, (0, 0, 0, 0), prec).set_logical(
        orc.extended_gauge_from_global(unit_gauge_lex(X), (0, 0, 0, 0), (1, 1, 1, 1), (0, 0, 0, 0)))
    src = _field(hip, psi, X, prec, order)
    dst = hip.SpinorField(X, prec, order)
    Uo = orc.extended_gauge_from_global(unit_gauge_lex(X), (0, 0, 0, 0), (1, 1, 1, 1), (0, 0, 0, 0))
    for dirn in range(4):
        for sign in (hip.DispSignPlus, hip.DispSignMinus):
            hip.performCovariantDisplacementVector(dst, src, U, dirn, sign)
            exp = orc.covariant_displacement(psi, Uo, dirn, sign, X)
            assert np.array_equal(dst.get_logical().astype(np.complex128), exp)


@pytest.mark.parametrize("prec,order", CASES)
def test_displacement_random_su3_and_inverse_property(hip, prec, order):
    X = (4, 4, 6, 8)
    rng = np.random.default_rng(17)
    psi = orc.lex_to_eo(random_spinor_lex(rng, X), X)
    U_lex = random_gauge_lex(rng, X)
    Uo = orc.extended_gauge_from_global(U_lex, (0, 0, 0, 0), (1, 1, 1, 1), (0, 0, 0, 0))
    U = hip.GaugeField(X, (0, 0, 0, 0), prec).set_logical(Uo)
    src = _field(hip, psi, X, prec, order)
    a = hip.SpinorField(X, prec, order)
    b = hip.SpinorField(X, prec, order)
    Ur = Uo.astype(_np_c(prec)).astype(np.complex128)
    for dirn in range(4):
        for sign in (hip.DispSignPlus, hip.DispSignMinus):
            hip.performCovariantDisplacementVector(a, src, U, dirn, sign)
            exp = orc.covariant_displacement(_rounded(psi, prec), Ur, dirn, sign, X)
            assert rel_err(a.get_logical(), exp) < (1e-14 if prec == 8 else 1e-6)
            hip.performCovariantDisplacementVector(b, a, U, dirn, 1 - sign)        # D_-mu D_+mu = 1
            assert rel_err(b.get_logical(), psi) < (1e-12 if prec == 8 else 1e-5)


@pytest.mark.parametrize("order", [2, 4])
def test_displacement_with_ghost_zones_matches_single_domain(hip, order):
    """Two domains along t and z emulated on one GPU: pack faces with the HIP packer, hand them over as ghost
    zones, displace with the border-2 extended gauge, compare with the single-domain oracle."""
    G = (4, 4, 8, 8)
    rng = np.random.default_rng(23)
    psi_lex = random_spinor_lex(rng, G)
    U_lex = random_gauge_lex(rng, G)
    single_U = orc.extended_gauge_from_global(U_lex, (0, 0, 0, 0), (1, 1, 1, 1), (0, 0, 0, 0))
    psi_g = orc.lex_to_eo(psi_lex, G)
    for grid in [(1, 1, 1, 2), (1, 1, 2, 1), (2, 1, 1, 1), (1, 2, 1, 1)]:
        comm = [1 if g > 1 else 0 for g in grid]
        brd = [2 * c for c in comm]
        pdim = comm.index(1)
        l = [G[d] // grid[d] for d in range(4)]
        ranks = [tuple(1 if (d == pdim and r == 1) else 0 for d in range(4)) for r in range(2)]
        f = {r: _field(hip, orc.lex_to_eo(orc.local_block(psi_lex, r, grid), l), l, 8, order) for r in ranks}
        Ue = {r: hip.GaugeField(l, brd, 8).set_logical(orc.extended_gauge_from_global(U_lex, r, grid, brd)) for r in ranks}
        # faces: HIP packer == oracle packer, then install as the neighbour's ghost zones
        for r in ranks:
            other = ranks[1 - ranks.index(r)]
            for high in (0, 1):
                face = torch.zeros(2 * 12 * f[r].face_cb(pdim), dtype=torch.complex128, device="cuda")
                hip.packFace(face, f[r], pdim, high)
                got = f[r].zone_to_logical(pdim, face)
                assert np.array_equal(got, orc.pack_face(f[r].get_logical(), l, pdim, high))
                # low face -> backward neighbour's forward zone (bnd 1); high face -> forward neighbour's bnd 0
                f[other].alloc_ghost(pdim, 1 - high).copy_(face)
        for dirn in range(4):
            for sign in (hip.DispSignPlus, hip.DispSignMinus):
                exp_g = orc.eo_to_lex(orc.covariant_displacement(psi_g, single_U, dirn, sign, G), G)
                for r in ranks:
                    dst = hip.SpinorField(l, 8, order)
                    hip.performCovariantDisplacementVector(dst, f[r], Ue[r], dirn, sign, comm)
                    got = orc.eo_to_lex(dst.get_logical(), l)
                    assert rel_err(got, orc.local_block(exp_g, r, grid)) < 1e-14, (grid, dirn, sign, r)


# ---------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("prec", [8, 4])
@pytest.mark.parametrize("FTSign", [1, -1])
def test_phase_matrix(hip, prec, FTSign):
    localL, totalL, cc = (4, 6, 8, 4), (8, 6, 16, 8), (1, 0, 1, 1)
    moms = momenta_p2_le(3)
    locV3 = localL[0] * localL[1] * localL[2]
    ph = torch.zeros(locV3 * len(moms), dtype=torch.complex128 if prec == 8 else torch.complex64, device="cuda")
    hip.createPhaseMatrixGPU(ph, moms, locV3, len(moms), FTSign, localL, totalL, cc)
    exp = orc.phase_matrix(moms, locV3, FTSign, localL, totalL, cc, np.float64 if prec == 8 else np.float32)
    assert rel_err(ph.cpu().numpy(), exp) < (1e-15 if prec == 8 else 2e-7)


@pytest.mark.parametrize("prec", [8, 4])
@pytest.mark.parametrize("X,nLoop", [((4, 4, 4, 4), 1), ((6, 4, 2, 8), 3), ((8, 8, 8, 4), 2)])
def test_convert_idx_order_map_gamma_is_exact(hip, prec, X, nLoop):
    rng = np.random.default_rng(5)
    V = int(np.prod(X))
    nData = 16 * nLoop
    cdt = np.complex128 if prec == 8 else np.complex64
    data = (rng.standard_normal(nData * V) + 1j * rng.standard_normal(nData * V)).astype(cdt)
    d_in = torch.from_numpy(data).cuda()
    d_out = torch.zeros_like(d_in)
    hip.convertIdxOrder_mapGamma(d_out, d_in, nData, nLoop, 2, V // 2, X)
    exp = orc.convert_idx_order_map_gamma(data, nData, nLoop, 2, V // 2, X)
    assert np.array_equal(d_out.cpu().numpy(), exp)          # permutation + sign flip: bit-exact


@pytest.mark.parametrize("prec", [8, 4])
@pytest.mark.parametrize("locT,nData,L3,Nmom", [(4, 16, (4, 4, 4), 7), (8, 32, (8, 6, 4), 19), (2, 16, (6, 2, 2), 1),
                                                (16, 48, (8, 8, 8), 33)])
def test_momentum_projection_vs_oracle_gemm(hip, prec, locT, nData, L3, Nmom):
    rng = np.random.default_rng(9)
    locV3 = L3[0] * L3[1] * L3[2]
    M = locT * nData
    cdt = np.complex128 if prec == 8 else np.complex64
    A = (rng.standard_normal(M * locV3) + 1j * rng.standard_normal(M * locV3)).astype(cdt)
    B = (rng.standard_normal(locV3 * Nmom) + 1j * rng.standard_normal(locV3 * Nmom)).astype(cdt)
    out = torch.zeros(M * Nmom, dtype=torch.complex128 if prec == 8 else torch.complex64, device="cuda")
    hip.momentumProjection(out, torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda(), locT, nData, locV3, Nmom)
    exp = orc.momentum_projection_local(A.astype(np.complex128), B.astype(np.complex128), locT, nData, locV3, Nmom)
    assert rel_err(out.cpu().numpy(), exp) < (1e-13 if prec == 8 else 1e-5)


@pytest.mark.parametrize("prec", [8, 4])
@pytest.mark.parametrize("L,tot,coord,locT,nData,moms", [
    ((4, 4, 4, 4), (4, 4, 4, 4), (0, 0, 0, 0), 4, 16, "p2le3"),
    ((8, 6, 4, 8), (8, 6, 4, 8), (0, 0, 0, 0), 8, 32, "p2le9"),
    ((4, 6, 2, 4), (8, 12, 6, 4), (1, 1, 2, 0), 4, 16, "p2le2"),          # a rank in the middle of a 2 x 2 x 3 spatial grid
    ((6, 2, 4, 2), (6, 2, 4, 2), (0, 0, 0, 0), 2, 48, "odd"),             # an unsorted list with a repeated momentum
    ((8, 8, 4, 2), (8, 8, 4, 2), (0, 0, 0, 0), 2, 16, "p2le30")])         # 600+ momenta: many groups per step
def test_separable_momentum_projection_matches_phase_matrix_product(hip, prec, L, tot, coord, locT, nData, moms):
    """mugiq_hip_momentum_projection_separable (one direction at a time) against the reference's formulation: the dense
    phase matrix of createPhaseMatrixGPU times the reordered loop data (oracle), and against the dense GPU product."""
    if moms == "odd":
        mom = [(1, -2, 0), (0, 0, 0), (-3, 1, 2), (1, -2, 0), (2, 2, -1), (0, 5, 0)]
    else:
        mom = momenta_p2_le(int(moms[4:]))
    rng = np.random.default_rng(17)
    locV3 = L[0] * L[1] * L[2]
    M = locT * nData
    cdt = np.complex128 if prec == 8 else np.complex64
    A = (rng.standard_normal(M * locV3) + 1j * rng.standard_normal(M * locV3)).astype(cdt)
    tdt = torch.complex128 if prec == 8 else torch.complex64
    A_d = torch.from_numpy(A).cuda()
    out = torch.zeros(M * len(mom), dtype=tdt, device="cuda")
    for FTSign in (1, -1):
        hip.momentumProjectionSeparable(out, A_d, mom, FTSign, L, tot, locT, nData, coord)
        ph = orc.phase_matrix(mom, locV3, FTSign, L, tot, coord, dtype=np.float64 if prec == 8 else np.float32)
        exp = orc.momentum_projection_local(A.astype(np.complex128), ph.astype(np.complex128), locT, nData, locV3, len(mom))
        assert rel_err(out.cpu().numpy(), exp) < (1e-13 if prec == 8 else 5e-6), (FTSign,)
        ph_d = torch.empty(locV3 * len(mom), dtype=tdt, device="cuda")
        hip.createPhaseMatrixGPU(ph_d, mom, locV3, len(mom), FTSign, L, tot, coord)
        out2 = torch.zeros_like(out)
        hip.momentumProjection(out2, A_d, ph_d, locT, nData, locV3, len(mom))
        assert rel_err(out.cpu().numpy(), out2.cpu().numpy()) < (1e-13 if prec == 8 else 5e-6)


@pytest.mark.parametrize("prec", [8, 4])
@pytest.mark.parametrize("X,nLoop,coord,tot", [((4, 4, 4, 4), 1, (0, 0, 0, 0), None), ((6, 4, 2, 8), 3, (0, 0, 0, 0), None),
                                               ((8, 8, 8, 4), 2, (1, 0, 1, 0), (16, 8, 24, 8)), ((12, 6, 4, 2), 5, (0, 0, 0, 0), None)])
def test_convert_and_project_in_one_call(hip, prec, X, nLoop, coord, tot):
    """mugiq_hip_convert_and_project (reorder + gamma5 map + x step in one kernel, then y and z) against the reference's
    sequence on the oracle: convertIdxOrder_mapGamma, createPhaseMatrix, dense product."""
    tot = tot or X
    rng = np.random.default_rng(31)
    V = int(np.prod(X))
    nData = 16 * nLoop
    cdt = _np_c(prec)
    pos = (rng.standard_normal(nData * V) + 1j * rng.standard_normal(nData * V)).astype(cdt)
    mom = momenta_p2_le(5)
    locV3 = X[0] * X[1] * X[2]
    out = torch.zeros(X[3] * nData * len(mom), dtype=torch.complex128 if prec == 8 else torch.complex64, device="cuda")
    hip.convertAndProject(out, torch.from_numpy(pos).cuda(), nData, nLoop, mom, -1, X, tot, coord)
    mp_ = orc.convert_idx_order_map_gamma(pos.astype(np.complex128), nData, nLoop, 2, V // 2, X)
    ph = orc.phase_matrix(mom, locV3, -1, X, tot, coord, dtype=np.float64 if prec == 8 else np.float32)
    exp = orc.momentum_projection_local(mp_, ph.astype(np.complex128), X[3], nData, locV3, len(mom))
    assert rel_err(out.cpu().numpy(), exp) < (1e-13 if prec == 8 else 5e-6)


@pytest.mark.parametrize("prec", [8, 4])
@pytest.mark.parametrize("X,pxs", [((4, 2, 2, 40), [0, 1]), ((12, 2, 2, 4), list(range(-5, 6))), ((8, 4, 2, 36), list(range(-4, 6))),
                                   ((32, 2, 2, 32), [0, 3, -1]),
                                   # (x, t) rows of a y pair beyond 64 KiB of LDS: the tile is cut into chunks of time slices
                                   # (48 x 48: two chunks of 24 in fp64, one tile in fp32; 48 x 96 -- one GPU's view of
                                   # 48^3 x 96: three of 32 / two of 48; 64 x 40 with 11 p_x: ragged chunks, several passes)
                                   ((48, 2, 2, 48), [0, 1, -1, 2]), ((48, 2, 4, 96), [0, -2, 1]), ((64, 2, 2, 40), list(range(-5, 6)))])
def test_convert_and_project_multi_pass_shapes(hip, prec, X, pxs):
    """The fused reorder + x step keeps one row (y, t) per lane and 8 distinct p_x per pass: more than 64 rows (Lt > 32)
    and more than 8 distinct p_x go through several passes over the staged tile; tiles beyond the LDS are cut along t."""
    rng = np.random.default_rng(47)
    V = int(np.prod(X))
    nLoop, nData = 2, 32
    cdt = _np_c(prec)
    pos = (rng.standard_normal(nData * V) + 1j * rng.standard_normal(nData * V)).astype(cdt)
    mom = [(px, py, (px + py) % 2) for px in pxs for py in (0, 1)]
    locV3 = X[0] * X[1] * X[2]
    out = torch.zeros(X[3] * nData * len(mom), dtype=torch.complex128 if prec == 8 else torch.complex64, device="cuda")
    hip.convertAndProject(out, torch.from_numpy(pos).cuda(), nData, nLoop, mom, 1, X, X)
    mp_ = orc.convert_idx_order_map_gamma(pos.astype(np.complex128), nData, nLoop, 2, V // 2, X)
    ph = orc.phase_matrix(mom, locV3, 1, X, X, dtype=np.float64 if prec == 8 else np.float32)
    exp = orc.momentum_projection_local(mp_, ph.astype(np.complex128), X[3], nData, locV3, len(mom))
    assert rel_err(out.cpu().numpy(), exp) < (1e-13 if prec == 8 else 5e-6)


@pytest.mark.parametrize("seed", range(int(os.environ.get("MUGIQ_TEST_SEEDS", 16))))
def test_random_projection_shapes(hip, seed):
    """Seeded random local lattices, process grids / rank coordinates, momentum lists (unsorted, with repeats) and loop
    counts through the two projection entry points of the OPT plan."""
    rng = np.random.default_rng(7000 + seed)
    X = tuple(int(v) for v in rng.choice([2, 4, 6, 8, 12], size=4))
    while np.prod(X) > 2048:
        X = tuple(int(v) for v in rng.choice([2, 4, 6, 8, 12], size=4))
    grid = [int(v) for v in rng.choice([1, 1, 2, 3], size=3)] + [1]
    coord = tuple(int(rng.integers(g)) for g in grid)
    tot = tuple(X[d] * grid[d] for d in range(4))
    prec = int(rng.choice([8, 4]))
    nLoop = int(rng.integers(1, 4))
    nData = 16 * nLoop
    nmom = int(rng.integers(1, 40))
    mom = [tuple(int(v) for v in rng.integers(-4, 5, size=3)) for _ in range(nmom)]
    FTSign = int(rng.choice([-1, 1]))
    V, locV3 = int(np.prod(X)), X[0] * X[1] * X[2]
    cdt = _np_c(prec)
    tdt = torch.complex128 if prec == 8 else torch.complex64
    pos = (rng.standard_normal(nData * V) + 1j * rng.standard_normal(nData * V)).astype(cdt)
    mp_ = orc.convert_idx_order_map_gamma(pos.astype(np.complex128), nData, nLoop, 2, V // 2, X)
    ph = orc.phase_matrix(mom, locV3, FTSign, X, tot, coord, dtype=np.float64 if prec == 8 else np.float32)
    exp = orc.momentum_projection_local(mp_, ph.astype(np.complex128), X[3], nData, locV3, nmom)
    tol = 1e-13 if prec == 8 else 1e-5
    out = torch.zeros(X[3] * nData * nmom, dtype=tdt, device="cuda")
    hip.convertAndProject(out, torch.from_numpy(pos).cuda(), nData, nLoop, mom, FTSign, X, tot, coord)
    assert rel_err(out.cpu().numpy(), exp) < tol, ("fused", X, grid, coord, prec, nLoop, nmom)
    out.zero_()
    hip.momentumProjectionSeparable(out, torch.from_numpy(mp_.astype(cdt)).cuda(), mom, FTSign, X, tot, X[3], nData, coord)
    assert rel_err(out.cpu().numpy(), exp) < tol, ("separable", X, grid, coord, prec, nLoop, nmom)


def test_full_pipeline_ultralocal_and_displaced_vs_oracle(hip):
    """cfg1-like plumbing on the GPU: 8^4... scaled to 4^3x8, N_ev=4, ultra-local + displaced loops,
    reorder, phases, momentum projection -- operator by operator in the reference's order."""
    X = (4, 4, 4, 8)
    rng = np.random.default_rng(77)
    nev = 4
    V = int(np.prod(X))
    ev = [orc.lex_to_eo(random_spinor_lex(rng, X), X) for _ in range(nev)]
    U_lex = random_gauge_lex(rng, X)
    Uo = orc.extended_gauge_from_global(U_lex, (0, 0, 0, 0), (1, 1, 1, 1), (0, 0, 0, 0))
    sg = sigmas(nev)
    cprm = orc.LoopComputeParam(["+z", "-x", "+t"], [1, 2, 1], [2, 2, 3])
    ref_pos = orc.compute_loop_position_space(ev, sg, cprm, Uo, X)

    f = [_field(hip, v, X, 8, 2) for v in ev]
    U = hip.GaugeField(X, (0, 0, 0, 0), 8).set_logical(Uo)
    perLoop = 16 * V
    dataPos = torch.zeros(perLoop * cprm.nLoop, dtype=torch.complex128, device="cuda")
    a, b = hip.SpinorField(X, 8, 2), hip.SpinorField(X, 8, 2)
    for idx in range(-1, cprm.nDispEntries):
        for n in range(nev):
            if idx < 0:
                hip.performLoopContraction(dataPos[:perLoop], f[n], f[n], sg[n])
                continue
            dirn, sign = orc.parse_displacement(cprm.dispString[idx])
            cur, nxt = f[n], a
            cnt = 0
            for idisp in range(1, cprm.dispStop[idx] + 1):
                hip.performCovariantDisplacementVector(nxt, cur, U, dirn, sign)
                cur, nxt = nxt, (b if nxt is a else a)
                if cprm.dispStart[idx] <= idisp:
                    s0 = perLoop * (cprm.nLoopOffset[idx] + cnt)
                    hip.performLoopContraction(dataPos[s0:s0 + perLoop], f[n], cur, sg[n])
                    cnt += 1
    assert rel_err(dataPos.cpu().numpy(), ref_pos) < 1e-12

    moms = momenta_p2_le(2)
    locV3 = X[0] * X[1] * X[2]
    ph = torch.zeros(locV3 * len(moms), dtype=torch.complex128, device="cuda")
    hip.createPhaseMatrixGPU(ph, moms, locV3, len(moms), -1, X, X)
    mp = torch.zeros_like(dataPos)
    hip.convertIdxOrder_mapGamma(mp, dataPos, cprm.nData, cprm.nLoop, 2, V // 2, X)
    mom = torch.zeros(X[3] * cprm.nData * len(moms), dtype=torch.complex128, device="cuda")
    hip.momentumProjection(mom, mp, ph, X[3], cprm.nData, locV3, len(moms))
    ref_mp = orc.convert_idx_order_map_gamma(ref_pos, cprm.nData, cprm.nLoop, 2, V // 2, X)
    ref_mom = orc.momentum_projection_local(ref_mp, orc.phase_matrix(moms, locV3, -1, X, X), X[3], cprm.nData, locV3, len(moms))
    assert rel_err(mom.cpu().numpy(), ref_mom) < 1e-12


def test_errors_are_loud(hip):
    X = (4, 4, 4, 4)
    a, b = hip.SpinorField(X, 8, 2), hip.SpinorField(X, 4, 4)
    loop = torch.zeros(16 * 256, dtype=torch.complex128, device="cuda")
    with pytest.raises(hip.MugiqHipError):
        hip.performLoopContractionBatched(loop, [a], [b], [1.0])          # mismatched precision/order
    with pytest.raises(hip.MugiqHipError):
        hip.performLoopContraction(loop, a, a, 0.0)                         # sigma = 0
    U = hip.GaugeField(X, (0, 0, 0, 0), 8)
    with pytest.raises(hip.MugiqHipError):
        hip.performCovariantDisplacementVector(a, a, U, 0, 1)               # aliasing
    c = hip.SpinorField(X, 8, 2)
    with pytest.raises(hip.MugiqHipError):
        hip.performCovariantDisplacementVector(c, a, U, 3, 1, (0, 0, 0, 1))  # partitioned but no ghost zone
    with pytest.raises(hip.MugiqHipError):
        big = torch.zeros(17 * 256, dtype=torch.complex128, device="cuda")
        hip.convertIdxOrder_mapGamma(torch.zeros_like(big), big, 17, 1, 2, 128, X)     # nData != 16*nLoop


@pytest.mark.parametrize("order", [2, 4])
def test_mixed_precision_contraction_fp32_storage_fp64_accumulation(hip, order):
    """configs[3] / f3: fp32 eigenvectors, all arithmetic and the loop buffer in fp64 -> equals the fp64 oracle on
    the fp32-rounded inputs to fp64 rounding (1e-13), far below the 1e-5 the plain fp32 path is held to."""
    X = (4, 6, 4, 8)
    rng = np.random.default_rng(55)
    nev = 7
    V = int(np.prod(X))
    evL = [orc.lex_to_eo(random_spinor_lex(rng, X), X) for _ in range(nev)]
    evR = [orc.lex_to_eo(random_spinor_lex(rng, X), X) for _ in range(nev)]
    sg = sigmas(nev)
    fL = [_field(hip, v, X, 4, order) for v in evL]
    fR = [_field(hip, v, X, 4, order) for v in evR]
    ref_lr = np.zeros(16 * V, dtype=np.complex128)
    ref_ll = np.zeros(16 * V, dtype=np.complex128)
    for n in range(nev):
        s32 = float(np.float32(sg[n]))                                     # (Float) eVals_sigma[n]
        orc.loop_contract(ref_lr, _rounded(evL[n], 4), _rounded(evR[n], 4), s32)
        orc.loop_contract(ref_ll, _rounded(evL[n], 4), _rounded(evL[n], 4), s32)
    a = torch.zeros(16 * V, dtype=torch.complex128, device="cuda")
    hip.performLoopContractionBatched(a, fL, fR, sg)
    assert rel_err(a.cpu().numpy(), ref_lr) < 1e-13
    b = torch.zeros(16 * V, dtype=torch.complex128, device="cuda")
    hip.performLoopContractionBatched(b, fL, fL, sg)
    assert rel_err(b.cpu().numpy(), ref_ll) < 1e-13
    # fp32 loop buffer over fp64 fields is refused
    f64 = [_field(hip, v, X, 8, order) for v in evL[:1]]
    with pytest.raises(hip.MugiqHipError):
        hip.performLoopContractionBatched(torch.zeros(16 * V, dtype=torch.complex64, device="cuda"), f64, f64, sg[:1])


# ---- f2: prolongator ---------------------------------------------------------------------------------------------
def _mg_problem(X, bs, nvec, nev, seed):
    rng = np.random.default_rng(seed)
    vcb = int(np.prod(X)) // 2
    Xc = [X[d] // bs[d] for d in range(4)]
    vcbc = int(np.prod(Xc)) // 2
    V = (rng.standard_normal((2, vcb, 4, 3, nvec)) + 1j * rng.standard_normal((2, vcb, 4, 3, nvec))) / np.sqrt(nvec * 12.0)
    phis = [rng.standard_normal((2, vcbc, 2, nvec)) + 1j * rng.standard_normal((2, vcbc, 2, nvec)) for _ in range(nev)]
    return V, phis, Xc


@pytest.mark.parametrize("prec,order", CASES)
@pytest.mark.parametrize("X,bs,nvec,nev", [((8, 8, 8, 8), (4, 4, 4, 4), 24, 5), ((8, 4, 12, 4), (2, 2, 3, 2), 6, 35), ((4, 4, 4, 6), (2, 2, 2, 1), 3, 2),
                                           # shapes the matrix-pipe form takes for fp64 FLOAT2 (n_vec 8 | 16 | 24, aggregates of 16 k sites):
                                           # ragged eigenvector counts, every wave with blocks, two passes (> 192 eigenvectors)
                                           ((8, 8, 4, 4), (4, 4, 2, 2), 16, 19), ((4, 4, 8, 8), (2, 2, 4, 4), 8, 70), ((4, 4, 4, 4), (2, 2, 2, 2), 8, 203),
                                           ((8, 8, 4, 4), (4, 2, 2, 2), 24, 66)])
def test_prolongator_matches_oracle(hip, prec, order, X, bs, nvec, nev, monkeypatch):
    V, phis, Xc = _mg_problem(X, bs, nvec, nev, 71)
    cdt = _np_c(prec)
    V = V.astype(cdt)
    phis = [p.astype(cdt) for p in phis]
    T = hip.Transfer(X, nvec, bs, 2, prec).set_logical(V)
    cf = [hip.CoarseField(Xc, nvec, prec).set_logical(p) for p in phis]
    ff = [hip.SpinorField(X, prec, order) for _ in range(nev)]
    hip.prolongateEvecs(ff, cf, T)
    exps = [orc.prolongate(phis[n].astype(np.complex128), V.astype(np.complex128), X, bs) for n in range(nev)]
    for n in range(nev):
        assert rel_err(ff[n].get_logical(), exps[n]) < (1e-14 if prec == 8 else 2e-6), n
    if prec == 8 and order == 2:                      # the vector kernel on the same input (MUGIQ_HIP_PROLONG_MFMA=0)
        monkeypatch.setenv("MUGIQ_HIP_PROLONG_MFMA", "0")
        for f in ff:
            f.data.zero_()
        hip.prolongateEvecs(ff, cf, T)
        for n in range(nev):
            assert rel_err(ff[n].get_logical(), exps[n]) < 1e-14, n


@pytest.mark.parametrize("prec", [8, 4])
@pytest.mark.parametrize("X,bs,ncf,nvec,nev,pad", [((4, 4, 4, 8), (2, 2, 2, 2), 6, 4, 11, 0), ((8, 4, 4, 4), (2, 1, 2, 2), 24, 32, 3, 5),
                                                   ((4, 4, 4, 4), (1, 1, 1, 1), 3, 3, 9, 0), ((12, 4, 4, 4), (3, 2, 2, 2), 8, 24, 17, 2)])
def test_coarse_to_coarse_prolongator_matches_oracle(hip, prec, X, bs, ncf, nvec, nev, pad):
    """One coarse -> coarse level (transfer[lev-1]->P, lib/loop_mugiq.cpp:310): nSpin 2 on both sides, spin_block_size 1;
    eigenvector counts that are not multiples of the per-lane batch, padded strides."""
    rng = np.random.default_rng(5150)
    cdt = _np_c(prec)
    vcb = int(np.prod(X)) // 2
    Xc = [X[d] // bs[d] for d in range(4)]
    vcbc = int(np.prod(Xc)) // 2
    V = ((rng.standard_normal((2, vcb, 2, ncf, nvec)) + 1j * rng.standard_normal((2, vcb, 2, ncf, nvec))) / np.sqrt(2.0 * ncf * nvec)).astype(cdt)
    phis = [(rng.standard_normal((2, vcbc, 2, nvec)) + 1j * rng.standard_normal((2, vcbc, 2, nvec))).astype(cdt) for _ in range(nev)]
    T = hip.Transfer(X, nvec, bs, 1, prec, pad=pad, fine_spin=2, fine_color=ncf).set_logical(V)
    cin = [hip.CoarseField(Xc, nvec, prec, pad=pad).set_logical(p) for p in phis]
    cout = [hip.CoarseField(X, ncf, prec, pad=pad) for _ in range(nev)]
    hip.prolongateCoarseEvecs(cout, cin, T)
    for n in range(nev):
        exp = orc.prolongate(phis[n].astype(np.complex128), V.astype(np.complex128), X, bs, 1)
        assert rel_err(cout[n].get_logical(), exp) < (1e-14 if prec == 8 else 2e-6), n
    with pytest.raises(hip.MugiqHipError):                                     # a finest-level transfer is not a coarse level
        hip.prolongateCoarseEvecs(cout, cin, hip.Transfer(X, nvec, bs, 2, prec, fine_spin=2, fine_color=ncf))
    with pytest.raises(hip.MugiqHipError):                                     # colour count of the coarser side != n_vec
        hip.prolongateCoarseEvecs(cout, [hip.CoarseField(Xc, nvec + 1, prec) for _ in range(nev)], T)


@pytest.mark.parametrize("plan", ["coarse", "direct"])
@pytest.mark.parametrize("prec,lprec", [(8, 8), (4, 4), (4, 8)])
@pytest.mark.parametrize("X,bs,nvec,nev", [((8, 8, 8, 8), (4, 4, 4, 4), 24, 37), ((4, 4, 4, 6), (2, 2, 2, 1), 3, 2),
                                           ((4, 4, 4, 4), (2, 2, 2, 2), 32, 9), ((8, 8, 4, 4), (4, 4, 2, 2), 12, 5),
                                           ((8, 4, 12, 4), (2, 2, 3, 2), 16, 11), ((4, 8, 4, 4), (2, 4, 2, 1), 8, 3)])
def test_fused_prolong_contract_matches_oracle(hip, prec, lprec, X, bs, nvec, nev, plan, monkeypatch):
    """MG ultra-local loop: (P c_n)^dag G (P c_n) summed over n, fine vectors never written.  Two plans behind one entry
    point: "coarse" = outer product of the eigenvectors on the coarse grid + one congruence per fine site (n_vec 8, 12,
    16, 24, 32), "direct" = prolong every eigenvector and contract on the spot (any n_vec; the fallback)."""
    if plan == "direct":
        monkeypatch.setenv("MUGIQ_HIP_MG_PLAN", "direct")
    else:
        monkeypatch.delenv("MUGIQ_HIP_MG_PLAN", raising=False)
    V, phis, Xc = _mg_problem(X, bs, nvec, nev, 72)
    cdt = _np_c(prec)
    V = V.astype(cdt)
    phis = [p.astype(cdt) for p in phis]
    sg = sigmas(nev)
    T = hip.Transfer(X, nvec, bs, 2, prec).set_logical(V)
    cf = [hip.CoarseField(Xc, nvec, prec).set_logical(p) for p in phis]
    Vt = int(np.prod(X))
    loop = torch.zeros(16 * Vt, dtype=torch.complex128 if lprec == 8 else torch.complex64, device="cuda")
    hip.prolongateContractBatched(loop, cf, sg, T)
    ref = np.zeros(16 * Vt, dtype=np.complex128)
    for n in range(nev):
        psi = orc.prolongate(phis[n].astype(np.complex128), V.astype(np.complex128), X, bs)
        orc.loop_contract(ref, psi, psi, float(np.float32(sg[n])) if prec == 4 else sg[n])
    tol = 1e-12 if prec == 8 else (1e-5 if lprec == 4 else 1e-12)
    assert rel_err(loop.cpu().numpy(), ref) < tol
    # and it equals prolongate-then-contract through the separate operators
    ff = [hip.SpinorField(X, prec, 2) for _ in range(nev)]
    hip.prolongateEvecs(ff, cf, T)
    loop2 = torch.zeros_like(loop)
    hip.performLoopContractionBatched(loop2, ff, ff, sg)
    assert rel_err(loop.cpu().numpy(), loop2.cpu().numpy()) < (1e-12 if lprec == 8 and prec == 8 else 1e-5)


@pytest.mark.parametrize("seed", range(int(os.environ.get("MUGIQ_TEST_SEEDS", 24))))   # MUGIQ_TEST_SEEDS=N widens the sweep
def test_random_geometry_contraction_and_prolongator(hip, seed):
    """Seeded random shapes through the operator entry points: batch sizes around the kernels' prefetch depth, padded
    strides, L == R and L != R, mixed precision; transfer operators with random aggregate shapes and n_vec."""
    rng = np.random.default_rng(5000 + seed)
    ext = [2, 4, 6, 8, 12]
    X = tuple(int(v) for v in rng.choice(ext, size=4))
    while np.prod(X) > 4096:
        X = tuple(int(v) for v in rng.choice(ext, size=4))
    V = int(np.prod(X))
    prec, order = CASES[int(rng.integers(4))]
    lprec = 8 if (prec == 4 and rng.integers(2)) else prec
    nev = int(rng.integers(1, 10))
    pad = int(rng.choice([0, 0, 2, 18, 64]))
    same = bool(rng.integers(2))
    evL = [orc.lex_to_eo(random_spinor_lex(rng, X), X) for _ in range(nev)]
    evR = evL if same else [orc.lex_to_eo(random_spinor_lex(rng, X), X) for _ in range(nev)]
    sg = sigmas(nev)
    fL = [_field(hip, v, X, prec, order, pad) for v in evL]
    fR = fL if same else [_field(hip, v, X, prec, order, pad) for v in evR]
    ref = np.zeros(16 * V, dtype=np.complex128)
    for n in range(nev):
        orc.loop_contract(ref, _rounded(evL[n], prec), _rounded(evR[n], prec), float(np.float32(sg[n])) if prec == 4 else sg[n])
    loop = torch.zeros(16 * V, dtype=torch.complex128 if lprec == 8 else torch.complex64, device="cuda")
    hip.performLoopContractionBatched(loop, fL, fR, sg)
    tol = 1e-12 if lprec == 8 else 1e-5
    assert rel_err(loop.cpu().numpy(), ref) < tol, (X, prec, order, lprec, nev, pad, same)

    # prolongator: aggregates that divide X into even coarse extents
    bs = tuple(int(rng.choice([b for b in (1, 2, 3, 4, 6) if X[d] % b == 0 and (X[d] // b) % 2 == 0])) for d in range(4))
    nvec = int(rng.choice([1, 2, 3, 5, 8, 12, 16, 24, 32]))
    ncv = int(rng.integers(1, 40))
    Vn, phis, Xc = _mg_problem(X, bs, nvec, ncv, 6000 + seed)
    cdt = _np_c(prec)
    Vn = Vn.astype(cdt)
    phis = [q.astype(cdt) for q in phis]
    vpad, cpad = int(rng.choice([0, 0, 4, 30])), int(rng.choice([0, 0, 2, 10]))          # padded strides of V and of the coarse fields
    T = hip.Transfer(X, nvec, bs, 2, prec, pad=vpad).set_logical(Vn)
    cf = [hip.CoarseField(Xc, nvec, prec, pad=cpad).set_logical(q) for q in phis]
    ff = [hip.SpinorField(X, prec, order, pad=pad) for _ in range(ncv)]
    hip.prolongateEvecs(ff, cf, T)
    sg2 = sigmas(ncv)
    ref2 = np.zeros(16 * V, dtype=np.complex128)
    for n in range(ncv):
        psi = orc.prolongate(phis[n].astype(np.complex128), Vn.astype(np.complex128), X, bs)
        assert rel_err(ff[n].get_logical(), psi) < (1e-13 if prec == 8 else 2e-6), (X, bs, nvec, ncv, n)
        orc.loop_contract(ref2, psi, psi, float(np.float32(sg2[n])) if prec == 4 else sg2[n])
    loop2 = torch.zeros(16 * V, dtype=torch.complex128 if lprec == 8 else torch.complex64, device="cuda")
    hip.prolongateContractBatched(loop2, cf, sg2, T)
    assert rel_err(loop2.cpu().numpy(), ref2) < tol, (X, bs, nvec, ncv, prec, lprec)


# ---- size-independent properties at BASELINE.json's full sizes -------------------------------------------------------
def test_full_size_cfg2_properties(hip):
    """configs[1] at full size (32^4, fp64, N_ev = 200; 40 GB of eigenvectors): the oracle cannot run this in seconds, so
    check identities: sum_x L_1(x) = sum_n 1/sigma_n (unit-norm vectors); L_1 real; additivity over eigenvector subsets;
    the trace identity tr over a Hermitian Gamma is real for the ultra-local loop (all 16 G(n) are Hermitian or
    anti-Hermitian: G^dag = +-G)."""
    X = (32, 32, 32, 32)
    nev = 200
    V = int(np.prod(X))
    per = 24 * (V // 2)
    big = torch.empty(nev * per, dtype=torch.complex128, device="cuda")
    g = torch.Generator(device="cuda")
    g.manual_seed(4321)
    f = []
    for n in range(nev):
        w = torch.complex(torch.randn(per, dtype=torch.float64, device="cuda", generator=g),
                          torch.randn(per, dtype=torch.float64, device="cuda", generator=g))
        w /= torch.linalg.vector_norm(w)
        big[n * per:(n + 1) * per] = w
        f.append(hip.SpinorField(X, 8, 2, data=big[n * per:(n + 1) * per]))
        del w
    sg = sigmas(nev)
    loop = torch.zeros(16 * V, dtype=torch.complex128, device="cuda")
    hip.performLoopContractionBatched(loop, f, f, sg)
    expect = np.sum(1.0 / sg)
    assert abs(loop[:V].sum().item() - expect) < 1e-10 * expect
    assert torch.max(torch.abs(loop[:V].imag)).item() == 0.0
    # additivity: contraction over [0,73) plus [73,200) on top == contraction over all
    part = torch.zeros_like(loop)
    hip.performLoopContractionBatched(part, f[:73], f[:73], sg[:73])
    hip.performLoopContractionBatched(part, f[73:], f[73:], sg[73:])
    assert (torch.max(torch.abs(part - loop)) / torch.max(torch.abs(loop))).item() < 1e-13
    # G^dag = eta G with eta = +-1  =>  v^dag G v is real (eta = +1) or imaginary (eta = -1)
    scale = torch.max(torch.abs(loop)).item()
    for iG in range(16):
        G = orc.gamma_dense(iG)
        eta = 1 if np.array_equal(G.conj().T, G) else -1
        assert np.array_equal(G.conj().T, eta * G)
        comp = loop[V * iG:V * (iG + 1)]
        off = comp.imag if eta == 1 else comp.real
        assert torch.max(torch.abs(off)).item() < 1e-13 * scale, iG


def test_full_size_cfg3_local_shape_plus_minus_displacement_identity(hip, monkeypatch):
    """configs[2] per-GPU shape (48x48x24x24), reduced N_ev: for Gamma = 1 the loop displaced by -k mu is the complex
    conjugate of the one displaced by +k mu shifted by k mu, so their lattice sums are complex conjugates -- a
    size-independent link between the two signs, the path links and the tiled kernels (all four axes).  Reflection is
    switched OFF here: both signs are computed from the eigenvectors, so the identity links two independent computations
    (with reflection on it would only restate what reflect_kernel enforces)."""
    monkeypatch.setenv("MUGIQ_HIP_REFLECT", "0")
    X = (48, 48, 24, 24)
    nev = 6
    V = int(np.prod(X))
    vcb = V // 2
    f = []
    for n in range(nev):
        w = torch.randn(24 * vcb, dtype=torch.complex128, device="cuda")
        w /= torch.linalg.vector_norm(w)
        f.append(hip.SpinorField(X, 8, 2, data=w))
    gauge = hip.GaugeField(X, (0, 0, 0, 0), 8)
    m = torch.randn(4 * 2 * vcb, 3, 3, dtype=torch.complex128, device="cuda")
    r0 = m[:, 0] / torch.linalg.vector_norm(m[:, 0], dim=-1, keepdim=True)
    r1 = m[:, 1] - (r0.conj() * m[:, 1]).sum(-1, keepdim=True) * r0
    r1 = r1 / torch.linalg.vector_norm(r1, dim=-1, keepdim=True)
    r2 = m[:, 2] - (r0.conj() * m[:, 2]).sum(-1, keepdim=True) * r0
    r2 = r2 - (r1.conj() * r2).sum(-1, keepdim=True) * r1
    r2 = r2 / torch.linalg.vector_norm(r2, dim=-1, keepdim=True)
    q = torch.stack([r0, r1, r2], dim=1).reshape(4, 2, vcb, 9).permute(1, 0, 3, 2).contiguous()
    gauge.data.copy_(q.reshape(-1))
    sg = sigmas(nev)
    prm = hip.MugiqLoopParam(gauge=gauge).set_displace_entry_string("+x:1,3;-x:1,3;+y:1,3;-y:1,3;+z:1,3;-z:1,3;+t:1,3;-t:1,3")
    loop = hip.Loop_Mugiq(prm, f, sg)
    loop.computeCoarseLoop()
    assert all(loop.derivedFrom(i) < 0 for i in range(8))
    pos = loop.dataPos_d.view(loop.nLoop, 16, V)
    for axis in range(4):
        for k in range(3):
            plus = pos[1 + 6 * axis + k, 0].sum().item()
            minus = pos[1 + 6 * axis + 3 + k, 0].sum().item()
            assert abs(plus - np.conj(minus)) < 1e-11 * max(abs(plus), 1e-3), (axis, k, plus, minus)
    loop.close()


# ---- reflected displacement entries --------------------------------------------------------------------------------
def _oracle_pm_slots(X, nev, name, kmax, seed):
    rng = np.random.default_rng(seed)
    ev = [orc.lex_to_eo(random_spinor_lex(rng, X), X) for _ in range(nev)]
    U_lex = random_gauge_lex(rng, X)
    Uo = orc.extended_gauge_from_global(U_lex, (0, 0, 0, 0), (1, 1, 1, 1), (0, 0, 0, 0))
    cprm = orc.LoopComputeParam(["+" + name, "-" + name], [1, 1], [kmax, kmax])
    V = int(np.prod(X))
    pos = orc.compute_loop_position_space(ev, sigmas(nev), cprm, Uo, X).reshape(cprm.nLoop, 16 * V)
    return pos[1:1 + kmax], pos[1 + kmax:1 + 2 * kmax]


@pytest.mark.parametrize("prec", [8, 4])
@pytest.mark.parametrize("X,name", [((4, 6, 4, 8), "x"), ((4, 6, 4, 8), "y"), ((6, 4, 2, 4), "z"), ((4, 4, 4, 2), "t")])
def test_reflected_entry_equals_the_directly_computed_one(hip, prec, X, name):
    """mugiq_hip_reflect_displaced_loop: the "-mu" slots from the "+mu" slots and back, lengths up to past the extent."""
    kmax = 5
    plus, minus = _oracle_pm_slots(X, 2, name, kmax, 88)
    cdt = torch.complex128 if prec == 8 else torch.complex64
    d = "xyzt".index(name)
    V = int(np.prod(X))
    for k in range(1, kmax + 1):
        src = torch.from_numpy(plus[k - 1]).to(cdt).cuda()
        dst = torch.zeros(16 * V, dtype=cdt, device="cuda")
        hip.reflectDisplacedLoop(dst, src, X, d, hip.DispSignMinus, k)
        assert rel_err(dst.cpu().numpy(), minus[k - 1]) < (1e-13 if prec == 8 else 1e-6), (name, k)
        src = torch.from_numpy(minus[k - 1]).to(cdt).cuda()
        hip.reflectDisplacedLoop(dst, src, X, d, hip.DispSignPlus, k)
        assert rel_err(dst.cpu().numpy(), plus[k - 1]) < (1e-13 if prec == 8 else 1e-6), (name, k)
    with pytest.raises(hip.MugiqHipError):
        hip.reflectDisplacedLoop(dst, dst, X, d, hip.DispSignPlus, 1)                       # aliased
    with pytest.raises(hip.MugiqHipError):
        hip.reflectDisplacedLoop(dst, src, X, d, hip.DispSignPlus, 1, commDim=(1, 1, 1, 1))  # partitioned without ghost layers


@pytest.mark.parametrize("dim", [0, 1, 2, 3])
def test_reflected_entry_with_ghost_layers_matches_single_domain(hip, dim):
    """Two domains along `dim` (one process): the k boundary layers of the source slot travel through
    mugiq_hip_pack_loop_layers exactly as the driver sends them."""
    G = [4, 4, 4, 4]
    G[dim] = 8
    G = tuple(G)
    grid = [1, 1, 1, 1]
    grid[dim] = 2
    name = "xyzt"[dim]
    kmax = 3
    plus, minus = _oracle_pm_slots(G, 2, name, kmax, 99)
    l = [G[d] // grid[d] for d in range(4)]
    Vg, Vl = int(np.prod(G)), int(np.prod(l))
    ranks = [tuple(1 if (d == dim and r == 1) else 0 for d in range(4)) for r in range(2)]
    comm = tuple(1 if d == dim else 0 for d in range(4))

    def local(slot, r):
        out = np.empty(16 * Vl, dtype=np.complex128)
        for ig in range(16):
            gl = orc.eo_to_lex(slot[Vg * ig:Vg * (ig + 1)].reshape(2, Vg // 2), G)
            out[Vl * ig:Vl * (ig + 1)] = orc.lex_to_eo(orc.local_block(gl, r, grid), l).reshape(-1)
        return out

    fcb = Vl // 2 // l[dim]
    for k in range(1, kmax + 1):
        for dst_sign, src_all, ref_all, high in ((hip.DispSignMinus, plus, minus, 1), (hip.DispSignPlus, minus, plus, 0)):
            src = {r: torch.from_numpy(local(src_all[k - 1], r)).cuda() for r in ranks}
            packed = {}
            for r in ranks:
                packed[r] = torch.zeros(32 * k * fcb, dtype=torch.complex128, device="cuda")
                hip.packLoopLayers(packed[r], src[r], l, dim, high, k)
            for i, r in enumerate(ranks):
                dst = torch.zeros(16 * Vl, dtype=torch.complex128, device="cuda")
                hip.reflectDisplacedLoop(dst, src[r], l, dim, dst_sign, k, comm, packed[ranks[1 - i]])   # 2 ranks: both neighbours are the other one
                assert rel_err(dst.cpu().numpy(), local(ref_all[k - 1], r)) < 1e-13, (name, k, dst_sign, r)


# ---- full-size property tests of the remaining BASELINE.json configurations ---------------------------------------
def _device_evecs(hip, X, nev, prec, order, seed):
    """N_ev unit-norm Gaussian eigenvectors generated on the device, one allocation; returns (fields, ||v_n||^2 in fp64
    of the values actually stored)."""
    vcb = int(np.prod(X)) // 2
    per = 24 * vcb
    cdt = torch.complex128 if prec == 8 else torch.complex64
    big = torch.empty(nev * per, dtype=cdt, device="cuda")
    g = torch.Generator(device="cuda").manual_seed(seed)
    f, n2 = [], []
    for n in range(nev):
        w = torch.complex(torch.randn(per, dtype=torch.float64, device="cuda", generator=g),
                          torch.randn(per, dtype=torch.float64, device="cuda", generator=g))
        w /= torch.linalg.vector_norm(w)
        v = big[n * per:(n + 1) * per]
        v.copy_(w.to(cdt))
        n2.append(float(torch.sum(v.real.double() ** 2 + v.imag.double() ** 2).item()))
        f.append(hip.SpinorField(X, prec, order, data=v))
        del w
    return big, f, np.array(n2)


def test_full_size_cfg3_mixed_precision_and_projection_properties(hip, record_max):
    """BASELINE.json configs[3] at its per-GPU size: 64x64x32x16, fp32 FLOAT4 eigenvectors, N_ev = 600 (121 GB), fp32 and
    mixed-precision loops, momentum projection onto p^2 <= 9.  Size-independent properties:
      * Gamma = 1: sum_x L(x) = sum_n ||v_n||^2 / sigma_n -- mixed precision accumulates the fp32 inputs in fp64, so it equals
        the fp64 evaluation on the rounded inputs to fp64 rounding; the fp32 loop agrees to north_star's 1e-5;
      * mixed == fp32 loop to 1e-5 on every element (relative to the largest);
      * the separable projection (reorder + x sum fused, then y, z) == the dense phase-matrix product of the reference's
        formulation (lib/loop_mugiq.cpp:343-378), and its p = 0 row == the plain spatial sum."""
    torch.cuda.empty_cache()
    X, nev = (64, 64, 32, 16), 600
    V = int(np.prod(X))
    big, f, n2 = _device_evecs(hip, X, nev, 4, 4, 2718)
    sg = sigmas(nev)
    inv = 1.0 / np.float32(sg).astype(np.float64)              # the kernel divides by sigma cast to Float (contract_util.cuh:130-134)
    loop64 = torch.zeros(16 * V, dtype=torch.complex128, device="cuda")
    hip.performLoopContractionBatched(loop64, f, f, sg)         # complex128 loops over fp32 fields = mixed mode
    expect = float(np.sum(n2 * inv))
    got = loop64[:V].sum().item()
    record_max("cfg3_full_size_mixed_gamma1_sum_rel", abs(got - expect) / expect)
    assert abs(got - expect) < 1e-11 * expect and torch.max(torch.abs(loop64[:V].imag)).item() == 0.0
    loop32 = torch.zeros(16 * V, dtype=torch.complex64, device="cuda")
    hip.performLoopContractionBatched(loop32, f, f, sg)
    e32 = (torch.max(torch.abs(loop32.to(torch.complex128) - loop64)) / torch.max(torch.abs(loop64))).item()
    record_max("cfg3_full_size_fp32_vs_mixed", e32)
    assert e32 < 1e-5
    del big, f, loop32
    torch.cuda.empty_cache()
    # projection of the mixed-precision loop buffer: separable (one pass over the even-odd buffer) vs dense (reorder + product)
    moms = momenta_p2_le(9)
    Nmom, locT, locV3 = len(moms), X[3], X[0] * X[1] * X[2]
    sep = torch.zeros(16 * locT * Nmom, dtype=torch.complex128, device="cuda")
    hip.convertAndProject(sep, loop64, 16, 1, moms, -1, X, X)
    mp_ = torch.zeros(16 * V, dtype=torch.complex128, device="cuda")
    hip.convertIdxOrder_mapGamma(mp_, loop64, 16, 1, 2, V // 2, X)
    ph = torch.zeros(locV3 * Nmom, dtype=torch.complex128, device="cuda")
    hip.createPhaseMatrixGPU(ph, moms, locV3, Nmom, -1, X, X)
    dense = torch.zeros_like(sep)
    hip.momentumProjection(dense, mp_, ph, locT, 16, locV3, Nmom)
    e = (torch.max(torch.abs(sep - dense)) / torch.max(torch.abs(dense))).item()
    record_max("cfg3_full_size_separable_vs_dense", e)
    assert e < 1e-12
    i0 = moms.index((0, 0, 0))
    # dataMom index t + locT*ig + locT*16*im; the reordered buffer is [v3][nData][t] with the G -> g5 G map applied
    p0 = sep.view(Nmom, 16, locT)[i0]
    plain = mp_.view(locV3, 16, locT).sum(0)
    assert (torch.max(torch.abs(p0 - plain)) / torch.max(torch.abs(plain))).item() < 1e-12


def test_full_size_cfg4_mg_plans_agree(hip, monkeypatch, record_max):
    """BASELINE.json configs[4] at full size: 32^4 fp64, 4^4 aggregates, n_vec = 24, N_ev = 200 coarse eigenvectors.
    The three routes to the MG ultra-local loop agree: coarse-grid plan (outer product on the coarse grid + one congruence
    per fine site) == direct plan (prolong + contract per eigenvector, fused) == prolongateEvecs then the ordinary batched
    contraction (the reference's sequence, lib/loop_mugiq.cpp:482,501-502)."""
    torch.cuda.empty_cache()
    X, bs, nvec, nev = (32, 32, 32, 32), (4, 4, 4, 4), 24, 200
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
    sg = sigmas(nev)
    out = {}
    for plan in ("coarse", "direct"):
        if plan == "direct":
            monkeypatch.setenv("MUGIQ_HIP_MG_PLAN", "direct")
        else:
            monkeypatch.delenv("MUGIQ_HIP_MG_PLAN", raising=False)
        out[plan] = torch.zeros(16 * V, dtype=torch.complex128, device="cuda")
        hip.prolongateContractBatched(out[plan], cf, sg, T)
    monkeypatch.delenv("MUGIQ_HIP_MG_PLAN", raising=False)
    big = torch.empty(nev * 24 * (V // 2), dtype=torch.complex128, device="cuda")
    ff = [hip.SpinorField(X, 8, 2, data=big[n * 24 * (V // 2):(n + 1) * 24 * (V // 2)]) for n in range(nev)]
    hip.prolongateEvecs(ff, cf, T)
    out["separate"] = torch.zeros(16 * V, dtype=torch.complex128, device="cuda")
    hip.performLoopContractionBatched(out["separate"], ff, ff, sg)
    scale = torch.max(torch.abs(out["separate"])).item()
    for a, b in (("coarse", "direct"), ("coarse", "separate"), ("direct", "separate")):
        e = torch.max(torch.abs(out[a] - out[b])).item() / scale
        record_max("cfg4_full_size_%s_vs_%s" % (a, b), e)
        assert e < 1e-12, (a, b, e)
    # Gamma = 1 slot: sum_x L(x) = sum_n ||P c_n||^2 / sigma_n with the norms taken from the prolonged vectors
    n2 = np.array([float(torch.sum(f.data.real ** 2 + f.data.imag ** 2).item()) for f in ff])
    expect = float(np.sum(n2 / sg))
    got = out["coarse"][:V].sum().item()
    assert abs(got - expect) < 1e-11 * abs(expect)
    # and through the driver (Loop_Mugiq with a transfer operator, no displacement entries -> fused route, fine vectors never stored)
    del big, ff
    torch.cuda.empty_cache()
    loop = hip.Loop_Mugiq(hip.MugiqLoopParam(), cf, sg, transfer=T)
    loop.computeCoarseLoop()
    e = torch.max(torch.abs(loop.dataPos_d - out["coarse"])).item() / scale
    assert e < 1e-13
    loop.close()


def test_spinor_alloc_copy_zero_helpers(hip):
    """mugiq_hip_alloc_spinor_like / _copy_spinor / _zero_spinor / _free_spinor: what Displace asks of QUDA's ColorSpinorField for
    its auxiliary vector (lib/displace.cpp:26-30,42,59) -- a zeroed twin with the same geometry, a value copy, blas::zero."""
    import ctypes
    from mugiq_amd import _lib
    lib = _lib.load()
    X = (4, 6, 4, 8)
    rng = np.random.default_rng(5)
    v = orc.lex_to_eo(random_spinor_lex(rng, X), X)
    src = hip.SpinorField(X, 8, 2, pad=6).set_logical(v)
    d_src = src.desc()
    twin = _lib.SpinorDesc()
    ghost = (ctypes.c_int * 4)(0, 0, 1, 0)
    assert lib.mugiq_hip_alloc_spinor_like(ctypes.byref(twin), ctypes.byref(d_src), 0, ghost) == 0
    assert twin.data and twin.stride == d_src.stride and twin.parity_offset == d_src.parity_offset and twin.precision == 8
    assert twin.ghost[2][0] and twin.ghost[2][1] and not twin.ghost[3][0]
    n = 2 * src.parity_offset
    back = torch.empty(n, dtype=torch.complex128, device="cuda")
    def fetch(desc):
        assert torch.cuda.current_stream().cuda_stream is not None
        import ctypes as C
        hipMemcpy = C.CDLL("libamdhip64.so").hipMemcpy
        assert hipMemcpy(C.c_void_p(back.data_ptr()), C.c_void_p(desc.data), C.c_size_t(n * 16), C.c_int(3)) == 0   # device to device
        torch.cuda.synchronize()
        return back.cpu().numpy().copy()
    assert not fetch(twin).any()                                                     # QUDA_ZERO_FIELD_CREATE
    assert lib.mugiq_hip_copy_spinor(ctypes.byref(twin), ctypes.byref(d_src), None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(fetch(twin), src.data.cpu().numpy())
    assert lib.mugiq_hip_zero_spinor(ctypes.byref(twin), None) == 0
    torch.cuda.synchronize()
    assert not fetch(twin).any()
    other = hip.SpinorField(X, 8, 4)                                                  # different order: the copy must refuse
    d_o = other.desc()
    assert lib.mugiq_hip_copy_spinor(ctypes.byref(twin), ctypes.byref(d_o), None) != 0
    assert lib.mugiq_hip_free_spinor(ctypes.byref(twin)) == 0 and not twin.data and not twin.ghost[2][0]
