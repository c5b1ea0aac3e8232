"""Analytic known-answer tests that anchor the CPU oracle (SURVEY.md §8c list, items 1-9).

The reference ships no golden vectors, so the oracle is pinned by identities that follow from the
cited reference code alone.  CPU-only; runs in seconds.
"""
import itertools

import numpy as np
import pytest

from util import orc, random_gauge_lex, random_spinor_lex, random_su3, unit_gauge_lex, gauge_eo_single_domain, \
    sigmas, momenta_p2_le, rel_err

LATTICES = [(4, 4, 4, 4), (4, 4, 4, 8), (6, 4, 8, 2)]


# ---- (1) gamma-table self-consistency ---------------------------------------------------------
def test_gamma_tables_are_ordered_products():
    g = [orc.gamma_dense(n) for n in (1, 2, 4, 8)]          # g1, g2, g3, g4
    for n in range(16):
        prod = np.eye(4, dtype=complex)
        for b in range(4):
            if (n >> b) & 1:
                prod = prod @ g[b]
        assert np.array_equal(orc.gamma_dense(n), prod), n


def test_gamma_clifford_hermitian_g5():
    g = [orc.gamma_dense(n) for n in (1, 2, 4, 8)]
    for mu in range(4):
        assert np.array_equal(g[mu], g[mu].conj().T)
        for nu in range(4):
            anti = g[mu] @ g[nu] + g[nu] @ g[mu]
            assert np.array_equal(anti, 2.0 * np.eye(4) * (mu == nu))
    assert np.array_equal(orc.gamma_dense(15), np.diag([1, 1, -1, -1]).astype(complex))


# ---- (2) gamma5 map ---------------------------------------------------------------------------
def test_gamma5_map_matches_names():
    g5 = orc.gamma_dense(15)
    sign = orc.gamma_map_sign()
    for ig in range(16):
        j = orc.INDEX_MAP_GAMMA[ig]
        assert j == 15 - ig
        lhs = g5 @ orc.gamma_dense(j)
        if ig in (1, 4):       # deliberate: output slots 14 / 11 are named g5g1 / g5g3 = -g5*G(14), -g5*G(11)
            assert np.array_equal(lhs, -orc.gamma_dense(ig)) and sign[ig] == 1.0
        else:
            assert np.array_equal(lhs, sign[ig] * orc.gamma_dense(ig)), ig
    assert orc.GAMMA_NAMES[14] == "g5g1" and orc.GAMMA_NAMES[11] == "g5g3"


# ---- (9) even-odd round trip -------------------------------------------------------------------
@pytest.mark.parametrize("X", LATTICES)
def test_even_odd_round_trip(X):
    vcb = int(np.prod(X)) // 2
    seen = set()
    for pty in range(2):
        c = orc.get_coords(np.arange(vcb), X, pty)
        assert np.all((c.sum(axis=1) & 1) == pty)
        assert np.array_equal(orc.link_index(c, X), np.arange(vcb))
        for row in c:
            seen.add(tuple(row))
        assert np.all(c >= 0) and np.all(c < np.asarray(X))
    assert len(seen) == 2 * vcb
    f = np.arange(int(np.prod(X))).reshape(X[3], X[2], X[1], X[0])
    assert np.array_equal(orc.eo_to_lex(orc.lex_to_eo(f, X), X), f)


@pytest.mark.parametrize("order", [orc.FLOAT2, orc.FLOAT4])
def test_native_spinor_layout_round_trip(order):
    rng = np.random.default_rng(5)
    v = rng.standard_normal((2, 32, 4, 3)) + 1j * rng.standard_normal((2, 32, 4, 3))
    for stride in (32, 40):
        buf = orc.spinor_to_native(v, order, stride=stride)
        assert np.array_equal(orc.spinor_from_native(buf, order, 32, stride=stride), v)
    # FLOAT4: two consecutive complex (k, k+1) of one site are adjacent in memory
    if order == orc.FLOAT4:
        assert orc.spinor_native_index(order, 0, 7, 0, 1, 32, 384) == orc.spinor_native_index(order, 0, 7, 0, 0, 32, 384) + 1


# ---- (3)+(4) contraction ------------------------------------------------------------------------
def test_contraction_unit_gamma_and_dense_gamma():
    X = (4, 4, 4, 4)
    rng = np.random.default_rng(11)
    nev = 3
    evL = [orc.lex_to_eo(random_spinor_lex(rng, X), X) for _ in range(nev)]
    evR = [orc.lex_to_eo(random_spinor_lex(rng, X), X) for _ in range(nev)]
    sg = sigmas(nev)
    V = int(np.prod(X))
    loop = np.zeros(16 * V, dtype=complex)
    for n in range(nev):
        orc.loop_contract(loop, evL[n], evR[n], sg[n])
    # dense check: loop[x, iG] = sum_n sigma_n^-1 vL^dag(x) G(iG) vR(x)   (colour traced)
    for iG in range(16):
        G = orc.gamma_dense(iG)
        ref = np.zeros(V, dtype=complex)
        for n in range(nev):
            L = evL[n].reshape(V, 4, 3)
            R = evR[n].reshape(V, 4, 3)
            ref += np.einsum("xbc,ba,xac->x", L.conj(), G, R) / sg[n]
        assert rel_err(loop[V * iG:V * (iG + 1)], ref) < 1e-14
    # Gamma = 1 slot with L == R: sum_x loop = sum_n ||v_n||^2 / sigma_n = sum_n 1/sigma_n
    loop1 = np.zeros(16 * V, dtype=complex)
    for n in range(nev):
        orc.loop_contract(loop1, evL[n], evL[n], sg[n])
    assert abs(loop1[:V].sum() - np.sum(1.0 / sg)) < 1e-10 * np.sum(1.0 / sg)
    assert np.max(np.abs(loop1[:V].imag)) < 1e-16 * np.max(np.abs(loop1[:V].real)) + 1e-30


# ---- (4)+(7) unit gauge: displacement is a pure shift; L_mu steps wrap around --------------------
@pytest.mark.parametrize("X", LATTICES)
def test_unit_gauge_displacement_is_shift_and_wraps(X):
    rng = np.random.default_rng(3)
    psi_lex = random_spinor_lex(rng, X)
    psi = orc.lex_to_eo(psi_lex, X)
    U = gauge_eo_single_domain(unit_gauge_lex(X), X)
    axis_of_dir = {0: 3, 1: 2, 2: 1, 3: 0}                 # lex array is [T, Z, Y, X]
    for dirn in range(4):
        for sign in (orc.DISP_SIGN_PLUS, orc.DISP_SIGN_MINUS):
            out = orc.covariant_displacement(psi, U, dirn, sign, X)
            shift = -1 if sign == orc.DISP_SIGN_PLUS else 1   # dst(x) = src(x + mu) for sign +
            expect = np.roll(psi_lex, shift, axis=axis_of_dir[dirn])
            assert np.array_equal(orc.eo_to_lex(out, X), expect)
            cur = psi
            for _ in range(X[dirn]):
                cur = orc.covariant_displacement(cur, U, dirn, sign, X)
            assert np.array_equal(cur, psi)


# ---- (5) D_-mu D_+mu = 1 -------------------------------------------------------------------------
def test_forward_backward_displacement_is_identity():
    X = (4, 4, 4, 8)
    rng = np.random.default_rng(7)
    psi = orc.lex_to_eo(random_spinor_lex(rng, X), X)
    U = gauge_eo_single_domain(random_gauge_lex(rng, X), X)
    for dirn in range(4):
        fwd = orc.covariant_displacement(psi, U, dirn, orc.DISP_SIGN_PLUS, X)
        back = orc.covariant_displacement(fwd, U, dirn, orc.DISP_SIGN_MINUS, X)
        assert rel_err(back, psi) < 1e-12
        bwd = orc.covariant_displacement(psi, U, dirn, orc.DISP_SIGN_MINUS, X)
        back = orc.covariant_displacement(bwd, U, dirn, orc.DISP_SIGN_PLUS, X)
        assert rel_err(back, psi) < 1e-12


# ---- explicit formula check in lexicographic coordinates -----------------------------------------
def test_displacement_matches_lexicographic_formula():
    X = (4, 6, 4, 2)
    rng = np.random.default_rng(17)
    psi_lex = random_spinor_lex(rng, X)
    U_lex = random_gauge_lex(rng, X)
    psi = orc.lex_to_eo(psi_lex, X)
    U = gauge_eo_single_domain(U_lex, X)
    axis_of_dir = {0: 3, 1: 2, 2: 1, 3: 0}
    for dirn in range(4):
        ax = axis_of_dir[dirn]
        # +: U_mu(x) psi(x+mu)
        exp = np.einsum("tzyxab,tzyxsb->tzyxsa", U_lex[dirn], np.roll(psi_lex, -1, axis=ax))
        got = orc.eo_to_lex(orc.covariant_displacement(psi, U, dirn, orc.DISP_SIGN_PLUS, X), X)
        assert rel_err(got, exp) < 1e-15
        # -: U_mu^dag(x-mu) psi(x-mu)
        Ub = np.roll(U_lex[dirn], 1, axis=ax)
        exp = np.einsum("tzyxba,tzyxsb->tzyxsa", Ub.conj(), np.roll(psi_lex, 1, axis=ax))
        got = orc.eo_to_lex(orc.covariant_displacement(psi, U, dirn, orc.DISP_SIGN_MINUS, X), X)
        assert rel_err(got, exp) < 1e-15


# ---- (6) gauge covariance of displaced loops -------------------------------------------------------
def test_displaced_loop_is_gauge_invariant():
    X = (4, 4, 4, 4)
    rng = np.random.default_rng(23)
    nev = 2
    ev_lex = [random_spinor_lex(rng, X) for _ in range(nev)]
    U_lex = random_gauge_lex(rng, X)
    g = random_su3(rng, (X[3], X[2], X[1], X[0]))
    axis_of_dir = {0: 3, 1: 2, 2: 1, 3: 0}
    Ug = np.stack([np.einsum("tzyxab,tzyxbc,tzyxdc->tzyxad", g, U_lex[mu],
                             np.roll(g, -1, axis=axis_of_dir[mu]).conj()) for mu in range(4)])
    evg = [np.einsum("tzyxab,tzyxsb->tzyxsa", g, v) for v in ev_lex]
    cprm = orc.LoopComputeParam(["+x", "-y", "+t", "-z"], [1, 1, 2, 1], [2, 1, 3, 3])
    assert cprm.nLoop == 1 + 2 + 1 + 2 + 3 and cprm.nLoopOffset == [1, 3, 4, 6]
    a = orc.compute_loop_position_space([orc.lex_to_eo(v, X) for v in ev_lex], sigmas(nev), cprm,
                                        gauge_eo_single_domain(U_lex, X), X)
    b = orc.compute_loop_position_space([orc.lex_to_eo(v, X) for v in evg], sigmas(nev), cprm,
                                        gauge_eo_single_domain(Ug, X), X)
    assert rel_err(b, a) < 1e-13


# ---- (8) FT: phase matrix + projection vs direct DFT ------------------------------------------------
def test_phase_matrix_and_projection_vs_direct_dft():
    X = (4, 6, 4, 8)
    rng = np.random.default_rng(29)
    V = int(np.prod(X))
    vcb = V // 2
    nLoop = 2
    nData = 16 * nLoop
    dataPos = rng.standard_normal(nData * V) + 1j * rng.standard_normal(nData * V)
    moms = momenta_p2_le(2)
    locV3 = X[0] * X[1] * X[2]
    for FTSign in (+1, -1):
        ph = orc.phase_matrix(moms, locV3, FTSign, X, X)
        mp = orc.convert_idx_order_map_gamma(dataPos, nData, nLoop, 2, vcb, X)
        mom = orc.momentum_projection_local(mp, ph, X[3], nData, locV3, len(moms))
        mom = mom.reshape(len(moms), nLoop, 16, X[3])
        # direct: lexicographic data, sum over x,y,z with exp(i FTSign 2pi p.x/L)
        sign = orc.gamma_map_sign()
        lex = np.stack([orc.eo_to_lex(dataPos[V * i:V * (i + 1)].reshape(2, vcb), X) for i in range(nData)])
        xs, ys, zs = np.arange(X[0]), np.arange(X[1]), np.arange(X[2])
        for im, p in enumerate(moms):
            phase = np.exp(1j * FTSign * 2 * np.pi * (p[2] * zs[:, None, None] / X[2] + p[1] * ys[None, :, None] / X[1]
                                                       + p[0] * xs[None, None, :] / X[0]))
            direct = np.einsum("itzyx,zyx->it", lex, phase)        # [idata, t]
            for iL in range(nLoop):
                for ig in range(16):
                    exp = sign[ig] * direct[ig + 16 * iL]
                    assert rel_err(mom[im, iL, 15 - ig], exp) < 1e-12
    # p = 0 is the plain spatial sum
    ph0 = orc.phase_matrix([(0, 0, 0)], locV3, 1, X, X)
    assert np.array_equal(ph0, np.ones(locV3, dtype=complex))


# ---- multi-domain == single-domain ---------------------------------------------------------------------
def _emulated_multi_domain_loop(ev_lex, U_lex, sg, cprm, G, grid):
    ranks = list(itertools.product(*[range(g) for g in grid]))          # (cx, cy, cz, ct)
    l = [G[d] // grid[d] for d in range(4)]
    comm_dim = [1 if grid[d] > 1 else 0 for d in range(4)]
    brd = [2 * c for c in comm_dim]                                      # lib/displace.cpp:16
    V = int(np.prod(l))
    perLoop = 16 * V
    Uloc = {r: orc.extended_gauge_from_global(U_lex, r, grid, brd) for r in ranks}
    data = {r: np.zeros(perLoop * cprm.nLoop, dtype=complex) for r in ranks}

    def nbr(r, d, s):
        q = list(r)
        q[d] = (q[d] + s) % grid[d]
        return tuple(q)

    for idx in range(-1, cprm.nDispEntries):
        for n, v_lex in enumerate(ev_lex):
            vL = {r: orc.lex_to_eo(orc.local_block(v_lex, r, grid), l) for r in ranks}
            if idx < 0:
                for r in ranks:
                    orc.loop_contract(data[r][:perLoop], vL[r], vL[r], sg[n])
                continue
            dirn, sign = orc.parse_displacement(cprm.dispString[idx])
            off = perLoop * cprm.nLoopOffset[idx]
            vR = {r: vL[r].copy() for r in ranks}
            cnt = 0
            for idisp in range(1, cprm.dispStop[idx] + 1):
                ghost = {}
                for r in ranks:                                           # exchangeGhostVec, all dims, both dirs
                    gh = [[None, None] for _ in range(4)]
                    for d in range(4):
                        if comm_dim[d]:
                            gh[d][1] = orc.pack_face(vR[nbr(r, d, +1)], l, d, high=0)
                            gh[d][0] = orc.pack_face(vR[nbr(r, d, -1)], l, d, high=1)
                    ghost[r] = gh
                vR = {r: orc.covariant_displacement(vR[r], Uloc[r], dirn, sign, l, comm_dim, brd, ghost[r])
                      for r in ranks}
                if cprm.dispStart[idx] <= idisp <= cprm.dispStop[idx]:
                    for r in ranks:
                        s0 = off + perLoop * cnt
                        orc.loop_contract(data[r][s0:s0 + perLoop], vL[r], vR[r], sg[n])
                    cnt += 1
    return data, l


@pytest.mark.parametrize("grid", [(1, 1, 1, 2), (1, 1, 2, 2), (2, 1, 1, 1), (1, 2, 1, 1)])
def test_multi_domain_equals_single_domain(grid):
    G = (4, 4, 4, 8)
    rng = np.random.default_rng(31)
    nev = 2
    ev_lex = [random_spinor_lex(rng, G) for _ in range(nev)]
    U_lex = random_gauge_lex(rng, G)
    sg = sigmas(nev)
    cprm = orc.LoopComputeParam(["+t", "-t", "+z", "-z", "+x", "-y"], [1, 1, 1, 2, 1, 1], [2, 1, 1, 2, 1, 2])
    single = orc.compute_loop_position_space([orc.lex_to_eo(v, G) for v in ev_lex], sg, cprm,
                                             gauge_eo_single_domain(U_lex, G), G)
    Vg = int(np.prod(G))
    data, l = _emulated_multi_domain_loop(ev_lex, U_lex, sg, cprm, G, grid)
    Vl = int(np.prod(l))
    for r, d in data.items():
        for idata in range(cprm.nData):
            glob = orc.eo_to_lex(single[Vg * idata:Vg * (idata + 1)].reshape(2, Vg // 2), G)
            loc = orc.eo_to_lex(d[Vl * idata:Vl * (idata + 1)].reshape(2, Vl // 2), l)
            assert rel_err(loc, orc.local_block(glob, r, grid)) < 1e-14


def test_parse_displacement_strings_and_bookkeeping():
    assert orc.parse_displacement("+x") == (0, orc.DISP_SIGN_PLUS)
    assert orc.parse_displacement("-t") == (3, orc.DISP_SIGN_MINUS)
    with pytest.raises(ValueError):
        orc.parse_displacement("+w")
    e, s, a, b = orc.parse_disp_entry_string("+z:1,8;-x:3;+y:5,2")
    assert s == ["+z", "-x", "+y"] and a == [1, 3, 5] and b == [8, 3, 2]
    c = orc.LoopComputeParam(s, a, b)
    assert c.dispStart == [1, 3, 2] and c.dispStop == [8, 3, 5]       # start > stop are swapped
    assert c.nLoopPerEntry == [8, 1, 4] and c.nLoopOffset == [1, 9, 10] and c.nLoop == 14 and c.nData == 224
    assert orc.LoopComputeParam(doNonLocal=False).nLoop == 1


# ---- the plain-C restatement (cpu_baseline "port") agrees with the numpy oracle ------------------------
@pytest.mark.parametrize("order", [orc.FLOAT2, orc.FLOAT4])
@pytest.mark.parametrize("dtype", [np.complex128, np.complex64])
def test_c_oracle_matches_numpy_oracle(order, dtype):
    from oracle import c_oracle
    X = (4, 4, 6, 2)
    rng = np.random.default_rng(41)
    nev = 3
    V = int(np.prod(X))
    evL = [orc.lex_to_eo(random_spinor_lex(rng, X), X).astype(dtype) for _ in range(nev)]
    evR = [orc.lex_to_eo(random_spinor_lex(rng, X), X).astype(dtype) for _ in range(nev)]
    sg = sigmas(nev)
    fdt = np.float64 if dtype == np.complex128 else np.float32
    ref = np.zeros(16 * V, dtype=dtype)
    for n in range(nev):
        orc.loop_contract(ref, evL[n], evR[n], sg[n], fdt)
    stride = V // 2 + 6
    bl = [orc.spinor_to_native(v, order, stride=stride) for v in evL]
    br = [orc.spinor_to_native(v, order, stride=stride) for v in evR]
    got = np.zeros(16 * V, dtype=dtype)
    c_oracle.loop_contract_native(got, bl, br, sg, V // 2, stride, 12 * stride, order)
    assert rel_err(got, ref) < (1e-15 if dtype == np.complex128 else 1e-6)
    assert c_oracle.num_threads() >= 1


# ---- f2: prolongator -----------------------------------------------------------------------------------------
def test_prolongator_block_structure_and_adjoint_identity():
    """P is block-local and chirality-preserving; with block-orthonormal null vectors P^dag P = 1 on the coarse space."""
    X, bs, nvec = (8, 4, 4, 8), (4, 2, 2, 4), 6
    Xc = [X[d] // bs[d] for d in range(4)]
    rng = np.random.default_rng(12)
    vcb, vcbc = int(np.prod(X)) // 2, int(np.prod(Xc)) // 2
    V = rng.standard_normal((2, vcb, 4, 3, nvec)) + 1j * rng.standard_normal((2, vcb, 4, 3, nvec))
    cp, cx = orc.fine_to_coarse_map(X, bs)
    # every coarse site owns prod(bs) fine sites, and the map agrees with coordinates
    counts = np.zeros((2, vcbc), dtype=int)
    np.add.at(counts, (cp.reshape(-1), cx.reshape(-1)), 1)
    assert np.all(counts == int(np.prod(bs)))
    # block-orthonormalise V per (aggregate, chirality): QR over the rows (fine site in block, spin in chirality, colour)
    for p in range(2):
        for xc in range(vcbc):
            for chi in range(2):
                rows = [(q, i) for q in range(2) for i in np.nonzero((cp[q] == p) & (cx[q] == xc))[0]]
                M = np.concatenate([V[q, i, 2 * chi:2 * chi + 2].reshape(6, nvec) for q, i in rows])
                Q, _ = np.linalg.qr(M)
                for r, (q, i) in enumerate(rows):
                    V[q, i, 2 * chi:2 * chi + 2] = Q[6 * r:6 * r + 6].reshape(2, 3, nvec)
    phi = rng.standard_normal((2, vcbc, 2, nvec)) + 1j * rng.standard_normal((2, vcbc, 2, nvec))
    psi = orc.prolongate(phi, V, X, bs)
    assert abs(np.linalg.norm(psi) - np.linalg.norm(phi)) < 1e-12 * np.linalg.norm(phi)
    # restrict back: R = P^dag
    back = np.zeros_like(phi)
    for pty in range(2):
        for s in range(4):
            contrib = np.einsum("xcj,xc->xj", V[pty, :, s].conj(), psi[pty, :, s])
            np.add.at(back, (cp[pty], cx[pty], s // 2), contrib)
    assert rel_err(back, phi) < 1e-12
    # a coarse vector supported on one aggregate / chirality prolongs to that aggregate's sites and those spins only
    one = np.zeros_like(phi)
    one[1, 3, 1, 2] = 1.0
    p1 = orc.prolongate(one, V, X, bs)
    support = (np.abs(p1).sum(axis=(2, 3)) > 0)
    assert np.array_equal(support, (cp == 1) & (cx == 3)) and np.all(p1[:, :, :2] == 0)
    # native layouts round-trip through the index formulas
    buf = orc.coarse_to_native(phi)
    assert buf[1 * (2 * nvec * vcbc) + (1 * nvec + 2) * vcbc + 3] == phi[1, 3, 1, 2]
    vb = orc.nullvec_to_native(V)
    assert vb[0 * (12 * nvec * vcb) + ((3 * 2 + 1) * nvec + 4) * vcb + 5] == V[0, 5, 2, 1, 4]


def _mg_hierarchy(rng, X0, bss, nvecs):
    """Synthetic 1 + len(bss) level hierarchy: Vs[l], lattice Xs[l] (finer side), block sizes, and the coarsest lattice."""
    Xs, Vs = [tuple(X0)], []
    for l, (bs, nv) in enumerate(zip(bss, nvecs)):
        X = Xs[l]
        vcb = int(np.prod(X)) // 2
        ns, nc = (4, 3) if l == 0 else (2, nvecs[l - 1])
        Vs.append((rng.standard_normal((2, vcb, ns, nc, nv)) + 1j * rng.standard_normal((2, vcb, ns, nc, nv))) / np.sqrt(ns * nc * nv))
        Xs.append(tuple(X[d] // bs[d] for d in range(4)))
    return Vs, Xs[:-1], Xs[-1]


def test_multilevel_prolongation_is_the_composition_of_the_levels():
    """prolongateEvec with nCoarseLevels = 2 (lib/loop_mugiq.cpp:306-314): P_0 P_1 phi.  Checks: linearity; chirality is
    preserved through BOTH levels (spin_bs 2 then 1: upper spins of the fine vector see only chirality 0 of the coarsest
    one); locality (a coarsest-site delta prolongs onto exactly the fine sites of its aggregate of aggregates); and the
    explicit double sum over (j1, j2) at a few sites."""
    rng = np.random.default_rng(31)
    X0, bss, nvecs = (8, 8, 8, 16), [(2, 2, 2, 2), (2, 2, 2, 4)], [4, 3]
    Vs, Xs, Xcc = _mg_hierarchy(rng, X0, bss, nvecs)
    vcbcc = int(np.prod(Xcc)) // 2
    phi = rng.standard_normal((2, vcbcc, 2, nvecs[1])) + 1j * rng.standard_normal((2, vcbcc, 2, nvecs[1]))
    chi = rng.standard_normal(phi.shape) + 1j * rng.standard_normal(phi.shape)
    psi = orc.prolongate_levels(phi, Vs, Xs, bss)
    assert psi.shape == (2, int(np.prod(X0)) // 2, 4, 3)
    assert rel_err(orc.prolongate_levels(2j * phi + chi, Vs, Xs, bss), 2j * psi + orc.prolongate_levels(chi, Vs, Xs, bss)) < 1e-13
    only0 = phi.copy()
    only0[:, :, 1] = 0
    p0 = orc.prolongate_levels(only0, Vs, Xs, bss)
    assert np.all(p0[:, :, 2:] == 0) and rel_err(p0[:, :, :2], psi[:, :, :2]) < 1e-14
    # explicit double sum at fine sites: psi(x; s, c) = sum_j1 V0(x; s,c,j1) sum_j2 V1(X1(x); s/2, j1, j2) phi(X2(X1(x)); s/2, j2)
    cp0, cx0 = orc.fine_to_coarse_map(Xs[0], bss[0])
    cp1, cx1 = orc.fine_to_coarse_map(Xs[1], bss[1])
    for pty, x in ((0, 0), (1, 17), (0, 1234), (1, int(np.prod(X0)) // 2 - 1)):
        p1, x1 = cp0[pty, x], cx0[pty, x]
        p2, x2 = cp1[p1, x1], cx1[p1, x1]
        for s in range(4):
            mid = np.einsum("jk,k->j", Vs[1][p1, x1, s // 2], phi[p2, x2, s // 2])
            exp = Vs[0][pty, x, s] @ mid
            assert np.allclose(psi[pty, x, s], exp, rtol=1e-12, atol=1e-14)
    one = np.zeros_like(phi)
    one[1, 2, 0, 1] = 1.0
    sup = np.abs(orc.prolongate_levels(one, Vs, Xs, bss)).sum(axis=(2, 3)) > 0
    want = (cp1[cp0, cx0] == 1) & (cx1[cp0, cx0] == 2)
    assert np.array_equal(sup, want) and sup.sum() == int(np.prod(bss[0])) * int(np.prod(bss[1]))


# ---- (10) reflection: a minus-direction loop is the shifted conjugate of the plus-direction one -----------
def gamma_dagger_sign():
    """G(n)^dagger = eta_n G(n) for G(n) = g1^n0 g2^n1 g3^n2 g4^n3 (Hermitian, anticommuting factors):
    eta = (-1)^(m(m-1)/2), m = number of factors."""
    return np.array([(-1) ** ((bin(n).count("1") * (bin(n).count("1") - 1)) // 2) for n in range(16)])


def test_gamma_dagger_signs():
    for n in range(16):
        G = orc.gamma_dense(n)
        assert np.array_equal(G.conj().T, gamma_dagger_sign()[n] * G)


def test_minus_entry_is_shifted_conjugate_of_plus_entry():
    """W_{-k}(x) = W_{+k}(x - k mu)^dagger and [Gamma, W] = 0 give, slot by slot,
        L^-_{k,G}(x) = eta_G * conj( L^+_{k,G}(x - k mu) ),
    which lets the engine derive every "-mu" entry from the "+mu" entry of the same lengths (and vice versa)."""
    X = (4, 6, 4, 8)
    rng = np.random.default_rng(2024)
    nev = 3
    ev = [orc.lex_to_eo(random_spinor_lex(rng, X), X) for _ in range(nev)]
    U = gauge_eo_single_domain(random_gauge_lex(rng, X), X)
    V = int(np.prod(X))
    eta = gamma_dagger_sign()
    axis_of_dir = {0: 3, 1: 2, 2: 1, 3: 0}
    for d, name in enumerate("xyzt"):
        cprm = orc.LoopComputeParam(["+" + name, "-" + name], [1, 1], [3, 3])
        pos = orc.compute_loop_position_space(ev, sigmas(nev), cprm, U, X).reshape(cprm.nLoop, 16, V)
        for k in (1, 2, 3):
            plus, minus = pos[1 + (k - 1)], pos[4 + (k - 1)]
            for ig in range(16):
                p_lex = orc.eo_to_lex(plus[ig].reshape(2, V // 2), X)
                m_lex = orc.eo_to_lex(minus[ig].reshape(2, V // 2), X)
                shifted = np.roll(p_lex, k, axis=axis_of_dir[d])          # value at x - k mu
                assert rel_err(m_lex, eta[ig] * shifted.conj()) < 1e-13, (name, k, ig)


# ---- (11) the same identity in momentum space, after the G -> g5 G map of the reorder ----------------------------------------
def reflect_momentum_space(mom_src, moms, FTSign, totalL, dirn, dst_sign_plus, k):
    """Momentum-space data of the derived slot from the momentum-space data of its opposite-sign source slot:
        dst[p, ig, t] = eta(15 - ig) * exp(-+ i FTSign 2 pi p_mu k / L_mu) * conj(src[-p, ig, t])            (mu spatial)
        dst[p, ig, t] = eta(15 - ig) * conj(src[-p, ig, t +- k])                                              (mu = t)
    upper signs for a derived "+" entry (source "-").  ig is the OUTPUT channel of convertIdxOrder_mapGamma (input channel 15 - ig,
    real sign).  mom_src: [Nmom][16][totT].  The momentum list must hold -p for every p."""
    eta = gamma_dagger_sign()
    moms = [tuple(m) for m in moms]
    out = np.empty_like(mom_src)
    T = mom_src.shape[-1]
    for im, p in enumerate(moms):
        jm = moms.index(tuple(-c for c in p))
        for ig in range(16):
            v = np.conj(mom_src[jm, ig])
            if dirn < 3:
                sgn = -1.0 if dst_sign_plus else 1.0
                out[im, ig] = eta[15 - ig] * np.exp(1j * sgn * FTSign * 2.0 * np.pi * p[dirn] * k / totalL[dirn]) * v
            else:
                out[im, ig] = eta[15 - ig] * np.roll(v, -k if dst_sign_plus else k)      # value at t + k | t - k
    return out


@pytest.mark.parametrize("FTSign", [-1, 1])
def test_momentum_space_of_a_reflected_entry_follows_from_its_source(FTSign):
    """Fourier transform of L^-_k = eta conj(L^+_k(x - k mu)): a phase, the momentum reversed, a conjugation -- so the engine can
    skip the reflected slots in position space altogether when only momentum-space output is asked for."""
    X = (4, 6, 4, 8)
    rng = np.random.default_rng(77)
    nev = 2
    ev = [orc.lex_to_eo(random_spinor_lex(rng, X), X) for _ in range(nev)]
    U = gauge_eo_single_domain(random_gauge_lex(rng, X), X)
    V = int(np.prod(X))
    moms = [m for m in momenta_p2_le(3)]
    locV3 = X[0] * X[1] * X[2]
    ph = orc.phase_matrix(moms, locV3, FTSign, X, X)
    for d, name in enumerate("xyzt"):
        cprm = orc.LoopComputeParam(["+" + name, "-" + name], [1, 1], [2, 2])
        pos = orc.compute_loop_position_space(ev, sigmas(nev), cprm, U, X)
        mp = orc.convert_idx_order_map_gamma(pos, cprm.nData, cprm.nLoop, 2, V // 2, X)
        mom = orc.momentum_projection_local(mp, ph, X[3], cprm.nData, locV3, len(moms)).reshape(len(moms), cprm.nLoop, 16, X[3])
        for k in (1, 2):
            plus, minus = mom[:, 1 + (k - 1)], mom[:, 3 + (k - 1)]
            assert rel_err(reflect_momentum_space(plus, moms, FTSign, X, d, False, k), minus) < 1e-13, (name, k, "minus from plus")
            assert rel_err(reflect_momentum_space(minus, moms, FTSign, X, d, True, k), plus) < 1e-13, (name, k, "plus from minus")
