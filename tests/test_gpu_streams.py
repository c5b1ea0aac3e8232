"""GPU tests of re-entrancy: operator calls and Loop objects on DIFFERENT HIP streams are independent.

The small device tables of a call (eigenvector pointer list, 1/sigma, ...) are kept per (device, stream)
(csrc/host_api.cpp, StreamArena); before that they lived in one per-device buffer and a call on a second stream
overwrote the pointer table a kernel of the first stream was still reading.  The reference's wrappers are
single-stream and synchronous (lib/contract_wrappers.cu:93-114), so it has no counterpart of these tests.
"""
import ctypes
import threading

import numpy as np
import pytest
import torch

from util import orc, random_gauge_lex, random_spinor_lex, sigmas, momenta_p2_le, rel_err

pytestmark = pytest.mark.gpu


def _fields(hip, X, nev, seed, prec=8, order=2):
    rng = np.random.default_rng(seed)
    ev = [orc.lex_to_eo(random_spinor_lex(rng, X), X) for _ in range(nev)]
    return ev, [hip.SpinorField(X, prec, order).set_logical(v) for v in ev]


def test_two_contractions_on_two_streams_do_not_share_tables(hip):
    """Two eigenvector sets, two streams, launches interleaved without any synchronisation in between: each stream's
    result equals the oracle's (and, the kernel being deterministic, the single-stream result bit for bit)."""
    X = (16, 16, 16, 16)
    V = int(np.prod(X))
    nev = 24
    evA, fA = _fields(hip, X, nev, 11)
    evB, fB = _fields(hip, X, nev, 12)
    sgA, sgB = sigmas(nev), 0.5 + 0.01 * np.arange(nev)
    refA = np.zeros(16 * V, dtype=np.complex128)
    refB = np.zeros(16 * V, dtype=np.complex128)
    for n in range(nev):
        orc.loop_contract(refA, evA[n], evA[n], sgA[n])
        orc.loop_contract(refB, evB[n], evB[n], sgB[n])
    oneA = torch.zeros(16 * V, dtype=torch.complex128, device="cuda")
    oneB = torch.zeros(16 * V, dtype=torch.complex128, device="cuda")
    hip.performLoopContractionBatched(oneA, fA, fA, sgA)
    hip.performLoopContractionBatched(oneB, fB, fB, sgB)
    torch.cuda.synchronize()
    assert rel_err(oneA.cpu().numpy(), refA) < 1e-12 and rel_err(oneB.cpu().numpy(), refB) < 1e-12

    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    reps = 40
    outA = torch.zeros(reps, 16 * V, dtype=torch.complex128, device="cuda")
    outB = torch.zeros(reps, 16 * V, dtype=torch.complex128, device="cuda")
    torch.cuda.synchronize()
    for r in range(reps):                       # A on s1 and B on s2, back to back: tables of both calls are live at once
        with torch.cuda.stream(s1):
            hip.performLoopContractionBatched(outA[r], fA, fA, sgA)
        with torch.cuda.stream(s2):
            hip.performLoopContractionBatched(outB[r], fB, fB, sgB)
    torch.cuda.synchronize()
    for r in range(reps):
        assert torch.equal(outA[r], oneA), "stream 1, repetition %d differs from the single-stream result" % r
        assert torch.equal(outB[r], oneB), "stream 2, repetition %d differs from the single-stream result" % r
    lib = hip._lib.load()
    for s in (s1, s2):
        assert lib.mugiq_hip_release_stream(ctypes.c_void_p(s.cuda_stream)) == 0
    assert lib.mugiq_hip_release_stream(ctypes.c_void_p(s1.cuda_stream)) == 0     # nothing left: still success


def test_two_loops_on_two_streams_from_two_threads(hip):
    """Two Loop_Mugiq objects (different eigenvectors, entries and momenta), each created on its own stream, computed
    at the same time from two host threads; both equal the oracle."""
    X = (4, 6, 4, 8)
    V = int(np.prod(X))
    rng = np.random.default_rng(77)
    Uo = orc.extended_gauge_from_global(random_gauge_lex(rng, X), (0, 0, 0, 0), (1, 1, 1, 1), (0, 0, 0, 0))
    U = hip.GaugeField(X, (0, 0, 0, 0), 8).set_logical(Uo)
    jobs = []
    for seed, nev, entry, p2 in ((1, 5, "+z:1,2;-z:1,2;+x:1", 2), (2, 7, "-t:1,3;+y:2;-y:2", 3)):
        ev, f = _fields(hip, X, nev, seed)
        moms = momenta_p2_le(p2)
        jobs.append({"ev": ev, "f": f, "sg": sigmas(nev) * (1 + seed), "entry": entry, "moms": moms, "stream": torch.cuda.Stream()})
    torch.cuda.synchronize()
    for reps in range(3):
        loops = []
        for j in jobs:
            prm = hip.MugiqLoopParam(FTSign=1, doMomProj=True, gauge=U, momMatrix=[list(m) for m in j["moms"]], Nmom=len(j["moms"]))
            prm.set_displace_entry_string(j["entry"])
            with torch.cuda.stream(j["stream"]):
                loops.append(hip.Loop_Mugiq(prm, j["f"], j["sg"]))
        errs = []

        def run(lp):
            try:
                lp.computeCoarseLoop()              # ctypes releases the GIL: the two computes overlap
            except Exception as e:                      # surfaced below
                errs.append(e)

        th = [threading.Thread(target=run, args=(lp,)) for lp in loops]
        for t in th:
            t.start()
        for t in th:
            t.join()
        assert not errs, errs
        for j, lp in zip(jobs, loops):
            _, s, a, b = orc.parse_disp_entry_string(j["entry"])
            cprm = orc.LoopComputeParam(s, a, b)
            ref_pos = orc.compute_loop_position_space(j["ev"], j["sg"], cprm, Uo, X)
            assert rel_err(lp.dataPos_d.cpu().numpy(), ref_pos) < 1e-12
            locV3 = X[0] * X[1] * X[2]
            ref_mom = orc.momentum_projection_local(orc.convert_idx_order_map_gamma(ref_pos, cprm.nData, cprm.nLoop, 2, V // 2, X),
                                                    orc.phase_matrix(j["moms"], locV3, 1, X, X), X[3], cprm.nData, locV3, len(j["moms"]))
            assert rel_err(lp.dataMom_bcast, ref_mom) < 1e-12
            lp.close()
