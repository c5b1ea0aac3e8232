"""mugiq_hip_reflect_momentum_space (host-only C-ABI entry, csrc/reflect_mom.cpp) against the oracle: the momentum-space data
of a "-mu" entry derived from the "+mu" entry's (and the other way round) equals the Fourier transform of the directly
computed slots, for every direction, both FT signs, both precisions, and with the gathered array cut into time-rank slabs
(lib/loop_mugiq.cpp:415-424) so that a shift along t crosses slab boundaries.  No GPU."""
import numpy as np
import pytest

from util import orc, random_gauge_lex, random_spinor_lex, gauge_eo_single_domain, sigmas, momenta_p2_le, rel_err


def _slabs(mom_g, nt):
    """[Nmom][nLoop][16][totT] -> dataMom_bcast: nt slabs of t + locT*ig + locT*16*iL + locT*16*nLoop*im"""
    Nmom, nLoop, _, T = mom_g.shape
    locT = T // nt
    return np.ascontiguousarray(mom_g.reshape(Nmom, nLoop, 16, nt, locT).transpose(3, 0, 1, 2, 4)).reshape(-1)


def _unslab(flat, Nmom, nLoop, T, nt):
    locT = T // nt
    return flat.reshape(nt, Nmom, nLoop, 16, locT).transpose(1, 2, 3, 0, 4).reshape(Nmom, nLoop, 16, T)


@pytest.fixture(scope="module")
def problem():
    X = (4, 6, 4, 8)
    rng = np.random.default_rng(314)
    ev = [orc.lex_to_eo(random_spinor_lex(rng, X), X) for _ in range(2)]
    U = gauge_eo_single_domain(random_gauge_lex(rng, X), X)
    return X, ev, U


@pytest.mark.parametrize("FTSign,dtype,nt", [(-1, np.complex128, 1), (1, np.complex128, 4), (-1, np.complex64, 2)])
def test_reflected_slots_in_momentum_space(hip, problem, FTSign, dtype, nt):
    X, ev, U = problem
    V = int(np.prod(X))
    moms = momenta_p2_le(3)
    locV3 = X[0] * X[1] * X[2]
    ph = orc.phase_matrix(moms, locV3, FTSign, X, X)
    tol = 1e-13 if dtype == np.complex128 else 2e-6
    for d, name in enumerate("xyzt"):
        cprm = orc.LoopComputeParam(["+" + name, "-" + name], [1, 1], [3, 3])          # slots: 0 | +: 1..3 | -: 4..6
        pos = orc.compute_loop_position_space(ev, sigmas(2), cprm, U, X)
        mp = orc.convert_idx_order_map_gamma(pos, cprm.nData, cprm.nLoop, 2, V // 2, X)
        full = orc.momentum_projection_local(mp, ph, X[3], cprm.nData, locV3, len(moms)).reshape(len(moms), cprm.nLoop, 16, X[3])
        for dst_plus in (False, True):
            work = full.copy()
            dst0, src0 = (1, 4) if dst_plus else (4, 1)
            work[:, dst0:dst0 + 3] = np.nan                                                   # the derived slots hold nothing yet
            flat = _slabs(work, nt).astype(dtype)
            for k in (1, 2, 3):
                hip.reflectMomentumSpace(flat, moms, FTSign, X, cprm.nLoop, X[3] // nt, X[3], dst0 + k - 1, src0 + k - 1, d, int(dst_plus), k)
            got = _unslab(flat, len(moms), cprm.nLoop, X[3], nt)
            assert rel_err(got, full.astype(dtype)) < tol, (name, "plus" if dst_plus else "minus")


def test_momentum_list_without_negatives_is_refused(hip):
    moms = [(0, 0, 0), (1, 0, 0)]
    flat = np.zeros(16 * len(moms) * 4 * 3, dtype=np.complex128)
    with pytest.raises(hip.MugiqHipError, match="not closed under p -> -p"):
        hip.reflectMomentumSpace(flat, moms, 1, (4, 4, 4, 4), 3, 4, 4, 2, 1, 0, 0, 1)
