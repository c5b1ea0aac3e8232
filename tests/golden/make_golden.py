#!/usr/bin/env python3
"""Regenerates tests/golden/oracle_regression_4x4x4x4.npz.

NOT a reference-pinned golden vector: the reference (ckallidonis/mugiq) ships none and cannot be run here
(SURVEY.md section 8c), so parity stays "unpinned".  This fixture is produced by OUR oracle (oracle/mugiq_oracle.py) at a
known-good commit and guards against accidental drift of the oracle or of the HIP path: inputs + expected outputs
of the whole pipeline (ultra-local + displaced loops, reorder, phases, momentum projection) on a 4^4 lattice.

usage: python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
from util import orc, random_gauge_lex, random_spinor_lex, sigmas, momenta_p2_le  # noqa: E402

X = (4, 4, 4, 4)
ENTRIES = "+x:1;-z:1,2;+t:2"
rng = np.random.default_rng(20261004)
nev = 3
ev = np.stack([orc.lex_to_eo(random_spinor_lex(rng, X), X) for _ in range(nev)])
U = orc.extended_gauge_from_global(random_gauge_lex(rng, X), (0, 0, 0, 0), (1, 1, 1, 1), (0, 0, 0, 0))
sg = sigmas(nev)
moms = np.array(momenta_p2_le(1), dtype=np.int32)
_, s, a, b = orc.parse_disp_entry_string(ENTRIES)
cprm = orc.LoopComputeParam(s, a, b)
pos = orc.compute_loop_position_space(list(ev), sg, cprm, U, X)
V = int(np.prod(X))
locV3 = X[0] * X[1] * X[2]
mom = orc.momentum_projection_local(orc.convert_idx_order_map_gamma(pos, cprm.nData, cprm.nLoop, 2, V // 2, X),
                                    orc.phase_matrix(moms, locV3, -1, X, X), X[3], cprm.nData, locV3, len(moms))
np.savez_compressed(os.path.join(HERE, "oracle_regression_4x4x4x4.npz"), X=np.array(X), entries=np.array(ENTRIES), ev=ev, U=U,
                    sigma=sg, moms=moms, FTSign=np.array(-1), dataPos=pos, dataMom=mom)
print("wrote", os.path.join(HERE, "oracle_regression_4x4x4x4.npz"), "nLoop", cprm.nLoop, "dataPos", pos.shape, "dataMom", mom.shape)
