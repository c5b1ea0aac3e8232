"""Shared helpers for the test-suite: seeded synthetic inputs and oracle import."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import mugiq_oracle as orc  # noqa: E402  (tests are allowed to import the oracle)


def random_su3(rng, shape):
    """Random SU(3): complex Gaussian 3x3 -> Gram-Schmidt on rows -> det phase fixed on the last row."""
    a = rng.standard_normal(shape + (3, 3)) + 1j * rng.standard_normal(shape + (3, 3))
    r0 = a[..., 0, :]
    r0 = r0 / np.linalg.norm(r0, axis=-1, keepdims=True)
    r1 = a[..., 1, :]
    r1 = r1 - np.sum(np.conj(r0) * r1, axis=-1, keepdims=True) * r0
    r1 = r1 / np.linalg.norm(r1, axis=-1, keepdims=True)
    r2 = a[..., 2, :]
    r2 = r2 - np.sum(np.conj(r0) * r2, axis=-1, keepdims=True) * r0
    r2 = r2 - np.sum(np.conj(r1) * r2, axis=-1, keepdims=True) * r1
    r2 = r2 / np.linalg.norm(r2, axis=-1, keepdims=True)
    u = np.stack([r0, r1, r2], axis=-2)
    det = np.linalg.det(u)
    u[..., 2, :] = u[..., 2, :] / det[..., None]
    return u


def random_gauge_lex(rng, G):
    """Global gauge field [4, T, Z, Y, X, 3, 3], G = (X, Y, Z, T)."""
    return random_su3(rng, (4, G[3], G[2], G[1], G[0]))


def unit_gauge_lex(G):
    u = np.zeros((4, G[3], G[2], G[1], G[0], 3, 3), dtype=np.complex128)
    u[..., 0, 0] = u[..., 1, 1] = u[..., 2, 2] = 1.0
    return u


def random_spinor_lex(rng, G, normalise=True):
    """Global spinor [T, Z, Y, X, 4, 3], unit norm."""
    v = rng.standard_normal((G[3], G[2], G[1], G[0], 4, 3)) + 1j * rng.standard_normal((G[3], G[2], G[1], G[0], 4, 3))
    if normalise:
        v /= np.linalg.norm(v)
    return v


def sigmas(nev):
    return 0.01 + 0.002 * np.arange(nev)


def momenta_p2_le(n):
    """All integer momenta with p^2 <= n in lexicographic order."""
    r = int(np.floor(np.sqrt(n)))
    out = []
    for px in range(-r, r + 1):
        for py in range(-r, r + 1):
            for pz in range(-r, r + 1):
                if px * px + py * py + pz * pz <= n:
                    out.append((px, py, pz))
    return out


def gauge_eo_single_domain(U_lex, G):
    """Logical [4, 2, volCB, 3, 3] for a single periodic domain (brd = 0)."""
    return orc.extended_gauge_from_global(U_lex, (0, 0, 0, 0), (1, 1, 1, 1), (0, 0, 0, 0))


def rel_err(a, b):
    a = np.asarray(a)
    b = np.asarray(b)
    d = np.max(np.abs(a - b))
    s = np.max(np.abs(b))
    return d / s if s > 0 else d
