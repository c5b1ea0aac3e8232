"""Device fields in QUDA-native layouts, as plain torch buffers + the POD descriptors of the C ABI.

These classes play the role of quda::ColorSpinorField / cudaGaugeField *as seen by the hot path*:
a device pointer plus layout and geometry.  torch is used for device memory only.
"""
import ctypes

import numpy as np
import torch

from . import _lib

FLOAT2 = 2
FLOAT4 = 4


def _cdtype(precision):
    return torch.complex128 if precision == 8 else torch.complex64


def _np_cdtype(precision):
    return np.complex128 if precision == 8 else np.complex64


def spinor_native_index(order, parity, x_cb, s, c, stride, parity_offset):
    """Complex-element index of component (s, c) at (parity, x_cb); see MugiqHipSpinorField in mugiq_hip.h."""
    k = 3 * s + c
    if order == FLOAT2:
        return parity * parity_offset + k * stride + x_cb
    return parity * parity_offset + ((k // 2) * stride + x_cb) * 2 + (k % 2)


class SpinorField:
    """nSpin=4, nColor=3 full-site-subset field in FLOAT2 or FLOAT4 order (plus optional depth-1 ghost zones)."""

    def __init__(self, X, precision=8, order=FLOAT2, pad=0, device="cuda", data=None):
        self.X = tuple(int(x) for x in X)
        assert all(x > 0 and x % 2 == 0 for x in self.X), "local dims must be even"
        self.precision = int(precision)
        self.order = int(order)
        self.volumeCB = int(np.prod(self.X)) // 2
        self.stride = self.volumeCB + int(pad)
        self.parity_offset = 12 * self.stride
        self.device = torch.device(device)
        n = 2 * self.parity_offset
        if data is None:
            self.data = torch.zeros(n, dtype=_cdtype(precision), device=self.device)
        else:
            assert data.numel() == n and data.dtype == _cdtype(precision)
            self.data = data
        self.ghost = [[None, None] for _ in range(4)]

    def face_cb(self, dim):
        return self.volumeCB // self.X[dim]

    def alloc_ghost(self, dim, bnd):
        if self.ghost[dim][bnd] is None:
            self.ghost[dim][bnd] = torch.zeros(2 * 12 * self.face_cb(dim), dtype=self.data.dtype, device=self.device)
        return self.ghost[dim][bnd]

    def desc(self):
        d = _lib.SpinorDesc()
        d.data = self.data.data_ptr()
        d.precision = self.precision
        d.field_order = self.order
        d.nParity = 2
        d.volumeCB = self.volumeCB
        d.stride = self.stride
        for i in range(4):
            d.X[i] = self.X[i]
            for b in range(2):
                g = self.ghost[i][b]
                d.ghost[i][b] = g.data_ptr() if g is not None else None
        d.parity_offset = self.parity_offset
        return d

    # ---- host <-> device plumbing in the logical shape [2, volumeCB, 4, 3] -------------------------------
    def _index_table(self, vcb=None, stride=None, parity_offset=None):
        vcb = self.volumeCB if vcb is None else vcb
        stride = self.stride if stride is None else stride
        parity_offset = self.parity_offset if parity_offset is None else parity_offset
        p = np.arange(2).reshape(2, 1, 1, 1)
        x = np.arange(vcb).reshape(1, vcb, 1, 1)
        s = np.arange(4).reshape(1, 1, 4, 1)
        c = np.arange(3).reshape(1, 1, 1, 3)
        return spinor_native_index(self.order, p, x, s, c, stride, parity_offset)

    def set_logical(self, v):
        v = np.asarray(v)
        assert v.shape == (2, self.volumeCB, 4, 3)
        buf = np.zeros(2 * self.parity_offset, dtype=_np_cdtype(self.precision))
        buf[self._index_table()] = v.astype(buf.dtype)
        self.data.copy_(torch.from_numpy(buf))
        return self

    def get_logical(self):
        buf = self.data.cpu().numpy()
        return buf[self._index_table()]

    def set_ghost_logical(self, dim, bnd, zone):
        fcb = self.face_cb(dim)
        zone = np.asarray(zone)
        assert zone.shape == (2, fcb, 4, 3)
        buf = np.zeros(2 * 12 * fcb, dtype=_np_cdtype(self.precision))
        buf[self._index_table(fcb, fcb, 12 * fcb)] = zone.astype(buf.dtype)
        self.alloc_ghost(dim, bnd).copy_(torch.from_numpy(buf))

    def zone_to_logical(self, dim, zone_tensor):
        fcb = self.face_cb(dim)
        return zone_tensor.cpu().numpy()[self._index_table(fcb, fcb, 12 * fcb)]


class GaugeField:
    """Border-extended gauge field in native FLOAT2 order, 18 reals per link (what Displace builds at
    lib/displace.cpp:104-134 of the reference)."""

    def __init__(self, X, R=(0, 0, 0, 0), precision=8, pad=0, device="cuda"):
        self.X = tuple(int(x) for x in X)
        self.R = tuple(int(r) for r in R)
        self.XE = tuple(self.X[d] + 2 * self.R[d] for d in range(4))
        self.precision = int(precision)
        self.volumeExCB = int(np.prod(self.XE)) // 2
        self.stride = self.volumeExCB + int(pad)
        self.parity_offset = 36 * self.stride
        self.device = torch.device(device)
        self.data = torch.zeros(2 * self.parity_offset, dtype=_cdtype(precision), device=self.device)

    def desc(self):
        d = _lib.GaugeDesc()
        d.data = self.data.data_ptr()
        d.precision = self.precision
        for i in range(4):
            d.X[i] = self.X[i]
            d.R[i] = self.R[i]
        d.stride = self.stride
        d.parity_offset = self.parity_offset
        return d

    def set_from_qdp_host(self, qdp_links, comm=None):
        """Displace::createExtendedCudaGaugeField (lib/displace.cpp:70-134): `qdp_links` = 4 host arrays of the LOCAL
        lattice in QDP order (loopParams.gauge[4]); borders come from the neighbours through `comm` (a GridComm)."""
        import ctypes
        import torch
        arrs = [np.ascontiguousarray(a) for a in qdp_links]
        cpu_prec = 8 if arrs[0].dtype == np.float64 else 4
        ptrs = (ctypes.c_void_p * 4)(*[a.ctypes.data for a in arrs])
        d = self.desc()
        c = comm.c_struct() if comm is not None else None
        _lib.check(_lib.load().mugiq_hip_create_extended_gauge(
            ctypes.byref(d), ptrs, cpu_prec, ctypes.byref(c) if c is not None else None,
            ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)))
        return self

    def get_logical(self):
        d = np.arange(4).reshape(4, 1, 1, 1, 1)
        p = np.arange(2).reshape(1, 2, 1, 1, 1)
        x = np.arange(self.volumeExCB).reshape(1, 1, -1, 1, 1)
        r = np.arange(3).reshape(1, 1, 1, 3, 1)
        c = np.arange(3).reshape(1, 1, 1, 1, 3)
        idx = p * self.parity_offset + (d * 9 + r * 3 + c) * self.stride + x
        return self.data.cpu().numpy()[idx]

    def set_logical(self, U):
        """U: [4, 2, volExCB, 3, 3] (dir, parity, extended even-odd index, row, col)."""
        U = np.asarray(U)
        assert U.shape == (4, 2, self.volumeExCB, 3, 3)
        d = np.arange(4).reshape(4, 1, 1, 1, 1)
        p = np.arange(2).reshape(1, 2, 1, 1, 1)
        x = np.arange(self.volumeExCB).reshape(1, 1, -1, 1, 1)
        r = np.arange(3).reshape(1, 1, 1, 3, 1)
        c = np.arange(3).reshape(1, 1, 1, 1, 3)
        idx = p * self.parity_offset + (d * 9 + r * 3 + c) * self.stride + x
        buf = np.zeros(2 * self.parity_offset, dtype=_np_cdtype(self.precision))
        buf[idx] = U.astype(buf.dtype)
        self.data.copy_(torch.from_numpy(buf))
        return self


class CoarseField:
    """Coarse-grid colour-spinor (nSpin 2, nColor n_vec) in FLOAT2 order: a coarse eigenvector of the MG path."""

    def __init__(self, Xc, n_vec, precision=8, pad=0, device="cuda"):
        self.X = tuple(int(x) for x in Xc)
        assert all(x > 0 and x % 2 == 0 for x in self.X), "coarse dims must be even"
        self.n_vec = int(n_vec)
        self.precision = int(precision)
        self.volumeCB = int(np.prod(self.X)) // 2
        self.stride = self.volumeCB + int(pad)
        self.parity_offset = 2 * self.n_vec * self.stride
        self.device = torch.device(device)
        self.data = torch.zeros(2 * self.parity_offset, dtype=_cdtype(precision), device=self.device)

    def desc(self):
        d = _lib.CoarseDesc()
        d.data = self.data.data_ptr()
        d.precision, d.nSpin, d.nColor = self.precision, 2, self.n_vec
        d.volumeCB, d.stride, d.parity_offset = self.volumeCB, self.stride, self.parity_offset
        for i in range(4):
            d.X[i] = self.X[i]
        return d

    def set_logical(self, phi):
        """phi: [2, volCB_c, 2, n_vec]"""
        phi = np.asarray(phi)
        assert phi.shape == (2, self.volumeCB, 2, self.n_vec)
        p = np.arange(2).reshape(2, 1, 1, 1)
        x = np.arange(self.volumeCB).reshape(1, -1, 1, 1)
        s = np.arange(2).reshape(1, 1, 2, 1)
        c = np.arange(self.n_vec).reshape(1, 1, 1, -1)
        buf = np.zeros(2 * self.parity_offset, dtype=_np_cdtype(self.precision))
        buf[p * self.parity_offset + (s * self.n_vec + c) * self.stride + x] = phi.astype(buf.dtype)
        self.data.copy_(torch.from_numpy(buf))
        return self

    def get_logical(self):
        """[2, volCB_c, 2, n_vec]"""
        p = np.arange(2).reshape(2, 1, 1, 1)
        x = np.arange(self.volumeCB).reshape(1, -1, 1, 1)
        s = np.arange(2).reshape(1, 1, 2, 1)
        c = np.arange(self.n_vec).reshape(1, 1, 1, -1)
        return self.data.cpu().numpy()[p * self.parity_offset + (s * self.n_vec + c) * self.stride + x]


class Transfer:
    """One level of QUDA's Transfer as the hot path sees it: the block-orthonormal null vectors V on the finer grid of
    the level (packed vector index), the aggregate size and the spin blocking.  Finest level: the finer side is the
    fine lattice (4 spins x 3 colours, spin_block_size 2).  A coarse -> coarse level: fine_spin = 2, fine_color = n_vec of
    the next finer level, spin_block_size = 1."""

    def __init__(self, X, n_vec=24, geo_block_size=(4, 4, 4, 4), spin_block_size=2, precision=8, pad=0, device="cuda",
                 fine_spin=4, fine_color=3):
        self.X = tuple(int(x) for x in X)
        self.n_vec = int(n_vec)
        self.geo_block_size = tuple(int(b) for b in geo_block_size)
        self.spin_block_size = int(spin_block_size)
        self.fine_spin, self.fine_color = int(fine_spin), int(fine_color)
        self.precision = int(precision)
        self.volumeCB = int(np.prod(self.X)) // 2
        self.stride = self.volumeCB + int(pad)
        self.parity_offset = self.fine_spin * self.fine_color * self.n_vec * self.stride
        self.device = torch.device(device)
        self.V = torch.zeros(2 * self.parity_offset, dtype=_cdtype(precision), device=self.device)
        self.Xc = tuple(self.X[d] // self.geo_block_size[d] for d in range(4))

    def desc(self):
        d = _lib.TransferDesc()
        d.V = self.V.data_ptr()
        d.precision, d.nVec, d.spinBlockSize = self.precision, self.n_vec, self.spin_block_size
        d.stride, d.parity_offset = self.stride, self.parity_offset
        for i in range(4):
            d.X[i] = self.X[i]
            d.geoBlockSize[i] = self.geo_block_size[i]
        return d

    def set_logical(self, V):
        """V: [2, volCB, fine_spin, fine_color, n_vec]"""
        V = np.asarray(V)
        ns, nc = self.fine_spin, self.fine_color
        assert V.shape == (2, self.volumeCB, ns, nc, self.n_vec)
        p = np.arange(2).reshape(2, 1, 1, 1, 1)
        x = np.arange(self.volumeCB).reshape(1, -1, 1, 1, 1)
        s = np.arange(ns).reshape(1, 1, ns, 1, 1)
        c = np.arange(nc).reshape(1, 1, 1, nc, 1)
        j = np.arange(self.n_vec).reshape(1, 1, 1, 1, -1)
        buf = np.zeros(2 * self.parity_offset, dtype=_np_cdtype(self.precision))
        buf[p * self.parity_offset + ((nc * s + c) * self.n_vec + j) * self.stride + x] = V.astype(buf.dtype)
        self.V.copy_(torch.from_numpy(buf))
        return self


def transfer_desc_array(transfers):
    arr = (_lib.TransferDesc * len(transfers))()
    for i, t in enumerate(transfers):
        arr[i] = t.desc()
    return arr


def coarse_desc_array(fields):
    arr = (_lib.CoarseDesc * len(fields))()
    for i, f in enumerate(fields):
        arr[i] = f.desc()
    return arr


def desc_array(fields):
    arr = (_lib.SpinorDesc * len(fields))()
    for i, f in enumerate(fields):
        arr[i] = f.desc()
    return arr
