"""CPU oracle for the MuGiq disconnected-loop hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference algorithm (ckallidonis/mugiq, mounted at
/root/reference while building).  It is the *checker* for the HIP path: only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.  Nothing under
`mugiq_amd/` (the product) imports, links or executes anything from `oracle/`.

PARITY UNPINNED.  The reference ships no unit tests, golden vectors or fixtures (SURVEY.md §4,
§8c) and cannot be built here (needs nvcc + QUDA + parallel HDF5).  The arithmetic it delegates
to QUDA (field accessors, even-odd index helpers, ghost-face index) is not vendored and not
version-pinned by the reference (CMakeLists.txt:112-114 only takes a path), so those conventions
are restated from upstream QUDA (include/index_helper.cuh, color_spinor_field_order.h,
gauge_field_order.h; circa v1.0 / early-2020 develop) and marked ASSUMED below.  The oracle is
anchored instead by analytic known-answer tests (tests/test_oracle_kat.py): gamma-algebra
identities, D_-mu D_+mu = 1, gauge covariance, periodic wrap, direct DFT, multi-domain ==
single-domain.  The literal tables of the reference's headers (gamma row values / column
indices / names, gamma5 map, displacement flags) ARE pinned against the header text where the
reference is mounted (tests/test_reference_tables.py).

Every function cites the reference file:line it follows (paths relative to /root/reference).

Logical array conventions used throughout the oracle
  spinor   v[parity, x_cb, spin, colour]          complex   (QUDA accessor F(pty, x_cb, s, c))
  gauge    U[dir, parity, x_cb_ext, row, col]     complex   (QUDA accessor U(dir, x_cb, pty)(row,col))
  loop     loop[idata * V + tid], tid = x_cb + parity * volumeCB, idata = iG + 16 * iLoop
Native memory layouts (what the C-ABI sees) are produced by the `*_to_native` helpers.
"""
import numpy as np

N_SPIN = 4
N_COLOR = 3
N_GAMMA = 16
N_DIM = 4

# ----------------------------------------------------------------------------------------------
# gamma tables                                                   include/gamma.h:11-20, 32-71, 99-109
# ----------------------------------------------------------------------------------------------
GAMMA_NAMES = ["1", "g1", "g2", "g1g2", "g3", "g1g3", "g2g3", "g5g4",
               "g4", "g1g4", "g2g4", "g5g3", "g3g4", "g5g2", "g5g1", "g5"]

# (re, im) of the non-zero element in rows 0..3 of G(n), DeGrand-Rossi basis  include/gamma.h:32-50
_ROW_VALUE = [
    [(1, 0), (1, 0), (1, 0), (1, 0)],
    [(0, 1), (0, 1), (0, -1), (0, -1)],
    [(-1, 0), (1, 0), (1, 0), (-1, 0)],
    [(0, -1), (0, 1), (0, -1), (0, 1)],
    [(0, 1), (0, -1), (0, -1), (0, 1)],
    [(-1, 0), (1, 0), (-1, 0), (1, 0)],
    [(0, -1), (0, -1), (0, -1), (0, -1)],
    [(1, 0), (1, 0), (-1, 0), (-1, 0)],
    [(1, 0), (1, 0), (1, 0), (1, 0)],
    [(0, 1), (0, 1), (0, -1), (0, -1)],
    [(-1, 0), (1, 0), (1, 0), (-1, 0)],
    [(0, -1), (0, 1), (0, -1), (0, 1)],
    [(0, 1), (0, -1), (0, -1), (0, 1)],
    [(-1, 0), (1, 0), (-1, 0), (1, 0)],
    [(0, -1), (0, -1), (0, -1), (0, -1)],
    [(1, 0), (1, 0), (-1, 0), (-1, 0)],
]
# column holding that element                                                 include/gamma.h:54-71
_COLUMN_INDEX = [
    [0, 1, 2, 3], [3, 2, 1, 0], [3, 2, 1, 0], [0, 1, 2, 3],
    [2, 3, 0, 1], [1, 0, 3, 2], [1, 0, 3, 2], [2, 3, 0, 1],
    [2, 3, 0, 1], [1, 0, 3, 2], [1, 0, 3, 2], [2, 3, 0, 1],
    [0, 1, 2, 3], [3, 2, 1, 0], [3, 2, 1, 0], [0, 1, 2, 3],
]
GAMMA_ROW_VALUE = np.array([[complex(re, im) for (re, im) in row] for row in _ROW_VALUE])
GAMMA_COLUMN_INDEX = np.array(_COLUMN_INDEX, dtype=np.int64)

MINUS_GAMMA = [3, 6, 9, 11, 12, 14]                       # include/gamma.h:99-102
INDEX_MAP_GAMMA = [N_GAMMA - i - 1 for i in range(N_GAMMA)]  # include/gamma.h:105-109


def gamma_map_sign():
    """sign[ig] of the G -> g5*G map (lib/contract_wrappers.cu:26-43)."""
    s = np.ones(N_GAMMA)
    s[MINUS_GAMMA] = -1.0
    return s


def gamma_dense(n):
    """4x4 dense G(n): G(n)_{ij} = RowValue[n][i] * delta(j, ColumnIndex[n][i])  (include/gamma.h:22-29)."""
    g = np.zeros((4, 4), dtype=np.complex128)
    for i in range(4):
        g[i, GAMMA_COLUMN_INDEX[n, i]] = GAMMA_ROW_VALUE[n, i]
    return g


# ----------------------------------------------------------------------------------------------
# QUDA index helpers  (ASSUMED: upstream QUDA include/index_helper.cuh)
# called at lib/mugiq_displace_kernels.cu:17,22,28,59,128,133,141,146,168 and lib/mugiq_util_kernels.cu:75
# ----------------------------------------------------------------------------------------------
def get_coords(x_cb, X, parity):
    """getCoords(x, cb_index, X, parity): even-odd index -> 4-d coordinates. Returns int array (...,4)."""
    x_cb = np.asarray(x_cb, dtype=np.int64)
    za = x_cb // (X[0] // 2)
    zb = za // X[1]
    x1 = za - zb * X[1]
    x3 = zb // X[2]
    x2 = zb - x3 * X[2]
    x1odd = (x1 + x2 + x3 + parity) & 1
    x0 = 2 * x_cb + x1odd - za * X[0]
    return np.stack([x0, x1, x2, x3], axis=-1)


def lex_index(x, X):
    return ((x[..., 3] * X[2] + x[..., 2]) * X[1] + x[..., 1]) * X[0] + x[..., 0]


def link_index(x, X):
    """linkIndex(x, X) = lexicographic >> 1."""
    return lex_index(x, X) >> 1


def link_index_shift(x, dx, X):
    """linkIndexShift(x, dx, X): y[i] = (x[i] + dx[i] + X[i]) % X[i]; lexicographic >> 1."""
    y = (x + np.asarray(dx) + np.asarray(X)) % np.asarray(X)
    return lex_index(y, X) >> 1


def link_index_p1(x, X, mu):
    dx = [0, 0, 0, 0]
    dx[mu] = 1
    return link_index_shift(x, dx, X)


def link_index_m1(x, X, mu):
    dx = [0, 0, 0, 0]
    dx[mu] = -1
    return link_index_shift(x, dx, X)


def ghost_face_index(bnd, x, X, dim, nFace):
    """ghostFaceIndex<bnd>(x, X, dim, nFace), 4-d: drop coordinate `dim`; leading index is x[dim]
    (bnd=0, backward face) or x[dim]-X[dim]+nFace (bnd=1, forward face); remaining coordinates
    lexicographic with the lowest dimension fastest; >> 1."""
    lead = x[..., dim] if bnd == 0 else x[..., dim] - X[dim] + nFace
    rest = [d for d in range(4) if d != dim]            # ascending; rest[0] fastest
    a, b, c = rest
    idx = ((lead * X[c] + x[..., c]) * X[b] + x[..., b]) * X[a] + x[..., a]
    return idx >> 1


# ----------------------------------------------------------------------------------------------
# native memory layouts (ASSUMED: upstream QUDA color_spinor_field_order.h / gauge_field_order.h)
# accessor typedefs at include/contract_util.cuh:20-24
# ----------------------------------------------------------------------------------------------
FLOAT2 = 2
FLOAT4 = 4


def spinor_native_index(order, parity, x_cb, s, c, stride, parity_offset):
    """Complex-element index of F(parity, x_cb, s, c) for FieldOrderCB<Float,4,3,1,order>."""
    k = s * N_COLOR + c
    if order == FLOAT2:
        return parity * parity_offset + k * stride + x_cb
    if order == FLOAT4:
        return parity * parity_offset + ((k // 2) * stride + x_cb) * 2 + (k % 2)
    raise ValueError("field order must be 2 or 4")


def spinor_to_native(v, order, stride=None, parity_offset=None):
    """Logical [2, volumeCB, 4, 3] -> flat native buffer of complex elements."""
    npar, vcb = v.shape[0], v.shape[1]
    stride = vcb if stride is None else stride
    parity_offset = 12 * stride if parity_offset is None else parity_offset
    buf = np.zeros(npar * parity_offset, dtype=v.dtype)
    x = np.arange(vcb)
    for p in range(npar):
        for s in range(4):
            for c in range(3):
                buf[spinor_native_index(order, p, x, s, c, stride, parity_offset)] = v[p, :, s, c]
    return buf


def spinor_from_native(buf, order, vcb, stride=None, parity_offset=None, npar=2):
    stride = vcb if stride is None else stride
    parity_offset = 12 * stride if parity_offset is None else parity_offset
    v = np.zeros((npar, vcb, 4, 3), dtype=buf.dtype)
    x = np.arange(vcb)
    for p in range(npar):
        for s in range(4):
            for c in range(3):
                v[p, :, s, c] = buf[spinor_native_index(order, p, x, s, c, stride, parity_offset)]
    return v


def gauge_native_index(dirn, parity, x_cb, row, col, stride, parity_offset):
    """Complex-element index for gauge_mapper<Float,QUDA_RECONSTRUCT_NO> (FLOAT2 order, 18 reals)."""
    return parity * parity_offset + (dirn * 9 + row * 3 + col) * stride + x_cb


def gauge_to_native(U, stride=None, parity_offset=None):
    """Logical [4, 2, volCB, 3, 3] -> flat native buffer."""
    vcb = U.shape[2]
    stride = vcb if stride is None else stride
    parity_offset = 36 * stride if parity_offset is None else parity_offset
    buf = np.zeros(2 * parity_offset, dtype=U.dtype)
    x = np.arange(vcb)
    for d in range(4):
        for p in range(2):
            for r in range(3):
                for c in range(3):
                    buf[gauge_native_index(d, p, x, r, c, stride, parity_offset)] = U[d, p, :, r, c]
    return buf


def gauge_to_qdp_host(U):
    """Logical [4, 2, volCB, 3, 3] -> list of 4 real arrays in QDP host order
    gauge[dir][(parity*V/2 + x_cb)*18 + (row*3+col)*2 + reim]  (tests/loop.cpp:88,106; lib/displace.cpp:82)."""
    out = []
    for d in range(4):
        a = np.ascontiguousarray(U[d].reshape(-1, 9))          # [(parity, x_cb), row*3+col]
        out.append(a.view(np.float64 if U.dtype == np.complex128 else np.float32).reshape(-1).copy())
    return out


# ----------------------------------------------------------------------------------------------
# lexicographic <-> even-odd helpers (test-input plumbing; QUDA even-odd convention, Appendix A)
# ----------------------------------------------------------------------------------------------
def eo_site_tables(X):
    """For local dims X: (parity[V], x_cb[V]) of every lexicographic site, and the inverse lex[2, volCB]."""
    V = int(np.prod(X))
    i = np.arange(V)
    x0 = i % X[0]
    x1 = (i // X[0]) % X[1]
    x2 = (i // (X[0] * X[1])) % X[2]
    x3 = i // (X[0] * X[1] * X[2])
    par = (x0 + x1 + x2 + x3) & 1
    xcb = i >> 1
    inv = np.zeros((2, V // 2), dtype=np.int64)
    inv[par, xcb] = i
    return par, xcb, inv


def lex_to_eo(f_lex, X):
    """f_lex[T, Z, Y, X, ...] -> f_eo[2, volCB, ...]."""
    V = int(np.prod(X))
    flat = f_lex.reshape((V,) + f_lex.shape[4:])
    _, _, inv = eo_site_tables(X)
    return flat[inv]


def eo_to_lex(f_eo, X):
    V = int(np.prod(X))
    _, _, inv = eo_site_tables(X)
    flat = np.zeros((V,) + f_eo.shape[2:], dtype=f_eo.dtype)
    flat[inv] = f_eo
    return flat.reshape((X[3], X[2], X[1], X[0]) + f_eo.shape[2:])


# ----------------------------------------------------------------------------------------------
# a1/a2  loop contraction         lib/mugiq_contract_kernels.cu:45-122, lib/contract_wrappers.cu:88-115
# ----------------------------------------------------------------------------------------------
def loop_contract(loop, vL, vR, sigma, dtype=np.float64):
    """loopData[tid + V*iG] += inv_sigma * sum_{s2} row_value[iG][s2] * resG[s2, column_index[iG][s2]],
    resG[be, al] = sum_{kc} conj(vL[be,kc]) * vR[al,kc]        (lib/mugiq_contract_kernels.cu:98-120)
    inv_sigma = Float(1.0 / sigma)                              (include/contract_util.cuh:130-134)
    `loop` is the flat complex view of one loop slot (16*V elements), accumulated in place.
    The debug printf at :90-95 is deliberately not reproduced."""
    cdt = np.complex128 if dtype == np.float64 else np.complex64
    npar, vcb = vL.shape[0], vL.shape[1]
    V = npar * vcb
    L = vL.reshape(V, 4, 3).astype(cdt, copy=False)
    R = vR.reshape(V, 4, 3).astype(cdt, copy=False)
    inv_sigma = dtype(1.0 / float(dtype(sigma)))
    # colour trace, kc = 0,1,2 in order, starting from 0            (:103-105)
    resG = np.zeros((V, 4, 4), dtype=cdt)                           # [x, be, al]
    for kc in range(3):
        resG += np.conj(L[:, :, None, kc]) * R[:, None, :, kc]
    for iG in range(N_GAMMA):                                       # (:110-117)
        trace = np.zeros(V, dtype=cdt)
        for s2 in range(4):
            s1 = GAMMA_COLUMN_INDEX[iG, s2]
            trace += cdt(GAMMA_ROW_VALUE[iG, s2]) * resG[:, s2, s1]
        loop[V * iG:V * (iG + 1)] += inv_sigma * trace              # (:120)
    return loop


# ----------------------------------------------------------------------------------------------
# a4/a5  covariant displacement   lib/mugiq_displace_kernels.cu:4-185, lib/contract_wrappers.cu:166-198
# ----------------------------------------------------------------------------------------------
DISP_SIGN_MINUS = 0     # include/enum_mugiq.h:80-84
DISP_SIGN_PLUS = 1
DISPLACE_FLAGS = ["+x", "-x", "+y", "-y", "+z", "-z", "+t", "-t"]   # include/displace.h:21


def parse_displacement(dstr):
    """Displace::setupDisplacement: string -> flag -> (dir, sign)   (lib/displace.cpp:137-223,
    include/enum_mugiq.h:59-85): flag = index in DISPLACE_FLAGS, dir = flag/2, sign = + for even flag."""
    if dstr not in DISPLACE_FLAGS:
        raise ValueError("Cannot parse given displacement string = %s" % dstr)
    flag = DISPLACE_FLAGS.index(dstr)
    return flag // 2, (DISP_SIGN_PLUS if flag % 2 == 0 else DISP_SIGN_MINUS)


def covariant_displacement(src, U, dirn, sign, dim, comm_dim=(0, 0, 0, 0), brd=(0, 0, 0, 0),
                           ghost=None, nFace=1):
    """dst(x) = U_d(x) * src(x+d)          (sign +)
       dst(x) = U_d^dag(x-d) * src(x-d)    (sign -)       lib/mugiq_displace_kernels.cu:156-185
    src    [2, volCB, 4, 3]; U [4, 2, volExCB, 3, 3] on the border-extended lattice dimEx = dim + 2*brd
    ghost  ghost[dir][bnd] = [2, faceCB, 4, 3] (bnd 0 = backward zone, 1 = forward zone), used only when
           comm_dim[dir] and the site is on the face (getNbrSiteVec, :116-151).
    The live link branch is getNbrLinkExtG (:68-74, 39-66); getNbrLink (:8-34) is dead code in the
    reference because Displace always builds an EXTENDED field (lib/displace.cpp:113-114)."""
    dim = list(dim)
    brd = list(brd)
    dimEx = [dim[i] + 2 * brd[i] for i in range(4)]
    vcb = src.shape[1]
    dst = np.zeros_like(src)
    for pty in range(2):
        x_cb = np.arange(vcb)
        coord = get_coords(x_cb, dim, pty)                              # :167-169
        nbr_pty = 1 - pty                                               # :120
        # ---- neighbouring vector, getNbrSiteVec :116-151
        if sign == DISP_SIGN_PLUS:
            idx = link_index_p1(coord, dim, dirn)
            nbrV = src[nbr_pty, idx].copy()
            if comm_dim[dirn]:
                on = coord[:, dirn] + nFace >= dim[dirn]
                if on.any():
                    g = ghost_face_index(1, coord[on], dim, dirn, nFace)
                    nbrV[on] = ghost[dirn][1][nbr_pty, g]
        else:
            idx = link_index_m1(coord, dim, dirn)
            nbrV = src[nbr_pty, idx].copy()
            if comm_dim[dirn]:
                on = coord[:, dirn] - nFace < 0
                if on.any():
                    g = ghost_face_index(0, coord[on], dim, dirn, nFace)
                    nbrV[on] = ghost[dirn][0][nbr_pty, g]
        # ---- neighbouring link, getNbrLinkDispExtG :39-66
        dx1 = [0, 0, 0, 0]
        c2 = coord + np.asarray(brd)
        if sign == DISP_SIGN_MINUS:
            dx1[dirn] -= 1
        # evenORodd(dx1) == 0 ? pty : 1 - pty      (C remainder: (-1) % 2 == -1 != 0)
        link_pty = pty if (sum(dx1) % 2 == 0) else 1 - pty
        lidx = link_index_shift(c2, dx1, dimEx)
        link = U[dirn, link_pty, lidx]
        if sign == DISP_SIGN_MINUS:
            link = np.conj(np.swapaxes(link, -1, -2))                   # conj(Matrix) = Hermitian conjugate
        # ---- R = nbrU * nbrV (Matrix * ColorSpinor: y(s,i) = sum_j A(i,j) x(s,j), j in order) :182
        R = np.zeros_like(nbrV)
        for j in range(3):
            R += link[:, None, :, j] * nbrV[:, :, None, j]
        dst[pty] = R                                                    # FillFermionSite :184
    return dst


# ----------------------------------------------------------------------------------------------
# a8  phase matrix                lib/mugiq_util_kernels.cu:3-35, include/contract_util.cuh:50-66
# ----------------------------------------------------------------------------------------------
def phase_matrix(mom, locV3, FTSign, localL, totalL, comm_coord=(0, 0, 0, 0), dtype=np.float64):
    """ph[v3 + locV3*im] = cos(2 pi phi) + i * FTSign * sin(2 pi phi),
    phi accumulated in Float over d<3 of mom[im][d] * gcoord[d] / (Float)totalL[d]; cos/sin evaluated in
    double with PI = 2.0*asin(1.0) (include/util_mugiq.h:7) and narrowed to Float."""
    cdt = np.complex128 if dtype == np.float64 else np.complex64
    mom = np.asarray(mom, dtype=np.int64).reshape(-1, 3)
    Nmom = mom.shape[0]
    tid = np.arange(locV3, dtype=np.int64)
    a1 = tid // localL[0]
    a2 = a1 // localL[1]
    lcoord = [tid - a1 * localL[0], a1 - a2 * localL[1], a2]
    gcoord = [lcoord[d] + comm_coord[d] * localL[d] for d in range(3)]
    PI = 2.0 * np.arcsin(1.0)
    out = np.zeros(locV3 * Nmom, dtype=cdt)
    for im in range(Nmom):
        phase = np.zeros(locV3, dtype=dtype)
        for d in range(3):
            phase = (phase + (mom[im, d] * gcoord[d]).astype(dtype) / dtype(totalL[d])).astype(dtype)
        arg = 2.0 * PI * phase.astype(np.float64)
        re = np.cos(arg).astype(dtype)
        im_ = (dtype(FTSign) * np.sin(arg).astype(dtype)).astype(dtype)
        out[locV3 * im:locV3 * (im + 1)] = re + 1j * im_
    return out


# ----------------------------------------------------------------------------------------------
# a9  index reorder + gamma5 map  lib/mugiq_util_kernels.cu:59-99, lib/contract_wrappers.cu:133-156
# ----------------------------------------------------------------------------------------------
def convert_idx_order_map_gamma(data_in, nData, nLoop, nParity, volumeCB, localL):
    """out[t + Lt*(index[ig] + 16*iL) + Lt*nData*v3] = sign[ig] * in[tid + V*(ig + 16*iL)],
    v3 = x + Lx*y + Lx*Ly*z, (x,y,z,t) = getCoords(x_cb, localL, pty)."""
    if nData != nLoop * N_GAMMA:
        raise ValueError("This function assumes that nData = nLoop * NGamma")
    V = nParity * volumeCB
    Lx, Ly, Lt = localL[0], localL[1], localL[3]
    sign = gamma_map_sign()
    out = np.zeros_like(data_in)
    for pty in range(nParity):
        x_cb = np.arange(volumeCB)
        crd = get_coords(x_cb, localL, pty)
        tid = x_cb + volumeCB * pty
        v3 = crd[:, 0] + Lx * crd[:, 1] + Lx * Ly * crd[:, 2]
        t = crd[:, 3]
        for ig in range(N_GAMMA):
            for iL in range(nLoop):
                idata_from = ig + N_GAMMA * iL
                idata_to = INDEX_MAP_GAMMA[ig] + N_GAMMA * iL
                out[t + Lt * idata_to + Lt * nData * v3] = sign[ig] * data_in[tid + V * idata_from]
    return out


# ----------------------------------------------------------------------------------------------
# a10  momentum projection        lib/loop_mugiq.cpp:322-434
# ----------------------------------------------------------------------------------------------
def momentum_projection_local(dataPosMP, phase, locT, nData, locV3, Nmom):
    """dataMom[M x N] = dataPosMP[M x K] * phase[K x N], column-major, M = locT*nData, K = locV3,
    N = Nmom (lib/loop_mugiq.cpp:363-378). Returns the flat column-major M x N result, i.e. index
    t + locT*idata + locT*nData*im."""
    M = locT * nData
    A = dataPosMP.reshape(locV3, M).T          # column-major M x K
    B = phase.reshape(Nmom, locV3).T           # column-major K x N
    C = A @ B
    return np.ascontiguousarray(C.T).reshape(-1)


# ----------------------------------------------------------------------------------------------
# a7  loop bookkeeping + driver   include/loop_mugiq.h:185-261, lib/loop_mugiq.cpp:439-525
# ----------------------------------------------------------------------------------------------
class LoopComputeParam:
    """Slot bookkeeping of Loop_Mugiq::LoopComputeParam (include/loop_mugiq.h:221-256)."""

    def __init__(self, disp_str=(), disp_start=(), disp_stop=(), doNonLocal=True):
        self.dispString, self.dispStart, self.dispStop = [], [], []
        self.nLoopPerEntry, self.nLoopOffset = [], []
        self.nLoop = 0
        self.doNonLocal = bool(doNonLocal)
        if doNonLocal:
            if not (len(disp_str) == len(disp_start) == len(disp_stop)):
                raise ValueError("Displacement string length not compatible with displacement limits length")
            for i in range(len(disp_str)):
                a, b = int(disp_start[i]), int(disp_stop[i])
                if a > b:                       # swapped with a warning, :234-239
                    a, b = b, a
                self.dispString.append(disp_str[i])
                self.dispStart.append(a)
                self.dispStop.append(b)
                self.nLoopPerEntry.append(b - a + 1)
                self.nLoop += b - a + 1
                self.nLoopOffset.append(1 + sum(self.nLoopPerEntry[:i]))
            self.nLoop += 1
        else:
            self.nLoop = 1
        self.nDispEntries = len(self.dispString)
        self.nData = self.nLoop * N_GAMMA


def parse_disp_entry_string(s):
    """tests/loop.cpp:607-705: "+z:1,8;-x:3" -> (disp_entry, disp_str, disp_start, disp_stop)."""
    entries = s.split(";")
    disp_entry, disp_str, start, stop = [], [], [], []
    for e in entries:
        parts = e.split(":")
        if len(parts) != 2:
            raise ValueError("Displacement entry has the wrong format: %r" % e)
        lims = [int(t) for t in parts[1].split(",")]
        if len(lims) == 0 or len(lims) > 2:
            raise ValueError("Wrong format of displacement entry %r" % e)
        disp_entry.append(e)
        disp_str.append(parts[0])
        start.append(lims[0])
        stop.append(lims[1] if len(lims) == 2 else lims[0])
    return disp_entry, disp_str, start, stop


def compute_loop_position_space(evecs, sigmas, cprm, U=None, dim=None, comm_dim=(0, 0, 0, 0),
                                brd=(0, 0, 0, 0), ghost_exchange=None, dtype=np.float64):
    """Loop_Mugiq::computeCoarseLoop up to the position-space buffer (lib/loop_mugiq.cpp:455-512).
    evecs: list of logical spinors [2, volCB, 4, 3]; sigmas: eVals_sigma (double, cast to Float :479).
    ghost_exchange(src) -> ghost[dir][bnd] arrays (the exchangeGhostVec of lib/contract_wrappers.cu:166-174);
    None on a single domain.  Returns dataPos flat [nLoop*16*V]."""
    cdt = np.complex128 if dtype == np.float64 else np.complex64
    vcb = evecs[0].shape[1]
    V = 2 * vcb
    perLoop = N_GAMMA * V
    dataPos = np.zeros(perLoop * cprm.nLoop, dtype=cdt)
    for idx in range(-1, cprm.nDispEntries):
        if idx >= 0:
            dirn, sign = parse_displacement(cprm.dispString[idx])
            off = perLoop * cprm.nLoopOffset[idx]
            dataPos[off:off + perLoop * cprm.nLoopPerEntry[idx]] = 0       # cudaMemset :476
        else:
            off = 0
            dataPos[0:perLoop] = 0
        for n in range(len(evecs)):
            sigma = dtype(sigmas[n])
            vL = evecs[n]
            if idx >= 0:
                vR = vL.copy()                                              # :487
                cnt = 0
                for idisp in range(1, cprm.dispStop[idx] + 1):              # :489
                    gh = ghost_exchange(vR) if ghost_exchange is not None else None
                    vR = covariant_displacement(vR, U, dirn, sign, dim, comm_dim, brd, gh)   # :490
                    if cprm.dispStart[idx] <= idisp <= cprm.dispStop[idx]:
                        s0 = off + perLoop * cnt
                        loop_contract(dataPos[s0:s0 + perLoop], vL, vR, sigma, dtype)        # :493
                        cnt += 1
            else:
                loop_contract(dataPos[0:perLoop], vL, vL, sigma, dtype)     # :501-502
    return dataPos


# ----------------------------------------------------------------------------------------------
# domain decomposition helpers (the role QUDA's comm grid / exchangeGhost / extended gauge play)
# ----------------------------------------------------------------------------------------------
def local_block(f_lex, coords, grid):
    """Slice the local block of rank `coords` (cx,cy,cz,ct) from a global lex field [T,Z,Y,X,...]."""
    T, Z, Y, X = f_lex.shape[:4]
    l = [X // grid[0], Y // grid[1], Z // grid[2], T // grid[3]]
    return f_lex[coords[3] * l[3]:(coords[3] + 1) * l[3],
                 coords[2] * l[2]:(coords[2] + 1) * l[2],
                 coords[1] * l[1]:(coords[1] + 1) * l[1],
                 coords[0] * l[0]:(coords[0] + 1) * l[0]]


def pack_face(v, dim, dirn, high):
    """Face of a local spinor [2, volCB, 4, 3] that a neighbour needs as a ghost zone (nFace = 1, the
    only value the reference supports: lib/contract_wrappers.cu:168).
      high = 0: the LOW face  x[dir] = 0         -> the backward neighbour's FORWARD ghost zone (bnd 1)
      high = 1: the HIGH face x[dir] = X[dir]-1  -> the forward neighbour's BACKWARD ghost zone (bnd 0)
    Output [2, faceCB, 4, 3] indexed [parity of the packed site, ghostFaceIndex], which is what the
    receiver addresses as Ghost(dir, bnd, nbrPty = 1 - pty, ghostFaceIndex<bnd>(coord)) because the
    leading term of ghostFaceIndex vanishes on the face for nFace = 1 and local dims are even."""
    vcb = v.shape[1]
    face_cb = int(np.prod(dim)) // dim[dirn] // 2
    out = np.zeros((2, face_cb) + v.shape[2:], dtype=v.dtype)
    for pty in range(2):
        coord = get_coords(np.arange(vcb), dim, pty)
        on = coord[:, dirn] == (dim[dirn] - 1 if high else 0)
        c = coord[on].copy()
        c[:, dirn] = 0
        g = ghost_face_index(0, c, dim, dirn, 1)
        out[pty, g] = v[pty, np.nonzero(on)[0]]
    return out


def extended_gauge_from_global(U_lex, coords, grid, brd):
    """Border-extended local gauge field (the product of lib/displace.cpp:104-134:
    copyExtendedGauge + exchangeExtendedGhost) built directly from the global field.
    U_lex [4, T, Z, Y, X, 3, 3] -> logical [4, 2, volExCB, 3, 3] on dimEx = local + 2*brd."""
    G = [U_lex.shape[4], U_lex.shape[3], U_lex.shape[2], U_lex.shape[1]]     # global X,Y,Z,T
    l = [G[d] // grid[d] for d in range(4)]
    dimEx = [l[d] + 2 * brd[d] for d in range(4)]
    if sum(brd) % 2 != 0:
        raise ValueError("sum of borders must be even (parity preserved by the border shift)")
    idx = [(np.arange(dimEx[d]) - brd[d] + coords[d] * l[d]) % G[d] for d in range(4)]
    out = []
    for mu in range(4):
        blk = U_lex[mu][np.ix_(idx[3], idx[2], idx[1], idx[0])]
        out.append(lex_to_eo(blk, dimEx))
    return np.stack(out, axis=0)


# ----------------------------------------------------------------------------------------------
# f2  coarse -> fine prolongation   Loop_Mugiq::prolongateEvec, lib/loop_mugiq.cpp:277-319
# ----------------------------------------------------------------------------------------------
# The reference calls QUDA's Transfer::P (lib/loop_mugiq.cpp:310,314) on a 2-level hierarchy with
# spin_block_size 2 (tests/loop.cpp:569), n_vec = 24 (:492) and 4^4 aggregates (:471).  ASSUMED from upstream
# QUDA (lib/transfer.cpp createGeoMap/createSpinMap, include/kernels/prolongator.cuh):
#   out(x; s, c) = sum_{j < n_vec} V(x; s, c, j) * in(X(x); s / spin_bs, j)
# with X(x) the aggregate of x (coarse coordinates = fine coordinates / geo_bs, even-odd indexed on the coarse
# lattice) and V the block-orthonormalised null vectors stored as a fine field with a packed vector index:
#   V:       FieldOrderCB<Float, 4, 3, n_vec, FLOAT2>  plane index (3*s + c)*n_vec + j
#   coarse:  FieldOrderCB<Float, 2, n_vec, 1, FLOAT2>  plane index  s*n_vec + j
# Logical shapes here: V[2, volCB, 4, 3, n_vec], coarse[2, volCB_coarse, 2, n_vec].
def fine_to_coarse_map(X, geo_bs):
    """For each parity: (coarse parity, coarse x_cb) of every fine checkerboard site. Returns int arrays [2, volCB]."""
    Xc = [X[d] // geo_bs[d] for d in range(4)]
    vcb = int(np.prod(X)) // 2
    cp = np.zeros((2, vcb), dtype=np.int64)
    cx = np.zeros((2, vcb), dtype=np.int64)
    for pty in range(2):
        c = get_coords(np.arange(vcb), X, pty)
        cc = c // np.asarray(geo_bs)
        cp[pty] = cc.sum(axis=1) & 1
        cx[pty] = lex_index(cc, Xc) >> 1
    return cp, cx


def prolongate(coarse, V, X, geo_bs=(4, 4, 4, 4), spin_bs=2):
    """Transfer::P for one level: finer[2, volCB, nSpin_f, nColor_f] from coarse[2, volCB_c, nSpin_f/spin_bs, n_vec] with
    V[2, volCB, nSpin_f, nColor_f, n_vec].  Finest level: nSpin_f 4, nColor_f 3, spin_bs 2.  Coarse -> coarse levels
    (transfer[lev-1]->P in lib/loop_mugiq.cpp:310): nSpin_f 2, nColor_f = n_vec of the finer level, spin_bs 1."""
    nvec = V.shape[-1]
    cp, cx = fine_to_coarse_map(X, geo_bs)
    out = np.zeros(V.shape[:4], dtype=V.dtype)
    for pty in range(2):
        phi = coarse[cp[pty], cx[pty]]                       # [volCB, nSpin_c, n_vec]
        for s in range(V.shape[2]):
            for j in range(nvec):                            # rotateFineColor: j in order
                out[pty, :, s, :] += V[pty, :, s, :, j] * phi[:, s // spin_bs, j, None]
    return out


def coarse_to_native(phi, stride=None, parity_offset=None):
    """Logical [2, volCB_c, nSpin_c, nColor_c] -> flat FLOAT2 buffer (plane index s*nColor_c + c)."""
    vcb, ns, nc = phi.shape[1], phi.shape[2], phi.shape[3]
    stride = vcb if stride is None else stride
    parity_offset = ns * nc * stride if parity_offset is None else parity_offset
    buf = np.zeros(2 * parity_offset, dtype=phi.dtype)
    x = np.arange(vcb)
    for p in range(2):
        for s in range(ns):
            for c in range(nc):
                buf[p * parity_offset + (s * nc + c) * stride + x] = phi[p, :, s, c]
    return buf


def nullvec_to_native(V, stride=None, parity_offset=None):
    """Logical [2, volCB, 4, 3, n_vec] -> flat FLOAT2 buffer (plane index (3*s + c)*n_vec + j)."""
    vcb, nvec = V.shape[1], V.shape[-1]
    stride = vcb if stride is None else stride
    parity_offset = 12 * nvec * stride if parity_offset is None else parity_offset
    buf = np.zeros(2 * parity_offset, dtype=V.dtype)
    x = np.arange(vcb)
    for p in range(2):
        for s in range(4):
            for c in range(3):
                for j in range(nvec):
                    buf[p * parity_offset + ((3 * s + c) * nvec + j) * stride + x] = V[p, :, s, c, j]
    return buf


def prolongate_levels(coarsest, Vs, Xs, geo_bss):
    """Loop_Mugiq::prolongateEvec for nCoarseLevels = len(Vs) (lib/loop_mugiq.cpp:306-314): Vs[0] / Xs[0] / geo_bss[0] belong
    to the finest transfer (spin_bs 2), Vs[l], l >= 1, to the transfer between level l and level l+1 (spin_bs 1, lattice
    Xs[l] = Xs[l-1] / geo_bss[l-1]); `coarsest` lives on level len(Vs)."""
    v = coarsest
    for l in range(len(Vs) - 1, 0, -1):
        v = prolongate(v, Vs[l], Xs[l], geo_bss[l], 1)       # transfer[lev-1]->P(tmpCSF[lev-1], tmpCSF[lev])
    return prolongate(v, Vs[0], Xs[0], geo_bss[0], 2)        # transfer[0]->P(fineEvec, tmpCSF[1])
