// Fused displaced contraction in an axial gauge on the fp64 matrix pipe (fourth generation).  Column tiles for mu = y, z, t, whole x rows
// for mu = x; ascending lengths up to 8 per entry; eigenvectors fp64 FLOAT2 (every tile geometry, and the face layers of posted halos
// written on the way) or fp64 FLOAT4 / fp32 FLOAT2 / fp32 FLOAT4 (converted on their way into LDS; 16-line tiles); slots fp64 or fp32.
//
// (1) The gauge.  Along every line of direction mu fix g(j + 1) = g(j) U_mu(x_j), g(0) = 1 (continued past both ends of
// the local line with the path-link products the driver has anyway: g(J + l) = g(J - 1) W_{l+1}(x_{J-1}), g(-l) = W^-_l(x_0)).
// Then g(x) U_mu(x) g(x + mu)^dag = 1, i.e. W_k(x) psi(x + k mu) = g(x)^dag [g psi](x + k mu), and because the colour trace
// does not see a unitary rotation of both factors
//        sum_c conj(v(x)[be][c]) (W_k(x) v(x + k mu))[al][c]  =  sum_c conj(v'(x)[be][c]) v'(x + k mu)[al][c],   v' = g v.
// The tile applies g ONCE per staged position and eigenvector (9 complex FMAs per spin) on the way into LDS, instead of W_k
// once per slot: 36 (4 + Kmax)/4 + 48 Kmax complex FMAs per site where csrc/fused_tile.hip spends 84 Kmax (Kmax = 3:
// 207 against 252), and no link field is read by the contraction any more.
//
// (2) The matrix pipe.  What is left per site, slot and eigenvector is a 4 x 3 times 3 x 4 complex product -- one block of
// v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 products per instruction, ONE LATTICE SITE PER BLOCK):
//     lane maps (tools/probes/mfma_4x4x4_layout.hip, one-hot operands):
//       A[b][i][k] in lane 16 k + 4 b + i,   B[b][k][j] in lane 16 k + 4 b + j,   D[b][i][j] in lane 16 i + 4 b + j
//     accR[be][al] += VR[be][c] PR[c][al] + VI[be][c] PI[c][al]          V = v'(x) / sigma_n, P = v'(x + k mu)
//     accI[be][al] += VR[be][c] PI[c][al] - VI[be][c] PR[c][al]          (colour index c padded 3 -> 4 with V = 0)
// Four products of 128 flops do the 384 flops of the outer product (75 %); V and P come from LDS with ONE ds_read_b128 per
// lane and 4-site group each (V is shared by the slots) and ARE the operands as they stand; the 4x4 colour-traced spin
// matrices accumulate in the D registers (2 x 2 VGPRs per group and slot).
//
// Tile and pipeline as csrc/fused_tile.hip: a workgroup owns 32 lines along mu x 4 consecutive positions and needs 4 + Kmax
// staged positions; thread (position, spin, line) loads its three colours two eigenvectors ahead into registers, rotates
// them with its g (in registers for the whole kernel) and commits v' to the other LDS buffer while the products of the
// current eigenvector run; one LDS-only barrier per eigenvector.
// LDS image: chunk (position pair, component) = [position & 1][32 lines] complex = 1 KiB, chunks 64 bytes apart in bank
// phase (stride 1088 B): an operand read -- lanes (colour, site, spin) -> component 3 spin + colour -- touches every bank
// once.  Measurements, and the form with W_k applied on the matrix pipe that this one replaces: profiles/r04_mfma_tile.txt.
#include "fused_mfma_kernel.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace mugiq {

// ---- the axial gauge of one (direction, sign) from the path-link fields E_k = W_k (FLOAT2, pad 0; component 3 j + i of
// E_k(x) is W_k(x)[i][j]): one thread per line, sequential along the line
template <typename F> struct AxialArgs {
  Cplx<double> *G;
  const Cplx<F> *E[kMT_MaxLength];  // E_1 .. E_kmax (storage precision; the gauge itself is kept in double)
  int kmax, sign, J, strideMu, H, numCols, volumeCB;
  int rowMode, X1, X2;  // mu = x: line = x row `cid`, site j <-> (parity p0 ^ (j & 1), entry cid J/2 + j/2); G is [9][row][position]
};
template <typename F> __device__ inline void mt_load_w(Cplx<double> w[9], const Cplx<F> *E, int par, int x_cb, int volumeCB) {
  const Cplx<F> *e = E + (int64_t)par * 12 * volumeCB + x_cb;
#pragma unroll
  for (int j = 0; j < 3; j++)
#pragma unroll
    for (int i = 0; i < 3; i++) {
      const Cplx<F> u = e[(int64_t)(j * 3 + i) * volumeCB];
      w[i * 3 + j] = Cplx<double>{(double)u.re, (double)u.im};
    }
}
// r = x y (DAG: x y^dag)
template <bool DAG> __device__ inline void mt_mul3(Cplx<double> r[9], const Cplx<double> x[9], const Cplx<double> y[9]) {
#pragma unroll
  for (int i = 0; i < 3; i++)
#pragma unroll
    for (int j = 0; j < 3; j++) {
      Cplx<double> s{0.0, 0.0};
#pragma unroll
      for (int m = 0; m < 3; m++) {
        if (DAG) cmadd(s, x[i * 3 + m], Cplx<double>{y[j * 3 + m].re, -y[j * 3 + m].im});
        else cmadd(s, x[i * 3 + m], y[m * 3 + j]);
      }
      r[i * 3 + j] = s;
    }
}
template <typename F> __global__ __launch_bounds__(64) void axial_gauge_kernel(AxialArgs<F> a) {
  const int cid = blockIdx.x * 64 + threadIdx.x;
  if (cid >= a.numCols) return;
  int p0, base;
  if (a.rowMode) {
    const int zt = cid / a.X1;
    p0 = (cid % a.X1 + zt % a.X2 + zt / a.X2) & 1;
    base = cid * (a.J >> 1);
  } else {
    mt_line(cid, a.H, a.strideMu, a.J, p0, base);
  }
  const int Jext = a.J + a.kmax;
  auto store = [&](int jext, const Cplx<double> g[9]) {
#pragma unroll
    for (int c = 0; c < 9; c++) a.G[a.rowMode ? ((int64_t)c * a.numCols + cid) * Jext + jext : ((int64_t)c * Jext + jext) * a.numCols + cid] = g[c];
  };
  auto site_xcb = [&](int j) { return a.rowMode ? base + (j >> 1) : base + j * a.strideMu; };
  Cplx<double> g[9], w[9], t[9];
#pragma unroll
  for (int c = 0; c < 9; c++) g[c] = Cplx<double>{c % 4 == 0 ? 1.0 : 0.0, 0.0};
  const int off = a.sign == MUGIQ_HIP_DISP_SIGN_PLUS ? 0 : a.kmax;
  if (a.sign == MUGIQ_HIP_DISP_SIGN_MINUS) {  // g(-l) = W^-_l(x_0)
    for (int l = 1; l <= a.kmax; l++) {
      mt_load_w(w, a.E[l - 1], p0, site_xcb(0), a.volumeCB);
      store(a.kmax - l, w);
    }
  }
  for (int j = 0; j < a.J; j++) {
    const int par = p0 ^ (j & 1), x_cb = site_xcb(j);
    if (a.sign == MUGIQ_HIP_DISP_SIGN_MINUS && j > 0) {  // g(j) = g(j - 1) W^-_1(x_j)^dag      (W^-_1(x) = U(x - mu)^dag)
      mt_load_w(w, a.E[0], par, x_cb, a.volumeCB);
      mt_mul3<true>(t, g, w);
#pragma unroll
      for (int c = 0; c < 9; c++) g[c] = t[c];
    }
    store(j + off, g);
    if (a.sign == MUGIQ_HIP_DISP_SIGN_PLUS) {
      if (j == a.J - 1) {  // g(J + l) = g(J - 1) W_{l+1}(x_{J-1})
        for (int l = 0; l < a.kmax; l++) {
          mt_load_w(w, a.E[l], par, x_cb, a.volumeCB);
          mt_mul3<false>(t, g, w);
          store(a.J + l, t);
        }
      } else {  // g(j + 1) = g(j) W_1(x_j)
        mt_load_w(w, a.E[0], par, x_cb, a.volumeCB);
        mt_mul3<false>(t, g, w);
#pragma unroll
        for (int c = 0; c < 9; c++) g[c] = t[c];
      }
    }
  }
}

// ---- the same gauge straight from the gauge field, for a direction that is NOT partitioned (the local line is the global, periodic
// one): W_1(x) = U_mu(x), so g(j + 1) = g(j) U_mu(x_j) for either sign, continued with the links of the wrapped sites --
// g(J + l) = g(J + l - 1) U(x_{(J + l - 1) mod J}), g(-l) = g(-l + 1) U(x_{J - l})^dag -- and no path-link field has to be built at all
// (the driver's chain of `stop` covariant displacements of the identity: 0.4 ms and a GB of scratch per entry at configs[2]).
template <typename F> struct AxialLinkArgs {
  Cplx<double> *G;
  const Cplx<F> *U;   // extended gauge field: parity * Upo + (dir * 9 + row * 3 + col) * Ustride + x_cb on the extended lattice
  int64_t Upo;
  int Ustride;
  int X[4], R[4];
  int dir, kmax, sign, J, strideMu, H, numCols;
  int rowMode;
};
template <typename F> __global__ __launch_bounds__(64) void axial_gauge_from_links_kernel(AxialLinkArgs<F> a) {
  const int cid = blockIdx.x * 64 + threadIdx.x;
  if (cid >= a.numCols) return;
  int p0, base;
  if (a.rowMode) {
    const int zt = cid / a.X[1];
    p0 = (cid % a.X[1] + zt % a.X[2] + zt / a.X[2]) & 1;
    base = cid * (a.J >> 1);
  } else {
    mt_line(cid, a.H, a.strideMu, a.J, p0, base);
  }
  int c0[4], XE[4];
  get_coords(c0, base, a.X, p0);  // the j = 0 site of the line (row mode: x = 0 or 1 -- only the other three coordinates are used)
#pragma unroll
  for (int d = 0; d < 4; d++) XE[d] = a.X[d] + 2 * a.R[d];
  // U_mu at position j of the line; j < 0 or j >= J: the wrapped site of a periodic line, or -- along a partitioned direction -- the
  // neighbour's link in the border of the extended field (-R <= j < J + R)
  auto load_u = [&](Cplx<double> u[9], int j) {
    const int jj = a.R[a.dir] > 0 ? j : ((j % a.J) + a.J) % a.J;
    int c[4];
#pragma unroll
    for (int d = 0; d < 4; d++) c[d] = (d == a.dir ? jj : c0[d]) + a.R[d];
    const int par = p0 ^ (j & 1);  // (borders are even in sum: the extended parity is the interior one)
    const Cplx<F> *q = a.U + (int64_t)par * a.Upo + (int64_t)(a.dir * 9) * a.Ustride + (lex_index(c, XE) >> 1);
#pragma unroll
    for (int e = 0; e < 9; e++) {
      const Cplx<F> v = q[(int64_t)e * a.Ustride];
      u[e] = Cplx<double>{(double)v.re, (double)v.im};
    }
  };
  const int Jext = a.J + a.kmax;
  auto store = [&](int jext, const Cplx<double> g[9]) {
#pragma unroll
    for (int c = 0; c < 9; c++) a.G[a.rowMode ? ((int64_t)c * a.numCols + cid) * Jext + jext : ((int64_t)c * Jext + jext) * a.numCols + cid] = g[c];
  };
  Cplx<double> g[9], w[9], t[9];
#pragma unroll
  for (int c = 0; c < 9; c++) g[c] = Cplx<double>{c % 4 == 0 ? 1.0 : 0.0, 0.0};
  const int off = a.sign == MUGIQ_HIP_DISP_SIGN_PLUS ? 0 : a.kmax;
  if (a.sign == MUGIQ_HIP_DISP_SIGN_MINUS) {  // g(-l) = g(-l + 1) U(x_{J - l})^dag
    for (int l = 1; l <= a.kmax; l++) {
      load_u(w, -l);
      mt_mul3<true>(t, g, w);
#pragma unroll
      for (int c = 0; c < 9; c++) g[c] = t[c];
      store(a.kmax - l, g);
    }
#pragma unroll
    for (int c = 0; c < 9; c++) g[c] = Cplx<double>{c % 4 == 0 ? 1.0 : 0.0, 0.0};
  }
  const int last = a.sign == MUGIQ_HIP_DISP_SIGN_PLUS ? a.J + a.kmax : a.J;
  for (int j = 0; j < last; j++) {
    store(j + off, g);
    if (j + 1 < last) {
      load_u(w, j);
      mt_mul3<false>(t, g, w);
#pragma unroll
      for (int c = 0; c < 9; c++) g[c] = t[c];
    }
  }
}

// The tile geometry for an entry: the first TJ of {8, 12, 4} that divides the extent and keeps TJ + Kmax within the staged
// positions of its line count (MUGIQ_HIP_MFMA_TJ = 4 | 8 | 12 fixes it); 0 = none.
static int mfma_tile_tj(int extent, int kmax, int nSlots = kMT_MaxSlots, bool partitioned = true, bool reduced = false) {
  int want = 0;
  if (const char *e = getenv("MUGIQ_HIP_MFMA_TJ")) want = atoi(e);
  // 12 x 16 sites (1 + K/12 units staged per site) where it keeps its registers -- three groups per wave: up to three slots -- and the
  // line is not partitioned (of 24 / 12 = 2 tiles along the line one would be a boundary tile): "+z:1,3;+t:1,3" N_ev 200 32.6 against
  // 33.4 ms.  Else 8 x 16, then 12 x 16 (spills with four slots), then 4 x 32 (512-byte runs, but 1 + K/4 units).
  const int first = (!partitioned && nSlots < kMT_MaxSlots) ? 12 : 8;
  for (int tj : {first, 8, 12, 4}) {
    if (want && tj != want) continue;
    if (reduced && tj == 4) continue;  // (storage types other than fp64 FLOAT2 come with the 16-line tiles only)
    if (extent % tj != 0 || tj + kmax > (tj == 4 ? 8 : 16)) continue;
    return tj;
  }
  return 0;
}

// mu = x: R whole rows per workgroup of W waves, G = 2 | 3 groups of 4 sites per wave: R X0 = 16 G W sites.  Two workgroups of 8
// waves per CU where the rows allow, else one of 16 (MUGIQ_HIP_MFMA_ROW_WAVES = 8 | 16 fixes it).
static bool mfma_reduced(const MugiqHipSpinorField &ev) { return !(ev.precision == 8 && ev.field_order == 2); }
static bool mfma_row_geometry(const MugiqHipSpinorField &ev, int *groups, int *rows, int *waves) {
  const int epr = ev.X[0] / 2, nRows = ev.volumeCB / epr;
  if (epr % 4 != 0) return false;
  int want = 0;
  if (const char *e = getenv("MUGIQ_HIP_MFMA_ROW_WAVES")) want = atoi(e);
  for (int w : {8, 16}) {  // (X0 = 48, N_ev 200, spill-free kernels: two workgroups of 8 waves per CU 13.1 ms per entry, one of 16 13.6)
    if (want && w != want) continue;
    if (mfma_reduced(ev) && w != 8) continue;  // (... and with the 8-wave row tile only)
    for (int g : {3, 2}) {
      if ((2 * g * w) % epr != 0) continue;
      const int r = 2 * g * w / epr;
      if (nRows % r != 0 || r * 8 * (epr + kMT_MaxLength / 2) > 64 * w) continue;
      if (24 * ((r * (epr + kMT_MaxLength / 2) + 12) / 16 * 16 + 4) > (w == 8 ? kMT_BufElems / 2 : kMT_BufElems)) continue;  // the LDS image of a tile buffer
      *groups = g;
      *rows = r;
      *waves = w;
      return true;
    }
  }
  return false;
}

// Can the axial-gauge tile take this entry?  fp64 FLOAT2 storage and loops, mu = y, z, t, lengths 1 .. Kmax (the gauge is
// built from W_1 .. W_Kmax).  MUGIQ_HIP_TILE_MFMA = 0 switches it off (the vector tiles of csrc/fused_tile.hip /
// fused_tile16.hip take over).
bool mfma_tile_applicable(const MugiqHipSpinorField &ev, int dir, const int *kvals, int nK, int partitioned, bool gaugeGiven) {
  if (const char *e = getenv("MUGIQ_HIP_TILE_MFMA"))
    if (atoi(e) == 0) return false;
  if (const char *e = getenv("MUGIQ_HIP_FUSED_TILE"))
    if (atoi(e) == 0) return false;  // streaming kernel only
  if (const char *e = getenv("MUGIQ_HIP_TILE_COLS"))
    if (atoi(e) != 0) return false;  // a vector-tile generation was asked for by name
  if (const char *e = getenv("MUGIQ_HIP_TILE_GLDS"))
    if (atoi(e) == 0) return false;  // register-staged vector tile asked for
  if ((ev.precision != 8 && ev.precision != 4) || (ev.field_order != 2 && ev.field_order != 4)) return false;
  if (const char *e = getenv("MUGIQ_HIP_MFMA_STORAGE"))
    if (atoi(e) == 0 && mfma_reduced(ev)) return false;  // fp64 FLOAT2 only, as before
  if (2 * (int64_t)ev.parity_offset >= (1LL << 31)) return false;  // the kernel keeps 32-bit element offsets
  // lengths ascending; from 1 without a gap where the tile has to build the gauge itself (from W_1 .. W_Kmax = the links it is
  // handed); any ascending list where the caller has built the gauge (the driver holds W_1 .. W_stop whatever the entry starts at)
  for (int i = 0; i < nK; i++)
    if (kvals[i] < 1 || (i > 0 && kvals[i] <= kvals[i - 1]) || (!gaugeGiven && kvals[i] != i + 1)) return false;
  const int kmax = kvals[nK - 1];
  if (kmax > kMT_MaxLength || kmax > ev.X[dir]) return false;
  if (dir == 0) {  // whole x rows: no ghost handling
    int g, r, w;
    if (const char *e = getenv("MUGIQ_HIP_MFMA_ROW"))
      if (atoi(e) == 0) return false;
    if (2 * (int64_t)ev.parity_offset >= (1LL << 28)) return false;  // (the row tile keeps 32-bit BYTE offsets)
    return !partitioned && mfma_row_geometry(ev, &g, &r, &w);
  }
  return mfma_tile_tj(ev.X[dir], kmax, kMT_MaxSlots, true, mfma_reduced(ev)) != 0;
}

static int launch_mfma_tile(const MTileArgs &a, int precision, int order, int dir, int sign, int ns, int tj, int rowGroups, int rowWaves, hipStream_t stream) {
  if (precision == 8 && order == 2) return launch_mfma_tile_t<double, 2, true>(a, dir, sign, ns, tj, rowGroups, rowWaves, stream);
  if (precision == 8) return launch_mfma_tile_d4(a, dir, sign, ns, tj, rowGroups, rowWaves, stream);
  if (order == 2) return launch_mfma_tile_f2(a, dir, sign, ns, tj, rowGroups, rowWaves, stream);
  return launch_mfma_tile_f4(a, dir, sign, ns, tj, rowGroups, rowWaves, stream);
}

namespace {
struct AxialHint {
  const void *G = nullptr, *E1 = nullptr;
  int dir = -1, sign = -1, kmax = 0;
};
thread_local AxialHint g_hint;
}  // namespace

namespace {
struct PackHint {
  int n = 0;
  bool taken = false;
  EntryPackTarget t[kMT_MaxPack];
};
thread_local PackHint g_pack;
}  // namespace

// Face layers the next mu = x entry of this host thread writes on its way through the eigenvectors (see MTileArgs::pack).
// entry_pack_capacity: how many targets such an entry can take (0: it cannot -- not the row tile, or rows of a workgroup would
// straddle a z / t coordinate); entry_pack_taken: did the entry launched since the last set_entry_pack_hint do it?
int entry_pack_capacity(const MugiqHipSpinorField &ev, const int *kvals, int nK) {
  if (const char *e = getenv("MUGIQ_HIP_PACK_IN_ENTRY"))
    if (atoi(e) == 0) return 0;
  if (mfma_reduced(ev)) return 0;
  if (!mfma_tile_applicable(ev, 0, kvals, nK, 0, true)) return 0;  // (the driver builds the gauge where the lengths do not start at 1)
  int g, r, w;
  if (!mfma_row_geometry(ev, &g, &r, &w) || ev.X[1] % r != 0) return 0;
  if ((int64_t)ev.X[1] * (ev.X[0] / 2) >= (1 << 20)) return 0;  // (face entry within its (z | t) slice: 20 bits in the kernel)
  return kMT_MaxPack;
}
void set_entry_pack_hint(const EntryPackTarget *targets, int n) {
  g_pack.n = targets ? std::min(n, kMT_MaxPack) : 0;
  g_pack.taken = false;
  for (int i = 0; i < g_pack.n; i++) g_pack.t[i] = targets[i];
}
bool entry_pack_taken() { return g_pack.taken; }

void set_axial_gauge_hint(const void *G_d, const void *E1_d, int dir, int sign, int kmax) {
  g_hint.G = G_d;
  g_hint.E1 = E1_d;
  g_hint.dir = dir;
  g_hint.sign = sign;
  g_hint.kmax = kmax;
}

size_t axial_gauge_bytes(const MugiqHipSpinorField &ev, int dir, const int *kvals, int nK, int partitioned) {
  if (!mfma_tile_applicable(ev, dir, kvals, nK, partitioned, true)) return 0;
  return (size_t)9 * (ev.X[dir] + kvals[nK - 1]) * (size_t)(2 * ev.volumeCB / ev.X[dir]) * sizeof(Cplx<double>);
}
bool axial_gauge_hint_matches(const void *E0_d, int dir, int sign, int kmax) {
  return g_hint.G && g_hint.E1 == E0_d && g_hint.dir == dir && g_hint.sign == sign && g_hint.kmax == kmax;
}

template <typename F>
static int build_axial_gauge_t(void *G_d, const MugiqHipSpinorField &ev, const void *const *E_d, int kmax, int dir, int sign, hipStream_t stream) {
  AxialArgs<F> g;
  g.G = static_cast<Cplx<double> *>(G_d);
  for (int l = 0; l < kMT_MaxLength; l++) g.E[l] = static_cast<const Cplx<F> *>(E_d[l < kmax ? l : 0]);
  long long strideMu = 1;
  for (int d = 0; d < dir; d++) strideMu *= ev.X[d];
  g.kmax = kmax;
  g.sign = sign;
  g.J = ev.X[dir];
  g.strideMu = dir == 0 ? 1 : (int)(strideMu / 2);
  g.H = (int)(ev.volumeCB / ((long long)ev.X[dir] * g.strideMu));
  g.numCols = 2 * ev.volumeCB / ev.X[dir];
  g.volumeCB = ev.volumeCB;
  g.rowMode = dir == 0;
  g.X1 = ev.X[1];
  g.X2 = ev.X[2];
  hipLaunchKernelGGL(axial_gauge_kernel<F>, dim3((g.numCols + 63) / 64), dim3(64), 0, stream, g);
  MUGIQ_CHECK_HIP(hipGetLastError());
  return MUGIQ_HIP_SUCCESS;
}
// (the path-link fields are FLOAT2, pad 0, in the eigenvectors' precision)
int build_axial_gauge(void *G_d, const MugiqHipSpinorField &ev, const void *const *E_d, int kmax, int dir, int sign, hipStream_t stream) {
  return ev.precision == 8 ? build_axial_gauge_t<double>(G_d, ev, E_d, kmax, dir, sign, stream)
                           : build_axial_gauge_t<float>(G_d, ev, E_d, kmax, dir, sign, stream);
}

template <typename F>
static int build_axial_gauge_links_t(void *G_d, const MugiqHipSpinorField &ev, const MugiqHipGaugeField &U, int kmax, int dir, int sign, hipStream_t stream) {
  AxialLinkArgs<F> g;
  g.G = static_cast<Cplx<double> *>(G_d);
  g.U = static_cast<const Cplx<F> *>(U.data);
  g.Upo = U.parity_offset;
  g.Ustride = U.stride;
  long long strideMu = 1;
  for (int d = 0; d < 4; d++) {
    g.X[d] = ev.X[d];
    g.R[d] = U.R[d];
    if (d < dir) strideMu *= ev.X[d];
  }
  g.dir = dir;
  g.kmax = kmax;
  g.sign = sign;
  g.J = ev.X[dir];
  g.strideMu = dir == 0 ? 1 : (int)(strideMu / 2);
  g.H = (int)(ev.volumeCB / ((long long)ev.X[dir] * g.strideMu));
  g.numCols = 2 * ev.volumeCB / ev.X[dir];
  g.rowMode = dir == 0;
  hipLaunchKernelGGL(axial_gauge_from_links_kernel<F>, dim3((g.numCols + 63) / 64), dim3(64), 0, stream, g);
  MUGIQ_CHECK_HIP(hipGetLastError());
  return MUGIQ_HIP_SUCCESS;
}
// Can the gauge of (dir, sign) with lengths up to kmax be taken from the gauge field?  Always along a direction that is not partitioned
// (border 0: periodic line); along a partitioned one as far as the border of the extended field reaches: the continued positions need the
// links at J .. J + kmax - 2 (sign +) or -1 .. -kmax (sign -).
bool axial_gauge_from_links_possible(const MugiqHipSpinorField &ev, const MugiqHipGaugeField &U, int kmax, int dir, int sign) {
  if (U.precision != ev.precision) return false;
  if (const char *e = getenv("MUGIQ_HIP_GAUGE_FROM_LINKS"))
    if (atoi(e) == 0) return false;
  const int R = U.R[dir];
  return R == 0 || (sign == MUGIQ_HIP_DISP_SIGN_PLUS ? kmax <= R + 1 : kmax <= R);
}
int build_axial_gauge_from_links(void *G_d, const MugiqHipSpinorField &ev, const MugiqHipGaugeField &U, int kmax, int dir, int sign, hipStream_t stream) {
  MUGIQ_REQUIRE(axial_gauge_from_links_possible(ev, U, kmax, dir, sign), "axial gauge from the links: precision %d / border %d along %d, lengths up to %d (internal)", U.precision, U.R[dir], dir, kmax);
  return ev.precision == 8 ? build_axial_gauge_links_t<double>(G_d, ev, U, kmax, dir, sign, stream)
                           : build_axial_gauge_links_t<float>(G_d, ev, U, kmax, dir, sign, stream);
}

// ultra_d != NULL: also produce the ultra-local loop (k = 0) into ultra_d as one more slot; *carried says whether that
// happened (only a launch over the whole lattice may: see csrc/fused_tile.hip)
int mfma_tile_entry(void *loop_d, int loopPrecision, const MugiqHipSpinorField *ev, const double *sigma, int nVec, const void *const *E_d,
                    const int *kvals, int nK, int dir, int sign, int partitioned, const void *ghost_d, int layers, int region,
                    hipStream_t stream, void *ultra_d, int *carried) {
  const size_t ptr_bytes = sizeof(void *) * (size_t)nVec;
  std::vector<unsigned char> host(ptr_bytes + sizeof(double) * (size_t)nVec);
  const void **hl = reinterpret_cast<const void **>(host.data());
  double *hs = reinterpret_cast<double *>(host.data() + ptr_bytes);
  for (int n = 0; n < nVec; n++) {
    hl[n] = ev[n].data;
    hs[n] = 1.0 / sigma[n];
  }
  void *dev = nullptr;
  int st = upload_table(&dev, host.data(), host.size(), stream);
  if (st) return st;
  MTileArgs a;
  const int64_t slot_stride = (int64_t)16 * 2 * ev[0].volumeCB * 2 * loopPrecision;  // bytes
  a.outFloat = loopPrecision == 4;
  if (carried) *carried = 0;
  a.L = reinterpret_cast<const void *const *>(dev);
  a.inv_sigma = reinterpret_cast<const double *>(static_cast<unsigned char *>(dev) + ptr_bytes);
  a.nVec = nVec;
  long long strideMu = 1;
  for (int d = 0; d < 4; d++) {
    a.X[d] = ev[0].X[d];
    if (d < dir) strideMu *= ev[0].X[d];
  }
  strideMu = dir == 0 ? 1 : strideMu / 2;  // (unused by the row tile: a step along x is half a checkerboard entry)
  a.volumeCB = ev[0].volumeCB;
  a.stride = ev[0].stride;
  a.parity_offset = ev[0].parity_offset;
  a.partitioned = partitioned;
  a.ghost = ghost_d;
  a.faceCB = ev[0].volumeCB / ev[0].X[dir];
  a.ghost_vec_stride = (int64_t)layers * 24 * a.faceCB;
  a.strideMu = (int)strideMu;
  a.H = (int)(ev[0].volumeCB / (ev[0].X[dir] * strideMu));
  a.numCols = 2 * ev[0].volumeCB / ev[0].X[dir];
  int tj = 0, rowGroups = 0, rowWaves = 0;
  a.rowsPerTile = a.rowChunk = 0;
  if (dir == 0) {
    MUGIQ_REQUIRE(mfma_row_geometry(ev[0], &rowGroups, &a.rowsPerTile, &rowWaves), "mfma tile: no row geometry for X0 = %d (internal)", ev[0].X[0]);
    a.rowChunk = (a.rowsPerTile * (ev[0].X[0] / 2 + kMT_MaxLength / 2) + 12) / 16 * 16 + 4;  // > R (X0/2 + 4) (the entry behind the rows holds the zero of the padded operand lanes), and 4 mod 16 entries: 16 banks of phase per component
    MUGIQ_REQUIRE(24 * a.rowChunk <= (rowWaves == 8 ? kMT_BufElems / 2 : kMT_BufElems), "mfma tile: row image of %d entries per chunk does not fit (internal)", a.rowChunk);
    ultra_d = nullptr;  // (the row tile takes no fourth slot)
    tj = ev[0].X[0];    // one "tile" along mu
  } else {
    tj = mfma_tile_tj(ev[0].X[dir], kvals[nK - 1], kMT_MaxSlots, true, mfma_reduced(ev[0]));
    MUGIQ_REQUIRE(tj != 0, "mfma tile: no tile geometry for extent %d, lengths up to %d (internal)", ev[0].X[dir], kvals[nK - 1]);
  }
  const int nJT = ev[0].X[dir] / tj;
  a.overwrite = (region & MUGIQ_HIP_REGION_OVERWRITE) ? 1 : 0;
  region &= 0xff;
  if (region != MUGIQ_HIP_REGION_ALL) ultra_d = nullptr;
  a.kmaxG = kvals[nK - 1];  // (mfma_tile_applicable: ascending; 1 .. nK unless the caller's gauge is at hand)
  // the axial gauge of this (direction, sign): the caller's, if it has built one from these links; else rebuilt into the stream's
  // workspace (one pass over W_1)
  if (axial_gauge_hint_matches(E_d[0], dir, sign, a.kmaxG)) {
    a.G = static_cast<const Cplx<double> *>(g_hint.G);
  } else {
    MUGIQ_REQUIRE(a.kmaxG == nK, "mfma tile: lengths %d .. %d without the caller's axial gauge (internal)", kvals[0], a.kmaxG);
    void *gbuf = nullptr;
    if ((st = stream_workspace(&gbuf, (size_t)9 * (ev[0].X[dir] + a.kmaxG) * a.numCols * sizeof(Cplx<double>), stream))) return st;
    if ((st = build_axial_gauge(gbuf, ev[0], E_d, a.kmaxG, dir, sign, stream))) return st;
    a.G = static_cast<const Cplx<double> *>(gbuf);
  }
  // launches of up to four slots each (the first one may carry the ultra-local loop as its fourth); a launch of lengths
  // k0 .. k1 stages the TJ + k1 positions its sites and their shifted partners live on
  for (int first = 0, ns = 0; first < nK; first += ns) {
    const bool takesUltra = ultra_d && first == 0;
    const int room = dir == 0 ? kMT_MaxSlots - 1 : kMT_MaxSlots;  // (the row tile has no four-slot instance)
    const int slotsLeft = nK - first + (takesUltra ? 1 : 0), launchesLeft = (slotsLeft + room - 1) / room;
    ns = (slotsLeft + launchesLeft - 1) / launchesLeft - (takesUltra ? 1 : 0);  // evenly: 1 .. 8 with the ultra-local loop = 3 + 3 + 3 slots
    a.kmax = kvals[first + ns - 1];
    for (int s = 0; s < kMT_MaxSlots; s++) {
      const int i = first + (s < ns ? s : 0);
      a.k[s] = kvals[i];
      a.out[s] = static_cast<char *>(loop_d) + (int64_t)i * slot_stride;
    }
    bool withUltra = false;
    int nSlots = ns;
    if (takesUltra) {
      a.k[nSlots] = 0;
      a.out[nSlots] = ultra_d;
      nSlots++;
      withUltra = true;
    }
    // the tile of THIS launch (its slots and the positions it stages; the gauge does not depend on it)
    int tjL = tj, nJTL = nJT;
    if (dir != 0) {
      const int t2 = mfma_tile_tj(ev[0].X[dir], a.kmax, nSlots, partitioned != 0, mfma_reduced(ev[0]));
      if (t2) tjL = t2;
      nJTL = ev[0].X[dir] / tjL;
    }
    // region 0: everything | 1: tiles whose shifted reads stay inside the local lattice | 2: tiles that read ghost layers
    // (the split is by the ENTRY's longest length, so that the interior and the boundary launch of a slot cover complementary tiles)
    a.nPack = 0;
    if (dir == 0 && first == 0 && g_pack.n > 0 && !g_pack.taken && ev[0].X[1] % a.rowsPerTile == 0) {  // the first launch of the entry packs
      a.nPack = g_pack.n;
      for (int i = 0; i < g_pack.n; i++) {
        const int fcb = ev[0].volumeCB / ev[0].X[g_pack.t[i].dim];
        a.pack[i].base = static_cast<Cplx<double> *>(g_pack.t[i].out_d);
        a.pack[i].dim = g_pack.t[i].dim;
        a.pack[i].high = g_pack.t[i].high;
        a.pack[i].layers = g_pack.t[i].layers;
        a.pack[i].from = g_pack.t[i].fromVec;
        a.pack[i].faceCB = fcb;
        a.pack[i].vecStride = (int64_t)g_pack.t[i].layers * 24 * fcb;
      }
      g_pack.taken = true;
    }
    a.jtBegin = 0;
    a.jtCount = nJTL;
    if (region != MUGIQ_HIP_REGION_ALL) {
      const int nb = partitioned ? std::min(nJTL, (a.kmaxG + tjL - 1) / tjL) : 0;  // boundary tiles
      if (region == MUGIQ_HIP_REGION_INTERIOR) {
        a.jtBegin = sign == MUGIQ_HIP_DISP_SIGN_PLUS ? 0 : nb;
        a.jtCount = nJTL - nb;
      } else {
        a.jtBegin = sign == MUGIQ_HIP_DISP_SIGN_PLUS ? nJTL - nb : 0;
        a.jtCount = nb;
      }
    }
    if (a.jtCount > 0) {
      st = launch_mfma_tile(a, ev[0].precision, ev[0].field_order, dir, sign, nSlots, tjL, rowGroups, rowWaves, stream);
      if (st) return st;
      if (withUltra && carried) *carried = 1;
    }
  }
  return MUGIQ_HIP_SUCCESS;
}

}  // namespace mugiq
