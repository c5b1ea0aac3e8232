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
#include "internal.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace mugiq {

// Tile geometries (TJ positions along mu x LN lines per workgroup; 16 waves = 1024 threads each):
//   TJ =  4, LN = 32: 128 sites, 4 + Kmax <=  8 staged positions, 1 + Kmax/4  units requested per site (the tile of csrc/fused_tile.hip)
//   TJ =  8, LN = 16: 128 sites, 8 + Kmax <= 16 staged positions, 1 + Kmax/8
//   TJ = 12, LN = 16: 192 sites, 12 + Kmax <= 16 staged positions, 1 + Kmax/12
// 4-site groups: TJ LN / 4, an equal share per wave (2 | 2 | 3), group -> (position, 4 consecutive lines).
constexpr int kMT_Waves = 16;
constexpr int kMT_MaxSlots = 4;   // 3 displaced slots + the ultra-local loop riding along (k = 0) per launch
constexpr int kMT_MaxLength = 8;  // lengths 1 .. 8 per entry (launches of three lengths; 4 x 32 tiles: 1 .. 4)
constexpr int kMT_MaxPack = 4;   // face-layer targets a row-tile launch can fill on the way (z and t, low and high side)
constexpr int kMT_Chunk = 68;     // complex elements per chunk: 64 + 4 of bank phase
constexpr int kMT_Chunks = 4 * 12;  // chunks of a tile buffer: 64 / LN positions each, 12 components, <= 4 * 64 / LN staged positions
constexpr int kMT_BufElems = kMT_Chunks * kMT_Chunk;

struct MTileArgs {
  void *out[kMT_MaxSlots];  // Cplx<double> | Cplx<float> (outFloat)
  int outFloat;
  const void *const *L;
  const double *inv_sigma;
  int nVec;
  int X[4];
  int volumeCB;
  int stride;
  int64_t parity_offset;
  const Cplx<double> *G;  // the axial gauge: [9][J + kmax][numCols] (sign +: position j | sign -: position j + kmax)
  int k[kMT_MaxSlots];
  int kmax;       // largest length of THIS launch (staged window: TJ + kmax positions)
  int kmaxG;      // largest length of the entry: the axial gauge is continued that far (G holds J + kmaxG positions per line)
  int partitioned;
  const void *ghost;  // ghost layers in the eigenvectors' precision and order
  int64_t ghost_vec_stride;
  int faceCB;
  int strideMu;   // x_cb distance of one step along DIR
  int H;          // volumeCB / (X[DIR] * strideMu)
  int numCols;    // V / X[DIR]
  int jtBegin;    // tiles along mu handled by this launch: [jtBegin, jtBegin + jtCount)
  int jtCount;
  int blockOrder; // bit 1: XCD-contiguous workgroup order
  int overwrite;  // store instead of accumulate (MUGIQ_HIP_REGION_OVERWRITE)
  int rowsPerTile;  // mu = x: whole x rows per workgroup (R), and the chunk stride of their LDS image
  int rowChunk;
  // mu = x only: face layers of partitioned z / t axes written on the way through (the raw eigenvector is in registers between
  // its fetch and its rotation): what mugiq_hip_pack_face_layers would read once more, [n][layer][parity][12][faceCB]
  int nPack;
  struct Pack {
    Cplx<double> *base;
    int64_t vecStride;  // layers * 24 * faceCB
    int dim, high, layers, faceCB;
    int from;  // eigenvectors from .. nVec - 1 (the first halo block may have gone out ahead, packed by its own kernel)
  } pack[kMT_MaxPack];
};

// line `cid` of direction mu: parity and checkerboard index of its j = 0 site
__host__ __device__ inline void mt_line(int cid, int H, int strideMu, int J, int &p0, int &base) {
  const int colsPerParity = H * strideMu;
  p0 = cid / colsPerParity;
  const int rem = cid - p0 * colsPerParity;
  const int hi = rem / strideMu;
  const int lo = rem - hi * strideMu;
  base = hi * (J * strideMu) + lo;
}

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

#define MUGIQ_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f64_4x4x4f64(a_, b_, c_, 0, 0, 0)
// ... with the left operand negated by the instruction (for the f64 forms the BLGP field holds the NEG bits of A, B, C)
#define MUGIQ_MFMA_NEGA(a_, b_, c_) __builtin_amdgcn_mfma_f64_4x4x4f64(a_, b_, c_, 0, 0, 1)

// mu = x (DIR == 0, "row tile"): the lines run along the coalescing direction, so a workgroup owns R whole x rows (both
// parities; R X0 = 128 | 192 sites, TJ = 0 and LN = 16 * groups per wave in the template) and there is no halo at all: the
// positions past the end of the row (sign +) or before its start (sign -) are the row's own first / last sites, staged a
// second time with the continued gauge g(J + l) | g(-l).  LDS image: chunk (parity, component) = [row][X0/2 + 2] complex.
template <int DIR, int SIGN, int NS, int TJ, int LN, bool PACK = false, typename F = double, int ORDER = 2>
__global__ __launch_bounds__(64 * (DIR == 0 ? TJ : kMT_Waves), 4) void mfma_tile_displaced_contract_kernel(MTileArgs a) {
  // F, ORDER: the eigenvectors' storage (double | float; FLOAT2 | FLOAT4).  They are converted on their way into LDS; the tile images, the
  // gauge and the products are double whatever the storage, the slots double or float (a.outFloat)
  static_assert(!PACK || (std::is_same<F, double>::value && ORDER == 2), "face layers are written on the way for fp64 FLOAT2 only");
  // element k = 3 spin + colour of checkerboard entry x in a field body of stride `stride` (complex elements from the parity base)
  auto fieldOff = [](int k, int x, int stride) { return ORDER == 2 ? k * stride + x : (((k >> 1) * stride + x) << 1) + (k & 1); };  // 4 waves per SIMD: <= 128 VGPRs
  constexpr bool kRow = DIR == 0;
  constexpr int kWaves = kRow ? TJ : kMT_Waves;  // (row tile: the TJ slot of the template carries the waves per workgroup, 8 | 16)
  constexpr int kMT_TJ = TJ, kMT_Cols = LN;
  constexpr int kPPC = kRow ? 1 : 64 / LN;                          // positions per chunk (2 | 4)
  constexpr int kMT_Groups = kRow ? LN / 16 : TJ * LN / 4 / kMT_Waves;  // 4-site groups per wave
  constexpr int kGP = LN / 4;                                       // groups per position
  constexpr int kSites = kMT_Groups * 4 * kWaves;
  extern __shared__ __align__(16) unsigned char smem[];
  Cplx<double> *tileBase = reinterpret_cast<Cplx<double> *>(smem);  // 2 x [pair][12][68]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int J = a.X[DIR];
  const int NP = kMT_TJ + a.kmax;

  int blk = blockIdx.x;
  if (a.blockOrder & 2) {  // XCD-contiguous: workgroups are dealt round-robin over the 8 XCDs
    const int per = gridDim.x >> 3;
    blk = (blk & 7) * per + (blk >> 3);
  }
  const int jt = a.jtBegin + blk % a.jtCount;
  const int cc = blk / a.jtCount;
  const int j0 = jt * kMT_TJ;

  // ---- staging role: thread <-> (position t / (4 LN), spin, line t % LN); three colours each
  const int spp = LN == 32 ? wave >> 1 : wave, sspin = (t / LN) & 3, sline = t & (LN - 1);
  const bool stages = kRow || spp < NP;  // (wave-uniform: the waves of the unused positions only compute)

  int soff = 0, cstride = a.stride;
  unsigned sByte = 0;  // (row tile) byte offset of this thread's first colour in an eigenvector body
  int offC[3] = {0, 0, 0};  // (FLOAT4) element offsets of the three colours (FLOAT2: soff + c * cstride)
  bool fromGhost = false;
  Cplx<double> g[9];
#pragma unroll
  for (int c = 0; c < 9; c++) g[c] = Cplx<double>{0.0, 0.0};
  int wIdx = 0;
  const int compStride = kRow ? a.rowChunk : kMT_Chunk;  // distance of two components in the LDS image
  constexpr int bufElems = kWaves == 8 ? kMT_BufElems / 2 : kMT_BufElems;  // one tile buffer (two 8-wave workgroups share a CU's LDS)
  bool commits = stages;
  // (row tile, PACK) this thread's (component << 20) | entry within the (y, x) plane of a face, or -1; everything else of the pack
  // addressing is per workgroup (its rows share z and t) and is told to the compiler to be: scalar registers, scalar arithmetic
  int pkAB = -1;
  uint64_t pkBase[kMT_MaxPack];  // target i at (layer, other coordinate) of this workgroup's rows, 0 = its rows are not on that face
  if constexpr (PACK) {
    const int zt = (blk * a.rowsPerTile) / a.X[1], zz = zt % a.X[2], tt = zt / a.X[2];
#pragma unroll
    for (int i = 0; i < kMT_MaxPack; i++) {
      const int isZ = a.pack[i].dim == 2, coord = isZ ? zz : tt, other = isZ ? tt : zz;
      const int layer = a.pack[i].high ? a.X[isZ ? 2 : 3] - 1 - coord : coord;
      const bool member = i < a.nPack && layer < a.pack[i].layers;
      const uint64_t p = member ? reinterpret_cast<uint64_t>(a.pack[i].base + (int64_t)layer * 24 * a.pack[i].faceCB + (int64_t)other * a.X[1] * (a.X[0] >> 1)) : 0;
      pkBase[i] = ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(p >> 32)) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)p);
    }
  }
  // row tile: rows of X0/2 entries per parity, + 4 slots for the continued positions (element m <-> position 2 (m - off) + delta)
  const int EPR = a.X[0] >> 1, EPRX = EPR + kMT_MaxLength / 2, rOff = SIGN == MUGIQ_HIP_DISP_SIGN_PLUS ? 0 : kMT_MaxLength / 2;
  auto row_delta = [&](int rowG, int parity) {  // x of the first entry of (row, parity): (parity + y + z + t) & 1
    const int y = rowG % a.X[1], zt = rowG / a.X[1];
    return (parity + y + zt % a.X[2] + zt / a.X[2]) & 1;
  };
  if constexpr (kRow) {
    const int R = a.rowsPerTile, nItems = R * 8 * EPRX;
    const int q = t < nItems ? t : nItems - 1;
    const int mm = q % EPRX, rest = q / EPRX;
    const int spin = rest & 3, pr = rest >> 2, parity = pr / R, row = pr - parity * R;
    const int rowG = blk * R + row;
    const int j = 2 * (mm - rOff) + row_delta(rowG, parity);  // position along x; beyond [0, J): a continued one
    const bool valid = t < nItems && (SIGN == MUGIQ_HIP_DISP_SIGN_PLUS ? (j < J + a.kmax) : (j >= -a.kmax && j < J));
    const int jv = valid ? j : row_delta(rowG, parity);  // (invalid items fetch the first entry of their row and commit nothing)
    const int js = jv < 0 ? jv + J : (jv >= J ? jv - J : jv);  // J is even: the wrapped site has the same parity
    const int jext = (SIGN == MUGIQ_HIP_DISP_SIGN_PLUS) ? jv : jv + a.kmaxG;
    const int Jext = J + a.kmaxG;
    const int par = parity;
#pragma unroll
    for (int c = 0; c < 9; c++) g[c] = a.G[((int64_t)c * a.numCols + rowG) * Jext + jext];
    soff = (int)((int64_t)par * a.parity_offset + (int64_t)(3 * spin) * a.stride + (int64_t)rowG * EPR + (js >> 1));
#pragma unroll
    for (int c = 0; c < 3; c++) offC[c] = (int)((int64_t)par * a.parity_offset + fieldOff(3 * spin + c, rowG * EPR + (js >> 1), a.stride));
    wIdx = (parity * 12 + 3 * spin) * a.rowChunk + row * EPRX + mm;
    commits = valid;
    sByte = (unsigned)soff * (unsigned)sizeof(Cplx<F>);  // (< 2^32: mfma_tile_applicable)
    if constexpr (PACK) {  // the R rows of a workgroup share z and t (X1 % R == 0, checked by the launcher)
      const int m = mm - rOff;  // a real position of the row (not a continued one): this thread owns the site
      if (valid && m >= 0 && m < EPR) pkAB = ((parity * 12 + 3 * spin) << 20) | ((rowG % a.X[1]) * EPR + m);
    }
  } else if (stages) {
    int cid = cc * kMT_Cols + sline;
    if (cid >= a.numCols) cid = a.numCols - 1;  // surplus lines shadow the last one (valid addresses, result dropped)
    int p0, base;
    mt_line(cid, a.H, a.strideMu, J, p0, base);
    int j = (SIGN == MUGIQ_HIP_DISP_SIGN_PLUS) ? j0 + spp : j0 - a.kmax + spp;
    const int jext = (SIGN == MUGIQ_HIP_DISP_SIGN_PLUS) ? j : j + a.kmaxG;
    const int Jext = J + a.kmaxG;
#pragma unroll
    for (int c = 0; c < 9; c++) g[c] = a.G[((int64_t)c * Jext + jext) * a.numCols + cid];
    const int par = p0 ^ (j & 1);
    if ((j < 0 || j >= J) && a.partitioned) {
      int c0[4];
      get_coords(c0, base, a.X, p0);  // c0[DIR] == 0
      const int faceIdx = ghost_face_index_on_face(c0, a.X, DIR);
      const int layer = (j >= J) ? j - J : -j - 1;
      fromGhost = true;
      cstride = a.faceCB;
      soff = (int)((int64_t)layer * 24 * a.faceCB + (int64_t)par * 12 * a.faceCB + (int64_t)(3 * sspin) * a.faceCB + faceIdx);
#pragma unroll
      for (int c = 0; c < 3; c++) offC[c] = (int)((int64_t)layer * 24 * a.faceCB + (int64_t)par * 12 * a.faceCB + fieldOff(3 * sspin + c, faceIdx, a.faceCB));
    } else {
      j = j < 0 ? j + J : (j >= J ? j - J : j);
      soff = (int)((int64_t)par * a.parity_offset + (int64_t)(3 * sspin) * a.stride + base + j * a.strideMu);
#pragma unroll
      for (int c = 0; c < 3; c++) offC[c] = (int)((int64_t)par * a.parity_offset + fieldOff(3 * sspin + c, base + j * a.strideMu, a.stride));
    }
    wIdx = ((spp / kPPC) * 12 + 3 * sspin) * kMT_Chunk + (spp % kPPC) * kMT_Cols + sline;
  }
  const Cplx<F> *ghostBase = reinterpret_cast<const Cplx<F> *>(a.ghost);

  // ---- arithmetic role: lane = 16 hi + 4 b + lo; group wave * G + gi = (position, line quad), site b = line 4 quad + b
  const int lo = lane & 3, b = (lane >> 2) & 3, hi = lane >> 4;
  const int compRd = 3 * lo + (hi < 2 ? hi : 2);  // component 3 spin + colour (the padding lanes hi = 3 re-read colour 2)
  auto elemIdx = [&](int pp, int line) { return ((pp / kPPC) * 12 + compRd) * kMT_Chunk + (pp % kPPC) * kMT_Cols + line; };
  int vIdx[kMT_Groups], pIdx[kMT_Groups][kRow ? 1 : NS];  // (row tile: pIdx[gi][0] = the odd-length base, see below)
#pragma unroll
  for (int gi = 0; gi < kMT_Groups; gi++) {
    if constexpr (kRow) {  // group = 4 consecutive entries of one (parity, row)
      const int R = a.rowsPerTile, gid = wave * kMT_Groups + gi, gpr = EPR / 4;
      const int m0 = 4 * (gid % gpr), pr = gid / gpr, parity = pr / R, row = pr - parity * R;
      const int j = 2 * (m0 + b) + row_delta(blk * R + row, parity);
      // the shifted partner of length k: same parity plane and k / 2 entries on for an even k; the other plane and (k -+ 1) / 2 +
      // (x of the row's first entry) on for an odd one -- two bases per group and a per-slot constant instead of NS addresses
      vIdx[gi] = (parity * 12 + compRd) * a.rowChunk + row * EPRX + m0 + b + rOff;
      pIdx[gi][0] = ((parity ^ 1) * 12 + compRd) * a.rowChunk + row * EPRX + m0 + b + rOff + (j & 1);
      continue;
    }
    const int gid = wave * kMT_Groups + gi, gpos = gid / kGP, gline = 4 * (gid % kGP) + b;
    vIdx[gi] = elemIdx((SIGN == MUGIQ_HIP_DISP_SIGN_PLUS) ? gpos : a.kmax + gpos, gline);
#pragma unroll
    for (int s = 0; s < NS; s++) pIdx[gi][s] = elemIdx((SIGN == MUGIQ_HIP_DISP_SIGN_PLUS) ? gpos + a.k[s] : a.kmax + gpos - a.k[s], gline);
  }
  int rowOdd = 0, rowShift[NS];  // (row tile) bit s: length k[s] is odd; entries from the base to the partner of slot s
#pragma unroll
  for (int s = 0; s < NS; s++) {
    const int k = a.k[s];
    rowOdd |= (k & 1) << s;
    rowShift[s] = (SIGN == MUGIQ_HIP_DISP_SIGN_PLUS) ? ((k & 1) ? (k - 1) / 2 : k / 2) : ((k & 1) ? -(k + 1) / 2 : -k / 2);
  }
  // the padded colour (lanes hi == 3) of the LEFT operand is zero: those lanes read a cell behind the tile buffers that holds 0 (the
  // right operand may hold anything finite there)
  constexpr int zeroCell = 2 * bufElems > 16 * kSites ? 2 * bufElems : 16 * kSites;  // (the launcher allocates one cell more)
  if (t == 0) tileBase[zeroCell] = Cplx<double>{0.0, 0.0};
  if constexpr (!kRow) {  // (row tile: vIdx is also the base of the even-length partners, which must stay real data -- a padded lane of the
                          //  RIGHT operand multiplies zeros but must be finite; the left operand's address is chosen at the read)
#pragma unroll
    for (int gi = 0; gi < kMT_Groups; gi++)
      if (hi == 3) vIdx[gi] = zeroCell;
  }

  double aR[kMT_Groups][NS], aI[kMT_Groups][NS];
#pragma unroll
  for (int gi = 0; gi < kMT_Groups; gi++)
#pragma unroll
    for (int s = 0; s < NS; s++) aR[gi][s] = aI[gi][s] = 0.0;

  typedef double vec2 __attribute__((ext_vector_type(2)));
  typedef F vecF __attribute__((ext_vector_type(2)));  // one complex number of the storage type
  vecF stageA[3], stageB[3];
#define MUGIQ_MT_BODY(n_) static_cast<const Cplx<F> *>(as_constant(a.L)[n_])
#define MUGIQ_MT_SIGMA(n_) as_constant(a.inv_sigma)[n_]
  // this thread's three colours of eigenvector n_ (unconditional for the staging waves: a known number of loads in flight)
#define MUGIQ_MT_FETCH(bodyExpr_, n_, stage)                                                                           \
  {                                                                                                                    \
    if constexpr (kRow && ORDER == 2) { /* scalar base + one 32-bit byte offset per lane: no 64-bit address arithmetic per lane */ \
      const char *b_ = reinterpret_cast<const char *>(bodyExpr_);                                                      \
      _Pragma("unroll") for (int c = 0; c < 3; c++) stage[c] = *as_global(reinterpret_cast<const vecF *>(b_ + (int64_t)c * a.stride * (int64_t)sizeof(Cplx<F>) + (uint64_t)sByte)); \
    } else if constexpr (kRow) { /* (FLOAT4: the colours of a spin are not a stride apart) */                          \
      const Cplx<F> *b_ = (bodyExpr_);                                                                                 \
      _Pragma("unroll") for (int c = 0; c < 3; c++) stage[c] = *as_global(reinterpret_cast<const vecF *>(b_ + offC[c])); \
    } else if (stages) {                                                                                               \
      const Cplx<F> *base_ = fromGhost ? ghostBase + (int64_t)(n_)*a.ghost_vec_stride : (bodyExpr_);                   \
      if constexpr (ORDER == 2) {                                                                                      \
        _Pragma("unroll") for (int c = 0; c < 3; c++) stage[c] = *as_global(reinterpret_cast<const vecF *>(base_ + soff + (int64_t)c * cstride)); \
      } else {                                                                                                         \
        _Pragma("unroll") for (int c = 0; c < 3; c++) stage[c] = *as_global(reinterpret_cast<const vecF *>(base_ + offC[c])); \
      }                                                                                                                \
    }                                                                                                                  \
  }
  // v' = g v into tile buffer buf_
#define MUGIQ_MT_COMMIT(stage, buf_, n_)                                                                               \
  {                                                                                                                    \
    if (commits) {                                                                                                     \
      Cplx<double> *dst_ = (buf_) + wIdx;                                                                              \
      _Pragma("unroll") for (int i = 0; i < 3; i++) {                                                                  \
        Cplx<double> r{0.0, 0.0};                                                                                      \
        _Pragma("unroll") for (int j = 0; j < 3; j++) cmadd(r, g[i * 3 + j], Cplx<double>{(double)stage[j].x, (double)stage[j].y}); \
        dst_[i * compStride] = r;                                                                                      \
      }                                                                                                                \
    }                                                                                                                  \
    if constexpr (PACK) { /* the raw eigenvector n_ onto the face-layer targets this workgroup's rows lie on */         \
      _Pragma("unroll") for (int i = 0; i < kMT_MaxPack; i++) {                                                        \
        if (pkBase[i] != 0) { /* (wave-uniform.  The target's numbers come from the kernel arguments HERE, behind an index the \
                                 compiler cannot see through: hoisted out of the eigenvector loop they cost 30 scalar registers \
                                 and the kernel spills) */                                                             \
          int o_ = 0;                                                                                                  \
          asm volatile("" : "+s"(o_));                                                                                 \
          const MTileArgs::Pack &t_ = a.pack[i + o_];                                                                  \
          if ((n_) < t_.from || (n_) >= a.nVec) continue;                                                              \
          char *pk_ = reinterpret_cast<char *>(pkBase[i]) + (int64_t)(n_)*t_.vecStride * 16;                           \
          if (pkAB >= 0) {                                                                                             \
            const unsigned f_ = (unsigned)t_.faceCB, v_ = (unsigned)(pkAB >> 20) * f_ + (unsigned)(pkAB & 0xfffff);    \
            _Pragma("unroll") for (int c = 0; c < 3; c++) *as_global(reinterpret_cast<vec2 *>(pk_ + (uint64_t)((v_ + (unsigned)c * f_) * 16u))) = stage[c]; \
          }                                                                                                            \
        }                                                                                                              \
      }                                                                                                                \
    }                                                                                                                  \
  }
#define MUGIQ_MT_BARRIER()                              \
  {                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    __builtin_amdgcn_s_barrier();                       \
    asm volatile("" ::: "memory");                      \
  }
  // the products of one eigenvector (scaled by s_) on the tile buffer tile_
#define MUGIQ_MT_COMPUTE(tile_, s_)                                                                                    \
  {                                                                                                                    \
    const Cplx<double> *tile = tile_;                                                                                  \
    const double sc = (s_);                                                                                            \
    _Pragma("unroll") for (int gi = 0; gi < kMT_Groups; gi++) {                                                        \
      int zc_ = zeroCell - (int)(tile - tileBase);                                                                     \
      if constexpr (kRow) asm volatile("" : "+s"(zc_)); /* (the select below stays inside the loop: no register for it) */ \
      const Cplx<double> v = tile[(kRow && hi == 3) ? zc_ : vIdx[gi]];                                                 \
      const double VR = sc * v.re, VI = sc * v.im;                                                                     \
      _Pragma("unroll") for (int s = 0; s < NS; s++) {                                                                 \
        int sh_ = kRow ? rowShift[s] : 0, odd_ = kRow ? (rowOdd >> s) & 1 : 0;                                         \
        if constexpr (kRow) asm volatile("" : "+s"(sh_), "+s"(odd_)); /* (keeps the NS x groups sums out of registers) */ \
        const Cplx<double> p = tile[kRow ? (odd_ ? pIdx[gi][0] : vIdx[gi]) + sh_ : pIdx[gi][kRow ? 0 : s]];            \
        aR[gi][s] = MUGIQ_MFMA(VR, p.re, aR[gi][s]);                                                                   \
        aI[gi][s] = MUGIQ_MFMA(VR, p.im, aI[gi][s]);                                                                   \
        aR[gi][s] = MUGIQ_MFMA(VI, p.im, aR[gi][s]);                                                                   \
        aI[gi][s] = MUGIQ_MFMA_NEGA(VI, p.re, aI[gi][s]);                                                              \
      }                                                                                                                \
    }                                                                                                                  \
  }
  // One step: v'(n_) is in buffer n_ & 1; `stage` holds the raw eigenvector n_ + 1 (fetched two steps ago).  Rotate and commit
  // it into the other buffer (everybody finished reading that one before the barrier that ended the previous step), refill
  // `stage` with n_ + 3, consume n_, one barrier.
#define MUGIQ_MT_STEP(n_, stage, GUARD)                                                                                \
  {                                                                                                                    \
    const double sNow = sigPre;                                                                                        \
    const Cplx<F> *bodyNow = bodyPre;                                                                                  \
    {                                                                                                                  \
      const int nb_ = (n_) + 4 < a.nVec ? (n_) + 4 : a.nVec - 1, ns_ = (n_) + 1 < a.nVec ? (n_) + 1 : a.nVec - 1;      \
      bodyPre = MUGIQ_MT_BODY(nb_);                                                                                    \
      sigPre = MUGIQ_MT_SIGMA(ns_);                                                                                    \
    }                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    if (GUARD == 0 || (n_) + 1 < a.nVec) MUGIQ_MT_COMMIT(stage, tileBase + (size_t)(((n_) + 1) & 1) * bufElems, (n_) + 1)        \
    if (GUARD == 0 || (n_) + 3 < a.nVec) MUGIQ_MT_FETCH(bodyNow, (n_) + 3, stage)                                      \
    MUGIQ_MT_COMPUTE(tileBase + (size_t)((n_) & 1) * bufElems, sNow)                                                   \
    MUGIQ_MT_BARRIER()                                                                                                 \
  }
  // prologue: eigenvector 0 -> buffer 0; eigenvectors 1 and 2 in flight (clamped, unconditional)
  {
    const int last = a.nVec - 1;
    MUGIQ_MT_FETCH(MUGIQ_MT_BODY(0), 0, stageB)
    MUGIQ_MT_COMMIT(stageB, tileBase, 0)
    MUGIQ_MT_FETCH(MUGIQ_MT_BODY((1 < last ? 1 : last)), (1 < last ? 1 : last), stageA)
    MUGIQ_MT_FETCH(MUGIQ_MT_BODY((2 < last ? 2 : last)), (2 < last ? 2 : last), stageB)
  }
  const Cplx<F> *bodyPre = MUGIQ_MT_BODY(a.nVec > 3 ? 3 : a.nVec - 1);
  double sigPre = MUGIQ_MT_SIGMA(0);
  MUGIQ_MT_BARRIER()
  int n = 0;
  for (; n + 4 < a.nVec; n += 2) {
    MUGIQ_MT_STEP(n, stageA, 0)
    MUGIQ_MT_STEP(n + 1, stageB, 0)
  }
  for (; n < a.nVec; n += 2) {
    MUGIQ_MT_STEP(n, stageA, 1)
    if (n + 1 < a.nVec) MUGIQ_MT_STEP(n + 1, stageB, 1)
  }
#undef MUGIQ_MT_STEP
#undef MUGIQ_MT_COMPUTE
#undef MUGIQ_MT_COMMIT
#undef MUGIQ_MT_FETCH
#undef MUGIQ_MT_BODY
#undef MUGIQ_MT_SIGMA

  // ---- epilogue: lane 16 be + 4 b + al holds element (be, al) of the spin matrix of site b.  Slot by slot through LDS (the
  // tile buffers are free now) as [be * 4 + al][site = position * LN + line], then one thread per (site, half of the gamma
  // channels): consecutive lanes <-> consecutive lines, so the stores stay coalesced per channel.
  Cplx<double> *scratch = tileBase;
#pragma unroll
  for (int s = 0; s < NS; s++) {
#pragma unroll
    for (int gi = 0; gi < kMT_Groups; gi++) {
      const int gid = wave * kMT_Groups + gi;
      // (row tile: site = (parity R + row) X0/2 + entry = 4 gid + b)
      const int site = kRow ? 4 * gid + b : (gid / kGP) * kMT_Cols + 4 * (gid % kGP) + b;
      scratch[(hi * 4 + lo) * kSites + site] = Cplx<double>{aR[gi][s], aI[gi][s]};
    }
    MUGIQ_MT_BARRIER()
    for (int item = t; item < 2 * kSites; item += 64 * kWaves) {
      const int half = item / kSites, site = item - half * kSites;
      int pmine, xmine;
      if constexpr (kRow) {
        const int R = a.rowsPerTile, pr = site / EPR;
        pmine = pr / R;
        xmine = (blk * R + pr - pmine * R) * EPR + site - pr * EPR;
      } else {
        const int pos = site / kMT_Cols;
        const int cid = cc * kMT_Cols + (site % kMT_Cols);
        if (cid >= a.numCols) continue;
        int p0, base;
        mt_line(cid, a.H, a.strideMu, J, p0, base);
        const int jmine = j0 + pos;
        pmine = p0 ^ (jmine & 1);
        xmine = base + jmine * a.strideMu;
      }
      Cplx<double> full[16];
#pragma unroll
      for (int e = 0; e < 16; e++) full[e] = scratch[e * kSites + site];
      const int siteIdx = xmine + pmine * a.volumeCB;
      if (std::is_same<F, double>::value || !a.outFloat) {  // (fp64 eigenvectors come with fp64 slots: no second store path in those kernels)
        Cplx<double> *o = static_cast<Cplx<double> *>(a.out[s]);
        if (half == 0) trace_and_store_range<double, 0, 8>(o, full, 2 * a.volumeCB, siteIdx, a.overwrite != 0);
        else trace_and_store_range<double, 8, 16>(o, full, 2 * a.volumeCB, siteIdx, a.overwrite != 0);
      } else {  // fp32 slots: the traces are taken in double and rounded once, on the way out
        Cplx<double> tr[4];
        Cplx<float> *o = static_cast<Cplx<float> *>(a.out[s]) + siteIdx;
#pragma unroll
        for (int q = 0; q < 2; q++) {
          if (half == 0) {
            if (q == 0) traces_range<double, 0>(tr, full);
            else traces_range<double, 4>(tr, full);
          } else {
            if (q == 0) traces_range<double, 8>(tr, full);
            else traces_range<double, 12>(tr, full);
          }
#pragma unroll
          for (int i = 0; i < 4; i++) {
            Cplx<float> *w = o + (int64_t)(2 * a.volumeCB) * (8 * half + 4 * q + i);
            Cplx<float> v = a.overwrite ? Cplx<float>{0.f, 0.f} : *w;
            v.re += (float)tr[i].re;
            v.im += (float)tr[i].im;
            *w = v;
          }
        }
      }
    }
    if (s + 1 < NS) MUGIQ_MT_BARRIER()
  }
#undef MUGIQ_MT_BARRIER
}
#undef MUGIQ_MFMA
#undef MUGIQ_MFMA_NEGA

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
      if (24 * ((r * (epr + kMT_MaxLength / 2) + 11) / 16 * 16 + 4) > (w == 8 ? kMT_BufElems / 2 : kMT_BufElems)) continue;  // the LDS image of a tile buffer
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

// F, ORDER: the eigenvectors' storage.  FULL: every tile geometry and the face-layer packing (fp64 FLOAT2); else the 16-line column tiles
// and the 8-wave row tile only (a sixth of the instances per storage type).
template <typename F, int ORDER, bool FULL>
static int launch_mfma_tile_t(MTileArgs a, int dir, int sign, int ns, int tj, int rowGroups, int rowWaves, hipStream_t stream) {
  const int ln = tj == 4 ? 32 : 16;
  const size_t bufElems = dir == 0 && rowWaves == 8 ? kMT_BufElems / 2 : kMT_BufElems;
  const size_t shmem = (std::max(2 * bufElems, (size_t)16 * (dir == 0 ? 4 * rowGroups * rowWaves : tj * ln)) + 1) * sizeof(Cplx<double>);  // (+ the zero cell)
  const unsigned nblocks = dir == 0 ? (unsigned)(a.numCols / a.rowsPerTile) : (unsigned)(((a.numCols + ln - 1) / ln) * a.jtCount);
  a.blockOrder = 2;
  if (const char *e = getenv("MUGIQ_HIP_TILE_ORDER")) a.blockOrder = atoi(e) & 2;
  if (nblocks % 8 != 0 || dir == 0) a.blockOrder = 0;
  const dim3 grid(nblocks), block(64 * (dir == 0 ? rowWaves : kMT_Waves));
  if (!FULL) {
    MUGIQ_REQUIRE(a.nPack == 0 && tj != 4 && (dir != 0 || rowWaves == 8), "mfma tile: geometry %d / %d waves / %d pack targets not built for this storage type (internal)", tj, rowWaves, a.nPack);
  }
#define MUGIQ_MT_ROW(S, N)                                                                                             \
  {                                                                                                                    \
    if (rowWaves == 8) {                                                                                               \
      if (rowGroups == 3) MUGIQ_MT_LAUNCH_P(0, S, N, 8, 48) else MUGIQ_MT_LAUNCH_P(0, S, N, 8, 32)                     \
    } else if constexpr (FULL) {                                                                                       \
      if (rowGroups == 3) MUGIQ_MT_LAUNCH_P(0, S, N, 16, 48) else MUGIQ_MT_LAUNCH_P(0, S, N, 16, 32)                   \
    }                                                                                                                  \
  }
#define MUGIQ_MT_ROWCASE(S)                                                                                            \
  case (S):                                                                                                            \
    if (ns == 1) MUGIQ_MT_ROW(S, 1) else if (ns == 2) MUGIQ_MT_ROW(S, 2) else MUGIQ_MT_ROW(S, 3)                       \
    break;
#define MUGIQ_MT_LAUNCH(D, S, N)                                                                                       \
  {                                                                                                                    \
    if (tj == 12) MUGIQ_MT_LAUNCH_(D, S, N, 12, 16) else if (tj == 8) MUGIQ_MT_LAUNCH_(D, S, N, 8, 16) else if constexpr (FULL) MUGIQ_MT_LAUNCH_(D, S, N, 4, 32) \
  }
#define MUGIQ_MT_LAUNCH_(D, S, N, T, LL)                                                                               \
  {                                                                                                                    \
    auto kern = mfma_tile_displaced_contract_kernel<D, S, N, T, LL, false, F, ORDER>;                                  \
    MUGIQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
    hipLaunchKernelGGL(kern, grid, block, shmem, stream, a);                                                           \
  }
#define MUGIQ_MT_LAUNCH_P(D, S, N, T, LL)                                                                              \
  {                                                                                                                    \
    auto kern = mfma_tile_displaced_contract_kernel<D, S, N, T, LL, false, F, ORDER>;                                  \
    if constexpr (FULL) {                                                                                              \
      if (a.nPack > 0) kern = mfma_tile_displaced_contract_kernel<D, S, N, T, LL, FULL, F, ORDER>;                     \
    }                                                                                                                  \
    MUGIQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
    hipLaunchKernelGGL(kern, grid, block, shmem, stream, a);                                                           \
  }
#define MUGIQ_MT_CASE(D, S)                                                                                            \
  case (D)*2 + (S):                                                                                                    \
    if (ns == 1) MUGIQ_MT_LAUNCH(D, S, 1) else if (ns == 2) MUGIQ_MT_LAUNCH(D, S, 2) else if (ns == 3) MUGIQ_MT_LAUNCH(D, S, 3) else MUGIQ_MT_LAUNCH(D, S, 4) \
    break;
  switch (dir * 2 + sign) {
    MUGIQ_MT_ROWCASE(0) MUGIQ_MT_ROWCASE(1)
    MUGIQ_MT_CASE(1, 0) MUGIQ_MT_CASE(1, 1) MUGIQ_MT_CASE(2, 0) MUGIQ_MT_CASE(2, 1) MUGIQ_MT_CASE(3, 0) MUGIQ_MT_CASE(3, 1)
  default:
    return set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "mfma tile: direction %d has no matrix-pipe tile (internal)", dir);
  }
#undef MUGIQ_MT_CASE
#undef MUGIQ_MT_LAUNCH
#undef MUGIQ_MT_LAUNCH_
#undef MUGIQ_MT_LAUNCH_P
#undef MUGIQ_MT_ROW
#undef MUGIQ_MT_ROWCASE
  MUGIQ_CHECK_HIP(hipGetLastError());
  return MUGIQ_HIP_SUCCESS;
}

static int launch_mfma_tile(const MTileArgs &a, int precision, int order, int dir, int sign, int ns, int tj, int rowGroups, int rowWaves, hipStream_t stream) {
  if (precision == 8 && order == 2) return launch_mfma_tile_t<double, 2, true>(a, dir, sign, ns, tj, rowGroups, rowWaves, stream);
  if (precision == 8) return launch_mfma_tile_t<double, 4, false>(a, dir, sign, ns, tj, rowGroups, rowWaves, stream);
  if (order == 2) return launch_mfma_tile_t<float, 2, false>(a, dir, sign, ns, tj, rowGroups, rowWaves, stream);
  return launch_mfma_tile_t<float, 4, false>(a, dir, sign, ns, tj, rowGroups, rowWaves, stream);
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
    a.rowChunk = (a.rowsPerTile * (ev[0].X[0] / 2 + kMT_MaxLength / 2) + 11) / 16 * 16 + 4;  // >= R (X0/2 + 4), and 4 mod 16 entries: 16 banks of phase per component
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
