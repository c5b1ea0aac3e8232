// The matrix-pipe tile of csrc/fused_mfma.hip as a header: the kernel template and its launcher, instantiated per storage type of the
// eigenvectors in translation units of their own (fused_mfma.hip: fp64 FLOAT2, every geometry; fused_mfma_d4 / _f2 / _f4.hip: the reduced
// sets), so that the 300 instances compile side by side.  Design and measurements: csrc/fused_mfma.hip, profiles/r04_mfma_tile.txt.
#pragma once
#include "internal.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

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
  int vIdx[kMT_Groups], pIdx[kMT_Groups][NS];
#pragma unroll
  for (int gi = 0; gi < kMT_Groups; gi++) {
    if constexpr (kRow) {  // group = 4 consecutive entries of one (parity, row)
      const int R = a.rowsPerTile, gid = wave * kMT_Groups + gi, gpr = EPR / 4;
      const int m0 = 4 * (gid % gpr), pr = gid / gpr, parity = pr / R, row = pr - parity * R;
      const int j = 2 * (m0 + b) + row_delta(blk * R + row, parity);
      // the shifted partner of length k: the same parity plane for an even k, the other one for an odd k
      vIdx[gi] = (parity * 12 + compRd) * a.rowChunk + row * EPRX + m0 + b + rOff;
#pragma unroll
      for (int s = 0; s < NS; s++) {
        const int js = (SIGN == MUGIQ_HIP_DISP_SIGN_PLUS) ? j + a.k[s] : j - a.k[s];
        pIdx[gi][s] = ((parity ^ (a.k[s] & 1)) * 12 + compRd) * a.rowChunk + row * EPRX + ((js + 2 * rOff) >> 1);
      }
      continue;
    }
    const int gid = wave * kMT_Groups + gi, gpos = gid / kGP, gline = 4 * (gid % kGP) + b;
    vIdx[gi] = elemIdx((SIGN == MUGIQ_HIP_DISP_SIGN_PLUS) ? gpos : a.kmax + gpos, gline);
#pragma unroll
    for (int s = 0; s < NS; s++) pIdx[gi][s] = elemIdx((SIGN == MUGIQ_HIP_DISP_SIGN_PLUS) ? gpos + a.k[s] : a.kmax + gpos - a.k[s], gline);
  }
  // the padded colour (lanes hi == 3) of the LEFT operand is zero: those lanes read a padding entry of the first chunk of the tile
  // image -- no commit ever writes there -- that holds 0 in BOTH buffers (the right operand may hold anything finite in its padded
  // lanes: its addresses point at real data)
  const int zeroCell = kRow ? a.rowChunk - 1 : 64;
  if (t < 2) tileBase[(size_t)t * bufElems + zeroCell] = Cplx<double>{0.0, 0.0};
#pragma unroll
  for (int gi = 0; gi < kMT_Groups; gi++)
    if (hi == 3) vIdx[gi] = zeroCell;

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
      const Cplx<double> v = tile[vIdx[gi]];                                                                           \
      const double VR = sc * v.re, VI = sc * v.im;                                                                     \
      _Pragma("unroll") for (int s = 0; s < NS; s++) {                                                                 \
        const Cplx<double> p = tile[pIdx[gi][s]];                                                                      \
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

// F, ORDER: the eigenvectors' storage.  FULL: every tile geometry and the face-layer packing (fp64 FLOAT2); else the 16-line column tiles
// and the 8-wave row tile only (a sixth of the instances per storage type).
template <typename F, int ORDER, bool FULL>
inline int launch_mfma_tile_t(MTileArgs a, int dir, int sign, int ns, int tj, int rowGroups, int rowWaves, hipStream_t stream) {
  const int ln = tj == 4 ? 32 : 16;
  const size_t bufElems = dir == 0 && rowWaves == 8 ? kMT_BufElems / 2 : kMT_BufElems;
  const size_t shmem = std::max(2 * bufElems, (size_t)16 * (dir == 0 ? 4 * rowGroups * rowWaves : tj * ln)) * sizeof(Cplx<double>);
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

// the reduced sets (fused_mfma_d4.hip, fused_mfma_f2.hip, fused_mfma_f4.hip)
int launch_mfma_tile_d4(const MTileArgs &a, int dir, int sign, int ns, int tj, int rowGroups, int rowWaves, hipStream_t stream);
int launch_mfma_tile_f2(const MTileArgs &a, int dir, int sign, int ns, int tj, int rowGroups, int rowWaves, hipStream_t stream);
int launch_mfma_tile_f4(const MTileArgs &a, int dir, int sign, int ns, int tj, int rowGroups, int rowWaves, hipStream_t stream);

}  // namespace mugiq
