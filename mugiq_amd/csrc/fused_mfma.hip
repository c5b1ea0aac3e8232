// Fused displaced contraction on the fp64 matrix pipe (fourth generation; fp64 FLOAT2 column tiles, mu = y, z, t).
//
// Same tile as csrc/fused_tile.hip -- a workgroup owns 32 lines along mu x 4 consecutive positions and stages the 4 + Kmax
// positions it needs global -> LDS into three rotating buffers, one barrier per eigenvector -- but the arithmetic runs as
// chained v_mfma_f64_4x4x4_4b_f64: four independent 4x4x4 products per instruction, ONE LATTICE SITE PER BLOCK.
//
//   lane maps of the instruction (tools/probes/mfma_4x4x4_layout.hip, one-hot operands):
//     A[b][i][k] in lane 16 k + 4 b + i,   B[b][k][j] in lane 16 k + 4 b + j,   D[b][i][j] in lane 16 i + 4 b + j
//   i.e. a result D is, as it stands, the B operand of the next product with its row index as the summation index.
//
//   per site, slot and eigenvector (real 4x4 tiles; colour index padded 3 -> 4, W has zeros there):
//     stage 1   tR[i][al] = WR[i][j] psiR[j][al] - WI[i][j] psiI[j][al]        t = W_k(x) psi(x + k mu), 4 products
//               tI[i][al] = WI[i][j] psiR[j][al] + WR[i][j] psiI[j][al]
//     stage 2   accR[be][al] += VR[be][i] tR[i][al] + VI[be][i] tI[i][al]      acc += conj(v(x)) (x) t, 4 products
//               accI[be][al] += VR[be][i] tI[i][al] - VI[be][i] tR[i][al]      (V = v(x) / sigma_n)
//   W sits in registers for the whole kernel, psi and v come from LDS with ONE ds_read_b128 per lane and 4-site group
//   each (v is shared by the slots), the 4x4 colour-traced spin matrices accumulate in the D registers (2 x 2 VGPRs per
//   group and slot).  8 products of 128 flops do the 672 flops of the mathematics (66 %), but the vector pipe issues
//   nothing but two multiplies per group: 96 MFMAs against 16 LDS reads and 8 VALU instructions per wave and eigenvector,
//   where the vector form of csrc/fused_tile.hip issues 205 VALU instructions and 18 LDS reads per wave for a third of
//   the sites.  Measured: profiles/r04_mfma_tile.txt.
//
// LDS image: chunk (position pair, component) = [position & 1][32 lines] complex = 1 KiB (one global_load_lds
// instruction), chunks 64 bytes apart from each other's bank phase (stride 1088 B): an operand read -- lanes (k | i, site,
// spin) -> component 3 spin + colour -- then touches every bank once (no conflicts; see the bank arithmetic in DESIGN.md).
#include "internal.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace mugiq {

constexpr int kMT_TJ = 4;         // positions along mu per workgroup
constexpr int kMT_Cols = 32;      // lines per workgroup
// wave w: position w & 3, lines 4 G (w >> 2) .. + 4 G - 1 (G groups of 4 sites): G = 4 -> 8 waves, G = 2 -> 16 waves
constexpr int kMT_MaxSlots = 4;   // 3 displaced slots + the ultra-local loop riding along (k = 0, W = 1)
constexpr int kMT_Pairs = 4;      // staged positions (TJ + Kmax <= 8) in pairs
constexpr int kMT_Chunk = 68;     // complex elements per chunk: 64 + 4 of bank phase
constexpr int kMT_BufElems = kMT_Pairs * 12 * kMT_Chunk;

struct MTileArgs {
  Cplx<double> *out[kMT_MaxSlots];
  const void *const *L;
  const double *inv_sigma;
  int nVec;
  int X[4];
  int volumeCB;
  int stride;
  int64_t parity_offset;
  const double *E[kMT_MaxSlots];  // path links W_k of every slot; NULL = the identity (k = 0)
  int k[kMT_MaxSlots];
  int kmax;
  int partitioned;
  const double *ghost;
  int64_t ghost_vec_stride;
  int faceCB;
  int strideMu;   // x_cb distance of one step along DIR
  int H;          // volumeCB / (X[DIR] * strideMu)
  int numCols;    // V / X[DIR]
  int jtBegin;    // tiles along mu handled by this launch: [jtBegin, jtBegin + jtCount)
  int jtCount;
  int blockOrder; // bit 1: XCD-contiguous workgroup order
  int overwrite;  // store instead of accumulate (MUGIQ_HIP_REGION_OVERWRITE)
};

#define MUGIQ_MFMA(a_, b_, c_) __builtin_amdgcn_mfma_f64_4x4x4f64(a_, b_, c_, 0, 0, 0)
#ifndef MUGIQ_MT_EXPERIMENT
#define MUGIQ_MT_EXPERIMENT 0  // probe builds only (tools/probes): 1 no global loads in the steps, 2 + no barrier, 3 + no LDS reads
#endif

template <int DIR, int SIGN, int NS, int G, bool HOIST>
__global__ __launch_bounds__(64 * 32 / G) void mfma_tile_displaced_contract_kernel(MTileArgs a) {
  constexpr int kMT_Groups = G, kMT_Waves = 32 / G;
  constexpr int kMT_PerWave = kMT_Pairs * 12 / kMT_Waves;  // chunks a wave stages per eigenvector (6 | 3)
  extern __shared__ __align__(16) unsigned char smem[];
  Cplx<double> *tileBase = reinterpret_cast<Cplx<double> *>(smem);  // 3 x [pair][12][68]
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
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

  // a line of the tile: parity of its j = 0 site, x_cb of that site, index on the face (ghost layers)
  auto line_info = [&](int c32, int &p0, int &base, int &faceIdx, bool &ok) {
    int cid = cc * kMT_Cols + c32;
    ok = cid < a.numCols;
    if (!ok) cid = a.numCols - 1;  // surplus lines shadow the last one (valid addresses, result dropped)
    const int colsPerParity = a.H * a.strideMu;
    p0 = cid / colsPerParity;
    const int rem = cid - p0 * colsPerParity;
    const int hi = rem / a.strideMu;
    const int lo = rem - hi * a.strideMu;
    base = hi * (J * a.strideMu) + lo;
    int c0[4];
    get_coords(c0, base, a.X, p0);  // c0[DIR] == 0
    faceIdx = ghost_face_index_on_face(c0, a.X, DIR);
  };

  // ---- staging: chunk q = wave * 6 + i <-> (pair q / 12, component q % 12); lane -> position 2 pair + (lane >> 5), line lane & 31
  const Cplx<double> *ghostBase = reinterpret_cast<const Cplx<double> *>(a.ghost);
  int soff[kMT_PerWave];
  unsigned sghost = 0;
  {
    int p0, base, faceIdx;
    bool ok;
    line_info(lane & 31, p0, base, faceIdx, ok);
#pragma unroll
    for (int i = 0; i < kMT_PerWave; i++) {
      const int q = wave * kMT_PerWave + i;
      const int pair = q / 12, comp = q - pair * 12;
      int pp = 2 * pair + (lane >> 5);
      pp = pp < NP ? pp : NP - 1;  // surplus positions re-read the last one
      int j = (SIGN == MUGIQ_HIP_DISP_SIGN_PLUS) ? j0 + pp : j0 - a.kmax + pp;
      const int par = p0 ^ (j & 1);
      if ((j < 0 || j >= J) && a.partitioned) {
        const int layer = (j >= J) ? j - J : -j - 1;
        sghost |= 1u << i;
        soff[i] = (int)((int64_t)layer * 24 * a.faceCB + (int64_t)par * 12 * a.faceCB + (int64_t)comp * a.faceCB + faceIdx);
      } else {
        j = j < 0 ? j + J : (j >= J ? j - J : j);
        soff[i] = (int)((int64_t)par * a.parity_offset + (int64_t)comp * a.stride + base + j * a.strideMu);
      }
    }
  }

  // ---- arithmetic: lane = 16 hi + 4 b + lo; site b of group g = line 16 (wave >> 2) + 4 g + b at position wave & 3
  const int lo = lane & 3, b = (lane >> 2) & 3, hi = lane >> 4;
  const int wpos = wave & 3, lineBase = 4 * G * (wave >> 2) + b;
  const int ppL = (SIGN == MUGIQ_HIP_DISP_SIGN_PLUS) ? wpos : a.kmax + wpos;
  const int compRd = 3 * lo + (hi < 2 ? hi : 2);  // component 3 spin + colour (the padding lanes hi = 3 re-read colour 2)
  auto elemIdx = [&](int pp) { return ((pp >> 1) * 12 + compRd) * kMT_Chunk + (pp & 1) * kMT_Cols + lineBase; };
  const int vIdx = elemIdx(ppL);
  int pIdx[NS];
#pragma unroll
  for (int s = 0; s < NS; s++) pIdx[s] = elemIdx((SIGN == MUGIQ_HIP_DISP_SIGN_PLUS) ? wpos + a.k[s] : a.kmax + wpos - a.k[s]);

  // W_k(x) of every (group, slot) as the A operand of stage 1: lane (k = hi, site b, i = lo) holds W[i = lo][j = hi]
  double WR[kMT_Groups][NS], WI[kMT_Groups][NS], nWI[kMT_Groups][NS];
#pragma unroll
  for (int g = 0; g < kMT_Groups; g++) {
    int p0, base, faceIdx;
    bool ok;
    line_info(lineBase + 4 * g, p0, base, faceIdx, ok);
    const int jmine = j0 + wpos;
    const int pmine = p0 ^ (jmine & 1), xmine = base + jmine * a.strideMu;
#pragma unroll
    for (int s = 0; s < NS; s++) {
      Cplx<double> w{0.0, 0.0};
      if (lo < 3 && hi < 3) {
        if (a.E[s]) w = reinterpret_cast<const Cplx<double> *>(a.E[s])[(int64_t)pmine * 12 * a.volumeCB + (int64_t)(hi * 3 + lo) * a.volumeCB + xmine];
        else w = Cplx<double>{lo == hi ? 1.0 : 0.0, 0.0};  // the carried ultra-local slot
      }
      WR[g][s] = w.re;
      WI[g][s] = w.im;
      nWI[g][s] = -w.im;
    }
  }
  double aR[kMT_Groups][NS], aI[kMT_Groups][NS];
#pragma unroll
  for (int g = 0; g < kMT_Groups; g++)
#pragma unroll
    for (int s = 0; s < NS; s++) aR[g][s] = aI[g][s] = 0.0;

#if defined(__HIP_DEVICE_COMPILE__)  // (global_load_lds is a device-only builtin: the host pass of hipcc must not see it)
  typedef double vec2 __attribute__((ext_vector_type(2)));
  typedef __attribute__((address_space(3))) void lds_void;
#define MUGIQ_MT_BODY(n_) static_cast<const Cplx<double> *>(as_constant(a.L)[n_])
#define MUGIQ_MT_SIGMA(n_) as_constant(a.inv_sigma)[n_]
  // this wave's share of eigenvector n_ -> tile buffer buf_: 6 transfers of 64 x 16 bytes
#define MUGIQ_MT_GLDS(bodyExpr_, n_, buf_)                                                                             \
  {                                                                                                                    \
    const Cplx<double> *body_ = bodyExpr_;                                                                             \
    const Cplx<double> *gh_ = ghostBase + (int64_t)(n_)*a.ghost_vec_stride;                                            \
    Cplx<double> *dst_ = (buf_) + (size_t)wave * kMT_PerWave * kMT_Chunk;                                              \
    _Pragma("unroll") for (int i = 0; i < kMT_PerWave; i++) {                                                          \
      const Cplx<double> *ptr_ = (((sghost >> i) & 1u) ? gh_ : body_) + soff[i];                                       \
      __builtin_amdgcn_global_load_lds(as_global(reinterpret_cast<const vec2 *>(ptr_)), (lds_void *)(dst_ + (size_t)i * kMT_Chunk), 16, 0, 0); \
    }                                                                                                                  \
  }
#define MUGIQ_MT_BARRIER()                              \
  {                                                     \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    if (MUGIQ_MT_EXPERIMENT < 2) __builtin_amdgcn_s_barrier(); \
    asm volatile("" ::: "memory");                      \
  }
  // the arithmetic of one eigenvector (scaled by s_) on the tile buffer tile_
#define MUGIQ_MT_COMPUTE(tile_, s_)                                                                                    \
  {                                                                                                                    \
    const Cplx<double> *tile = tile_;                                                                                  \
    const double sc = s_;                                                                                              \
    Cplx<double> vv[kMT_Groups], pp_[kMT_Groups][NS];                                                                \
    _Pragma("unroll") for (int g = 0; g < kMT_Groups; g++) {                                                           \
      if (MUGIQ_MT_EXPERIMENT >= 3) {                                                                                  \
        vv[g] = Cplx<double>{WR[g][0] + sc, WI[g][0] - sc};                                                            \
        _Pragma("unroll") for (int s = 0; s < NS; s++) pp_[g][s] = Cplx<double>{WI[g][s] * sc, WR[g][s] + sc};         \
        continue;                                                                                                      \
      }                                                                                                                \
      vv[g] = tile[vIdx + 4 * g];                                                                                      \
      _Pragma("unroll") for (int s = 0; s < NS; s++) pp_[g][s] = tile[pIdx[s] + 4 * g];                                \
    }                                                                                                                  \
    if (HOIST) __builtin_amdgcn_sched_barrier(0); /* all operand reads of the step in flight before the first product */ \
    _Pragma("unroll") for (int g = 0; g < kMT_Groups; g++) {                                                           \
      const Cplx<double> v = vv[g];                                                                                    \
      const double VR = sc * v.re, VI = sc * v.im, nVI = -VI;                                                          \
      _Pragma("unroll") for (int s = 0; s < NS; s++) {                                                                 \
        const Cplx<double> p = pp_[g][s];                                                                              \
        double tR = MUGIQ_MFMA(WR[g][s], p.re, 0.0);                                                                   \
        double tI = MUGIQ_MFMA(WI[g][s], p.re, 0.0);                                                                   \
        tR = MUGIQ_MFMA(nWI[g][s], p.im, tR);                                                                          \
        tI = MUGIQ_MFMA(WR[g][s], p.im, tI);                                                                           \
        aR[g][s] = MUGIQ_MFMA(VR, tR, aR[g][s]);                                                                       \
        aI[g][s] = MUGIQ_MFMA(VR, tI, aI[g][s]);                                                                       \
        aR[g][s] = MUGIQ_MFMA(VI, tI, aR[g][s]);                                                                       \
        aI[g][s] = MUGIQ_MFMA(nVI, tR, aI[g][s]);                                                                      \
      }                                                                                                                \
    }                                                                                                                  \
  }
  // One step: eigenvector n_ lands in buffer cur_ (own share: counted vmcnt wait; everybody's: the barrier, which also says
  // that nobody reads buffer nxt2_ = the one consumed in the previous step any more); eigenvector n_+2 is sent there, n_+1
  // stays in flight, n_ is consumed.
#define MUGIQ_MT_STEP(n_, cur_, nxt2_, STEADY)                                                                         \
  {                                                                                                                    \
    const double sNow = sigPre;                                                                                        \
    const Cplx<double> *bodyNow = bodyPre;                                                                             \
    if (MUGIQ_MT_EXPERIMENT == 0 && (STEADY || (n_) + 1 < a.nVec)) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kMT_PerWave) : "memory"); \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                              \
    MUGIQ_MT_BARRIER()                                                                                                 \
    if (MUGIQ_MT_EXPERIMENT == 0 && (STEADY || (n_) + 2 < a.nVec)) MUGIQ_MT_GLDS(bodyNow, (n_) + 2, nxt2_)             \
    MUGIQ_MT_COMPUTE(cur_, sNow)                                                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    {                                                                                                                  \
      const int nb_ = (n_) + 3 < a.nVec ? (n_) + 3 : a.nVec - 1, ns_ = (n_) + 1 < a.nVec ? (n_) + 1 : a.nVec - 1;      \
      bodyPre = MUGIQ_MT_BODY(nb_);                                                                                    \
      sigPre = MUGIQ_MT_SIGMA(ns_);                                                                                    \
    }                                                                                                                  \
  }
  Cplx<double> *const buf0 = tileBase, *const buf1 = tileBase + kMT_BufElems, *const buf2 = tileBase + 2 * kMT_BufElems;
  const int last = a.nVec - 1;
  MUGIQ_MT_GLDS(MUGIQ_MT_BODY(0), 0, buf0)
  MUGIQ_MT_GLDS(MUGIQ_MT_BODY((1 < last ? 1 : last)), (1 < last ? 1 : last), buf1)  // (unconditional: known count in flight)
  const Cplx<double> *bodyPre = MUGIQ_MT_BODY((2 < last ? 2 : last));
  double sigPre = MUGIQ_MT_SIGMA(0);
  int n = 0;
  for (; n + 4 < a.nVec; n += 3) {
    MUGIQ_MT_STEP(n, buf0, buf2, 1)
    MUGIQ_MT_STEP(n + 1, buf1, buf0, 1)
    MUGIQ_MT_STEP(n + 2, buf2, buf1, 1)
  }
  for (; n < a.nVec; n += 3) {  // n % 3 == 0 here
    MUGIQ_MT_STEP(n, buf0, buf2, 0)
    if (n + 1 < a.nVec) MUGIQ_MT_STEP(n + 1, buf1, buf0, 0)
    if (n + 2 < a.nVec) MUGIQ_MT_STEP(n + 2, buf2, buf1, 0)
  }
#undef MUGIQ_MT_STEP
#undef MUGIQ_MT_COMPUTE
#undef MUGIQ_MT_GLDS
#undef MUGIQ_MT_BODY
#undef MUGIQ_MT_SIGMA

  // ---- epilogue: lane 16 be + 4 b + al holds element (be, al) of the spin matrix of site b.  Through LDS (the tile buffers
  // are free now) as [slot][be * 4 + al][site = position * 32 + line], then one thread per (slot, site, half of the gamma
  // channels) as in csrc/fused_tile.hip: consecutive lanes <-> consecutive lines, so the stores stay coalesced per channel.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  MUGIQ_MT_BARRIER()
  Cplx<double> *scratch = tileBase;
#pragma unroll
  for (int g = 0; g < kMT_Groups; g++)
#pragma unroll
    for (int s = 0; s < NS; s++)
      scratch[(s * 16 + hi * 4 + lo) * 128 + wpos * kMT_Cols + lineBase + 4 * g] = Cplx<double>{aR[g][s], aI[g][s]};
  MUGIQ_MT_BARRIER()
  for (int item = threadIdx.x; item < 256 * NS; item += 64 * kMT_Waves) {
    const int site = item & 127, half = (item >> 7) & 1, s = item >> 8;
    const int pos = site >> 5;
    int p0, base, faceIdx;
    bool ok;
    line_info(site & 31, p0, base, faceIdx, ok);
    if (!ok) continue;
    const int jmine = j0 + pos;
    const int pmine = p0 ^ (jmine & 1), xmine = base + jmine * a.strideMu;
    Cplx<double> full[16];
#pragma unroll
    for (int e = 0; e < 16; e++) full[e] = scratch[(s * 16 + e) * 128 + site];
    const int siteIdx = xmine + pmine * a.volumeCB;
    if (half == 0) trace_and_store_range<double, 0, 8>(a.out[s], full, 2 * a.volumeCB, siteIdx, a.overwrite != 0);
    else trace_and_store_range<double, 8, 16>(a.out[s], full, 2 * a.volumeCB, siteIdx, a.overwrite != 0);
  }
#undef MUGIQ_MT_BARRIER
#endif
}
#undef MUGIQ_MFMA

// Can the matrix-pipe tile take this entry?  fp64 FLOAT2 storage and loops, mu = y, z, t, at most 8 staged positions.
// MUGIQ_HIP_TILE_MFMA = 0 switches it off (the vector tiles of csrc/fused_tile.hip / fused_tile16.hip take over).
bool mfma_tile_applicable(const MugiqHipSpinorField &ev, int dir, int kmax, int partitioned) {
  if (const char *e = getenv("MUGIQ_HIP_TILE_MFMA"))
    if (atoi(e) == 0) return false;
  if (const char *e = getenv("MUGIQ_HIP_FUSED_TILE"))
    if (atoi(e) == 0) return false;  // streaming kernel only
  if (const char *e = getenv("MUGIQ_HIP_TILE_COLS"))
    if (atoi(e) != 0) return false;  // a vector-tile generation was asked for by name
  if (const char *e = getenv("MUGIQ_HIP_TILE_GLDS"))
    if (atoi(e) == 0) return false;  // register-staged vector tile asked for
  if (ev.precision != 8 || ev.field_order != 2 || dir < 1) return false;
  if (2 * (int64_t)ev.parity_offset >= (1LL << 31)) return false;  // the kernel keeps 32-bit element offsets
  if (ev.X[dir] % kMT_TJ != 0 || kmax > ev.X[dir]) return false;
  return kMT_TJ + kmax <= 2 * kMT_Pairs;
}

static int launch_mfma_tile(MTileArgs a, int dir, int sign, int ns, hipStream_t stream) {
  const size_t shmem = (size_t)3 * kMT_BufElems * sizeof(Cplx<double>);
  const unsigned nblocks = (unsigned)(((a.numCols + kMT_Cols - 1) / kMT_Cols) * a.jtCount);
  a.blockOrder = 2;
  if (const char *e = getenv("MUGIQ_HIP_TILE_ORDER")) a.blockOrder = atoi(e) & 2;
  if (nblocks % 8 != 0) a.blockOrder = 0;
  int groups = 4, hoist = 0;
  if (const char *e = getenv("MUGIQ_HIP_MFMA_GROUPS")) groups = atoi(e) == 2 ? 2 : 4;
  if (const char *e = getenv("MUGIQ_HIP_MFMA_HOIST")) hoist = atoi(e) != 0;
  const dim3 grid(nblocks), block(64 * 32 / groups);
#define MUGIQ_MT_LAUNCH_(D, S, N, GG, HH)                                                                                     \
  {                                                                                                                    \
    auto kern = mfma_tile_displaced_contract_kernel<D, S, N, GG, HH>;                                                  \
    MUGIQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
    hipLaunchKernelGGL(kern, grid, block, shmem, stream, a);                                                           \
  }
#define MUGIQ_MT_LAUNCH(D, S, N)                                                                                       \
  {                                                                                                                    \
    if (groups == 2 && hoist) MUGIQ_MT_LAUNCH_(D, S, N, 2, true) else if (groups == 2) MUGIQ_MT_LAUNCH_(D, S, N, 2, false)   \
    else if (hoist) MUGIQ_MT_LAUNCH_(D, S, N, 4, true) else MUGIQ_MT_LAUNCH_(D, S, N, 4, false)                           \
  }
#define MUGIQ_MT_CASE(D, S)                                                                                            \
  case (D)*2 + (S):                                                                                                    \
    if (ns == 1) MUGIQ_MT_LAUNCH(D, S, 1) else if (ns == 2) MUGIQ_MT_LAUNCH(D, S, 2) else if (ns == 3) MUGIQ_MT_LAUNCH(D, S, 3) else MUGIQ_MT_LAUNCH(D, S, 4) \
    break;
  switch (dir * 2 + sign) {
    MUGIQ_MT_CASE(1, 0) MUGIQ_MT_CASE(1, 1) MUGIQ_MT_CASE(2, 0) MUGIQ_MT_CASE(2, 1) MUGIQ_MT_CASE(3, 0) MUGIQ_MT_CASE(3, 1)
  default:
    return set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "mfma tile: direction %d has no matrix-pipe tile (internal)", dir);
  }
#undef MUGIQ_MT_CASE
#undef MUGIQ_MT_LAUNCH
#undef MUGIQ_MT_LAUNCH_
  MUGIQ_CHECK_HIP(hipGetLastError());
  return MUGIQ_HIP_SUCCESS;
}

// ultra_d != NULL: also produce the ultra-local loop (k = 0, W = 1) into ultra_d as one more slot of the first launch; *carried
// says whether that happened (only a launch over the whole lattice may: see csrc/fused_tile.hip)
int mfma_tile_entry(void *loop_d, const MugiqHipSpinorField *ev, const double *sigma, int nVec, const void *const *E_d, const int *kvals,
                    int nK, int dir, int sign, int partitioned, const void *ghost_d, int layers, int region, hipStream_t stream,
                    void *ultra_d, int *carried) {
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
  const int64_t slot_stride = (int64_t)16 * 2 * ev[0].volumeCB;
  if (carried) *carried = 0;
  a.L = reinterpret_cast<const void *const *>(dev);
  a.inv_sigma = reinterpret_cast<const double *>(static_cast<unsigned char *>(dev) + ptr_bytes);
  a.nVec = nVec;
  long long strideMu = 1;
  for (int d = 0; d < 4; d++) {
    a.X[d] = ev[0].X[d];
    if (d < dir) strideMu *= ev[0].X[d];
  }
  strideMu /= 2;
  a.volumeCB = ev[0].volumeCB;
  a.stride = ev[0].stride;
  a.parity_offset = ev[0].parity_offset;
  a.partitioned = partitioned;
  a.ghost = static_cast<const double *>(ghost_d);
  a.faceCB = ev[0].volumeCB / ev[0].X[dir];
  a.ghost_vec_stride = (int64_t)layers * 24 * a.faceCB;
  a.strideMu = (int)strideMu;
  a.H = (int)(ev[0].volumeCB / (ev[0].X[dir] * strideMu));
  a.numCols = 2 * ev[0].volumeCB / ev[0].X[dir];
  const int nJT = ev[0].X[dir] / kMT_TJ;
  a.overwrite = (region & MUGIQ_HIP_REGION_OVERWRITE) ? 1 : 0;
  region &= 0xff;
  if (region != MUGIQ_HIP_REGION_ALL) ultra_d = nullptr;
  const int perLaunch = kMT_MaxSlots - 1;  // displaced slots per launch
  for (int k0 = 0; k0 < nK; k0 += perLaunch) {
    int ns = std::min(nK - k0, perLaunch);
    a.kmax = 0;
    for (int s = 0; s < kMT_MaxSlots; s++) {
      const int i = k0 + (s < ns ? s : 0);
      a.E[s] = static_cast<const double *>(E_d[i]);
      a.k[s] = kvals[i];
      a.out[s] = static_cast<Cplx<double> *>(loop_d) + (int64_t)i * slot_stride;
      if (s < ns && kvals[i] > a.kmax) a.kmax = kvals[i];
    }
    bool withUltra = false;
    if (ultra_d && k0 == 0) {
      a.E[ns] = nullptr;
      a.k[ns] = 0;
      a.out[ns] = static_cast<Cplx<double> *>(ultra_d);
      ns++;
      withUltra = true;
    }
    // region 0: everything | 1: tiles whose shifted reads stay inside the local lattice | 2: tiles that read ghost layers
    a.jtBegin = 0;
    a.jtCount = nJT;
    if (region != MUGIQ_HIP_REGION_ALL) {
      const int nb = partitioned ? std::min(nJT, (a.kmax + kMT_TJ - 1) / kMT_TJ) : 0;  // boundary tiles
      if (region == MUGIQ_HIP_REGION_INTERIOR) {
        a.jtBegin = sign == MUGIQ_HIP_DISP_SIGN_PLUS ? 0 : nb;
        a.jtCount = nJT - nb;
      } else {
        a.jtBegin = sign == MUGIQ_HIP_DISP_SIGN_PLUS ? nJT - nb : 0;
        a.jtCount = nb;
      }
    }
    if (a.jtCount > 0) {
      st = launch_mfma_tile(a, dir, sign, ns, stream);
      if (st) return st;
      if (withUltra && carried) *carried = 1;
    }
  }
  return MUGIQ_HIP_SUCCESS;
}

}  // namespace mugiq
