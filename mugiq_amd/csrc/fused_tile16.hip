// LDS-tiled fused displaced contraction with 16-line items: the tile of csrc/fused_tile.hip cut into 16-line pieces.
//
// The unit of work ("item") is one (position, slot) pair on 16 lines = 32 lanes (two spin halves), two items per wave:
//   * row tile (mu = x): positions are 16-entry pieces of whole x rows, lcm(16, X0/2)/16 per parity -- at X0 = 48 two rows =
//     three pieces per parity, 6 positions x 3 slots = 18 items = 9 waves, every lane busy (the 32-line positions of
//     csrc/fused_tile.hip hold one 24-entry row each there: a quarter of every wave idle);
//   * column tile (mu = y, z, t): 4 positions x 3 slots = 12 items = 6 waves, 2 x 21.5 KB of LDS, two workgroups per CU.
// Staging is by linear element index (element e of the [position][12][16] tile <-> thread e mod #threads), so any number of
// waves stages any tile.  Arithmetic, prefetch depth, barrier discipline and the ghost / wrap handling are those of
// csrc/fused_tile.hip.  Which generation takes what is decided by measurement (tile16_applicable below).
#include "internal.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace mugiq {

constexpr int kT16Cols = 16;       // lines per item (each line is held by two lanes: one per spin half)
static int t16_tj() {              // positions along mu per column tile (MUGIQ_HIP_TILE16_TJ = 4 | 8)
  if (const char *e = getenv("MUGIQ_HIP_TILE16_TJ")) return atoi(e) == 8 ? 8 : 4;
  return 4;
}
constexpr int kT16MaxSlots = 3;
constexpr int kT16MaxItems = 24;   // 12 waves
constexpr int kT16MaxPos = 16;     // staged positions: TJ + Kmax upper bound
constexpr int kT16Row = 12 * kT16Cols;  // elements of one staged position

template <typename F, typename A> struct Tile16Args {
  Cplx<A> *loop;
  int64_t slot_stride;
  const void *const *L;
  const A *inv_sigma;
  int nVec;
  int X[4];
  int volumeCB;
  int stride;
  int64_t parity_offset;
  const F *E[kT16MaxSlots];
  int k[kT16MaxSlots];
  int nslot;
  int kmax;
  int partitioned;
  const F *ghost;
  int64_t ghost_vec_stride;
  int faceCB;
  int strideMu;   // x_cb distance of one step along DIR
  int H;          // volumeCB / (X[DIR] * strideMu)
  int numCols;    // V / X[DIR]
  int jtBegin;    // column tile: tiles along mu handled by this launch: [jtBegin, jtBegin + jtCount)
  int jtCount;
  int tj;         // column tile: positions along mu per tile
  int npc;        // computed positions per tile (column: TJ; row: 2 m)
  int np;         // staged positions per tile (column: TJ + kmax; row: 2 m)
  int m;          // row tile: 16-entry pieces per parity
  int blockOrder; // bit 1: XCD-contiguous workgroup order
  int overwrite;  // store instead of accumulate (MUGIQ_HIP_REGION_OVERWRITE)
};

template <int ORDER> __device__ inline int64_t comp_offset16(int comp, int64_t stride, int64_t idx) {
  if constexpr (ORDER == 2) return (int64_t)comp * stride + idx;
  else return ((int64_t)(comp >> 1) * stride + idx) * 2 + (comp & 1);
}

// PHL bounds the staging loads per lane and eigenvector (ceil(np * 192 / #threads)).
// GLDS (fp64 FLOAT2 storage; round 3): the tile goes global -> LDS directly (global_load_lds_dwordx4) into THREE buffers -- one
// consumed, two in flight -- exactly as in csrc/fused_tile.hip: no stage registers, no ds_write pass.  The staging index is
// already linear in the thread (element e <-> thread e mod #threads), which is the lane-linear image the transfer wants.
template <typename F, typename A, int ORDER, int DIR, int SIGN, int PHL, bool GLDS>
__device__ __forceinline__ void tile16_body(const Tile16Args<F, A> &a) {
  extern __shared__ __align__(16) unsigned char smem[];
  Cplx<F> *tileBase = reinterpret_cast<Cplx<F> *>(smem);  // 2 (GLDS: 3) x [PHL * #threads] (buffered over the eigenvectors)
  const int nthreads = blockDim.x;
  const size_t tileElems = (size_t)PHL * nthreads;        // padded: commits are unconditional
  const int t = threadIdx.x, lane = t & 63;
  const int col = lane & 15, half = (lane >> 4) & 1;
  const int item = 2 * (t >> 6) + (lane >> 5);
  const bool computes = item < a.npc * a.nslot;
  const int wpos = item % a.npc;  // this item's own position
  const int slot = computes ? item / a.npc : 0;
  const int k = a.k[slot];
  const int J = a.X[DIR];
  const int nElem = a.np * kT16Row;

  int blk = blockIdx.x;
  if (a.blockOrder & 2) {  // XCD-contiguous: workgroups are dealt round-robin over the 8 XCDs
    const int per = gridDim.x >> 3;
    blk = (blk & 7) * per + (blk >> 3);
  }
  // ---- the tile.  Column tile: 16 lines along mu x TJ consecutive positions j0 .. j0 + TJ - 1; staged position pp <->
  // coordinate j = j0 + pp (sign +) | j0 - kmax + pp (sign -), parity alternating with j.  Row tile: m 16-entry pieces of
  // whole x rows per parity, position pp = parity | piece << 1; entry f = piece * 16 + col of the tile's run of m * 16
  // checkerboard entries.
  int j0 = 0, cc = blk, tileBaseX = 0;
  if constexpr (DIR >= 1) {
    const int jt = a.jtBegin + blk % a.jtCount;
    cc = blk / a.jtCount;
    j0 = jt * a.tj;
  } else {
    tileBaseX = blk * (a.m * kT16Cols);
  }
  // per-line data of the column tile (as a function of the column index within the tile)
  auto line = [&](int c16, int &p0, int &base, int &faceIdx, bool &ok) {
    int cid = cc * kT16Cols + c16;
    ok = cid < a.numCols;
    if (!ok) cid = a.numCols - 1;  // surplus lanes shadow the last line (valid addresses, result dropped)
    const int colsPerParity = a.H * a.strideMu;
    p0 = cid / colsPerParity;
    const int rem = cid - p0 * colsPerParity;
    const int hi = rem / a.strideMu;
    const int lo = rem - hi * a.strideMu;
    base = hi * (J * a.strideMu) + lo;  // x_cb of the line's j = 0 site (parity p0)
    int c0[4];
    get_coords(c0, base, a.X, p0);      // c0[DIR] == 0
    faceIdx = ghost_face_index_on_face(c0, a.X, DIR);
  };
  int p0 = 0, base = 0, faceIdx = 0;
  bool active = true;
  if constexpr (DIR >= 1) line(col, p0, base, faceIdx, active);
  const Cplx<F> *ghostBase = reinterpret_cast<const Cplx<F> *>(a.ghost);

  // ---- staging: element e = (pp * 12 + comp) * 16 + c of the tile <-> thread e mod #threads (c == col: #threads is a
  // multiple of 16); source offset and "is a ghost layer" bit are fixed for the whole kernel
  int soff[PHL];
  unsigned sghost = 0;
#pragma unroll
  for (int i = 0; i < PHL; i++) {
    int e = t + nthreads * i;
    e = e < nElem ? e : nElem - kT16Cols + col;  // surplus slots re-read an element of the last row (same value, same place)
    const int pp = e / kT16Row, comp = (e - pp * kT16Row) >> 4;
    soff[i] = 0;
    if constexpr (DIR >= 1) {
      int j = (SIGN == MUGIQ_HIP_DISP_SIGN_PLUS) ? j0 + pp : j0 - a.kmax + pp;
      const int par = p0 ^ (j & 1);
      if ((j < 0 || j >= J) && a.partitioned) {
        const int layer = (j >= J) ? j - J : -j - 1;
        sghost |= 1u << i;
        soff[i] = (int)((int64_t)layer * 24 * a.faceCB + (int64_t)par * 12 * a.faceCB + comp_offset16<ORDER>(comp, a.faceCB, faceIdx));
      } else {
        j = j < 0 ? j + J : (j >= J ? j - J : j);
        soff[i] = (int)((int64_t)par * a.parity_offset + comp_offset16<ORDER>(comp, a.stride, base + j * a.strideMu));
      }
    } else {
      soff[i] = (int)((int64_t)(pp & 1) * a.parity_offset + comp_offset16<ORDER>(comp, a.stride, tileBaseX + (pp >> 1) * kT16Cols + col));
    }
  }
  // where thread t commits its element i: e itself (clamped like above)
  auto commit_index = [&](int i) {
    const int e = t + nthreads * i;
    return e < nElem ? e : nElem - kT16Cols + col;
  };

  // ---- my site, and the LDS slots of my v(x) and of my shifted v(x +- k mu)
  int pmine, xmine, ppL, ppS, colS = col;
  if constexpr (DIR >= 1) {
    ppL = (SIGN == MUGIQ_HIP_DISP_SIGN_PLUS) ? wpos : a.kmax + wpos;
    ppS = (SIGN == MUGIQ_HIP_DISP_SIGN_PLUS) ? wpos + k : a.kmax + wpos - k;
    const int jmine = j0 + wpos;
    pmine = p0 ^ (jmine & 1);
    xmine = base + jmine * a.strideMu;
  } else {
    const int ePR = a.X[0] >> 1;  // checkerboard entries per x row; the tile's run starts on a row boundary
    ppL = wpos;
    pmine = wpos & 1;
    const int f = (wpos >> 1) * kT16Cols + col;
    xmine = tileBaseX + f;
    int c[4];
    get_coords(c, xmine, a.X, pmine);
    int xs = c[0] + ((SIGN == MUGIQ_HIP_DISP_SIGN_PLUS) ? k : -k);
    xs %= a.X[0];
    if (xs < 0) xs += a.X[0];
    const int fs = (f / ePR) * ePR + (xs >> 1);  // same row, entry x' / 2; the parity flips for odd k
    ppS = ((fs >> 4) << 1) | (pmine ^ (k & 1));
    colS = fs & 15;
  }

  // W_k(x) of this lane's site (both lane halves hold the same 3x3): in VGPRs
  Cplx<A> Wr[9];
  {
    const Cplx<F> *e = reinterpret_cast<const Cplx<F> *>(a.E[slot]) + (int64_t)pmine * 12 * a.volumeCB + xmine;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const Cplx<F> w = e[(int64_t)(j * 3 + i) * a.volumeCB];
        Wr[i * 3 + j] = Cplx<A>{(A)w.re, (A)w.im};
      }
  }
  Cplx<A> acc[8];  // acc[be*2 + a2], al = 2*half + a2
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = Cplx<A>{A(0), A(0)};

  typedef F vec2 __attribute__((ext_vector_type(2)));
  constexpr int kDepth = sizeof(F) == 8 ? 2 : 3;  // eigenvectors in flight ahead of the one being consumed
  vec2 stageA[PHL], stageB[PHL], stageC[PHL];
#define MUGIQ_T16_BODY(n_) static_cast<const Cplx<F> *>(as_constant(a.L)[n_])
#define MUGIQ_T16_SIGMA(n_) as_constant(a.inv_sigma)[n_]
#define MUGIQ_T16_FETCH(n_, stage) MUGIQ_T16_FETCH_AT(MUGIQ_T16_BODY(n_), n_, stage)
#define MUGIQ_T16_FETCH_AT(bodyExpr_, n_, stage)                                                                       \
  {                                                                                                                    \
    const Cplx<F> *body_ = bodyExpr_;                                                                                  \
    const Cplx<F> *gh_ = ghostBase + (int64_t)(n_)*a.ghost_vec_stride;                                                 \
    _Pragma("unroll") for (int i = 0; i < PHL; i++) { /* unconditional: every lane and slot has a valid source */      \
      const Cplx<F> *ptr_ = (((sghost >> i) & 1u) ? gh_ : body_) + soff[i];                                            \
      stage[i] = *as_global(reinterpret_cast<const vec2 *>(ptr_));                                                     \
    }                                                                                                                  \
  }
// workgroup barrier that orders LDS traffic only (__syncthreads() would also drain the global loads in flight)
#define MUGIQ_T16_BARRIER()                              \
  {                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
    __builtin_amdgcn_s_barrier();                        \
    asm volatile("" ::: "memory");                       \
  }
// the arithmetic of one eigenvector (scaled by s_) on the tile buffer tile_
#define MUGIQ_T16_COMPUTE(tile_, s_)                                                                                   \
  {                                                                                                                    \
    const Cplx<F> *tile = tile_;                                                                                       \
    const A s = s_;                                                                                                    \
    if (computes) {                                                                                                    \
      const Cplx<F> *tl = tile + (ppL * 12) * kT16Cols + col;                                                          \
      const Cplx<F> *ts = tile + (ppS * 12 + half * 6) * kT16Cols + colS; /* spins 2*half, 2*half + 1 */               \
      Cplx<A> t0[3], t1[3];                                                                                            \
      _Pragma("unroll") for (int i = 0; i < 3; i++) t0[i] = t1[i] = Cplx<A>{A(0), A(0)};                               \
      _Pragma("unroll") for (int j = 0; j < 3; j++) {                                                                  \
        const Cplx<F> w0 = ts[j * kT16Cols], w1 = ts[(3 + j) * kT16Cols];                                              \
        const Cplx<A> p0j{(A)w0.re, (A)w0.im}, p1j{(A)w1.re, (A)w1.im};                                                \
        _Pragma("unroll") for (int i = 0; i < 3; i++) {                                                                \
          const Cplx<A> w = Wr[i * 3 + j];                                                                             \
          cmadd(t0[i], w, p0j);                                                                                        \
          cmadd(t1[i], w, p1j);                                                                                        \
        }                                                                                                              \
      }                                                                                                                \
      _Pragma("unroll") for (int i = 0; i < 3; i++) {                                                                  \
        t0[i] = Cplx<A>{s * t0[i].re, s * t0[i].im};                                                                   \
        t1[i] = Cplx<A>{s * t1[i].re, s * t1[i].im};                                                                   \
      }                                                                                                                \
      _Pragma("unroll") for (int be = 0; be < 4; be++) {                                                               \
        if (be == 2) __builtin_amdgcn_sched_barrier(0); /* bound how many LDS reads the scheduler hoists (VGPRs) */    \
        _Pragma("unroll") for (int c = 0; c < 3; c++) {                                                                \
          const Cplx<F> u = tl[(be * 3 + c) * kT16Cols];                                                               \
          const Cplx<A> lv{(A)u.re, (A)u.im};                                                                          \
          cmadd_conj(acc[be * 2 + 0], lv, t0[c]);                                                                      \
          cmadd_conj(acc[be * 2 + 1], lv, t1[c]);                                                                      \
        }                                                                                                              \
      }                                                                                                                \
    }                                                                                                                  \
  }
// One step: eigenvector n_ is in LDS buffer n_ % 2; `stage` holds eigenvector n_+1.  Commit n_+1 into the other buffer,
// refill `stage` with n_ + kDepth + 1, consume n_, one barrier.
#define MUGIQ_T16_STEP(n_, stage, GUARD)                                                                               \
  {                                                                                                                    \
    const A sNow = sigPre;                                                                                             \
    const Cplx<F> *bodyNow = bodyPre;                                                                                  \
    {                                                                                                                  \
      const int nb_ = (n_) + kDepth + 2 < a.nVec ? (n_) + kDepth + 2 : a.nVec - 1, ns_ = (n_) + 1 < a.nVec ? (n_) + 1 : a.nVec - 1; \
      bodyPre = MUGIQ_T16_BODY(nb_);                                                                                   \
      sigPre = MUGIQ_T16_SIGMA(ns_);                                                                                   \
    }                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    if (GUARD == 0 || (n_) + 1 < a.nVec) {                                                                             \
      Cplx<F> *nxt = tileBase + (size_t)(((n_) + 1) & 1) * tileElems;                                                  \
      _Pragma("unroll") for (int i = 0; i < PHL; i++) nxt[commit_index(i)] = Cplx<F>{stage[i].x, stage[i].y};          \
    }                                                                                                                  \
    if (GUARD == 0 || (n_) + kDepth + 1 < a.nVec) MUGIQ_T16_FETCH_AT(bodyNow, (n_) + kDepth + 1, stage)                \
    MUGIQ_T16_COMPUTE(tileBase + (size_t)((n_) & 1) * tileElems, sNow)                                                 \
    MUGIQ_T16_BARRIER()                                                                                                \
  }

  if constexpr (GLDS) {
#if defined(__HIP_DEVICE_COMPILE__)  // (global_load_lds is a device-only builtin: the host pass of hipcc must not see it)
    typedef __attribute__((address_space(3))) void lds_void;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // this lane's share of eigenvector n_ -> tile buffer buf_: PHL transfers of 64 x 16 bytes (elements wave * 64 + lane + #threads * i)
#define MUGIQ_T16_GLDS(bodyExpr_, n_, buf_)                                                                            \
  {                                                                                                                    \
    const Cplx<F> *body_ = bodyExpr_;                                                                                  \
    const Cplx<F> *gh_ = ghostBase + (int64_t)(n_)*a.ghost_vec_stride;                                                 \
    Cplx<F> *dst_ = (buf_) + (size_t)wave * 64;                                                                        \
    _Pragma("unroll") for (int i = 0; i < PHL; i++) {                                                                  \
      const Cplx<F> *ptr_ = (((sghost >> i) & 1u) ? gh_ : body_) + soff[i];                                            \
      __builtin_amdgcn_global_load_lds(as_global(reinterpret_cast<const vec2 *>(ptr_)), (lds_void *)(dst_ + (size_t)i * nthreads), 16, 0, 0); \
    }                                                                                                                  \
  }
#define MUGIQ_T16_GSTEP(n_, cur_, nxt2_, STEADY)                                                                       \
  {                                                                                                                    \
    const A sNow = sigPre;                                                                                             \
    const Cplx<F> *bodyNow = bodyPre;                                                                                  \
    if (STEADY || (n_) + 1 < a.nVec) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PHL) : "memory");                        \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                              \
    MUGIQ_T16_BARRIER()                                                                                                \
    if (STEADY || (n_) + 2 < a.nVec) MUGIQ_T16_GLDS(bodyNow, (n_) + 2, nxt2_)                                          \
    MUGIQ_T16_COMPUTE(cur_, sNow)                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    {                                                                                                                  \
      const int nb_ = (n_) + 3 < a.nVec ? (n_) + 3 : a.nVec - 1, ns_ = (n_) + 1 < a.nVec ? (n_) + 1 : a.nVec - 1;      \
      bodyPre = MUGIQ_T16_BODY(nb_);                                                                                   \
      sigPre = MUGIQ_T16_SIGMA(ns_);                                                                                   \
    }                                                                                                                  \
  }
    Cplx<F> *const buf0 = tileBase, *const buf1 = tileBase + tileElems, *const buf2 = tileBase + 2 * tileElems;
    const int last = a.nVec - 1;
    MUGIQ_T16_GLDS(MUGIQ_T16_BODY(0), 0, buf0)
    MUGIQ_T16_GLDS(MUGIQ_T16_BODY((1 < last ? 1 : last)), (1 < last ? 1 : last), buf1)  // (unconditional: known count in flight)
    const Cplx<F> *bodyPre = MUGIQ_T16_BODY((2 < last ? 2 : last));
    A sigPre = MUGIQ_T16_SIGMA(0);
    int n = 0;
    for (; n + 4 < a.nVec; n += 3) {
      MUGIQ_T16_GSTEP(n, buf0, buf2, 1)
      MUGIQ_T16_GSTEP(n + 1, buf1, buf0, 1)
      MUGIQ_T16_GSTEP(n + 2, buf2, buf1, 1)
    }
    for (; n < a.nVec; n += 3) {  // n % 3 == 0 here
      MUGIQ_T16_GSTEP(n, buf0, buf2, 0)
      if (n + 1 < a.nVec) MUGIQ_T16_GSTEP(n + 1, buf1, buf0, 0)
      if (n + 2 < a.nVec) MUGIQ_T16_GSTEP(n + 2, buf2, buf1, 0)
    }
#undef MUGIQ_T16_GSTEP
#undef MUGIQ_T16_GLDS
#endif
  } else {
  // prologue: eigenvector 0 -> LDS buffer 0; the next kDepth eigenvectors in flight in the stage registers (clamped,
  // unconditional loads: the steady-state loop is entered with a KNOWN number of loads in flight)
  MUGIQ_T16_FETCH(0, stageC)
#pragma unroll
  for (int i = 0; i < PHL; i++) tileBase[commit_index(i)] = Cplx<F>{stageC[i].x, stageC[i].y};
  {
    const int last = a.nVec - 1;
    MUGIQ_T16_FETCH((1 < last ? 1 : last), stageA)
    MUGIQ_T16_FETCH((2 < last ? 2 : last), stageB)
    if constexpr (kDepth == 3) MUGIQ_T16_FETCH((3 < last ? 3 : last), stageC)
  }
  const Cplx<F> *bodyPre = MUGIQ_T16_BODY(a.nVec > kDepth + 1 ? kDepth + 1 : a.nVec - 1);
  A sigPre = MUGIQ_T16_SIGMA(0);
  MUGIQ_T16_BARRIER()
  int n = 0;
  for (; n + 2 * kDepth < a.nVec; n += kDepth) {
    MUGIQ_T16_STEP(n, stageA, 0)
    MUGIQ_T16_STEP(n + 1, stageB, 0)
    if constexpr (kDepth == 3) MUGIQ_T16_STEP(n + 2, stageC, 0)
  }
  for (; n < a.nVec; n += kDepth) {
    MUGIQ_T16_STEP(n, stageA, 1)
    if (n + 1 < a.nVec) MUGIQ_T16_STEP(n + 1, stageB, 1)
    if constexpr (kDepth == 3)
      if (n + 2 < a.nVec) MUGIQ_T16_STEP(n + 2, stageC, 1)
  }
  }  // register-staged form
#undef MUGIQ_T16_STEP
#undef MUGIQ_T16_COMPUTE
  // ---- epilogue: the two lane halves of an item hold complementary halves of the 4x4 colour-traced spin matrix of the
  // same 16 sites.  Exchange them with wavefront shuffles (lane ^ 16), then each half takes 8 of the 16 gamma traces.
  {
    Cplx<A> full[16];
#pragma unroll
    for (int be = 0; be < 4; be++)
#pragma unroll
      for (int a2 = 0; a2 < 2; a2++) {
        const Cplx<A> mine = acc[be * 2 + a2];
        Cplx<A> theirs;
        theirs.re = __shfl_xor(mine.re, 16);
        theirs.im = __shfl_xor(mine.im, 16);
        full[be * 4 + a2] = half == 0 ? mine : theirs;
        full[be * 4 + 2 + a2] = half == 0 ? theirs : mine;
      }
    if (computes && active) {
      Cplx<A> *out = a.loop + (int64_t)slot * a.slot_stride;
      const int siteIdx = xmine + pmine * a.volumeCB;
      if (half == 0) trace_and_store_range<A, 0, 8>(out, full, 2 * a.volumeCB, siteIdx, a.overwrite != 0);
      else trace_and_store_range<A, 8, 16>(out, full, 2 * a.volumeCB, siteIdx, a.overwrite != 0);
    }
  }
}

template <typename F, typename A, int ORDER, int DIR, int SIGN, int PHL, bool GLDS>
__global__ __launch_bounds__(64 * 12) void tile16_displaced_contract_kernel(Tile16Args<F, A> a) {
  tile16_body<F, A, ORDER, DIR, SIGN, PHL, GLDS>(a);
}
#undef MUGIQ_T16_FETCH
#undef MUGIQ_T16_FETCH_AT
#undef MUGIQ_T16_BODY
#undef MUGIQ_T16_SIGMA
#undef MUGIQ_T16_BARRIER

static int gcd_int(int x, int y) { return y == 0 ? x : gcd_int(y, x % y); }

// geometry of a launch: computed / staged positions, threads, loads per lane; false if the tile does not apply
struct Tile16Plan {
  int npc, np, m, maxSlots, minWaves;
};
static bool tile16_plan(const MugiqHipSpinorField &ev, int dir, int kmax, int partitioned, Tile16Plan &p) {
  if (2 * (int64_t)ev.parity_offset >= (1LL << 31)) return false;  // the kernel keeps 32-bit element offsets
  if (dir == 0) {  // row tile: whole x rows in LDS, no ghost handling
    const int ePR = ev.X[0] / 2;
    if (partitioned || kmax >= ev.X[0]) return false;
    p.m = ePR / gcd_int(kT16Cols, ePR);  // lcm(16, ePR) / 16
    p.npc = p.np = 2 * p.m;
    if (p.npc > kT16MaxItems) return false;
    if (ev.volumeCB % (p.m * kT16Cols) != 0) return false;
    p.maxSlots = std::min(kT16MaxSlots, kT16MaxItems / p.npc);
    p.minWaves = 2;
    return true;
  }
  const int tj = t16_tj();
  if (ev.X[dir] % tj != 0) return false;
  if (kmax > ev.X[dir]) return false;  // the staged window wraps at most once around the lattice
  if (tj + kmax > kT16MaxPos) return false;
  p.m = 0;
  p.npc = tj;
  p.np = tj + kmax;
  p.maxSlots = kT16MaxSlots;
  p.minWaves = 6 * tj / 4;  // idle waves of a launch with fewer slots still stage
  return true;
}

// Measured (48.48.24.24 fp64, 100 eigenvectors, three slots): row tile 9.3 ms here against 11.3 ms with 32-line positions
// (24 of 32 lanes busy at X0 = 48); column tiles 10.6-10.9 ms here against 8.9 ms (256-byte instead of 512-byte runs per
// load instruction; the two workgroups per CU did not buy the overlap hoped for).  So by default this kernel takes the ROW
// tiles whose rows do not fill 32-line positions (32 % (X0/2) != 0) and whatever the second generation cannot take;
// MUGIQ_HIP_TILE_COLS = 16 | 32 forces one generation for everything it can take.
bool tile16_applicable(const MugiqHipSpinorField &ev, int dir, int kmax, int precision, int partitioned, bool secondGenerationApplies) {
  int cols = 0;
  if (const char *e = getenv("MUGIQ_HIP_TILE_COLS")) cols = atoi(e);
  if (cols == 32) return false;
  if (cols != 16 && secondGenerationApplies && !(dir == 0 && 32 % (ev.X[0] / 2) != 0)) return false;
  int mode = 1;  // MUGIQ_HIP_FUSED_TILE: 0 = streaming kernel only, 2 = column tile only (no row tile), default both
  if (const char *e = getenv("MUGIQ_HIP_FUSED_TILE")) mode = atoi(e);
  if (mode == 0 || (mode == 2 && dir == 0)) return false;
  Tile16Plan p;
  if (!tile16_plan(ev, dir, kmax, partitioned, p)) return false;
  // staging loads per lane with the fewest threads a launch may have (one slot)
  const int waves = std::max(p.minWaves, (p.npc + 1) / 2);
  const int phl = (p.np * kT16Row + 64 * waves - 1) / (64 * waves);
  const int phlSel = phl <= 2 ? 2 : (phl <= 4 ? 4 : 8);
  return phl <= 8 && (size_t)2 * 2 * precision * phlSel * 64 * waves <= 160 * 1024;  // two staging tiles (three when they fit: launch_tile16)
}

template <typename F, typename A, int ORDER> static int launch_tile16(Tile16Args<F, A> a, int dir, int sign, int minWaves, hipStream_t stream) {
  const int items = a.npc * a.nslot;
  const int waves = std::max(minWaves, (items + 1) / 2);
  const int nthreads = 64 * waves;
  const int phl = (a.np * kT16Row + nthreads - 1) / nthreads;
  const int PHLsel = phl <= 2 ? 2 : (phl <= 4 ? 4 : 8);
  // global -> LDS staging with three buffers: fp64 FLOAT2 storage (MUGIQ_HIP_TILE16_GLDS=0: register staging, two buffers)
  bool glds = std::is_same<F, double>::value && ORDER == 2 && 3 * sizeof(Cplx<F>) * (size_t)PHLsel * nthreads <= 160 * 1024;
  if (const char *e = getenv("MUGIQ_HIP_TILE16_GLDS")) glds = glds && atoi(e) != 0;
  const size_t shmem = (glds ? 3 : 2) * sizeof(Cplx<F>) * (size_t)PHLsel * nthreads;
  unsigned nblocks = dir == 0 ? (unsigned)(a.volumeCB / (a.m * kT16Cols)) : (unsigned)(((a.numCols + kT16Cols - 1) / kT16Cols) * a.jtCount);
  a.blockOrder = 2;
  if (const char *e = getenv("MUGIQ_HIP_TILE_ORDER")) a.blockOrder = atoi(e) & 2;
  if (nblocks % 8 != 0) a.blockOrder = 0;
  const dim3 grid(nblocks), block(nthreads);
#define MUGIQ_T16_LAUNCH(D, S, P) \
  if constexpr (std::is_same<F, double>::value && ORDER == 2) { \
    if (glds) MUGIQ_T16_LAUNCH_G(D, S, P, true) else MUGIQ_T16_LAUNCH_G(D, S, P, false) \
  } else MUGIQ_T16_LAUNCH_G(D, S, P, false)
#define MUGIQ_T16_LAUNCH_G(D, S, P, G)                                                                                \
  {                                                                                                                   \
    auto kern = tile16_displaced_contract_kernel<F, A, ORDER, D, S, P, G>;                                            \
    if (shmem > 64 * 1024)                                                                                            \
      MUGIQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
    hipLaunchKernelGGL(kern, grid, block, shmem, stream, a);                                                          \
  }
#define MUGIQ_T16_CASE(D, S)                                                                                          \
  case (D)*2 + (S):                                                                                                   \
    if (PHLsel == 2) MUGIQ_T16_LAUNCH(D, S, 2) else if (PHLsel == 4) MUGIQ_T16_LAUNCH(D, S, 4) else MUGIQ_T16_LAUNCH(D, S, 8) \
    break;
  switch (dir * 2 + sign) {
    MUGIQ_T16_CASE(0, 0) MUGIQ_T16_CASE(0, 1) MUGIQ_T16_CASE(1, 0) MUGIQ_T16_CASE(1, 1)
    MUGIQ_T16_CASE(2, 0) MUGIQ_T16_CASE(2, 1) MUGIQ_T16_CASE(3, 0) MUGIQ_T16_CASE(3, 1)
  }
#undef MUGIQ_T16_CASE
#undef MUGIQ_T16_LAUNCH
#undef MUGIQ_T16_LAUNCH_G
  MUGIQ_CHECK_HIP(hipGetLastError());
  return MUGIQ_HIP_SUCCESS;
}

template <typename F, typename A, int ORDER>
int tile16_entry(void *loop_d, const MugiqHipSpinorField *ev, const double *sigma, int nVec, const void *const *E_d, const int *kvals,
                 int nK, int dir, int sign, int partitioned, const void *ghost_d, int layers, int region, hipStream_t stream) {
  const size_t ptr_bytes = sizeof(void *) * (size_t)nVec;
  std::vector<unsigned char> host(ptr_bytes + sizeof(A) * (size_t)nVec);
  const void **hl = reinterpret_cast<const void **>(host.data());
  A *hs = reinterpret_cast<A *>(host.data() + ptr_bytes);
  for (int n = 0; n < nVec; n++) {
    hl[n] = ev[n].data;
    const F sg = static_cast<F>(sigma[n]);
    hs[n] = static_cast<A>(1.0 / sg);
  }
  void *dev = nullptr;
  int st = upload_table(&dev, host.data(), host.size(), stream);
  if (st) return st;
  Tile16Args<F, A> a;
  a.slot_stride = (int64_t)16 * 2 * ev[0].volumeCB;
  a.L = reinterpret_cast<const void *const *>(dev);
  a.inv_sigma = reinterpret_cast<const A *>(static_cast<unsigned char *>(dev) + ptr_bytes);
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
  a.ghost = static_cast<const F *>(ghost_d);
  a.faceCB = ev[0].volumeCB / ev[0].X[dir];
  a.ghost_vec_stride = (int64_t)layers * 24 * a.faceCB;
  if (dir == 0) strideMu = 1;  // unused by the row tile
  a.strideMu = (int)strideMu;
  a.H = (int)(ev[0].volumeCB / (ev[0].X[dir] * strideMu));
  a.numCols = 2 * ev[0].volumeCB / ev[0].X[dir];
  const int tj = t16_tj();
  a.tj = tj;
  const int nJT = dir == 0 ? 1 : ev[0].X[dir] / tj;
  a.overwrite = (region & MUGIQ_HIP_REGION_OVERWRITE) ? 1 : 0;
  region &= 0xff;
  int kmaxAll = 0;
  for (int i = 0; i < nK; i++) kmaxAll = std::max(kmaxAll, kvals[i]);
  Tile16Plan plan;
  MUGIQ_REQUIRE(tile16_plan(ev[0], dir, kmaxAll, partitioned, plan), "displacedLoopContraction: the 16-line tile does not apply (internal)");
  for (int k0 = 0; k0 < nK; k0 += plan.maxSlots) {
    a.nslot = std::min(nK - k0, plan.maxSlots);
    a.loop = static_cast<Cplx<A> *>(loop_d) + (int64_t)k0 * a.slot_stride;
    a.kmax = 0;
    for (int s = 0; s < kT16MaxSlots; s++) {
      const int i = k0 + (s < a.nslot ? s : 0);
      a.E[s] = static_cast<const F *>(E_d[i]);
      a.k[s] = kvals[i];
      if (s < a.nslot && kvals[i] > a.kmax) a.kmax = kvals[i];
    }
    a.m = plan.m;
    a.npc = plan.npc;
    a.np = dir == 0 ? plan.np : tj + a.kmax;
    // region 0: everything | 1: tiles whose shifted reads stay inside the local lattice | 2: tiles that read ghost layers
    a.jtBegin = 0;
    a.jtCount = nJT;
    if (region != MUGIQ_HIP_REGION_ALL && dir >= 1) {
      const int nb = partitioned ? std::min(nJT, (a.kmax + tj - 1) / tj) : 0;  // boundary tiles
      if (region == MUGIQ_HIP_REGION_INTERIOR) {
        a.jtBegin = sign == MUGIQ_HIP_DISP_SIGN_PLUS ? 0 : nb;
        a.jtCount = nJT - nb;
      } else {
        a.jtBegin = sign == MUGIQ_HIP_DISP_SIGN_PLUS ? nJT - nb : 0;
        a.jtCount = nb;
      }
    } else if (region == MUGIQ_HIP_REGION_BOUNDARY) {
      a.jtCount = 0;  // the row tile (x, never partitioned here) has no boundary part
    }
    if (a.jtCount > 0) {
      st = launch_tile16<F, A, ORDER>(a, dir, sign, plan.minWaves, stream);
      if (st) return st;
    }
  }
  return MUGIQ_HIP_SUCCESS;
}

#define MUGIQ_T16_INST(F, A, O)                                                                                                    \
  template int tile16_entry<F, A, O>(void *, const MugiqHipSpinorField *, const double *, int, const void *const *, const int *, int, \
                                     int, int, int, const void *, int, int, hipStream_t);
MUGIQ_T16_INST(double, double, 2)
MUGIQ_T16_INST(double, double, 4)
MUGIQ_T16_INST(float, float, 2)
MUGIQ_T16_INST(float, float, 4)
MUGIQ_T16_INST(float, double, 2)
MUGIQ_T16_INST(float, double, 4)
#undef MUGIQ_T16_INST

}  // namespace mugiq
