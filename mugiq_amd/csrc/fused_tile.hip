// LDS-tiled fused displaced contraction (second generation of csrc/fused.hip, same mathematics and same C entry point).
//
// Measured on MI355X (profiles/r01_fused_*): the first-generation kernel streams v_n(x) and v_n(x +- k mu) for every slot
// k from HBM/fabric -- 1 + nslots "units" of 24*B bytes per site and eigenvector -- and sits at the ~6 TB/s fabric limit
// with no cache reuse.  But v_n(x + k mu) IS v_n at another site of the same straight line along mu.  Here a workgroup owns
// 32 such lines ("columns": fixed other coordinates, consecutive coordinate j along mu, parity alternating with j) and TJ
// consecutive positions on them.  Per eigenvector it stages the TJ + Kmax positions it needs in LDS once (one component
// plane per wave, 512-byte runs over the lines), and every (position, slot) pair -- one wave each -- reads both its v_n(x)
// and its shifted v_n(x +- k mu) from LDS.  A workgroup requests (TJ + Kmax)/TJ units per site and eigenvector (TJ = 4,
// 3 slots: 1.75); with the XCD-contiguous workgroup order the Kmax halo positions are the core positions of the
// neighbouring tile on the same XCD and come out of its L2, so the HBM traffic is read-once (PMC: 27-28 GB per entry
// against 45 GB with the plain order, profiles/r02_pmc_extra_traffic.txt).  What bounds the kernel is fp64 issue at the
// clock the device grants under this mix of FMAs, LDS reads and loads (1.72-1.78 GHz, profiles/r02_kernel_clocks.txt).
// Staging: global -> LDS directly with three tile buffers (fp64 FLOAT2, <= 8 positions), otherwise through registers with
// two buffers; one LDS-only barrier per eigenvector either way.
// Positions beyond the local extent come from the ghost layers (partitioned) or wrap around (periodic), exactly as in
// the first-generation kernel.  Requirements: X[DIR] % TJ == 0, at most 3 slots per launch; DIR == 0 is the row tile
// (whole x-rows per workgroup, see the kernel; rows that do not fill 32 lanes go to csrc/fused_tile16.hip).
#include "internal.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

namespace mugiq {

constexpr int kTileTJ = 4;        // positions along mu per workgroup
constexpr int kTileMaxSlots = 3;  // displaced slots per launch: waves = kTileTJ * nslot <= 12
constexpr int kTileCarry = kTileMaxSlots + 1;  // + the ultra-local loop riding along as a slot with k = 0 (16 waves; GLDS form only)
constexpr int kTileMaxPos = 16;   // TJ + Kmax upper bound
constexpr int kTileCols = 32;     // lines per workgroup (each line is held by two lanes: one per spin half)

template <typename F, typename A> struct TileArgs {
  Cplx<A> *out[kTileCarry];  // where every slot goes
  const void *const *L;
  const A *inv_sigma;
  int nVec;
  int X[4];
  int volumeCB;
  int stride;
  int64_t parity_offset;
  const F *E[kTileCarry];  // path links W_k of every slot; NULL = the identity (k = 0)
  int k[kTileCarry];
  int nslot;
  int kmax;
  int partitioned;
  const F *ghost;
  int64_t ghost_vec_stride;
  int faceCB;
  int strideMu;   // x_cb distance of one step along DIR
  int H;          // volumeCB / (X[DIR] * strideMu)
  int numCols;    // V / X[DIR]
  int nJT;        // X[DIR] / TJ
  int jtBegin;    // tiles along mu handled by this launch: [jtBegin, jtBegin + jtCount)
  int jtCount;
  int tileBytes;  // LDS bytes of the two staging tiles
  int blockOrder; // workgroup -> tile map (see the kernel)
  int overwrite;  // store instead of accumulate (MUGIQ_HIP_REGION_OVERWRITE)
};

// complex-element offset of component `comp` at checkerboard index idx inside one parity block of a field / ghost zone
template <int ORDER> __device__ inline int64_t comp_offset(int comp, int64_t stride, int64_t idx) {
  if constexpr (ORDER == 2) return (int64_t)comp * stride + idx;
  else return ((int64_t)(comp >> 1) * stride + idx) * 2 + (comp & 1);
}

// Workgroup = 12 waves over 32 lines.  Wave w <-> (position jj = w % TJ, slot = w / TJ); inside a wave lanes 0-31 and
// 32-63 hold the SAME 32 lines and split the right-hand spin index (al in {0,1} | {2,3}), so a lane carries 8 of the 16
// accumulators -- at three waves per SIMD a lane has ~168 VGPRs, which 16 fp64 accumulators + W + the prefetch
// registers do not fit into.  For staging, wave w owns plane w (component 3*spin + colour): its two lane halves fetch
// alternate positions (32 lanes x 16 B = one 512-byte run each).  PH bounds the positions staged per lane.
//
// GLDS (fp64 FLOAT2 storage, column tiles of at most 8 positions): the staged positions go global -> LDS directly
// (global_load_lds_dwordx4: no stage registers, no ds_write pass) into THREE tile buffers -- one consumed, two in flight.  The
// transfer wants a lane-linear LDS image, so a buffer is [position pair][12][position & 1][32]: the two lane halves of an
// instruction (alternate positions, 512 bytes each) land next to each other.
template <typename F, typename A, int ORDER, int DIR, int SIGN, int PH, bool GLDS, int NW = 12>
__global__ __launch_bounds__(64 * NW) void tile_displaced_contract_kernel(TileArgs<F, A> a) {
  extern __shared__ __align__(16) unsigned char smem[];
  Cplx<F> *tileBase = reinterpret_cast<Cplx<F> *>(smem);  // 2 (3: GLDS) x [NP][12][32] (buffered over the eigenvectors)
  const size_t tileElems = (size_t)(2 * PH) * 12 * kTileCols;  // padded to 2*PH positions: commits are unconditional
  const int lane = threadIdx.x & 63;
  const int col = lane & 31, half = lane >> 5;
  const int wave = GLDS ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)(threadIdx.x >> 6);
  // element index of (position pp, component, line) inside a tile buffer, and the distance between components
  constexpr int kCompStride = GLDS ? 2 * kTileCols : kTileCols;
  auto tileIdx = [&](int pp, int comp, int c) {
    return GLDS ? (((pp >> 1) * 12 + comp) * 2 + (pp & 1)) * kTileCols + c : (pp * 12 + comp) * kTileCols + c;
  };
  const bool computes = wave < kTileTJ * a.nslot;
  const int NP = (DIR >= 1) ? kTileTJ + a.kmax : kTileTJ;
  const int J = a.X[DIR];

  // ---- which sites this lane works on, and where every staged position lives
  //  DIR >= 1 ("column" tile): 32 lines along mu x TJ consecutive positions j0..j0+TJ-1; position pp <-> coordinate
  //            j = j0 + pp (sign +) | j0 - kmax + pp (sign -), parity alternating with j, x_cb = base + j*strideMu.
  //  DIR == 0 ("row" tile): the lines run along x, i.e. along the coalescing direction, so the tile is 2 row groups
  //            x 2 parities of whole x-rows (32 checkerboard entries each): position pp = parity | rowgroup << 1; a
  //            shifted site is another entry of the same row in the other (k odd) or same (k even) parity -- every
  //            eigenvector is read exactly once (1 unit).
  const int wpos = wave % kTileTJ;  // this wave's own position index (jj for the column tile)
  const int slot = wave / kTileTJ;
  const int k = a.k[slot];
  int blk = blockIdx.x;
  if (a.blockOrder & 2) {  // XCD-contiguous: workgroups are dealt round-robin over the 8 XCDs
    const int per = gridDim.x >> 3;
    blk = (blk & 7) * per + (blk >> 3);
  }
  int jt = 0, cc = blk, j0 = 0;
  bool active;
  int p0 = 0, base = 0, faceIdx = 0;         // column tile
  int ePR = 1, rowsPerGroup = 1;             // row tile
  if constexpr (DIR >= 1) {
    if (a.blockOrder & 1) {  // consecutive workgroups = consecutive column groups (adjacent 512-byte runs)
      const int nCC = gridDim.x / a.jtCount;
      jt = a.jtBegin + blk / nCC;
      cc = blk % nCC;
    } else {
      jt = a.jtBegin + blk % a.jtCount;
      cc = blk / a.jtCount;
    }
    j0 = jt * kTileTJ;
    int cid = cc * kTileCols + col;
    active = cid < a.numCols;
    if (!active) cid = a.numCols - 1;  // surplus lanes shadow the last line (valid addresses, result dropped)
    const int colsPerParity = a.H * a.strideMu;
    p0 = cid / colsPerParity;
    const int rem = cid - p0 * colsPerParity;
    const int hi = rem / a.strideMu;
    const int lo = rem - hi * a.strideMu;
    base = hi * (J * a.strideMu) + lo;  // x_cb of the line's j = 0 site (parity p0)
    int c0[4];
    get_coords(c0, base, a.X, p0);      // c0[DIR] == 0
    faceIdx = ghost_face_index_on_face(c0, a.X, DIR);
  } else {
    ePR = a.X[0] >> 1;                  // checkerboard entries per x-row
    rowsPerGroup = kTileCols / ePR;
    active = col < rowsPerGroup * ePR;  // the host guarantees an even number of whole row groups
  }
  const Cplx<F> *ghostBase = reinterpret_cast<const Cplx<F> *>(a.ghost);
  // x_cb of (row-tile position pp, this lane's column)
  const int colClamped = active ? col : 0;
  auto rowTileXcb = [&](int pp) { return ((2 * cc + (pp >> 1)) * rowsPerGroup) * ePR + colClamped; };

  // source of every position this lane stages (fixed for the whole kernel): element offset + "is a ghost layer" bit
  int soff[PH];
  unsigned sghost = 0;
#pragma unroll
  for (int i = 0; i < PH; i++) {
    const int pp = (2 * i + half < NP) ? 2 * i + half : NP - 1;  // surplus slots re-read the last position
    soff[i] = 0;
    {
      if constexpr (DIR >= 1) {
        int j = (SIGN == MUGIQ_HIP_DISP_SIGN_PLUS) ? j0 + pp : j0 - a.kmax + pp;
        const int par = p0 ^ (j & 1);
        if ((j < 0 || j >= J) && a.partitioned) {
          const int layer = (j >= J) ? j - J : -j - 1;
          sghost |= 1u << i;
          soff[i] = (int)((int64_t)layer * 24 * a.faceCB + (int64_t)par * 12 * a.faceCB + comp_offset<ORDER>(wave, a.faceCB, faceIdx));
        } else {
          j = j < 0 ? j + J : (j >= J ? j - J : j);
          soff[i] = (int)((int64_t)par * a.parity_offset + comp_offset<ORDER>(wave, a.stride, base + j * a.strideMu));
        }
      } else {
        soff[i] = (int)((int64_t)(pp & 1) * a.parity_offset + comp_offset<ORDER>(wave, a.stride, rowTileXcb(pp)));
      }
    }
  }

  // ---- my site, and the LDS slots of my v(x) and of my shifted v(x +- k mu)
  int pmine, xmine, ppL, ppS, colS = col;
  if constexpr (DIR >= 1) {
    ppL = (SIGN == MUGIQ_HIP_DISP_SIGN_PLUS) ? wpos : a.kmax + wpos;
    ppS = (SIGN == MUGIQ_HIP_DISP_SIGN_PLUS) ? wpos + k : a.kmax + wpos - k;
    const int jmine = j0 + wpos;
    pmine = p0 ^ (jmine & 1);
    xmine = base + jmine * a.strideMu;
  } else {
    ppL = wpos;
    pmine = wpos & 1;
    xmine = rowTileXcb(wpos);
    int c[4];
    get_coords(c, xmine, a.X, pmine);
    int xs = c[0] + ((SIGN == MUGIQ_HIP_DISP_SIGN_PLUS) ? k : -k);
    xs %= a.X[0];
    if (xs < 0) xs += a.X[0];
    ppS = (wpos ^ (k & 1));                       // parity flips for odd k, same row group
    colS = (col / ePR) * ePR + (xs >> 1);        // same row, entry x0' / 2
  }

  // W_k(x) of this lane's site (both lane halves hold the same 3x3): in VGPRs.  It used to sit in LDS (9 of the 27 LDS
  // reads per eigenvector) because the registers were short; fp32 storage has the room (stage registers half as wide:
  // -7 % / -9 % per entry), fp64 gets it by prefetching two eigenvectors ahead instead of three (the kernel is bound by
  // vector issue, not by memory latency: -3.6 % column, -2 % row tile; three-deep AND W in registers spills, +35 %).
  Cplx<A> Wr[9];
  const bool unitW = a.E[slot] == nullptr;
  if (!unitW) {
    const Cplx<F> *e = reinterpret_cast<const Cplx<F> *>(a.E[slot]) + (int64_t)pmine * 12 * a.volumeCB + xmine;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const Cplx<F> t = e[(int64_t)(j * 3 + i) * a.volumeCB];
        Wr[i * 3 + j] = Cplx<A>{(A)t.re, (A)t.im};
      }
  } else {  // the ultra-local loop carried as a slot: k = 0, W = 1
#pragma unroll
    for (int i = 0; i < 9; i++) Wr[i] = Cplx<A>{A(i % 4 == 0 ? 1 : 0), A(0)};
  }
  Cplx<A> acc[8];  // acc[be*2 + a2], al = 2*half + a2
#pragma unroll
  for (int i = 0; i < 8; i++) acc[i] = Cplx<A>{A(0), A(0)};

  typedef F vec2 __attribute__((ext_vector_type(2)));
  // Three eigenvectors in flight ahead of the one being consumed.  Diagnostic builds on MI355X (tools/probes/
  // fused_tile_probes.patch; 48.48.24.24 fp64, 100 eigenvectors, 10.0 ms per entry): staging only 5.9 ms, compute only 7.0 ms (6.9 ms without the
  // barrier); SQ counters: the VALU pipe of a SIMD is ~64 % busy, 82 % of its instructions are the FMAs of the
  // mathematics -- the kernel is bound by vector issue, not by HBM or LDS.  Tried without gain: 6-deep prefetch for
  // fp32 storage (25 % slower), a pair-wise LDS layout so that fp32 reads are ds_read_b128 instead of ds_read2_b64 (no
  // change), column groups as the fastest workgroup index (4 % slower).
  constexpr int kDepth = sizeof(F) == 8 ? 2 : 3;  // eigenvectors in flight ahead of the one being consumed
  vec2 stageA[PH], stageB[PH], stageC[PH];
  // fetch this lane's share of eigenvector n_: plane `wave`, column `col`, positions pp = 2*i + half
#define MUGIQ_TILE_BODY(n_) static_cast<const Cplx<F> *>(as_constant(a.L)[n_])
#define MUGIQ_TILE_SIGMA(n_) as_constant(a.inv_sigma)[n_]
#define MUGIQ_TILE_FETCH(n_, stage) MUGIQ_TILE_FETCH_AT(MUGIQ_TILE_BODY(n_), n_, stage)
#define MUGIQ_TILE_FETCH_AT(bodyExpr_, n_, stage)                                                                      \
  {                                                                                                                    \
    const Cplx<F> *body_ = bodyExpr_;                                                                                  \
    const Cplx<F> *gh_ = ghostBase + (int64_t)(n_)*a.ghost_vec_stride;                                                 \
    _Pragma("unroll") for (int i = 0; i < PH; i++) { /* unconditional: every lane and slot has a valid source */       \
      const Cplx<F> *ptr_ = (((sghost >> i) & 1u) ? gh_ : body_) + soff[i];                                            \
      stage[i] = *as_global(reinterpret_cast<const vec2 *>(ptr_));                                                     \
    }                                                                                                                  \
  }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() lowers to `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier`,
// i.e. it also drains every global load in flight -- which would serialise the register prefetch of the next
// eigenvectors behind each barrier.  The stage registers are guarded by the compiler's own counted vmcnt waits.
#define MUGIQ_LDS_BARRIER()                              \
  {                                                      \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  \
    __builtin_amdgcn_s_barrier();                        \
    asm volatile("" ::: "memory");                       \
  }

// The arithmetic of one eigenvector (scaled by s_) on the tile buffer tile_.
#define MUGIQ_TILE_COMPUTE(tile_, s_)                                                                                \
  {                                                                                                                    \
    const Cplx<F> *tile = tile_;                                                                                        \
    const A s = s_;                                                                                                     \
    if (computes) {                                                                                                    \
      const Cplx<F> *tl = tile + tileIdx(ppL, 0, col);                                                                 \
      const Cplx<F> *ts = tile + tileIdx(ppS, half * 6, colS); /* spins 2*half, 2*half + 1 */                          \
      /* t[a2] = s * W * psi[2*half + a2]: each W element and each psi element is read from LDS once */               \
      Cplx<A> t0[3], t1[3];                                                                                            \
      if (unitW) { /* the carried ultra-local slot: W = 1 (wave-uniform branch) */                                     \
        _Pragma("unroll") for (int j = 0; j < 3; j++) {                                                                \
          const Cplx<F> w0 = ts[j * kCompStride], w1 = ts[(3 + j) * kCompStride];                                      \
          t0[j] = Cplx<A>{(A)w0.re, (A)w0.im};                                                                         \
          t1[j] = Cplx<A>{(A)w1.re, (A)w1.im};                                                                         \
        }                                                                                                              \
      } else {                                                                                                         \
      _Pragma("unroll") for (int i = 0; i < 3; i++) t0[i] = t1[i] = Cplx<A>{A(0), A(0)};                               \
      _Pragma("unroll") for (int j = 0; j < 3; j++) {                                                                  \
        const Cplx<F> w0 = ts[j * kCompStride], w1 = ts[(3 + j) * kCompStride];                                            \
        const Cplx<A> p0j{(A)w0.re, (A)w0.im}, p1j{(A)w1.re, (A)w1.im};                                                \
        _Pragma("unroll") for (int i = 0; i < 3; i++) {                                                                \
          const Cplx<A> w = Wr[i * 3 + j];                                                                             \
          cmadd(t0[i], w, p0j);                                                                                        \
          cmadd(t1[i], w, p1j);                                                                                        \
        }                                                                                                              \
      }                                                                                                                \
      }                                                                                                                \
      _Pragma("unroll") for (int i = 0; i < 3; i++) {                                                                  \
        t0[i] = Cplx<A>{s * t0[i].re, s * t0[i].im};                                                                   \
        t1[i] = Cplx<A>{s * t1[i].re, s * t1[i].im};                                                                   \
      }                                                                                                                \
      _Pragma("unroll") for (int be = 0; be < 4; be++) {                                                               \
        if (be == 2) __builtin_amdgcn_sched_barrier(0); /* bound how many LDS reads the scheduler hoists (VGPRs) */    \
        _Pragma("unroll") for (int c = 0; c < 3; c++) {                                                                \
          const Cplx<F> u = tl[(be * 3 + c) * kCompStride];                                                              \
          const Cplx<A> lv{(A)u.re, (A)u.im};                                                                          \
          cmadd_conj(acc[be * 2 + 0], lv, t0[c]);                                                                      \
          cmadd_conj(acc[be * 2 + 1], lv, t1[c]);                                                                      \
        }                                                                                                              \
      }                                                                                                                \
    }                                                                                                                  \
  }

// One step: eigenvector n_ is in LDS buffer n_ % 2; `stage` holds eigenvector n_+1 (fetched three steps ago).
// Commit n_+1 into the other buffer (everyone finished reading it before the barrier that ended the previous step),
// refill `stage` with n_+4, consume n_, one barrier.  (Little's law: with one workgroup per CU the bytes in flight
// are what the stage registers hold -- 3 x 42 KB per CU sustains ~6 TB/s at ~4 us loaded latency, 2 x 42 KB does not.)
#define MUGIQ_TILE_STEP(n_, stage, GUARD)                                                                              \
  {                                                                                                                    \
    const A sNow = sigPre;                                                                                             \
    const Cplx<F> *bodyNow = bodyPre;                                                                                  \
    { /* table look-ups of the NEXT step, issued first: they complete while this step waits for its staged loads, and  \
         being scalar loads in flight they would otherwise turn the first LDS wait of the arithmetic into lgkmcnt(0) */ \
      const int nb_ = (n_) + kDepth + 2 < a.nVec ? (n_) + kDepth + 2 : a.nVec - 1, ns_ = (n_) + 1 < a.nVec ? (n_) + 1 : a.nVec - 1; \
      bodyPre = MUGIQ_TILE_BODY(nb_);                                                                                  \
      sigPre = MUGIQ_TILE_SIGMA(ns_);                                                                                  \
    }                                                                                                                  \
    __builtin_amdgcn_sched_barrier(0); /* keep them first */                                                           \
    if (GUARD == 0 || (n_) + 1 < a.nVec) {                                                                             \
      Cplx<F> *nxt = tileBase + (size_t)(((n_) + 1) & 1) * tileElems;                                                  \
      _Pragma("unroll") for (int i = 0; i < PH; i++)                                                                   \
        nxt[((2 * i + half) * 12 + wave) * kTileCols + col] = Cplx<F>{stage[i].x, stage[i].y};                         \
    }                                                                                                                  \
    if (GUARD == 0 || (n_) + kDepth + 1 < a.nVec) MUGIQ_TILE_FETCH_AT(bodyNow, (n_) + kDepth + 1, stage)               \
    MUGIQ_TILE_COMPUTE(tileBase + (size_t)((n_) & 1) * tileElems, sNow)                                                \
    MUGIQ_LDS_BARRIER()                                                                                                \
  }

  if constexpr (GLDS) {
#if defined(__HIP_DEVICE_COMPILE__)  // (global_load_lds is a device-only builtin: the host pass of hipcc must not see it)
    typedef __attribute__((address_space(3))) void lds_void;
    // this lane's share of eigenvector n_ -> tile buffer buf_: PH transfers of 64 x 16 bytes (plane `wave`; the lane halves
    // carry positions 2 i | 2 i + 1)
#define MUGIQ_TILE_GLDS(bodyExpr_, n_, buf_)                                                                           \
  {                                                                                                                    \
    const Cplx<F> *body_ = bodyExpr_;                                                                                  \
    const Cplx<F> *gh_ = ghostBase + (int64_t)(n_)*a.ghost_vec_stride;                                                 \
    Cplx<F> *dst_ = (buf_) + (size_t)wave * 2 * kTileCols;                                                             \
    if (NW == 12 || wave < 12) { /* (waves 12-15 of the 16-wave form only compute) */                                  \
    _Pragma("unroll") for (int i = 0; i < PH; i++) {                                                                   \
      const Cplx<F> *ptr_ = (((sghost >> i) & 1u) ? gh_ : body_) + soff[i];                                            \
      __builtin_amdgcn_global_load_lds(as_global(reinterpret_cast<const vec2 *>(ptr_)),                                \
                                       (lds_void *)(dst_ + (size_t)i * 24 * kTileCols), 16, 0, 0);                     \
    }                                                                                                                  \
    }                                                                                                                  \
  }
    // One step: eigenvector n_ lands in buffer cur_ (own share: counted vmcnt wait; everybody's: the barrier, which also
    // says that nobody reads buffer nxt2_ = the one consumed in the previous step any more); eigenvector n_+2 is sent
    // there, n_+1 stays in flight, n_ is consumed.  One barrier per step, no register staging, no ds_write pass.
#define MUGIQ_TILE_GSTEP(n_, cur_, nxt2_, STEADY)                                                                      \
  {                                                                                                                    \
    const A sNow = sigPre;                                                                                             \
    const Cplx<F> *bodyNow = bodyPre;                                                                                  \
    if (STEADY || (n_) + 1 < a.nVec) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PH) : "memory");                         \
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                              \
    MUGIQ_LDS_BARRIER()                                                                                                \
    if (STEADY || (n_) + 2 < a.nVec) MUGIQ_TILE_GLDS(bodyNow, (n_) + 2, nxt2_)                                         \
    MUGIQ_TILE_COMPUTE(cur_, sNow)                                                                                     \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    { /* table look-ups of the next step (scalar loads: they complete under the tail of the arithmetic) */            \
      const int nb_ = (n_) + 3 < a.nVec ? (n_) + 3 : a.nVec - 1, ns_ = (n_) + 1 < a.nVec ? (n_) + 1 : a.nVec - 1;      \
      bodyPre = MUGIQ_TILE_BODY(nb_);                                                                                  \
      sigPre = MUGIQ_TILE_SIGMA(ns_);                                                                                  \
    }                                                                                                                  \
  }
    Cplx<F> *const buf0 = tileBase, *const buf1 = tileBase + tileElems, *const buf2 = tileBase + 2 * tileElems;
    const int last = a.nVec - 1;
    MUGIQ_TILE_GLDS(MUGIQ_TILE_BODY(0), 0, buf0)
    MUGIQ_TILE_GLDS(MUGIQ_TILE_BODY((1 < last ? 1 : last)), (1 < last ? 1 : last), buf1)  // (unconditional: known count in flight)
    const Cplx<F> *bodyPre = MUGIQ_TILE_BODY((2 < last ? 2 : last));
    A sigPre = MUGIQ_TILE_SIGMA(0);
    int n = 0;
    for (; n + 4 < a.nVec; n += 3) {
      MUGIQ_TILE_GSTEP(n, buf0, buf2, 1)
      MUGIQ_TILE_GSTEP(n + 1, buf1, buf0, 1)
      MUGIQ_TILE_GSTEP(n + 2, buf2, buf1, 1)
    }
    for (; n < a.nVec; n += 3) {  // n % 3 == 0 here
      MUGIQ_TILE_GSTEP(n, buf0, buf2, 0)
      if (n + 1 < a.nVec) MUGIQ_TILE_GSTEP(n + 1, buf1, buf0, 0)
      if (n + 2 < a.nVec) MUGIQ_TILE_GSTEP(n + 2, buf2, buf1, 0)
    }
#undef MUGIQ_TILE_GSTEP
#undef MUGIQ_TILE_GLDS
#endif
  } else {
    // prologue: eigenvector 0 -> LDS buffer 0; eigenvectors 1, 2, 3 in flight in stageA / stageB / stageC
    MUGIQ_TILE_FETCH(0, stageC)
#pragma unroll
    for (int i = 0; i < PH; i++) tileBase[((2 * i + half) * 12 + wave) * kTileCols + col] = Cplx<F>{stageC[i].x, stageC[i].y};
    // unconditional (clamped) so that the steady-state loop is entered with a KNOWN number of loads in flight: with
    // conditional prologue loads hipcc's wait-count pass waited for vmcnt(0) at the first commit of every loop iteration,
    // i.e. for the loads issued one step earlier -- the three-deep prefetch was in effect one deep
    {
      const int last = a.nVec - 1;
      MUGIQ_TILE_FETCH((1 < last ? 1 : last), stageA)
      MUGIQ_TILE_FETCH((2 < last ? 2 : last), stageB)
      if constexpr (kDepth == 3) MUGIQ_TILE_FETCH((3 < last ? 3 : last), stageC)
    }
    const Cplx<F> *bodyPre = MUGIQ_TILE_BODY(a.nVec > kDepth + 1 ? kDepth + 1 : a.nVec - 1);
    A sigPre = MUGIQ_TILE_SIGMA(0);
    MUGIQ_LDS_BARRIER()
    // steady state without any data-dependent branch (hipcc's wait-count pass turns every conditional load into a
    // conservative `vmcnt(0)`, which would serialise the prefetch), then a guarded tail
    int n = 0;
    for (; n + 2 * kDepth < a.nVec; n += kDepth) {
      MUGIQ_TILE_STEP(n, stageA, 0)
      MUGIQ_TILE_STEP(n + 1, stageB, 0)
      if constexpr (kDepth == 3) MUGIQ_TILE_STEP(n + 2, stageC, 0)
    }
    for (; n < a.nVec; n += kDepth) {
      MUGIQ_TILE_STEP(n, stageA, 1)
      if (n + 1 < a.nVec) MUGIQ_TILE_STEP(n + 1, stageB, 1)
      if constexpr (kDepth == 3)
        if (n + 2 < a.nVec) MUGIQ_TILE_STEP(n + 2, stageC, 1)
    }
  }
#undef MUGIQ_TILE_STEP
#undef MUGIQ_TILE_COMPUTE
  // ---- epilogue: the two lane halves of a wave hold complementary halves of the 4x4 colour-traced spin matrix of the
  // same 32 sites.  Exchange them with wavefront shuffles (lane ^ 32), then each half takes 8 of the 16 gamma traces.
  if (computes) {
    Cplx<A> full[16];
#pragma unroll
    for (int be = 0; be < 4; be++)
#pragma unroll
      for (int a2 = 0; a2 < 2; a2++) {
        const Cplx<A> mine = acc[be * 2 + a2];
        Cplx<A> theirs;
        theirs.re = __shfl_xor(mine.re, 32);
        theirs.im = __shfl_xor(mine.im, 32);
        // al = 2*half + a2 is mine, al = 2*(1-half) + a2 the partner's (selects keep the register indices static)
        full[be * 4 + a2] = half == 0 ? mine : theirs;
        full[be * 4 + 2 + a2] = half == 0 ? theirs : mine;
      }
    if (active) {
      Cplx<A> *out = a.out[slot];
      const int siteIdx = xmine + pmine * a.volumeCB;
      if (half == 0) trace_and_store_range<A, 0, 8>(out, full, 2 * a.volumeCB, siteIdx, a.overwrite != 0);
      else trace_and_store_range<A, 8, 16>(out, full, 2 * a.volumeCB, siteIdx, a.overwrite != 0);
    }
  }
}

#undef MUGIQ_TILE_FETCH
#undef MUGIQ_TILE_FETCH_AT
#undef MUGIQ_TILE_BODY
#undef MUGIQ_TILE_SIGMA

// global -> LDS staging with three buffers: fp64 FLOAT2 column tiles of at most 8 positions (MUGIQ_HIP_TILE_GLDS=0: off)
template <typename F, int ORDER> static bool tile_glds(int dir, int kmax) {
  bool glds = std::is_same<F, double>::value && ORDER == 2 && dir >= 1 && kTileTJ + kmax <= 8;
  if (const char *e = getenv("MUGIQ_HIP_TILE_GLDS")) glds = glds && atoi(e) != 0;
  return glds;
}

template <typename F, typename A, int ORDER> static int launch_tile(TileArgs<F, A> a, int dir, int sign, hipStream_t stream) {
  const int NP = dir >= 1 ? kTileTJ + a.kmax : kTileTJ;
  const int PHsel = NP <= 8 ? 4 : kTileMaxPos / 2;
  const bool glds = tile_glds<F, ORDER>(dir, a.kmax);
  const bool carry = a.nslot == kTileCarry;  // (tile_entry adds the fourth slot only where glds holds)
  const size_t tileBytes = (glds ? 3 : 2) * sizeof(Cplx<F>) * (size_t)(2 * PHsel) * 12 * kTileCols;  // padded positions
  const size_t shmem = tileBytes;  // the staging tiles
  a.tileBytes = (int)tileBytes;
  unsigned nblocks = ((a.numCols + kTileCols - 1) / kTileCols) * a.jtCount;
  if (dir == 0) {  // row tile: 2 groups of kTileCols/(X0/2) whole x-rows per workgroup
    const int ePR = a.X[0] / 2, rpg = kTileCols / ePR;
    nblocks = (unsigned)(a.volumeCB / ePR / rpg / 2);
  }
  // workgroup order, measured on MI355X (48.48.24.24, 100 eigenvectors): XCD-contiguous with the tiles along mu as the
  // fastest index is 2.5 % (fp64) / 1.5 % (fp32) faster than the plain order
  a.blockOrder = 2;
  if (const char *e = getenv("MUGIQ_HIP_TILE_ORDER")) a.blockOrder = atoi(e) & 3;
  if (nblocks % 8 != 0) a.blockOrder &= 1;
  const dim3 grid(nblocks), block(64 * (carry ? 16 : 12));
#define MUGIQ_TILE_LAUNCH(D, S, P, G, W)                                                                              \
  {                                                                                                                   \
    auto kern = tile_displaced_contract_kernel<F, A, ORDER, D, S, P, G, W>;                                           \
    if (shmem > 64 * 1024)                                                                                            \
      MUGIQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
    hipLaunchKernelGGL(kern, grid, block, shmem, stream, a);                                                          \
  }
#define MUGIQ_TILE_CASE(D, S)                                                                                         \
  case (D)*2 + (S):                                                                                                   \
    if constexpr (std::is_same<F, double>::value && ORDER == 2 && (D) >= 1) {                                         \
      if (glds) {                                                                                                     \
        if (carry) MUGIQ_TILE_LAUNCH(D, S, 4, true, 16) else MUGIQ_TILE_LAUNCH(D, S, 4, true, 12)                     \
        break;                                                                                                        \
      }                                                                                                               \
    }                                                                                                                 \
    if (NP <= 8) MUGIQ_TILE_LAUNCH(D, S, 4, false, 12) else MUGIQ_TILE_LAUNCH(D, S, kTileMaxPos / 2, false, 12)       \
    break;
  switch (dir * 2 + sign) {
    MUGIQ_TILE_CASE(0, 0) MUGIQ_TILE_CASE(0, 1) MUGIQ_TILE_CASE(1, 0) MUGIQ_TILE_CASE(1, 1)
    MUGIQ_TILE_CASE(2, 0) MUGIQ_TILE_CASE(2, 1) MUGIQ_TILE_CASE(3, 0) MUGIQ_TILE_CASE(3, 1)
  }
#undef MUGIQ_TILE_CASE
#undef MUGIQ_TILE_LAUNCH
  MUGIQ_CHECK_HIP(hipGetLastError());
  return MUGIQ_HIP_SUCCESS;
}

// Can the tiled kernel take this entry?  (otherwise the caller uses the first-generation streaming kernel)
bool tile_applicable(const MugiqHipSpinorField &ev, int dir, int kmax, int precision, int partitioned) {
  int mode = 1;  // MUGIQ_HIP_FUSED_TILE: 0 = streaming kernel only, 2 = column tile only (no row tile), default both
  if (const char *e = getenv("MUGIQ_HIP_FUSED_TILE")) mode = atoi(e);
  if (mode == 0 || (mode == 2 && dir == 0)) return false;
  if (2 * (int64_t)ev.parity_offset >= (1LL << 31)) return false;  // the kernel keeps 32-bit element offsets
  if (dir == 0) {  // row tile: whole x-rows in LDS, no ghost handling
    const int ePR = ev.X[0] / 2;
    if (partitioned || ePR > kTileCols || kmax >= ev.X[0]) return false;
    const int rpg = kTileCols / ePR;
    if ((ev.volumeCB / ePR) % (2 * rpg) != 0) return false;
    return true;
  }
  if (ev.X[dir] % kTileTJ != 0) return false;
  if (kmax > ev.X[dir]) return false;  // the staged window wraps at most once around the lattice
  if (kTileTJ + kmax > kTileMaxPos) return false;
  const int PHsel = kTileTJ + kmax <= 8 ? 4 : kTileMaxPos / 2;
  const size_t lds = (size_t)2 * 2 * precision * (2 * PHsel) * 12 * kTileCols;
  return lds <= 160 * 1024;
}

// ultra_d != NULL: also produce the ultra-local loop (k = 0, W = 1) into ultra_d, as a fourth slot of the 16-wave form; *carried
// says whether that was possible (fp64 FLOAT2 column tiles staged global -> LDS, at most three displaced slots)
template <typename F, typename A, int ORDER>
int tile_entry(void *loop_d, const MugiqHipSpinorField *ev, const double *sigma, int nVec, const void *const *E_d, const int *kvals,
               int nK, int dir, int sign, int partitioned, const void *ghost_d, int layers, int region, hipStream_t stream,
               void *ultra_d, int *carried) {
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
  TileArgs<F, A> a;
  const int64_t slot_stride = (int64_t)16 * 2 * ev[0].volumeCB;
  if (carried) *carried = 0;
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
  if (dir == 0) strideMu = 1;  // unused by the row tile (a step along x is half a checkerboard entry)
  a.strideMu = (int)strideMu;
  a.H = (int)(ev[0].volumeCB / (ev[0].X[dir] * strideMu));
  a.numCols = 2 * ev[0].volumeCB / ev[0].X[dir];
  a.nJT = dir == 0 ? 1 : ev[0].X[dir] / kTileTJ;  // (the row tile is not cut along x: one "tile", or X0 = 2 would launch nothing)
  a.overwrite = (region & MUGIQ_HIP_REGION_OVERWRITE) ? 1 : 0;
  region &= 0xff;
  // the ultra-local slot is produced for the WHOLE lattice or not at all: a launch restricted to the interior or the boundary
  // tiles would write only part of it (and none of it where that region is empty), so it is taken along with REGION_ALL only
  if (region != MUGIQ_HIP_REGION_ALL) ultra_d = nullptr;
  for (int k0 = 0; k0 < nK; k0 += kTileMaxSlots) {
    a.nslot = (nK - k0 < kTileMaxSlots) ? nK - k0 : kTileMaxSlots;
    a.kmax = 0;
    bool withUltra = false;
    for (int s = 0; s < kTileCarry; s++) {
      const int i = k0 + (s < a.nslot ? s : 0);
      a.E[s] = static_cast<const F *>(E_d[i]);
      a.k[s] = kvals[i];
      a.out[s] = static_cast<Cplx<A> *>(loop_d) + (int64_t)i * slot_stride;
      if (s < a.nslot && kvals[i] > a.kmax) a.kmax = kvals[i];
    }
    // room for it: a free slot of the 12-wave forms, or the fourth slot of the 16-wave form (global -> LDS staging only)
    if (ultra_d && k0 == 0 && nK <= kTileMaxSlots && (nK < kTileMaxSlots || tile_glds<F, ORDER>(dir, a.kmax))) {
      a.E[a.nslot] = nullptr;
      a.k[a.nslot] = 0;
      a.out[a.nslot] = static_cast<Cplx<A> *>(ultra_d);
      a.nslot++;
      withUltra = true;
    }
    // region 0: everything | 1: tiles whose shifted reads stay inside the local lattice | 2: tiles that read ghost layers
    a.jtBegin = 0;
    a.jtCount = a.nJT;
    if (region != MUGIQ_HIP_REGION_ALL && dir >= 1) {
      const int nb = partitioned ? std::min(a.nJT, (a.kmax + kTileTJ - 1) / kTileTJ) : 0;  // boundary tiles
      if (region == MUGIQ_HIP_REGION_INTERIOR) {
        a.jtBegin = sign == MUGIQ_HIP_DISP_SIGN_PLUS ? 0 : nb;
        a.jtCount = a.nJT - nb;
      } else {
        a.jtBegin = sign == MUGIQ_HIP_DISP_SIGN_PLUS ? a.nJT - nb : 0;
        a.jtCount = nb;
      }
    } else if (region == MUGIQ_HIP_REGION_BOUNDARY) {
      a.jtCount = 0;  // the row tile (x, never partitioned here) has no boundary part
    }
    if (a.jtCount > 0) {
      st = launch_tile<F, A, ORDER>(a, dir, sign, stream);
      if (st) return st;
      if (withUltra && carried) *carried = 1;  // only now: a launch that covers the whole lattice did write the slot
    }
  }
  return MUGIQ_HIP_SUCCESS;
}

#define MUGIQ_TILE_INST(F, A, O)                                                                                                 \
  template int tile_entry<F, A, O>(void *, const MugiqHipSpinorField *, const double *, int, const void *const *, const int *, int, \
                                   int, int, int, const void *, int, int, hipStream_t, void *, int *);
MUGIQ_TILE_INST(double, double, 2)
MUGIQ_TILE_INST(double, double, 4)
MUGIQ_TILE_INST(float, float, 2)
MUGIQ_TILE_INST(float, float, 4)
MUGIQ_TILE_INST(float, double, 2)
MUGIQ_TILE_INST(float, double, 4)
#undef MUGIQ_TILE_INST

}  // namespace mugiq
