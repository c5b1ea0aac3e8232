// a10: momentum projection  dataMom[M x N] = dataPosMP[M x K] * phaseMatrix[K x N]  (column-major),
// M = locT*nData, K = locV3, N = Nmom -- the cublasZgemm / cublasCgemm call of lib/loop_mugiq.cpp:363-378.
//
// The product is skinny (N = number of momenta, typically tens to ~100; M a few hundred to a few
// thousand; K = the local spatial volume), so it is bound by streaming A once per N-tile.  One lane owns
// one row m (A is read coalesced along m, 16 B per lane), a tile of NT momenta is accumulated in
// registers, the phase tile is broadcast from LDS, and K is split over workgroups; partial sums are
// combined in a fixed order by a second kernel (deterministic, no atomics).
#include "internal.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace mugiq {

constexpr int kMpBlock = 256;
constexpr int kMpNT = 8;    // momenta per register tile
constexpr int kMpKC = 64;   // k-values staged in LDS per step

struct MomProjGeom {
  int M;
  int N;
  long long K;
  long long kChunk;  // k-range per split
  int nSplit;
  int RT;            // rows per lane (1, 2 or 4)
};

// RT rows per lane (m, m + 256, ...): every phase read from LDS feeds RT complex multiply-adds.  With one row per lane
// the broadcast LDS reads (one ds_read_b128 per 4 FMAs, shared by the four SIMDs of a CU) cost as much as the
// arithmetic; wider momentum tiles do not change that ratio (measured: NT = 32 was 40 % slower than NT = 8).
template <typename F, int RT>
__global__ __launch_bounds__(kMpBlock) void momproj_partial_kernel(Cplx<F> *part, const Cplx<F> *A, const Cplx<F> *B,
                                                                   MomProjGeom g) {
  __shared__ Cplx<F> Bs[kMpKC][kMpNT];
  const int m0 = blockIdx.x * (kMpBlock * RT) + threadIdx.x;
  const int n0 = blockIdx.y * kMpNT;
  const int split = blockIdx.z;
  const long long kBeg = split * g.kChunk;
  const long long kEnd = (kBeg + g.kChunk < g.K) ? kBeg + g.kChunk : g.K;

  Cplx<F> acc[RT][kMpNT];
#pragma unroll
  for (int r = 0; r < RT; r++)
#pragma unroll
    for (int n = 0; n < kMpNT; n++) acc[r][n] = Cplx<F>{F(0), F(0)};
  // rows beyond M re-read the last row (valid address, result dropped): no divergence around the loads
  int mr[RT];
#pragma unroll
  for (int r = 0; r < RT; r++) mr[r] = (m0 + r * kMpBlock < g.M) ? m0 + r * kMpBlock : g.M - 1;

  for (long long k0 = kBeg; k0 < kEnd; k0 += kMpKC) {
    __syncthreads();
    for (int i = threadIdx.x; i < kMpKC * kMpNT; i += kMpBlock) {
      const int n = i / kMpKC, kk = i - n * kMpKC;  // consecutive lanes -> consecutive k (B is column-major K x N)
      Cplx<F> b{F(0), F(0)};
      if (k0 + kk < kEnd && n0 + n < g.N) b = B[(k0 + kk) + g.K * (n0 + n)];
      Bs[kk][n] = b;
    }
    __syncthreads();
    const int kn = (int)((kEnd - k0 < kMpKC) ? (kEnd - k0) : kMpKC);
    const Cplx<F> *a = A + (long long)g.M * k0;
#pragma unroll 2
    for (int kk = 0; kk < kn; kk++) {
      Cplx<F> av[RT];
#pragma unroll
      for (int r = 0; r < RT; r++) av[r] = a[(long long)g.M * kk + mr[r]];
#pragma unroll
      for (int n = 0; n < kMpNT; n++) {
        const Cplx<F> b = Bs[kk][n];
#pragma unroll
        for (int r = 0; r < RT; r++) cmadd(acc[r][n], av[r], b);
      }
    }
  }
#pragma unroll
  for (int r = 0; r < RT; r++) {
    const int m = m0 + r * kMpBlock;
    if (m < g.M) {
#pragma unroll
      for (int n = 0; n < kMpNT; n++)
        if (n0 + n < g.N) part[((long long)split * g.N + (n0 + n)) * g.M + m] = acc[r][n];
    }
  }
}

template <typename F>
__global__ __launch_bounds__(kMpBlock) void momproj_reduce_kernel(Cplx<F> *C, const Cplx<F> *part, long long MN, int nSplit) {
  const long long i = (long long)blockIdx.x * kMpBlock + threadIdx.x;
  if (i >= MN) return;
  Cplx<F> s{F(0), F(0)};
  for (int p = 0; p < nSplit; p++) {  // fixed order: reproducible run to run
    const Cplx<F> v = part[(long long)p * MN + i];
    s.re += v.re;
    s.im += v.im;
  }
  C[i] = s;
}

static void choose_split(int M, int N, long long K, MomProjGeom &g) {
  g.M = M;
  g.N = N;
  g.K = K;
  g.RT = M >= 4 * kMpBlock ? 4 : (M >= 2 * kMpBlock ? 2 : 1);
  const int rowsPerWg = kMpBlock * g.RT;
  const long long tiles = (long long)((M + rowsPerWg - 1) / rowsPerWg) * ((N + kMpNT - 1) / kMpNT);
  long long want = (2048 + tiles - 1) / tiles;  // ~8 workgroups per CU
  const long long maxSplit = (K + 4 * kMpKC - 1) / (4 * kMpKC);  // keep >= 256 k-values per split
  if (want > maxSplit) want = maxSplit;
  if (want < 1) want = 1;
  long long chunk = (K + want - 1) / want;
  chunk = (chunk + kMpKC - 1) / kMpKC * kMpKC;
  g.kChunk = chunk;
  g.nSplit = (int)((K + chunk - 1) / chunk);
}

template <typename F>
static int launch_momproj(void *C, const void *A, const void *B, const MomProjGeom &g, void *ws, hipStream_t stream) {
  const int rowsPerWg = kMpBlock * g.RT;
  const dim3 grid((g.M + rowsPerWg - 1) / rowsPerWg, (g.N + kMpNT - 1) / kMpNT, g.nSplit);
  Cplx<F> *part = g.nSplit == 1 ? static_cast<Cplx<F> *>(C) : static_cast<Cplx<F> *>(ws);
  const Cplx<F> *Ap = static_cast<const Cplx<F> *>(A), *Bp = static_cast<const Cplx<F> *>(B);
  if (g.RT == 4) hipLaunchKernelGGL((momproj_partial_kernel<F, 4>), grid, dim3(kMpBlock), 0, stream, part, Ap, Bp, g);
  else if (g.RT == 2) hipLaunchKernelGGL((momproj_partial_kernel<F, 2>), grid, dim3(kMpBlock), 0, stream, part, Ap, Bp, g);
  else hipLaunchKernelGGL((momproj_partial_kernel<F, 1>), grid, dim3(kMpBlock), 0, stream, part, Ap, Bp, g);
  MUGIQ_CHECK_HIP(hipGetLastError());
  if (g.nSplit > 1) {
    const long long MN = (long long)g.M * g.N;
    hipLaunchKernelGGL((momproj_reduce_kernel<F>), dim3((unsigned)((MN + kMpBlock - 1) / kMpBlock)), dim3(kMpBlock), 0, stream,
                       static_cast<Cplx<F> *>(C), part, MN, g.nSplit);
    MUGIQ_CHECK_HIP(hipGetLastError());
  }
  return MUGIQ_HIP_SUCCESS;
}



// ---- separable form of the same projection -------------------------------------------------------------------------
// The phase of lib/mugiq_util_kernels.cu:3-35 factorises, exp(i s 2 pi p.x/L) = f_x(p_x, x) f_y(p_y, y) f_z(p_z, z), so the
// M x K x N product can be taken one spatial direction at a time:
//   step x:  T1[z,y,ipx,m]  = sum_x  A[z,y,x,m]        f_x(px,x)      for the DISTINCT p_x of the momentum list
//   step y:  T2[z,ipxy,m]   = sum_y  T1[z,y,ipx,m]     f_y(py,y)      for the distinct (p_x,p_y) pairs
//   step z:  C[m,n]         = sum_z  T2[z,ipxy(n),m]   f_z(pz,z)      for the momenta themselves
// 123 momenta with p^2 <= 9 have 7 distinct p_x: step x costs 7 complex multiply-adds per element of A instead of 123
// and is bound by reading A once (the bound SURVEY.md section 8 names for this row); the later steps work on arrays
// Lx and Lx*Ly times smaller.  Same sums in a different order: agrees with the dense product to rounding.
constexpr int kDftJT = 8;  // outputs per lane

template <typename F> struct DftStepArgs {
  const Cplx<F> *in;   // [outer][Lsum][innerIn][M]
  Cplx<F> *out;        // [outer][nOut][M]
  const Cplx<F> *ph;   // [rows][Lsum]
  const int *groupSrc, *groupFirst, *groupCount;  // per group: source index in `innerIn`, first output, #outputs (<= kDftJT)
  const int *outRow, *outPos;                     // per output: phase row, position in the out array
  int M, Lsum, innerIn, nOut;
  // last step of a projection over a SUBSET of the loop slots: row m = r + rowsPerSlot * c of the compact arrays goes to row
  // r + rowsPerSlot * rowSlot[c] of an output with Mout rows per momentum (the full [locT*16*nLoop] layout); NULL: m itself
  const int *rowSlot;
  int rowsPerSlot, Mout;
};

template <typename F> __global__ __launch_bounds__(256) void partial_dft_kernel(DftStepArgs<F> a) {
  extern __shared__ __align__(16) unsigned char smem[];
  Cplx<F> *phs = reinterpret_cast<Cplx<F> *>(smem);  // [kDftJT][Lsum], zero rows beyond the group's outputs
  const int g = blockIdx.y, o = blockIdx.z;
  const int first = a.groupFirst[g], cnt = a.groupCount[g], src = a.groupSrc[g];
  for (int i = threadIdx.x; i < kDftJT * a.Lsum; i += 256) {
    const int j = i / a.Lsum, sI = i - j * a.Lsum;
    phs[i] = j < cnt ? a.ph[(int64_t)a.outRow[first + j] * a.Lsum + sI] : Cplx<F>{F(0), F(0)};
  }
  __syncthreads();
  const int m = blockIdx.x * 256 + threadIdx.x;
  if (m >= a.M) return;
  Cplx<F> acc[kDftJT];
#pragma unroll
  for (int j = 0; j < kDftJT; j++) acc[j] = Cplx<F>{F(0), F(0)};
  typedef F vec2 __attribute__((ext_vector_type(2)));
  const Cplx<F> *p = a.in + ((int64_t)o * a.Lsum * a.innerIn + src) * a.M + m;
  const int64_t step = (int64_t)a.innerIn * a.M;
#pragma unroll 4
  for (int sI = 0; sI < a.Lsum; sI++) {
    const vec2 u = __builtin_nontemporal_load(as_global(reinterpret_cast<const vec2 *>(p + step * sI)));
    const Cplx<F> v{u.x, u.y};
#pragma unroll
    for (int j = 0; j < kDftJT; j++) cmadd(acc[j], v, phs[j * a.Lsum + sI]);
  }
  const int c = a.rowSlot ? m / a.rowsPerSlot : 0;
  const int mo = a.rowSlot ? m - c * a.rowsPerSlot + a.rowsPerSlot * a.rowSlot[c] : m;
#pragma unroll
  for (int j = 0; j < kDftJT; j++)
    if (j < cnt) a.out[((int64_t)o * a.nOut + a.outPos[first + j]) * a.Mout + mo] = acc[j];
}

// Step x taken straight from the even-odd position-space buffer: the reorder of convertIdxOrder_mapGamma
// (lib/mugiq_util_kernels.cu:59-99: tid = x_cb + volumeCB*parity -> time-major rows, gamma -> g5 gamma with sign) and the sum
// over x in one pass, so the reordered copy (as large as the loop data itself) is neither written nor read.
//   T1[(z*Ly + y)*nPx + ipx][t + Lt*idataTo] = s(ig) * sum_x dataPos[idataFrom][x,y,z,t] * f_x(ipx, x)
// One workgroup per (pair of y rows, z, idataFrom): the rows x = 0..Lx-1 of both y and all t land in LDS (runs of Lx
// checkerboard entries per parity and t), then lane <-> (y, ipx, t) sums its row.
constexpr int kEoYG = 2;  // y rows per workgroup (their checkerboard entries are adjacent)

template <typename F> struct EoDftArgs {
  const Cplx<F> *in;   // dataPos [nData][2*volumeCB]
  Cplx<F> *out;        // T1 [Lz*Ly][nPx][M]
  const Cplx<F> *ph;   // [Lx][nPxPad]: the phases of one x side by side (nPxPad = nPx rounded up to kEoCh, zero-filled)
  int X[4];
  int volumeCB, nPx, nPxPad, M;
  int tilesPerWg;      // pipelined kernel: consecutive y pairs per workgroup
  int redOffset;       // complex elements from the tile to the partial-sum area (0: the tile itself, single pass)
  int tChunk;          // general kernel: time slices per tile (= X[3] unless the (x, t) rows of a y pair do not fit the LDS)
  const int *slotMap;  // NULL: every loop slot; else compact slot c of the output <- slot slotMap[c] of dataPos (a subset of the slots)
};

// blockIdx.z = 16 * (compact slot) + ig of the OUTPUT: which channel of dataPos it is read from, and which row block it feeds
template <typename F> __device__ inline void eo_dft_channels(const EoDftArgs<F> &a, int &ig, int &idataFrom, int &idataTo) {
  const int ia = blockIdx.z, slotC = ia >> 4;
  ig = ia & 15;
  const int slotFrom = a.slotMap ? as_constant(a.slotMap)[slotC] : slotC;
  idataFrom = slotFrom * 16 + ig;
  idataTo = (15 - ig) + 16 * slotC;  // gammaMap->index[ig] + N_GAMMA_*iL   :89
}

// Measured on MI355X (48.48.24.24 fp64, 25 slots, 7 distinct p_x): the first version of this kernel let lane <-> (y, p_x, t)
// read a data element AND a phase from LDS for every complex multiply-add -- 32 LDS bytes per 4 FMAs, four times what the
// LDS pipe of a CU delivers at the fp64 FMA rate: 3.45 ms to read 8.49 GB (2.5 TB/s, 31 % of HBM).  Now lane <-> row (y, t),
// wave <-> a class of x values (x = wave, wave + 4, ...): every staged element is read from LDS exactly once and feeds the
// multiply-adds of ALL p_x; the phases of a wave's x are wave-uniform, so they come through the scalar cache (s_load) and
// enter the FMAs as SGPR operands.  The four partial sums of a row are combined through LDS in a fixed order.
constexpr int kEoCh = 8;  // p_x values per pass (accumulators per lane)

// the sums of one staged tile (rows y0, y0 + 1 of plane z, all t) and their store; ends with the LDS still being read
// (t0, tn: the time slices [t0, t0 + tn) the tile holds -- all of them except in the chunked general kernel)
template <typename F> __device__ inline void eo_dft_x_sums(const EoDftArgs<F> &a, Cplx<F> *tile, int y0, int z, int idataTo, int t0, int tn) {
  typedef F vec2 __attribute__((ext_vector_type(2)));
  const int Lx = a.X[0], Ly = a.X[1], Lt = a.X[3], ld = Lx + 1;
  const int rows = kEoYG * tn, lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const Cplx<F> *phc = a.ph;
  for (int p0 = 0; p0 < a.nPx; p0 += kEoCh) {
    const int cnt = a.nPx - p0 < kEoCh ? a.nPx - p0 : kEoCh;
    for (int r0 = 0; r0 < rows; r0 += 64) {
      const int row = r0 + lane < rows ? r0 + lane : rows - 1;
      const Cplx<F> *rp = tile + row * ld;
      Cplx<F> acc[kEoCh];
#pragma unroll
      for (int j = 0; j < kEoCh; j++) acc[j] = Cplx<F>{F(0), F(0)};
      // wave-uniform address: the kEoCh phases of an x in one run of scalar loads (zero beyond the list); the operands of
      // the next x are requested before the multiply-adds of this one (clamped index: no conditional load)
      const MUGIQ_CONSTANT vec2 *fbase = as_constant(reinterpret_cast<const vec2 *>(phc)) + p0;
      Cplx<F> vn = rp[wave < Lx ? wave : Lx - 1];
      vec2 fn[kEoCh];
#pragma unroll
      for (int j = 0; j < kEoCh; j++) fn[j] = fbase[(wave < Lx ? wave : Lx - 1) * a.nPxPad + j];
      for (int x = wave; x < Lx; x += 4) {
        const Cplx<F> v = vn;
        vec2 fv[kEoCh];
#pragma unroll
        for (int j = 0; j < kEoCh; j++) fv[j] = fn[j];
        const int xn = x + 4 < Lx ? x + 4 : x;
        vn = rp[xn];
#pragma unroll
        for (int j = 0; j < kEoCh; j++) fn[j] = fbase[xn * a.nPxPad + j];
#pragma unroll
        for (int j = 0; j < kEoCh; j++) cmadd(acc[j], v, Cplx<F>{fv[j].x, fv[j].y});
      }
      __syncthreads();  // everyone is done reading the tile rows (single pass) / the previous pass's partial sums
      Cplx<F> *red = tile + a.redOffset;  // [kEoCh][4 waves][64 lanes]: consecutive lanes, consecutive 16-byte slots
#pragma unroll
      for (int j = 0; j < kEoCh; j++) red[(j * 4 + wave) * 64 + lane] = acc[j];
      __syncthreads();
      for (int o = threadIdx.x; o < 64 * cnt; o += 256) {
        const int ln = o % 64, j = o / 64, rr = r0 + ln;
        if (rr < rows) {
          Cplx<F> s = red[(j * 4 + 0) * 64 + ln];
#pragma unroll
          for (int w = 1; w < 4; w++) {  // fixed order: x classes 0, 1, 2, 3
            const Cplx<F> q = red[(j * 4 + w) * 64 + ln];
            s.re += q.re;
            s.im += q.im;
          }
          const int yy = rr / tn, t = t0 + rr - yy * tn;
          a.out[((int64_t)(z * Ly + y0 + yy) * a.nPx + p0 + j) * a.M + t + Lt * idataTo] = s;
        }
      }
    }
  }
}

#define MUGIQ_EO_LOAD(dst_, ptr_) dst_ = __builtin_nontemporal_load(as_global(reinterpret_cast<const vec2 *>(ptr_)));
constexpr int kEoLd = 12;  // loads per lane that cover a whole tile in the pipelined kernel (48 x 24: 12)

// Staging of the pipelined kernels (a run fits a wave and kEoLd loads per lane cover a tile): lane <-> entry of a run, several
// runs side by side when a run is shorter than half a wave; load slot q of wave wv <-> run group wv + 4 q.
template <typename F> struct EoStager {
  typedef F vec2 __attribute__((ext_vector_type(2)));
  int wv, rpw, groups, nRuns, tStride, base, dstBase, parBase, tpLane, ld, run, volumeCB;
  vec2 u[kEoLd];

  // a tile holds the time slices [t0, t0 + a.tChunk) of a y pair (all of them when tChunk == Lt)
  __device__ inline void init(const EoDftArgs<F> &a, int z) {
    const int Lx = a.X[0], Ly = a.X[1], Lz = a.X[2], Lt = a.tChunk;
    ld = Lx + 1;
    const int hx = Lx >> 1;
    run = kEoYG * hx;
    nRuns = 2 * Lt;
    const int ln = threadIdx.x & 63;
    wv = threadIdx.x >> 6;
    rpw = 64 / run;  // runs a wave fetches side by side (run <= 64)
    const int sub = ln / run;
    groups = (nRuns + rpw - 1) / rpw;
    const bool idle = sub >= rpw;
    const int rr = idle ? 0 : ln - sub * run;
    const int yy = rr / hx, xh = rr - yy * hx;
    tStride = (Lz * Ly * Lx) >> 1;  // elements; the host guarantees 2 * volumeCB < 2^31
    base = ((z * Ly * Lx) >> 1) + rr;
    dstBase = yy * Lt * ld + 2 * xh;
    parBase = yy + z;  // y0 is even: the parity of y is that of yy
    tpLane = idle ? 0 : sub;
    volumeCB = a.volumeCB;
  }
  // (t, parity) of load slot q: recomputed where needed (a dozen integer instructions) instead of kept in 36 registers -- with
  // the stage registers live across the sums the kernel must stay within 128 VGPRs to keep four waves per SIMD; the empty
  // asm keeps hipcc from hoisting the arithmetic out of the tile loop again
  __device__ inline void slot(int q, int &pty, int &t) const {
    const int grp = wv + 4 * q < groups ? wv + 4 * q : groups - 1;  // surplus slots repeat the last group (same value, same place)
    int tp = grp * rpw + tpLane;
    tp = tp < nRuns ? tp : nRuns - 1;
    pty = tp & 1;
    t = tp >> 1;
  }
  __device__ inline void fetch(const Cplx<F> *src, int ty, int t0 = 0) {
    const Cplx<F> *sp = src + (int64_t)ty * run + base + (int64_t)t0 * tStride;  // rows y0 = kEoYG * ty: run = kEoYG * Lx / 2 entries further
    asm volatile("" : "+v"(tpLane));
#pragma unroll
    for (int q = 0; q < kEoLd; q++) {
      int pty_, t_;
      slot(q, pty_, t_);
      MUGIQ_EO_LOAD(u[q], sp + (pty_ * volumeCB + t_ * tStride))
    }
  }
  __device__ inline void commit(Cplx<F> *tile, F sign, int t0 = 0) {
    asm volatile("" : "+v"(tpLane));
#pragma unroll
    for (int q = 0; q < kEoLd; q++) {
      int pty_, t_;
      slot(q, pty_, t_);
      tile[dstBase + t_ * ld + ((pty_ - (parBase + t0 + t_)) & 1)] = Cplx<F>{sign * u[q].x, sign * u[q].y};
    }
  }
};

// Pipelined form: a workgroup walks `tilesPerWg` consecutive y pairs of its (z, idataFrom); the loads of the NEXT tile are issued
// into registers as soon as the current one has been committed to LDS, so they travel while the sums of the current tile
// are taken.
template <typename F> __global__ __launch_bounds__(256) void eo_dft_x_pipelined_kernel(EoDftArgs<F> a) {
  extern __shared__ __align__(16) unsigned char smem[];
  Cplx<F> *tile = reinterpret_cast<Cplx<F> *>(smem);
  const int z = blockIdx.y;
  int ig, idataFrom, idataTo;
  eo_dft_channels(a, ig, idataFrom, idataTo);
  const F sign = (F)kGammaMapSign[ig];               // gammaMap->sign[ig]                    :93
  const Cplx<F> *src = a.in + (int64_t)idataFrom * 2 * a.volumeCB;
  EoStager<F> st;
  st.init(a, z);
  const int nCh = a.X[3] / a.tChunk, tiles = (a.X[1] / kEoYG) * nCh;  // tile tt = (y pair tt / nCh, time chunk tt % nCh)
  const int tb = blockIdx.x * a.tilesPerWg, te = tb + a.tilesPerWg < tiles ? tb + a.tilesPerWg : tiles;
  if (tb < te) st.fetch(src, tb / nCh, (tb % nCh) * a.tChunk);
  for (int tt = tb; tt < te; tt++) {
    const int t0 = (tt % nCh) * a.tChunk;
    __syncthreads();  // the previous tile's sums and stores are done with the LDS
    st.commit(tile, sign, t0);
    if (tt + 1 < te) st.fetch(src, (tt + 1) / nCh, ((tt + 1) % nCh) * a.tChunk);
    __syncthreads();
    eo_dft_x_sums(a, tile, (tt / nCh) * kEoYG, z, idataTo, t0, a.tChunk);
  }
}

// The same with the sums on the matrix pipe (fp64 only; MUGIQ_HIP_EO_MFMA selects it).  Per tile the sums are a small real
// GEMM, out'[row][n'] = sum_k' A'[row][k'] B'[k'][n'], with the complex structure unfolded: k' = 2 x + (re | im of the
// staged element), n' = 2 j + (re | im of the sum for p_x number j), B'[2x][2j] = Re f, B'[2x+1][2j] = -Im f,
// B'[2x][2j+1] = Im f, B'[2x+1][2j+1] = Re f.  v_mfma_f64_16x16x4_f64 takes A[i = lane & 15][k = lane >> 4] and
// B[k = lane >> 4][j = lane & 15], one double per lane, and returns C[row = (lane >> 4) + 4 r][col = lane & 15] in register r
// (checked with exact integers by tools/probes/mfma_f64_probe.hip).  Wave w owns the x range [w Lx/4, (w+1) Lx/4): its B
// fragments (the phases) are loaded ONCE into Lx/8 registers and stay there for the whole kernel -- no phase traffic at
// all in the tile loop; an A fragment is one ds_read_b64 of the tile, every element of which is read exactly once.  The four
// partial products of a row are combined through LDS in a fixed order.  On gfx950 an fp64 MFMA has the rate of the vector
// FMAs and shares their pipe (profiles/r02_mfma_f64_probe.json): what it saves here is operand delivery, not arithmetic.
constexpr int kEoMfmaLdC = 18; // padded row of the partial-product area (doubles)
template <int NKS, int MB> __global__ __launch_bounds__(256) void eo_dft_x_mfma_kernel(EoDftArgs<double> a) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  extern __shared__ __align__(16) unsigned char smem[];
  Cplx<double> *tile = reinterpret_cast<Cplx<double> *>(smem);
  const int Lx = a.X[0], Ly = a.X[1], Lt = a.X[3], tn = a.tChunk, ld = Lx + 1;
  const int z = blockIdx.y;
  int ig, idataFrom, idataTo;
  eo_dft_channels(a, ig, idataFrom, idataTo);
  const double sign = (double)kGammaMapSign[ig];
  const Cplx<double> *src = a.in + (int64_t)idataFrom * 2 * a.volumeCB;
  EoStager<double> st;
  st.init(a, z);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int rows = kEoYG * tn, xw = wave * (Lx >> 2);  // NKS = Lx / 8 k-steps per wave, MB = ceil(rows / 16) row blocks: compile
                                                       // time, so the tile loop is straight-line code (with run-time bounds
                                                       // every MFMA sat in its own basic block behind its own LDS wait)
  const int kq = lane >> 4, col = lane & 15;
  // B fragments of this wave
  double bfrag[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ks++) {
    const int x = xw + 2 * ks + (kq >> 1), j = col >> 1;
    const Cplx<double> f = a.ph[x * a.nPxPad + (j < a.nPx ? j : 0)];
    const double v = (col & 1) == 0 ? ((kq & 1) == 0 ? f.re : -f.im) : ((kq & 1) == 0 ? f.im : f.re);
    bfrag[ks] = j < a.nPx ? v : 0.0;
  }
  // A fragment addresses (doubles in the tile): row block mb, k-step ks -> ((row * ld + xw + 2 ks + (kq >> 1)) * 2 + (kq & 1))
  int aoff[MB];
#pragma unroll
  for (int mb = 0; mb < MB; mb++) {
    const int row = 16 * mb + col < rows ? 16 * mb + col : rows - 1;
    aoff[mb] = (row * ld + xw + (kq >> 1)) * 2 + (kq & 1);
  }
  const int nCh = Lt / tn, tiles = (Ly / kEoYG) * nCh;  // tile tt = (y pair tt / nCh, time chunk tt % nCh)
  const int tb = blockIdx.x * a.tilesPerWg, te = tb + a.tilesPerWg < tiles ? tb + a.tilesPerWg : tiles;
  if (tb < te) st.fetch(src, tb / nCh, (tb % nCh) * tn);
  for (int tt = tb; tt < te; tt++) {
    const int ty = tt / nCh, t0 = (tt % nCh) * tn;
    __syncthreads();  // the previous tile's partial products have been consumed
    st.commit(tile, sign, t0);
    if (tt + 1 < te) st.fetch(src, (tt + 1) / nCh, ((tt + 1) % nCh) * tn);
    __syncthreads();
    const double *td = reinterpret_cast<const double *>(tile);
    d4 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; mb++) acc[mb] = d4{0, 0, 0, 0};
    double af[NKS][MB];  // all A fragments of the tile first (NKS * MB LDS reads in flight), then the products
#pragma unroll
    for (int ks = 0; ks < NKS; ks++)
#pragma unroll
      for (int mb = 0; mb < MB; mb++) af[ks][mb] = td[aoff[mb] + 4 * ks];
#pragma unroll
    for (int ks = 0; ks < NKS; ks++)
#pragma unroll
      for (int mb = 0; mb < MB; mb++) acc[mb] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[ks][mb], bfrag[ks], acc[mb], 0, 0, 0);
    __syncthreads();  // every wave is done reading the tile
    double *red = reinterpret_cast<double *>(tile + a.redOffset);  // [4 waves][64 rows][kEoMfmaLdC]
#pragma unroll
    for (int mb = 0; mb < MB; mb++)
#pragma unroll
      for (int r = 0; r < 4; r++) red[((wave * 64) + 16 * mb + kq + 4 * r) * kEoMfmaLdC + col] = acc[mb][r];
    __syncthreads();
    const int y0 = ty * kEoYG;
    for (int o = threadIdx.x; o < rows * a.nPx; o += 256) {
      const int t = o % tn, rest = o / tn, j = rest % a.nPx, yy = rest / a.nPx, row = yy * tn + t;
      Cplx<double> s{0.0, 0.0};
#pragma unroll
      for (int w = 0; w < 4; w++) {  // fixed order: x ranges 0, 1, 2, 3
        const double *q = red + ((w * 64) + row) * kEoMfmaLdC + 2 * j;
        s.re += q[0];
        s.im += q[1];
      }
      a.out[((int64_t)(z * Ly + y0 + yy) * a.nPx + j) * a.M + t0 + t + Lt * idataTo] = s;
    }
  }
}

// General form (any Lx, Lt): one tile per workgroup, loads in batches of kEoLd per lane.  A tile holds the time slices
// [t0, t0 + tn) of a y pair: all of them when they fit the LDS, else chunks of a.tChunk (48^3 x 96 on one GPU: 3 x 32) -- the
// sums of different t are independent, so the chunks are independent workgroups.
template <typename F> __global__ __launch_bounds__(256) void eo_dft_x_kernel(EoDftArgs<F> a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int Lx = a.X[0], Ly = a.X[1], Lz = a.X[2], Lt = a.X[3], ld = Lx + 1;  // padded rows: lanes walk the rows at a fixed x
  Cplx<F> *tile = reinterpret_cast<Cplx<F> *>(smem);                         // [kEoYG * tn rows][Lx + 1]; later the partial sums
  const int tilesY = Ly / kEoYG, tc = blockIdx.x / tilesY;
  const int t0 = tc * a.tChunk, tn = Lt - t0 < a.tChunk ? Lt - t0 : a.tChunk;
  const int y0 = (blockIdx.x - tc * tilesY) * kEoYG, z = blockIdx.y;
  int ig, idataFrom, idataTo;
  eo_dft_channels(a, ig, idataFrom, idataTo);
  const F sign = (F)kGammaMapSign[ig];               // gammaMap->sign[ig]                    :93
  const Cplx<F> *src = a.in + (int64_t)idataFrom * 2 * a.volumeCB;
  // per (t, parity): the kEoYG * Lx/2 checkerboard entries of rows y0, y0+1 are contiguous ("run"); a wave fetches 64-entry
  // pieces of runs
  const int hx = Lx >> 1, run = kEoYG * hx, nRuns = 2 * tn;
  typedef F vec2 __attribute__((ext_vector_type(2)));
  const int ln = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int chunks = (run + 63) / 64;
  const int items = nRuns * chunks;
  for (int it0 = wv; it0 < items; it0 += 4 * kEoLd) {
    vec2 u[kEoLd];
    int dst[kEoLd];
#pragma unroll
    for (int q = 0; q < kEoLd; q++) {
      const int it = it0 + 4 * q < items ? it0 + 4 * q : items - 1;  // surplus slots repeat the last item (same value, same place)
      const int tp = it / chunks, ch = it - tp * chunks;
      int r = ln + 64 * ch;
      r = r < run ? r : 0;
      const int pty = tp & 1, t = t0 + (tp >> 1);
      const int yy = r / hx, xh = r - yy * hx, y = y0 + yy;
      const int x = 2 * xh + ((pty - (y + z + t)) & 1);
      const int64_t x_cb = ((((int64_t)t * Lz + z) * Ly + y0) * Lx >> 1) + r;
      u[q] = __builtin_nontemporal_load(as_global(reinterpret_cast<const vec2 *>(src + (int64_t)pty * a.volumeCB + x_cb)));
      dst[q] = (yy * tn + (t - t0)) * ld + x;
    }
#pragma unroll
    for (int q = 0; q < kEoLd; q++) tile[dst[q]] = Cplx<F>{sign * u[q].x, sign * u[q].y};
  }
  __syncthreads();
  eo_dft_x_sums(a, tile, y0, z, idataTo, t0, tn);
}

// LDS bytes of eo_dft_x_kernel: the tile, and the partial sums -- in the tile's place when one pass covers all rows (<= 64) and
// all distinct p_x (<= kEoCh), behind it otherwise (a later pass needs the rows again)
size_t eo_dft_x_lds_bytes(int precision, const int localL[4], int nPx, int *redOffsetElems) {
  // (the matrix-pipe variant keeps [4][64][kEoMfmaLdC] doubles = 2304 complex there; it runs single pass only)
  const size_t tile = (size_t)kEoYG * localL[3] * (localL[0] + 1), red = std::max<size_t>((size_t)4 * 64 * kEoCh, (size_t)4 * 64 * kEoMfmaLdC / 2);
  const bool single = kEoYG * localL[3] <= 64 && nPx <= kEoCh;
  if (redOffsetElems) *redOffsetElems = single ? 0 : (int)tile;
  return (single ? std::max(tile, red) : tile + red) * 2 * (size_t)precision;
}

// Time slices per tile of the fused reorder + x step: all of them if the tile then fits 64 KiB of LDS (the pipelined and the
// matrix-pipe forms need that), else the largest chunk whose tile does and whose rows go through in one pass; 0 if even one
// time slice does not fit (Lx of several thousand).
int eo_dft_x_time_chunk(int precision, const int localL[4], int nPx) {
  if (eo_dft_x_lds_bytes(precision, localL, nPx, nullptr) <= 64 * 1024) return localL[3];
  int L[4] = {localL[0], localL[1], localL[2], localL[3]};
  int tcMax = 0;
  for (int tc = std::min(localL[3], 64 / kEoYG); tc >= 1; tc--) {
    L[3] = tc;
    if (eo_dft_x_lds_bytes(precision, L, nPx, nullptr) <= 64 * 1024) {
      tcMax = tc;
      break;
    }
  }
  if (tcMax == 0) return 0;
  // the largest chunk that divides Lt and that the pipelined forms can stage (4 waves x kEoLd loads cover its 2 tc runs) ...
  const int run = kEoYG * localL[0] / 2, rpw = run <= 64 ? 64 / run : 0;
  if (rpw)
    for (int tc = tcMax; tc >= 1; tc--)
      if (localL[3] % tc == 0 && (2 * tc + rpw - 1) / rpw <= 4 * kEoLd) return tc;
  // ... else even chunks for the general kernel
  const int nChunks = (localL[3] + tcMax - 1) / tcMax;
  return (localL[3] + nChunks - 1) / nChunks;
}

// the plan: distinct p_x, distinct (p_x, p_y) pairs, and the tables of the three steps
struct SeparablePlan {
  std::vector<int> px, py, pz;             // distinct values
  std::vector<int> pairIpx, pairIpy;       // pairs sorted by ipx
  std::vector<int> momPair, momIpz;        // per momentum n
  // per step: groups and outputs
  std::vector<int> gSrc[3], gFirst[3], gCount[3], oRow[3], oPos[3];
  int nOut[3];
};

static int index_of(std::vector<int> &v, int x) {
  for (size_t i = 0; i < v.size(); i++)
    if (v[i] == x) return (int)i;
  v.push_back(x);
  return (int)v.size() - 1;
}

static void add_groups(SeparablePlan &P, int st, int src, const std::vector<int> &rows, const std::vector<int> &pos) {
  for (size_t b = 0; b < rows.size(); b += kDftJT) {
    const int cnt = (int)std::min<size_t>(kDftJT, rows.size() - b);
    P.gSrc[st].push_back(src);
    P.gFirst[st].push_back((int)P.oRow[st].size());
    P.gCount[st].push_back(cnt);
    for (int j = 0; j < cnt; j++) {
      P.oRow[st].push_back(rows[b + j]);
      P.oPos[st].push_back(pos[b + j]);
    }
  }
}

static void build_plan(const int *mom, int Nmom, SeparablePlan &P) {
  std::vector<int> momIpx(Nmom), momIpy(Nmom);
  P.momPair.resize(Nmom);
  P.momIpz.resize(Nmom);
  for (int n = 0; n < Nmom; n++) {
    momIpx[n] = index_of(P.px, mom[3 * n + 0]);
    momIpy[n] = index_of(P.py, mom[3 * n + 1]);
    P.momIpz[n] = index_of(P.pz, mom[3 * n + 2]);
  }
  // pairs, grouped by ipx
  for (int ix = 0; ix < (int)P.px.size(); ix++)
    for (int n = 0; n < Nmom; n++)
      if (momIpx[n] == ix) {
        bool seen = false;
        for (size_t q = 0; q < P.pairIpx.size(); q++) seen = seen || (P.pairIpx[q] == ix && P.pairIpy[q] == momIpy[n]);
        if (!seen) {
          P.pairIpx.push_back(ix);
          P.pairIpy.push_back(momIpy[n]);
        }
      }
  for (int n = 0; n < Nmom; n++)
    for (size_t q = 0; q < P.pairIpx.size(); q++)
      if (P.pairIpx[q] == momIpx[n] && P.pairIpy[q] == momIpy[n]) P.momPair[n] = (int)q;
  // step x: outputs = distinct p_x, one source
  {
    std::vector<int> rows, pos;
    for (int ix = 0; ix < (int)P.px.size(); ix++) {
      rows.push_back(ix);
      pos.push_back(ix);
    }
    add_groups(P, 0, 0, rows, pos);
    P.nOut[0] = (int)P.px.size();
  }
  // step y: outputs = pairs; source = the pair's ipx
  for (int ix = 0; ix < (int)P.px.size(); ix++) {
    std::vector<int> rows, pos;
    for (size_t q = 0; q < P.pairIpx.size(); q++)
      if (P.pairIpx[q] == ix) {
        rows.push_back(P.pairIpy[q]);
        pos.push_back((int)q);
      }
    add_groups(P, 1, ix, rows, pos);
  }
  P.nOut[1] = (int)P.pairIpx.size();
  // step z: outputs = momenta (written at their own index n); source = the momentum's pair
  for (size_t q = 0; q < P.pairIpx.size(); q++) {
    std::vector<int> rows, pos;
    for (int n = 0; n < Nmom; n++)
      if (P.momPair[n] == (int)q) {
        rows.push_back(P.momIpz[n]);
        pos.push_back(n);
      }
    add_groups(P, 2, (int)q, rows, pos);
  }
  P.nOut[2] = Nmom;
}

// f_d(q, g) = cos(2 pi phi) + i FTSign sin(2 pi phi), phi = Float(q * g) / Float(totalL_d): the reference's rounding of the
// phase (lib/mugiq_util_kernels.cu:20-31), one direction at a time
template <typename F> static void phase_rows(std::vector<Cplx<F>> &out, const std::vector<int> &q, int L, int g0, int totalL, int FTSign) {
  const double PI = 2.0 * asin(1.0);  // include/util_mugiq.h:7
  for (size_t r = 0; r < q.size(); r++)
    for (int x = 0; x < L; x++) {
      const F phi = static_cast<F>(q[r] * (x + g0)) / static_cast<F>(totalL);
      const double arg = 2.0 * PI * static_cast<double>(phi);
      out.push_back(Cplx<F>{static_cast<F>(cos(arg)), static_cast<F>(FTSign) * static_cast<F>(sin(arg))});
    }
}

static size_t separable_workspace_elems(const SeparablePlan &P, const int localL[4], int M) {
  return ((size_t)localL[2] * localL[1] * P.px.size() + (size_t)localL[2] * P.pairIpx.size()) * (size_t)M;
}

template <typename F>
static int launch_separable(void *C, const void *A, const void *dataPosEO, int nData, const int *mom, int Nmom, int FTSign,
                            const int localL[4], const int totalL[4], const int commCoord[4], int M, void *ws, hipStream_t stream,
                            const int *slotMap_h = nullptr, int nLoopOut = 0) {
  SeparablePlan P;
  build_plan(mom, Nmom, P);
  // one table: [phases x | phases y | phases z | int tables of the three steps]
  std::vector<Cplx<F>> ph;
  size_t phOff[3];
  const std::vector<int> *qs[3] = {&P.px, &P.py, &P.pz};
  for (int d = 0; d < 3; d++) {
    phOff[d] = ph.size();
    phase_rows<F>(ph, *qs[d], localL[d], (commCoord ? commCoord[d] : 0) * localL[d], totalL[d], FTSign);
  }
  // step x of the fused kernel reads the x phases transposed: [Lx][nPxPad], zero-filled (see eo_dft_x_kernel)
  const int nPxPad = ((int)P.px.size() + kEoCh - 1) / kEoCh * kEoCh;
  const size_t phXT = ph.size();
  if (dataPosEO != nullptr) {
    ph.resize(phXT + (size_t)localL[0] * nPxPad, Cplx<F>{F(0), F(0)});
    for (size_t r = 0; r < P.px.size(); r++)
      for (int x = 0; x < localL[0]; x++) ph[phXT + (size_t)x * nPxPad + r] = ph[phOff[0] + r * localL[0] + x];
  }
  std::vector<int> ints;
  size_t iOff[3][5];
  for (int st = 0; st < 3; st++) {
    const std::vector<int> *v[5] = {&P.gSrc[st], &P.gFirst[st], &P.gCount[st], &P.oRow[st], &P.oPos[st]};
    for (int t = 0; t < 5; t++) {
      iOff[st][t] = ints.size();
      ints.insert(ints.end(), v[t]->begin(), v[t]->end());
    }
  }
  const size_t slotOff = ints.size();
  if (slotMap_h) ints.insert(ints.end(), slotMap_h, slotMap_h + nData / 16);
  const size_t phBytes = (ph.size() * sizeof(Cplx<F>) + 255) / 256 * 256;
  std::vector<unsigned char> host(phBytes + ints.size() * sizeof(int));
  memcpy(host.data(), ph.data(), ph.size() * sizeof(Cplx<F>));
  memcpy(host.data() + phBytes, ints.data(), ints.size() * sizeof(int));
  void *dev = nullptr;
  int rc = upload_table(&dev, host.data(), host.size(), stream);
  if (rc) return rc;
  const Cplx<F> *ph_d = static_cast<const Cplx<F> *>(dev);
  const int *int_d = reinterpret_cast<const int *>(static_cast<unsigned char *>(dev) + phBytes);

  Cplx<F> *t1 = static_cast<Cplx<F> *>(ws);
  Cplx<F> *t2 = t1 + (size_t)localL[2] * localL[1] * P.px.size() * (size_t)M;
  const Cplx<F> *ins[3] = {static_cast<const Cplx<F> *>(A), t1, t2};
  Cplx<F> *outs[3] = {t1, t2, static_cast<Cplx<F> *>(C)};
  const int Lsum[3] = {localL[0], localL[1], localL[2]};
  const int innerIn[3] = {1, (int)P.px.size(), (int)P.pairIpx.size()};
  const int outer[3] = {localL[2] * localL[1], localL[2], 1};
  int firstStep = 0;
  if (dataPosEO != nullptr) {  // step x straight from the even-odd buffer (A, the reordered copy, is not needed)
    EoDftArgs<F> e;
    const int tChunk = eo_dft_x_time_chunk((int)sizeof(F), localL, (int)P.px.size());
    MUGIQ_REQUIRE(tChunk >= 1 && localL[2] <= 65535 && nData <= 65535, "performMomentumProjection: lattice too large for the fused reorder + x step");
    int Lc[4] = {localL[0], localL[1], localL[2], tChunk};
    const size_t shmem = eo_dft_x_lds_bytes((int)sizeof(F), Lc, (int)P.px.size(), &e.redOffset);
    e.tChunk = tChunk;
    e.in = static_cast<const Cplx<F> *>(dataPosEO);
    e.out = t1;
    e.ph = ph_d + phXT;
    e.nPxPad = nPxPad;
    long long vol = 1;
    for (int d = 0; d < 4; d++) {
      e.X[d] = localL[d];
      vol *= localL[d];
    }
    e.volumeCB = (int)(vol / 2);
    e.nPx = (int)P.px.size();
    e.M = M;
    e.slotMap = slotMap_h ? int_d + slotOff : nullptr;
    // the pipelined forms walk tiles = (y pair, chunk of time slices): whole time slabs when they fit the LDS, else even chunks
    const int nCh = localL[3] % tChunk == 0 ? localL[3] / tChunk : 0;
    const int run = kEoYG * localL[0] / 2, tiles = (localL[1] / kEoYG) * std::max(nCh, 1);
    const bool pipelined = nCh >= 1 && run <= 64 && (2 * tChunk + 64 / run - 1) / (64 / run) <= 4 * kEoLd;
    // the sums on the matrix pipe where it applies (fp64; one pass: <= 64 rows, <= 8 distinct p_x; Lx = 24, 32, 48 or 64):
    // 1.95 ms against 2.06 ms for the vector form at 48.48.24.24 x 25 slots (profiles/r02_eo_dft_x_kernel_stats_*.csv);
    // MUGIQ_HIP_EO_MFMA = 0 keeps the vector form
    bool mfma = true;
    if (const char *m = getenv("MUGIQ_HIP_EO_MFMA")) mfma = atoi(m) != 0;
    const int mfmaKs = localL[0] / 8, mfmaMb = (kEoYG * tChunk + 15) / 16;
    mfma = mfma && sizeof(F) == 8 && pipelined && kEoYG * tChunk <= 64 && (int)P.px.size() <= 8 && localL[0] % 8 == 0 &&
           (mfmaKs == 3 || mfmaKs == 4 || mfmaKs == 6 || mfmaKs == 8) && mfmaMb >= 2;
    if (pipelined) {
      // about 32 workgroups per CU (8 rounds of 4): enough to balance, few enough to amortise the pipeline fill
      const long long slabs = (long long)localL[2] * nData;
      int perWg = (int)((slabs * tiles + 8191) / 8192);
      perWg = perWg < 1 ? 1 : (perWg > tiles ? tiles : perWg);
      if (const char *t = getenv("MUGIQ_HIP_EO_TILES_PER_WG")) perWg = std::max(1, std::min(tiles, atoi(t)));
      e.tilesPerWg = perWg;
      if constexpr (sizeof(F) == 8) {
        if (mfma) {
          const dim3 grid((tiles + perWg - 1) / perWg, localL[2], nData);
#define MUGIQ_EO_MFMA_CASE(K_, M_) \
  if (mfmaKs == K_ && mfmaMb == M_) hipLaunchKernelGGL((eo_dft_x_mfma_kernel<K_, M_>), grid, dim3(256), shmem, stream, e);
          MUGIQ_EO_MFMA_CASE(3, 2) MUGIQ_EO_MFMA_CASE(3, 3) MUGIQ_EO_MFMA_CASE(3, 4) MUGIQ_EO_MFMA_CASE(4, 2) MUGIQ_EO_MFMA_CASE(4, 3)
          MUGIQ_EO_MFMA_CASE(4, 4) MUGIQ_EO_MFMA_CASE(6, 2) MUGIQ_EO_MFMA_CASE(6, 3) MUGIQ_EO_MFMA_CASE(6, 4) MUGIQ_EO_MFMA_CASE(8, 2)
          MUGIQ_EO_MFMA_CASE(8, 3) MUGIQ_EO_MFMA_CASE(8, 4)
#undef MUGIQ_EO_MFMA_CASE
        }
      }
      if (!mfma)
        hipLaunchKernelGGL((eo_dft_x_pipelined_kernel<F>), dim3((tiles + perWg - 1) / perWg, localL[2], nData), dim3(256), shmem, stream, e);
    } else {
      e.tilesPerWg = 1;
      const int nChunks = (localL[3] + tChunk - 1) / tChunk;
      hipLaunchKernelGGL((eo_dft_x_kernel<F>), dim3((localL[1] / kEoYG) * nChunks, localL[2], nData), dim3(256), shmem, stream, e);
    }
    MUGIQ_CHECK_HIP(hipGetLastError());
    firstStep = 1;
  }
  for (int st = firstStep; st < 3; st++) {
    DftStepArgs<F> a;
    a.in = ins[st];
    a.out = outs[st];
    a.ph = ph_d + phOff[st];
    a.groupSrc = int_d + iOff[st][0];
    a.groupFirst = int_d + iOff[st][1];
    a.groupCount = int_d + iOff[st][2];
    a.outRow = int_d + iOff[st][3];
    a.outPos = int_d + iOff[st][4];
    a.M = M;
    a.Lsum = Lsum[st];
    a.innerIn = innerIn[st];
    a.nOut = P.nOut[st];
    a.rowSlot = nullptr;
    a.rowsPerSlot = 16 * localL[3];
    a.Mout = M;
    if (st == 2 && slotMap_h) {  // the subset's rows land in the full layout
      a.rowSlot = int_d + slotOff;
      a.Mout = a.rowsPerSlot * nLoopOut;
    }
    const dim3 grid((M + 255) / 256, (unsigned)P.gSrc[st].size(), outer[st]);
    const size_t shmem = sizeof(Cplx<F>) * kDftJT * (size_t)Lsum[st];
    MUGIQ_REQUIRE(shmem <= 64 * 1024 && grid.y <= 65535 && grid.z <= 65535, "performMomentumProjection: lattice / momentum list too large for the separable plan");
    hipLaunchKernelGGL((partial_dft_kernel<F>), grid, dim3(256), shmem, stream, a);
    MUGIQ_CHECK_HIP(hipGetLastError());
  }
  return MUGIQ_HIP_SUCCESS;
}

}  // namespace mugiq

using namespace mugiq;

extern "C" {

size_t mugiq_hip_momentum_projection_workspace(int locT, int nData, long long locV3, int Nmom, int precision) {
  if (locT < 1 || nData < 1 || locV3 < 1 || Nmom < 1 || (precision != 4 && precision != 8)) return 0;
  MomProjGeom g;
  choose_split(locT * nData, Nmom, locV3, g);
  if (g.nSplit == 1) return 0;
  return (size_t)g.nSplit * (size_t)g.M * (size_t)g.N * 2 * (size_t)precision;
}

int mugiq_hip_momentum_projection(void *dataMom_d, const void *dataPosMP_d, const void *phaseMatrix_d, int locT, int nData,
                                  long long locV3, int Nmom, int precision, void *workspace_d, size_t workspace_bytes,
                                  void *stream) {
  if (int dbg_ = mugiq::debug_poison_lds_if_asked(static_cast<hipStream_t>(stream))) return dbg_;
  const char *who = "performMomentumProjection";
  MUGIQ_REQUIRE(dataMom_d && dataPosMP_d && phaseMatrix_d, "%s: NULL argument", who);
  MUGIQ_REQUIRE(precision == 4 || precision == 8, "%s: Precision not supported!", who);  // lib/loop_mugiq.cpp:379
  MUGIQ_REQUIRE(locT >= 1 && nData >= 1 && locV3 >= 1 && Nmom >= 1, "%s: invalid sizes locT=%d nData=%d locV3=%lld Nmom=%d", who,
                locT, nData, locV3, Nmom);
  MUGIQ_REQUIRE((long long)locT * nData < (1LL << 31), "%s: locT*nData overflows int", who);
  MomProjGeom g;
  choose_split(locT * nData, Nmom, locV3, g);
  const size_t need = mugiq_hip_momentum_projection_workspace(locT, nData, locV3, Nmom, precision);
  void *ws = workspace_d;
  if (need > 0 && (ws == nullptr || workspace_bytes < need)) {
    int st = stream_workspace(&ws, need, static_cast<hipStream_t>(stream));
    if (st) return st;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (precision == 8) return launch_momproj<double>(dataMom_d, dataPosMP_d, phaseMatrix_d, g, ws, s);
  return launch_momproj<float>(dataMom_d, dataPosMP_d, phaseMatrix_d, g, ws, s);
}

size_t mugiq_hip_momentum_projection_separable_workspace(const int *momMatrix_h, int Nmom, const int localL[4], int locT, int nData,
                                                          int precision) {
  if (!momMatrix_h || !localL || Nmom < 1 || locT < 1 || nData < 1 || (precision != 4 && precision != 8)) return 0;
  SeparablePlan P;
  build_plan(momMatrix_h, Nmom, P);
  return separable_workspace_elems(P, localL, locT * nData) * 2 * (size_t)precision;
}

int mugiq_hip_momentum_projection_separable(void *dataMom_d, const void *dataPosMP_d, const int *momMatrix_h, int Nmom, int FTSign,
                                            const int localL[4], const int totalL[4], const int commCoord[4], int locT, int nData,
                                            int precision, void *workspace_d, size_t workspace_bytes, void *stream) {
  if (int dbg_ = mugiq::debug_poison_lds_if_asked(static_cast<hipStream_t>(stream))) return dbg_;
  const char *who = "performMomentumProjection";
  MUGIQ_REQUIRE(dataMom_d && dataPosMP_d && momMatrix_h && localL && totalL, "%s: NULL argument", who);
  MUGIQ_REQUIRE(precision == 4 || precision == 8, "%s: Precision not supported!", who);  // lib/loop_mugiq.cpp:379
  MUGIQ_REQUIRE(locT >= 1 && nData >= 1 && Nmom >= 1, "%s: invalid sizes locT=%d nData=%d Nmom=%d", who, locT, nData, Nmom);
  MUGIQ_REQUIRE(FTSign == 1 || FTSign == -1, "%s: FTSign = %d must be +1 or -1", who, FTSign);
  MUGIQ_REQUIRE((long long)locT * nData < (1LL << 31), "%s: locT*nData overflows int", who);
  for (int d = 0; d < 3; d++) MUGIQ_REQUIRE(localL[d] > 0 && totalL[d] > 0, "%s: localL / totalL [%d]", who, d);
  const size_t need = mugiq_hip_momentum_projection_separable_workspace(momMatrix_h, Nmom, localL, locT, nData, precision);
  void *ws = workspace_d;
  if (ws == nullptr || workspace_bytes < need) {
    int st = stream_workspace(&ws, need, static_cast<hipStream_t>(stream));
    if (st) return st;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (precision == 8)
    return launch_separable<double>(dataMom_d, dataPosMP_d, nullptr, nData, momMatrix_h, Nmom, FTSign, localL, totalL, commCoord, locT * nData, ws, s);
  return launch_separable<float>(dataMom_d, dataPosMP_d, nullptr, nData, momMatrix_h, Nmom, FTSign, localL, totalL, commCoord, locT * nData, ws, s);
}

int mugiq_hip_convert_and_project(void *dataMom_d, const void *dataPos_d, int nData, int nLoop, const int *momMatrix_h, int Nmom, int FTSign,
                                  const int localL[4], const int totalL[4], const int commCoord[4], int precision, void *workspace_d,
                                  size_t workspace_bytes, void *stream) {
  const char *who = "performMomentumProjection";
  MUGIQ_REQUIRE(dataMom_d && dataPos_d && momMatrix_h && localL && totalL, "%s: NULL argument", who);
  MUGIQ_REQUIRE(nData == nLoop * 16 && nLoop >= 1, "%s: This function assumes that nData = nLoop * NGamma", who);  // lib/contract_wrappers.cu:138
  MUGIQ_REQUIRE(precision == 4 || precision == 8, "%s: Precision not supported!", who);
  MUGIQ_REQUIRE(Nmom >= 1 && (FTSign == 1 || FTSign == -1), "%s: Nmom = %d, FTSign = %d", who, Nmom, FTSign);
  long long vol = 1;
  for (int d = 0; d < 4; d++) {
    MUGIQ_REQUIRE(localL[d] > 0 && (localL[d] & 1) == 0 && totalL[d] > 0, "%s: localL[%d] = %d must be positive and even", who, d, localL[d]);
    vol *= localL[d];
  }
  MUGIQ_REQUIRE(vol < (1LL << 31) && (long long)localL[3] * nData < (1LL << 31), "%s: local volume overflows int", who);
  const int locT = localL[3];
  const size_t need = mugiq_hip_momentum_projection_separable_workspace(momMatrix_h, Nmom, localL, locT, nData, precision);
  void *ws = workspace_d;
  if (ws == nullptr || workspace_bytes < need) {
    int st = stream_workspace(&ws, need, static_cast<hipStream_t>(stream));
    if (st) return st;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (precision == 8)
    return launch_separable<double>(dataMom_d, nullptr, dataPos_d, nData, momMatrix_h, Nmom, FTSign, localL, totalL, commCoord, locT * nData, ws, s);
  return launch_separable<float>(dataMom_d, nullptr, dataPos_d, nData, momMatrix_h, Nmom, FTSign, localL, totalL, commCoord, locT * nData, ws, s);
}

int mugiq_hip_convert_and_project_slots(void *dataMom_d, const void *dataPos_d, int nLoop, const int *slots_h, int nSlots,
                                        const int *momMatrix_h, int Nmom, int FTSign, const int localL[4], const int totalL[4],
                                        const int commCoord[4], int precision, void *workspace_d, size_t workspace_bytes, void *stream) {
  if (int dbg_ = mugiq::debug_poison_lds_if_asked(static_cast<hipStream_t>(stream))) return dbg_;
  const char *who = "performMomentumProjection";
  MUGIQ_REQUIRE(dataMom_d && dataPos_d && slots_h && momMatrix_h && localL && totalL, "%s: NULL argument", who);
  MUGIQ_REQUIRE(nLoop >= 1 && nSlots >= 1 && nSlots <= nLoop, "%s: nSlots = %d of nLoop = %d", who, nSlots, nLoop);
  MUGIQ_REQUIRE(precision == 4 || precision == 8, "%s: Precision not supported!", who);
  MUGIQ_REQUIRE(Nmom >= 1 && (FTSign == 1 || FTSign == -1), "%s: Nmom = %d, FTSign = %d", who, Nmom, FTSign);
  long long vol = 1;
  for (int d = 0; d < 4; d++) {
    MUGIQ_REQUIRE(localL[d] > 0 && (localL[d] & 1) == 0 && totalL[d] > 0, "%s: localL[%d] = %d must be positive and even", who, d, localL[d]);
    vol *= localL[d];
  }
  const int nData = 16 * nSlots, locT = localL[3];
  for (int c = 0; c < nSlots; c++) MUGIQ_REQUIRE(slots_h[c] >= 0 && slots_h[c] < nLoop, "%s: slot %d is not in [0, %d)", who, slots_h[c], nLoop);
  MUGIQ_REQUIRE(vol < (1LL << 31) && (long long)locT * 16 * nLoop < (1LL << 31), "%s: local volume overflows int", who);
  const size_t need = mugiq_hip_momentum_projection_separable_workspace(momMatrix_h, Nmom, localL, locT, nData, precision);
  void *ws = workspace_d;
  if (ws == nullptr || workspace_bytes < need) {
    int st = stream_workspace(&ws, need, static_cast<hipStream_t>(stream));
    if (st) return st;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (precision == 8)
    return launch_separable<double>(dataMom_d, nullptr, dataPos_d, nData, momMatrix_h, Nmom, FTSign, localL, totalL, commCoord, locT * nData, ws, s, slots_h, nLoop);
  return launch_separable<float>(dataMom_d, nullptr, dataPos_d, nData, momMatrix_h, Nmom, FTSign, localL, totalL, commCoord, locT * nData, ws, s, slots_h, nLoop);
}

}  // extern "C"
