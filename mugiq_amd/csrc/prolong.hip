// f2: coarse -> fine prolongation of the eigenvectors (MG coarse path, BASELINE.json configs[4]), batched over the
// eigenvectors, optionally fused with the ultra-local 16-gamma contraction.
//
// Reference: Loop_Mugiq::prolongateEvec (lib/loop_mugiq.cpp:277-319) calls QUDA's Transfer::P once per
// eigenvector and per displacement entry (:482), allocating and freeing its temporaries each time:
//     out(x; s, c) = sum_{j < n_vec} V(x; s, c, j) * in(X(x); s / spin_bs, j)
// (QUDA include/kernels/prolongator.cuh; V = block-orthonormal null vectors, X(x) = aggregate of x).
//
// MI355X design: per fine site the work is a (12 x n_vec) . (n_vec x N_ev) complex product -- GEMM-shaped, but on
// gfx950 the fp64 (and fp32) MFMA rate equals the vector rate, so the tile runs on the VALU and what matters is
// feeding it: a workgroup stages the V rows of 16 consecutive checkerboard sites in LDS ONCE (n_vec*12*16 complex,
// 74 KB for fp64 n_vec = 24) and sweeps all N_ev eigenvectors over them, 16 sites x 16 eigenvector groups per
// workgroup, two eigenvectors per lane per pass so each LDS operand feeds two FMAs chains.  The coarse vectors
// (3 MB each) are served by L2.  V is read from HBM exactly once per call (the reference re-reads it N_ev times).
// With CONTRACT the prolonged vectors are consumed on the spot by the Hermitian 16-gamma accumulation and never
// written: the ultra-local loop of the MG path costs one pass over V instead of N_ev fine-vector writes + reads.
#include "internal.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace mugiq {

constexpr int kPrTile = 16;    // sites per workgroup
constexpr int kPrGroups = 16;  // eigenvector groups per workgroup

template <typename F, typename A> struct ProlongArgs {
  const Cplx<F> *V;             // [parity][(3s+c)*NV + j][x_cb]
  int64_t Vpo;
  int Vstride;
  int NV;
  int X[4], Xc[4], bs[4];
  int spinBs;
  int volumeCB;
  const void *const *coarse;    // device table: nVec coarse bodies [parity][chi*NV + j][x_cb_c]
  int64_t Cpo;
  int Cstride;
  const void *const *fine;      // device table: nVec fine bodies (WRITE)
  int Fstride;
  int64_t Fpo;
  int nVec;
  const A *inv_sigma;           // (CONTRACT)
  Cplx<A> *loop;                // (CONTRACT) [16][V]
};

template <typename F, typename A, int ORDER, bool WRITE, bool CONTRACT>
__global__ __launch_bounds__(kPrTile *kPrGroups) void prolong_kernel(ProlongArgs<F, A> a) {
  extern __shared__ __align__(16) unsigned char smem[];
  Cplx<F> *Vt = reinterpret_cast<Cplx<F> *>(smem);  // [12*NV][16 sites]
  const int tilesPerParity = (a.volumeCB + kPrTile - 1) / kPrTile;
  const int pty = blockIdx.x / tilesPerParity;
  const int x0 = (blockIdx.x - pty * tilesPerParity) * kPrTile;
  const int t = threadIdx.x, site = t & (kPrTile - 1), g = t / kPrTile;
  const int x_cb = x0 + site;
  const bool valid = x_cb < a.volumeCB;

  const int nPl = 12 * a.NV;
  for (int k = g; k < nPl; k += kPrGroups)
    Vt[k * kPrTile + site] = valid ? a.V[pty * a.Vpo + (int64_t)k * a.Vstride + x_cb] : Cplx<F>{F(0), F(0)};
  __syncthreads();

  // aggregate of this fine site: coarse coordinates = fine / block, even-odd on the coarse lattice
  int c[4] = {0, 0, 0, 0};
  if (valid) get_coords(c, x_cb, a.X, pty);
  int cc[4];
#pragma unroll
  for (int d = 0; d < 4; d++) cc[d] = c[d] / a.bs[d];
  const int cpar = (cc[0] + cc[1] + cc[2] + cc[3]) & 1;
  const int64_t coff = (int64_t)cpar * a.Cpo + (lex_index(cc, a.Xc) >> 1);

  A diag[4] = {A(0), A(0), A(0), A(0)};
  Cplx<A> up[6];
#pragma unroll
  for (int i = 0; i < 6; i++) up[i] = Cplx<A>{A(0), A(0)};

  for (int nb = 0; nb < a.nVec; nb += 2 * kPrGroups) {
    const int n0 = nb + 2 * g, n1 = n0 + 1;
    if (n0 >= a.nVec) continue;
    const bool has1 = n1 < a.nVec;
    const Cplx<F> *c0 = static_cast<const Cplx<F> *>(a.coarse[n0]) + coff;
    const Cplx<F> *c1 = static_cast<const Cplx<F> *>(a.coarse[has1 ? n1 : n0]) + coff;
    Cplx<A> o0[12], o1[12];
#pragma unroll
    for (int k = 0; k < 12; k++) o0[k] = o1[k] = Cplx<A>{A(0), A(0)};
    // The null-vector index j runs through a three-stage register pipeline: the 12 LDS reads of V(x; :, j) and the four
    // L2 reads of the coarse components are issued two steps before the 96 FMAs that consume them (the plain loop
    // waited for both latencies in every iteration: the VALU was ~45 % busy; 32^4, n_vec 24, 200 eigenvectors:
    // prolongate-to-fine 36 -> 29 ms).  All loads are unconditional (index clamped) so that hipcc's wait-count pass
    // keeps counted waits.
    typedef F vec2 __attribute__((ext_vector_type(2)));
    Cplx<F> sv0[12], sv1[12], sv2[12];
    vec2 sp0[4], sp1[4], sp2[4];  // [chi] of eigenvector n0, [2 + chi] of eigenvector n1
#define MUGIQ_PR_LOAD(sv, sp, jexpr)                                                                                   \
  {                                                                                                                    \
    const int j_ = (jexpr) < a.NV ? (jexpr) : a.NV - 1;                                                                \
    _Pragma("unroll") for (int chi = 0; chi < 2; chi++) {                                                              \
      sp[chi] = *as_global(reinterpret_cast<const vec2 *>(c0 + (int64_t)(chi * a.NV + j_) * a.Cstride));               \
      sp[2 + chi] = *as_global(reinterpret_cast<const vec2 *>(c1 + (int64_t)(chi * a.NV + j_) * a.Cstride));           \
    }                                                                                                                  \
    _Pragma("unroll") for (int sc = 0; sc < 12; sc++) sv[sc] = Vt[(sc * a.NV + j_) * kPrTile + site];                  \
  }
#define MUGIQ_PR_COMPUTE(sv, sp)                                                                                       \
  {                                                                                                                    \
    const Cplx<A> p0[2] = {Cplx<A>{(A)sp[0].x, (A)sp[0].y}, Cplx<A>{(A)sp[1].x, (A)sp[1].y}};                          \
    const Cplx<A> p1[2] = {Cplx<A>{(A)sp[2].x, (A)sp[2].y}, Cplx<A>{(A)sp[3].x, (A)sp[3].y}};                          \
    _Pragma("unroll") for (int sc = 0; sc < 12; sc++) {                                                                \
      const Cplx<A> v{(A)sv[sc].re, (A)sv[sc].im};                                                                     \
      const int chi = (sc / 3) / 2; /* spin_map(s) = s / spin_bs with spin_bs = 2 (tests/loop.cpp:569) */              \
      cmadd(o0[sc], v, p0[chi]);                                                                                       \
      cmadd(o1[sc], v, p1[chi]);                                                                                       \
    }                                                                                                                  \
  }
    if constexpr (CONTRACT) {
      // with the 16 Hermitian accumulators on top, three stages do not fit 256 VGPRs any more (fp64: one wave per SIMD
      // plus accumulation-register traffic): measured 21.6 ms against 17.5 ms for the plain loop in fp64, 12.3 ms
      // against 10.9 ms in fp32 -- the fused variant keeps the plain loop
      for (int j = 0; j < a.NV; j++) {
        MUGIQ_PR_LOAD(sv0, sp0, j)
        MUGIQ_PR_COMPUTE(sv0, sp0)
      }
    } else {
      MUGIQ_PR_LOAD(sv0, sp0, 0)
      MUGIQ_PR_LOAD(sv1, sp1, 1)
      for (int j = 0; j < a.NV; j += 3) {
        MUGIQ_PR_LOAD(sv2, sp2, j + 2)
        MUGIQ_PR_COMPUTE(sv0, sp0)
        MUGIQ_PR_LOAD(sv0, sp0, j + 3)
        if (j + 1 < a.NV) MUGIQ_PR_COMPUTE(sv1, sp1)
        MUGIQ_PR_LOAD(sv1, sp1, j + 4)
        if (j + 2 < a.NV) MUGIQ_PR_COMPUTE(sv2, sp2)
      }
    }
#undef MUGIQ_PR_LOAD
#undef MUGIQ_PR_COMPUTE
    if constexpr (WRITE) {
      if (valid) {
        Cplx<F> w[12];
#pragma unroll
        for (int k = 0; k < 12; k++) w[k] = Cplx<F>{(F)o0[k].re, (F)o0[k].im};
        SpinorView<F, ORDER>{const_cast<F *>(static_cast<const F *>(a.fine[n0])), a.Fstride, a.Fpo}.store(w, pty, x_cb);
        if (has1) {
#pragma unroll
          for (int k = 0; k < 12; k++) w[k] = Cplx<F>{(F)o1[k].re, (F)o1[k].im};
          SpinorView<F, ORDER>{const_cast<F *>(static_cast<const F *>(a.fine[n1])), a.Fstride, a.Fpo}.store(w, pty, x_cb);
        }
      }
    }
    if constexpr (CONTRACT) {
      accumulate_herm(diag, up, o0, a.inv_sigma[n0]);
      if (has1) accumulate_herm(diag, up, o1, a.inv_sigma[n1]);
    }
  }

  if constexpr (CONTRACT) {
    // combine the 16 eigenvector groups in a fixed order (deterministic), then the 16 gamma traces
    __syncthreads();
    A *red = reinterpret_cast<A *>(smem);  // [16 values][16 groups][16 sites]
#pragma unroll
    for (int i = 0; i < 4; i++) red[(i * kPrGroups + g) * kPrTile + site] = diag[i];
#pragma unroll
    for (int i = 0; i < 6; i++) {
      red[((4 + 2 * i) * kPrGroups + g) * kPrTile + site] = up[i].re;
      red[((5 + 2 * i) * kPrGroups + g) * kPrTile + site] = up[i].im;
    }
    __syncthreads();
    if (g == 0 && valid) {
      A sum[16];
#pragma unroll
      for (int i = 0; i < 16; i++) {
        A s = A(0);
        for (int gg = 0; gg < kPrGroups; gg++) s += red[(i * kPrGroups + gg) * kPrTile + site];
        sum[i] = s;
      }
      Cplx<A> acc[16];
      int p = 0;
#pragma unroll
      for (int be = 0; be < 4; be++) {
        acc[be * 4 + be] = Cplx<A>{sum[be], A(0)};
#pragma unroll
        for (int al = be + 1; al < 4; al++) {
          acc[be * 4 + al] = Cplx<A>{sum[4 + 2 * p], sum[5 + 2 * p]};
          acc[al * 4 + be] = Cplx<A>{sum[4 + 2 * p], -sum[5 + 2 * p]};
          p++;
        }
      }
      trace_and_store(a.loop, acc, 2 * a.volumeCB, x_cb + pty * a.volumeCB);
    }
  }
}

static int validate_transfer(const MugiqHipTransfer *T, const MugiqHipCoarseField *c0, const char *who) {
  MUGIQ_REQUIRE(T && T->V, "%s: transfer / null vectors are NULL", who);
  MUGIQ_REQUIRE(T->precision == 4 || T->precision == 8, "%s: transfer precision %d", who, T->precision);
  MUGIQ_REQUIRE(T->nVec >= 1 && T->nVec <= 64, "%s: n_vec = %d must be in [1, 64]", who, T->nVec);
  MUGIQ_REQUIRE(T->spinBlockSize == 2, "%s: spin_block_size = %d (the reference's drivers use 2, tests/loop.cpp:569)", who, T->spinBlockSize);
  long long vol = 1, volc = 1;
  for (int d = 0; d < 4; d++) {
    MUGIQ_REQUIRE(T->X[d] > 0 && (T->X[d] & 1) == 0, "%s: fine X[%d] = %d must be positive and even", who, d, T->X[d]);
    MUGIQ_REQUIRE(T->geoBlockSize[d] >= 1 && T->X[d] % T->geoBlockSize[d] == 0, "%s: geo_block_size[%d] = %d does not divide X = %d", who, d, T->geoBlockSize[d], T->X[d]);
    const int xc = T->X[d] / T->geoBlockSize[d];
    MUGIQ_REQUIRE((xc & 1) == 0, "%s: coarse extent %d in dim %d must be even (even-odd coarse field)", who, xc, d);
    MUGIQ_REQUIRE(c0->X[d] == xc, "%s: coarse field X[%d] = %d, expected %d", who, d, c0->X[d], xc);
    vol *= T->X[d];
    volc *= xc;
  }
  MUGIQ_REQUIRE(T->stride >= vol / 2 && T->parity_offset >= (int64_t)12 * T->nVec * T->stride, "%s: V stride / parity_offset too small", who);
  MUGIQ_REQUIRE(c0->data && c0->precision == T->precision && c0->nSpin == 2 && c0->nColor == T->nVec, "%s: coarse field must have precision %d, nSpin 2, nColor %d", who, T->precision, T->nVec);
  MUGIQ_REQUIRE(c0->volumeCB == volc / 2 && c0->stride >= c0->volumeCB && c0->parity_offset >= (int64_t)2 * T->nVec * c0->stride, "%s: coarse field geometry mismatch", who);
  return MUGIQ_HIP_SUCCESS;
}

template <typename F, typename A, int ORDER, bool WRITE, bool CONTRACT>
static int launch_prolong(const MugiqHipTransfer *T, const MugiqHipCoarseField *coarse, const MugiqHipSpinorField *fine,
                          const double *sigma, void *loop_d, int nVec, hipStream_t stream) {
  const size_t pb = sizeof(void *) * (size_t)nVec;
  std::vector<unsigned char> host(2 * pb + sizeof(A) * (size_t)nVec);
  const void **hc = reinterpret_cast<const void **>(host.data());
  const void **hf = reinterpret_cast<const void **>(host.data() + pb);
  A *hs = reinterpret_cast<A *>(host.data() + 2 * pb);
  for (int n = 0; n < nVec; n++) {
    hc[n] = coarse[n].data;
    hf[n] = WRITE ? fine[n].data : nullptr;
    hs[n] = CONTRACT ? static_cast<A>(1.0 / static_cast<F>(sigma[n])) : A(0);
  }
  void *dev = nullptr;
  int st = upload_table(&dev, host.data(), host.size(), stream);
  if (st) return st;
  ProlongArgs<F, A> a;
  a.V = static_cast<const Cplx<F> *>(T->V);
  a.Vpo = T->parity_offset;
  a.Vstride = T->stride;
  a.NV = T->nVec;
  long long vol = 1;
  for (int d = 0; d < 4; d++) {
    a.X[d] = T->X[d];
    a.bs[d] = T->geoBlockSize[d];
    a.Xc[d] = T->X[d] / T->geoBlockSize[d];
    vol *= T->X[d];
  }
  a.spinBs = T->spinBlockSize;
  a.volumeCB = (int)(vol / 2);
  a.coarse = reinterpret_cast<const void *const *>(dev);
  a.Cpo = coarse[0].parity_offset;
  a.Cstride = coarse[0].stride;
  a.fine = reinterpret_cast<const void *const *>(static_cast<unsigned char *>(dev) + pb);
  a.Fstride = WRITE ? fine[0].stride : 0;
  a.Fpo = WRITE ? fine[0].parity_offset : 0;
  a.nVec = nVec;
  a.inv_sigma = reinterpret_cast<const A *>(static_cast<unsigned char *>(dev) + 2 * pb);
  a.loop = static_cast<Cplx<A> *>(loop_d);
  size_t shmem = sizeof(Cplx<F>) * 12 * (size_t)T->nVec * kPrTile;
  const size_t redBytes = sizeof(A) * 16 * kPrGroups * kPrTile;
  if (CONTRACT && shmem < redBytes) shmem = redBytes;
  auto kern = prolong_kernel<F, A, ORDER, WRITE, CONTRACT>;
  if (shmem > 64 * 1024)
    MUGIQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  const int tiles = 2 * ((a.volumeCB + kPrTile - 1) / kPrTile);
  hipLaunchKernelGGL(kern, dim3(tiles), dim3(kPrTile * kPrGroups), shmem, stream, a);
  MUGIQ_CHECK_HIP(hipGetLastError());
  return MUGIQ_HIP_SUCCESS;
}


// ---- coarse -> coarse levels of the hierarchy --------------------------------------------------------------------------
// Loop_Mugiq::prolongateEvec walks the MG hierarchy from the coarsest level up (lib/loop_mugiq.cpp:306-311:
// transfer[lev-1]->P(tmpCSF[lev-1], tmpCSF[lev]) for lev = nCoarseLevels .. 2) before the finest transfer (:314).  Both sides
// of such a level are coarse fields with nSpin = 2 (the chirality survives every level: spin_block_size = 1 there), so
//     out(x; s, c) = sum_{j < n_vec} V(x; s, c, j) * in(X(x); s, j),    c < nColor(out) = n_vec of the next finer level,
// V = FieldOrderCB<Float, 2, nColor(out), n_vec, FLOAT2>: plane (s * nColor(out) + c) * n_vec + j.
// The intermediate lattices are 256 x smaller than the fine one, so this is a plain kernel: lane = site (coalesced along
// x_cb), blockIdx.y = output component, a lane carries kPcEvecs eigenvectors so that every V element it loads is used
// kPcEvecs times; sums run over j in ascending order in the field precision.
constexpr int kPcEvecs = 8;
template <typename F> struct ProlongCoarseArgs {
  const Cplx<F> *V;
  int64_t Vpo;
  int Vstride, NV, NCf;
  int X[4], Xc[4], bs[4];
  int volumeCB;
  const void *const *in;  // device table: nVec coarser bodies, then nVec finer bodies
  int64_t Ipo, Opo;
  int Istride, Ostride, nVec;
};

template <typename F> __global__ __launch_bounds__(128) void prolong_coarse_kernel(ProlongCoarseArgs<F> a) {
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid >= 2 * a.volumeCB) return;
  const int pty = tid / a.volumeCB, x_cb = tid - pty * a.volumeCB;
  const int k = blockIdx.y, s = k / a.NCf;  // output component (s, c); the coarser field's spin is the same chirality
  const int n0 = blockIdx.z * kPcEvecs;
  int c[4], cc[4];
  get_coords(c, x_cb, a.X, pty);
#pragma unroll
  for (int d = 0; d < 4; d++) cc[d] = c[d] / a.bs[d];
  const int cpar = (cc[0] + cc[1] + cc[2] + cc[3]) & 1;
  const int64_t coff = (int64_t)cpar * a.Ipo + (lex_index(cc, a.Xc) >> 1);
  const Cplx<F> *inp[kPcEvecs];
#pragma unroll
  for (int i = 0; i < kPcEvecs; i++) {
    const int n = n0 + i < a.nVec ? n0 + i : a.nVec - 1;
    inp[i] = static_cast<const Cplx<F> *>(as_constant(a.in)[n]) + coff + (int64_t)(s * a.NV) * a.Istride;
  }
  Cplx<F> acc[kPcEvecs];
#pragma unroll
  for (int i = 0; i < kPcEvecs; i++) acc[i] = Cplx<F>{F(0), F(0)};
  const Cplx<F> *vp = a.V + (int64_t)pty * a.Vpo + (int64_t)k * a.NV * a.Vstride + x_cb;
  for (int j = 0; j < a.NV; j++) {
    const Cplx<F> v = vp[(int64_t)j * a.Vstride];
#pragma unroll
    for (int i = 0; i < kPcEvecs; i++) {
      typedef F vec2 __attribute__((ext_vector_type(2)));
      const vec2 u = *as_global(reinterpret_cast<const vec2 *>(inp[i] + (int64_t)j * a.Istride));
      cmadd(acc[i], v, Cplx<F>{u.x, u.y});
    }
  }
#pragma unroll
  for (int i = 0; i < kPcEvecs; i++)
    if (n0 + i < a.nVec) {
      Cplx<F> *o = static_cast<Cplx<F> *>(const_cast<void *>(as_constant(a.in)[a.nVec + n0 + i]));
      o[(int64_t)pty * a.Opo + (int64_t)k * a.Ostride + x_cb] = acc[i];
    }
}

template <typename F>
static int launch_prolong_coarse(const MugiqHipCoarseField *out, const MugiqHipCoarseField *in, int nVec, const MugiqHipTransfer *T,
                                 hipStream_t stream) {
  std::vector<const void *> host(2 * (size_t)nVec);
  for (int n = 0; n < nVec; n++) {
    host[n] = in[n].data;
    host[nVec + n] = out[n].data;
  }
  void *dev = nullptr;
  int st = upload_table(&dev, host.data(), host.size() * sizeof(void *), stream);
  if (st) return st;
  ProlongCoarseArgs<F> a;
  a.V = static_cast<const Cplx<F> *>(T->V);
  a.Vpo = T->parity_offset;
  a.Vstride = T->stride;
  a.NV = T->nVec;
  a.NCf = out[0].nColor;
  for (int d = 0; d < 4; d++) {
    a.X[d] = T->X[d];
    a.bs[d] = T->geoBlockSize[d];
    a.Xc[d] = T->X[d] / T->geoBlockSize[d];
  }
  a.volumeCB = out[0].volumeCB;
  a.in = reinterpret_cast<const void *const *>(dev);
  a.Ipo = in[0].parity_offset;
  a.Istride = in[0].stride;
  a.Opo = out[0].parity_offset;
  a.Ostride = out[0].stride;
  a.nVec = nVec;
  const dim3 grid((2 * a.volumeCB + 127) / 128, 2 * a.NCf, (nVec + kPcEvecs - 1) / kPcEvecs);
  hipLaunchKernelGGL((prolong_coarse_kernel<F>), grid, dim3(128), 0, stream, a);
  MUGIQ_CHECK_HIP(hipGetLastError());
  return MUGIQ_HIP_SUCCESS;
}

// ---- MG ultra-local loop through the coarse-grid outer product -----------------------------------------------------
// With psi_n(x) = V(x) phi_n(X(x)) the weighted sum over the eigenvectors can be taken on the COARSE grid first:
//     sum_n s_n psi_n(x) psi_n(x)^dagger = V(x) C(X) V(x)^dagger,     C(X) = sum_n s_n phi_n(X) phi_n(X)^dagger,
// one (2 n_vec)^2 Hermitian matrix per aggregate, shared by all its fine sites.  The N_ev-fold prolongation
// (N_ev * 336 complex multiply-adds per fine site) becomes one congruence per fine site (about 15 000, independent of
// N_ev): 4.5 x fewer flops at N_ev = 200, n_vec = 24, and the eigenvectors are read once from the 3 MB coarse fields.
// The colour-traced spin matrix the gamma traces need is resG[be][al] = sum_c M[(al,c),(be,c)] with M = V C V^dagger
// (lib/mugiq_contract_kernels.cu:98-105 with vL = vR = psi_n, summed over n with weights 1/sigma_n).
template <typename F, typename A> struct CoarseOuterArgs {
  const void *const *coarse;  // device table: nVec coarse bodies [parity][chi*NV + j][x_cb_c]
  const A *inv_sigma;
  int nVec, NV, volumeCBc, Cstride;
  int64_t Cpo;
  Cplx<A> *C;                 // [2*volumeCBc][NC][NC]
};

constexpr int kCoEvecs = 8;   // eigenvectors staged per barrier
constexpr int kCoMaxNC = 64;  // 2 * n_vec handled by the coarse path

__device__ inline int xcd_contiguous_block(int blk, int nblk);

// One workgroup per coarse site; the 256 threads tile C as 16 x 16 blocks of B x B entries (B = ceil(NC / 16)), so a
// thread reads 2B components from LDS per B^2 complex multiply-adds.
// A workgroup reads ONE 16-byte element of every (eigenvector, component) plane; the eight coarse sites of a 128-byte line are
// eight workgroups.  Dealt round-robin over the 8 XCDs they pulled the line into eight L2s (PMC, round 2: 5.03 GB fetched for
// 0.63 GB of coarse eigenvectors); XCD k now walks the k-th contiguous eighth of the sites, so the sharers of a line run side
// by side on one XCD.
template <typename F, typename A, int B> __global__ __launch_bounds__(256) void coarse_outer_kernel(CoarseOuterArgs<F, A> a) {
  __shared__ Cplx<A> ph[kCoEvecs][kCoMaxNC], phs[kCoEvecs][kCoMaxNC];
  const int NC = 2 * a.NV;
  const int site = xcd_contiguous_block(blockIdx.x, gridDim.x);
  const int pty = site / a.volumeCBc, x_cb = site - pty * a.volumeCBc;
  const int r0 = (threadIdx.x >> 4) * B, c0 = (threadIdx.x & 15) * B;
  Cplx<A> acc[B][B];
#pragma unroll
  for (int i = 0; i < B; i++)
#pragma unroll
    for (int j = 0; j < B; j++) acc[i][j] = Cplx<A>{A(0), A(0)};
  for (int n0 = 0; n0 < a.nVec; n0 += kCoEvecs) {
    __syncthreads();
    for (int idx = threadIdx.x; idx < kCoEvecs * kCoMaxNC; idx += 256) {
      const int nn = idx / kCoMaxNC, comp = idx - nn * kCoMaxNC, n = n0 + nn;
      Cplx<A> v{A(0), A(0)}, vs{A(0), A(0)};  // components beyond NC and eigenvectors beyond nVec contribute zero
      if (n < a.nVec && comp < NC) {
        typedef F vec2 __attribute__((ext_vector_type(2)));
        const Cplx<F> *c = static_cast<const Cplx<F> *>(as_constant(a.coarse)[n]) + (int64_t)pty * a.Cpo + (int64_t)comp * a.Cstride + x_cb;
        const vec2 u = *as_global(reinterpret_cast<const vec2 *>(c));
        const A s = as_constant(a.inv_sigma)[n];
        v = Cplx<A>{(A)u.x, (A)u.y};
        vs = Cplx<A>{s * v.re, s * v.im};
      }
      ph[nn][comp] = v;
      phs[nn][comp] = vs;
    }
    __syncthreads();
#pragma unroll
    for (int nn = 0; nn < kCoEvecs; nn++) {
      Cplx<A> x[B], y[B];
#pragma unroll
      for (int i = 0; i < B; i++) {
        x[i] = phs[nn][(r0 + i) & (kCoMaxNC - 1)];
        y[i] = ph[nn][(c0 + i) & (kCoMaxNC - 1)];
      }
#pragma unroll
      for (int i = 0; i < B; i++)
#pragma unroll
        for (int j = 0; j < B; j++) {  // acc += (s phi[r]) * conj(phi[c])
          acc[i][j].re = fma(x[i].re, y[j].re, acc[i][j].re);
          acc[i][j].re = fma(x[i].im, y[j].im, acc[i][j].re);
          acc[i][j].im = fma(x[i].im, y[j].re, acc[i][j].im);
          acc[i][j].im = fma(-x[i].re, y[j].im, acc[i][j].im);
        }
    }
  }
  Cplx<A> *out = a.C + (int64_t)site * NC * NC;
#pragma unroll
  for (int i = 0; i < B; i++)
#pragma unroll
    for (int j = 0; j < B; j++)
      if (r0 + i < NC && c0 + j < NC) out[(r0 + i) * NC + c0 + j] = acc[i][j];
}

// Aggregate of a workgroup.  The V rows of an aggregate are 32-byte pieces (two checkerboard entries of an x row of four
// sites) of 128-byte lines that four x-adjacent aggregates share.  Workgroups are dealt round-robin over the 8 XCDs, so with
// aggregate = blockIdx the four sharers sit on four XCDs and each pulls the line into its own L2.  Here XCD k walks the k-th
// contiguous eighth of the lexicographic aggregate order, so the sharers run side by side on one XCD (measured: -8 % on both
// congruence kernels at 32^4, n_vec 24).
__device__ inline int xcd_contiguous_block(int blk, int nblk) {
  if (nblk & 7) return blk;
  return (blk & 7) * (nblk >> 3) + (blk >> 3);
}

template <typename F, typename A> struct FineCongruenceArgs {
  const Cplx<F> *V;       // [parity][(3s+c)*NV + j][x_cb]
  int64_t Vpo;
  int Vstride, NV;
  int X[4], Xc[4], bs[4];
  int volumeCB, volumeCBc, aggVol;
  const Cplx<A> *C;       // [2*volumeCBc][NC][NC]
  Cplx<A> *loop;          // [16][V]
};

// One workgroup per aggregate; SPR fine sites per round, 4 * NH lanes per site: lane (h, q) <-> column chunk h of JC null
// vectors and (chi, chi') = (q >> 1, q & 1); it accumulates its share of the 2x2 block  sum_c M[(be,c),(al,c)],
// be in {2chi, 2chi+1}, al in {2chi', 2chi'+1}.  C(X) sits in LDS for the whole workgroup; per round and colour the rows
// V(x; :, c, :) of the SPR sites are staged in LDS.  T = u C is built for the lane's JC columns (2 rows x JC accumulators
// in registers) and folded into the block at once; the NH partial blocks of a site are summed with wavefront shuffles.
// (Spreading the chunks over lanes instead of looping over them doubles the waves per CU at the same LDS footprint.)
template <typename F, typename A, int JC, int SPR, int NH>
__global__ __launch_bounds__(4 * NH * SPR) void fine_congruence_kernel(FineCongruenceArgs<F, A> a) {
  extern __shared__ __align__(16) unsigned char smem[];
  const int NV = a.NV, NC = 2 * NV;
  Cplx<A> *Cs = reinterpret_cast<Cplx<A> *>(smem);                                    // [NC][NC]
  Cplx<A> *red = Cs + NC * NC;                                                         // [SPR][16]
  Cplx<F> *Vs = reinterpret_cast<Cplx<F> *>(red + SPR * 16);                           // [4 spins][NV][SPR]
  constexpr int LPS = 4 * NH, NT = LPS * SPR;  // lanes per site, threads per workgroup
  const int t = threadIdx.x, sidx = t / LPS, lane = t % LPS, hh = lane >> 2, q = lane & 3, chi = q >> 1, chip = q & 1;

  int cc[4], r = xcd_contiguous_block(blockIdx.x, gridDim.x);
#pragma unroll
  for (int d = 0; d < 4; d++) {
    cc[d] = r % a.Xc[d];
    r /= a.Xc[d];
  }
  const int cpar = (cc[0] + cc[1] + cc[2] + cc[3]) & 1;
  const Cplx<A> *Cg = a.C + ((int64_t)cpar * a.volumeCBc + (lex_index(cc, a.Xc) >> 1)) * (int64_t)(NC * NC);
  for (int i = t; i < NC * NC; i += NT) Cs[i] = Cg[i];

  // fine site (parity, x_cb) of aggregate member k (lexicographic inside the block); k >= aggVol shadows the last one
  auto member = [&](int k, int &pty, int &x_cb) {
    if (k >= a.aggVol) k = a.aggVol - 1;
    int x[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
      x[d] = cc[d] * a.bs[d] + k % a.bs[d];
      k /= a.bs[d];
    }
    pty = (x[0] + x[1] + x[2] + x[3]) & 1;
    x_cb = lex_index(x, a.X) >> 1;
  };

  const int rounds = (a.aggVol + SPR - 1) / SPR;
  const int stSite = t % SPR, stRow0 = t / SPR;  // staging: this lane always fetches site stSite, rows stRow0 + LPS*i
  for (int rd = 0; rd < rounds; rd++) {
    int myP, myX, stP, stX;
    member(rd * SPR + sidx, myP, myX);
    member(rd * SPR + stSite, stP, stX);
    Cplx<A> blk[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++) blk[i][0] = blk[i][1] = Cplx<A>{A(0), A(0)};
    for (int c = 0; c < 3; c++) {
      __syncthreads();  // previous consumers of Vs (and the first use of Cs) are done / ready
      for (int row = stRow0; row < 4 * NV; row += LPS) {  // row = spin * NV + j
        const int sp = row / NV, j = row - sp * NV;
        typedef F vec2 __attribute__((ext_vector_type(2)));
        const Cplx<F> *src = a.V + (int64_t)stP * a.Vpo + (int64_t)((3 * sp + c) * NV + j) * a.Vstride + stX;
        const vec2 u = *as_global(reinterpret_cast<const vec2 *>(src));
        Vs[row * SPR + stSite] = Cplx<F>{u.x, u.y};
      }
      __syncthreads();
      const Cplx<F> *u0p = Vs + (2 * chi) * NV * SPR + sidx, *u1p = u0p + NV * SPR;      // rows be = 2chi, 2chi+1
      const Cplx<F> *w0p = Vs + (2 * chip) * NV * SPR + sidx, *w1p = w0p + NV * SPR;    // rows al = 2chi', 2chi'+1
      {
        const int h = hh * JC;
        Cplx<A> T0[JC], T1[JC];
#pragma unroll
        for (int jj = 0; jj < JC; jj++) T0[jj] = T1[jj] = Cplx<A>{A(0), A(0)};
        const Cplx<A> *crow = Cs + (chi * NV) * NC + chip * NV + h;
        for (int j = 0; j < NV; j++) {
          const Cplx<F> a0 = u0p[j * SPR], a1 = u1p[j * SPR];
          const Cplx<A> u0{(A)a0.re, (A)a0.im}, u1{(A)a1.re, (A)a1.im};
#pragma unroll
          for (int jj = 0; jj < JC; jj++) {
            const Cplx<A> cv = crow[j * NC + jj];
            cmadd(T0[jj], u0, cv);
            cmadd(T1[jj], u1, cv);
          }
        }
#pragma unroll
        for (int jj = 0; jj < JC; jj++) {  // blk[b2][a2] += T_b2[j'] * conj(V[al, c, j'])
          const Cplx<F> b0 = w0p[(h + jj) * SPR], b1 = w1p[(h + jj) * SPR];
          const Cplx<A> w0{(A)b0.re, (A)b0.im}, w1{(A)b1.re, (A)b1.im};
          cmadd_conj(blk[0][0], w0, T0[jj]);
          cmadd_conj(blk[0][1], w1, T0[jj]);
          cmadd_conj(blk[1][0], w0, T1[jj]);
          cmadd_conj(blk[1][1], w1, T1[jj]);
        }
      }
    }
    // sum the NH column chunks (lanes q + 4h of the same site sit in one wavefront)
#pragma unroll
    for (int m = 4; m < LPS; m <<= 1)
#pragma unroll
      for (int b2 = 0; b2 < 2; b2++)
#pragma unroll
        for (int a2 = 0; a2 < 2; a2++) {
          blk[b2][a2].re += __shfl_xor(blk[b2][a2].re, m);
          blk[b2][a2].im += __shfl_xor(blk[b2][a2].im, m);
        }
    // resG[al][be] = sum_c M[(be,c),(al,c)]: collect the four 2x2 blocks of a site, then 4 of the 16 gamma traces per lane
    if (hh == 0) {
#pragma unroll
      for (int b2 = 0; b2 < 2; b2++)
#pragma unroll
        for (int a2 = 0; a2 < 2; a2++) red[sidx * 16 + (2 * chip + a2) * 4 + (2 * chi + b2)] = blk[b2][a2];
    }
    __syncthreads();
    if (hh == 0 && rd * SPR + sidx < a.aggVol) {
      Cplx<A> full[16];
#pragma unroll
      for (int i = 0; i < 16; i++) full[i] = red[sidx * 16 + i];
      const int site = myX + myP * a.volumeCB;
      if (q == 0) trace_and_store_range<A, 0, 4>(a.loop, full, 2 * a.volumeCB, site);
      else if (q == 1) trace_and_store_range<A, 4, 8>(a.loop, full, 2 * a.volumeCB, site);
      else if (q == 2) trace_and_store_range<A, 8, 12>(a.loop, full, 2 * a.volumeCB, site);
      else trace_and_store_range<A, 12, 16>(a.loop, full, 2 * a.volumeCB, site);
    }
  }
}

// ---- the same congruence on the matrix pipe (fp64 accumulation; n_vec = 8, 16, 24, 32; aggregates of 16 k sites) ----------------
// Per colour c and left spin be the first half of the congruence is a real GEMM shared by all sites of the aggregate:
//   T(x; be, c; j'') = sum_j V(x; be, c, j) C[(chi(be), j), j''],   j'' = (chi', j') over the 2 n_vec columns of C(X),
// i.e. T' = A' C' with the complex structure unfolded (k' <-> (j, re|im of V), n' = 2 j'' + re|im of T; C'[(j,re)][2j''] = Re C,
// C'[(j,im)][2j''] = -Im C, C'[(j,re)][2j''+1] = Im C, C'[(j,im)][2j''+1] = Re C).  v_mfma_f64_16x16x4_f64 computes it
// TRANSPOSED, D = C'^T A'^T: the A operand is a fragment of C'^T (lane: n' = 16 cb + (lane & 15), k = lane >> 4), the B operand
// a fragment of A'^T (lane: k = lane >> 4, site = lane & 15), and lane l receives D[n' = 16 cb + (l >> 4) + 4 r][site = l & 15]
// in register r -- the SITE stays on the lane, so the second half of the congruence,
//   blk[be][al] += sum_{j'} T(be; chi(al), j') conj V(x; al, c, j'),
// is plain per-lane arithmetic on the MFMA result (no transposition of T).  The order of the summation index inside a k-step
// is free as long as both operands agree: k-steps 2 m and 2 m + 1 take Re and Im of V(.., j = 4 m + (lane >> 4)), so ONE
// ds_read_b128 of the staged V tile feeds two MFMAs of a lane (with (j, re|im) in natural order every MFMA needed its own
// ds_read_b64, and the LDS pipe, not the matrix pipe, set the pace).  Wave w <-> (chi, column block cb of 8 complex columns):
// its n_vec / 2 fragments of C'^T stay in registers for the whole aggregate -- C(X) never enters LDS.  (Tried: four waves, one
// per SIMD, each owning a whole 2 x 2 block with n_vec / 8 independent accumulator chains, one barrier per round, a third of the
// B-fragment reads -- 5.1 ms against 3.6 ms: with one wave per SIMD nothing hides the LDS latency of stage two and the staging.)  Per round 16 sites x 12
// (spin, colour) rows of V are staged.  fp64 MFMA shares the fp64 vector pipe (profiles/r02_mfma_f64_probe.json): the gain is
// operand delivery -- one LDS read feeds 2048 multiply-adds instead of 48.
template <typename F> struct CongruenceMfmaArgs {
  const Cplx<F> *V;       // [parity][(3s+c)*NV + j][x_cb]
  int64_t Vpo;
  int Vstride;
  int X[4], Xc[4], bs[4];
  int volumeCB, volumeCBc, aggVol;
  const Cplx<double> *C;  // [2*volumeCBc][NC][NC]
  Cplx<double> *loop;     // [16][V]
};

constexpr int kCmS = 16;  // sites per round = columns of one MFMA

// GLDS (fp64 storage, two tiles fit the LDS): the next round's tile goes global -> LDS directly (global_load_lds_dwordx4, no
// registers, no ds_write pass) into the other buffer while this round is consumed; otherwise (fp32 storage: the tile is
// widened to double on the way; n_vec = 32: one buffer only) the next round waits in registers.
template <typename F, int NV, bool GLDS> __global__ __launch_bounds__(64 * (NV / 2)) void fine_congruence_mfma_kernel(CongruenceMfmaArgs<F> a) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  typedef F vec2 __attribute__((ext_vector_type(2)));
  constexpr int NC = 2 * NV, KS = NV / 2, NCB = NV / 4, NW = 2 * NCB, NT = 64 * NW;  // k-steps, column blocks, waves, threads
  constexpr int ROWS = 12 * NV;                                                        // (spin, colour, j) rows of the V tile
  constexpr int LDV = kCmS;                                                            // row of the V tile (complex)
  constexpr int NLD = (ROWS * kCmS) / NT;                                              // staging loads per lane and round (= 6)
  extern __shared__ __align__(16) unsigned char smem[];
  Cplx<double> *Vs0 = reinterpret_cast<Cplx<double> *>(smem);  // [(s*3 + c)*NV + j][LDV], one or two buffers
  double *red = reinterpret_cast<double *>(Vs0 + (GLDS ? 2 : 1) * ROWS * LDV);  // [NW][kCmS][8]: partial blk of every wave
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int chi = wave / NCB, cb = wave - chi * NCB;          // row block of C / left spins 2 chi, 2 chi + 1; column block
  const int chip = cb / (NCB / 2), jb = (cb - chip * (NCB / 2)) * 8;  // right chirality of these columns; first j' of the block
  const int kq = lane >> 4, site = lane & 15;

  int cc[4], rr = xcd_contiguous_block(blockIdx.x, gridDim.x);
#pragma unroll
  for (int d = 0; d < 4; d++) {
    cc[d] = rr % a.Xc[d];
    rr /= a.Xc[d];
  }
  const int cpar = (cc[0] + cc[1] + cc[2] + cc[3]) & 1;
  const Cplx<double> *Cg = a.C + ((int64_t)cpar * a.volumeCBc + (lex_index(cc, a.Xc) >> 1)) * (int64_t)(NC * NC);
  // resident fragments of C'^T: lane (n' = 16 cb + site, k = kq), k-steps 2 m | 2 m + 1 <-> (j = 4 m + kq, re | im):
  // C[(chi NV + j), j''] with j'' = 8 cb + (site >> 1), output component site & 1
  double cfrag[KS];
#pragma unroll
  for (int m = 0; m < KS / 2; m++) {
    const Cplx<double> cv = Cg[(chi * NV + 4 * m + kq) * NC + 8 * cb + (site >> 1)];
    cfrag[2 * m] = (site & 1) == 0 ? cv.re : cv.im;       // V_re contributes  Re C to T_re,  Im C to T_im
    cfrag[2 * m + 1] = (site & 1) == 0 ? -cv.im : cv.re;  // V_im contributes -Im C to T_re,  Re C to T_im
  }
  // fine site (parity, x_cb) of aggregate member k (lexicographic inside the block)
  auto member = [&](int k, int &pty, int &x_cb) {
    int x[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
      x[d] = cc[d] * a.bs[d] + k % a.bs[d];
      k /= a.bs[d];
    }
    pty = (x[0] + x[1] + x[2] + x[3]) & 1;
    x_cb = lex_index(x, a.X) >> 1;
  };
  // staging: element e = row * 16 + site of the round's tile, e = t + NT * q (six per lane for every n_vec); consecutive lanes
  // take the 16 sites of a row, a wave instruction four consecutive rows = 1 KiB of the LDS image
  vec2 u[GLDS ? 1 : NLD];
  const int stSite = t & 15;
  auto fetch = [&](int rd, Cplx<double> *buf) {
    int pty, x_cb;
    member(rd * kCmS + stSite, pty, x_cb);
    const Cplx<F> *base = a.V + (int64_t)pty * a.Vpo + x_cb;
#pragma unroll
    for (int q = 0; q < NLD; q++) {
      const int row = (t + NT * q) >> 4;
      if constexpr (GLDS) {
#if defined(__HIP_DEVICE_COMPILE__)  // (a device-only builtin: the host pass of hipcc must not see it)
        typedef __attribute__((address_space(3))) void lds_void;
        __builtin_amdgcn_global_load_lds(as_global(reinterpret_cast<const vec2 *>(base + (int64_t)row * a.Vstride)),
                                         (lds_void *)(buf + (wave * 64 + NT * q)), 16, 0, 0);
#endif
      } else {
        u[q] = *as_global(reinterpret_cast<const vec2 *>(base + (int64_t)row * a.Vstride));
      }
    }
  };
  auto commit = [&](Cplx<double> *buf) {
    if constexpr (!GLDS) {
#pragma unroll
      for (int q = 0; q < NLD; q++) buf[t + NT * q] = Cplx<double>{(double)u[q].x, (double)u[q].y};
    }
  };
  // operand addresses (complex elements of the tile): B fragments of (be, c), k-step pair m: V(site; be, c, j = 4 m + kq);
  // W(al, c, j' = jb + (kq >> 1) + 2 r)
  const int bBase = kq * LDV + site, wBase = (jb + (kq >> 1)) * LDV + site;
  const int rT = kq & 1;  // this lane holds Re (0) or Im (1) of T;  blk = T conj(W): Re += T_re W_re + T_im W_im, Im += T_im W_re - T_re W_im

  const int rounds = a.aggVol / kCmS;
  fetch(0, Vs0);
  for (int rd = 0; rd < rounds; rd++) {
    Cplx<double> *Vs = Vs0 + (GLDS ? (rd & 1) * ROWS * LDV : 0);
    if constexpr (GLDS) {
      // my share of this round's tile has landed; after the barrier everybody's has, and nobody reads the other buffer or
      // the partial sums of the previous round any more
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (rd + 1 < rounds) fetch(rd + 1, Vs0 + ((rd + 1) & 1) * ROWS * LDV);
    } else {
      __syncthreads();  // the previous round's consumers are done with Vs / red
      commit(Vs);
      if (rd + 1 < rounds) fetch(rd + 1, Vs);
      __syncthreads();
    }
    double blk[2][2][2];  // [be - 2 chi][al - 2 chip][re | im]
#pragma unroll
    for (int i = 0; i < 8; i++) (&blk[0][0][0])[i] = 0.0;
    // Six units (colour c, left spin be): the B fragments of unit u + 1 are requested BEFORE the MFMAs of unit u, so the LDS
    // reads run under the matrix pipe (with load-then-multiply per unit the twelve waves of the workgroup -- in step after
    // every barrier -- all read LDS, then all multiply: the two times added up).  The W values of stage two do not depend on
    // be: read once per colour.
    Cplx<double> bfA[KS / 2], bfB[KS / 2], w[2][4];
#define MUGIQ_CM_LOAD_B(bf_, u_)                                                                          \
  {                                                                                                       \
    const int rowB_ = ((2 * chi + ((u_) & 1)) * 3 + ((u_) >> 1)) * NV;                                    \
    _Pragma("unroll") for (int m = 0; m < KS / 2; m++) bf_[m] = Vs[(rowB_ + 4 * m) * LDV + bBase];        \
  }
#define MUGIQ_CM_UNIT(bf_, bfNext_, u_)                                                                   \
  {                                                                                                       \
    if ((u_) + 1 < 6) MUGIQ_CM_LOAD_B(bfNext_, (u_) + 1)                                                  \
    if (((u_) & 1) == 0) {                                                                                \
      _Pragma("unroll") for (int ai = 0; ai < 2; ai++)                                                    \
        _Pragma("unroll") for (int r = 0; r < 4; r++)                                                     \
          w[ai][r] = Vs[(((2 * chip + ai) * 3 + ((u_) >> 1)) * NV + 2 * r) * LDV + wBase];                \
    }                                                                                                     \
    __builtin_amdgcn_sched_barrier(0); /* the reads above stay above the MFMAs (hipcc sinks them next to their use) */ \
    d4 acc = {0, 0, 0, 0};                                                                                \
    _Pragma("unroll") for (int m = 0; m < KS / 2; m++) {                                                  \
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cfrag[2 * m], bf_[m].re, acc, 0, 0, 0);                  \
      acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cfrag[2 * m + 1], bf_[m].im, acc, 0, 0, 0);              \
    }                                                                                                     \
    /* acc[r] = Re|Im (kq & 1) of T(site; be, c; chi', j' = jb + (kq >> 1) + 2 r) */                      \
    _Pragma("unroll") for (int ai = 0; ai < 2; ai++)                                                      \
      _Pragma("unroll") for (int r = 0; r < 4; r++) {                                                     \
        const double wa = rT == 0 ? w[ai][r].re : w[ai][r].im;   /* multiplies into Re blk */             \
        const double wb = rT == 0 ? -w[ai][r].im : w[ai][r].re;  /* multiplies into Im blk */             \
        blk[(u_) & 1][ai][0] = fma(acc[r], wa, blk[(u_) & 1][ai][0]);                                     \
        blk[(u_) & 1][ai][1] = fma(acc[r], wb, blk[(u_) & 1][ai][1]);                                     \
      }                                                                                                   \
  }
    MUGIQ_CM_LOAD_B(bfA, 0)
    MUGIQ_CM_UNIT(bfA, bfB, 0)
    MUGIQ_CM_UNIT(bfB, bfA, 1)
    MUGIQ_CM_UNIT(bfA, bfB, 2)
    MUGIQ_CM_UNIT(bfB, bfA, 3)
    MUGIQ_CM_UNIT(bfA, bfB, 4)
    MUGIQ_CM_UNIT(bfB, bfA, 5)
#undef MUGIQ_CM_UNIT
#undef MUGIQ_CM_LOAD_B
    // sum the four kq groups of a site (lanes l, l ^ 16, l ^ 32, l ^ 48), then hand the wave's partial block to LDS
#pragma unroll
    for (int i = 0; i < 8; i++) {
      double v = (&blk[0][0][0])[i];
      v += __shfl_xor(v, 16);
      v += __shfl_xor(v, 32);
      (&blk[0][0][0])[i] = v;
    }
    if (kq == 0) {
#pragma unroll
      for (int i = 0; i < 8; i++) red[(wave * kCmS + site) * 8 + i] = (&blk[0][0][0])[i];
    }
    // LDS-only barrier: __syncthreads() would also wait for the global -> LDS transfers of the next round (vmcnt(0))
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // full[al * 4 + be] = sum over the NCB / 2 waves of (chi(be), chi'(al)), fixed order; 4 of the 16 gamma traces per lane
    if (t < 4 * kCmS) {
      const int s = t >> 2, q = t & 3;
      Cplx<double> full[16];
#pragma unroll
      for (int al = 0; al < 4; al++)
#pragma unroll
        for (int be = 0; be < 4; be++) {
          const int w0 = (be >> 1) * NCB + (al >> 1) * (NCB / 2);
          double re = 0.0, im = 0.0;
#pragma unroll
          for (int wv = 0; wv < NCB / 2; wv++) {
            const double *p = red + ((w0 + wv) * kCmS + s) * 8 + ((be & 1) * 2 + (al & 1)) * 2;
            re += p[0];
            im += p[1];
          }
          full[al * 4 + be] = Cplx<double>{re, im};
        }
      Cplx<double> tr[4];
      if (q == 0) traces_range<double, 0>(tr, full);
      else if (q == 1) traces_range<double, 4>(tr, full);
      else if (q == 2) traces_range<double, 8>(tr, full);
      else traces_range<double, 12>(tr, full);
      // loopData += trace, as fire-and-forget fp64 atomic adds: a read-modify-write would have to wait for its load, and a
      // wait for ANY load result also waits for the global -> LDS transfers of the next round issued before it (one in-order
      // counter) -- a full memory latency per round on wave 0, with the other eleven waves parked at the next barrier
      // (measured: 0.5 ms of 4.1).  Every element receives exactly one addend per call, so the result does not depend on order.
      int pty, x_cb;
      member(rd * kCmS + s, pty, x_cb);
      double *out = reinterpret_cast<double *>(a.loop + (int64_t)(2 * a.volumeCB) * (4 * q) + x_cb + pty * a.volumeCB);
#pragma unroll
      for (int i = 0; i < 4; i++) {
        unsafeAtomicAdd(out + (int64_t)(4 * a.volumeCB) * i, tr[i].re);
        unsafeAtomicAdd(out + (int64_t)(4 * a.volumeCB) * i + 1, tr[i].im);
      }
    }
  }
}

template <typename F, int NV> static int launch_congruence_mfma(const CongruenceMfmaArgs<F> &a, hipStream_t stream) {
  constexpr int NW = NV / 2;
  constexpr size_t tileB = sizeof(Cplx<double>) * (size_t)12 * NV * kCmS, redB = sizeof(double) * (size_t)NW * kCmS * 8;
  constexpr bool GLDS = sizeof(F) == 8 && 2 * tileB + redB <= 160 * 1024;
  const size_t shmem = (GLDS ? 2 : 1) * tileB + redB;
  auto kern = fine_congruence_mfma_kernel<F, NV, GLDS>;
  if (shmem > 64 * 1024)
    MUGIQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(kern, dim3(2 * a.volumeCBc), dim3(64 * NW), shmem, stream, a);
  MUGIQ_CHECK_HIP(hipGetLastError());
  return MUGIQ_HIP_SUCCESS;
}

// LDS bytes of the congruence kernel; 0 if this (n_vec, precision) does not fit
template <typename F, typename A> static size_t congruence_lds(int NV, int SPR) {
  const size_t NC = 2 * (size_t)NV;
  return sizeof(Cplx<A>) * (NC * NC + (size_t)SPR * 16) + sizeof(Cplx<F>) * 4 * NV * (size_t)SPR;
}

template <typename F, typename A, int JC, int SPR, int NH> static int launch_congruence(const FineCongruenceArgs<F, A> &a, size_t shmem, hipStream_t stream) {
  auto kern = fine_congruence_kernel<F, A, JC, SPR, NH>;
  if (shmem > 64 * 1024)
    MUGIQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(kern, dim3(2 * a.volumeCBc), dim3(4 * NH * SPR), shmem, stream, a);
  MUGIQ_CHECK_HIP(hipGetLastError());
  return MUGIQ_HIP_SUCCESS;
}

// returns -1 if the coarse-grid plan does not apply (caller falls back to the per-eigenvector kernel)
template <typename F, typename A>
static int coarse_plan(const MugiqHipTransfer *T, const MugiqHipCoarseField *coarse, const double *sigma, void *loop_d, int nVec,
                       hipStream_t stream) {
  const int NV = T->nVec, NC = 2 * NV;
  // supported shapes: n_vec = 8, 16, 32 (chunks of 8 columns) and 12, 24 (chunks of 12); NH = n_vec / chunk lanes per
  // (site, chi, chi') must be a power of two (shuffle reduction) and 4 * NH * SPR <= 1024 threads
  if (!(NV == 8 || NV == 16 || NV == 32 || NV == 12 || NV == 24)) return -1;
  const int JC = NV % 12 == 0 ? 12 : 8, NH = NV / JC;  // (6-column chunks on 16 lanes per site measured 8 % slower)
  int SPR = 64;
  if (congruence_lds<F, A>(NV, SPR) > 150 * 1024 || 4 * NH * SPR > 1024) SPR = 32;
  if (congruence_lds<F, A>(NV, SPR) > 150 * 1024) return -1;
  long long volc = 1, aggVol = 1;
  for (int d = 0; d < 4; d++) {
    volc *= T->X[d] / T->geoBlockSize[d];
    aggVol *= T->geoBlockSize[d];
  }
  const size_t pb = sizeof(void *) * (size_t)nVec;
  const size_t tabBytes = (pb + sizeof(A) * (size_t)nVec + 255) / 256 * 256;
  const size_t cBytes = sizeof(Cplx<A>) * (size_t)volc * NC * NC;
  // one scratch region holds [pointer table | 1/sigma | C]: reserve it in full first, so that the table upload below
  // (which draws on the same per-stream scratch) cannot move it
  void *base = nullptr;
  int st = stream_scratch(&base, tabBytes + cBytes, stream);
  if (st) return st;
  std::vector<unsigned char> host(tabBytes, 0);
  const void **hc = reinterpret_cast<const void **>(host.data());
  A *hs = reinterpret_cast<A *>(host.data() + pb);
  for (int n = 0; n < nVec; n++) {
    hc[n] = coarse[n].data;
    hs[n] = static_cast<A>(1.0 / static_cast<F>(sigma[n]));
  }
  void *dev = nullptr;
  if ((st = upload_table(&dev, host.data(), tabBytes, stream))) return st;
  MUGIQ_REQUIRE(dev == base, "prolongateContract: scratch moved under the coarse-grid plan");
  CoarseOuterArgs<F, A> o;
  o.coarse = reinterpret_cast<const void *const *>(dev);
  o.inv_sigma = reinterpret_cast<const A *>(static_cast<unsigned char *>(dev) + pb);
  o.nVec = nVec;
  o.NV = NV;
  o.volumeCBc = (int)(volc / 2);
  o.Cstride = coarse[0].stride;
  o.Cpo = coarse[0].parity_offset;
  o.C = reinterpret_cast<Cplx<A> *>(static_cast<unsigned char *>(dev) + tabBytes);
  switch ((NC + 15) / 16) {
  case 1: hipLaunchKernelGGL((coarse_outer_kernel<F, A, 1>), dim3((unsigned)volc), dim3(256), 0, stream, o); break;
  case 2: hipLaunchKernelGGL((coarse_outer_kernel<F, A, 2>), dim3((unsigned)volc), dim3(256), 0, stream, o); break;
  case 3: hipLaunchKernelGGL((coarse_outer_kernel<F, A, 3>), dim3((unsigned)volc), dim3(256), 0, stream, o); break;
  default: hipLaunchKernelGGL((coarse_outer_kernel<F, A, 4>), dim3((unsigned)volc), dim3(256), 0, stream, o); break;
  }
  MUGIQ_CHECK_HIP(hipGetLastError());

  // fp64 accumulation, n_vec = 8, 16, 24, 32, aggregates of a multiple of 16 sites: the congruence on the matrix pipe
  // (MUGIQ_HIP_MG_MFMA=0 keeps the vector kernel)
  if constexpr (sizeof(A) == 8) {
    bool mfma = (NV == 8 || NV == 16 || NV == 24 || NV == 32) && aggVol % kCmS == 0;
    if (const char *e = getenv("MUGIQ_HIP_MG_MFMA")) mfma = mfma && atoi(e) != 0;
    if (mfma) {
      CongruenceMfmaArgs<F> m;
      m.V = static_cast<const Cplx<F> *>(T->V);
      m.Vpo = T->parity_offset;
      m.Vstride = T->stride;
      long long volf = 1;
      for (int d = 0; d < 4; d++) {
        m.X[d] = T->X[d];
        m.bs[d] = T->geoBlockSize[d];
        m.Xc[d] = T->X[d] / T->geoBlockSize[d];
        volf *= T->X[d];
      }
      m.volumeCB = (int)(volf / 2);
      m.volumeCBc = (int)(volc / 2);
      m.aggVol = (int)aggVol;
      m.C = reinterpret_cast<const Cplx<double> *>(o.C);
      m.loop = static_cast<Cplx<double> *>(loop_d);
      switch (NV) {
      case 8: return launch_congruence_mfma<F, 8>(m, stream);
      case 16: return launch_congruence_mfma<F, 16>(m, stream);
      case 24: return launch_congruence_mfma<F, 24>(m, stream);
      default: return launch_congruence_mfma<F, 32>(m, stream);
      }
    }
  }
  FineCongruenceArgs<F, A> a;
  a.V = static_cast<const Cplx<F> *>(T->V);
  a.Vpo = T->parity_offset;
  a.Vstride = T->stride;
  a.NV = NV;
  long long vol = 1;
  for (int d = 0; d < 4; d++) {
    a.X[d] = T->X[d];
    a.bs[d] = T->geoBlockSize[d];
    a.Xc[d] = T->X[d] / T->geoBlockSize[d];
    vol *= T->X[d];
  }
  a.volumeCB = (int)(vol / 2);
  a.volumeCBc = (int)(volc / 2);
  a.aggVol = (int)aggVol;
  a.C = o.C;
  a.loop = static_cast<Cplx<A> *>(loop_d);
  const size_t shmem = congruence_lds<F, A>(NV, SPR);
#define MUGIQ_CONGRUENCE_CASE(J, S, H) \
  if (JC == J && SPR == S && NH == H) return launch_congruence<F, A, J, S, H>(a, shmem, stream);
  MUGIQ_CONGRUENCE_CASE(12, 64, 2) MUGIQ_CONGRUENCE_CASE(12, 32, 2) MUGIQ_CONGRUENCE_CASE(12, 64, 1) MUGIQ_CONGRUENCE_CASE(12, 32, 1)
  MUGIQ_CONGRUENCE_CASE(8, 64, 1) MUGIQ_CONGRUENCE_CASE(8, 32, 1) MUGIQ_CONGRUENCE_CASE(8, 64, 2) MUGIQ_CONGRUENCE_CASE(8, 32, 2)
  MUGIQ_CONGRUENCE_CASE(8, 64, 4) MUGIQ_CONGRUENCE_CASE(8, 32, 4)
#undef MUGIQ_CONGRUENCE_CASE
  return -1;
}


// ---- prolongate-to-fine on the matrix pipe -------------------------------------------------------------------------------
// out_n(x; s, c) = sum_j V(x; s, c, j) phi_n(X(x); chi(s), j) for all eigenvectors n is, per aggregate X and chirality, the
// complex GEMM [256 sites x 6 (s, c)] x [n_vec] . [n_vec x N_ev] with phi shared by every site of the aggregate -- the same
// operand sharing as the congruence above, and the same transposed real form:  D[(n, re | im)][site] = Phi'^T V'^T  on
// v_mfma_f64_16x16x4_f64, the site staying on the lane (column l & 15).  One workgroup per aggregate, eight waves: a wave owns
// one chirality and up to kPmPairs blocks of eight eigenvectors, whose fragments of Phi'^T (n_vec / 2 doubles per block) stay
// in registers for the whole aggregate; a B fragment (Re and Im of one V element = two k-steps) is one ds_read_b128 and
// feeds every block of the wave.  The rows of Phi'^T are ordered so that a lane ends up with (Re, Im) of eigenvectors kq
// and kq + 4 of a block at its site: two 16-byte stores per block, (s, c) and site.  V tile of 16 sites x 12 n_vec rows
// double-buffered in LDS by global_load_lds_dwordx4 exactly as in fine_congruence_mfma_kernel; XCD-contiguous aggregate
// order; fragments of Phi'^T from an eigenvector-major copy of the coarse eigenvectors (coarse_pack_kernel): whole lines.
//
// Measured (32^4, n_vec 24, 200 eigenvectors; profiles/r02_prolong_mfma_probes.txt): 17.8 ms against 28.9 ms for the vector
// kernel above (which gives every lane its own coarse operands -- consecutive x_cb belong to different aggregates -- and
// reaches 17 TFLOP/s).  Without its stores the kernel takes 10.4 ms (48 TFLOP/s); the stores are what is left: the 16 sites
// of an aggregate are eight 32-byte pieces of a fine field, and L2 hands about half of them to memory as 32-byte requests
// (TCC_EA0_WRREQ against _64B).  Tried and dropped, all slower: non-temporal stores (51 ms); the x-neighbouring aggregates as
// waves of one workgroup in step, so that the pieces of a line meet in L2 (28.8 ms); alternating the neighbours inside one
// workgroup tile by tile (51 ms); aggregate pairs per wave with the 64-byte sector completed inside the store instruction
// by a quad permutation (39.9 ms).  Every wait in the loop is vmcnt(0): loads and stores share that counter and are not
// documented to complete in order with respect to each other, so no counted wait is used while stores are in flight.
constexpr int kPmWaves = 8;  // waves per workgroup: even ones take chirality 0, odd ones chirality 1
constexpr int kPmPairs = 4;  // eight-eigenvector blocks a wave keeps resident (measured: 12-16 blocks per pass beat 24 and 8)

struct ProlongMfmaArgs {
  const Cplx<double> *V;  // [parity][(3s+c)*NV + j][x_cb]
  int64_t Vpo;
  int Vstride;
  int X[4], Xc[4], bs[4];
  int volumeCB, volumeCBc, aggVol;
  const void *const *table;    // device table: nVec coarse bodies, then nVec fine bodies
  const Cplx<double> *packed;  // [coarse site, even-odd][chi][j][n < nVec8]: see coarse_pack_kernel
  int64_t Cpo;
  int Cstride;
  int Fstride;
  int64_t Fpo;
  int nVec, nVec8;             // all eigenvectors of the call (nVec8: rounded up to whole blocks)
  int blkBegin, blkCount;      // this launch: blocks [blkBegin, blkBegin + blkCount) of eight eigenvectors
};

// packed[((site * 2 + chi) * NV + j) * nVec8 + n] = phi_n(site; chi, j), site = parity * volumeCBc + x_cb_c, zero for
// nVec <= n < nVec8.  A fragment load of a wave (16 rows = 8 eigenvectors x re | im, 4 values of j) is then four 128-byte
// lines; from the fields themselves it would be 32 scattered 16-byte pieces of 32 different lines.
// A workgroup transposes a tile of 16 coarse sites x 16 eigenvectors of one (parity, chi, j) through LDS: the reads run along
// x_cb inside each eigenvector's field (256-byte runs), the writes along n inside the packed copy (256-byte runs).  (With
// one thread per packed element and n fastest, every 16-byte read came from a different field: 4.0 x the bytes fetched.)
__global__ __launch_bounds__(256) void coarse_pack_kernel(ProlongMfmaArgs a, int NV) {
  __shared__ Cplx<double> tile[16][17];
  const int nT = (a.nVec8 + 15) / 16;  // tiles of 16 eigenvectors (nVec8 is a multiple of 8: the last tile may be half empty)
  const int xt = blockIdx.x, cj = blockIdx.y, nt = blockIdx.z % nT, cpar = blockIdx.z / nT;
  const int hi = threadIdx.x >> 4, lo = threadIdx.x & 15;
  {
    const int n = nt * 16 + hi, xc = xt * 16 + lo;
    Cplx<double> v{0.0, 0.0};
    if (n < a.nVec && xc < a.volumeCBc) v = static_cast<const Cplx<double> *>(a.table[n])[(int64_t)cpar * a.Cpo + (int64_t)cj * a.Cstride + xc];
    tile[hi][lo] = v;
  }
  __syncthreads();
  const int xc = xt * 16 + hi, n = nt * 16 + lo;
  if (xc < a.volumeCBc && n < a.nVec8)
    const_cast<Cplx<double> *>(a.packed)[(((int64_t)cpar * a.volumeCBc + xc) * (2 * NV) + cj) * a.nVec8 + n] = tile[lo][hi];
}

template <int NV> __global__ __launch_bounds__(64 * kPmWaves) void prolong_mfma_kernel(ProlongMfmaArgs a) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  typedef double vec2 __attribute__((ext_vector_type(2)));
  constexpr int KS = NV / 2;  // k-steps of one (s, c): 2 n_vec / 4
  constexpr int ROWS = 12 * NV, LDV = kCmS, NT = 64 * kPmWaves, NLD = (ROWS * kCmS) / NT;
  static_assert((ROWS * kCmS) % NT == 0 && NV % 8 == 0, "tile staging assumes whole wave instructions");
  extern __shared__ __align__(16) unsigned char smem[];
  Cplx<double> *Vs0 = reinterpret_cast<Cplx<double> *>(smem);  // two buffers [(s*3 + c)*NV + j][16 sites]
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int kq = lane >> 4, site = lane & 15;
  const int chi = wave & 1, b0 = wave >> 1;  // my blocks: blkBegin + b0, + 4, + 8, ...
  int npairs = 0;
#pragma unroll
  for (int i = 0; i < kPmPairs; i++)
    if (b0 + 4 * i < a.blkCount) npairs = i + 1;

  int cc[4], rr = xcd_contiguous_block(blockIdx.x, gridDim.x);
#pragma unroll
  for (int d = 0; d < 4; d++) {
    cc[d] = rr % a.Xc[d];
    rr /= a.Xc[d];
  }
  const int cpar = (cc[0] + cc[1] + cc[2] + cc[3]) & 1;

  // resident fragments of Phi'^T: lane (row = l & 15, k = kq); row <-> (eigenvector (row & 3) + 4 (row >> 3) of the block,
  // re | im = (row >> 2) & 1); k-steps 2 m | 2 m + 1 <-> (j = 4 m + kq, V_re | V_im):
  //   re row:  phi_re V_re - phi_im V_im      im row:  phi_im V_re + phi_re V_im
  double afrag[kPmPairs][KS];
  Cplx<double> *fineP[kPmPairs][2];  // bodies of eigenvectors 8 b + kq and 8 b + kq + 4 (what this lane stores), or NULL
  {
    const int rowA = lane & 15, riA = (rowA >> 2) & 1;
    const Cplx<double> *pk = a.packed + (((int64_t)cpar * a.volumeCBc + (lex_index(cc, a.Xc) >> 1)) * 2 + chi) * (int64_t)NV * a.nVec8 +
                             (int64_t)kq * a.nVec8 + (rowA & 3) + 4 * (rowA >> 3);
#pragma unroll
    for (int i = 0; i < kPmPairs; i++) {
      // (blocks past the last one shadow a valid one -- block b0 of the wave, or the first block of the pass for a wave that
      // has none at all: their fragments are loaded but never used, and the loads stay inside the packed array)
      const int b = a.blkBegin + (npairs > 0 ? b0 + (i < npairs ? 4 * i : 0) : 0);
#pragma unroll
      for (int m = 0; m < KS / 2; m++) {
        const vec2 cv = *as_global(reinterpret_cast<const vec2 *>(pk + (int64_t)(4 * m) * a.nVec8 + 8 * b));
        afrag[i][2 * m] = riA == 0 ? cv.x : cv.y;
        afrag[i][2 * m + 1] = riA == 0 ? -cv.y : cv.x;
      }
#pragma unroll
      for (int h = 0; h < 2; h++) {
        const int ns = 8 * b + kq + 4 * h;
        fineP[i][h] = ns < a.nVec ? static_cast<Cplx<double> *>(const_cast<void *>(a.table[a.nVec + ns])) : nullptr;
      }
    }
  }
  // fine site (parity, x_cb) of aggregate member k (lexicographic inside the block)
  auto member = [&](int k, int &pty, int &x_cb) {
    int x[4];
#pragma unroll
    for (int d = 0; d < 4; d++) {
      x[d] = cc[d] * a.bs[d] + k % a.bs[d];
      k /= a.bs[d];
    }
    pty = (x[0] + x[1] + x[2] + x[3]) & 1;
    x_cb = lex_index(x, a.X) >> 1;
  };
  const int stSite = t & 15;
  auto fetch = [&](int rd, Cplx<double> *buf) {
    int pty, x_cb;
    member(rd * kCmS + stSite, pty, x_cb);
    const Cplx<double> *base = a.V + (int64_t)pty * a.Vpo + x_cb;
#pragma unroll
    for (int q = 0; q < NLD; q++) {
      const int row = (t + NT * q) >> 4;
      (void)row, (void)base, (void)buf;
#if defined(__HIP_DEVICE_COMPILE__)  // (a device-only builtin: the host pass of hipcc must not see it)
      typedef __attribute__((address_space(3))) void lds_void;
      __builtin_amdgcn_global_load_lds(as_global(reinterpret_cast<const vec2 *>(base + (int64_t)row * a.Vstride)),
                                       (lds_void *)(buf + (wave * 64 + NT * q)), 16, 0, 0);
#endif
    }
  };

  const int rounds = a.aggVol / kCmS;
  fetch(0, Vs0);
  for (int rd = 0; rd < rounds; rd++) {
    const Cplx<double> *Vs = Vs0 + (rd & 1) * ROWS * LDV;
    // my share of this round's tile has landed and my stores of the previous round have left (one counter for both, with no
    // documented order between the two kinds: zero is the safe wait); after the barrier everybody's has, and nobody reads
    // the other buffer any more
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (rd + 1 < rounds) fetch(rd + 1, Vs0 + ((rd + 1) & 1) * ROWS * LDV);
    int pty, x_cb;
    member(rd * kCmS + site, pty, x_cb);
    const int64_t foff = (int64_t)pty * a.Fpo + x_cb;
#pragma unroll 1
    for (int scl = 0; scl < 6; scl++) {
      const int sc = 6 * chi + scl;  // (s, c) = (2 chi + scl / 3, scl % 3): row block (s*3 + c) * NV of the tile
      Cplx<double> bf[KS / 2];
#pragma unroll
      for (int m = 0; m < KS / 2; m++) bf[m] = Vs[(sc * NV + 4 * m + kq) * LDV + site];
      const int64_t soff = foff + (int64_t)sc * a.Fstride;
      // plain stores: the pieces of a line must be able to meet in L2 (non-temporal stores of 32-byte pieces: 3 x slower)
#define MUGIQ_PM_STORE(i_, acc_)                                                                                        \
  {                                                                                                                     \
    if (fineP[i_][0]) *as_global(reinterpret_cast<vec2 *>(fineP[i_][0] + soff)) = vec2{acc_[0], acc_[1]};               \
    if (fineP[i_][1]) *as_global(reinterpret_cast<vec2 *>(fineP[i_][1] + soff)) = vec2{acc_[2], acc_[3]};               \
  }
      // two blocks at a time: their accumulation chains are independent, so the matrix pipe always has a second instruction
#pragma unroll
      for (int i = 0; i < kPmPairs; i += 2) {
        if (i + 1 < npairs) {
          const int i1 = i + 1 < kPmPairs ? i + 1 : i;
          d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
#pragma unroll
          for (int m = 0; m < KS / 2; m++) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[i][2 * m], bf[m].re, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[i1][2 * m], bf[m].re, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[i][2 * m + 1], bf[m].im, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[i1][2 * m + 1], bf[m].im, acc1, 0, 0, 0);
          }
          MUGIQ_PM_STORE(i, acc0)
          MUGIQ_PM_STORE(i1, acc1)
        } else if (i < npairs) {
          d4 acc0 = {0, 0, 0, 0};
#pragma unroll
          for (int m = 0; m < KS / 2; m++) {
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[i][2 * m], bf[m].re, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(afrag[i][2 * m + 1], bf[m].im, acc0, 0, 0, 0);
          }
          MUGIQ_PM_STORE(i, acc0)
        }
      }
#undef MUGIQ_PM_STORE
    }
  }
}

// The fine fields must be FLOAT2 fp64, n_vec 8 | 16 | 24 (two V tiles fit the LDS), aggregates of a multiple of 16 sites.
// Returns -1 if the shape is not covered (the caller uses the vector kernel); MUGIQ_HIP_PROLONG_MFMA=0 switches it off.
static int prolong_mfma_plan(const MugiqHipTransfer *T, const MugiqHipCoarseField *coarse, const MugiqHipSpinorField *fine, int nVec,
                             hipStream_t stream) {
  const int NV = T->nVec;
  if (!(NV == 8 || NV == 16 || NV == 24)) return -1;
  if (const char *e = getenv("MUGIQ_HIP_PROLONG_MFMA"))
    if (atoi(e) == 0) return -1;
  long long vol = 1, volc = 1, aggVol = 1;
  for (int d = 0; d < 4; d++) {
    vol *= T->X[d];
    volc *= T->X[d] / T->geoBlockSize[d];
    aggVol *= T->geoBlockSize[d];
  }
  if (aggVol % kCmS != 0) return -1;
  ProlongMfmaArgs a;
  a.V = static_cast<const Cplx<double> *>(T->V);
  a.Vpo = T->parity_offset;
  a.Vstride = T->stride;
  for (int d = 0; d < 4; d++) {
    a.X[d] = T->X[d];
    a.bs[d] = T->geoBlockSize[d];
    a.Xc[d] = T->X[d] / T->geoBlockSize[d];
  }
  a.volumeCB = (int)(vol / 2);
  a.volumeCBc = (int)(volc / 2);
  a.aggVol = (int)aggVol;
  a.Cpo = coarse[0].parity_offset;
  a.Cstride = coarse[0].stride;
  a.Fstride = fine[0].stride;
  a.Fpo = fine[0].parity_offset;
  a.nVec = nVec;
  a.nVec8 = (nVec + 7) / 8 * 8;
  std::vector<const void *> host(2 * (size_t)nVec);
  for (int n = 0; n < nVec; n++) {
    host[n] = coarse[n].data;
    host[nVec + n] = fine[n].data;
  }
  void *dev = nullptr;
  int st = upload_table(&dev, host.data(), sizeof(void *) * host.size(), stream);
  if (st) return st;
  a.table = reinterpret_cast<const void *const *>(dev);
  void *ws = nullptr;
  const size_t packBytes = sizeof(Cplx<double>) * (size_t)volc * 2 * NV * (size_t)a.nVec8;
  if ((st = stream_workspace(&ws, packBytes, stream))) return st;
  a.packed = static_cast<const Cplx<double> *>(ws);
  const int nVec16 = (a.nVec8 + 15) / 16;  // (tiles of 16 eigenvectors; the last one may be half empty)
  hipLaunchKernelGGL(coarse_pack_kernel, dim3((unsigned)((a.volumeCBc + 15) / 16), (unsigned)(2 * NV), (unsigned)(2 * nVec16)), dim3(256), 0, stream, a, NV);
  MUGIQ_CHECK_HIP(hipGetLastError());
  // passes: a workgroup keeps 4 * kPmPairs blocks of eight eigenvectors per chirality resident; more eigenvectors than that
  // are split evenly (V is staged once per pass: 12 n_vec 16 B per site against 192 B per site and eigenvector written)
  int cap = 4 * kPmPairs;
  if (const char *e = getenv("MUGIQ_HIP_PROLONG_PASS_BLOCKS")) {  // experiments: blocks of eight eigenvectors per pass (<= 4 kPmPairs)
    const int c = atoi(e);
    if (c >= 1 && c <= 4 * kPmPairs) cap = c;
  }
  const int blocks = a.nVec8 / 8;
  const int passes = (blocks + cap - 1) / cap;
  const int blocksPerPass = (blocks + passes - 1) / passes;
  const size_t shmem = 2 * sizeof(Cplx<double>) * (size_t)12 * NV * kCmS;
  for (int b = 0; b < blocks; b += blocksPerPass) {
    a.blkBegin = b;
    a.blkCount = std::min(blocksPerPass, blocks - b);
#define MUGIQ_PM_LAUNCH(N_)                                                                                             \
  {                                                                                                                     \
    auto kern = prolong_mfma_kernel<N_>;                                                                                \
    if (shmem > 64 * 1024)                                                                                              \
      MUGIQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem)); \
    hipLaunchKernelGGL(kern, dim3((unsigned)volc), dim3(64 * kPmWaves), shmem, stream, a);                              \
  }
    if (NV == 8) MUGIQ_PM_LAUNCH(8) else if (NV == 16) MUGIQ_PM_LAUNCH(16) else MUGIQ_PM_LAUNCH(24)
#undef MUGIQ_PM_LAUNCH
    MUGIQ_CHECK_HIP(hipGetLastError());
  }
  return MUGIQ_HIP_SUCCESS;
}

}  // namespace mugiq

using namespace mugiq;

extern "C" {

int mugiq_hip_prolongate_batched(const MugiqHipSpinorField *fine_h, const MugiqHipCoarseField *coarse_h, int nVec,
                                 const MugiqHipTransfer *transfer, void *stream) {
  if (int dbg_ = mugiq::debug_poison_lds_if_asked(static_cast<hipStream_t>(stream))) return dbg_;
  const char *who = "prolongateEvec";
  MUGIQ_REQUIRE(fine_h && coarse_h && nVec >= 1, "%s: NULL / empty argument", who);
  int st = validate_transfer(transfer, &coarse_h[0], who);
  if (st) return st;
  for (int n = 0; n < nVec; n++) {
    if ((st = validate_spinor(&fine_h[n], who, "fineEvec"))) return st;
    MUGIQ_REQUIRE(same_geometry(fine_h[n], fine_h[0]), "%s: fine field %d differs in geometry from field 0", who, n);
    MUGIQ_REQUIRE(coarse_h[n].data && coarse_h[n].stride == coarse_h[0].stride && coarse_h[n].parity_offset == coarse_h[0].parity_offset &&
                      coarse_h[n].precision == coarse_h[0].precision, "%s: coarse field %d differs from field 0", who, n);
  }
  MUGIQ_REQUIRE(fine_h[0].precision == transfer->precision, "%s: fine precision %d != transfer precision %d", who, fine_h[0].precision, transfer->precision);
  for (int d = 0; d < 4; d++) MUGIQ_REQUIRE(fine_h[0].X[d] == transfer->X[d], "%s: fine X[%d] mismatch", who, d);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int p = transfer->precision, o = fine_h[0].field_order;
  if (p == 8 && o == 2) {
    const int rc = prolong_mfma_plan(transfer, coarse_h, fine_h, nVec, s);  // the matrix-pipe form where the shape allows
    if (rc >= 0) return rc;
    return launch_prolong<double, double, 2, true, false>(transfer, coarse_h, fine_h, nullptr, nullptr, nVec, s);
  }
  if (p == 8 && o == 4) return launch_prolong<double, double, 4, true, false>(transfer, coarse_h, fine_h, nullptr, nullptr, nVec, s);
  if (p == 4 && o == 2) return launch_prolong<float, float, 2, true, false>(transfer, coarse_h, fine_h, nullptr, nullptr, nVec, s);
  return launch_prolong<float, float, 4, true, false>(transfer, coarse_h, fine_h, nullptr, nullptr, nVec, s);
}

int mugiq_hip_prolongate_coarse_batched(const MugiqHipCoarseField *out_h, const MugiqHipCoarseField *in_h, int nVec,
                                        const MugiqHipTransfer *T, void *stream) {
  if (int dbg_ = mugiq::debug_poison_lds_if_asked(static_cast<hipStream_t>(stream))) return dbg_;
  const char *who = "prolongateEvec(coarse level)";
  MUGIQ_REQUIRE(out_h && in_h && nVec >= 1, "%s: NULL / empty argument", who);
  MUGIQ_REQUIRE(T && T->V, "%s: Transfer operator for this level does not exist!", who);  // lib/loop_mugiq.cpp:309
  MUGIQ_REQUIRE(T->precision == 4 || T->precision == 8, "%s: transfer precision %d", who, T->precision);
  MUGIQ_REQUIRE(T->nVec >= 1 && T->nVec <= 96, "%s: n_vec = %d must be in [1, 96]", who, T->nVec);
  MUGIQ_REQUIRE(T->spinBlockSize == 1, "%s: spin_block_size = %d: a coarse level keeps both chiralities (1)", who, T->spinBlockSize);
  const MugiqHipCoarseField &o = out_h[0], &i = in_h[0];
  MUGIQ_REQUIRE(o.nSpin == 2 && i.nSpin == 2 && i.nColor == T->nVec && o.nColor >= 1 && o.nColor <= 96, "%s: fields must have nSpin 2; coarser nColor = n_vec = %d", who, T->nVec);
  long long vol = 1, volc = 1;
  for (int d = 0; d < 4; d++) {
    MUGIQ_REQUIRE(T->X[d] > 0 && (T->X[d] & 1) == 0 && o.X[d] == T->X[d], "%s: finer lattice X[%d] = %d (field: %d) must be positive, even and equal", who, d, T->X[d], o.X[d]);
    MUGIQ_REQUIRE(T->geoBlockSize[d] >= 1 && T->X[d] % T->geoBlockSize[d] == 0, "%s: geo_block_size[%d] = %d does not divide X = %d", who, d, T->geoBlockSize[d], T->X[d]);
    const int xc = T->X[d] / T->geoBlockSize[d];
    MUGIQ_REQUIRE((xc & 1) == 0 && i.X[d] == xc, "%s: coarser extent in dim %d must be even and equal X/block = %d (field: %d)", who, d, xc, i.X[d]);
    vol *= T->X[d];
    volc *= xc;
  }
  MUGIQ_REQUIRE(o.volumeCB == vol / 2 && i.volumeCB == volc / 2, "%s: volumeCB mismatch", who);
  MUGIQ_REQUIRE(T->stride >= vol / 2 && T->parity_offset >= (int64_t)2 * o.nColor * T->nVec * T->stride, "%s: V stride / parity_offset too small", who);
  for (int n = 0; n < nVec; n++) {
    MUGIQ_REQUIRE(out_h[n].data && in_h[n].data && out_h[n].data != in_h[n].data, "%s: field %d is NULL or aliased", who, n);
    MUGIQ_REQUIRE(out_h[n].precision == T->precision && in_h[n].precision == T->precision, "%s: field %d: precision differs from the transfer's", who, n);
    MUGIQ_REQUIRE(out_h[n].stride == o.stride && out_h[n].parity_offset == o.parity_offset && out_h[n].nColor == o.nColor &&
                      in_h[n].stride == i.stride && in_h[n].parity_offset == i.parity_offset && in_h[n].nColor == i.nColor,
                  "%s: field %d differs in layout from field 0", who, n);
  }
  MUGIQ_REQUIRE(o.stride >= o.volumeCB && o.parity_offset >= (int64_t)2 * o.nColor * o.stride && i.stride >= i.volumeCB &&
                    i.parity_offset >= (int64_t)2 * i.nColor * i.stride, "%s: field stride / parity_offset too small", who);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (T->precision == 8) return launch_prolong_coarse<double>(out_h, in_h, nVec, T, s);
  return launch_prolong_coarse<float>(out_h, in_h, nVec, T, s);
}

int mugiq_hip_prolongate_contract_batched(void *loopData_d, int loopPrecision, const MugiqHipCoarseField *coarse_h,
                                          const double *sigma_h, int nVec, const MugiqHipTransfer *transfer, void *stream) {
  if (int dbg_ = mugiq::debug_poison_lds_if_asked(static_cast<hipStream_t>(stream))) return dbg_;
  const char *who = "prolongateContract";
  MUGIQ_REQUIRE(loopData_d && coarse_h && sigma_h && nVec >= 1, "%s: NULL / empty argument", who);
  int st = validate_transfer(transfer, &coarse_h[0], who);
  if (st) return st;
  for (int n = 0; n < nVec; n++) {
    MUGIQ_REQUIRE(coarse_h[n].data && coarse_h[n].stride == coarse_h[0].stride && coarse_h[n].parity_offset == coarse_h[0].parity_offset &&
                      coarse_h[n].precision == coarse_h[0].precision, "%s: coarse field %d differs from field 0", who, n);
    MUGIQ_REQUIRE(sigma_h[n] != 0.0, "%s: sigma[%d] is zero", who, n);
  }
  const int p = transfer->precision;
  if (loopPrecision == 0) loopPrecision = p;
  MUGIQ_REQUIRE(loopPrecision == p || (loopPrecision == 8 && p == 4), "%s: loop precision %d with field precision %d is not supported", who, loopPrecision, p);
  hipStream_t s = static_cast<hipStream_t>(stream);
  // MUGIQ_HIP_MG_PLAN=direct keeps the per-eigenvector kernel (prolong every eigenvector, contract on the spot)
  const char *plan = getenv("MUGIQ_HIP_MG_PLAN");
  if (!(plan && strcmp(plan, "direct") == 0)) {
    int rc;
    if (p == 8) rc = coarse_plan<double, double>(transfer, coarse_h, sigma_h, loopData_d, nVec, s);
    else if (loopPrecision == 8) rc = coarse_plan<float, double>(transfer, coarse_h, sigma_h, loopData_d, nVec, s);
    else rc = coarse_plan<float, float>(transfer, coarse_h, sigma_h, loopData_d, nVec, s);
    if (rc >= 0) return rc;
  }
  if (p == 8) return launch_prolong<double, double, 2, false, true>(transfer, coarse_h, nullptr, sigma_h, loopData_d, nVec, s);
  if (loopPrecision == 8) return launch_prolong<float, double, 2, false, true>(transfer, coarse_h, nullptr, sigma_h, loopData_d, nVec, s);
  return launch_prolong<float, float, 2, false, true>(transfer, coarse_h, nullptr, sigma_h, loopData_d, nVec, s);
}

}  // extern "C"
