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

}  // namespace mugiq

using namespace mugiq;

extern "C" {

int mugiq_hip_prolongate_batched(const MugiqHipSpinorField *fine_h, const MugiqHipCoarseField *coarse_h, int nVec,
                                 const MugiqHipTransfer *transfer, void *stream) {
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
  if (p == 8 && o == 2) return launch_prolong<double, double, 2, true, false>(transfer, coarse_h, fine_h, nullptr, nullptr, nVec, s);
  if (p == 8 && o == 4) return launch_prolong<double, double, 4, true, false>(transfer, coarse_h, fine_h, nullptr, nullptr, nVec, s);
  if (p == 4 && o == 2) return launch_prolong<float, float, 2, true, false>(transfer, coarse_h, fine_h, nullptr, nullptr, nVec, s);
  return launch_prolong<float, float, 4, true, false>(transfer, coarse_h, fine_h, nullptr, nullptr, nVec, s);
}

int mugiq_hip_prolongate_contract_batched(void *loopData_d, int loopPrecision, const MugiqHipCoarseField *coarse_h,
                                          const double *sigma_h, int nVec, const MugiqHipTransfer *transfer, void *stream) {
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
  if (p == 8) return launch_prolong<double, double, 2, false, true>(transfer, coarse_h, nullptr, sigma_h, loopData_d, nVec, s);
  if (loopPrecision == 8) return launch_prolong<float, double, 2, false, true>(transfer, coarse_h, nullptr, sigma_h, loopData_d, nVec, s);
  return launch_prolong<float, float, 2, false, true>(transfer, coarse_h, nullptr, sigma_h, loopData_d, nVec, s);
}

}  // extern "C"
