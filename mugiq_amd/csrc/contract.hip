// a1/a2: 16-gamma loop contraction, eigenvector-batched.
//
// Reference: loopContract_kernel (lib/mugiq_contract_kernels.cu:45-122) launched once per eigenvector by
// performLoopContraction (lib/contract_wrappers.cu:88-115) with a (16,2,16) block, LDS staging done by
// one z-thread in 16 and a global read-modify-write of the 16-gamma accumulator per eigenvector.
//
// MI355X design: one lattice site per lane, sites coalesced over the even-odd index (each of the 12
// FLOAT2 planes is read as one 1 KiB request per wave), the eigenvector loop INSIDE the kernel.  Because
// the gamma trace is linear, the per-eigenvector work is only the colour-traced spin matrix
//   resG[be][al] += (1/sigma_n) * sum_c conj(vL_n[be,c]) * vR_n[al,c]           (:98-105)
// held in registers; the 16 sparse gamma traces (:110-117) are taken once after the loop and added to
// loopData (:120).  HBM traffic is the algorithmic minimum: 24*sizeof(F) per site per eigenvector
// (12 complex, once) plus one read+write of the 16 outputs.  When vL == vR (ultra-local loop) resG is
// Hermitian and only its upper triangle is accumulated.
#include "internal.h"

#include <cstdlib>
#include <type_traits>
#include <vector>

namespace mugiq {


// F = storage type of the eigenvectors, A = arithmetic / accumulation type (= F, or double over float storage:
// the mixed-precision mode of BASELINE.json configs[3])
template <typename A> struct ContractArgs {
  Cplx<A> *loop;           // [16][V]
  const void *const *L;    // device table: nVec field bodies
  const void *const *R;    // device table (unused when SAME)
  const A *inv_sigma;      // device [nVec]
  int nVec;
  int volumeCB;
  int stride;
  int64_t parity_offset;
  int xcdSwizzle;          // 1: workgroups of one XCD (blockIdx % 8) cover one contiguous eighth of the sites
};

// streaming load of one site's 12 complex; NT = non-temporal (the eigenvectors are read exactly once)
template <typename F, typename A, int ORDER, bool NT>
__device__ inline void load_spinor(Cplx<A> v[12], const void *body, int64_t parity_offset, int stride, int parity, int x_cb) {
  typedef F vec2 __attribute__((ext_vector_type(2)));
  typedef F vec4 __attribute__((ext_vector_type(4)));
  const Cplx<F> *p = static_cast<const Cplx<F> *>(body) + parity * parity_offset;
  if constexpr (ORDER == 2) {
#pragma unroll
    for (int k = 0; k < 12; k++) {
      const MUGIQ_GLOBAL vec2 *q = as_global(reinterpret_cast<const vec2 *>(p + (int64_t)k * stride + x_cb));
      vec2 t = NT ? __builtin_nontemporal_load(q) : *q;
      v[k] = Cplx<A>{(A)t.x, (A)t.y};
    }
  } else {
#pragma unroll
    for (int j = 0; j < 6; j++) {
      const MUGIQ_GLOBAL vec4 *q = as_global(reinterpret_cast<const vec4 *>(p + ((int64_t)j * stride + x_cb) * 2));
      vec4 t = NT ? __builtin_nontemporal_load(q) : *q;
      v[2 * j] = Cplx<A>{(A)t.x, (A)t.y};
      v[2 * j + 1] = Cplx<A>{(A)t.z, (A)t.w};
    }
  }
}

// DEPTH eigenvectors are in flight per lane (DEPTH-1 loads issued ahead of the one being consumed).
template <typename F, typename A, int ORDER, bool SAME, int BLOCK, int DEPTH, bool NT>
__global__ __launch_bounds__(BLOCK) void loop_contract_kernel(ContractArgs<A> a) {
  const int V = 2 * a.volumeCB;
  int blk = blockIdx.x;
  if (a.xcdSwizzle) {  // dispatch deals workgroups round-robin over the 8 XCDs: give each XCD a contiguous site range
    const int per = gridDim.x >> 3;
    blk = (blk & 7) * per + (blk >> 3);
  }
  const int site = blk * BLOCK + threadIdx.x;  // tid = x_cb + parity*volumeCB  (:52)
  if (site >= V) return;
  const int parity = site >= a.volumeCB ? 1 : 0;
  const int x_cb = site - parity * a.volumeCB;

  Cplx<A> acc[16];
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = Cplx<A>{A(0), A(0)};

  if constexpr (SAME) {
    A diag[4] = {A(0), A(0), A(0), A(0)};
    Cplx<A> up[6];
#pragma unroll
    for (int i = 0; i < 6; i++) up[i] = Cplx<A>{A(0), A(0)};
    Cplx<A> v[DEPTH][12];
#pragma unroll
    for (int j = 0; j < DEPTH - 1; j++)
      if (j < a.nVec) load_spinor<F, A, ORDER, NT>(v[j], a.L[j], a.parity_offset, a.stride, parity, x_cb);
    // steady state without data-dependent branches (a conditional load makes hipcc's wait-count pass fall back to
    // vmcnt(0), which defeats the prefetch), then a guarded tail
    int n = 0;
    for (; n + 2 * DEPTH - 1 <= a.nVec; n += DEPTH) {
#pragma unroll
      for (int j = 0; j < DEPTH; j++) {
        const int m = n + j;
        if constexpr (DEPTH > 1) load_spinor<F, A, ORDER, NT>(v[(j + DEPTH - 1) % DEPTH], a.L[m + DEPTH - 1], a.parity_offset, a.stride, parity, x_cb);
        else load_spinor<F, A, ORDER, NT>(v[0], a.L[m], a.parity_offset, a.stride, parity, x_cb);
        accumulate_herm(diag, up, v[j], a.inv_sigma[m]);
      }
    }
    for (; n < a.nVec; n += DEPTH) {
#pragma unroll
      for (int j = 0; j < DEPTH; j++) {
        const int m = n + j;
        if (m < a.nVec) {
          const int pre = m + DEPTH - 1;
          if (pre < a.nVec && (DEPTH == 1 || pre >= DEPTH - 1)) load_spinor<F, A, ORDER, NT>(v[(j + DEPTH - 1) % DEPTH], a.L[pre], a.parity_offset, a.stride, parity, x_cb);
          accumulate_herm(diag, up, v[j], a.inv_sigma[m]);
        }
      }
    }
    int p = 0;
#pragma unroll
    for (int be = 0; be < 4; be++) {
      acc[be * 4 + be] = Cplx<A>{diag[be], A(0)};
#pragma unroll
      for (int al = be + 1; al < 4; al++) {
        acc[be * 4 + al] = up[p];
        acc[al * 4 + be] = Cplx<A>{up[p].re, -up[p].im};
        p++;
      }
    }
  } else {
    Cplx<A> l[DEPTH][12], r[DEPTH][12];
#pragma unroll
    for (int j = 0; j < DEPTH - 1; j++)
      if (j < a.nVec) {
        load_spinor<F, A, ORDER, NT>(l[j], a.L[j], a.parity_offset, a.stride, parity, x_cb);
        load_spinor<F, A, ORDER, NT>(r[j], a.R[j], a.parity_offset, a.stride, parity, x_cb);
      }
    for (int n = 0; n < a.nVec; n += DEPTH) {
#pragma unroll
      for (int j = 0; j < DEPTH; j++) {
        const int m = n + j;
        if (m < a.nVec) {
          const int pre = m + DEPTH - 1;
          if (pre < a.nVec) {
            load_spinor<F, A, ORDER, NT>(l[(j + DEPTH - 1) % DEPTH], a.L[pre], a.parity_offset, a.stride, parity, x_cb);
            load_spinor<F, A, ORDER, NT>(r[(j + DEPTH - 1) % DEPTH], a.R[pre], a.parity_offset, a.stride, parity, x_cb);
          }
          accumulate_full(acc, l[j], r[j], a.inv_sigma[m]);
        }
      }
    }
  }
  trace_and_store(a.loop, acc, V, site);
}

// Launch configuration.  Defaults were picked by sweeping on MI355X (profiles/); MUGIQ_HIP_CONTRACT_TUNE="block,depth,nt"
// overrides them for experiments (e.g. "512,3,1").
struct ContractTune {
  int block, depth, nt, swz;
};
static ContractTune contract_tune(bool same, bool fp64Storage) {
  // sweep on MI355X, 32^4 x 200 fp64 (profiles/r01_contract_sweep.txt): non-temporal loads +4 %; block size and prefetch
  // depth within 1 % of each other INSIDE one process (4-6 waves/SIMD already cover the latency); XCD-contiguous order +2 %.
  // Across processes the kernel's time moves with where its 40 GB landed (profiles/r03_headline_layout.txt), and there the
  // variants differ: over 8 fresh processes each, workgroups of 512 with three eigenvectors in flight averaged 6.65 ms (6.40-6.86,
  // 0.91-0.98 of the read probe of their process) against 6.83 ms (6.51-7.08, 0.86-0.94) for 256 / two
  // (profiles/r03_headline_tune.txt) -- faster on average and less exposed to a bad placement.  fp64 storage only: the other
  // storage types showed no such difference (profiles/r02_contract_cfg3_sweep.txt) and the mixed mode is built for 256.
  ContractTune t{256, 2, 1, 1};
  if (same && fp64Storage) t = ContractTune{512, 3, 1, 1};
  if (const char *e = getenv("MUGIQ_HIP_CONTRACT_TUNE")) {
    int b = 0, d = 0, n = 0, w = 0;
    if (sscanf(e, "%d,%d,%d,%d", &b, &d, &n, &w) >= 3 && (b == 64 || b == 128 || b == 256 || b == 512) && d >= 1 && d <= 3 && (n == 0 || n == 1)) {
      t.block = b;
      t.depth = (!same && d > 2) ? 2 : d;
      t.nt = n;
      t.swz = w ? 1 : 0;
    }
  }
  return t;
}

template <typename F, typename A, int ORDER, bool SAME, int BLOCK, int DEPTH>
static void launch_variant(const ContractArgs<A> &a, int nt, hipStream_t stream) {
  const int V = 2 * a.volumeCB;
  const dim3 grid((V + BLOCK - 1) / BLOCK), block(BLOCK);
  if (nt) hipLaunchKernelGGL((loop_contract_kernel<F, A, ORDER, SAME, BLOCK, DEPTH, true>), grid, block, 0, stream, a);
  else hipLaunchKernelGGL((loop_contract_kernel<F, A, ORDER, SAME, BLOCK, DEPTH, false>), grid, block, 0, stream, a);
}

template <typename F, typename A, int ORDER, bool SAME, int BLOCK>
static void launch_depth(const ContractArgs<A> &a, const ContractTune &t, hipStream_t stream) {
  if (t.depth == 1) launch_variant<F, A, ORDER, SAME, BLOCK, 1>(a, t.nt, stream);
  else if (t.depth == 2 || !SAME) launch_variant<F, A, ORDER, SAME, BLOCK, 2>(a, t.nt, stream);
  else launch_variant<F, A, ORDER, SAME, BLOCK, (SAME ? 3 : 2)>(a, t.nt, stream);
}

template <typename F, typename A, int ORDER, bool SAME>
static void launch_block(const ContractArgs<A> &a, const ContractTune &t, hipStream_t stream) {
  if constexpr (std::is_same<F, A>::value) {
    switch (t.block) {
    case 64: launch_depth<F, A, ORDER, SAME, 64>(a, t, stream); return;
    case 128: launch_depth<F, A, ORDER, SAME, 128>(a, t, stream); return;
    case 512: launch_depth<F, A, ORDER, SAME, 512>(a, t, stream); return;
    default: break;
    }
  }
  launch_depth<F, A, ORDER, SAME, 256>(a, t, stream);  // the mixed mode is built for the default block size only
}

template <typename F, typename A, int ORDER>
static int launch_contract(void *loop_d, const MugiqHipSpinorField *L, const MugiqHipSpinorField *R, const double *sigma,
                           int nVec, bool same, hipStream_t stream) {
  // device tables: [L pointers][R pointers][inv_sigma]
  const size_t ptr_bytes = sizeof(void *) * (size_t)nVec;
  const size_t tab_bytes = 2 * ptr_bytes + sizeof(A) * (size_t)nVec;
  std::vector<unsigned char> host(tab_bytes);
  const void **hl = reinterpret_cast<const void **>(host.data());
  const void **hr = reinterpret_cast<const void **>(host.data() + ptr_bytes);
  A *hs = reinterpret_cast<A *>(host.data() + 2 * ptr_bytes);
  for (int n = 0; n < nVec; n++) {
    hl[n] = L[n].data;
    hr[n] = R[n].data;
    const F sg = static_cast<F>(sigma[n]);   // (Float) eVals_sigma[n]   lib/loop_mugiq.cpp:479
    hs[n] = static_cast<A>(1.0 / sg);        // inv_sigma(1.0/sigma)     include/contract_util.cuh:132
  }
  void *dev = nullptr;
  int st = upload_table(&dev, host.data(), tab_bytes, stream);
  if (st) return st;

  ContractArgs<A> a;
  a.loop = static_cast<Cplx<A> *>(loop_d);
  a.L = reinterpret_cast<const void *const *>(dev);
  a.R = reinterpret_cast<const void *const *>(static_cast<unsigned char *>(dev) + ptr_bytes);
  a.inv_sigma = reinterpret_cast<const A *>(static_cast<unsigned char *>(dev) + 2 * ptr_bytes);
  a.nVec = nVec;
  a.volumeCB = L[0].volumeCB;
  a.stride = L[0].stride;
  a.parity_offset = L[0].parity_offset;
  const ContractTune t = contract_tune(same, std::is_same<F, double>::value && std::is_same<A, double>::value);
  a.xcdSwizzle = (t.swz && (((2 * a.volumeCB + t.block - 1) / t.block) % 8 == 0)) ? 1 : 0;
  if (same) launch_block<F, A, ORDER, true>(a, t, stream);
  else launch_block<F, A, ORDER, false>(a, t, stream);
  MUGIQ_CHECK_HIP(hipGetLastError());
  return MUGIQ_HIP_SUCCESS;
}

static int contract_dispatch(void *loop_d, int loopPrecision, const MugiqHipSpinorField *L, const MugiqHipSpinorField *R,
                             const double *sigma, int nVec, void *stream, const char *who) {
  MUGIQ_REQUIRE(loop_d != nullptr, "%s: loopData_d is NULL", who);
  MUGIQ_REQUIRE(L != nullptr && R != nullptr && sigma != nullptr, "%s: NULL argument", who);
  MUGIQ_REQUIRE(nVec >= 1, "%s: nVec = %d must be >= 1", who, nVec);
  bool same = true;
  for (int n = 0; n < nVec; n++) {
    int st = validate_spinor(&L[n], who, "eVecL");
    if (st) return st;
    st = validate_spinor(&R[n], who, "eVecR");
    if (st) return st;
    MUGIQ_REQUIRE(same_geometry(L[n], L[0]) && same_geometry(R[n], L[0]),
                  "%s: eigenvector %d differs in precision, field order or geometry from eigenvector 0", who, n);
    MUGIQ_REQUIRE(sigma[n] != 0.0, "%s: sigma[%d] is zero", who, n);
    same = same && (L[n].data == R[n].data);
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int prec = L[0].precision, order = L[0].field_order;
  if (loopPrecision == 0) loopPrecision = prec;
  MUGIQ_REQUIRE(loopPrecision == prec || (loopPrecision == 8 && prec == 4),
                "%s: loop precision %d with field precision %d is not supported (same precision, or fp64 loops over fp32 fields)",
                who, loopPrecision, prec);
  if (prec == 8 && order == 2) return launch_contract<double, double, 2>(loop_d, L, R, sigma, nVec, same, s);
  if (prec == 8 && order == 4) return launch_contract<double, double, 4>(loop_d, L, R, sigma, nVec, same, s);
  if (loopPrecision == 8) {
    if (order == 2) return launch_contract<float, double, 2>(loop_d, L, R, sigma, nVec, same, s);
    return launch_contract<float, double, 4>(loop_d, L, R, sigma, nVec, same, s);
  }
  if (order == 2) return launch_contract<float, float, 2>(loop_d, L, R, sigma, nVec, same, s);
  return launch_contract<float, float, 4>(loop_d, L, R, sigma, nVec, same, s);
}

}  // namespace mugiq

extern "C" {

int mugiq_hip_perform_loop_contraction(void *loopData_d, const MugiqHipSpinorField *eVecL, const MugiqHipSpinorField *eVecR,
                                       double sigma, void *stream) {
  return mugiq::contract_dispatch(loopData_d, 0, eVecL, eVecR, &sigma, 1, stream, "performLoopContraction");
}

int mugiq_hip_perform_loop_contraction_batched(void *loopData_d, const MugiqHipSpinorField *eVecL_h,
                                               const MugiqHipSpinorField *eVecR_h, const double *sigma_h, int nVec,
                                               void *stream) {
  return mugiq::contract_dispatch(loopData_d, 0, eVecL_h, eVecR_h, sigma_h, nVec, stream, "performLoopContractionBatched");
}

int mugiq_hip_perform_loop_contraction_batched_mixed(void *loopData_d, int loopPrecision, const MugiqHipSpinorField *eVecL_h,
                                                     const MugiqHipSpinorField *eVecR_h, const double *sigma_h, int nVec,
                                                     void *stream) {
  if (int dbg_ = mugiq::debug_poison_lds_if_asked(static_cast<hipStream_t>(stream))) return dbg_;
  return mugiq::contract_dispatch(loopData_d, loopPrecision, eVecL_h, eVecR_h, sigma_h, nVec, stream,
                                  "performLoopContractionBatchedMixed");
}

}  // extern "C"
