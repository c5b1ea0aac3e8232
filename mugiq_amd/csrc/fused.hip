// Fused displaced loop contraction (the MI355X-first form of lib/loop_mugiq.cpp:485-497).
//
// The reference computes, per eigenvector and per displacement entry "+mu:start,stop",
//     R_0 = v;  R_k = D_mu R_{k-1}  (one kernel + one halo exchange + 2-3 field copies per step);
//     loop_k += sigma^-1 v^dag Gamma R_k   for start <= k <= stop      (one kernel + a global RMW per k)
// i.e. ~1.5 kB of HBM traffic per site, eigenvector and step.  A straight covariant displacement is a
// path-ordered link product applied to the shifted ORIGINAL vector:
//     R_k(x) = W_k(x) v(x + k mu),   W_k(x) = U_mu(x) U_mu(x+mu) ... U_mu(x+(k-1)mu)           (sign +)
//     R_k(x) = W_k(x) v(x - k mu),   W_k(x) = U^dag_mu(x-mu) ... U^dag_mu(x-k mu)              (sign -)
// and W_k is itself the k-fold displacement of the "identity" field E_0(x)(s,c) = delta_sc (s < 3), so it is
// built once per entry with the ordinary displacement kernel (csrc/displace.hip).  The kernel below then
// streams, per site and eigenvector, v(x) once and v(x +- k mu) once per slot k: 192 + 192 B (fp64) per slot,
// no displaced vector is ever written, and the 16 accumulators per slot stay in registers over the whole
// eigenvector block.  Each wave of a workgroup handles one slot k for the same 64 sites, so the v(x) read is
// shared through L1/L2.  Halo: `layers` = stop face layers of the original eigenvectors, exchanged once per
// entry and eigenvector block (pack_layers_kernel), instead of one exchange per step.
#include "internal.h"

#include <cstdlib>
#include <vector>

namespace mugiq {

constexpr int kFusedMaxSlots = 4;

// F = storage type of eigenvectors / links, A = arithmetic and loop-buffer type (F, or double over float storage)
template <typename F, typename A> struct FusedArgs {
  Cplx<A> *loop;            // first slot handled by this launch
  int64_t slot_stride;      // complex elements between consecutive slots (16*V)
  int overwrite;            // store instead of accumulate (MUGIQ_HIP_REGION_OVERWRITE)
  const void *const *L;     // device table of eigenvector bodies
  const A *inv_sigma;       // device [nVec]
  int nVec;
  int X[4];
  int volumeCB;
  int stride;
  int64_t parity_offset;
  const F *E[kFusedMaxSlots];  // W_k as a FLOAT2 spinor field (stride volumeCB): W[i][j] = E(spin j, colour i)
  int k[kFusedMaxSlots];
  int nslot;
  int partitioned;          // commDim[DIR]
  const F *ghost;           // [nVec][layers][2][12][faceCB] in the eigenvectors' field order
  int64_t ghost_vec_stride; // complex elements per eigenvector = layers*24*faceCB
  int faceCB;
  int xcdSwizzle;           // workgroups of one XCD cover one contiguous eighth of the (logical) workgroup order
  // Displacement-aligned workgroup order (0 = off): consecutive workgroups of an XCD step along the displacement axis
  // (and alternate parity), so v(x +- k mu) requested by one workgroup is the v(x) another one streams at about the
  // same time: the shifted reads become L2 hits instead of HBM reads.  x_cb = hi*(J*strideMu) + j*strideMu + lo.
  int remapJ;               // X[DIR]
  int remapS;               // strideMu / 64
  int strideMu;             // x_cb distance of one step along DIR = prod_{d<DIR} X[d] / 2
};

template <typename F, typename A, int ORDER, bool NT = false>
__device__ inline void load_lane(Cplx<A> v[12], const Cplx<F> *p, int64_t stride, int64_t idx) {
  typedef F vec2 __attribute__((ext_vector_type(2)));
  typedef F vec4 __attribute__((ext_vector_type(4)));
  if constexpr (ORDER == 2) {
#pragma unroll
    for (int k = 0; k < 12; k++) {
      const MUGIQ_GLOBAL vec2 *q = as_global(reinterpret_cast<const vec2 *>(p + k * stride + idx));
      const vec2 t = NT ? __builtin_nontemporal_load(q) : *q;
      v[k] = Cplx<A>{(A)t.x, (A)t.y};
    }
  } else {
#pragma unroll
    for (int j = 0; j < 6; j++) {
      const MUGIQ_GLOBAL vec4 *q = as_global(reinterpret_cast<const vec4 *>(p + (j * stride + idx) * 2));
      const vec4 t = NT ? __builtin_nontemporal_load(q) : *q;
      v[2 * j] = Cplx<A>{(A)t.x, (A)t.y};
      v[2 * j + 1] = Cplx<A>{(A)t.z, (A)t.w};
    }
  }
}

template <typename F, typename A, int ORDER, int DIR, int SIGN, bool NT>
__global__ __launch_bounds__(64 * kFusedMaxSlots) void fused_displaced_contract_kernel(FusedArgs<F, A> a) {
  const int V = 2 * a.volumeCB;
  int blk = blockIdx.x;
  if (a.xcdSwizzle) {
    const int per = gridDim.x >> 3;
    blk = (blk & 7) * per + (blk >> 3);
  }
  int site;
  if (a.remapJ > 0) {
    const int p = blk & 1;
    int r = blk >> 1;
    const int j = r % a.remapJ;
    r /= a.remapJ;
    const int c = r % a.remapS;
    const int hi = r / a.remapS;
    site = p * a.volumeCB + hi * (a.remapJ * a.strideMu) + j * a.strideMu + c * 64 + threadIdx.x;
  } else {
    site = blk * 64 + threadIdx.x;
  }
  const int slot = threadIdx.y;
  if (site >= V) return;
  const int pty = site >= a.volumeCB ? 1 : 0;
  const int x_cb = site - pty * a.volumeCB;
  const int k = a.k[slot];

  int coord[4];
  get_coords(coord, x_cb, a.X, pty);
  // ---- where v(x +- k mu) lives: body (periodic wrap) or ghost layer --------------------------------------
  const int nbrPty = (pty + k) & 1;
  int cn = coord[DIR] + (SIGN == MUGIQ_HIP_DISP_SIGN_PLUS ? k : -k);
  bool inGhost = false;
  int layer = 0;
  if (a.partitioned) {
    if (SIGN == MUGIQ_HIP_DISP_SIGN_PLUS && cn >= a.X[DIR]) {
      inGhost = true;
      layer = cn - a.X[DIR];          // forward neighbour's x[DIR] = layer
    } else if (SIGN == MUGIQ_HIP_DISP_SIGN_MINUS && cn < 0) {
      inGhost = true;
      layer = -cn - 1;                // backward neighbour's x[DIR] = X-1-layer
    }
  }
  int64_t nIdx, nStride, nOff;
  if (inGhost) {
    nIdx = ghost_face_index_on_face(coord, a.X, DIR);
    nStride = a.faceCB;
    nOff = (int64_t)layer * 24 * a.faceCB + (int64_t)nbrPty * 12 * a.faceCB;
  } else {
    int c[4] = {coord[0], coord[1], coord[2], coord[3]};
    cn %= a.X[DIR];
    if (cn < 0) cn += a.X[DIR];
    c[DIR] = cn;
    nIdx = lex_index(c, a.X) >> 1;
    nStride = a.stride;
    nOff = (int64_t)nbrPty * a.parity_offset;
  }

  // ---- W_k(x): 3x3 from the first 9 planes of E_k -----------------------------------------------------------
  Cplx<A> W[9];  // W[i*3+j]
  {
    const Cplx<F> *e = reinterpret_cast<const Cplx<F> *>(a.E[slot]) + (int64_t)pty * 12 * a.volumeCB + x_cb;
#pragma unroll
    for (int j = 0; j < 3; j++)
#pragma unroll
      for (int i = 0; i < 3; i++) {
        const Cplx<F> t = e[(int64_t)(j * 3 + i) * a.volumeCB];
        W[i * 3 + j] = Cplx<A>{(A)t.re, (A)t.im};
      }
  }

  Cplx<A> acc[16];
#pragma unroll
  for (int i = 0; i < 16; i++) acc[i] = Cplx<A>{A(0), A(0)};

  const Cplx<F> *ghostBase = reinterpret_cast<const Cplx<F> *>(a.ghost);
  for (int n = 0; n < a.nVec; n++) {
    const Cplx<F> *body = reinterpret_cast<const Cplx<F> *>(a.L[n]);
    Cplx<A> l[12], psi[12];
    load_lane<F, A, ORDER>(l, body + (int64_t)pty * a.parity_offset, a.stride, x_cb);
    const Cplx<F> *src = inGhost ? ghostBase + (int64_t)n * a.ghost_vec_stride : body;
    load_lane<F, A, ORDER, NT>(psi, src + nOff, nStride, nIdx);  // the shifted vector is read once per slot
    const A s = a.inv_sigma[n];
    Cplx<A> r[12];
#pragma unroll
    for (int sp = 0; sp < 4; sp++)
#pragma unroll
      for (int i = 0; i < 3; i++) {
        Cplx<A> t{A(0), A(0)};
#pragma unroll
        for (int j = 0; j < 3; j++) cmadd(t, W[i * 3 + j], psi[sp * 3 + j]);
        r[sp * 3 + i] = Cplx<A>{s * t.re, s * t.im};
      }
#pragma unroll
    for (int be = 0; be < 4; be++)
#pragma unroll
      for (int al = 0; al < 4; al++)
#pragma unroll
        for (int c = 0; c < 3; c++) cmadd_conj(acc[be * 4 + al], l[be * 3 + c], r[al * 3 + c]);
  }

  Cplx<A> *loop = a.loop + (int64_t)slot * a.slot_stride;
#pragma unroll
  for (int iG = 0; iG < 16; iG++) {
    Cplx<A> t{A(0), A(0)};
#pragma unroll
    for (int s2 = 0; s2 < 4; s2++) add_phase(t, kGammaPhase[iG][s2], acc[s2 * 4 + kGammaColumn[iG][s2]]);
    Cplx<A> *out = loop + (int64_t)V * iG + site;
    Cplx<A> o = a.overwrite ? Cplx<A>{A(0), A(0)} : *out;
    o.re += t.re;
    o.im += t.im;
    *out = o;
  }
}

// ---- batched multi-layer face packer ---------------------------------------------------------------------
template <typename F> struct PackLayersArgs {
  F *out;                   // [nVec][layers][2][12][faceCB]
  const void *const *src;   // device table of field bodies
  int X[4];
  int stride;
  int64_t parity_offset;
  int dim;
  int high;                 // 0: layers x[dim] = j ; 1: layers x[dim] = X-1-j
  int faceCB;
  int layers;
};

template <typename F, int ORDER>
__global__ __launch_bounds__(256) void pack_layers_kernel(PackLayersArgs<F> g) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * g.faceCB) return;
  const int j = blockIdx.y, n = blockIdx.z;
  const int pty = i >= g.faceCB ? 1 : 0;
  const int idx = i - pty * g.faceCB;
  const int r0 = g.dim == 0 ? 1 : 0, r1 = g.dim <= 1 ? 2 : 1;
  auto Xd = [&](int d) { return d == 0 ? g.X[0] : d == 1 ? g.X[1] : d == 2 ? g.X[2] : g.X[3]; };
  const int fixed = g.high ? Xd(g.dim) - 1 - j : j;
  int l = 2 * idx;
  const int c0 = l % Xd(r0);
  l /= Xd(r0);
  const int c1 = l % Xd(r1);
  const int c2 = l / Xd(r1);
  const int bit = (pty - (c0 + c1 + c2 + fixed)) & 1;
  int c[4];
#pragma unroll
  for (int d = 0; d < 4; d++) c[d] = d == g.dim ? fixed : d == r0 ? c0 + bit : d == r1 ? c1 : c2;
  Cplx<F> v[12];
  SpinorView<F, ORDER> src{const_cast<F *>(static_cast<const F *>(g.src[n])), g.stride, g.parity_offset};
  src.load(v, pty, lex_index(c, g.X) >> 1);
  SpinorView<F, ORDER> dst{g.out + 2 * ((int64_t)n * g.layers + j) * 24 * g.faceCB, g.faceCB, (int64_t)12 * g.faceCB};
  dst.store(v, pty, idx);
}

template <typename F, typename A, int ORDER>
static int launch_fused(FusedArgs<F, A> a, int dir, int sign, hipStream_t stream) {
  const int V = 2 * a.volumeCB;
  const dim3 grid((V + 63) / 64), block(64, a.nslot);
  int nt = 1, swz = 1, remap = 1;  // MUGIQ_HIP_FUSED_TUNE = "nt,swizzle,remap" overrides (profiles/r01_fused_tune.txt)
  if (const char *e = getenv("MUGIQ_HIP_FUSED_TUNE")) sscanf(e, "%d,%d,%d", &nt, &swz, &remap);
  a.xcdSwizzle = (swz && grid.x % 8 == 0) ? 1 : 0;
  long long strideMu = 1;
  for (int d = 0; d < dir; d++) strideMu *= a.X[d];
  strideMu /= 2;
  a.remapJ = 0;
  a.remapS = 0;
  a.strideMu = (int)strideMu;
  if (remap && dir >= 1 && strideMu % 64 == 0 && a.volumeCB % 64 == 0) {
    a.remapJ = a.X[dir];
    a.remapS = (int)(strideMu / 64);
    nt = 0;  // the shifted reads are meant to hit in L2
  }
#define MUGIQ_FUSED_CASE(D, S)                                                                                     \
  case (D)*2 + (S):                                                                                                \
    if (nt) hipLaunchKernelGGL((fused_displaced_contract_kernel<F, A, ORDER, D, S, true>), grid, block, 0, stream, a);  \
    else hipLaunchKernelGGL((fused_displaced_contract_kernel<F, A, ORDER, D, S, false>), grid, block, 0, stream, a);    \
    break;
  switch (dir * 2 + sign) {
    MUGIQ_FUSED_CASE(0, 0) MUGIQ_FUSED_CASE(0, 1) MUGIQ_FUSED_CASE(1, 0) MUGIQ_FUSED_CASE(1, 1)
    MUGIQ_FUSED_CASE(2, 0) MUGIQ_FUSED_CASE(2, 1) MUGIQ_FUSED_CASE(3, 0) MUGIQ_FUSED_CASE(3, 1)
  }
#undef MUGIQ_FUSED_CASE
  MUGIQ_CHECK_HIP(hipGetLastError());
  return MUGIQ_HIP_SUCCESS;
}

// third-generation (16-line tiles, two items per wave), csrc/fused_tile16.hip
bool tile16_applicable(const MugiqHipSpinorField &ev, int dir, int kmax, int precision, int partitioned, bool secondGenerationApplies);
template <typename F, typename A, int ORDER>
int tile16_entry(void *loop_d, const MugiqHipSpinorField *ev, const double *sigma, int nVec, const void *const *E_d, const int *kvals,
                 int nK, int dir, int sign, int partitioned, const void *ghost_d, int layers, int region, hipStream_t stream);
// second-generation (LDS-tiled) kernel, csrc/fused_tile.hip
bool tile_applicable(const MugiqHipSpinorField &ev, int dir, int kmax, int precision, int partitioned);
template <typename F, typename A, int ORDER>
int tile_entry(void *loop_d, const MugiqHipSpinorField *ev, const double *sigma, int nVec, const void *const *E_d, const int *kvals,
               int nK, int dir, int sign, int partitioned, const void *ghost_d, int layers, int region, hipStream_t stream,
               void *ultra_d, int *carried);

// fourth generation: the same tile in an axial gauge on the fp64 matrix pipe (any storage type, ascending lengths up to 8), csrc/fused_mfma.hip
bool mfma_tile_applicable(const MugiqHipSpinorField &ev, int dir, const int *kvals, int nK, int partitioned, bool gaugeGiven);
int mfma_tile_entry(void *loop_d, int loopPrecision, const MugiqHipSpinorField *ev, const double *sigma, int nVec, const void *const *E_d,
                    const int *kvals, int nK, int dir, int sign, int partitioned, const void *ghost_d, int layers, int region,
                    hipStream_t stream, void *ultra_d, int *carried);

template <typename F, typename A, int ORDER>
static int fused_entry(void *loop_d, const MugiqHipSpinorField *ev, const double *sigma, int nVec, const void *const *E_d,
                       const int *kvals, int nK, int dir, int sign, int partitioned, const void *ghost_d, int layers,
                       int region, hipStream_t stream, void *ultra_d, int *carried) {
  if (carried) *carried = 0;
  {
    int kmax = 0;
    for (int i = 0; i < nK; i++) kmax = kvals[i] > kmax ? kvals[i] : kmax;
    // (every storage type: the tile converts on the way into LDS and works in double; float slots are rounded once, on the way out)
    if (mfma_tile_applicable(ev[0], dir, kvals, nK, partitioned, axial_gauge_hint_matches(E_d[0], dir, sign, kmax)))
      return mfma_tile_entry(loop_d, (int)sizeof(A), ev, sigma, nVec, E_d, kvals, nK, dir, sign, partitioned, ghost_d, layers, region, stream, ultra_d, carried);
    const bool gen2 = tile_applicable(ev[0], dir, kmax, ev[0].precision, partitioned);
    if (tile16_applicable(ev[0], dir, kmax, ev[0].precision, partitioned, gen2))
      return tile16_entry<F, A, ORDER>(loop_d, ev, sigma, nVec, E_d, kvals, nK, dir, sign, partitioned, ghost_d, layers, region, stream);
    if (gen2)
      return tile_entry<F, A, ORDER>(loop_d, ev, sigma, nVec, E_d, kvals, nK, dir, sign, partitioned, ghost_d, layers, region, stream, ultra_d, carried);
  }
  // the streaming kernel has no interior / boundary split: when the dimension is partitioned it counts as boundary
  const int overwrite = (region & MUGIQ_HIP_REGION_OVERWRITE) ? 1 : 0;
  region &= 0xff;
  if (region == MUGIQ_HIP_REGION_INTERIOR && partitioned) return MUGIQ_HIP_SUCCESS;
  if (region == MUGIQ_HIP_REGION_BOUNDARY && !partitioned) return MUGIQ_HIP_SUCCESS;
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
  FusedArgs<F, A> a;
  a.slot_stride = (int64_t)16 * 2 * ev[0].volumeCB;
  a.overwrite = overwrite;
  a.L = reinterpret_cast<const void *const *>(dev);
  a.inv_sigma = reinterpret_cast<const A *>(static_cast<unsigned char *>(dev) + ptr_bytes);
  a.nVec = nVec;
  for (int d = 0; d < 4; d++) a.X[d] = ev[0].X[d];
  a.volumeCB = ev[0].volumeCB;
  a.stride = ev[0].stride;
  a.parity_offset = ev[0].parity_offset;
  a.partitioned = partitioned;
  a.ghost = static_cast<const F *>(ghost_d);
  a.faceCB = ev[0].volumeCB / ev[0].X[dir];
  a.ghost_vec_stride = (int64_t)layers * 24 * a.faceCB;
  for (int k0 = 0; k0 < nK; k0 += kFusedMaxSlots) {
    a.nslot = (nK - k0 < kFusedMaxSlots) ? nK - k0 : kFusedMaxSlots;
    a.loop = static_cast<Cplx<A> *>(loop_d) + (int64_t)k0 * a.slot_stride;
    for (int s = 0; s < kFusedMaxSlots; s++) {
      a.E[s] = static_cast<const F *>(E_d[k0 + (s < a.nslot ? s : 0)]);
      a.k[s] = kvals[k0 + (s < a.nslot ? s : 0)];
    }
    st = launch_fused<F, A, ORDER>(a, dir, sign, stream);
    if (st) return st;
  }
  return MUGIQ_HIP_SUCCESS;
}

template <typename F, int ORDER>
static int pack_layers(void *out_d, const MugiqHipSpinorField *ev, int nVec, int dim, int high, int layers, hipStream_t stream) {
  std::vector<const void *> host(nVec);
  for (int n = 0; n < nVec; n++) host[n] = ev[n].data;
  void *dev = nullptr;
  int st = upload_table(&dev, host.data(), sizeof(void *) * (size_t)nVec, stream);
  if (st) return st;
  PackLayersArgs<F> g;
  g.out = static_cast<F *>(out_d);
  g.src = reinterpret_cast<const void *const *>(dev);
  for (int d = 0; d < 4; d++) g.X[d] = ev[0].X[d];
  g.stride = ev[0].stride;
  g.parity_offset = ev[0].parity_offset;
  g.dim = dim;
  g.high = high;
  g.faceCB = ev[0].volumeCB / ev[0].X[dim];
  g.layers = layers;
  const dim3 grid((2 * g.faceCB + 255) / 256, layers, nVec);
  hipLaunchKernelGGL((pack_layers_kernel<F, ORDER>), grid, dim3(256), 0, stream, g);
  MUGIQ_CHECK_HIP(hipGetLastError());
  return MUGIQ_HIP_SUCCESS;
}

}  // namespace mugiq

using namespace mugiq;

extern "C" {

int mugiq_hip_pack_face_layers(void *faces_d, const MugiqHipSpinorField *eVecs_h, int nVec, int dim, int high, int layers,
                               void *stream) {
  const char *who = "mugiq_hip_pack_face_layers";
  MUGIQ_REQUIRE(faces_d && eVecs_h && nVec >= 1, "%s: NULL / empty argument", who);
  MUGIQ_REQUIRE(dim >= 0 && dim < 4 && (high == 0 || high == 1), "%s: invalid dim %d / high %d", who, dim, high);
  for (int n = 0; n < nVec; n++) {
    int st = validate_spinor(&eVecs_h[n], who, "eVecs");
    if (st) return st;
    MUGIQ_REQUIRE(same_geometry(eVecs_h[n], eVecs_h[0]), "%s: eigenvector %d differs in geometry from eigenvector 0", who, n);
  }
  MUGIQ_REQUIRE(layers >= 1 && layers <= eVecs_h[0].X[dim], "%s: layers = %d must be in [1, X[dim] = %d]", who, layers,
                eVecs_h[0].X[dim]);
  MUGIQ_REQUIRE(nVec <= 65535 && layers <= 65535, "%s: nVec / layers exceed the grid limits", who);
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int p = eVecs_h[0].precision, o = eVecs_h[0].field_order;
  if (p == 8 && o == 2) return pack_layers<double, 2>(faces_d, eVecs_h, nVec, dim, high, layers, s);
  if (p == 8 && o == 4) return pack_layers<double, 4>(faces_d, eVecs_h, nVec, dim, high, layers, s);
  if (p == 4 && o == 2) return pack_layers<float, 2>(faces_d, eVecs_h, nVec, dim, high, layers, s);
  return pack_layers<float, 4>(faces_d, eVecs_h, nVec, dim, high, layers, s);
}

int mugiq_hip_displaced_loop_contraction_fused_carry(void *loopData_d, int loopPrecision, const MugiqHipSpinorField *eVecs_h,
                                                     const double *sigma_h, int nVec, const void *const *pathLinkFields_h,
                                                     const int *kValues_h, int nK, int dispDir, int dispSign,
                                                     const int commDim[4], const void *ghostLayers_d, int layers, int region,
                                                     void *ultraLocalSlot_d, int *carried, void *stream) {
  if (int dbg_ = mugiq::debug_poison_lds_if_asked(static_cast<hipStream_t>(stream))) return dbg_;
  if (carried) *carried = 0;
  const char *who = "mugiq_hip_displaced_loop_contraction_fused";
  MUGIQ_REQUIRE((region & 0xff) == MUGIQ_HIP_REGION_ALL || (region & 0xff) == MUGIQ_HIP_REGION_INTERIOR || (region & 0xff) == MUGIQ_HIP_REGION_BOUNDARY,
                "%s: invalid region %d", who, region);
  MUGIQ_REQUIRE((region & ~(0xff | MUGIQ_HIP_REGION_OVERWRITE)) == 0, "%s: invalid region flags %d", who, region);
  MUGIQ_REQUIRE(ultraLocalSlot_d == nullptr || carried != nullptr, "%s: ultraLocalSlot_d given without `carried` (the caller could not tell whether the slot was produced)", who);
  MUGIQ_REQUIRE(loopData_d && eVecs_h && sigma_h && pathLinkFields_h && kValues_h, "%s: NULL argument", who);
  MUGIQ_REQUIRE(nVec >= 1 && nK >= 1, "%s: nVec = %d, nK = %d must be >= 1", who, nVec, nK);
  MUGIQ_REQUIRE(dispDir >= 0 && dispDir < 4 && (dispSign == 0 || dispSign == 1), "%s: Got invalid dispDir and/or dispSign.", who);
  for (int n = 0; n < nVec; n++) {
    int st = validate_spinor(&eVecs_h[n], who, "eVecs");
    if (st) return st;
    MUGIQ_REQUIRE(same_geometry(eVecs_h[n], eVecs_h[0]), "%s: eigenvector %d differs in geometry from eigenvector 0", who, n);
    MUGIQ_REQUIRE(sigma_h[n] != 0.0, "%s: sigma[%d] is zero", who, n);
  }
  const int part = commDim ? (commDim[dispDir] != 0) : 0;
  int kmax = 0;
  for (int i = 0; i < nK; i++) {
    MUGIQ_REQUIRE(kValues_h[i] >= 1, "%s: displacement length %d must be >= 1", who, kValues_h[i]);
    MUGIQ_REQUIRE(pathLinkFields_h[i] != nullptr, "%s: pathLinkFields_h[%d] is NULL", who, i);
    if (kValues_h[i] > kmax) kmax = kValues_h[i];
  }
  if (part && (region & 0xff) != MUGIQ_HIP_REGION_INTERIOR) {
    MUGIQ_REQUIRE(ghostLayers_d != nullptr, "%s: dim %d is partitioned but ghostLayers_d is NULL (halo exchange missing)", who,
                  dispDir);
    MUGIQ_REQUIRE(layers >= kmax && layers <= eVecs_h[0].X[dispDir],
                  "%s: need %d ghost layers (<= local extent %d), got %d", who, kmax, eVecs_h[0].X[dispDir], layers);
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  const int p = eVecs_h[0].precision, o = eVecs_h[0].field_order;
  if (loopPrecision == 0) loopPrecision = p;
  MUGIQ_REQUIRE(loopPrecision == p || (loopPrecision == 8 && p == 4),
                "%s: loop precision %d with field precision %d is not supported", who, loopPrecision, p);
#define MUGIQ_FUSED_GO(F, A, O)                                                                                                 \
  return fused_entry<F, A, O>(loopData_d, eVecs_h, sigma_h, nVec, pathLinkFields_h, kValues_h, nK, dispDir, dispSign, part,   \
                              ghostLayers_d, layers, region, s, ultraLocalSlot_d, carried)
  if (p == 8 && o == 2) MUGIQ_FUSED_GO(double, double, 2);
  if (p == 8 && o == 4) MUGIQ_FUSED_GO(double, double, 4);
  if (loopPrecision == 8 && o == 2) MUGIQ_FUSED_GO(float, double, 2);
  if (loopPrecision == 8) MUGIQ_FUSED_GO(float, double, 4);
  if (o == 2) MUGIQ_FUSED_GO(float, float, 2);
  MUGIQ_FUSED_GO(float, float, 4);
#undef MUGIQ_FUSED_GO
}

int mugiq_hip_displaced_loop_contraction_fused_region(void *loopData_d, int loopPrecision, const MugiqHipSpinorField *eVecs_h,
                                                      const double *sigma_h, int nVec, const void *const *pathLinkFields_h,
                                                      const int *kValues_h, int nK, int dispDir, int dispSign,
                                                      const int commDim[4], const void *ghostLayers_d, int layers, int region,
                                                      void *stream) {
  return mugiq_hip_displaced_loop_contraction_fused_carry(loopData_d, loopPrecision, eVecs_h, sigma_h, nVec, pathLinkFields_h, kValues_h,
                                                          nK, dispDir, dispSign, commDim, ghostLayers_d, layers, region, nullptr, nullptr,
                                                          stream);
}

int mugiq_hip_displaced_loop_contraction_fused_mixed(void *loopData_d, int loopPrecision, const MugiqHipSpinorField *eVecs_h,
                                                     const double *sigma_h, int nVec, const void *const *pathLinkFields_h,
                                                     const int *kValues_h, int nK, int dispDir, int dispSign,
                                                     const int commDim[4], const void *ghostLayers_d, int layers, void *stream) {
  return mugiq_hip_displaced_loop_contraction_fused_region(loopData_d, loopPrecision, eVecs_h, sigma_h, nVec, pathLinkFields_h,
                                                           kValues_h, nK, dispDir, dispSign, commDim, ghostLayers_d, layers,
                                                           MUGIQ_HIP_REGION_ALL, stream);
}

int mugiq_hip_displaced_loop_contraction_fused(void *loopData_d, const MugiqHipSpinorField *eVecs_h, const double *sigma_h,
                                               int nVec, const void *const *pathLinkFields_h, const int *kValues_h, int nK,
                                               int dispDir, int dispSign, const int commDim[4], const void *ghostLayers_d,
                                               int layers, void *stream) {
  return mugiq_hip_displaced_loop_contraction_fused_mixed(loopData_d, 0, eVecs_h, sigma_h, nVec, pathLinkFields_h, kValues_h, nK,
                                                          dispDir, dispSign, commDim, ghostLayers_d, layers, stream);
}

}  // extern "C"
