// Internal helpers shared by the HIP translation units of libmugiq_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "mugiq_hip.h"

namespace mugiq {

// ---- error plumbing ------------------------------------------------------------------------------
int set_error(int status, const char *fmt, ...);
#define MUGIQ_CHECK_HIP(call)                                                                     \
  do {                                                                                            \
    hipError_t e_ = (call);                                                                       \
    if (e_ != hipSuccess)                                                                         \
      return ::mugiq::set_error(MUGIQ_HIP_ERROR_HIP, "%s:%d: %s failed: %s", __FILE__, __LINE__,  \
                                #call, hipGetErrorString(e_));                                    \
  } while (0)
#define MUGIQ_REQUIRE(cond, ...)                                                                  \
  do {                                                                                            \
    if (!(cond)) return ::mugiq::set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, __VA_ARGS__);        \
  } while (0)

// Per-(device, stream) scratch (host_api.cpp).  Small tables (pointer lists, 1/sigma, momenta) go through
// upload_table: a pinned staging copy, then an H2D copy ordered on `stream`, so the caller's memory may be released
// on return.  Successive calls on one stream reuse the same buffer -- safe because the stream orders the next upload
// behind the previous kernel; calls on different streams (or threads, or Loop objects) never share one.
// stream_scratch returns the table buffer itself (at least `bytes` large; the next upload_table of at most that size
// on the same stream lands at the same address); stream_workspace is a second, independent buffer of the same arena.
int stream_scratch(void **ptr, size_t bytes, hipStream_t stream);
int stream_workspace(void **ptr, size_t bytes, hipStream_t stream);
int upload_table(void **dev, const void *host, size_t bytes, hipStream_t stream);
int release_stream_scratch(hipStream_t stream);
size_t eo_dft_x_lds_bytes(int precision, const int localL[4], int nPx, int *redOffsetElems);  // momproj.hip
int eo_dft_x_time_chunk(int precision, const int localL[4], int nPx);                            // momproj.hip: 0 = the fused x step does not apply
int fill_identity_links(const MugiqHipSpinorField *f, hipStream_t stream);  // displace.hip
// MUGIQ_HIP_DEBUG_POISON_LDS=1 (test aid): every compute entry point of the C ABI first overwrites the LDS of all CUs with NaN bit
// patterns (mugiq_hip_debug_poison_lds), so that a kernel reading a cell it never wrote shows it in its result.
int debug_poison_lds_if_asked(hipStream_t stream);
// csrc/fused_mfma.hip.  The axial gauge of a (direction, sign) is rebuilt by every launch of the matrix-pipe tile (one pass over
// W_1, 0.2 ms) -- unless the caller, who launches the same entry several times (the driver: interior tiles, then the boundary tiles
// block by block), has built it once and says so: axial_gauge_bytes = 0 where that tile does not apply; the hint is per host
// thread, names the W_1 field it was built from, and is cleared with G_d = NULL.
size_t axial_gauge_bytes(const MugiqHipSpinorField &ev, int dir, const int *kvals, int nK, int partitioned);
int build_axial_gauge(void *G_d, const MugiqHipSpinorField &ev, const void *const *E_d, int kmax, int dir, int sign, hipStream_t stream);
// ... straight from the gauge field (no path-link fields needed at all): along a direction that is not partitioned, or as far as the border
// of the extended field reaches along a partitioned one
bool axial_gauge_from_links_possible(const MugiqHipSpinorField &ev, const MugiqHipGaugeField &U, int kmax, int dir, int sign);
int build_axial_gauge_from_links(void *G_d, const MugiqHipSpinorField &ev, const MugiqHipGaugeField &U, int kmax, int dir, int sign, hipStream_t stream);
void set_axial_gauge_hint(const void *G_d, const void *E1_d, int dir, int sign, int kmax);
bool axial_gauge_hint_matches(const void *E0_d, int dir, int sign, int kmax);  // E0_d: the first link field of the call

// Face layers packed by a mu = x entry on its way through the eigenvectors (csrc/fused_mfma.hip, row tile): the driver hands the
// targets of the halos it is about to post to the entry that runs first, instead of launching mugiq_hip_pack_face_layers beside it
// (the pack kernels and a tile kernel that fills every CU's registers and LDS take turns, they do not overlap).  Per host thread,
// like the axial-gauge hint.  Layout of out_d: [nVec][layers][parity][12][faceCB], what mugiq_hip_pack_face_layers writes.
struct EntryPackTarget {
  void *out_d;
  int dim;     // 2 | 3
  int high;    // 0: layers x[dim] = j | 1: x[dim] = X - 1 - j
  int layers;
  int fromVec;  // eigenvectors fromVec .. nVec - 1 (the ones before went out packed by mugiq_hip_pack_face_layers)
};
int entry_pack_capacity(const MugiqHipSpinorField &ev, const int *kvals, int nK);
void set_entry_pack_hint(const EntryPackTarget *targets, int n);  // (NULL, 0) clears it
bool entry_pack_taken();
}  // namespace mugiq
#include <vector>
namespace mugiq {
bool momenta_negation_table(const int *mom, int Nmom, std::vector<int> &neg);  // reflect_mom.cpp

// ---- address spaces ---------------------------------------------------------------------------------------------
// Pointers fetched from a device-side table (eigenvector bodies) have no known address space, so hipcc emits
// flat_load: those count on vmcnt AND lgkmcnt and return out of order, which forces one `s_waitcnt vmcnt(0)
// lgkmcnt(0)` in front of the first use -- no load can overlap arithmetic inside a wave.  Casting to the global
// address space turns them into global_load (ordered, counted waits).
#define MUGIQ_GLOBAL __attribute__((address_space(1)))
template <typename T> __device__ inline const MUGIQ_GLOBAL T *as_global(const T *p) { return (const MUGIQ_GLOBAL T *)p; }
template <typename T> __device__ inline MUGIQ_GLOBAL T *as_global(T *p) { return (MUGIQ_GLOBAL T *)p; }
// Tables the host uploads before the launch and the kernel never writes (pointer / 1/sigma tables): the constant
// address space lets hipcc fetch a wave-uniform entry with s_load (lgkmcnt) instead of a vector load, which would
// share the in-order vmcnt queue with the prefetched eigenvector loads and drain it at every use.
#define MUGIQ_CONSTANT __attribute__((address_space(4)))
template <typename T> __device__ inline const MUGIQ_CONSTANT T *as_constant(const T *p) { return (const MUGIQ_CONSTANT T *)p; }

// ---- complex arithmetic in registers -------------------------------------------------------------
template <typename F> struct alignas(2 * sizeof(F)) Cplx {
  F re, im;
};
// a += conj(x) * y
template <typename F> __device__ inline void cmadd_conj(Cplx<F> &a, const Cplx<F> &x, const Cplx<F> &y) {
  a.re = fma(x.re, y.re, a.re);
  a.re = fma(x.im, y.im, a.re);
  a.im = fma(x.re, y.im, a.im);
  a.im = fma(-x.im, y.re, a.im);
}
// a += x * y
template <typename F> __device__ inline void cmadd(Cplx<F> &a, const Cplx<F> &x, const Cplx<F> &y) {
  a.re = fma(x.re, y.re, a.re);
  a.re = fma(-x.im, y.im, a.re);
  a.im = fma(x.re, y.im, a.im);
  a.im = fma(x.im, y.re, a.im);
}

// ---- DeGrand-Rossi gamma tables, include/gamma.h:32-71 of the reference ----------------------------
// G(n)_{ij} = value[n][i] * delta(j, column[n][i]); value in {+1,-1,+i,-i} encoded as a power of i:
// 0 -> +1, 1 -> +i, 2 -> -1, 3 -> -i.
constexpr int kGammaPhase[16][4] = {
    {0, 0, 0, 0}, {1, 1, 3, 3}, {2, 0, 0, 2}, {3, 1, 3, 1}, {1, 3, 3, 1}, {2, 0, 2, 0}, {3, 3, 3, 3}, {0, 0, 2, 2},
    {0, 0, 0, 0}, {1, 1, 3, 3}, {2, 0, 0, 2}, {3, 1, 3, 1}, {1, 3, 3, 1}, {2, 0, 2, 0}, {3, 3, 3, 3}, {0, 0, 2, 2}};
constexpr int kGammaColumn[16][4] = {
    {0, 1, 2, 3}, {3, 2, 1, 0}, {3, 2, 1, 0}, {0, 1, 2, 3}, {2, 3, 0, 1}, {1, 0, 3, 2}, {1, 0, 3, 2}, {2, 3, 0, 1},
    {2, 3, 0, 1}, {1, 0, 3, 2}, {1, 0, 3, 2}, {2, 3, 0, 1}, {0, 1, 2, 3}, {3, 2, 1, 0}, {3, 2, 1, 0}, {0, 1, 2, 3}};
// G -> g5*G map, include/gamma.h:99-109: index[ig] = 15 - ig, sign = -1 for ig in {3,6,9,11,12,14}
constexpr int kGammaMapSign[16] = {1, 1, 1, -1, 1, 1, -1, 1, 1, -1, 1, -1, -1, 1, -1, 1};

// t += i^phase * z
template <typename F> __device__ inline void add_phase(Cplx<F> &t, int phase, const Cplx<F> &z) {
  switch (phase) {
  case 0: t.re += z.re; t.im += z.im; break;
  case 1: t.re -= z.im; t.im += z.re; break;
  case 2: t.re -= z.re; t.im -= z.im; break;
  default: t.re += z.im; t.im -= z.re; break;
  }
}

// ---- per-site contraction arithmetic shared by the contraction, fused and prolong-contract kernels -----------
// acc (full 4x4) += conj(l[be,c]) * (s * r[al,c])
template <typename F> __device__ inline void accumulate_full(Cplx<F> acc[16], const Cplx<F> l[12], const Cplx<F> r[12], F s) {
  Cplx<F> sr[12];
#pragma unroll
  for (int k = 0; k < 12; k++) sr[k] = Cplx<F>{s * r[k].re, s * r[k].im};
#pragma unroll
  for (int be = 0; be < 4; be++)
#pragma unroll
    for (int al = 0; al < 4; al++)
#pragma unroll
      for (int c = 0; c < 3; c++) cmadd_conj(acc[be * 4 + al], l[be * 3 + c], sr[al * 3 + c]);
}

// Hermitian case (l == r): diagonal kept in diag[4] (real), strict upper triangle in up[6]
// pair order (be,al): (0,1) (0,2) (0,3) (1,2) (1,3) (2,3)
template <typename F> __device__ inline void accumulate_herm(F diag[4], Cplx<F> up[6], const Cplx<F> v[12], F s) {
  Cplx<F> sv[12];
#pragma unroll
  for (int k = 0; k < 12; k++) sv[k] = Cplx<F>{s * v[k].re, s * v[k].im};
#pragma unroll
  for (int a = 0; a < 4; a++)
#pragma unroll
    for (int c = 0; c < 3; c++) {
      diag[a] = fma(v[a * 3 + c].re, sv[a * 3 + c].re, diag[a]);
      diag[a] = fma(v[a * 3 + c].im, sv[a * 3 + c].im, diag[a]);
    }
  int p = 0;
#pragma unroll
  for (int be = 0; be < 4; be++)
#pragma unroll
    for (int al = be + 1; al < 4; al++) {
#pragma unroll
      for (int c = 0; c < 3; c++) cmadd_conj(up[p], v[be * 3 + c], sv[al * 3 + c]);
      p++;
    }
}

// trace = sum_{s2} row_value[iG][s2] * resG[s2][column_index[iG][s2]]; loopData[tid + V*iG] += trace (:110-120)
template <typename F> __device__ inline void trace_and_store(Cplx<F> *loop, const Cplx<F> acc[16], int V, int site, bool overwrite = false) {
#pragma unroll
  for (int iG = 0; iG < 16; iG++) {
    Cplx<F> t{F(0), F(0)};
#pragma unroll
    for (int s2 = 0; s2 < 4; s2++) add_phase(t, kGammaPhase[iG][s2], acc[s2 * 4 + kGammaColumn[iG][s2]]);
    Cplx<F> *out = loop + (int64_t)V * iG + site;
    Cplx<F> o = overwrite ? Cplx<F>{F(0), F(0)} : *out;  // overwrite: the slot holds nothing yet (no memset, no read)
    o.re += t.re;
    o.im += t.im;
    *out = o;
  }
}

// the same for the gamma channels [G0, G1) only (the tile kernel splits the 16 channels over the two lane halves)
template <typename F, int G0, int G1> __device__ inline void trace_and_store_range(Cplx<F> *loop, const Cplx<F> acc[16], int V, int site, bool overwrite = false) {
#pragma unroll
  for (int iG = G0; iG < G1; iG++) {
    Cplx<F> t{F(0), F(0)};
#pragma unroll
    for (int s2 = 0; s2 < 4; s2++) add_phase(t, kGammaPhase[iG][s2], acc[s2 * 4 + kGammaColumn[iG][s2]]);
    Cplx<F> *out = loop + (int64_t)V * iG + site;
    Cplx<F> o = overwrite ? Cplx<F>{F(0), F(0)} : *out;  // overwrite: the slot holds nothing yet (no memset, no read)
    o.re += t.re;
    o.im += t.im;
    *out = o;
  }
}

// the four gamma traces G0 .. G0 + 3 of a colour-traced spin matrix, returned instead of stored
template <typename F, int G0> __device__ inline void traces_range(Cplx<F> out[4], const Cplx<F> acc[16]) {
#pragma unroll
  for (int i = 0; i < 4; i++) {
    Cplx<F> t{F(0), F(0)};
#pragma unroll
    for (int s2 = 0; s2 < 4; s2++) add_phase(t, kGammaPhase[G0 + i][s2], acc[s2 * 4 + kGammaColumn[G0 + i][s2]]);
    out[i] = t;
  }
}

// ---- QUDA even-odd index helpers (upstream QUDA index_helper.cuh) --------------
__host__ __device__ inline void get_coords(int c[4], int x_cb, const int X[4], int parity) {
  const int za = x_cb / (X[0] >> 1);
  const int zb = za / X[1];
  c[1] = za - zb * X[1];
  c[3] = zb / X[2];
  c[2] = zb - c[3] * X[2];
  const int x1odd = (c[1] + c[2] + c[3] + parity) & 1;
  c[0] = 2 * x_cb + x1odd - za * X[0];
}
__host__ __device__ inline int lex_index(const int c[4], const int X[4]) {
  return ((c[3] * X[2] + c[2]) * X[1] + c[1]) * X[0] + c[0];
}
// linkIndexShift(x, dx, X) with y[i] = (x[i] + dx[i] + X[i]) % X[i]
__host__ __device__ inline int link_index_shift(const int c[4], const int dx[4], const int X[4]) {
  int y[4];
#pragma unroll
  for (int i = 0; i < 4; i++) y[i] = (c[i] + dx[i] + X[i]) % X[i];
  return lex_index(y, X) >> 1;
}
// ghostFaceIndex<bnd>(x, X, dim, nFace = 1) on the face: the leading term vanishes
__host__ __device__ inline int ghost_face_index_on_face(const int c[4], const int X[4], int dim) {
  int idx;
  switch (dim) {
  case 0: idx = (c[3] * X[2] + c[2]) * X[1] + c[1]; break;
  case 1: idx = (c[3] * X[2] + c[2]) * X[0] + c[0]; break;
  case 2: idx = (c[3] * X[1] + c[1]) * X[0] + c[0]; break;
  default: idx = (c[2] * X[1] + c[1]) * X[0] + c[0]; break;
  }
  return idx >> 1;
}

// ---- native field accessors -------------------------------------------------------------------------
// Spinor body (or ghost zone) in FLOAT2 / FLOAT4 order; see MugiqHipSpinorField in mugiq_hip.h.
template <typename F, int ORDER> struct SpinorView {
  F *base;
  int stride;
  int64_t parity_offset;  // complex elements

  __device__ inline void load(Cplx<F> v[12], int parity, int x_cb) const {
    const Cplx<F> *p = reinterpret_cast<const Cplx<F> *>(base) + parity * parity_offset;
    if constexpr (ORDER == 2) {
#pragma unroll
      for (int k = 0; k < 12; k++) v[k] = p[(int64_t)k * stride + x_cb];
    } else {
      struct alignas(4 * sizeof(F)) Pair {
        Cplx<F> a, b;
      };
      const Pair *q = reinterpret_cast<const Pair *>(p);
#pragma unroll
      for (int j = 0; j < 6; j++) {
        Pair t = q[(int64_t)j * stride + x_cb];
        v[2 * j] = t.a;
        v[2 * j + 1] = t.b;
      }
    }
  }
  __device__ inline void store(const Cplx<F> v[12], int parity, int x_cb) const {
    Cplx<F> *p = reinterpret_cast<Cplx<F> *>(base) + parity * parity_offset;
    if constexpr (ORDER == 2) {
#pragma unroll
      for (int k = 0; k < 12; k++) p[(int64_t)k * stride + x_cb] = v[k];
    } else {
      struct alignas(4 * sizeof(F)) Pair {
        Cplx<F> a, b;
      };
      Pair *q = reinterpret_cast<Pair *>(p);
#pragma unroll
      for (int j = 0; j < 6; j++) q[(int64_t)j * stride + x_cb] = Pair{v[2 * j], v[2 * j + 1]};
    }
  }
};

template <typename F, int ORDER> inline SpinorView<F, ORDER> make_view(const MugiqHipSpinorField &f) {
  return SpinorView<F, ORDER>{static_cast<F *>(f.data), f.stride, f.parity_offset};
}
template <typename F, int ORDER> inline SpinorView<F, ORDER> make_ghost_view(void *zone, int faceCB) {
  return SpinorView<F, ORDER>{static_cast<F *>(zone), faceCB, (int64_t)12 * faceCB};
}

// Gauge field in native FLOAT2 order, 18 reals per link (reconstruct NO).
template <typename F> struct GaugeView {
  const F *base;
  int stride;
  int64_t parity_offset;
  __device__ inline void load(Cplx<F> u[9], int dir, int parity, int x_cb) const {
    const Cplx<F> *p = reinterpret_cast<const Cplx<F> *>(base) + parity * parity_offset + (int64_t)dir * 9 * stride + x_cb;
#pragma unroll
    for (int i = 0; i < 9; i++) u[i] = p[(int64_t)i * stride];
  }
};

int validate_spinor(const MugiqHipSpinorField *f, const char *who, const char *name);
// comm_dim_partitioned(d) (include/contract_util.cuh:89): more than one rank along d, or the partitioned code path forced
// on an axis of extent 1 (the rank is then its own neighbour; QUDA's comm_dim_partitioned_set)
inline bool comm_partitioned(const MugiqHipComm *c, int d) { return c != nullptr && (c->grid[d] > 1 || c->partitioned[d] != 0); }
bool same_geometry(const MugiqHipSpinorField &a, const MugiqHipSpinorField &b);

}  // namespace mugiq
