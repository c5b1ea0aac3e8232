// a4/a5: SU(3) covariant displacement of a colour-spinor field, and the face packer for its halo.
//
// Reference: covariantDisplacementVector_kernel (lib/mugiq_displace_kernels.cu:156-185) with helpers
// getNbrSiteVec (:116-151) and getNbrLinkExtG/getNbrLinkDispExtG (:39-74); host wrapper
// performCovariantDisplacementVector (lib/contract_wrappers.cu:171-198) which first runs
// exchangeGhostVec (:166-169).  The reference launches (16,2,1) blocks (32 threads = half a CDNA wave);
// here one site per lane over the flattened even-odd index, 256-lane blocks, every plane access a
// coalesced 1 KiB request, index arithmetic identical (it must be bit-exact).
#include "internal.h"

namespace mugiq {

constexpr int kDispBlock = 256;

struct DispGeom {
  int X[4];    // local dims                      (arg->dim)
  int XE[4];   // extended gauge dims X + 2*brd   (arg->dimEx)
  int brd[4];  // border                          (arg->brd)
  int volumeCB;
  int partitioned;  // commDim[dir]
};

template <typename F, int ORDER, int DIR, int SIGN>
__global__ __launch_bounds__(kDispBlock) void cov_displace_kernel(SpinorView<F, ORDER> dst, SpinorView<F, ORDER> src,
                                                                  SpinorView<F, ORDER> ghost, GaugeView<F> U, DispGeom g) {
  const int site = blockIdx.x * kDispBlock + threadIdx.x;
  if (site >= 2 * g.volumeCB) return;
  const int pty = site >= g.volumeCB ? 1 : 0;
  const int x_cb = site - pty * g.volumeCB;

  int coord[4];
  get_coords(coord, x_cb, g.X, pty);  // :167-169
  constexpr int dir = DIR;  // compile-time so coord[dir] stays in registers
  const int nbrPty = 1 - pty;  // :120

  // ---- neighbouring vector V(x+d) or V(x-d): getNbrSiteVec :116-151 (nFace = 1)
  Cplx<F> v[12];
  int dx[4] = {0, 0, 0, 0};
  if (SIGN == MUGIQ_HIP_DISP_SIGN_PLUS) {
    if (g.partitioned && (coord[dir] + 1 >= g.X[dir])) {
      ghost.load(v, nbrPty, ghost_face_index_on_face(coord, g.X, dir));  // F.Ghost(dir, 1, nbrPty, idx, s, c)
    } else {
      dx[dir] = 1;
      src.load(v, nbrPty, link_index_shift(coord, dx, g.X));  // linkIndexP1
    }
  } else {
    if (g.partitioned && (coord[dir] - 1 < 0)) {
      ghost.load(v, nbrPty, ghost_face_index_on_face(coord, g.X, dir));  // F.Ghost(dir, 0, nbrPty, idx, s, c)
    } else {
      dx[dir] = -1;
      src.load(v, nbrPty, link_index_shift(coord, dx, g.X));  // linkIndexM1
    }
  }

  // ---- neighbouring link U_d(x) or U_d^dag(x-d) from the extended field: getNbrLinkDispExtG :39-66
  int dx1[4] = {0, 0, 0, 0};
  int c2[4];
#pragma unroll
  for (int i = 0; i < 4; i++) c2[i] = coord[i] + g.brd[i];
  if (SIGN == MUGIQ_HIP_DISP_SIGN_MINUS) dx1[dir] -= 1;
  const int linkPty = (SIGN == MUGIQ_HIP_DISP_SIGN_MINUS) ? 1 - pty : pty;  // evenORodd(dx1) == 0 ? pty : 1-pty
  Cplx<F> u[9];
  U.load(u, dir, linkPty, link_index_shift(c2, dx1, g.XE));
  if (SIGN == MUGIQ_HIP_DISP_SIGN_MINUS) {  // conj(Matrix) = Hermitian conjugate
    Cplx<F> w[9];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
      for (int j = 0; j < 3; j++) w[i * 3 + j] = Cplx<F>{u[j * 3 + i].re, -u[j * 3 + i].im};
#pragma unroll
    for (int i = 0; i < 9; i++) u[i] = w[i];
  }

  // ---- R = nbrU * nbrV: y(s,i) = sum_j A(i,j) x(s,j)   :182
  Cplx<F> r[12];
#pragma unroll
  for (int s = 0; s < 4; s++)
#pragma unroll
    for (int i = 0; i < 3; i++) {
      Cplx<F> a{F(0), F(0)};
#pragma unroll
      for (int j = 0; j < 3; j++) cmadd(a, u[i * 3 + j], v[s * 3 + j]);
      r[s * 3 + i] = a;
    }
  dst.store(r, pty, x_cb);  // FillFermionSite :184
}

struct FaceGeom {
  int X[4];
  int dim;
  int fixed;   // coordinate of the face along dim
  int faceCB;
};

// One lane per (parity, ghostFaceIndex) pair of the face.
template <typename F, int ORDER>
__global__ __launch_bounds__(kDispBlock) void pack_face_kernel(SpinorView<F, ORDER> face, SpinorView<F, ORDER> src, FaceGeom g) {
  const int i = blockIdx.x * kDispBlock + threadIdx.x;
  if (i >= 2 * g.faceCB) return;
  const int pty = i >= g.faceCB ? 1 : 0;
  const int idx = i - pty * g.faceCB;
  // invert ghostFaceIndex: the three remaining coordinates are lexicographic, lowest dim fastest, >> 1
  const int r0 = g.dim == 0 ? 1 : 0, r1 = g.dim <= 1 ? 2 : 1;
  auto Xd = [&](int d) { return d == 0 ? g.X[0] : d == 1 ? g.X[1] : d == 2 ? g.X[2] : g.X[3]; };
  int l = 2 * idx;
  const int c0 = l % Xd(r0);
  l /= Xd(r0);
  const int c1 = l % Xd(r1);
  const int c2 = l / Xd(r1);
  const int bit = (pty - (c0 + c1 + c2 + g.fixed)) & 1;  // pick the site of this parity in the pair
  int c[4];
#pragma unroll
  for (int d = 0; d < 4; d++) c[d] = d == g.dim ? g.fixed : d == r0 ? c0 + bit : d == r1 ? c1 : c2;
  Cplx<F> v[12];
  src.load(v, pty, lex_index(c, g.X) >> 1);
  face.store(v, pty, idx);
}

template <typename F, int ORDER>
static int launch_displace(const MugiqHipSpinorField *dst, const MugiqHipSpinorField *src, const MugiqHipGaugeField *U,
                           int dir, int sign, int partitioned, hipStream_t stream) {
  DispGeom g;
  for (int d = 0; d < 4; d++) {
    g.X[d] = src->X[d];
    g.brd[d] = U->R[d];
    g.XE[d] = src->X[d] + 2 * U->R[d];
  }
  g.volumeCB = src->volumeCB;
  g.partitioned = partitioned;
  const int faceCB = src->volumeCB / src->X[dir];
  void *zone = partitioned ? src->ghost[dir][sign == MUGIQ_HIP_DISP_SIGN_PLUS ? 1 : 0] : src->data;
  GaugeView<F> gv{static_cast<const F *>(U->data), U->stride, U->parity_offset};
  const int V = 2 * src->volumeCB;
  const dim3 grid((V + kDispBlock - 1) / kDispBlock), block(kDispBlock);
  auto d = make_view<F, ORDER>(*dst);
  auto s = make_view<F, ORDER>(*src);
  auto z = make_ghost_view<F, ORDER>(zone, faceCB);
#define MUGIQ_DISP_CASE(D, S)                                                                            \
  case (D)*2 + (S):                                                                                      \
    hipLaunchKernelGGL((cov_displace_kernel<F, ORDER, D, S>), grid, block, 0, stream, d, s, z, gv, g);   \
    break;
  switch (dir * 2 + sign) {
    MUGIQ_DISP_CASE(0, 0) MUGIQ_DISP_CASE(0, 1) MUGIQ_DISP_CASE(1, 0) MUGIQ_DISP_CASE(1, 1)
    MUGIQ_DISP_CASE(2, 0) MUGIQ_DISP_CASE(2, 1) MUGIQ_DISP_CASE(3, 0) MUGIQ_DISP_CASE(3, 1)
  }
#undef MUGIQ_DISP_CASE
  MUGIQ_CHECK_HIP(hipGetLastError());
  return MUGIQ_HIP_SUCCESS;
}

template <typename F, int ORDER>
static int launch_pack(void *face_d, const MugiqHipSpinorField *src, int dim, int high, hipStream_t stream) {
  FaceGeom g;
  for (int d = 0; d < 4; d++) g.X[d] = src->X[d];
  g.dim = dim;
  g.fixed = high ? src->X[dim] - 1 : 0;
  g.faceCB = src->volumeCB / src->X[dim];
  const int n = 2 * g.faceCB;
  hipLaunchKernelGGL((pack_face_kernel<F, ORDER>), dim3((n + kDispBlock - 1) / kDispBlock), dim3(kDispBlock), 0, stream,
                     make_ghost_view<F, ORDER>(face_d, g.faceCB), make_view<F, ORDER>(*src), g);
  MUGIQ_CHECK_HIP(hipGetLastError());
  return MUGIQ_HIP_SUCCESS;
}

// E_0 of the fused plan: the FLOAT2 field whose first three spin rows are the 3x3 identity, E_0(x)(s,c) = delta_sc for
// s < 3, and zero elsewhere (its k-fold displacement is the path-ordered link product W_k)
template <typename F> __global__ __launch_bounds__(256) void identity_links_kernel(Cplx<F> *data, int volumeCB, int64_t parity_offset) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * volumeCB) return;
  const int pty = i >= volumeCB ? 1 : 0;
  Cplx<F> *p = data + (int64_t)pty * parity_offset + (i - pty * volumeCB);
#pragma unroll
  for (int k = 0; k < 12; k++) p[(int64_t)k * volumeCB] = Cplx<F>{(k == 0 || k == 4 || k == 8) ? F(1) : F(0), F(0)};
}

int fill_identity_links(const MugiqHipSpinorField *f, hipStream_t stream) {
  MUGIQ_REQUIRE(f && f->data && f->field_order == 2 && f->stride == f->volumeCB, "fill_identity_links: need a pad-0 FLOAT2 field");
  const dim3 grid((2 * f->volumeCB + 255) / 256);
  if (f->precision == 8)
    hipLaunchKernelGGL(identity_links_kernel<double>, grid, dim3(256), 0, stream, static_cast<Cplx<double> *>(f->data), f->volumeCB, f->parity_offset);
  else
    hipLaunchKernelGGL(identity_links_kernel<float>, grid, dim3(256), 0, stream, static_cast<Cplx<float> *>(f->data), f->volumeCB, f->parity_offset);
  MUGIQ_CHECK_HIP(hipGetLastError());
  return MUGIQ_HIP_SUCCESS;
}

int validate_gauge(const MugiqHipGaugeField *U, const MugiqHipSpinorField *ref, const char *who) {
  MUGIQ_REQUIRE(U != nullptr && U->data != nullptr, "%s: gauge field is NULL", who);
  MUGIQ_REQUIRE(U->precision == ref->precision,
                "%s: Incompatible precision settings between the spinor (%d) and gauge field (%d)", who, ref->precision,
                U->precision);  // lib/displace.cpp:84-87
  long long volEx = 1;
  int sumR = 0;
  for (int d = 0; d < 4; d++) {
    MUGIQ_REQUIRE(U->X[d] == ref->X[d], "%s: gauge X[%d] = %d differs from the spinor's %d", who, d, U->X[d], ref->X[d]);
    MUGIQ_REQUIRE(U->R[d] >= 0, "%s: gauge R[%d] = %d is negative", who, d, U->R[d]);
    volEx *= U->X[d] + 2 * U->R[d];
    sumR += U->R[d];
  }
  // the reference takes the link parity from the interior coordinates (lib/mugiq_displace_kernels.cu:55),
  // which addresses the extended even-odd field correctly only if the border shift preserves parity
  MUGIQ_REQUIRE((sumR & 1) == 0, "%s: the sum of the gauge borders R must be even", who);
  MUGIQ_REQUIRE(U->stride >= volEx / 2, "%s: gauge stride %d < extended volumeCB %lld", who, U->stride, volEx / 2);
  MUGIQ_REQUIRE(U->parity_offset >= (int64_t)36 * U->stride, "%s: gauge parity_offset %lld < 36*stride", who,
                (long long)U->parity_offset);
  return MUGIQ_HIP_SUCCESS;
}

}  // namespace mugiq

using namespace mugiq;

extern "C" {

int mugiq_hip_perform_covariant_displacement_vector(const MugiqHipSpinorField *dst, const MugiqHipSpinorField *src,
                                                    const MugiqHipGaugeField *gauge, int dispDir, int dispSign,
                                                    const int commDim[4], void *stream) {
  if (int dbg_ = mugiq::debug_poison_lds_if_asked(static_cast<hipStream_t>(stream))) return dbg_;
  const char *who = "performCovariantDisplacementVector";
  int st = validate_spinor(dst, who, "dst");
  if (st) return st;
  st = validate_spinor(src, who, "src");
  if (st) return st;
  MUGIQ_REQUIRE(same_geometry(*dst, *src), "%s: dst and src differ in precision, field order or geometry", who);
  MUGIQ_REQUIRE(dst->data != src->data, "%s: dst and src must not alias (the kernel reads neighbours of src)", who);
  // lib/displace.cpp:217-222
  MUGIQ_REQUIRE(dispDir >= 0 && dispDir < 4 && (dispSign == 0 || dispSign == 1), "%s: Got invalid dispDir and/or dispSign.",
                who);
  st = validate_gauge(gauge, src, who);
  if (st) return st;
  const int part = commDim ? (commDim[dispDir] != 0) : 0;
  if (part)
    MUGIQ_REQUIRE(src->ghost[dispDir][dispSign == MUGIQ_HIP_DISP_SIGN_PLUS ? 1 : 0] != nullptr,
                  "%s: dim %d is partitioned but src->ghost[%d][%d] is NULL (halo exchange missing)", who, dispDir, dispDir,
                  dispSign == MUGIQ_HIP_DISP_SIGN_PLUS ? 1 : 0);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (src->precision == 8 && src->field_order == 2) return launch_displace<double, 2>(dst, src, gauge, dispDir, dispSign, part, s);
  if (src->precision == 8 && src->field_order == 4) return launch_displace<double, 4>(dst, src, gauge, dispDir, dispSign, part, s);
  if (src->precision == 4 && src->field_order == 2) return launch_displace<float, 2>(dst, src, gauge, dispDir, dispSign, part, s);
  return launch_displace<float, 4>(dst, src, gauge, dispDir, dispSign, part, s);
}

int mugiq_hip_pack_face(void *face_d, const MugiqHipSpinorField *src, int dim, int high, void *stream) {
  const char *who = "mugiq_hip_pack_face";
  int st = validate_spinor(src, who, "src");
  if (st) return st;
  MUGIQ_REQUIRE(face_d != nullptr, "%s: face_d is NULL", who);
  MUGIQ_REQUIRE(dim >= 0 && dim < 4 && (high == 0 || high == 1), "%s: invalid dim %d / high %d", who, dim, high);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (src->precision == 8 && src->field_order == 2) return launch_pack<double, 2>(face_d, src, dim, high, s);
  if (src->precision == 8 && src->field_order == 4) return launch_pack<double, 4>(face_d, src, dim, high, s);
  if (src->precision == 4 && src->field_order == 2) return launch_pack<float, 2>(face_d, src, dim, high, s);
  return launch_pack<float, 4>(face_d, src, dim, high, s);
}

}  // extern "C"
