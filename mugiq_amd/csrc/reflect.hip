// Reflected displacement entries.
//
// The reference computes the "+mu" and the "-mu" entry of a displacement independently (one displace + contract
// sequence per eigenvector, step and sign: lib/loop_mugiq.cpp:478-500).  But the backward path-ordered link product
// is the dagger of the forward one started k sites earlier, W_{-k}(x) = W_{+k}(x - k mu)^dagger, the gamma matrices act
// on spin and W on colour, and sigma_n is real, so slot by slot
//
//     L^-_{k,G}(x) = sum_n v_n^dag(x) G W_{-k}(x) v_n(x - k mu) / sigma_n = eta_G * conj( L^+_{k,G}(x - k mu) ),
//     eta_G = +1 / -1 for G^dagger = +G / -G   (G(n) = g1^n0 g2^n1 g3^n2 g4^n3: eta = (-1)^(m(m-1)/2), m factors)
//
// and symmetrically L^+_k(x) = eta * conj(L^-_k(x + k mu)).  Whenever the entry list holds both signs of a direction
// with the same lengths (BASELINE.json configs[2] does: "+x:1,3;-x:1,3;..."), the second one costs one read and one
// write of 16 complex per site and slot instead of a pass over all eigenvectors.  On a partitioned direction the k
// boundary layers of the SOURCE SLOT (16 complex per face site) come from the neighbour -- N_ev times less than an
// eigenvector halo.  (tests/test_oracle_kat.py checks the identity on the CPU restatement of the reference.)
#include "internal.h"

namespace mugiq {

template <typename A> struct ReflectArgs {
  Cplx<A> *dst;          // [16][V] slot being derived
  const Cplx<A> *src;    // [16][V] computed slot of the opposite sign, same length
  const Cplx<A> *ghost;  // [k layers][16][2 parities][faceCB] of the neighbour's src slot (partitioned), else NULL
  int X[4];
  int volumeCB;
  int dir;
  int shift;             // source site = x + shift * mu  (-k: dst is the "-" entry, +k: dst is the "+" entry)
  int k;
  int partitioned;
  int faceCB;
};

__device__ inline int gamma_dagger_sign(int n) {
  const int m = __popc(n);
  return ((m * (m - 1) / 2) & 1) ? -1 : 1;
}

template <typename A> __global__ __launch_bounds__(256) void reflect_kernel(ReflectArgs<A> a) {
  const int V = 2 * a.volumeCB;
  const int tid = blockIdx.x * 256 + threadIdx.x;
  if (tid >= V) return;
  const int pty = tid >= a.volumeCB ? 1 : 0;
  int c[4];
  get_coords(c, tid - pty * a.volumeCB, a.X, pty);
  const int J = a.X[a.dir];
  int cs = c[a.dir] + a.shift;
  const int spty = pty ^ (a.k & 1);
  const Cplx<A> *base;
  int64_t stride;
  if (a.partitioned && (cs < 0 || cs >= J)) {
    // shift < 0: the backward neighbour's layer x = J + cs, packed as layer j = cs + k of its top k layers;
    // shift > 0: the forward neighbour's layer x = cs - J = j of its bottom k layers
    const int j = cs < 0 ? cs + a.k : cs - J;
    base = a.ghost + ((int64_t)j * 32 + spty) * a.faceCB + ghost_face_index_on_face(c, a.X, a.dir);
    stride = 2 * (int64_t)a.faceCB;
  } else {
    cs %= J;
    if (cs < 0) cs += J;
    c[a.dir] = cs;
    base = a.src + (int64_t)spty * a.volumeCB + (lex_index(c, a.X) >> 1);
    stride = V;
  }
#pragma unroll
  for (int ig = 0; ig < 16; ig++) {
    const Cplx<A> v = base[stride * ig];
    const A eta = (A)gamma_dagger_sign(ig);
    a.dst[(int64_t)V * ig + tid] = Cplx<A>{eta * v.re, -eta * v.im};
  }
}

// k boundary layers of a slot: out[((j*16 + ig)*2 + parity)*faceCB + face index], x[dim] = X-k+j (high) | j (low)
template <typename A> struct PackLoopArgs {
  Cplx<A> *out;
  const Cplx<A> *slot;
  int X[4];
  int volumeCB, dim, high, layers, faceCB;
};

template <typename A> __global__ __launch_bounds__(256) void pack_loop_layers_kernel(PackLoopArgs<A> g) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * g.faceCB) return;
  const int j = blockIdx.y;
  const int pty = i >= g.faceCB ? 1 : 0;
  const int idx = i - pty * g.faceCB;
  // invert ghost_face_index_on_face: the three other coordinates in ascending dimension order, parity fixes the low bit
  const int r0 = g.dim == 0 ? 1 : 0, r1 = g.dim <= 1 ? 2 : 1;
  auto Xd = [&](int d) { return d == 0 ? g.X[0] : d == 1 ? g.X[1] : d == 2 ? g.X[2] : g.X[3]; };
  const int fixed = g.high ? Xd(g.dim) - g.layers + j : j;
  int l = 2 * idx;
  const int c0 = l % Xd(r0);
  l /= Xd(r0);
  const int c1 = l % Xd(r1);
  const int c2 = l / Xd(r1);
  const int bit = (pty - (c0 + c1 + c2 + fixed)) & 1;
  int c[4];
#pragma unroll
  for (int d = 0; d < 4; d++) c[d] = d == g.dim ? fixed : d == r0 ? c0 + bit : d == r1 ? c1 : c2;
  const int64_t V = 2 * (int64_t)g.volumeCB;
  const Cplx<A> *src = g.slot + (int64_t)pty * g.volumeCB + (lex_index(c, g.X) >> 1);
#pragma unroll
  for (int ig = 0; ig < 16; ig++) g.out[(((int64_t)j * 16 + ig) * 2 + pty) * g.faceCB + idx] = src[V * ig];
}

template <typename A>
static int launch_reflect(void *dst, const void *src, const void *ghost, const int X[4], int dir, int dstSign, int k, int partitioned,
                          hipStream_t stream) {
  ReflectArgs<A> a;
  a.dst = static_cast<Cplx<A> *>(dst);
  a.src = static_cast<const Cplx<A> *>(src);
  a.ghost = static_cast<const Cplx<A> *>(ghost);
  long long vol = 1;
  for (int d = 0; d < 4; d++) {
    a.X[d] = X[d];
    vol *= X[d];
  }
  a.volumeCB = (int)(vol / 2);
  a.dir = dir;
  a.k = k;
  a.shift = dstSign == MUGIQ_HIP_DISP_SIGN_MINUS ? -k : k;
  a.partitioned = partitioned;
  a.faceCB = a.volumeCB / X[dir];
  hipLaunchKernelGGL(reflect_kernel<A>, dim3((unsigned)((vol + 255) / 256)), dim3(256), 0, stream, a);
  MUGIQ_CHECK_HIP(hipGetLastError());
  return MUGIQ_HIP_SUCCESS;
}

}  // namespace mugiq

using namespace mugiq;

extern "C" {

int mugiq_hip_reflect_displaced_loop(void *dstSlot_d, const void *srcSlot_d, const void *ghostLayers_d, const int localL[4],
                                     int dispDir, int dstDispSign, int length, const int commDim[4], int precision, void *stream) {
  if (int dbg_ = mugiq::debug_poison_lds_if_asked(static_cast<hipStream_t>(stream))) return dbg_;
  const char *who = "mugiq_hip_reflect_displaced_loop";
  MUGIQ_REQUIRE(dstSlot_d && srcSlot_d && localL && dstSlot_d != srcSlot_d, "%s: NULL / aliased argument", who);
  MUGIQ_REQUIRE(precision == 4 || precision == 8, "%s: Precision not supported! (%d)", who, precision);
  MUGIQ_REQUIRE(dispDir >= 0 && dispDir < 4, "%s: dispDir = %d", who, dispDir);
  MUGIQ_REQUIRE(dstDispSign == MUGIQ_HIP_DISP_SIGN_MINUS || dstDispSign == MUGIQ_HIP_DISP_SIGN_PLUS, "%s: dstDispSign = %d", who, dstDispSign);
  MUGIQ_REQUIRE(length >= 1, "%s: length = %d", who, length);
  long long vol = 1;
  for (int d = 0; d < 4; d++) {
    MUGIQ_REQUIRE(localL[d] > 0 && (localL[d] & 1) == 0, "%s: localL[%d] = %d must be positive and even", who, d, localL[d]);
    vol *= localL[d];
  }
  MUGIQ_REQUIRE(vol < (1LL << 31), "%s: local volume overflows int", who);
  const int part = commDim ? (commDim[dispDir] != 0) : 0;
  if (part) {
    MUGIQ_REQUIRE(ghostLayers_d != nullptr, "%s: direction %d is partitioned but no ghost layers were given", who, dispDir);
    MUGIQ_REQUIRE(length <= localL[dispDir], "%s: length %d exceeds the local extent %d of a partitioned direction", who, length, localL[dispDir]);
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (precision == 8) return launch_reflect<double>(dstSlot_d, srcSlot_d, ghostLayers_d, localL, dispDir, dstDispSign, length, part, s);
  return launch_reflect<float>(dstSlot_d, srcSlot_d, ghostLayers_d, localL, dispDir, dstDispSign, length, part, s);
}

int mugiq_hip_pack_loop_layers(void *layers_d, const void *slot_d, const int localL[4], int dim, int high, int layers, int precision,
                               void *stream) {
  const char *who = "mugiq_hip_pack_loop_layers";
  MUGIQ_REQUIRE(layers_d && slot_d && localL, "%s: NULL argument", who);
  MUGIQ_REQUIRE(precision == 4 || precision == 8, "%s: Precision not supported! (%d)", who, precision);
  MUGIQ_REQUIRE(dim >= 0 && dim < 4 && (high == 0 || high == 1), "%s: dim = %d, high = %d", who, dim, high);
  long long vol = 1;
  for (int d = 0; d < 4; d++) {
    MUGIQ_REQUIRE(localL[d] > 0 && (localL[d] & 1) == 0, "%s: localL[%d] = %d must be positive and even", who, d, localL[d]);
    vol *= localL[d];
  }
  MUGIQ_REQUIRE(layers >= 1 && layers <= localL[dim] && layers <= 65535, "%s: layers = %d must be in [1, X[dim] = %d]", who, layers, localL[dim]);
  const int volumeCB = (int)(vol / 2), faceCB = volumeCB / localL[dim];
  hipStream_t s = static_cast<hipStream_t>(stream);
  const dim3 grid((2 * faceCB + 255) / 256, layers);
  if (precision == 8) {
    PackLoopArgs<double> g{static_cast<Cplx<double> *>(layers_d), static_cast<const Cplx<double> *>(slot_d), {localL[0], localL[1], localL[2], localL[3]}, volumeCB, dim, high, layers, faceCB};
    hipLaunchKernelGGL(pack_loop_layers_kernel<double>, grid, dim3(256), 0, s, g);
  } else {
    PackLoopArgs<float> g{static_cast<Cplx<float> *>(layers_d), static_cast<const Cplx<float> *>(slot_d), {localL[0], localL[1], localL[2], localL[3]}, volumeCB, dim, high, layers, faceCB};
    hipLaunchKernelGGL(pack_loop_layers_kernel<float>, grid, dim3(256), 0, s, g);
  }
  MUGIQ_CHECK_HIP(hipGetLastError());
  return MUGIQ_HIP_SUCCESS;
}

}  // extern "C"
