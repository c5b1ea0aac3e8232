// a8: Fourier phase matrix; a9: even-odd -> time-major reorder with the G -> g5*G map.
//
// Reference: phaseMatrix_kernel (lib/mugiq_util_kernels.cu:3-35, wrapper lib/contract_wrappers.cu:50-77) and
// convertIdxOrder_mapGamma_kernel (lib/mugiq_util_kernels.cu:59-99, wrapper lib/contract_wrappers.cu:133-156).
#include "internal.h"

#include <vector>

namespace mugiq {

constexpr int kUtilBlock = 256;

struct PhaseGeom {
  long long locV3;
  int Nmom;
  int FTSign;
  int localL[3];
  int totalL[3];
  int commCoord[3];
};

// One lane per local spatial site, looping over the momenta; stores are coalesced over v3.
template <typename F>
__global__ __launch_bounds__(kUtilBlock) void phase_matrix_kernel(Cplx<F> *phaseMatrix, const int *momMatrix, PhaseGeom g) {
  const long long tid = (long long)blockIdx.x * kUtilBlock + threadIdx.x;
  if (tid >= g.locV3) return;
  const int a1 = (int)(tid / g.localL[0]);
  const int a2 = a1 / g.localL[1];
  int gcoord[3];
  gcoord[0] = (int)(tid - (long long)a1 * g.localL[0]) + g.commCoord[0] * g.localL[0];
  gcoord[1] = (a1 - a2 * g.localL[1]) + g.commCoord[1] * g.localL[1];
  gcoord[2] = a2 + g.commCoord[2] * g.localL[2];
  const F sgn = (F)g.FTSign;
  // reference: 2.0*PI*phase with PI = 2.0*asin(1.0) (include/util_mugiq.h:7) = (4*asin(1))*phase in double
  const double twoPi = 4.0 * 1.5707963267948966;
  for (int im = 0; im < g.Nmom; im++) {
    F phase = 0.0;
#pragma unroll
    for (int id = 0; id < 3; id++) phase += momMatrix[id + 3 * im] * gcoord[id] / (F)g.totalL[id];  // :25-26
    double s, c;
    sincos(twoPi * phase, &s, &c);
    Cplx<F> ph;
    ph.re = (F)c;          // :28
    ph.im = (F)(sgn * s);  // :29
    phaseMatrix[tid + g.locV3 * im] = ph;
  }
}

struct ConvertGeom {
  int X[4];
  int volumeCB;
  int nData;
  long long locV3;
};

// The reference scatters 16-byte elements with a stride of Lt*nData elements (one lane per input site).
// Here a workgroup transposes a [TV spatial sites] x [Lt] tile through LDS: global reads run along the
// even-odd site index, global writes along t, both in >= 256-byte runs.
constexpr int kTileV = 64;

template <typename F>
__global__ __launch_bounds__(kUtilBlock) void convert_idx_map_gamma_kernel(Cplx<F> *out, const Cplx<F> *in, ConvertGeom g) {
  extern __shared__ __align__(16) unsigned char smem[];
  Cplx<F> *tile = reinterpret_cast<Cplx<F> *>(smem);  // [kTileV][Lt + 1]
  const int Lt = g.X[3];
  const int ld = Lt + 1;
  const long long v0 = (long long)blockIdx.x * kTileV;
  const int idataFrom = blockIdx.y;                 // ig + 16*iL
  const int ig = idataFrom & 15;
  const int idataTo = (15 - ig) + (idataFrom - ig);  // gammaMap->index[ig] + N_GAMMA_*iL   :89
  const F sign = (F)kGammaMapSign[ig];               // gammaMap->sign[ig]                    :93
  const long long V = 2LL * g.volumeCB;
  const Cplx<F> *src = in + V * idataFrom;

  for (int idx = threadIdx.x; idx < kTileV * Lt; idx += kUtilBlock) {
    const int t = idx / kTileV;
    const int j = idx - t * kTileV;
    const long long v3 = v0 + j;
    if (v3 < g.locV3) {
      const int x = (int)(v3 % g.X[0]);
      const int yz = (int)(v3 / g.X[0]);
      const int y = yz % g.X[1];
      const int z = yz / g.X[1];
      const int pty = (x + y + z + t) & 1;
      const long long lex = v3 + g.locV3 * t;        // x + Lx*(y + Ly*(z + Lz*t))
      Cplx<F> val = src[(lex >> 1) + (long long)pty * g.volumeCB];   // tid = x_cb + volumeCB*pty   :70
      tile[j * ld + t] = Cplx<F>{sign * val.re, sign * val.im};
    }
  }
  __syncthreads();
  for (int idx = threadIdx.x; idx < kTileV * Lt; idx += kUtilBlock) {
    const int j = idx / Lt;
    const int t = idx - j * Lt;
    const long long v3 = v0 + j;
    if (v3 < g.locV3) out[t + (long long)Lt * idataTo + (long long)Lt * g.nData * v3] = tile[j * ld + t];  // :91
  }
}

// ---- roofline calibration: a pure streaming read of `n16` 16-byte words (what "achievable HBM read bandwidth" means
// on the device at hand).  Every lane keeps 8 independent 16-B loads in flight; the xor-sum is stored by one lane only
// if it hits an impossible value, so the loads cannot be optimised away and nothing is written.
template <bool NT>
__global__ __launch_bounds__(256) void read_probe_kernel(const uint4 *buf, size_t n16, unsigned *sink) {
  typedef unsigned vec4u __attribute__((ext_vector_type(4)));
  const vec4u *p = reinterpret_cast<const vec4u *>(buf);
  const size_t stride = (size_t)gridDim.x * 256;
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  vec4u acc = {0, 0, 0, 0};
  for (; i + 7 * stride < n16; i += 8 * stride) {
    vec4u t[8];
#pragma unroll
    for (int j = 0; j < 8; j++) t[j] = NT ? __builtin_nontemporal_load(p + i + j * stride) : p[i + j * stride];
#pragma unroll
    for (int j = 0; j < 8; j++) acc ^= t[j];
  }
  for (; i < n16; i += stride) acc ^= (NT ? __builtin_nontemporal_load(p + i) : p[i]);
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u && threadIdx.x == 0 && blockIdx.x == 12345678u) *sink = acc.x;
}

template <typename F>
static int launch_phase(void *ph, const int *mom_h, long long locV3, int Nmom, int FTSign, const int localL[4],
                        const int totalL[4], const int commCoord[4], hipStream_t stream) {
  void *mom_d = nullptr;
  int st = upload_table(&mom_d, mom_h, sizeof(int) * 3 * (size_t)Nmom, stream);
  if (st) return st;
  PhaseGeom g;
  g.locV3 = locV3;
  g.Nmom = Nmom;
  g.FTSign = FTSign;
  for (int d = 0; d < 3; d++) {
    g.localL[d] = localL[d];
    g.totalL[d] = totalL[d];
    g.commCoord[d] = commCoord ? commCoord[d] : 0;
  }
  const unsigned grid = (unsigned)((locV3 + kUtilBlock - 1) / kUtilBlock);
  hipLaunchKernelGGL((phase_matrix_kernel<F>), dim3(grid), dim3(kUtilBlock), 0, stream, static_cast<Cplx<F> *>(ph),
                     static_cast<const int *>(mom_d), g);
  MUGIQ_CHECK_HIP(hipGetLastError());
  return MUGIQ_HIP_SUCCESS;
}

template <typename F>
static int launch_convert(void *out, const void *in, int nData, int volumeCB, const int localL[4], hipStream_t stream) {
  ConvertGeom g;
  for (int d = 0; d < 4; d++) g.X[d] = localL[d];
  g.volumeCB = volumeCB;
  g.nData = nData;
  g.locV3 = (long long)localL[0] * localL[1] * localL[2];
  const size_t shmem = sizeof(Cplx<F>) * kTileV * (size_t)(localL[3] + 1);
  MUGIQ_REQUIRE(shmem <= 160 * 1024, "convertIdxOrder_mapGamma: local time extent %d too large for the LDS tile", localL[3]);
  if (shmem > 64 * 1024)
    MUGIQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(convert_idx_map_gamma_kernel<F>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  const dim3 grid((unsigned)((g.locV3 + kTileV - 1) / kTileV), (unsigned)nData);
  hipLaunchKernelGGL((convert_idx_map_gamma_kernel<F>), grid, dim3(kUtilBlock), shmem, stream, static_cast<Cplx<F> *>(out),
                     static_cast<const Cplx<F> *>(in), g);
  MUGIQ_CHECK_HIP(hipGetLastError());
  return MUGIQ_HIP_SUCCESS;
}

}  // namespace mugiq

using namespace mugiq;

namespace mugiq {
// Fills every CU's LDS with signalling-NaN bit patterns (test aid): LDS is not cleared between kernels, so a kernel that reads a cell
// it never wrote -- padding of a tile image, a lane of an operand that "multiplies zero anyway" -- sees whatever the last kernel left;
// with this in front it sees NaN and the result shows it.
__global__ __launch_bounds__(1024) void poison_lds_kernel(unsigned long long *sink, int words) {
  extern __shared__ __align__(16) unsigned long long lds_words[];
  for (int i = threadIdx.x; i < words; i += 1024) lds_words[i] = 0x7ff4dead7ff4deadULL;
  __syncthreads();
  if (words < 0) sink[0] = lds_words[threadIdx.x];  // (never: keeps the stores)
}
}  // namespace mugiq

extern "C" int mugiq_hip_debug_poison_lds(void *stream);
namespace mugiq {
int debug_poison_lds_if_asked(hipStream_t stream) {
  const char *e = getenv("MUGIQ_HIP_DEBUG_POISON_LDS");
  if (!e || atoi(e) == 0) return MUGIQ_HIP_SUCCESS;
  return mugiq_hip_debug_poison_lds(stream);
}
}  // namespace mugiq

extern "C" {

int mugiq_hip_debug_poison_lds(void *stream) {
  static void *sink = nullptr;
  if (!sink) MUGIQ_CHECK_HIP(hipMalloc(&sink, 64));
  const int bytes = 160 * 1024 - 64;
  MUGIQ_CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(mugiq::poison_lds_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  // one workgroup owns a CU's whole LDS: 4096 of them pass over every CU many times
  hipLaunchKernelGGL(mugiq::poison_lds_kernel, dim3(4096), dim3(1024), bytes, static_cast<hipStream_t>(stream), static_cast<unsigned long long *>(sink), bytes / 8);
  MUGIQ_CHECK_HIP(hipGetLastError());
  return MUGIQ_HIP_SUCCESS;
}

int mugiq_hip_create_phase_matrix(void *phaseMatrix_d, const int *momMatrix_h, long long locV3, int Nmom, int FTSign,
                                  const int localL[4], const int totalL[4], const int commCoord[4], int precision,
                                  void *stream) {
  const char *who = "createPhaseMatrixGPU";
  MUGIQ_REQUIRE(phaseMatrix_d && momMatrix_h && localL && totalL, "%s: NULL argument", who);
  MUGIQ_REQUIRE(precision == 4 || precision == 8, "%s: Precision not supported! (%d)", who, precision);
  MUGIQ_REQUIRE(Nmom >= 1, "%s: Nmom = %d", who, Nmom);
  MUGIQ_REQUIRE(FTSign == 1 || FTSign == -1, "%s: FTSign = %d must be +1 or -1", who, FTSign);
  MUGIQ_REQUIRE(locV3 == (long long)localL[0] * localL[1] * localL[2], "%s: locV3 = %lld does not match localL", who, locV3);
  for (int d = 0; d < 3; d++) MUGIQ_REQUIRE(totalL[d] > 0, "%s: totalL[%d] = %d", who, d, totalL[d]);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (precision == 8) return launch_phase<double>(phaseMatrix_d, momMatrix_h, locV3, Nmom, FTSign, localL, totalL, commCoord, s);
  return launch_phase<float>(phaseMatrix_d, momMatrix_h, locV3, Nmom, FTSign, localL, totalL, commCoord, s);
}

int mugiq_hip_probe_read_bandwidth(const void *buf_d, size_t bytes, int nonTemporal, void *stream) {
  MUGIQ_REQUIRE(buf_d != nullptr && bytes >= 16 && (reinterpret_cast<uintptr_t>(buf_d) & 15) == 0,
                "mugiq_hip_probe_read_bandwidth: need a 16-byte aligned buffer of >= 16 bytes");
  hipStream_t s = static_cast<hipStream_t>(stream);
  void *sink = nullptr;
  int st = stream_scratch(&sink, 64, s);
  if (st) return st;
  const size_t n16 = bytes / 16;
  const unsigned grid = 256 * 16;  // 16 workgroups of 256 lanes per CU
  if (nonTemporal)
    hipLaunchKernelGGL((read_probe_kernel<true>), dim3(grid), dim3(256), 0, s, static_cast<const uint4 *>(buf_d), n16, static_cast<unsigned *>(sink));
  else
    hipLaunchKernelGGL((read_probe_kernel<false>), dim3(grid), dim3(256), 0, s, static_cast<const uint4 *>(buf_d), n16, static_cast<unsigned *>(sink));
  MUGIQ_CHECK_HIP(hipGetLastError());
  return MUGIQ_HIP_SUCCESS;
}

int mugiq_hip_convert_idx_order_map_gamma(void *dataPosMP_d, const void *dataPos_d, int nData, int nLoop, int nParity,
                                          int volumeCB, const int localL[4], int precision, void *stream) {
  const char *who = "convertIdxOrder_mapGamma";
  MUGIQ_REQUIRE(dataPosMP_d && dataPos_d && localL, "%s: NULL argument", who);
  // lib/contract_wrappers.cu:138
  MUGIQ_REQUIRE(nData == nLoop * 16, "%s: This function assumes that nData = nLoop * NGamma", who);
  MUGIQ_REQUIRE(nLoop >= 1, "%s: nLoop = %d", who, nLoop);
  MUGIQ_REQUIRE(nParity == 2, "%s: only Full Site Subset (nParity = 2) loop buffers are supported, got %d", who, nParity);
  MUGIQ_REQUIRE(precision == 4 || precision == 8, "%s: Precision not supported! (%d)", who, precision);
  MUGIQ_REQUIRE((long long)localL[0] * localL[1] * localL[2] * localL[3] == 2LL * volumeCB,
                "%s: volumeCB = %d does not match localL", who, volumeCB);
  MUGIQ_REQUIRE(dataPosMP_d != dataPos_d, "%s: in-place conversion is not supported", who);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (precision == 8) return launch_convert<double>(dataPosMP_d, dataPos_d, nData, volumeCB, localL, s);
  return launch_convert<float>(dataPosMP_d, dataPos_d, nData, volumeCB, localL, s);
}

}  // extern "C"
