// a10: momentum projection  dataMom[M x N] = dataPosMP[M x K] * phaseMatrix[K x N]  (column-major),
// M = locT*nData, K = locV3, N = Nmom -- the cublasZgemm / cublasCgemm call of lib/loop_mugiq.cpp:363-378.
//
// The product is skinny (N = number of momenta, typically tens to ~100; M a few hundred to a few
// thousand; K = the local spatial volume), so it is bound by streaming A once per N-tile.  One lane owns
// one row m (A is read coalesced along m, 16 B per lane), a tile of NT momenta is accumulated in
// registers, the phase tile is broadcast from LDS, and K is split over workgroups; partial sums are
// combined in a fixed order by a second kernel (deterministic, no atomics).
#include "internal.h"

#include <mutex>

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

static int own_workspace(void **ptr, size_t bytes) {
  static std::mutex mtx;
  static void *buf[16] = {nullptr};
  static size_t cap[16] = {0};
  std::lock_guard<std::mutex> lock(mtx);
  int dev = 0;
  MUGIQ_CHECK_HIP(hipGetDevice(&dev));
  MUGIQ_REQUIRE(dev >= 0 && dev < 16, "device ordinal %d out of range", dev);
  if (bytes > cap[dev]) {
    if (buf[dev]) {
      MUGIQ_CHECK_HIP(hipDeviceSynchronize());
      MUGIQ_CHECK_HIP(hipFree(buf[dev]));
      buf[dev] = nullptr;
      cap[dev] = 0;
    }
    MUGIQ_CHECK_HIP(hipMalloc(&buf[dev], bytes));
    cap[dev] = bytes;
  }
  *ptr = buf[dev];
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
    int st = own_workspace(&ws, need);
    if (st) return st;
  }
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (precision == 8) return launch_momproj<double>(dataMom_d, dataPosMP_d, phaseMatrix_d, g, ws, s);
  return launch_momproj<float>(dataMom_d, dataPosMP_d, phaseMatrix_d, g, ws, s);
}

}  // extern "C"
