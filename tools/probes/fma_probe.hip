// Diagnostic: sustained vector FMA rate (fp64 / fp32 / packed fp32) of the device at hand -> effective clock under load.
// build: hipcc --offload-arch=gfx950 -O3 -o fma_probe fma_probe.hip ; run: ./fma_probe
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename T, int CHAINS> __global__ __launch_bounds__(256) void fma_kernel(T *out, int iters, T a, T b) {
  T x[CHAINS];
#pragma unroll
  for (int i = 0; i < CHAINS; i++) x[i] = T(threadIdx.x + i);
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < CHAINS; i++) {
      if constexpr (sizeof(T) == 8) x[i] = __builtin_fma(x[i], a, b);
      else x[i] = __builtin_fmaf(x[i], a, b);
    }
  }
  T s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; i++) s += x[i];
  if (s == T(-1.2345)) out[0] = s;
}

typedef float float2v __attribute__((ext_vector_type(2)));
template <int CHAINS> __global__ __launch_bounds__(256) void pkfma_kernel(float *out, int iters, float a, float b) {
  float2v x[CHAINS];
  const float2v av = {a, a}, bv = {b, b};
#pragma unroll
  for (int i = 0; i < CHAINS; i++) x[i] = float2v{(float)threadIdx.x, (float)i};
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int i = 0; i < CHAINS; i++) x[i] = __builtin_elementwise_fma(x[i], av, bv);
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < CHAINS; i++) s += x[i].x + x[i].y;
  if (s == -1.2345f) out[0] = s;
}

template <typename K> static double time_ms(K launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; r++) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 5;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  void *out;
  hipMalloc(&out, 64);
  const int blocks = cus * 8, iters = 20000;
  constexpr int C = 16;
  const double n = (double)blocks * 256 * iters * C;
  double ms = time_ms([&] { hipLaunchKernelGGL((fma_kernel<double, C>), dim3(blocks), dim3(256), 0, 0, (double *)out, iters, 1.0000001, 1e-9); });
  printf("{\"cus\": %d, \"clock_MHz_reported\": %d, \"fp64_TFLOPs\": %.2f, \"fp64_implied_GHz\": %.3f", cus, p.clockRate / 1000, 2 * n / ms / 1e9,
         n / ms / 1e6 / (cus * 4.0 * 16.0));
  ms = time_ms([&] { hipLaunchKernelGGL((fma_kernel<float, C>), dim3(blocks), dim3(256), 0, 0, (float *)out, iters, 1.0000001f, 1e-9f); });
  printf(", \"fp32_TFLOPs\": %.2f, \"fp32_implied_GHz\": %.3f", 2 * n / ms / 1e9, n / ms / 1e6 / (cus * 4.0 * 16.0));
  ms = time_ms([&] { hipLaunchKernelGGL((pkfma_kernel<C>), dim3(blocks), dim3(256), 0, 0, (float *)out, iters, 1.0000001f, 1e-9f); });
  printf(", \"pk_fp32_TFLOPs\": %.2f, \"pk_fp32_implied_GHz\": %.3f}\n", 4 * n / ms / 1e9, n / ms / 1e6 / (cus * 4.0 * 16.0));
  return 0;
}
