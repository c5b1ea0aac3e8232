// Diagnostic: fp64 matrix-pipe rate of the device at hand (v_mfma_f64_16x16x4_f64, v_mfma_f64_4x4x4_4b_f64) alone, the fp64
// vector FMA rate alone, and the two TOGETHER -- interleaved inside every wave, and on separate waves of a SIMD.  north_star
// asks that MFMA be used only where it beats the vector path: fp64 MFMA has the vector rate per instruction stream, so it
// pays only if the two pipes really run side by side.  Also checks the operand / result lane maps with exact integer data.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_probe mfma_f64_probe.hip ; run: ./mfma_f64_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

// mode 0: MFMA 16x16x4 only | 1: VALU fma only | 2: both interleaved in every wave (NM mfma : NV fma per step)
// mode 3: waves with even index MFMA only, odd index VALU only (4 waves per SIMD: 2 + 2) | 4: MFMA 4x4x4_4b only
template <int MODE, int NM, int NV> __global__ __launch_bounds__(256) void probe(double *out, int iters, double a, double b) {
  d4 acc[4];
  double x[16];
#pragma unroll
  for (int i = 0; i < 4; i++) acc[i] = d4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 16; i++) x[i] = threadIdx.x + i;
  const double av = threadIdx.x * 1e-3, bv = 1.0 + threadIdx.x * 1e-6;
  const int wave = threadIdx.x >> 6;
  const bool doM = MODE == 0 || MODE == 2 || MODE == 4 || (MODE == 3 && (wave & 1) == 0);
  const bool doV = MODE == 1 || MODE == 2 || (MODE == 3 && (wave & 1) == 1);
  for (int it = 0; it < iters; it++) {
    if (doM) {
#pragma unroll
      for (int i = 0; i < NM; i++) {
        if constexpr (MODE == 4) {
          double r = acc[i & 3].x;
          r = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, r, 0, 0, 0);
          acc[i & 3].x = r;
        } else {
          acc[i & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[i & 3], 0, 0, 0);
        }
      }
    }
    if (doV) {
#pragma unroll
      for (int i = 0; i < NV; i++) x[i & 15] = __builtin_fma(x[i & 15], a, b);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
#pragma unroll
  for (int i = 0; i < 16; i++) s += x[i];
  if (s == -1.2345) out[0] = s;
}

// lane maps with exact integer data: C = A * B with A[i][k] = 1 + i + 16 k (16 x 4), B[k][j] = 2 + 3 j + 7 k (4 x 16, asymmetric)
__global__ void layout_check(double *c_out) {
  const int l = threadIdx.x;
  const double av = 1 + (l & 15) + 16 * (l >> 4);       // A[i = l & 15][k = l >> 4]
  const double bv = 2 + 3 * (l & 15) + 7 * (l >> 4);    // B[k = l >> 4][j = l & 15]
  d4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) c_out[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];  // row = (l >> 4) + 4 r, col = l & 15
}
// 4x4x4_4b: 4 independent 4x4x4 products; guess: block = l >> 4, A[b][i = l & 3][k = (l >> 2) & 3], B[b][k = (l >> 2) & 3][j = l & 3],
// C[b][i = (l >> 2) & 3][j = l & 3] -- the check below prints which of a few candidate maps reproduces the exact products
__global__ void layout_check_4b(double *raw) {
  const int l = threadIdx.x;
  const int b = l >> 4, p = l & 3, q = (l >> 2) & 3;
  const double av = 1 + p + 4 * q + 100 * b;   // candidate: A[b][i = p][k = q]
  const double bv = 2 + 3 * p + 7 * q + 10 * b;  // candidate: B[b][k = q][j = p]
  double c = 0;
  c = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, c, 0, 0, 0);
  raw[l] = c;
}

template <typename K> static double time_ms(K launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 3; r++) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms / 3;
}

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int cus = p.multiProcessorCount;
  double *out;
  hipMalloc(&out, 4096 * sizeof(double));
  // ---- lane maps
  {
    layout_check<<<1, 64>>>(out);
    std::vector<double> c(256);
    hipMemcpy(c.data(), out, 256 * sizeof(double), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 16; i++)
      for (int j = 0; j < 16; j++) {
        double e = 0;
        for (int k = 0; k < 4; k++) e += (1.0 + i + 16 * k) * (2.0 + 3 * j + 7 * k);
        bad += c[i * 16 + j] != e;
      }
    printf("{\"layout_16x16x4\": \"%s\"", bad ? "MISMATCH" : "A[l&15][l>>4], B[l>>4][l&15], C row=(l>>4)+4r col=l&15: exact");
    layout_check_4b<<<1, 64>>>(out);
    std::vector<double> r(64);
    hipMemcpy(r.data(), out, 64 * sizeof(double), hipMemcpyDeviceToHost);
    // candidate result maps: lane l holds C[b][i][j] with (i, j) = (q, p) or (p, q)
    int badQP = 0, badPQ = 0;
    for (int l = 0; l < 64; l++) {
      const int b = l >> 4, pp = l & 3, qq = (l >> 2) & 3;
      double eQP = 0, ePQ = 0;
      for (int k = 0; k < 4; k++) {
        eQP += (1.0 + qq + 4 * k + 100 * b) * (2.0 + 3 * pp + 7 * k + 10 * b);  // C[i = q][j = p]
        ePQ += (1.0 + pp + 4 * k + 100 * b) * (2.0 + 3 * qq + 7 * k + 10 * b);  // C[i = p][j = q]
      }
      badQP += r[l] != eQP;
      badPQ += r[l] != ePQ;
    }
    printf(", \"layout_4x4x4_4b\": \"A[b=l>>4][i=l&3][k=(l>>2)&3], B[b][k=(l>>2)&3][j=l&3]; C at lane: %s\"",
           badQP == 0 ? "[i=(l>>2)&3][j=l&3] exact" : (badPQ == 0 ? "[i=l&3][j=(l>>2)&3] exact" : "NEITHER candidate"));
  }
  // ---- rates
  const int blocks = cus * 4, iters = 4000;  // 4 workgroups of 4 waves per CU: 4 waves per SIMD
  const double waves = (double)blocks * 4;
  auto tf = [&](double flop, double ms) { return flop / ms / 1e9; };
  double ms;
  ms = time_ms([&] { hipLaunchKernelGGL((probe<0, 8, 0>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
  const double mfmaFlop = waves * iters * 8 * 2048.0;
  printf(", \"mfma_16x16x4_only_TFLOPs\": %.2f", tf(mfmaFlop, ms));
  ms = time_ms([&] { hipLaunchKernelGGL((probe<4, 8, 0>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
  printf(", \"mfma_4x4x4_4b_only_TFLOPs\": %.2f", tf(waves * iters * 8 * 512.0, ms));
  ms = time_ms([&] { hipLaunchKernelGGL((probe<1, 0, 64>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
  const double valuFlop = waves * iters * 64 * 128.0;
  printf(", \"valu_fma_only_TFLOPs\": %.2f", tf(valuFlop, ms));
  ms = time_ms([&] { hipLaunchKernelGGL((probe<2, 4, 64>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
  printf(", \"interleaved_in_wave_4mfma_64fma\": {\"total_TFLOPs\": %.2f, \"mfma_TFLOPs\": %.2f, \"valu_TFLOPs\": %.2f}",
         tf(waves * iters * (4 * 2048.0 + 64 * 128.0), ms), tf(waves * iters * 4 * 2048.0, ms), tf(valuFlop, ms));
  ms = time_ms([&] { hipLaunchKernelGGL((probe<3, 4, 64>), dim3(blocks), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9); });
  printf(", \"separate_waves_2mfma_2valu_per_simd\": {\"total_TFLOPs\": %.2f, \"mfma_TFLOPs\": %.2f, \"valu_TFLOPs\": %.2f}}\n",
         tf(waves / 2 * iters * (4 * 2048.0 + 64 * 128.0), ms), tf(waves / 2 * iters * 4 * 2048.0, ms), tf(waves / 2 * iters * 64 * 128.0, ms));
  return 0;
}
