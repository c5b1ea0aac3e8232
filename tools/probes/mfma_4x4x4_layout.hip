// Diagnostic: operand and result lane maps of v_mfma_f64_4x4x4_4b_f64, found with one-hot operands (no guess needed).
// For every pair (la, lb): A = 1 in lane la only, B = 1 in lane lb only; the lanes where D != 0 are written out.  D[b][i][j] =
// sum_k A[b][i][k] B[b][k][j], so (la, lb) gives a non-zero D lane iff la and lb are in the same block and share k.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_4x4x4_layout mfma_4x4x4_layout.hip ; run: ./mfma_4x4x4_layout
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void onehot(int *hit) {
  const int la = blockIdx.x, lb = blockIdx.y, l = threadIdx.x;
  const double a = l == la ? 1.0 : 0.0, b = l == lb ? 1.0 : 0.0;
  double c = 0;
  c = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
  if (c != 0.0) hit[la * 64 + lb] = l + 1;  // at most one lane per pair
}

int main() {
  int *hit;
  hipMalloc(&hit, 64 * 64 * sizeof(int));
  hipMemset(hit, 0, 64 * 64 * sizeof(int));
  hipLaunchKernelGGL(onehot, dim3(64, 64), dim3(64), 0, 0, hit);
  std::vector<int> h(64 * 64);
  hipMemcpy(h.data(), hit, h.size() * sizeof(int), hipMemcpyDeviceToHost);
  int n = 0;
  for (int la = 0; la < 64; la++) {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; lb++)
      if (h[la * 64 + lb]) {
        printf("  (B %2d -> D %2d)", lb, h[la * 64 + lb] - 1);
        n++;
      }
    printf("\n");
  }
  printf("pairs: %d (expected 256)\n", n);
  // test the hypothesis  A: lane = 16 b + 4 k + i, B: lane = 16 b + 4 k + j, D: lane = 16 b + 4 i + j  and the three others
  const char *names[4] = {"D lane = 16b + 4i + j", "D lane = 16b + 4j + i", "D lane = 16j + 4b + i ?", "D lane = 16i + 4b + j ?"};
  for (int hyp = 0; hyp < 4; hyp++) {
    int bad = 0;
    for (int b = 0; b < 4; b++)
      for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
          for (int k = 0; k < 4; k++) {
            const int la = 16 * b + 4 * k + i, lb = 16 * b + 4 * k + j;
            const int ld = hyp == 0 ? 16 * b + 4 * i + j : hyp == 1 ? 16 * b + 4 * j + i : hyp == 2 ? 16 * j + 4 * b + i : 16 * i + 4 * b + j;
            bad += h[la * 64 + lb] != ld + 1;
          }
    printf("A = 16b+4k+i, B = 16b+4k+j, %s: %s\n", names[hyp], bad ? "no" : "EXACT");
  }
  return 0;
}
