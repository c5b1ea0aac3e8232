// How fast can one MI355X store 40 GB of 16-byte elements, as a function of how the lanes of a store instruction are laid
// out in memory?  Mimics prolongateEvecs at 32^4, 200 eigenvectors: "fields" of 12 planes x V/2 x 2 parities complex doubles.
//   mode 0: a wave instruction writes 64 consecutive checkerboard sites of one plane of one field (1 KB run)
//   mode 1: ... 16 consecutive sites of 4 different fields (256-byte runs; the vector prolongation kernel's pattern)
//   mode 3 / 4: like mode 2 but 64-byte / 128-byte pieces (2 / 4 x-neighbouring aggregates side by side in the instruction)
//   mode 5: the 32-byte pieces of mode 2, but the wave that writes aggregate ax's pieces writes those of its three x-neighbours in the next
//           three instructions (the lines are completed by consecutive instructions of one wave instead of inside one instruction)
//   mode 2: the matrix-pipe kernel's pattern: 16 lanes = the (x 0..3) x (y 0..3) sites of one 4^4 aggregate at fixed (z, t):
//           eight 32-byte pieces (two x_cb of one parity) in eight different lines, 4 fields per instruction
// build: hipcc --offload-arch=gfx950 -O3 -o store_pattern_probe store_pattern_probe.hip ; run: ./store_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef double vec2 __attribute__((ext_vector_type(2)));
constexpr int L = 32, VCB = L * L * L * L / 2, NF = 200, NPL = 12;

template <int MODE> __global__ __launch_bounds__(256) void store_kernel(vec2 *base, int nf) {
  // one workgroup: 4 waves; unit of work u = blockIdx.x: MODE 0/1: (tile of 64 | 16 x_cb, parity); MODE 2: one (aggregate, round)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t fieldStride = (size_t)2 * NPL * VCB;
  const vec2 val = {1.0 + lane, 2.0};
  if (MODE == 0) {
    const size_t tile = blockIdx.x;  // 64 consecutive x_cb of one parity
    const size_t pty = tile / (VCB / 64), x0 = (tile % (VCB / 64)) * 64;
    for (int f = wave; f < nf; f += 4)
      for (int p = 0; p < NPL; p++) base[f * fieldStride + pty * NPL * VCB + (size_t)p * VCB + x0 + lane] = val;
  } else if (MODE == 1) {
    const size_t tile = blockIdx.x;  // 16 consecutive x_cb of one parity
    const size_t pty = tile / (VCB / 16), x0 = (tile % (VCB / 16)) * 16;
    const int kq = lane >> 4, s = lane & 15;
    for (int f = 4 * wave + kq; f < nf; f += 16)
      for (int p = 0; p < NPL; p++) base[f * fieldStride + pty * NPL * VCB + (size_t)p * VCB + x0 + s] = val;
  } else {
    // aggregate (ax, ay, az, at) of 4^4, round r = (z, t) inside it: sites x 0..3, y 0..3
    int u = blockIdx.x;
    const int r = u & 15; u >>= 4;
    const int ax = u & 7; u >>= 3;
    const int ay = u & 7; u >>= 3;
    const int az = u & 7; u >>= 3;
    const int at = u & 7;
    const int kq = lane >> 4, s = lane & 15;
    const int x = 4 * ax + (s & 3), y = 4 * ay + (s >> 2), z = 4 * az + (r & 3), t = 4 * at + (r >> 2);
    const size_t pty = (x + y + z + t) & 1, xcb = ((size_t)x + L * ((size_t)y + L * ((size_t)z + L * (size_t)t))) >> 1;
    if (MODE == 5) {
      if (ax % 4) return;
      for (int f = 4 * wave + kq; f < nf; f += 16)
        for (int p = 0; p < NPL; p++)
#pragma unroll
          for (int gg = 0; gg < 4; gg++) base[f * fieldStride + pty * NPL * VCB + (size_t)p * VCB + xcb + 2 * gg] = val;
    } else if (MODE == 2) {
      for (int f = 4 * wave + kq; f < nf; f += 16)
        for (int p = 0; p < NPL; p++) base[f * fieldStride + pty * NPL * VCB + (size_t)p * VCB + xcb] = val;
    } else {
      // G = 2 | 4 aggregates side by side: 16 lanes = (G x 2 x_cb) x (16 / (2 G) rows y) of ONE parity; the unit of work covers the
      // same 16 sites x G aggregates in G x 2 instructions (both parities)
      constexpr int G = MODE == 3 ? 2 : 4;
      const int ag = ax / G * G;                       // first aggregate of the group (every G-th workgroup does the group's work ...)
      if (ax % G) return;                              // ... the others exit: same bytes in total as mode 2
      const int rows = 16 / (2 * G);                   // y rows per instruction
      for (int par = 0; par < 2; par++)
        for (int yb = 0; yb < 4; yb += rows) {
          const int q = s % (2 * G), yy = 4 * ay + yb + s / (2 * G);
          const size_t rowcb = ((size_t)(4 * ag) + L * ((size_t)yy + L * ((size_t)z + L * (size_t)t))) >> 1;   // x_cb of x = 4 ag in this row
          for (int f = 4 * wave + kq; f < nf; f += 16)
            for (int p = 0; p < NPL; p++) base[f * fieldStride + (size_t)par * NPL * VCB + (size_t)p * VCB + rowcb + q] = val;
        }
    }
  }
}

int main() {
  const size_t elems = (size_t)NF * 2 * NPL * VCB;
  vec2 *d = nullptr;
  CHK(hipMalloc(&d, elems * sizeof(vec2)));
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  const char *names[6] = {"1 KB runs (64 lanes contiguous, one field)", "256-byte runs, 4 fields per instruction", "32-byte pieces of an aggregate, 4 fields per instruction",
                          "64-byte pieces (2 aggregates side by side)", "128-byte pieces (4 aggregates side by side)",
                          "32-byte pieces, the other three quarters of each line in the next three instructions of the same wave"};
  for (int mode = 0; mode < 6; mode++) {
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
      CHK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(2 * VCB / 64), dim3(256), 0, 0, d, NF);
      else if (mode == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(2 * VCB / 16), dim3(256), 0, 0, d, NF);
      else if (mode == 2) hipLaunchKernelGGL(store_kernel<2>, dim3(8 * 8 * 8 * 8 * 16), dim3(256), 0, 0, d, NF);
      else if (mode == 3) hipLaunchKernelGGL(store_kernel<3>, dim3(8 * 8 * 8 * 8 * 16), dim3(256), 0, 0, d, NF);
      else if (mode == 4) hipLaunchKernelGGL(store_kernel<4>, dim3(8 * 8 * 8 * 8 * 16), dim3(256), 0, 0, d, NF);
      else hipLaunchKernelGGL(store_kernel<5>, dim3(8 * 8 * 8 * 8 * 16), dim3(256), 0, 0, d, NF);
      CHK(hipEventRecord(e1));
      CHK(hipEventSynchronize(e1));
      float ms;
      CHK(hipEventElapsedTime(&ms, e0, e1));
      if (rep > 0 && ms < best) best = ms;
    }
    printf("{\"mode\": %d, \"pattern\": \"%s\", \"GB\": %.2f, \"ms\": %.3f, \"TBps\": %.3f}\n", mode, names[mode], elems * 16 / 1e9, best, elems * 16 / 1e9 / best);
  }
  CHK(hipFree(d));
  return 0;
}
