// How fast do the eigenvectors ARRIVE when a workgroup walks them the way the tiled displaced contraction does -- with no arithmetic,
// no LDS and no barrier at all?  48.48.24.24 checkerboard fields (12 planes x 2 parities x V/2 complex doubles = 255 MB each), N of them;
// every workgroup loops over the fields and each thread keeps two fields in flight (the kernels' register prefetch).
//   mode 0: 1024 threads <-> 1024 consecutive checkerboard entries of one plane, planes in turn (the streaming reference: 16 KB runs)
//   mode 1: the 8 x 16 column tile along z: thread <-> (position 0..10, spin, line 0..15), three colour planes each; 256-byte pieces,
//           positions 1152 entries apart; 11 of 16 waves load (1.375 units per site)
//   mode 2: the same tile along t (positions 27648 entries apart)
//   mode 3: the 12 x 16 tile along z (15 positions)
//   mode 4: the 4 x 32 tile along z (7 positions, 512-byte pieces)
//   mode 5: the x row tile: 2 rows x 24 entries x (parity, spin) = 384 of 512 threads, three colours each (768-byte pieces)
//   mode 6: 64 lines x 1 position per tile (1 KB pieces, 4 positions staged: what long pieces alone would buy)
// (column tiles: workgroups dealt XCD-contiguously like the kernels' -- the tiles of one line group run side by side on one XCD and the
//  positions two tiles share are L2 hits there; with the plain round-robin order they are second HBM reads: 6.6 TB/s REQUESTED in every mode)
//   mode 7: mode 1 with the workgroups of one tile column walking the eigenvectors in a staggered order (workgroup b starts at field b % N)
// build: hipcc --offload-arch=gfx950 -O3 [-DPROBE_NT] -o load_pattern_probe load_pattern_probe.hip ; run: ./load_pattern_probe [N]
// (-DPROBE_NT: non-temporal loads -- nothing stays in L2, every request is an HBM read)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef double vec2 __attribute__((ext_vector_type(2)));
#ifdef PROBE_NT
#define LOADV(p_) __builtin_nontemporal_load(p_)
#else
#define LOADV(p_) (*(p_))
#endif
constexpr int X0 = 48, X1 = 48, X2 = 24, X3 = 24;
constexpr long long VCB = (long long)X0 * X1 * X2 * X3 / 2, FIELD = 24 * VCB;  // elements (vec2) per field

template <int MODE> __global__ __launch_bounds__(1024) void load_kernel(const vec2 *base, int nf, vec2 *sink) {
  const int t = threadIdx.x;
  long long off[3];
  bool active = true;
  long long blk = blockIdx.x;
  if (MODE != 0 && MODE != 5 && (gridDim.x & 7) == 0) blk = (long long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // XCD-contiguous, as the kernels
  int start = 0;
  if (MODE == 0) {
    // 1024 consecutive entries; the 24 planes in turn are covered by 24 x as many workgroups
    const long long chunk = blk;  // chunk of 1024 entries over all planes: FIELD / 1024 chunks; each thread loads ... 3 of them
    for (int c = 0; c < 3; c++) off[c] = (chunk * 3 + c) * 1024 + t;
    active = (chunk * 3 + 2) * 1024 + 1023 < FIELD;
  } else if (MODE == 5) {
    // rows: blk <-> 2 consecutive x rows (both parities); thread <-> (parity, spin, row, entry)
    const int m = t % 24, rest = t / 24, spin = rest & 3, pr = rest >> 2, parity = pr >> 1, row = pr & 1;
    active = t < 384;
    const long long x_cb = (blk * 2 + row) * 24 + m;
    for (int c = 0; c < 3; c++) off[c] = (long long)parity * 12 * VCB + (long long)(3 * spin + c) * VCB + (active ? x_cb : 0);
  } else {
    constexpr int TJ = MODE == 3 ? 12 : (MODE == 4 ? 4 : (MODE == 6 ? 1 : 8));
    constexpr int LN = MODE == 4 ? 32 : (MODE == 6 ? 64 : 16);
    constexpr int NP = TJ + 3;
    constexpr int DIRT = MODE == 2;
    const long long strideMu = DIRT ? (long long)X0 * X1 * X2 / 2 : (long long)X0 * X1 / 2;
    const int J = 24, nJT = J / TJ;
    const long long colsPerParity = VCB / J;  // lines per parity
    const int pos = t / (4 * LN), spin = (t / LN) & 3, line = t % LN;
    active = pos < NP;
    const int jt = (int)(blk % nJT);
    const long long cc = blk / nJT;
    long long cid = cc * LN + line;  // line index over both parities
    const int p0 = (int)(cid / colsPerParity);
    const long long rem = cid % colsPerParity;
    const long long hi = rem / strideMu, lo = rem % strideMu;
    const long long b0 = hi * (J * strideMu) + lo;
    int j = jt * TJ + (active ? pos : 0);
    if (j >= J) j -= J;
    const int par = p0 ^ (j & 1);
    for (int c = 0; c < 3; c++) off[c] = (long long)par * 12 * VCB + (long long)(3 * spin + c) * VCB + b0 + j * strideMu;
    if (MODE == 7) start = (int)(blk % nf);
  }
  vec2 acc = {0.0, 0.0};
  if (!active)
    for (int c = 0; c < 3; c++) off[c] = 0;
  vec2 a[3], b[3];
  auto field = [&](int n) { n += start; if (n >= nf) n -= nf; return base + (long long)n * FIELD; };
  for (int c = 0; c < 3; c++) a[c] = LOADV(field(0) + off[c]);
  for (int c = 0; c < 3; c++) b[c] = LOADV(field(nf > 1 ? 1 : 0) + off[c]);
  for (int n = 0; n < nf; n += 2) {
    for (int c = 0; c < 3; c++) acc += a[c];
    const int n2 = n + 2 < nf ? n + 2 : nf - 1;
    for (int c = 0; c < 3; c++) a[c] = LOADV(field(n2) + off[c]);
    for (int c = 0; c < 3; c++) acc += b[c];
    const int n3 = n + 3 < nf ? n + 3 : nf - 1;
    for (int c = 0; c < 3; c++) b[c] = LOADV(field(n3) + off[c]);
  }
  if (acc.x == 12345.678 && acc.y == 1.0) sink[0] = acc;  // (never: keeps the loads)
}

template <int MODE> static int run(const vec2 *base, int nf, vec2 *sink, const char *what, double unique_units) {
  long long nblocks;
  int threads = 1024;
  if (MODE == 0) nblocks = FIELD / 1024 / 3;
  else if (MODE == 5) { nblocks = VCB / 24 / 2; threads = 512; }
  else {
    const int TJ = MODE == 3 ? 12 : (MODE == 4 ? 4 : (MODE == 6 ? 1 : 8)), LN = MODE == 4 ? 32 : (MODE == 6 ? 64 : 16);
    nblocks = (2 * VCB / 24 / LN) * (24 / TJ);
  }
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0));
  CHK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 4; rep++) {
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(load_kernel<MODE>, dim3((unsigned)nblocks), dim3(threads), 0, 0, base, nf, sink);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    if (rep > 0 && ms < best) best = ms;
  }
  const double bytes = (double)nf * FIELD * 16;
  printf("mode %d  %-58s %8.3f ms   %6.2f TB/s of the fields read once   (%4.2f units requested per site: %6.2f TB/s requested)\n", MODE, what, best,
         bytes / best / 1e9, unique_units, bytes * unique_units / best / 1e9);
  return 0;
}

int main(int argc, char **argv) {
  const int nf = argc > 1 ? atoi(argv[1]) : 100;
  vec2 *base, *sink;
  CHK(hipMalloc(&base, (size_t)nf * FIELD * 16));
  CHK(hipMalloc(&sink, 64));
  CHK(hipMemset(base, 0, (size_t)nf * FIELD * 16));
  printf("%d fields of %.1f MB\n", nf, FIELD * 16 / 1e6);
  if (run<0>(base, nf, sink, "streaming: 16 KB runs per workgroup and plane", 1.0)) return 1;
  if (run<1>(base, nf, sink, "8 x 16 tile along z (256-byte pieces, 11 positions)", 11.0 / 8)) return 1;
  if (run<2>(base, nf, sink, "8 x 16 tile along t", 11.0 / 8)) return 1;
  if (run<3>(base, nf, sink, "12 x 16 tile along z (15 positions)", 15.0 / 12)) return 1;
  if (run<4>(base, nf, sink, "4 x 32 tile along z (512-byte pieces, 7 positions)", 7.0 / 4)) return 1;
  if (run<5>(base, nf, sink, "x row tile (2 rows x 24 entries, 768-byte pieces)", 1.0)) return 1;
  if (run<6>(base, nf, sink, "1 x 64 tile along z (1 KB pieces, 4 positions)", 4.0)) return 1;
  if (run<7>(base, nf, sink, "8 x 16 tile along z, staggered start field per workgroup", 11.0 / 8)) return 1;
  return 0;
}
