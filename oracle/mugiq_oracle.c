/*
 * CPU oracle (plain C) for the 16-gamma loop contraction -- TEST INFRASTRUCTURE ONLY.
 *
 * Restates loopContract_kernel of ckallidonis/mugiq (lib/mugiq_contract_kernels.cu:45-122; gamma tables
 * include/gamma.h:32-71; inv_sigma include/contract_util.cuh:130-134) on QUDA-native FLOAT2 / FLOAT4
 * spinor layouts (ASSUMED from upstream QUDA color_spinor_field_order.h).
 * It is the checker and the `cpu_baseline` ("port") of bench.py; nothing under mugiq_amd/ links it.
 * PARITY UNPINNED: the reference holds no golden vectors for this path; this file is
 * cross-checked against the numpy oracle and the analytic KATs in tests/.
 *
 * Build: make -C oracle   (gcc -O3 -fcx-limited-range -fopenmp; the site-block routine is compiled for avx512f / avx2 / baseline
 * x86-64 side by side and picked at load time, so the library built in one container runs on the GPU box's host)
 */
#include <complex.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static const int ROW_VALUE[16][4][2] = {
    {{1, 0}, {1, 0}, {1, 0}, {1, 0}},     {{0, 1}, {0, 1}, {0, -1}, {0, -1}},   {{-1, 0}, {1, 0}, {1, 0}, {-1, 0}},
    {{0, -1}, {0, 1}, {0, -1}, {0, 1}},   {{0, 1}, {0, -1}, {0, -1}, {0, 1}},   {{-1, 0}, {1, 0}, {-1, 0}, {1, 0}},
    {{0, -1}, {0, -1}, {0, -1}, {0, -1}}, {{1, 0}, {1, 0}, {-1, 0}, {-1, 0}},   {{1, 0}, {1, 0}, {1, 0}, {1, 0}},
    {{0, 1}, {0, 1}, {0, -1}, {0, -1}},   {{-1, 0}, {1, 0}, {1, 0}, {-1, 0}},   {{0, -1}, {0, 1}, {0, -1}, {0, 1}},
    {{0, 1}, {0, -1}, {0, -1}, {0, 1}},   {{-1, 0}, {1, 0}, {-1, 0}, {1, 0}},   {{0, -1}, {0, -1}, {0, -1}, {0, -1}},
    {{1, 0}, {1, 0}, {-1, 0}, {-1, 0}}};
static const int COLUMN_INDEX[16][4] = {{0, 1, 2, 3}, {3, 2, 1, 0}, {3, 2, 1, 0}, {0, 1, 2, 3}, {2, 3, 0, 1}, {1, 0, 3, 2},
                                        {1, 0, 3, 2}, {2, 3, 0, 1}, {2, 3, 0, 1}, {1, 0, 3, 2}, {1, 0, 3, 2}, {2, 3, 0, 1},
                                        {0, 1, 2, 3}, {3, 2, 1, 0}, {3, 2, 1, 0}, {0, 1, 2, 3}};

void oracle_set_num_threads(int n) {
#ifdef _OPENMP
  if (n >= 1) omp_set_num_threads(n);
#else
  (void)n;
#endif
}

int oracle_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

static inline int64_t spinor_index(int order, int parity, int64_t x_cb, int k, int64_t stride, int64_t parity_offset) {
  if (order == 2) return parity * parity_offset + k * stride + x_cb;
  return parity * parity_offset + ((k / 2) * stride + x_cb) * 2 + (k % 2);
}

/* One block of W consecutive sites x all eigenvectors.  Per site the arithmetic and its order are those of :98-120 (colour sum
 * kc = 0..2 from zero, the four-term gamma sum in s2 order from zero, inv_sigma * trace, += into loopData per eigenvector);
 * the W sites of a block are independent lanes (real / imaginary parts in separate arrays so that the compiler can use the
 * host's vector units), and a block touches whole cache lines of every plane. */
#define ORACLE_W 8
/* A thread takes ORACLE_CHUNK consecutive sites at a time and walks the eigenvectors over them (n outer, blocks of W sites inner):
 * every (eigenvector, component) plane is then read in runs of ORACLE_CHUNK * 16 bytes instead of 128 -- with N_ev * 24 planes a
 * thread touching a different 4 KiB page at every step was bound by TLB misses (18 GB/s on 128 threads) -- while the chunk's
 * 16 x ORACLE_CHUNK accumulators (loopData) stay in the thread's cache.  Per site nothing changes: eigenvectors in ascending
 * order, the same arithmetic in the same order. */
#define ORACLE_CHUNK 256
#define DEFINE_CONTRACT(NAME, REAL, CPLX)                                                                        \
  /* one eigenvector (scaled by inv_sigma) on one block of w <= W sites starting at tid0 */                       \
  __attribute__((target_clones("avx512f", "avx2", "default"))) static void NAME##_block(                         \
      CPLX *loop, const CPLX *vLn, const CPLX *vRn, REAL inv_sigma, int64_t tid0, int w, int volumeCB, int64_t stride,  \
      int64_t parity_offset, int order) {                                                                         \
    const int64_t V = 2 * (int64_t)volumeCB;                                                                      \
    REAL lr[12][ORACLE_W], li[12][ORACLE_W], rr[12][ORACLE_W], ri[12][ORACLE_W], gr[16][ORACLE_W], gi[16][ORACLE_W]; \
    for (int k = 0; k < 12; k++)                                                                                   \
      for (int j = 0; j < ORACLE_W; j++) {                                                                         \
        const int64_t tid = tid0 + (j < w ? j : 0);                                                                \
        const int pty = tid >= volumeCB;                                                                           \
        const int64_t i = spinor_index(order, pty, tid - (int64_t)pty * volumeCB, k, stride, parity_offset);       \
        lr[k][j] = creal(vLn[i]);                                                                                  \
        li[k][j] = cimag(vLn[i]);                                                                                  \
        rr[k][j] = creal(vRn[i]);                                                                                  \
        ri[k][j] = cimag(vRn[i]);                                                                                  \
      }                                                                                                            \
    for (int be = 0; be < 4; be++)                                                                                 \
      for (int al = 0; al < 4; al++) {                                                                             \
        _Pragma("omp simd") for (int j = 0; j < ORACLE_W; j++) {                                                   \
          REAL sr = 0, si = 0;                                                                                     \
          for (int kc = 0; kc < 3; kc++) { /* s += conj(l) * r */                                                  \
            const REAL a = lr[be * 3 + kc][j], b = li[be * 3 + kc][j], c = rr[al * 3 + kc][j], d = ri[al * 3 + kc][j]; \
            sr += a * c + b * d;                                                                                   \
            si += a * d - b * c;                                                                                   \
          }                                                                                                        \
          gr[be * 4 + al][j] = sr;                                                                                 \
          gi[be * 4 + al][j] = si;                                                                                 \
        }                                                                                                          \
      }                                                                                                            \
    for (int iG = 0; iG < 16; iG++) {                                                                              \
      REAL tr[ORACLE_W], ti[ORACLE_W];                                                                             \
      for (int j = 0; j < ORACLE_W; j++) tr[j] = ti[j] = 0;                                                        \
      for (int s2 = 0; s2 < 4; s2++) {                                                                             \
        const int e = s2 * 4 + COLUMN_INDEX[iG][s2];                                                               \
        const REAL a = (REAL)ROW_VALUE[iG][s2][0], b = (REAL)ROW_VALUE[iG][s2][1];                                 \
        _Pragma("omp simd") for (int j = 0; j < ORACLE_W; j++) { /* trace += g * resG */                           \
          tr[j] += a * gr[e][j] - b * gi[e][j];                                                                    \
          ti[j] += a * gi[e][j] + b * gr[e][j];                                                                    \
        }                                                                                                          \
      }                                                                                                            \
      for (int j = 0; j < w; j++) loop[tid0 + j + V * iG] += inv_sigma * tr[j] + inv_sigma * ti[j] * I;            \
    }                                                                                                              \
  }                                                                                                                \
  void NAME(CPLX *loop, const CPLX *const *vL, const CPLX *const *vR, const double *sigma, int nVec,             \
            int64_t site_begin, int64_t site_end, int volumeCB, int64_t stride, int64_t parity_offset, int order) { \
    const int64_t nchunk = (site_end - site_begin + ORACLE_CHUNK - 1) / ORACLE_CHUNK;                              \
    _Pragma("omp parallel for schedule(static)") for (int64_t c = 0; c < nchunk; c++) {                            \
      const int64_t c0 = site_begin + c * ORACLE_CHUNK;                                                            \
      const int64_t c1 = c0 + ORACLE_CHUNK < site_end ? c0 + ORACLE_CHUNK : site_end;                              \
      for (int n = 0; n < nVec; n++) {                                                                             \
        const REAL inv_sigma = (REAL)(1.0 / (REAL)sigma[n]);                                                       \
        for (int64_t tid0 = c0; tid0 < c1; tid0 += ORACLE_W) {                                                     \
          /* a block must not straddle the parity boundary with a partial lane set: lanes beyond w re-read lane 0 */ \
          const int w = (int)(c1 - tid0 < ORACLE_W ? c1 - tid0 : ORACLE_W);                                        \
          NAME##_block(loop, vL[n], vR[n], inv_sigma, tid0, w, volumeCB, stride, parity_offset, order);            \
        }                                                                                                          \
      }                                                                                                            \
    }                                                                                                              \
  }

DEFINE_CONTRACT(oracle_loop_contract_f64, double, double complex)
DEFINE_CONTRACT(oracle_loop_contract_f32, float, float complex)

/* First-touch placement for the cpu_baseline sample ([2 parities][nPlanes][S] items of itemBytes): copy src to dst with the
 * SAME static partition of the site index tid = x + parity * S the contraction uses, so that every thread's share of every
 * plane lands in memory next to the core that will read it.  dst must be freshly mapped (never written). */
void oracle_first_touch_copy(void *dst, const void *src, int nPlanes, int64_t S, int itemBytes) {
  const int64_t nblk = (S + ORACLE_CHUNK - 1) / ORACLE_CHUNK;  /* (the contraction's chunks; exact when S is a multiple of the chunk) */
#pragma omp parallel for schedule(static)
  for (int64_t b = 0; b < 2 * nblk; b++) {
    const int64_t p = b >= nblk, x0 = (b - p * nblk) * ORACLE_CHUNK;
    const int64_t w = S - x0 < ORACLE_CHUNK ? S - x0 : ORACLE_CHUNK;
    for (int k = 0; k < nPlanes; k++) {
      const int64_t off = ((p * nPlanes + k) * S + x0) * itemBytes;
      char *d = (char *)dst + off;
      const char *s_ = (const char *)src + off;
      for (int64_t i = 0; i < w * itemBytes; i++) d[i] = s_[i];
    }
  }
}
