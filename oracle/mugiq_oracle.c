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
 * Build: make -C oracle   (gcc -O3 -fcx-limited-range -fopenmp)
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

#define DEFINE_CONTRACT(NAME, REAL, CPLX)                                                                        \
  /* site-outer, eigenvector-inner; per site the arithmetic and its order are those of :98-120 */               \
  void NAME(CPLX *loop, const CPLX *const *vL, const CPLX *const *vR, const double *sigma, int nVec,             \
            int64_t site_begin, int64_t site_end, int volumeCB, int64_t stride, int64_t parity_offset, int order) { \
    const int64_t V = 2 * (int64_t)volumeCB;                                                                      \
    _Pragma("omp parallel for schedule(static)") for (int64_t tid = site_begin; tid < site_end; tid++) {          \
      const int pty = tid >= volumeCB;                                                                             \
      const int64_t x_cb = tid - (int64_t)pty * volumeCB;                                                          \
      for (int n = 0; n < nVec; n++) {                                                                             \
        const REAL inv_sigma = (REAL)(1.0 / (REAL)sigma[n]);                                                       \
        CPLX l[12], r[12], resG[16];                                                                               \
        for (int k = 0; k < 12; k++) {                                                                             \
          const int64_t i = spinor_index(order, pty, x_cb, k, stride, parity_offset);                              \
          l[k] = vL[n][i];                                                                                         \
          r[k] = vR[n][i];                                                                                         \
        }                                                                                                          \
        for (int be = 0; be < 4; be++)                                                                             \
          for (int al = 0; al < 4; al++) {                                                                         \
            CPLX s = 0;                                                                                            \
            for (int kc = 0; kc < 3; kc++) s += conj(l[be * 3 + kc]) * r[al * 3 + kc];                             \
            resG[be * 4 + al] = s;                                                                                 \
          }                                                                                                        \
        for (int iG = 0; iG < 16; iG++) {                                                                          \
          CPLX trace = 0;                                                                                          \
          for (int s2 = 0; s2 < 4; s2++) {                                                                         \
            const int s1 = COLUMN_INDEX[iG][s2];                                                                   \
            const CPLX g = (REAL)ROW_VALUE[iG][s2][0] + (REAL)ROW_VALUE[iG][s2][1] * I;                            \
            trace += g * resG[s2 * 4 + s1];                                                                        \
          }                                                                                                        \
          loop[tid + V * iG] += inv_sigma * trace;                                                                 \
        }                                                                                                          \
      }                                                                                                            \
    }                                                                                                              \
  }

DEFINE_CONTRACT(oracle_loop_contract_f64, double, double complex)
DEFINE_CONTRACT(oracle_loop_contract_f32, float, float complex)
