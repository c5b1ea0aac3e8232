// Reflected displacement entries in MOMENTUM space (host side; no device code).
//
// csrc/reflect.hip derives the slots of a "-mu" entry from those of the "+mu" entry (or the other way round) in position space:
// L^-_{k,G}(x) = eta_G conj(L^+_{k,G}(x - k mu)).  Its Fourier transform is a phase, the momentum reversed and a conjugation,
//     F^-_{k}(p, ig, t) = eta(15 - ig) e^{+ i s 2 pi p_mu k / L_mu} conj( F^+_{k}(-p, ig, t) )          mu = x, y, z
//     F^-_{k}(p, ig, t) = eta(15 - ig) conj( F^+_{k}(-p, ig, t - k) )                                      mu = t
// (s = FTSign; opposite phase / shift for a "+" entry derived from a "-" one; ig is the OUTPUT channel of the G -> g5 G map
// of convertIdxOrder_mapGamma, lib/mugiq_util_kernels.cu:59-99, whose sign is real: the channel it came from is 15 - ig).
// So when only momentum-space output is asked for (the reference writes nothing else: lib/loop_mugiq.cpp:660-663) and the
// momentum list holds -p for every p, the reflected slots need not exist in position space at all: the OPT plan leaves them
// out of the reorder + Fourier kernels (12 of the 25 slots of BASELINE.json configs[2]) and fills them in here, on the
// gathered array dataMom_bcast (16 Nmom totT complex per slot: microseconds), where a shift in t is a shift of the global
// index -- no halo, no matter how t is partitioned.  Identity checked on the CPU restatement of the reference
// (tests/test_oracle_kat.py), this function against the oracle in tests/test_reflect_mom_cpu.py.
#include <cmath>
#include <complex>
#include <vector>

#include "internal.h"

namespace mugiq {

// neg[im] = index of -p(im) in the list, or false if some momentum has no partner
bool momenta_negation_table(const int *mom, int Nmom, std::vector<int> &neg) {
  neg.assign(Nmom, -1);
  for (int i = 0; i < Nmom; i++) {
    for (int j = 0; j < Nmom; j++)
      if (mom[3 * j] == -mom[3 * i] && mom[3 * j + 1] == -mom[3 * i + 1] && mom[3 * j + 2] == -mom[3 * i + 2]) {
        neg[i] = j;
        break;
      }
    if (neg[i] < 0) return false;
  }
  return true;
}

static int gamma_dagger_sign_host(int n) {
  const int m = __builtin_popcount((unsigned)n);
  return ((m * (m - 1) / 2) & 1) ? -1 : 1;
}

template <typename F>
static void reflect_mom(std::complex<F> *data, int Nmom, const int *mom, const std::vector<int> &neg, int FTSign, const int totalL[4],
                        int nLoop, int locT, int totT, int dstSlot, int srcSlot, int dir, int dstSign, int k) {
  const int nRanksT = totT / locT;
  const size_t slab = (size_t)16 * Nmom * locT * nLoop;  // nElemMomLoc: one time rank's share
  // element (im, iL, ig, global t = r * locT + t) sits at r * slab + t + locT * (ig + 16 * (iL + nLoop * im)): runs of locT
  // consecutive t.  A shift along t by +-k maps a run onto (at most) two runs of the source: positions through a table.
  const bool dstPlus = dstSign == MUGIQ_HIP_DISP_SIGN_PLUS;
  std::vector<size_t> srcT(totT);  // offset (rank slab + local t) of the SOURCE time slice of every global t
  for (int tg = 0; tg < totT; tg++) {
    const int ts = dir == 3 ? ((tg + (dstPlus ? k : -k)) % totT + totT) % totT : tg;
    srcT[tg] = (size_t)(ts / locT) * slab + (size_t)(ts % locT);
  }
  for (int im = 0; im < Nmom; im++) {
    double cr = 1.0, ci = 0.0;
    if (dir < 3) {
      const double arg = (dstPlus ? -1.0 : 1.0) * FTSign * 2.0 * M_PI * (double)mom[3 * im + dir] * (double)k / (double)totalL[dir];
      cr = std::cos(arg);
      ci = std::sin(arg);
    }
    for (int ig = 0; ig < 16; ig++) {
      const double eta = gamma_dagger_sign_host(15 - ig);
      const double er = eta * cr, ei = eta * ci;  // dst = (er + i ei) * conj(src)
      const size_t dOff = (size_t)locT * (ig + 16 * ((size_t)dstSlot + (size_t)nLoop * im));
      const size_t sOff = (size_t)locT * (ig + 16 * ((size_t)srcSlot + (size_t)nLoop * neg[im]));
      for (int r = 0; r < nRanksT; r++) {
        F *d = reinterpret_cast<F *>(data + (size_t)r * slab + dOff);
        const size_t *st = &srcT[(size_t)r * locT];
        // the source time slices of a run are consecutive except where the shift crosses a slab boundary: contiguous pieces
        for (int t0 = 0; t0 < locT;) {
          int t1 = t0 + 1;
          while (t1 < locT && st[t1] == st[t1 - 1] + 1) t1++;
          const F *__restrict v = reinterpret_cast<const F *>(data + st[t0] + sOff);
          F *__restrict o = d + 2 * t0;
          const int n = t1 - t0;
          for (int t = 0; t < n; t++) {
            const double vr = v[2 * t], vi = v[2 * t + 1];
            o[2 * t] = (F)(er * vr + ei * vi);      // (er + i ei)(vr - i vi)
            o[2 * t + 1] = (F)(ei * vr - er * vi);
          }
          t0 = t1;
        }
      }
    }
  }
}

}  // namespace mugiq

extern "C" int mugiq_hip_reflect_momentum_space(void *dataMom_bcast_h, int precision, int Nmom, const int *momMatrix_h, int FTSign,
                                                const int totalL[4], int nLoop, int locT, int totT, int dstSlot, int srcSlot,
                                                int dispDir, int dstDispSign, int length) {
  using namespace mugiq;
  const char *who = "mugiq_hip_reflect_momentum_space";
  MUGIQ_REQUIRE(dataMom_bcast_h && momMatrix_h && totalL, "%s: NULL argument", who);
  MUGIQ_REQUIRE(precision == 4 || precision == 8, "%s: precision %d", who, precision);
  MUGIQ_REQUIRE(Nmom >= 1 && (FTSign == 1 || FTSign == -1), "%s: Nmom = %d, FTSign = %d", who, Nmom, FTSign);
  MUGIQ_REQUIRE(nLoop >= 1 && locT >= 1 && totT >= locT && totT % locT == 0, "%s: invalid sizes nLoop=%d locT=%d totT=%d", who, nLoop, locT, totT);
  MUGIQ_REQUIRE(dstSlot >= 0 && dstSlot < nLoop && srcSlot >= 0 && srcSlot < nLoop && dstSlot != srcSlot, "%s: slots %d <- %d of %d", who, dstSlot, srcSlot, nLoop);
  MUGIQ_REQUIRE(dispDir >= 0 && dispDir <= 3 && (dstDispSign == 0 || dstDispSign == 1) && length >= 1, "%s: displacement (%d, %d, %d)", who, dispDir, dstDispSign, length);
  for (int d = 0; d < 4; d++) MUGIQ_REQUIRE(totalL[d] > 0, "%s: totalL[%d] = %d", who, d, totalL[d]);
  std::vector<int> neg;
  if (!momenta_negation_table(momMatrix_h, Nmom, neg))
    return set_error(MUGIQ_HIP_ERROR_UNSUPPORTED, "%s: the momentum list is not closed under p -> -p", who);
  if (precision == 8)
    reflect_mom<double>(static_cast<std::complex<double> *>(dataMom_bcast_h), Nmom, momMatrix_h, neg, FTSign, totalL, nLoop, locT, totT, dstSlot, srcSlot, dispDir, dstDispSign, length);
  else
    reflect_mom<float>(static_cast<std::complex<float> *>(dataMom_bcast_h), Nmom, momMatrix_h, neg, FTSign, totalL, nLoop, locT, totT, dstSlot, srcSlot, dispDir, dstDispSign, length);
  return MUGIQ_HIP_SUCCESS;
}
