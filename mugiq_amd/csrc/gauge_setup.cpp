// a6 (setup half): the border-extended device gauge field Displace builds once from the host links.
//
// Reference: Displace::createCudaGaugeField + createExtendedCudaGaugeField (lib/displace.cpp:70-134): the host
// links arrive in QDP order (gauge[dir][(parity*V/2 + x_cb)*18 + (row*3+col)*2 + re/im], tests/loop.cpp:88,106),
// are copied into a native FLOAT2 device field, extended by R[d] = 2*commDimPartitioned(d) (:16) with
// copyExtendedGauge, and the borders are filled by cudaGaugeField::exchangeExtendedGhost (:126-127).
// Here the interior is laid out on the host, each dimension's borders are filled in turn (neighbour slabs through
// the comm callback, or a periodic wrap when the dimension is not partitioned) -- later dimensions carry the
// earlier ones' borders along, so edges and corners are filled as in QUDA's extended exchange -- and the result is
// uploaded once.  Setup-time code: plain host loops, device memory only as the transport's staging area.
#include <cstring>
#include <vector>

#include "internal.h"

namespace mugiq {

template <typename F> struct HostGauge {
  std::vector<F> buf;  // native FLOAT2 layout, 2*parity_offset complex
  int XE[4];
  int64_t stride, parity_offset;
  inline int64_t index(int dir, const int c[4], int comp) const {
    const int lex = ((c[3] * XE[2] + c[2]) * XE[1] + c[1]) * XE[0] + c[0];
    const int pty = (c[0] + c[1] + c[2] + c[3]) & 1;
    return pty * parity_offset + (int64_t)(dir * 9 + comp) * stride + (lex >> 1);
  }
};

// visit the slab c[d] in [lo, lo+n) x full extended range in the other dims, in a canonical order
template <typename Fn> static void for_slab(const int XE[4], int d, int lo, int n, Fn fn) {
  int beg[4] = {0, 0, 0, 0}, end[4] = {XE[0], XE[1], XE[2], XE[3]};
  beg[d] = lo;
  end[d] = lo + n;
  int c[4];
  for (c[3] = beg[3]; c[3] < end[3]; c[3]++)
    for (c[2] = beg[2]; c[2] < end[2]; c[2]++)
      for (c[1] = beg[1]; c[1] < end[1]; c[1]++)
        for (c[0] = beg[0]; c[0] < end[0]; c[0]++) fn(c);
}

template <typename F, typename H>
static int fill_extended(const MugiqHipGaugeField *g, const void *const qdp[4], const MugiqHipComm *comm, hipStream_t stream) {
  HostGauge<F> hg;
  int X[4], R[4];
  for (int d = 0; d < 4; d++) {
    X[d] = g->X[d];
    R[d] = g->R[d];
    hg.XE[d] = X[d] + 2 * R[d];
  }
  hg.stride = g->stride;
  hg.parity_offset = g->parity_offset;
  hg.buf.assign((size_t)4 * hg.parity_offset, F(0));
  const int64_t volCB = (int64_t)X[0] * X[1] * X[2] * X[3] / 2;
  // ---- interior: QDP host order -> native extended layout (copyExtendedGauge)
  for (int dir = 0; dir < 4; dir++) {
    const H *src = static_cast<const H *>(qdp[dir]);
    int c[4];
    for (c[3] = 0; c[3] < X[3]; c[3]++)
      for (c[2] = 0; c[2] < X[2]; c[2]++)
        for (c[1] = 0; c[1] < X[1]; c[1]++)
          for (c[0] = 0; c[0] < X[0]; c[0]++) {
            const int lex = ((c[3] * X[2] + c[2]) * X[1] + c[1]) * X[0] + c[0];
            const int pty = (c[0] + c[1] + c[2] + c[3]) & 1;
            const H *s = src + ((int64_t)pty * volCB + (lex >> 1)) * 18;
            int ce[4] = {c[0] + R[0], c[1] + R[1], c[2] + R[2], c[3] + R[3]};
            for (int comp = 0; comp < 9; comp++) {
              const int64_t i = hg.index(dir, ce, comp);
              hg.buf[2 * i] = (F)s[2 * comp];
              hg.buf[2 * i + 1] = (F)s[2 * comp + 1];
            }
          }
  }
  // ---- borders, one dimension after the other (exchangeExtendedGhost)
  for (int d = 0; d < 4; d++) {
    if (R[d] == 0) continue;
    size_t nSites = (size_t)R[d];
    for (int e = 0; e < 4; e++)
      if (e != d) nSites *= hg.XE[e];
    const size_t n = nSites * 36 * 2;  // reals per slab
    std::vector<F> sendLow(n), sendHigh(n), recvLow(n), recvHigh(n);
    auto pack = [&](std::vector<F> &out, int lo) {
      size_t k = 0;
      for_slab(hg.XE, d, lo, R[d], [&](const int c[4]) {
        for (int dir = 0; dir < 4; dir++)
          for (int comp = 0; comp < 9; comp++) {
            const int64_t i = hg.index(dir, c, comp);
            out[k++] = hg.buf[2 * i];
            out[k++] = hg.buf[2 * i + 1];
          }
      });
    };
    auto unpack = [&](const std::vector<F> &in, int lo) {
      size_t k = 0;
      for_slab(hg.XE, d, lo, R[d], [&](const int c[4]) {
        for (int dir = 0; dir < 4; dir++)
          for (int comp = 0; comp < 9; comp++) {
            const int64_t i = hg.index(dir, c, comp);
            hg.buf[2 * i] = in[k++];
            hg.buf[2 * i + 1] = in[k++];
          }
      });
    };
    pack(sendLow, R[d]);          // first R interior layers  -> backward neighbour's HIGH border
    pack(sendHigh, X[d]);         // last  R interior layers  -> forward  neighbour's LOW border
    const bool part = comm_partitioned(comm, d);
    if (part) {
      void *sd = nullptr, *rd = nullptr;
      const size_t bytes = n * sizeof(F);
      MUGIQ_CHECK_HIP(hipMalloc(&sd, bytes));
      MUGIQ_CHECK_HIP(hipMalloc(&rd, bytes));
      int st = 0;
      for (int pass = 0; pass < 2 && !st; pass++) {
        const std::vector<F> &snd = pass == 0 ? sendLow : sendHigh;
        std::vector<F> &rcv = pass == 0 ? recvHigh : recvLow;  // what arrives from the forward nbr fills my HIGH border
        if (hipMemcpyAsync(sd, snd.data(), bytes, hipMemcpyHostToDevice, stream) != hipSuccess ||
            hipStreamSynchronize(stream) != hipSuccess) {
          st = set_error(MUGIQ_HIP_ERROR_HIP, "createExtendedCudaGaugeField: staging copy failed");
          break;
        }
        st = comm->sendrecv(comm->ctx, sd, rd, bytes, d, pass == 0 ? -1 : +1, stream);
        if (st) {
          st = set_error(MUGIQ_HIP_ERROR_HIP, "createExtendedCudaGaugeField: halo sendrecv callback failed with status %d", st);
          break;
        }
        if (hipStreamSynchronize(stream) != hipSuccess ||
            hipMemcpy(rcv.data(), rd, bytes, hipMemcpyDeviceToHost) != hipSuccess)
          st = set_error(MUGIQ_HIP_ERROR_HIP, "createExtendedCudaGaugeField: staging copy failed");
      }
      (void)hipFree(sd);
      (void)hipFree(rd);
      if (st) return st;
    } else {  // periodic wrap inside the domain
      recvHigh = sendLow;
      recvLow = sendHigh;
    }
    unpack(recvHigh, R[d] + X[d]);
    unpack(recvLow, 0);
  }
  MUGIQ_CHECK_HIP(hipMemcpyAsync(g->data, hg.buf.data(), hg.buf.size() * sizeof(F), hipMemcpyHostToDevice, stream));
  MUGIQ_CHECK_HIP(hipStreamSynchronize(stream));
  return MUGIQ_HIP_SUCCESS;
}

}  // namespace mugiq

using namespace mugiq;

extern "C" {

size_t mugiq_hip_extended_gauge_bytes(const int X[4], const int R[4], int precision) {
  if (!X || !R || (precision != 4 && precision != 8)) return 0;
  size_t vol = 1;
  for (int d = 0; d < 4; d++) vol *= (size_t)(X[d] + 2 * R[d]);
  return vol / 2 * 36 * 2 * 2 * (size_t)precision;  // volExCB * 36 planes * 2 parities * complex
}

// the allocation half of Displace::createExtendedCudaGaugeField (lib/displace.cpp:104-124: gParamEx from X + 2 r, pad 0,
// QUDA_ZERO_FIELD_CREATE), for hosts that do not manage device memory themselves
int mugiq_hip_alloc_extended_gauge(MugiqHipGaugeField *gauge, const int X[4], const int R[4], int precision) {
  const char *who = "createExtendedCudaGaugeField";
  MUGIQ_REQUIRE(gauge && X && R, "%s: NULL argument", who);
  MUGIQ_REQUIRE(precision == 4 || precision == 8, "%s: precision %d", who, precision);
  long long volEx = 1;
  for (int d = 0; d < 4; d++) {
    MUGIQ_REQUIRE(X[d] > 0 && (X[d] & 1) == 0 && R[d] >= 0, "%s: X[%d] = %d must be positive and even, R[%d] = %d non-negative", who, d, X[d], d, R[d]);
    gauge->X[d] = X[d];
    gauge->R[d] = R[d];
    volEx *= X[d] + 2 * R[d];
  }
  MUGIQ_REQUIRE(volEx / 2 < (1LL << 31), "%s: extended volume overflows int", who);
  gauge->precision = precision;
  gauge->stride = (int)(volEx / 2);
  gauge->parity_offset = (int64_t)36 * gauge->stride;
  gauge->data = nullptr;
  const size_t bytes = mugiq_hip_extended_gauge_bytes(X, R, precision);
  MUGIQ_CHECK_HIP(hipMalloc(&gauge->data, bytes));
  MUGIQ_CHECK_HIP(hipMemset(gauge->data, 0, bytes));
  return MUGIQ_HIP_SUCCESS;
}

int mugiq_hip_free_extended_gauge(MugiqHipGaugeField *gauge) {
  if (gauge && gauge->data) {
    MUGIQ_CHECK_HIP(hipFree(gauge->data));
    gauge->data = nullptr;
  }
  return MUGIQ_HIP_SUCCESS;
}

int mugiq_hip_create_extended_gauge(const MugiqHipGaugeField *gauge, const void *const qdpLinks_h[4], int cpuPrecision,
                                    const MugiqHipComm *comm, void *stream) {
  const char *who = "createExtendedCudaGaugeField";
  MUGIQ_REQUIRE(gauge && gauge->data && qdpLinks_h, "%s: NULL argument", who);
  MUGIQ_REQUIRE((gauge->precision == 4 || gauge->precision == 8) && (cpuPrecision == 4 || cpuPrecision == 8),
                "%s: precisions must be 4 or 8", who);
  long long volEx = 1;
  int sumR = 0;
  bool anyPart = false;
  for (int d = 0; d < 4; d++) {
    MUGIQ_REQUIRE(qdpLinks_h[d] != nullptr, "%s: host link pointer %d is NULL", who, d);
    MUGIQ_REQUIRE(gauge->X[d] > 0 && (gauge->X[d] & 1) == 0 && gauge->R[d] >= 0 && gauge->R[d] <= gauge->X[d],
                  "%s: invalid X[%d] = %d / R[%d] = %d", who, d, gauge->X[d], d, gauge->R[d]);
    volEx *= gauge->X[d] + 2 * gauge->R[d];
    sumR += gauge->R[d];
    if (comm_partitioned(comm, d)) {
      anyPart = true;
      MUGIQ_REQUIRE(gauge->R[d] >= 1, "%s: dim %d is partitioned but R[%d] = 0", who, d, d);
    }
  }
  MUGIQ_REQUIRE((sumR & 1) == 0, "%s: the sum of the borders R must be even", who);
  MUGIQ_REQUIRE(gauge->stride >= volEx / 2 && gauge->parity_offset >= (int64_t)36 * gauge->stride,
                "%s: stride / parity_offset too small for the extended volume", who);
  if (anyPart) MUGIQ_REQUIRE(comm->sendrecv != nullptr, "%s: comm->sendrecv is NULL", who);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (gauge->precision == 8 && cpuPrecision == 8) return fill_extended<double, double>(gauge, qdpLinks_h, comm, s);
  if (gauge->precision == 8 && cpuPrecision == 4) return fill_extended<double, float>(gauge, qdpLinks_h, comm, s);
  if (gauge->precision == 4 && cpuPrecision == 8) return fill_extended<float, double>(gauge, qdpLinks_h, comm, s);
  return fill_extended<float, float>(gauge, qdpLinks_h, comm, s);
}

}  // extern "C"
