// Host-side plumbing of libmugiq_hip.so: error reporting, gamma tables, descriptor validation.
#include <cstring>
#include <mutex>
#include <string>

#include "internal.h"

namespace mugiq {

static thread_local std::string g_last_error;

int set_error(int status, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  g_last_error = buf;
  return status;
}

int device_scratch(void **ptr, size_t bytes) {
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
    size_t want = bytes < (1u << 16) ? (1u << 16) : bytes * 2;
    MUGIQ_CHECK_HIP(hipMalloc(&buf[dev], want));
    cap[dev] = want;
  }
  *ptr = buf[dev];
  return MUGIQ_HIP_SUCCESS;
}

int upload_table(void **dev, const void *host, size_t bytes, hipStream_t stream) {
  static std::mutex mtx;
  static void *pinned[16] = {nullptr};
  static size_t cap[16] = {0};
  static hipEvent_t done[16] = {nullptr};
  int st = device_scratch(dev, bytes);
  if (st) return st;
  std::lock_guard<std::mutex> lock(mtx);
  int d = 0;
  MUGIQ_CHECK_HIP(hipGetDevice(&d));
  if (!done[d]) MUGIQ_CHECK_HIP(hipEventCreateWithFlags(&done[d], hipEventDisableTiming));
  else MUGIQ_CHECK_HIP(hipEventSynchronize(done[d]));  // previous table has left the staging buffer
  if (bytes > cap[d]) {
    if (pinned[d]) MUGIQ_CHECK_HIP(hipHostFree(pinned[d]));
    pinned[d] = nullptr;
    size_t want = bytes < (1u << 16) ? (1u << 16) : bytes * 2;
    MUGIQ_CHECK_HIP(hipHostMalloc(&pinned[d], want, hipHostMallocDefault));
    cap[d] = want;
  }
  memcpy(pinned[d], host, bytes);
  MUGIQ_CHECK_HIP(hipMemcpyAsync(*dev, pinned[d], bytes, hipMemcpyHostToDevice, stream));
  MUGIQ_CHECK_HIP(hipEventRecord(done[d], stream));
  return MUGIQ_HIP_SUCCESS;
}

int validate_spinor(const MugiqHipSpinorField *f, const char *who, const char *name) {
  MUGIQ_REQUIRE(f != nullptr, "%s: %s is NULL", who, name);
  MUGIQ_REQUIRE(f->data != nullptr, "%s: %s->data is NULL", who, name);
  MUGIQ_REQUIRE(f->precision == 4 || f->precision == 8, "%s: %s->precision = %d (must be 4 or 8)", who, name,
                f->precision);
  MUGIQ_REQUIRE(f->field_order == 2 || f->field_order == 4, "%s: %s->field_order = %d (must be 2 or 4)", who,
                name, f->field_order);
  // lib/contract_wrappers.cu:100,185
  MUGIQ_REQUIRE(f->nParity == 2, "%s: Loop contraction kernels support only Full Site Subset spinors! (%s->nParity = %d)",
                who, name, f->nParity);
  long long vol = 1;
  for (int d = 0; d < 4; d++) {
    MUGIQ_REQUIRE(f->X[d] > 0 && (f->X[d] & 1) == 0, "%s: %s->X[%d] = %d must be positive and even", who, name, d,
                  f->X[d]);
    vol *= f->X[d];
  }
  MUGIQ_REQUIRE(vol / 2 == f->volumeCB, "%s: %s->volumeCB = %d does not match X (%lld)", who, name, f->volumeCB,
                vol / 2);
  MUGIQ_REQUIRE(f->stride >= f->volumeCB, "%s: %s->stride = %d < volumeCB = %d", who, name, f->stride, f->volumeCB);
  MUGIQ_REQUIRE(f->parity_offset >= (int64_t)12 * f->stride, "%s: %s->parity_offset = %lld < 12*stride", who, name,
                (long long)f->parity_offset);
  return MUGIQ_HIP_SUCCESS;
}

bool same_geometry(const MugiqHipSpinorField &a, const MugiqHipSpinorField &b) {
  return a.precision == b.precision && a.field_order == b.field_order && a.nParity == b.nParity &&
         a.volumeCB == b.volumeCB && a.stride == b.stride && a.parity_offset == b.parity_offset &&
         a.X[0] == b.X[0] && a.X[1] == b.X[1] && a.X[2] == b.X[2] && a.X[3] == b.X[3];
}

}  // namespace mugiq

using namespace mugiq;

extern "C" {

int mugiq_hip_version(void) { return MUGIQ_HIP_VERSION; }

const char *mugiq_hip_last_error(void) { return g_last_error.c_str(); }

int mugiq_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int mugiq_hip_copy_gamma_coeff_to_symbol(int precision) {
  MUGIQ_REQUIRE(precision == 4 || precision == 8, "copyGammaCoeffStructToSymbol: Precision not supported! (%d)",
                precision);
  return MUGIQ_HIP_SUCCESS;
}

int mugiq_hip_copy_gamma_map_to_symbol(int precision) {
  MUGIQ_REQUIRE(precision == 4 || precision == 8, "copyGammaMapStructToSymbol: Precision not supported! (%d)",
                precision);
  return MUGIQ_HIP_SUCCESS;
}

int mugiq_hip_get_gamma_tables(double *row_value_h, int *column_index_h, double *map_sign_h, int *map_index_h) {
  static const double re[4] = {1, 0, -1, 0}, im[4] = {0, 1, 0, -1};
  for (int m = 0; m < 16; m++) {
    for (int n = 0; n < 4; n++) {
      if (row_value_h) {
        row_value_h[(m * 4 + n) * 2 + 0] = re[kGammaPhase[m][n]];
        row_value_h[(m * 4 + n) * 2 + 1] = im[kGammaPhase[m][n]];
      }
      if (column_index_h) column_index_h[m * 4 + n] = kGammaColumn[m][n];
    }
    if (map_sign_h) map_sign_h[m] = kGammaMapSign[m];
    if (map_index_h) map_index_h[m] = 15 - m;
  }
  return MUGIQ_HIP_SUCCESS;
}

const char *mugiq_hip_gamma_name(int m) {
  // include/gamma.h:11-20
  static const char *names[16] = {"1",  "g1",   "g2",   "g1g2", "g3",   "g1g3", "g2g3", "g5g4",
                                  "g4", "g1g4", "g2g4", "g5g3", "g3g4", "g5g2", "g5g1", "g5"};
  return (m >= 0 && m < 16) ? names[m] : nullptr;
}

}  // extern "C"
