// Host-side plumbing of libmugiq_hip.so: error reporting, gamma tables, descriptor validation.
#include <cstring>
#include <map>
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

// ---- per-stream scratch ---------------------------------------------------------------------------------------------
// Every C-ABI entry takes the stream its kernels run on; the small tables those kernels read (eigenvector pointer lists,
// 1/sigma, momenta, phases) and the workspaces the library allocates for the caller live in an arena keyed by
// (device, stream).  Calls on ONE stream are ordered by the stream itself (the table upload of call n+1 is enqueued behind
// the kernel of call n), calls on different streams, threads or Loop objects never share a buffer.
namespace {
struct StreamArena {
  void *tab = nullptr, *ws = nullptr, *pinned = nullptr;
  size_t tabCap = 0, wsCap = 0, pinnedCap = 0;
  hipEvent_t staged = nullptr;  // the last table has left the pinned staging buffer
};
std::mutex g_arenaMtx;
std::map<std::pair<int, hipStream_t>, StreamArena> g_arenas;

// grow-only buffer of an arena; the old buffer may still be read by kernels of THIS stream only, so that stream is drained
int arena_reserve(void **buf, size_t *cap, size_t bytes, size_t floor, hipStream_t stream) {
  if (bytes <= *cap) return MUGIQ_HIP_SUCCESS;
  if (*buf) {
    MUGIQ_CHECK_HIP(hipStreamSynchronize(stream));
    MUGIQ_CHECK_HIP(hipFree(*buf));
    *buf = nullptr;
    *cap = 0;
  }
  const size_t want = bytes < floor ? floor : bytes + bytes / 2;
  MUGIQ_CHECK_HIP(hipMalloc(buf, want));
  *cap = want;
  return MUGIQ_HIP_SUCCESS;
}
int current_device(int *dev) {
  MUGIQ_CHECK_HIP(hipGetDevice(dev));
  return MUGIQ_HIP_SUCCESS;
}
}  // namespace

int stream_scratch(void **ptr, size_t bytes, hipStream_t stream) {
  int dev = 0, st = current_device(&dev);
  if (st) return st;
  std::lock_guard<std::mutex> lock(g_arenaMtx);
  StreamArena &a = g_arenas[{dev, stream}];
  if ((st = arena_reserve(&a.tab, &a.tabCap, bytes, 1u << 16, stream))) return st;
  *ptr = a.tab;
  return MUGIQ_HIP_SUCCESS;
}

int stream_workspace(void **ptr, size_t bytes, hipStream_t stream) {
  int dev = 0, st = current_device(&dev);
  if (st) return st;
  std::lock_guard<std::mutex> lock(g_arenaMtx);
  StreamArena &a = g_arenas[{dev, stream}];
  if ((st = arena_reserve(&a.ws, &a.wsCap, bytes, 256, stream))) return st;
  *ptr = a.ws;
  return MUGIQ_HIP_SUCCESS;
}

int upload_table(void **dev_out, const void *host, size_t bytes, hipStream_t stream) {
  int dev = 0, st = current_device(&dev);
  if (st) return st;
  std::lock_guard<std::mutex> lock(g_arenaMtx);
  StreamArena &a = g_arenas[{dev, stream}];
  if ((st = arena_reserve(&a.tab, &a.tabCap, bytes, 1u << 16, stream))) return st;
  if (!a.staged) MUGIQ_CHECK_HIP(hipEventCreateWithFlags(&a.staged, hipEventDisableTiming));
  else MUGIQ_CHECK_HIP(hipEventSynchronize(a.staged));  // previous table has left the staging buffer
  if (bytes > a.pinnedCap) {
    if (a.pinned) MUGIQ_CHECK_HIP(hipHostFree(a.pinned));
    a.pinned = nullptr;
    a.pinnedCap = 0;
    const size_t want = bytes < (1u << 16) ? (1u << 16) : bytes * 2;
    MUGIQ_CHECK_HIP(hipHostMalloc(&a.pinned, want, hipHostMallocDefault));
    a.pinnedCap = want;
  }
  memcpy(a.pinned, host, bytes);
  MUGIQ_CHECK_HIP(hipMemcpyAsync(a.tab, a.pinned, bytes, hipMemcpyHostToDevice, stream));
  MUGIQ_CHECK_HIP(hipEventRecord(a.staged, stream));
  *dev_out = a.tab;
  return MUGIQ_HIP_SUCCESS;
}

int release_stream_scratch(hipStream_t stream) {
  int dev = 0, st = current_device(&dev);
  if (st) return st;
  std::lock_guard<std::mutex> lock(g_arenaMtx);
  auto it = g_arenas.find({dev, stream});
  if (it == g_arenas.end()) return MUGIQ_HIP_SUCCESS;
  MUGIQ_CHECK_HIP(hipStreamSynchronize(stream));
  StreamArena &a = it->second;
  if (a.tab) (void)hipFree(a.tab);
  if (a.ws) (void)hipFree(a.ws);
  if (a.pinned) (void)hipHostFree(a.pinned);
  if (a.staged) (void)hipEventDestroy(a.staged);
  g_arenas.erase(it);
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

int mugiq_hip_release_stream(void *stream) { return release_stream_scratch(static_cast<hipStream_t>(stream)); }

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
