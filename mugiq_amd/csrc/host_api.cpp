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
  std::mutex mtx;  // serialises the users of THIS arena (two host threads must not drive one stream anyway)
  void *tab = nullptr, *ws = nullptr;
  size_t tabCap = 0, wsCap = 0;
  // two pinned staging buffers, used in turn: an upload waits for the upload TWO calls back to have left its buffer -- not for
  // the previous one, which sits on the stream behind the previous call's kernel and would make every call a host sync
  void *pinned[2] = {nullptr, nullptr};
  size_t pinnedCap[2] = {0, 0};
  hipEvent_t staged[2] = {nullptr, nullptr};
  int turn = 0;
};
std::mutex g_arenaMtx;  // guards the map only; no HIP call is made while it is held
std::map<std::pair<int, hipStream_t>, StreamArena> g_arenas;

StreamArena &arena_of(int dev, hipStream_t stream) {
  std::lock_guard<std::mutex> lock(g_arenaMtx);
  return g_arenas[{dev, stream}];  // (std::map: the reference stays valid while other arenas come and go)
}

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
  StreamArena &a = arena_of(dev, stream);
  std::lock_guard<std::mutex> lock(a.mtx);
  if ((st = arena_reserve(&a.tab, &a.tabCap, bytes, 1u << 16, stream))) return st;
  *ptr = a.tab;
  return MUGIQ_HIP_SUCCESS;
}

int stream_workspace(void **ptr, size_t bytes, hipStream_t stream) {
  int dev = 0, st = current_device(&dev);
  if (st) return st;
  StreamArena &a = arena_of(dev, stream);
  std::lock_guard<std::mutex> lock(a.mtx);
  if ((st = arena_reserve(&a.ws, &a.wsCap, bytes, 256, stream))) return st;
  *ptr = a.ws;
  return MUGIQ_HIP_SUCCESS;
}

int upload_table(void **dev_out, const void *host, size_t bytes, hipStream_t stream) {
  int dev = 0, st = current_device(&dev);
  if (st) return st;
  StreamArena &a = arena_of(dev, stream);
  std::lock_guard<std::mutex> lock(a.mtx);
  if ((st = arena_reserve(&a.tab, &a.tabCap, bytes, 1u << 16, stream))) return st;
  const int i = a.turn;
  a.turn ^= 1;
  if (!a.staged[i]) MUGIQ_CHECK_HIP(hipEventCreateWithFlags(&a.staged[i], hipEventDisableTiming));
  else MUGIQ_CHECK_HIP(hipEventSynchronize(a.staged[i]));  // the table of two calls back has left this staging buffer
  if (bytes > a.pinnedCap[i]) {
    if (a.pinned[i]) MUGIQ_CHECK_HIP(hipHostFree(a.pinned[i]));
    a.pinned[i] = nullptr;
    a.pinnedCap[i] = 0;
    const size_t want = bytes < (1u << 16) ? (1u << 16) : bytes * 2;
    MUGIQ_CHECK_HIP(hipHostMalloc(&a.pinned[i], want, hipHostMallocDefault));
    a.pinnedCap[i] = want;
  }
  memcpy(a.pinned[i], host, bytes);
  MUGIQ_CHECK_HIP(hipMemcpyAsync(a.tab, a.pinned[i], bytes, hipMemcpyHostToDevice, stream));
  MUGIQ_CHECK_HIP(hipEventRecord(a.staged[i], stream));
  *dev_out = a.tab;
  return MUGIQ_HIP_SUCCESS;
}

int release_stream_scratch(hipStream_t stream) {
  int dev = 0, st = current_device(&dev);
  if (st) return st;
  StreamArena *ap = nullptr;
  {
    std::lock_guard<std::mutex> lock(g_arenaMtx);
    auto it = g_arenas.find({dev, stream});
    if (it == g_arenas.end()) return MUGIQ_HIP_SUCCESS;
    ap = &it->second;
  }
  MUGIQ_CHECK_HIP(hipStreamSynchronize(stream));
  {
    StreamArena &a = *ap;
    std::lock_guard<std::mutex> lock(a.mtx);
    if (a.tab) (void)hipFree(a.tab);
    if (a.ws) (void)hipFree(a.ws);
    for (int i = 0; i < 2; i++) {
      if (a.pinned[i]) (void)hipHostFree(a.pinned[i]);
      if (a.staged[i]) (void)hipEventDestroy(a.staged[i]);
    }
  }
  std::lock_guard<std::mutex> lock(g_arenaMtx);
  g_arenas.erase({dev, stream});
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
