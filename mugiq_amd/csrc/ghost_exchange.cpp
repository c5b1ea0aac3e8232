// exchangeGhostVec (lib/contract_wrappers.cu:166-169 of the reference: x->exchangeGhost(QUDA_INVALID_PARITY, nFace = 1, 0)) as a
// C-ABI entry: depth-1 ghost zones of one spinor field on every partitioned dimension, both directions, through the
// caller's transport.  The BASIC plan of the driver does the same per step for the one face a displacement reads
// (loop_driver.cpp: exchange_face); this is the stand-alone operator for callers that sequence the steps themselves.
#include "internal.h"

extern "C" int mugiq_hip_exchange_ghost_vec(const MugiqHipSpinorField *v, const MugiqHipComm *comm, void *stream) {
  using namespace mugiq;
  const char *who = "mugiq_hip_exchange_ghost_vec";
  MUGIQ_REQUIRE(v != nullptr && v->data != nullptr, "%s: field is NULL", who);
  MUGIQ_REQUIRE(v->nParity == 2, "%s: This function supports only Full Site Subset fields!", who);  // lib/contract_wrappers.cu:185
  MUGIQ_REQUIRE(v->precision == 4 || v->precision == 8, "%s: precision %d", who, v->precision);
  if (comm == nullptr) return MUGIQ_HIP_SUCCESS;  // one process: nothing is partitioned
  MUGIQ_REQUIRE(comm->sendrecv != nullptr, "%s: comm->sendrecv is NULL", who);
  hipStream_t s = static_cast<hipStream_t>(stream);
  size_t total = 0, bytes[4] = {0, 0, 0, 0};
  for (int d = 0; d < 4; d++) {
    if (!comm_partitioned(comm, d)) continue;
    MUGIQ_REQUIRE(v->ghost[d][0] != nullptr && v->ghost[d][1] != nullptr, "%s: dimension %d is partitioned but the field has no ghost zones for it", who, d);
    bytes[d] = (size_t)24 * (size_t)(v->volumeCB / v->X[d]) * 2 * (size_t)v->precision;
    total += 2 * bytes[d];
  }
  if (total == 0) return MUGIQ_HIP_SUCCESS;
  void *ws = nullptr;
  int st = stream_workspace(&ws, total, s);  // one send buffer per message: nothing is overwritten before it left
  if (st) return st;
  unsigned char *send = static_cast<unsigned char *>(ws);
  const bool grouped = comm->group_begin && comm->group_end;
  if (grouped && (st = comm->group_begin(comm->ctx))) return set_error(MUGIQ_HIP_ERROR_HIP, "%s: group_begin callback failed with status %d", who, st);
  for (int d = 0; d < 4; d++) {
    if (!comm_partitioned(comm, d)) continue;
    for (int high = 0; high < 2; high++) {
      // my low face is the backward neighbour's ghost[d][1] (forward zone); my high face the forward neighbour's ghost[d][0]
      if ((st = mugiq_hip_pack_face(send, v, d, high, stream))) return st;
      st = comm->sendrecv(comm->ctx, send, v->ghost[d][1 - high], bytes[d], d, high ? +1 : -1, stream);
      if (st) return set_error(MUGIQ_HIP_ERROR_HIP, "%s: halo sendrecv callback failed with status %d", who, st);
      send += bytes[d];
    }
  }
  if (grouped && (st = comm->group_end(comm->ctx, stream))) return set_error(MUGIQ_HIP_ERROR_HIP, "%s: group_end callback failed with status %d", who, st);
  return MUGIQ_HIP_SUCCESS;
}

// ---- what Displace asks of QUDA's ColorSpinorField for its auxiliary vector (lib/displace.cpp:26-30,40-60): Create with
// QUDA_ZERO_FIELD_CREATE, operator=, blas::zero -- for hosts that do not manage device memory themselves.
extern "C" int mugiq_hip_free_spinor(MugiqHipSpinorField *f);

extern "C" int mugiq_hip_alloc_spinor_like(MugiqHipSpinorField *out, const MugiqHipSpinorField *like, int precision, const int ghostDims[4]) {
  using namespace mugiq;
  const char *who = "mugiq_hip_alloc_spinor_like";
  MUGIQ_REQUIRE(out != nullptr && like != nullptr, "%s: NULL descriptor", who);
  MUGIQ_REQUIRE(precision == 0 || precision == 4 || precision == 8, "%s: precision %d", who, precision);
  *out = *like;
  out->precision = precision ? precision : like->precision;
  out->data = nullptr;
  for (int d = 0; d < 4; d++) out->ghost[d][0] = out->ghost[d][1] = nullptr;
  const size_t cb = 2 * (size_t)out->precision;
  const size_t bytes = (size_t)2 * (size_t)out->parity_offset * cb;
  // (on any failure everything allocated so far is released again: the caller gets either a complete field or nothing)
  auto fail = [&](hipError_t e, const char *what) {
    (void)mugiq_hip_free_spinor(out);
    return set_error(MUGIQ_HIP_ERROR_HIP, "%s: %s failed: %s", who, what, hipGetErrorString(e));
  };
  hipError_t e = hipMalloc(&out->data, bytes);
  if (e != hipSuccess) {
    out->data = nullptr;
    return fail(e, "hipMalloc");
  }
  if ((e = hipMemset(out->data, 0, bytes)) != hipSuccess) return fail(e, "hipMemset");
  for (int d = 0; d < 4 && ghostDims; d++) {
    if (!ghostDims[d]) continue;
    const size_t gb = (size_t)24 * (size_t)(out->volumeCB / out->X[d]) * cb;
    for (int b = 0; b < 2; b++) {
      if ((e = hipMalloc(&out->ghost[d][b], gb)) != hipSuccess) {
        out->ghost[d][b] = nullptr;
        return fail(e, "hipMalloc");
      }
      if ((e = hipMemset(out->ghost[d][b], 0, gb)) != hipSuccess) return fail(e, "hipMemset");
    }
  }
  return MUGIQ_HIP_SUCCESS;
}

extern "C" int mugiq_hip_free_spinor(MugiqHipSpinorField *f) {
  using namespace mugiq;
  if (!f) return MUGIQ_HIP_SUCCESS;
  if (f->data) MUGIQ_CHECK_HIP(hipFree(f->data));
  f->data = nullptr;
  for (int d = 0; d < 4; d++)
    for (int b = 0; b < 2; b++) {
      if (f->ghost[d][b]) MUGIQ_CHECK_HIP(hipFree(f->ghost[d][b]));
      f->ghost[d][b] = nullptr;
    }
  return MUGIQ_HIP_SUCCESS;
}

// *dst = *src (same geometry, order and precision: a device copy of the body; ghost zones are not part of the value)
extern "C" int mugiq_hip_copy_spinor(const MugiqHipSpinorField *dst, const MugiqHipSpinorField *src, void *stream) {
  using namespace mugiq;
  const char *who = "mugiq_hip_copy_spinor";
  int st = validate_spinor(dst, who, "dst");
  if (st) return st;
  if ((st = validate_spinor(src, who, "src"))) return st;
  MUGIQ_REQUIRE(dst->precision == src->precision && dst->field_order == src->field_order && dst->volumeCB == src->volumeCB &&
                    dst->stride == src->stride && dst->parity_offset == src->parity_offset,
                "%s: fields differ in precision, order or geometry", who);
  const size_t bytes = (size_t)2 * (size_t)src->parity_offset * 2 * (size_t)src->precision;
  MUGIQ_CHECK_HIP(hipMemcpyAsync(dst->data, src->data, bytes, hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
  return MUGIQ_HIP_SUCCESS;
}

// blas::zero(*f)
extern "C" int mugiq_hip_zero_spinor(const MugiqHipSpinorField *f, void *stream) {
  using namespace mugiq;
  int st = validate_spinor(f, "mugiq_hip_zero_spinor", "f");
  if (st) return st;
  const size_t bytes = (size_t)2 * (size_t)f->parity_offset * 2 * (size_t)f->precision;
  MUGIQ_CHECK_HIP(hipMemsetAsync(f->data, 0, bytes, static_cast<hipStream_t>(stream)));
  return MUGIQ_HIP_SUCCESS;
}
