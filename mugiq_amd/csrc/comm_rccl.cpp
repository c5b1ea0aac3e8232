// A transport inside the library: MugiqHipComm filled from RCCL.
//
// The driver (csrc/loop_driver.cpp) sees its transport through the callback table MugiqHipComm: the nearest-neighbour face
// exchange of ColorSpinorField::exchangeGhost (lib/contract_wrappers.cu:166-169) and the MPI_Reduce / MPI_Gather / MPI_Bcast of
// Loop_Mugiq::performMomentumProjection over COMM_SPACE / COMM_TIME (lib/loop_mugiq.cpp:61-88, 406-424).  mugiq_amd's Python host
// serves the table from torch.distributed; a C++ host (a MuGiq build) had MPI or nothing.  Here the table is served by RCCL
// itself, with no host language in the data path:
//   sendrecv      ncclSend + ncclRecv on the caller's stream inside one ncclGroupStart/End (a self-neighbour -- an axis of extent 1
//                 under forced partitioning -- is a send to self inside the same group)
//   group_*       ncclGroupStart ... ncclGroupEnd around the sendrecv calls of one halo block: the messages to different
//                 neighbours leave on different xGMI links at once
//   reduce_space  ncclReduce(sum) over the ranks sharing coord[3] (ncclCommSplit colour coord[3], key = (x, y, z) lexicographic:
//                 the rank with x = y = z = 0 is rank 0 of its group) on device staging buffers
//   gather_time   ncclAllGather over the ranks with x = y = z = 0 (colour 0 for them, NCCL_SPLIT_NOCOLOR for the rest; key coord[3])
//   bcast         ncclBroadcast from world rank 0
// librccl is loaded on first use (dlopen; an RCCL the process has loaded already -- torch ships one -- is reused), so the
// library itself keeps no link dependency on it.  Rank <-> coordinate map: QUDA's default, t fastest (as mugiq_amd/comm.py).
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <vector>

#include "internal.h"

namespace {

// the slice of rccl.h this file uses (types by value / opaque pointer; the ABI is NCCL's)
typedef struct { char internal[128]; } UniqueId;
typedef void *Comm;
enum { kNcclSuccess = 0 };
enum { kNcclInt8 = 0, kNcclFloat32 = 7, kNcclFloat64 = 8 };  // ncclDataType_t
enum { kNcclSum = 0 };                                       // ncclRedOp_t
constexpr int kSplitNoColor = -1;

struct Api {
  void *lib = nullptr;
  int (*GetUniqueId)(UniqueId *) = nullptr;
  int (*CommInitRank)(Comm *, int, UniqueId, int) = nullptr;
  int (*CommSplit)(Comm, int, int, Comm *, void *) = nullptr;
  int (*CommDestroy)(Comm) = nullptr;
  int (*CommCount)(Comm, int *) = nullptr;
  int (*CommUserRank)(Comm, int *) = nullptr;
  const char *(*GetErrorString)(int) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*Send)(const void *, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Recv)(void *, size_t, int, int, Comm, hipStream_t) = nullptr;
  int (*Reduce)(const void *, void *, size_t, int, int, int, Comm, hipStream_t) = nullptr;
  int (*AllGather)(const void *, void *, size_t, int, Comm, hipStream_t) = nullptr;
  int (*Broadcast)(const void *, void *, size_t, int, int, Comm, hipStream_t) = nullptr;
};
Api g_api;

int load_api() {
  if (g_api.lib) return MUGIQ_HIP_SUCCESS;
  void *h = nullptr;
  for (const char *name : {"librccl.so.1", "librccl.so"}) {  // an RCCL that is in the process already, else the system's
    if ((h = dlopen(name, RTLD_NOW | RTLD_NOLOAD))) break;
  }
  if (!h)
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
      if ((h = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) break;
  if (!h) return mugiq::set_error(MUGIQ_HIP_ERROR_UNSUPPORTED, "mugiq_hip_rccl: librccl.so could not be loaded (%s)", dlerror());
#define MUGIQ_SYM(member, name)                                                                       \
  g_api.member = reinterpret_cast<decltype(g_api.member)>(dlsym(h, name));                            \
  if (!g_api.member) return mugiq::set_error(MUGIQ_HIP_ERROR_UNSUPPORTED, "mugiq_hip_rccl: %s is missing from librccl", name);
  MUGIQ_SYM(GetUniqueId, "ncclGetUniqueId")
  MUGIQ_SYM(CommInitRank, "ncclCommInitRank")
  MUGIQ_SYM(CommSplit, "ncclCommSplit")
  MUGIQ_SYM(CommDestroy, "ncclCommDestroy")
  MUGIQ_SYM(CommCount, "ncclCommCount")
  MUGIQ_SYM(CommUserRank, "ncclCommUserRank")
  MUGIQ_SYM(GetErrorString, "ncclGetErrorString")
  MUGIQ_SYM(GroupStart, "ncclGroupStart")
  MUGIQ_SYM(GroupEnd, "ncclGroupEnd")
  MUGIQ_SYM(Send, "ncclSend")
  MUGIQ_SYM(Recv, "ncclRecv")
  MUGIQ_SYM(Reduce, "ncclReduce")
  MUGIQ_SYM(AllGather, "ncclAllGather")
  MUGIQ_SYM(Broadcast, "ncclBroadcast")
#undef MUGIQ_SYM
  g_api.lib = h;
  return MUGIQ_HIP_SUCCESS;
}

#define MUGIQ_CHECK_NCCL(call)                                                                                        \
  do {                                                                                                                \
    int r_ = (call);                                                                                                  \
    if (r_ != kNcclSuccess)                                                                                           \
      return mugiq::set_error(MUGIQ_HIP_ERROR_HIP, "%s:%d: %s failed: %s", __FILE__, __LINE__, #call, g_api.GetErrorString(r_)); \
  } while (0)

}  // namespace

struct MugiqHipRcclComm_s {
  Comm world = nullptr, space = nullptr, time = nullptr;
  bool ownsWorld = false;
  int rank = 0, size = 1;
  int grid[4] = {1, 1, 1, 1}, coord[4] = {0, 0, 0, 0}, partitioned[4] = {0, 0, 0, 0};
  bool isTimeProcess = true;
  int groupDepth = 0;
  // multi-path halos (see relay_plan): the messages of a transfer group are collected and issued at group_end in two phases
  bool multipath = false;
  struct Msg {
    const void *send;
    void *recv;
    size_t bytes;
    int dim, dir;
    hipStream_t stream;
  };
  std::vector<Msg> pending;
  void *bounce = nullptr;
  size_t bounceBytes = 0;
  hipStream_t hostStream = nullptr;  // the reductions of host payloads run here
  void *stage[2] = {nullptr, nullptr};
  size_t stageBytes[2] = {0, 0};

  int rank_of(const int c[4]) const { return ((c[0] * grid[1] + c[1]) * grid[2] + c[2]) * grid[3] + c[3]; }
  int neighbour(int dim, int dir) const {
    int c[4] = {coord[0], coord[1], coord[2], coord[3]};
    c[dim] = (c[dim] + dir + grid[dim]) % grid[dim];
    return rank_of(c);
  }
  int staging(int which, size_t bytes, void **p) {
    if (bytes > stageBytes[which]) {
      if (stage[which]) (void)hipFree(stage[which]);
      stage[which] = nullptr;
      stageBytes[which] = 0;
      MUGIQ_CHECK_HIP(hipMalloc(&stage[which], bytes));
      stageBytes[which] = bytes;
    }
    *p = stage[which];
    return MUGIQ_HIP_SUCCESS;
  }
};

namespace {

// ---- multi-path halos ------------------------------------------------------------------------------------------------------
// A halo message goes to ONE neighbour, i.e. over ONE of a GPU's seven xGMI links, while the links to the GPUs that are no
// neighbour on that axis idle.  With `multipath` a message of B bytes is cut into 1 + R parts (R = the ranks that are neither its
// origin nor its destination, at most 6): part 0 travels directly, part p goes to relay p in a first phase and from there to
// the destination in a second one; every rank is origin, destination and relay at once, the schedule is a pure function of
// (rank, grid, dim, dir, B), and each phase is one ncclGroup in which every send has its receive on the peer.  Between a pair of
// ranks a message causes at most one transfer per phase (a rank is either the destination or a relay of an origin, never both),
// so the order in which NCCL pairs sends and receives of a peer is the order of the messages, the same on every rank.
enum { kOpSendUser = 0, kOpRecvUser = 1, kOpRecvBounce = 2, kOpSendBounce = 3 };
struct RelayOp {
  int phase, kind, peer;
  size_t offset, len;  // offset into the user's send / receive buffer, or into this message's bounce area
};
int rank_of(const int grid[4], const int c[4]) { return ((c[0] * grid[1] + c[1]) * grid[2] + c[2]) * grid[3] + c[3]; }
void coords_of(const int grid[4], int r, int c[4]) {
  c[3] = r % grid[3];
  r /= grid[3];
  c[2] = r % grid[2];
  r /= grid[2];
  c[1] = r % grid[1];
  c[0] = r / grid[1];
}
int shifted(const int grid[4], int r, int dim, int dir) {
  int c[4];
  coords_of(grid, r, c);
  c[dim] = (c[dim] + dir + grid[dim]) % grid[dim];
  return rank_of(grid, c);
}
// the relays of origin o, in the order of their parts: every rank but o and its destination (at most kMaxRelays of them, the
// ones closest after o in rank order, so that different origins spread over different relays)
constexpr int kMaxRelays = 6;
void relays_of(const int grid[4], int size, int o, int dim, int dir, std::vector<int> &out) {
  out.clear();
  const int f = shifted(grid, o, dim, dir);
  for (int i = 1; i < size && (int)out.size() < kMaxRelays; i++) {
    const int r = (o + i) % size;
    if (r != o && r != f) out.push_back(r);
  }
}
size_t relay_chunk(size_t bytes, int nRelays) {
  const size_t c = (bytes + (size_t)nRelays) / ((size_t)nRelays + 1);
  return (c + 255) / 256 * 256;
}
void relay_plan(int rank, const int grid[4], int dim, int dir, size_t bytes, std::vector<RelayOp> &ops, size_t *bounceBytes) {
  ops.clear();
  const int size = grid[0] * grid[1] * grid[2] * grid[3];
  const int dst = shifted(grid, rank, dim, dir), src = shifted(grid, rank, dim, -dir);
  std::vector<int> rel;
  relays_of(grid, size, rank, dim, dir, rel);
  const int R = dst == rank ? 0 : (int)rel.size();  // (a self-neighbour gets everything "directly")
  const size_t chunk = relay_chunk(bytes, R);
  auto part = [&](int p, size_t &off, size_t &len) {
    off = std::min(bytes, (size_t)p * chunk);
    len = std::min(bytes, (size_t)(p + 1) * chunk) - off;
  };
  size_t off, len;
  part(0, off, len);
  ops.push_back({1, kOpSendUser, dst, off, len});
  ops.push_back({1, kOpRecvUser, src, off, len});
  for (int p = 1; p <= R; p++) {  // my own parts to their relays
    part(p, off, len);
    if (len) ops.push_back({1, kOpSendUser, rel[p - 1], off, len});
  }
  size_t slot = 0;
  if (R > 0)
    for (int o = 0; o < size; o++) {  // the parts I relay: origin o, I am entry q - 1 of its list
      if (o == rank) continue;
      std::vector<int> ro;
      relays_of(grid, size, o, dim, dir, ro);
      for (int q = 1; q <= (int)ro.size(); q++)
        if (ro[q - 1] == rank) {
          part(q, off, len);
          if (len) {
            ops.push_back({1, kOpRecvBounce, o, slot * chunk, len});
            ops.push_back({2, kOpSendBounce, shifted(grid, o, dim, dir), slot * chunk, len});
          }
          slot++;
        }
    }
  if (R > 0) {  // the relayed parts of the message meant for me (origin: src)
    std::vector<int> rs;
    relays_of(grid, size, src, dim, dir, rs);
    for (int p = 1; p <= (int)rs.size(); p++) {
      part(p, off, len);
      if (len) ops.push_back({2, kOpRecvUser, rs[p - 1], off, len});
    }
  }
  if (bounceBytes) *bounceBytes = slot * chunk;
}

// the collected messages of a group over all paths: phase 1 (direct parts, first hops), then phase 2 (second hops), one ncclGroup each
int flush_multipath(MugiqHipRcclComm *c) {
  std::vector<std::vector<RelayOp>> plans(c->pending.size());
  std::vector<size_t> bounceOff(c->pending.size());
  size_t total = 0;
  for (size_t i = 0; i < c->pending.size(); i++) {
    size_t b = 0;
    relay_plan(c->rank, c->grid, c->pending[i].dim, c->pending[i].dir, c->pending[i].bytes, plans[i], &b);
    bounceOff[i] = total;
    total += b;
  }
  if (total > c->bounceBytes) {
    if (c->bounce) {  // (messages of an earlier group may still be reading it on their stream)
      for (auto &m : c->pending) MUGIQ_CHECK_HIP(hipStreamSynchronize(m.stream));
      (void)hipFree(c->bounce);
    }
    c->bounce = nullptr;
    c->bounceBytes = 0;
    MUGIQ_CHECK_HIP(hipMalloc(&c->bounce, total));
    c->bounceBytes = total;
  }
  for (int phase = 1; phase <= 2; phase++) {
    MUGIQ_CHECK_NCCL(g_api.GroupStart());
    int r = kNcclSuccess;
    for (size_t i = 0; i < c->pending.size() && r == kNcclSuccess; i++) {
      const auto &m = c->pending[i];
      char *bnc = static_cast<char *>(c->bounce) + bounceOff[i];
      for (const RelayOp &op : plans[i]) {
        if (op.phase != phase || r != kNcclSuccess) continue;
        if (op.kind == kOpSendUser) r = g_api.Send(static_cast<const char *>(m.send) + op.offset, op.len, kNcclInt8, op.peer, c->world, m.stream);
        else if (op.kind == kOpRecvUser) r = g_api.Recv(static_cast<char *>(m.recv) + op.offset, op.len, kNcclInt8, op.peer, c->world, m.stream);
        else if (op.kind == kOpRecvBounce) r = g_api.Recv(bnc + op.offset, op.len, kNcclInt8, op.peer, c->world, m.stream);
        else r = g_api.Send(bnc + op.offset, op.len, kNcclInt8, op.peer, c->world, m.stream);
      }
    }
    const int r2 = g_api.GroupEnd();
    if (r == kNcclSuccess) r = r2;
    if (r != kNcclSuccess) {
      c->pending.clear();
      return mugiq::set_error(MUGIQ_HIP_ERROR_HIP, "mugiq_hip_rccl: multi-path halo, phase %d: %s", phase, g_api.GetErrorString(r));
    }
  }
  c->pending.clear();
  return MUGIQ_HIP_SUCCESS;
}

int cb_group_begin(void *ctx) {
  auto *c = static_cast<MugiqHipRcclComm *>(ctx);
  if (c->groupDepth++ == 0 && !c->multipath) MUGIQ_CHECK_NCCL(g_api.GroupStart());
  return MUGIQ_HIP_SUCCESS;
}
int cb_group_end(void *ctx, void * /*stream: the members of the group carry their own*/) {
  auto *c = static_cast<MugiqHipRcclComm *>(ctx);
  if (c->groupDepth <= 0) return mugiq::set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "mugiq_hip_rccl: group_end without group_begin");
  if (--c->groupDepth == 0) {
    if (c->multipath) return flush_multipath(c);
    MUGIQ_CHECK_NCCL(g_api.GroupEnd());
  }
  return MUGIQ_HIP_SUCCESS;
}
int cb_sendrecv(void *ctx, const void *send_d, void *recv_d, size_t bytes, int dim, int dir, void *stream) {
  auto *c = static_cast<MugiqHipRcclComm *>(ctx);
  if (dim < 0 || dim > 3 || (dir != 1 && dir != -1)) return mugiq::set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "mugiq_hip_rccl: sendrecv dim %d dir %d", dim, dir);
  const int dst = c->neighbour(dim, dir), src = c->neighbour(dim, -dir);
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (c->multipath) {  // collected; issued over all paths when the group closes (a lone message is a group of one)
    c->pending.push_back({send_d, recv_d, bytes, dim, dir, s});
    return c->groupDepth > 0 ? MUGIQ_HIP_SUCCESS : flush_multipath(c);
  }
  const bool open = c->groupDepth > 0;
  if (!open) MUGIQ_CHECK_NCCL(g_api.GroupStart());
  int r = g_api.Send(send_d, bytes, kNcclInt8, dst, c->world, s);
  if (r == kNcclSuccess) r = g_api.Recv(recv_d, bytes, kNcclInt8, src, c->world, s);
  if (!open) {
    const int r2 = g_api.GroupEnd();
    if (r == kNcclSuccess) r = r2;
  }
  if (r != kNcclSuccess) return mugiq::set_error(MUGIQ_HIP_ERROR_HIP, "mugiq_hip_rccl: ncclSend / ncclRecv failed: %s", g_api.GetErrorString(r));
  return MUGIQ_HIP_SUCCESS;
}
int nccl_type(int precision) { return precision == 8 ? kNcclFloat64 : kNcclFloat32; }

int cb_reduce_space(void *ctx, const void *send_h, void *recv_h, size_t n, int precision) {
  auto *c = static_cast<MugiqHipRcclComm *>(ctx);
  const size_t bytes = n * (size_t)precision;
  void *s = nullptr, *r = nullptr;
  int st;
  if ((st = c->staging(0, bytes, &s)) || (st = c->staging(1, bytes, &r))) return st;
  MUGIQ_CHECK_HIP(hipMemcpyAsync(s, send_h, bytes, hipMemcpyHostToDevice, c->hostStream));
  MUGIQ_CHECK_NCCL(g_api.Reduce(s, r, n, nccl_type(precision), kNcclSum, 0, c->space, c->hostStream));
  if (c->isTimeProcess) MUGIQ_CHECK_HIP(hipMemcpyAsync(recv_h, r, bytes, hipMemcpyDeviceToHost, c->hostStream));
  MUGIQ_CHECK_HIP(hipStreamSynchronize(c->hostStream));
  return MUGIQ_HIP_SUCCESS;
}
int cb_gather_time(void *ctx, const void *send_h, void *recv_h, size_t n, int precision) {
  auto *c = static_cast<MugiqHipRcclComm *>(ctx);
  if (!c->isTimeProcess) return MUGIQ_HIP_SUCCESS;  // (MPI_Gather over COMM_TIME: the others are not in the communicator)
  const size_t bytes = n * (size_t)precision;
  void *s = nullptr, *r = nullptr;
  int st;
  if ((st = c->staging(0, bytes, &s)) || (st = c->staging(1, bytes * (size_t)c->grid[3], &r))) return st;
  MUGIQ_CHECK_HIP(hipMemcpyAsync(s, send_h, bytes, hipMemcpyHostToDevice, c->hostStream));
  MUGIQ_CHECK_NCCL(g_api.AllGather(s, r, n, nccl_type(precision), c->time, c->hostStream));
  if (c->coord[3] == 0) MUGIQ_CHECK_HIP(hipMemcpyAsync(recv_h, r, bytes * (size_t)c->grid[3], hipMemcpyDeviceToHost, c->hostStream));
  MUGIQ_CHECK_HIP(hipStreamSynchronize(c->hostStream));
  return MUGIQ_HIP_SUCCESS;
}
int cb_bcast(void *ctx, void *buf_h, size_t n, int precision) {
  auto *c = static_cast<MugiqHipRcclComm *>(ctx);
  const size_t bytes = n * (size_t)precision;
  void *s = nullptr;
  int st;
  if ((st = c->staging(0, bytes, &s))) return st;
  if (c->rank == 0) MUGIQ_CHECK_HIP(hipMemcpyAsync(s, buf_h, bytes, hipMemcpyHostToDevice, c->hostStream));
  MUGIQ_CHECK_NCCL(g_api.Broadcast(s, s, n, nccl_type(precision), 0, c->world, c->hostStream));
  if (c->rank != 0) MUGIQ_CHECK_HIP(hipMemcpyAsync(buf_h, s, bytes, hipMemcpyDeviceToHost, c->hostStream));
  MUGIQ_CHECK_HIP(hipStreamSynchronize(c->hostStream));
  return MUGIQ_HIP_SUCCESS;
}

// coordinates, sub-communicators, staging stream
int finish_create(MugiqHipRcclComm *c, const int grid[4], const int partitioned[4]) {
  long long prod = 1;
  for (int d = 0; d < 4; d++) {
    if (grid[d] < 1) return mugiq::set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "mugiq_hip_rccl: grid[%d] = %d", d, grid[d]);
    c->grid[d] = grid[d];
    c->partitioned[d] = partitioned ? (partitioned[d] != 0) : 0;
    prod *= grid[d];
  }
  MUGIQ_CHECK_NCCL(g_api.CommCount(c->world, &c->size));
  MUGIQ_CHECK_NCCL(g_api.CommUserRank(c->world, &c->rank));
  if (prod != c->size) return mugiq::set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "mugiq_hip_rccl: process grid %dx%dx%dx%d does not match %d ranks", grid[0], grid[1], grid[2], grid[3], c->size);
  int r = c->rank;  // QUDA's comm_rank_from_coords: x slowest, t fastest
  c->coord[3] = r % grid[3];
  r /= grid[3];
  c->coord[2] = r % grid[2];
  r /= grid[2];
  c->coord[1] = r % grid[1];
  c->coord[0] = r / grid[1];
  c->isTimeProcess = c->coord[0] == 0 && c->coord[1] == 0 && c->coord[2] == 0;
  if (c->size > 1) {  // COMM_SPACE and COMM_TIME (lib/loop_mugiq.cpp:61-88); collective over the world communicator
    const int keySpace = (c->coord[0] * grid[1] + c->coord[1]) * grid[2] + c->coord[2];
    MUGIQ_CHECK_NCCL(g_api.CommSplit(c->world, c->coord[3], keySpace, &c->space, nullptr));
    MUGIQ_CHECK_NCCL(g_api.CommSplit(c->world, c->isTimeProcess ? 0 : kSplitNoColor, c->coord[3], &c->time, nullptr));
  }
  MUGIQ_CHECK_HIP(hipStreamCreateWithFlags(&c->hostStream, hipStreamNonBlocking));
  return MUGIQ_HIP_SUCCESS;
}

}  // namespace

extern "C" int mugiq_hip_rccl_relay_plan(int rank, const int grid[4], int dim, int dir, size_t bytes, int max_ops, int *phase, int *kind,
                                         int *peer, size_t *offset, size_t *len, size_t *bounce_bytes) {
  if (!grid || dim < 0 || dim > 3 || (dir != 1 && dir != -1)) return -MUGIQ_HIP_ERROR_INVALID_ARGUMENT;
  std::vector<RelayOp> ops;
  relay_plan(rank, grid, dim, dir, bytes, ops, bounce_bytes);
  for (int i = 0; i < (int)ops.size() && i < max_ops; i++) {
    if (phase) phase[i] = ops[i].phase;
    if (kind) kind[i] = ops[i].kind;
    if (peer) peer[i] = ops[i].peer;
    if (offset) offset[i] = ops[i].offset;
    if (len) len[i] = ops[i].len;
  }
  return (int)ops.size();
}

extern "C" int mugiq_hip_rccl_comm_set_multipath(MugiqHipRcclComm *c, int on) {
  MUGIQ_REQUIRE(c != nullptr, "mugiq_hip_rccl_comm_set_multipath: NULL communicator");
  MUGIQ_REQUIRE(c->groupDepth == 0 && c->pending.empty(), "mugiq_hip_rccl_comm_set_multipath: inside a transfer group");
  c->multipath = on != 0 && c->size > 2;
  return MUGIQ_HIP_SUCCESS;
}

extern "C" {

int mugiq_hip_rccl_get_unique_id(void *id128_out) {
  MUGIQ_REQUIRE(id128_out != nullptr, "mugiq_hip_rccl_get_unique_id: NULL output");
  int st = load_api();
  if (st) return st;
  UniqueId id;
  MUGIQ_CHECK_NCCL(g_api.GetUniqueId(&id));
  memcpy(id128_out, &id, sizeof(id));
  return MUGIQ_HIP_SUCCESS;
}

int mugiq_hip_rccl_comm_create(MugiqHipRcclComm **out, const void *id128, int rank, int size, const int grid[4], const int partitioned[4]) {
  MUGIQ_REQUIRE(out && id128 && grid, "mugiq_hip_rccl_comm_create: NULL argument");
  MUGIQ_REQUIRE(size >= 1 && rank >= 0 && rank < size, "mugiq_hip_rccl_comm_create: rank %d of %d", rank, size);
  *out = nullptr;
  int st = load_api();
  if (st) return st;
  UniqueId id;
  memcpy(&id, id128, sizeof(id));
  auto *c = new MugiqHipRcclComm_s();
  int r = g_api.CommInitRank(&c->world, size, id, rank);  // (collective: every rank of the job calls this)
  if (r != kNcclSuccess) {
    delete c;
    return mugiq::set_error(MUGIQ_HIP_ERROR_HIP, "mugiq_hip_rccl_comm_create: ncclCommInitRank failed: %s", g_api.GetErrorString(r));
  }
  c->ownsWorld = true;
  if ((st = finish_create(c, grid, partitioned))) {
    mugiq_hip_rccl_comm_destroy(c);
    return st;
  }
  *out = c;
  return MUGIQ_HIP_SUCCESS;
}

int mugiq_hip_rccl_comm_from_nccl(MugiqHipRcclComm **out, void *ncclComm_world, const int grid[4], const int partitioned[4]) {
  MUGIQ_REQUIRE(out && ncclComm_world && grid, "mugiq_hip_rccl_comm_from_nccl: NULL argument");
  *out = nullptr;
  int st = load_api();
  if (st) return st;
  auto *c = new MugiqHipRcclComm_s();
  c->world = static_cast<Comm>(ncclComm_world);
  c->ownsWorld = false;
  if ((st = finish_create(c, grid, partitioned))) {
    mugiq_hip_rccl_comm_destroy(c);
    return st;
  }
  *out = c;
  return MUGIQ_HIP_SUCCESS;
}

int mugiq_hip_rccl_comm_fill(MugiqHipRcclComm *c, MugiqHipComm *out) {
  MUGIQ_REQUIRE(c && out, "mugiq_hip_rccl_comm_fill: NULL argument");
  memset(out, 0, sizeof(*out));
  out->ctx = c;
  out->rank = c->rank;
  out->size = c->size;
  for (int d = 0; d < 4; d++) {
    out->grid[d] = c->grid[d];
    out->coord[d] = c->coord[d];
    out->partitioned[d] = c->partitioned[d];
  }
  out->sendrecv = cb_sendrecv;
  out->reduce_space = cb_reduce_space;
  out->gather_time = cb_gather_time;
  out->bcast = cb_bcast;
  out->group_begin = cb_group_begin;
  out->group_end = cb_group_end;
  return MUGIQ_HIP_SUCCESS;
}

int mugiq_hip_rccl_comm_destroy(MugiqHipRcclComm *c) {
  if (!c) return MUGIQ_HIP_SUCCESS;
  if (c->hostStream) {
    (void)hipStreamSynchronize(c->hostStream);
    (void)hipStreamDestroy(c->hostStream);
  }
  for (int i = 0; i < 2; i++)
    if (c->stage[i]) (void)hipFree(c->stage[i]);
  if (c->bounce) (void)hipFree(c->bounce);
  if (g_api.lib) {
    if (c->space) (void)g_api.CommDestroy(c->space);
    if (c->time) (void)g_api.CommDestroy(c->time);
    if (c->world && c->ownsWorld) (void)g_api.CommDestroy(c->world);
  }
  delete c;
  return MUGIQ_HIP_SUCCESS;
}

}  // extern "C"
