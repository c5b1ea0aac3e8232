// Host-side driver: the Loop_Mugiq<Float,order> and Displace<Float,order> classes of the reference
// (include/loop_mugiq.h, lib/loop_mugiq.cpp, include/displace.h, lib/displace.cpp) over the C-ABI operators.
//
// What is kept: LoopComputeParam's slot bookkeeping (include/loop_mugiq.h:221-256), the buffer set and element
// counts of allocateDataMemory (lib/loop_mugiq.cpp:101-158), the loop nest of computeCoarseLoop (:455-509), the
// displacement string table (include/displace.h:21, lib/displace.cpp:137-223), and the sequence of
// performMomentumProjection (:343-424: reorder -> GEMM -> D2H -> reduce over space ranks -> gather over time ranks
// -> broadcast).  What is dropped: the per-eigenvector field copies (:483,487,501), blas::zero + the two copies
// of swapAuxDispVec (lib/displace.cpp:47-59), the per-launch cudaMalloc/cudaMemcpy/cudaFree/cudaDeviceSynchronize
// (lib/contract_wrappers.cu:93-114), and the exchange of all four faces in both directions per step.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "internal.h"

namespace mugiq {

static const char *kDisplaceFlagArray[8] = {"+x", "-x", "+y", "-y", "+z", "-z", "+t", "-t"};  // include/displace.h:21

// Displace::WhichDisplaceFlag / WhichDisplaceDir / WhichDisplaceSign  (lib/displace.cpp:137-202):
// flag = index in the table; dir = flag/2; even flags are "+" (DispSignPlus = 1), odd "-" (DispSignMinus = 0)
static int parse_displacement(const char *s, int *dir, int *sign) {
  for (int i = 0; i < 8; i++)
    if (s && strcmp(s, kDisplaceFlagArray[i]) == 0) {
      *dir = i / 2;
      *sign = (i % 2 == 0) ? MUGIQ_HIP_DISP_SIGN_PLUS : MUGIQ_HIP_DISP_SIGN_MINUS;
      return MUGIQ_HIP_SUCCESS;
    }
  return set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "WhichDisplaceFlag: Cannot parse given displacement string = %s.",
                   s ? s : "(null)");
}

}  // namespace mugiq

using namespace mugiq;

struct MugiqHipLoop_s {
  // ---- LoopComputeParam (include/loop_mugiq.h:141-271)
  int Nmom = 0, FTSign = 1, calcType = MUGIQ_HIP_LOOP_CALC_TYPE_OPT_KERNEL;
  bool doMomProj = false, doNonLocal = false;
  std::vector<int> momMatrix;
  int localL[4], totalL[4];
  int volumeCB = 0, locT = 0, totT = 0;
  long long locV4 = 1, locV3 = 1, totV3 = 1;
  std::vector<std::string> dispEntry, dispString;
  std::vector<int> dispStart, dispStop, nLoopPerEntry, nLoopOffset, dispDir, dispSign;
  std::vector<int> derivedFrom;  // per entry: the entry it was reflected from in the last compute, or -1
  int nDispEntries = 0, nLoop = 0, nData = 0;
  std::string fnameMom, fnamePos;
  bool writeMom = false, writePos = false;
  // ---- inputs
  std::vector<MugiqHipSpinorField> eVecs;
  std::vector<double> sigma;
  int nEv = 0, precision = 8, order = 2;
  int loopPrecision = 8;  // precision of the loop buffers / FT (= precision, or 8 over fp32 fields: mixed mode)
  MugiqHipGaugeField gauge;
  bool haveGauge = false;
  MugiqHipComm comm;
  bool haveComm = false;
  int commDim[4] = {0, 0, 0, 0};
  hipStream_t stream = nullptr;
  // halo transfers run on their own stream so they overlap the interior part of the fused contraction
  hipStream_t commStream = nullptr;
  hipStream_t packStream = nullptr;  // the face layers are packed here, block by block, while the previous block travels
  hipEvent_t evPacked = nullptr, evHalo = nullptr, evEntryPacked = nullptr;
  // ---- data buffers (include/loop_mugiq.h:49-57, lib/loop_mugiq.cpp:101-158)
  long long nElemMomTotPerLoop = 0, nElemMomLocPerLoop = 0, nElemPosLocPerLoop = 0;
  long long nElemMomTot = 0, nElemMomLoc = 0, nElemPosLoc = 0, nElemPhMat = 0;
  void *dataPos_d = nullptr, *dataPosMP_d = nullptr, *dataMom_d = nullptr, *phaseMatrix_d = nullptr;
  void *dataPos = nullptr, *dataMom_h = nullptr, *dataMom = nullptr, *dataMom_bcast = nullptr;
  bool dataPosCopied = false, dataPosPinned = false, momProjDone = false, computed = false;
  // OPT plan: the ultra-local loop rides along with one displaced entry when the tiled kernel has room for it (see
  // mugiq_hip_displaced_loop_contraction_fused_carry); ultraCarried says whether an entry of this compute has produced it
  bool carryUltra = false, ultraCarried = false;
  int ultraCarrier = -1;  // the entry that took it along in the last compute, or -1
  // OPT plan, momentum-space output: reflected entries are derived on the gathered momentum-space array (csrc/reflect_mom.cpp)
  // and exist in position space only once somebody asks for dataPos (posReflectPending: not materialised yet)
  bool momReflect = false, posReflectPending = false;
  // ---- MG coarse path (eigsolve->computeCoarse): coarse eigenvectors + one Transfer level (lib/loop_mugiq.cpp:277-319,482)
  bool coarseMode = false;
  std::vector<MugiqHipCoarseField> coarseVecs;
  MugiqHipTransfer transfer;
  void *fineStore = nullptr;  // prolonged eigenvectors, owned; NULL when only the fused prolong-contract is needed
  // more than one coarse level (mg_env.nCoarseLevels > 1): the eigenvectors live on the coarsest level; upper[l] is the
  // transfer between level l+1 and level l+2, levelVecs[l] the eigenvectors on level l+1 (levelVecs.back() = the input,
  // levelVecs[0] = coarseVecs, what the finest transfer prolongs); levelStore[l] owns the intermediate fields of level l+1
  std::vector<MugiqHipTransfer> upper;
  std::vector<std::vector<MugiqHipCoarseField>> levelVecs;
  std::vector<void *> levelStore;
  // ---- displacement scratch (Displace::auxDispVec and friends)
  // Scratch lives in a pool owned by the loop object: hipMalloc/hipFree of ~GB buffers per displacement entry cost
  // 10-100 ms and synchronise the device (measured), so buffers are recycled across entries and computes.
  struct PoolBuf {
    void *ptr;
    size_t bytes;
    bool inUse;
  };
  std::vector<PoolBuf> pool;
  std::vector<void *> scratch;  // pool buffers handed out for the current entry (returned by free_scratch)
  // halos posted ahead of their entry (OPT plan): the eigenvector layers of every partitioned entry are packed and sent
  // at the start of the compute, the entries of unpartitioned directions run while they travel
  struct HaloPost {
    void *gsend = nullptr, *grecv = nullptr;
    hipEvent_t evPacked = nullptr, evHalo = nullptr;
    // the halo travels in blocks of eigenvectors: evPackedBlk[b] (pack stream) / evBlock[b] (halo stream: block b has landed)
    std::vector<hipEvent_t> evPackedBlk, evBlock;
    int nBlocks = 0, blockN = 0;
    bool posted = false;
    int entryPacksFrom = -1;     // >= 0: the face layers of eigenvectors entryPacksFrom .. are written by the entry that runs first, on its
                                 // way through the eigenvectors (csrc/fused_mfma.hip, row tile); -1: by mugiq_hip_pack_face_layers
    void *axialGauge = nullptr;  // the axial gauge of the entry, built once for all its launches (csrc/fused_mfma.hip), or NULL
    bool selfAlias = false;  // the neighbour is this rank itself (an axis of extent 1 under forced partitioning): the face layers are
                             // packed straight into the ghost buffer, no send buffer and no message (MUGIQ_HIP_SELF_HALO_COPY=1: keep them)
    std::vector<MugiqHipSpinorField> E;  // path-link fields built ahead (their small face exchanges go first)
  };
  std::vector<HaloPost> halo;   // per displacement entry
  std::vector<void *> held;     // pool buffers held until the end of the compute
  int halosPackedInEntry = 0;   // posted halos whose face layers the first entry of the last compute wrote (0: pack kernels only)

  // ---- optional phase timing (mugiq_hip_loop_set_profiling): device time between two events bracketing each phase
  struct Phase {
    int kind, entry;
    double bytes, ms;
    int e0, e1;  // indices into `events`, -1 = host-timed (ms already set)
  };
  bool profiling = false;
  std::vector<Phase> phases;
  std::vector<hipEvent_t> events;
  size_t eventsUsed = 0;

  size_t cplxBytes() const { return 2 * (size_t)precision; }      // eigenvector / link storage
  size_t loopBytes() const { return 2 * (size_t)loopPrecision; }  // loop buffers, phases, momentum projection
};

namespace mugiq {

int write_loops_hdf5_mom(const char *filename, const void *dataMom_bcast, int precision, int Nmom, const int *momMatrix,
                         int nDispEntries, const std::vector<std::string> &dispString, const std::vector<int> &dispStart,
                         const std::vector<int> &dispStop, int nLoop, int locT, int totT);  // hdf5_writer.cpp

static int dev_alloc(MugiqHipLoop *lp, void **p, size_t bytes, bool zero) {
  MUGIQ_CHECK_HIP(hipMalloc(p, bytes ? bytes : 16));
  if (zero) MUGIQ_CHECK_HIP(hipMemsetAsync(*p, 0, bytes, lp->stream));
  return MUGIQ_HIP_SUCCESS;
}

// ---- phase timing ----------------------------------------------------------------------------------------------
static int timing_event(MugiqHipLoop *lp, hipStream_t s) {
  if (lp->eventsUsed == lp->events.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return -1;
    lp->events.push_back(e);
  }
  const int i = (int)lp->eventsUsed++;
  if (hipEventRecord(lp->events[i], s) != hipSuccess) return -1;
  return i;
}
// opens a phase on stream `s`; returns its index (or -1 when profiling is off)
static int phase_begin(MugiqHipLoop *lp, int kind, int entry, hipStream_t s, double bytes = 0) {
  if (!lp->profiling) return -1;
  lp->phases.push_back({kind, entry, bytes, 0.0, timing_event(lp, s), -1});
  return (int)lp->phases.size() - 1;
}
static void phase_end(MugiqHipLoop *lp, int idx, hipStream_t s) {
  if (idx >= 0) lp->phases[idx].e1 = timing_event(lp, s);
}
static void phase_host(MugiqHipLoop *lp, int kind, double ms, double bytes = 0) {
  if (lp->profiling) lp->phases.push_back({kind, -1, bytes, ms, -1, -1});
}
// after the streams have been synchronised: events -> milliseconds
static void phases_resolve(MugiqHipLoop *lp) {
  for (auto &ph : lp->phases)
    if (ph.e0 >= 0 && ph.e1 >= 0) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, lp->events[ph.e0], lp->events[ph.e1]) == hipSuccess) ph.ms = ms;
      ph.e0 = ph.e1 = -1;
    }
}

// scratch from the loop's pool (best fit among the free buffers, else a new allocation); returned by free_scratch
static int scratch_alloc(MugiqHipLoop *lp, void **p, size_t bytes, bool zero) {
  int best = -1;
  for (size_t i = 0; i < lp->pool.size(); i++)
    if (!lp->pool[i].inUse && lp->pool[i].bytes >= bytes && (best < 0 || lp->pool[i].bytes < lp->pool[best].bytes)) best = (int)i;
  if (best < 0) {
    void *q = nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    MUGIQ_CHECK_HIP(hipMalloc(&q, bytes ? bytes : 16));
    if (lp->profiling)
      lp->phases.push_back({MUGIQ_HIP_PHASE_SCRATCH_ALLOC, -1, (double)bytes,
                            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), -1, -1});
    lp->pool.push_back({q, bytes, false});
    best = (int)lp->pool.size() - 1;
  }
  lp->pool[best].inUse = true;
  *p = lp->pool[best].ptr;
  lp->scratch.push_back(*p);
  if (zero) MUGIQ_CHECK_HIP(hipMemsetAsync(*p, 0, bytes, lp->stream));
  return MUGIQ_HIP_SUCCESS;
}

// A FLOAT2, pad-0 scratch field with the eigenvectors' geometry (+ room for both depth-1 ghost zones of `dim`).
// (likeEvecs: keep the eigenvectors' own stride / pad, as Displace's auxDispVec does -- lib/displace.cpp:32-37 creates it
// from the eigenvectors' parameters -- so that it can stand in for an eigenvector in the batched kernels)
static int make_scratch_field(MugiqHipLoop *lp, MugiqHipSpinorField *f, int order, bool zero = false, bool likeEvecs = false) {
  *f = lp->eVecs[0];
  f->field_order = order;
  if (!likeEvecs) {
    f->stride = lp->volumeCB;
    f->parity_offset = (int64_t)12 * lp->volumeCB;
  }
  for (int d = 0; d < 4; d++) f->ghost[d][0] = f->ghost[d][1] = nullptr;
  void *p = nullptr;
  int st = scratch_alloc(lp, &p, (size_t)2 * f->parity_offset * lp->cplxBytes(), zero);
  if (st) return st;
  f->data = p;
  return MUGIQ_HIP_SUCCESS;
}

// exchangeGhostVec for ONE face: the face the displacement (dir, sign) reads (lib/contract_wrappers.cu:166-169
// exchanges all partitioned dims in both directions).
static int exchange_face(MugiqHipLoop *lp, MugiqHipSpinorField *src, int dir, int sign, void *send_d, void *recv_d) {
  const int high = (sign == MUGIQ_HIP_DISP_SIGN_PLUS) ? 0 : 1;  // sign +: my LOW face feeds the backward neighbour
  int st = mugiq_hip_pack_face(send_d, src, dir, high, lp->stream);
  if (st) return st;
  const size_t bytes = (size_t)24 * (lp->volumeCB / lp->localL[dir]) * lp->cplxBytes();
  st = lp->comm.sendrecv(lp->comm.ctx, send_d, recv_d, bytes, dir, high ? +1 : -1, lp->stream);
  if (st) return set_error(MUGIQ_HIP_ERROR_HIP, "halo sendrecv callback failed with status %d", st);
  src->ghost[dir][sign == MUGIQ_HIP_DISP_SIGN_PLUS ? 1 : 0] = recv_d;
  return MUGIQ_HIP_SUCCESS;
}

// ---- the reference's own plan: one displacement + one contraction launch per eigenvector and step -----------
static int entry_basic(MugiqHipLoop *lp, int id, void *slot0) {
  const int dir = lp->dispDir[id], sign = lp->dispSign[id];
  const bool part = lp->commDim[dir] != 0;
  MugiqHipSpinorField aux[2];
  int st;
  for (int i = 0; i < 2; i++)
    if ((st = make_scratch_field(lp, &aux[i], lp->order, false, true))) return st;  // fully written by every displacement
  void *send_d = nullptr, *recv_d = nullptr;
  if (part) {
    const size_t fb = (size_t)24 * (lp->volumeCB / lp->localL[dir]) * lp->cplxBytes();
    if ((st = scratch_alloc(lp, &send_d, fb, false))) return st;
    if ((st = scratch_alloc(lp, &recv_d, fb, false))) return st;
  }
  const size_t slotBytes = (size_t)lp->nElemPosLocPerLoop * lp->loopBytes();
  for (int n = 0; n < lp->nEv; n++) {  // lib/loop_mugiq.cpp:478
    MugiqHipSpinorField cur = lp->eVecs[n];
    int dispCount = 0;
    for (int idisp = 1; idisp <= lp->dispStop[id]; idisp++) {  // :489
      MugiqHipSpinorField *dst = &aux[idisp & 1];
      if (part && (st = exchange_face(lp, &cur, dir, sign, send_d, recv_d))) return st;
      if ((st = mugiq_hip_perform_covariant_displacement_vector(dst, &cur, &lp->gauge, dir, sign, lp->commDim, lp->stream)))
        return st;  // Displace::doVectorDisplacement, lib/displace.cpp:55-67
      cur = *dst;
      if (idisp >= lp->dispStart[id] && idisp <= lp->dispStop[id]) {  // :491-496
        void *slot = static_cast<char *>(slot0) + slotBytes * dispCount;
        if ((st = mugiq_hip_perform_loop_contraction_batched_mixed(slot, lp->loopPrecision, &lp->eVecs[n], &cur, &lp->sigma[n], 1,
                                                                   lp->stream)))
          return st;
        dispCount++;
      }
    }
  }
  return MUGIQ_HIP_SUCCESS;
}

// The same sequence for a block of eigenvectors at a time (OPT plan, displacement longer than the local extent of a
// partitioned dimension: the multi-layer halo cannot reach past the nearest neighbour, single steps can): per step ONE
// message carries the faces of all eigenvectors of the block and ONE contraction launch takes the whole block, instead of
// one exchange and one launch per eigenvector and step (thousands of small messages at configs[2] sizes).
static int entry_stepwise_blocked(MugiqHipLoop *lp, int id, void *slot0) {
  const int dir = lp->dispDir[id], sign = lp->dispSign[id], start = lp->dispStart[id], stop = lp->dispStop[id];
  const bool part = lp->commDim[dir] != 0;
  const size_t fieldB = (size_t)2 * lp->eVecs[0].parity_offset * lp->cplxBytes();
  const size_t faceB = (size_t)24 * (lp->volumeCB / lp->localL[dir]) * lp->cplxBytes();
  const size_t budget = (size_t)8 << 30;  // two auxiliary fields per eigenvector of the block
  const int nb = (int)std::max<size_t>(1, std::min<size_t>((size_t)lp->nEv, budget / (2 * fieldB)));
  int st;
  std::vector<MugiqHipSpinorField> aux[2];
  for (int h = 0; h < 2; h++) {
    aux[h].resize(nb);
    for (int i = 0; i < nb; i++)
      if ((st = make_scratch_field(lp, &aux[h][i], lp->order, false, true))) return st;  // fully written by every displacement
  }
  void *gsend = nullptr, *grecv = nullptr;
  if (part) {
    if ((st = scratch_alloc(lp, &gsend, faceB * nb, false))) return st;
    if ((st = scratch_alloc(lp, &grecv, faceB * nb, false))) return st;
  }
  const size_t slotBytes = (size_t)lp->nElemPosLocPerLoop * lp->loopBytes();
  const int high = (sign == MUGIQ_HIP_DISP_SIGN_PLUS) ? 0 : 1;  // sign +: my LOW face feeds the backward neighbour
  std::vector<MugiqHipSpinorField> cur(nb);
  for (int n0 = 0; n0 < lp->nEv; n0 += nb) {
    const int nv = std::min(nb, lp->nEv - n0);
    for (int i = 0; i < nv; i++) cur[i] = lp->eVecs[n0 + i];
    for (int idisp = 1; idisp <= stop; idisp++) {
      if (part) {
        if ((st = mugiq_hip_pack_face_layers(gsend, cur.data(), nv, dir, high, 1, lp->stream))) return st;
        st = lp->comm.sendrecv(lp->comm.ctx, gsend, grecv, faceB * nv, dir, high ? +1 : -1, lp->stream);
        if (st) return set_error(MUGIQ_HIP_ERROR_HIP, "halo sendrecv callback failed with status %d", st);
      }
      for (int i = 0; i < nv; i++) {
        if (part) cur[i].ghost[dir][sign == MUGIQ_HIP_DISP_SIGN_PLUS ? 1 : 0] = static_cast<char *>(grecv) + faceB * i;
        MugiqHipSpinorField *dst = &aux[idisp & 1][i];
        if ((st = mugiq_hip_perform_covariant_displacement_vector(dst, &cur[i], &lp->gauge, dir, sign, lp->commDim, lp->stream))) return st;
        cur[i] = *dst;
      }
      if (idisp >= start) {
        void *slot = static_cast<char *>(slot0) + slotBytes * (size_t)(idisp - start);
        if ((st = mugiq_hip_perform_loop_contraction_batched_mixed(slot, lp->loopPrecision, &lp->eVecs[n0], cur.data(), &lp->sigma[n0], nv,
                                                                   lp->stream)))
          return st;
      }
    }
  }
  return MUGIQ_HIP_SUCCESS;
}

// path-ordered link products W_k as E_k = D^k E_0, E_0(x)(s,c) = delta_sc, s < 3 (scratch fields; k = 0 .. stop)
static int build_path_links(MugiqHipLoop *lp, int id, std::vector<MugiqHipSpinorField> &E) {
  const int dir = lp->dispDir[id], sign = lp->dispSign[id], stop = lp->dispStop[id];
  const bool part = lp->commDim[dir] != 0;
  int st;
  E.assign(stop + 1, MugiqHipSpinorField());
  for (int k = 0; k <= stop; k++)
    if ((st = make_scratch_field(lp, &E[k], 2, false))) return st;  // every site of E_k is written below
  if ((st = fill_identity_links(&E[0], lp->stream))) return st;
  void *send_d = nullptr, *recv_d = nullptr;
  if (part) {
    const size_t fb = (size_t)24 * (lp->volumeCB / lp->localL[dir]) * lp->cplxBytes();
    if ((st = scratch_alloc(lp, &send_d, fb, false))) return st;
    if ((st = scratch_alloc(lp, &recv_d, fb, false))) return st;
  }
  for (int k = 1; k <= stop; k++) {
    if (part && (st = exchange_face(lp, &E[k - 1], dir, sign, send_d, recv_d))) return st;
    if ((st = mugiq_hip_perform_covariant_displacement_vector(&E[k], &E[k - 1], &lp->gauge, dir, sign, lp->commDim, lp->stream)))
      return st;
  }
  return MUGIQ_HIP_SUCCESS;
}

// bytes of the multi-layer eigenvector halo of entry `id` (one direction): `stop` face layers of all eigenvectors
static size_t halo_bytes(const MugiqHipLoop *lp, int id) {
  return (size_t)lp->dispStop[id] * 24 * (size_t)(lp->volumeCB / lp->localL[lp->dispDir[id]]) * lp->cplxBytes() * (size_t)lp->nEv;
}

static int reflection_source(const MugiqHipLoop *lp, int id);

// The OPT plan: which entries are reflected from which (derivedFrom), and which of the computed entries along partitioned axes
// get their eigenvector halo posted AHEAD, at the start of the compute (ahead[id] = 1).  The ghost-layer buffers posted ahead
// may take a quarter of the device memory.  The rule must not depend on anything that can differ between ranks (such as the
// memory free right now): every rank has to take the same decision, or the transfers would not pair up.
static int plan_opt(MugiqHipLoop *lp, std::vector<char> &ahead) {
  ahead.assign(lp->nDispEntries, 0);
  for (int id = 0; id < lp->nDispEntries; id++) {
    lp->derivedFrom[id] = -1;  // entries after `id` are still -1 here: reflection_source only looks at jd < id
    lp->derivedFrom[id] = reflection_source(lp, id);
  }
  size_t freeB = 0, totalB = 0;
  MUGIQ_CHECK_HIP(hipMemGetInfo(&freeB, &totalB));
  size_t budget = totalB / 4;
  if (const char *e = getenv("MUGIQ_HIP_HALO_AHEAD"))
    if (atoi(e) == 0) budget = 0;
  for (int id = 0; id < lp->nDispEntries; id++) {
    const int dir = lp->dispDir[id];
    if (lp->derivedFrom[id] >= 0 || !lp->commDim[dir] || lp->dispStop[id] > lp->localL[dir]) continue;
    const size_t bytes = halo_bytes(lp, id);
    if (2 * bytes > budget) continue;  // this entry exchanges eigenvector blocks of <= 4 GiB inside its own turn instead
    budget -= 2 * bytes;
    ahead[id] = 1;
  }
  return MUGIQ_HIP_SUCCESS;
}

// a buffer of `bytes` into the pool, free for scratch_alloc to hand out (hipMalloc of multi-GB buffers costs ~40 ms per GB: the
// driver maps and clears the pages -- so what the plan is known to need is allocated when the loop object is built, like the
// reference's allocateDataMemory, lib/loop_mugiq.cpp:101-158, not inside the first compute)
static int pool_reserve(MugiqHipLoop *lp, size_t bytes) {
  void *q = nullptr;
  MUGIQ_CHECK_HIP(hipMalloc(&q, bytes ? bytes : 16));
  lp->pool.push_back({q, bytes, false});
  return MUGIQ_HIP_SUCCESS;
}

// An axis of extent 1 that is partitioned all the same (MugiqHipComm.partitioned): this rank is its own forward and backward
// neighbour, the "message" would be a device copy of what the pack kernel has just written.  Pack into the ghost buffer instead.
static bool self_neighbour_alias(const MugiqHipLoop *lp, int dir) {
  if (!lp->haveComm || lp->comm.grid[dir] != 1) return false;
  if (const char *e = getenv("MUGIQ_HIP_SELF_HALO_COPY"))
    if (atoi(e) != 0) return false;
  return true;
}

static int reserve_plan_buffers(MugiqHipLoop *lp) {
  if (lp->calcType == MUGIQ_HIP_LOOP_CALC_TYPE_BASIC_KERNEL || lp->nDispEntries == 0) return MUGIQ_HIP_SUCCESS;
  std::vector<char> ahead;
  int st = plan_opt(lp, ahead);
  if (st) return st;
  const size_t fieldB = (size_t)24 * lp->volumeCB * lp->cplxBytes();  // a FLOAT2 pad-0 path-link field
  bool anyAhead = false;
  for (int id = 0; id < lp->nDispEntries; id++) {
    if (!ahead[id]) continue;
    const size_t faceB = (size_t)24 * (lp->volumeCB / lp->localL[lp->dispDir[id]]) * lp->cplxBytes();
    size_t gb = 0;
    {  // the entry's axial gauge (csrc/fused_mfma.hip), where that tile takes the entry
      std::vector<int> kv;
      for (int k = lp->dispStart[id]; k <= lp->dispStop[id]; k++) kv.push_back(k);
      gb = axial_gauge_bytes(lp->eVecs[0], lp->dispDir[id], kv.data(), (int)kv.size(), 1);
    }
    if (!(gb && axial_gauge_from_links_possible(lp->eVecs[0], lp->gauge, lp->dispStop[id], lp->dispDir[id], lp->dispSign[id]))) {
      // (prepare_halo: the gauge from the extended gauge field where its border reaches far enough -- then no link fields)
      for (int k = 0; k <= lp->dispStop[id]; k++)
        if ((st = pool_reserve(lp, fieldB))) return st;  // E_0 .. E_stop, held until the entry has run
      if ((st = pool_reserve(lp, faceB)) || (st = pool_reserve(lp, faceB))) return st;
    }
    if ((st = pool_reserve(lp, halo_bytes(lp, id)))) return st;
    if (!self_neighbour_alias(lp, lp->dispDir[id]) && (st = pool_reserve(lp, halo_bytes(lp, id)))) return st;
    if (gb && (st = pool_reserve(lp, gb))) return st;
    anyAhead = true;
  }
  // the entry that runs before the halos are posted keeps its link fields out of the pool until the compute ends (see
  // mugiq_hip_loop_compute): they come on top of what the posted entries hold
  if (anyAhead)
    for (int id = 0; id < lp->nDispEntries; id++)
      if (lp->derivedFrom[id] < 0 && !lp->commDim[lp->dispDir[id]]) {
        std::vector<int> kv;
        for (int k = lp->dispStart[id]; k <= lp->dispStop[id]; k++) kv.push_back(k);
        const bool direct = axial_gauge_from_links_possible(lp->eVecs[0], lp->gauge, lp->dispStop[id], lp->dispDir[id], lp->dispSign[id]);
        const size_t gb = direct ? axial_gauge_bytes(lp->eVecs[0], lp->dispDir[id], kv.data(), (int)kv.size(), 0) : 0;
        if (gb) {  // (entry_fused: the gauge straight from the gauge field, no link fields)
          if ((st = pool_reserve(lp, gb))) return st;
        } else {
          for (int k = 0; k <= lp->dispStop[id]; k++)
            if ((st = pool_reserve(lp, fieldB))) return st;
        }
        break;
      }
  return MUGIQ_HIP_SUCCESS;
}

static int ensure_comm_stream(MugiqHipLoop *lp) {
  if (!lp->packStream) MUGIQ_CHECK_HIP(hipStreamCreateWithFlags(&lp->packStream, hipStreamNonBlocking));
  if (!lp->commStream) {
    MUGIQ_CHECK_HIP(hipStreamCreateWithFlags(&lp->commStream, hipStreamNonBlocking));
    MUGIQ_CHECK_HIP(hipEventCreateWithFlags(&lp->evPacked, hipEventDisableTiming));
    MUGIQ_CHECK_HIP(hipEventCreateWithFlags(&lp->evHalo, hipEventDisableTiming));
    MUGIQ_CHECK_HIP(hipEventCreateWithFlags(&lp->evEntryPacked, hipEventDisableTiming));
  }
  return MUGIQ_HIP_SUCCESS;
}

// Halo of entry `id` posted ahead, step 1: link fields (their small face exchanges happen here, at once), ghost buffers for
// ALL eigenvectors, and the cut into blocks of eigenvectors the halo travels in.  (Whether an entry is posted ahead is
// plan_opt's decision.)
static int prepare_halo(MugiqHipLoop *lp, int id) {
  const size_t bytes = halo_bytes(lp, id);
  int st;
  MugiqHipLoop::HaloPost &h = lp->halo[id];
  if (!h.evPacked) {
    MUGIQ_CHECK_HIP(hipEventCreateWithFlags(&h.evPacked, hipEventDisableTiming));
    MUGIQ_CHECK_HIP(hipEventCreateWithFlags(&h.evHalo, hipEventDisableTiming));
  }
  {  // the entry is launched once for its interior tiles and once per halo block for its boundary tiles: one gauge for all of them
    std::vector<int> kv;
    for (int k = lp->dispStart[id]; k <= lp->dispStop[id]; k++) kv.push_back(k);
    const size_t gb = axial_gauge_bytes(lp->eVecs[0], lp->dispDir[id], kv.data(), (int)kv.size(), 1);
    h.axialGauge = nullptr;
    h.E.clear();
    if (gb && axial_gauge_from_links_possible(lp->eVecs[0], lp->gauge, lp->dispStop[id], lp->dispDir[id], lp->dispSign[id])) {
      // the neighbour's links of the continued positions are in the border of the extended gauge field: no path-link fields (and
      // none of their face exchanges) for this entry
      if ((st = scratch_alloc(lp, &h.axialGauge, gb, false))) return st;
      if ((st = build_axial_gauge_from_links(h.axialGauge, lp->eVecs[0], lp->gauge, lp->dispStop[id], lp->dispDir[id], lp->dispSign[id], lp->stream))) return st;
    } else if ((st = build_path_links(lp, id, h.E))) {  // compute stream: the entry's kernels read them there
      return st;
    } else if (gb) {
      std::vector<const void *> lk;
      for (int k = 1; k <= lp->dispStop[id]; k++) lk.push_back(h.E[k].data);
      if ((st = scratch_alloc(lp, &h.axialGauge, gb, false))) return st;
      if ((st = build_axial_gauge(h.axialGauge, lp->eVecs[0], lk.data(), lp->dispStop[id], lp->dispDir[id], lp->dispSign[id], lp->stream))) return st;
    }
  }
  h.selfAlias = self_neighbour_alias(lp, lp->dispDir[id]);
  if ((st = scratch_alloc(lp, &h.grecv, bytes, false))) return st;
  if (h.selfAlias) h.gsend = h.grecv;
  else if ((st = scratch_alloc(lp, &h.gsend, bytes, false))) return st;
  // all of these outlive the entries processed in between: move them from the per-entry list to the held list
  for (void *q : lp->scratch) lp->held.push_back(q);
  lp->scratch.clear();
  // Blocks of about 2 GiB (at most 8): the first block is on its way after a fraction of the packing, and the boundary tiles
  // of the first blocks run while the last ones still travel -- with ONE message the transfer could not start before all
  // face layers were packed (12 ms at configs[2]) and no boundary tile before the last byte had landed.  MUGIQ_HIP_HALO_BLOCKS
  // fixes the number (1 = the single message of round 2).
  int nb = (int)std::min<size_t>(8, std::max<size_t>(1, (bytes + ((size_t)1 << 31) - 1) >> 31));
  if (h.selfAlias) nb = 1;  // nothing travels: one block, one launch of the boundary tiles
  if (const char *e = getenv("MUGIQ_HIP_HALO_BLOCKS")) nb = std::max(1, std::min(64, atoi(e)));
  nb = std::min(nb, lp->nEv);
  h.blockN = (lp->nEv + nb - 1) / nb;
  h.nBlocks = (lp->nEv + h.blockN - 1) / h.blockN;
  while ((int)h.evBlock.size() < h.nBlocks) {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    MUGIQ_CHECK_HIP(hipEventCreateWithFlags(&e0, hipEventDisableTiming));
    MUGIQ_CHECK_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming));
    h.evPackedBlk.push_back(e0);
    h.evBlock.push_back(e1);
  }
  h.posted = true;
  h.entryPacksFrom = -1;
  return MUGIQ_HIP_SUCCESS;
}

// step 2: block b of every posted entry -- packed on the pack stream, handed to the transport on the halo stream (the entries
// of one block inside one transfer group: different axes, different links).  which: -1 every posted entry | 0 only those whose
// block b is packed by mugiq_hip_pack_face_layers | 1 only those whose block b the first entry has written (the pack stream
// waits for that entry; see mugiq_hip_loop_compute)
static bool entry_packs_block(const MugiqHipLoop::HaloPost &h, int b) { return h.entryPacksFrom >= 0 && b * h.blockN >= h.entryPacksFrom; }
static int send_halo_block(MugiqHipLoop *lp, int b, bool grouped, int which = -1) {
  int st = MUGIQ_HIP_SUCCESS;
  auto takes = [&](const MugiqHipLoop::HaloPost &h) {
    return h.posted && b < h.nBlocks && (which < 0 || (which == 1) == entry_packs_block(h, b));
  };
  for (int id = 0; id < lp->nDispEntries; id++) {
    MugiqHipLoop::HaloPost &h = lp->halo[id];
    if (!takes(h)) continue;
    const int dir = lp->dispDir[id], sign = lp->dispSign[id], stop = lp->dispStop[id];
    const int n0 = b * h.blockN, nv = std::min(h.blockN, lp->nEv - n0);
    const size_t perVec = halo_bytes(lp, id) / (size_t)lp->nEv;
    const int high = (sign == MUGIQ_HIP_DISP_SIGN_PLUS) ? 0 : 1;
    if (!entry_packs_block(h, b) &&
        (st = mugiq_hip_pack_face_layers(static_cast<char *>(h.gsend) + perVec * n0, &lp->eVecs[n0], nv, dir, high, stop, lp->packStream)))
      return st;
    MUGIQ_CHECK_HIP(hipEventRecord(h.evPackedBlk[b], lp->packStream));
    MUGIQ_CHECK_HIP(hipStreamWaitEvent(lp->commStream, h.evPackedBlk[b], 0));
  }
  bool anyMessage = false;
  for (int id = 0; id < lp->nDispEntries; id++) anyMessage = anyMessage || (takes(lp->halo[id]) && !lp->halo[id].selfAlias);
  grouped = grouped && anyMessage;
  if (grouped && (st = lp->comm.group_begin(lp->comm.ctx))) return set_error(MUGIQ_HIP_ERROR_HIP, "group_begin callback failed with status %d", st);
  for (int id = 0; id < lp->nDispEntries && !st; id++) {
    MugiqHipLoop::HaloPost &h = lp->halo[id];
    if (!takes(h) || h.selfAlias) continue;
    const int n0 = b * h.blockN, nv = std::min(h.blockN, lp->nEv - n0);
    const size_t perVec = halo_bytes(lp, id) / (size_t)lp->nEv;
    const int high = (lp->dispSign[id] == MUGIQ_HIP_DISP_SIGN_PLUS) ? 0 : 1;
    const int rc = lp->comm.sendrecv(lp->comm.ctx, static_cast<char *>(h.gsend) + perVec * n0, static_cast<char *>(h.grecv) + perVec * n0,
                                     perVec * nv, lp->dispDir[id], high ? +1 : -1, lp->commStream);
    if (rc) st = set_error(MUGIQ_HIP_ERROR_HIP, "halo sendrecv callback failed with status %d", rc);
  }
  if (grouped) {
    const int st2 = lp->comm.group_end(lp->comm.ctx, lp->commStream);
    if (!st && st2) st = set_error(MUGIQ_HIP_ERROR_HIP, "group_end callback failed with status %d", st2);
  }
  if (st) return st;
  for (int id = 0; id < lp->nDispEntries; id++) {
    MugiqHipLoop::HaloPost &h = lp->halo[id];
    if (takes(h)) MUGIQ_CHECK_HIP(hipEventRecord(h.evBlock[b], lp->commStream));
  }
  return MUGIQ_HIP_SUCCESS;
}

// ---- the fused plan -------------------------------------------------------------------------------------------
// part_sel: 0 = the whole entry; for an entry whose halo was posted ahead 1 = the interior tiles only, 2 = the boundary tiles
// only (the driver runs the interiors of ALL such entries before the first boundary: nothing then waits for a halo while there
// is still work that needs none)
static int entry_fused(MugiqHipLoop *lp, int id, void *slot0, int part_sel = 0) {
  const int dir = lp->dispDir[id], sign = lp->dispSign[id];
  const bool part = lp->commDim[dir] != 0;
  const int stop = lp->dispStop[id], start = lp->dispStart[id];
  int st;
  // a displacement longer than the local extent of a partitioned dimension reaches past the nearest neighbour: the
  // multi-layer halo cannot serve it, the step-by-step sequence (one face per step) can
  if (part && stop > lp->localL[dir]) return entry_stepwise_blocked(lp, id, slot0);
  std::vector<MugiqHipSpinorField> Elocal;
  const bool ahead = part && lp->halo[id].posted;
  std::vector<int> kv;
  for (int k = start; k <= stop; k++) kv.push_back(k);
  // A direction that is not partitioned and an entry the matrix-pipe tile takes: its axial gauge comes straight from the gauge field
  // (W_1 = U_mu, the continued positions are the wrapped sites) and the path-link fields W_1 .. W_stop are not built at all
  // (MUGIQ_HIP_GAUGE_FROM_LINKS = 0: build them and the gauge from them, as for the partitioned directions)
  void *directGauge = nullptr;
  if (!part && axial_gauge_from_links_possible(lp->eVecs[0], lp->gauge, stop, dir, sign)) {
    const size_t gb = axial_gauge_bytes(lp->eVecs[0], dir, kv.data(), (int)kv.size(), 0);
    if (gb) {
      if ((st = scratch_alloc(lp, &directGauge, gb, false))) return st;
      if ((st = build_axial_gauge_from_links(directGauge, lp->eVecs[0], lp->gauge, stop, dir, sign, lp->stream))) return st;
    }
  }
  if (!ahead && !directGauge && (st = build_path_links(lp, id, Elocal))) return st;
  std::vector<MugiqHipSpinorField> &E = ahead ? lp->halo[id].E : Elocal;
  std::vector<const void *> links;
  // (with the gauge at hand the fused call never looks at the link fields: the gauge buffer stands in, and names the hint)
  if (ahead && E.empty()) directGauge = lp->halo[id].axialGauge;  // (prepare_halo took the gauge from the gauge field: no link fields)
  for (int k = start; k <= stop; k++) links.push_back(directGauge ? directGauge : E[k].data);
  if (part && lp->halo[id].posted) {
    // the halo of all eigenvectors was posted at the start of the compute: interior tiles, then (once it has landed) the
    // boundary tiles
    MugiqHipLoop::HaloPost &h = lp->halo[id];
    struct HintScope {  // the entry's own axial gauge for the launches below (cleared on every way out)
      HintScope(const MugiqHipLoop::HaloPost &h, const std::vector<const void *> &links, int dir, int sign, int kmax) {
        if (h.axialGauge) set_axial_gauge_hint(h.axialGauge, links[0], dir, sign, kmax);
      }
      ~HintScope() { set_axial_gauge_hint(nullptr, nullptr, -1, -1, 0); }
    } hintScope(h, links, dir, sign, stop);
    int ph;
    if (part_sel != 2) {
      ph = phase_begin(lp, MUGIQ_HIP_PHASE_ENTRY_INTERIOR, id, lp->stream);
      if ((st = mugiq_hip_displaced_loop_contraction_fused_region(slot0, lp->loopPrecision, lp->eVecs.data(), lp->sigma.data(), lp->nEv,
                                                                  links.data(), kv.data(), (int)kv.size(), dir, sign, lp->commDim,
                                                                  h.grecv, stop, MUGIQ_HIP_REGION_INTERIOR | MUGIQ_HIP_REGION_OVERWRITE, lp->stream)))
        return st;
      phase_end(lp, ph, lp->stream);
    }
    if (part_sel == 1) return MUGIQ_HIP_SUCCESS;
    // boundary tiles, block of eigenvectors by block as the halo lands: the first block writes the boundary sites, the others add
    const size_t perVec = halo_bytes(lp, id) / (size_t)lp->nEv;
    for (int b = 0; b < h.nBlocks; b++) {
      const int n0 = b * h.blockN, nv = std::min(h.blockN, lp->nEv - n0);
      ph = phase_begin(lp, MUGIQ_HIP_PHASE_HALO_WAIT, id, lp->stream);  // idle time of the compute stream: what the overlap did not hide
      MUGIQ_CHECK_HIP(hipStreamWaitEvent(lp->stream, h.evBlock[b], 0));
      phase_end(lp, ph, lp->stream);
      ph = phase_begin(lp, MUGIQ_HIP_PHASE_ENTRY_BOUNDARY, id, lp->stream);
      st = mugiq_hip_displaced_loop_contraction_fused_region(slot0, lp->loopPrecision, &lp->eVecs[n0], &lp->sigma[n0], nv, links.data(), kv.data(),
                                                             (int)kv.size(), dir, sign, lp->commDim, static_cast<char *>(h.grecv) + perVec * n0, stop,
                                                             MUGIQ_HIP_REGION_BOUNDARY | (b == 0 ? MUGIQ_HIP_REGION_OVERWRITE : 0), lp->stream);
      phase_end(lp, ph, lp->stream);
      if (st) return st;
    }
    return MUGIQ_HIP_SUCCESS;
  }
  // Lengths that do not start at 1 ("-x:3"): the matrix-pipe tile cannot build its axial gauge from the links of the call (W_start ..
  // W_stop), the driver can (it holds W_1 .. W_stop) and says so for the launches below.
  struct GaugeScope {
    bool on = false;
    ~GaugeScope() {
      if (on) set_axial_gauge_hint(nullptr, nullptr, -1, -1, 0);
    }
  } gaugeScope;
  if (directGauge) {
    set_axial_gauge_hint(directGauge, links[0], dir, sign, stop);
    gaugeScope.on = true;
  } else if (start > 1) {
    const size_t gb = axial_gauge_bytes(lp->eVecs[0], dir, kv.data(), (int)kv.size(), part ? 1 : 0);
    if (gb) {
      void *G = nullptr;
      std::vector<const void *> lk;
      for (int k = 1; k <= stop; k++) lk.push_back(E[k].data);
      if ((st = scratch_alloc(lp, &G, gb, false))) return st;
      if ((st = build_axial_gauge(G, lp->eVecs[0], lk.data(), stop, dir, sign, lp->stream))) return st;
      set_axial_gauge_hint(G, links[0], dir, sign, stop);
      gaugeScope.on = true;
    }
  }
  // eigenvector blocks: bounded by the ghost-layer buffers when the dimension is partitioned
  int nb = lp->nEv;
  void *gsend = nullptr, *grecv = nullptr;
  size_t perVec = 0;
  if (part) {
    perVec = (size_t)stop * 24 * (lp->volumeCB / lp->localL[dir]) * lp->cplxBytes();
    const size_t budget = (size_t)4 << 30;  // 4 GiB per direction buffer
    nb = (int)std::max<size_t>(1, std::min<size_t>((size_t)lp->nEv, budget / perVec));
    if ((st = scratch_alloc(lp, &grecv, perVec * nb, false))) return st;
    if (self_neighbour_alias(lp, dir)) gsend = grecv;  // packed straight into the ghost buffer, no message
    else if ((st = scratch_alloc(lp, &gsend, perVec * nb, false))) return st;
  }
  if (part && (st = ensure_comm_stream(lp))) return st;
  for (int n0 = 0; n0 < lp->nEv; n0 += nb) {
    const int nv = std::min(nb, lp->nEv - n0);
    // the slots were not zeroed (see mugiq_hip_loop_compute): the first eigenvector block writes them, later ones add
    const int ow = n0 == 0 ? MUGIQ_HIP_REGION_OVERWRITE : 0;
    if (!part) {
      // (one block here: nb = nEv.)  The first such entry also takes the ultra-local loop along, if the kernel has room
      int carried = 0;
      void *ultra = (lp->carryUltra && !lp->ultraCarried && nb == lp->nEv) ? lp->dataPos_d : nullptr;
      if ((st = mugiq_hip_displaced_loop_contraction_fused_carry(slot0, lp->loopPrecision, &lp->eVecs[n0], &lp->sigma[n0], nv,
                                                                 links.data(), kv.data(), (int)kv.size(), dir, sign, lp->commDim,
                                                                 nullptr, 0, MUGIQ_HIP_REGION_ALL | ow, ultra, &carried, lp->stream)))
        return st;
      if (carried) {
        lp->ultraCarried = true;
        lp->ultraCarrier = id;
      }
      continue;
    }
    // pack the face layers -> [comm stream] exchange them  ||  [compute stream] interior sites -> boundary sites
    const int high = (sign == MUGIQ_HIP_DISP_SIGN_PLUS) ? 0 : 1;
    if ((st = mugiq_hip_pack_face_layers(gsend, &lp->eVecs[n0], nv, dir, high, stop, lp->stream))) return st;
    MUGIQ_CHECK_HIP(hipEventRecord(lp->evPacked, lp->stream));
    MUGIQ_CHECK_HIP(hipStreamWaitEvent(lp->commStream, lp->evPacked, 0));
    int ph = phase_begin(lp, MUGIQ_HIP_PHASE_HALO_TRANSFER, id, lp->commStream, (double)(perVec * nv));
    if (gsend != grecv) {
      st = lp->comm.sendrecv(lp->comm.ctx, gsend, grecv, perVec * nv, dir, high ? +1 : -1, lp->commStream);
      if (st) return set_error(MUGIQ_HIP_ERROR_HIP, "halo sendrecv callback failed with status %d", st);
    }
    phase_end(lp, ph, lp->commStream);
    MUGIQ_CHECK_HIP(hipEventRecord(lp->evHalo, lp->commStream));
    ph = phase_begin(lp, MUGIQ_HIP_PHASE_ENTRY_INTERIOR, id, lp->stream);
    if ((st = mugiq_hip_displaced_loop_contraction_fused_region(slot0, lp->loopPrecision, &lp->eVecs[n0], &lp->sigma[n0], nv,
                                                                links.data(), kv.data(), (int)kv.size(), dir, sign, lp->commDim,
                                                                grecv, stop, MUGIQ_HIP_REGION_INTERIOR | ow, lp->stream)))
      return st;
    phase_end(lp, ph, lp->stream);
    ph = phase_begin(lp, MUGIQ_HIP_PHASE_HALO_WAIT, id, lp->stream);
    MUGIQ_CHECK_HIP(hipStreamWaitEvent(lp->stream, lp->evHalo, 0));
    phase_end(lp, ph, lp->stream);
    ph = phase_begin(lp, MUGIQ_HIP_PHASE_ENTRY_BOUNDARY, id, lp->stream);
    if ((st = mugiq_hip_displaced_loop_contraction_fused_region(slot0, lp->loopPrecision, &lp->eVecs[n0], &lp->sigma[n0], nv,
                                                                links.data(), kv.data(), (int)kv.size(), dir, sign, lp->commDim,
                                                                grecv, stop, MUGIQ_HIP_REGION_BOUNDARY | ow, lp->stream)))
      return st;
    phase_end(lp, ph, lp->stream);
  }
  return MUGIQ_HIP_SUCCESS;
}

// Reflected entries (csrc/reflect.hip): entry `id` can be derived from an entry `jd` computed earlier in this run if jd
// has the same direction, the opposite sign and covers id's lengths.  Returns the source entry or -1.
static int reflection_source(const MugiqHipLoop *lp, int id) {
  if (const char *e = getenv("MUGIQ_HIP_REFLECT"))
    if (atoi(e) == 0) return -1;
  // a length that reaches past the nearest neighbour cannot be served by one halo of the source slot
  if (lp->commDim[lp->dispDir[id]] && lp->dispStop[id] > lp->localL[lp->dispDir[id]]) return -1;
  for (int jd = 0; jd < id; jd++)
    if (lp->dispDir[jd] == lp->dispDir[id] && lp->dispSign[jd] != lp->dispSign[id] && lp->dispStart[jd] <= lp->dispStart[id] &&
        lp->dispStop[id] <= lp->dispStop[jd] && lp->derivedFrom[jd] < 0)
      return jd;
  return -1;
}

static int entry_reflected(MugiqHipLoop *lp, int id, int jd, void *slot0) {
  const int dir = lp->dispDir[id], sign = lp->dispSign[id];
  const bool part = lp->commDim[dir] != 0;
  const size_t slotBytes = (size_t)lp->nElemPosLocPerLoop * lp->loopBytes();
  const char *src0 = static_cast<const char *>(lp->dataPos_d) + slotBytes * (size_t)lp->nLoopOffset[jd];
  const int faceCB = lp->volumeCB / lp->localL[dir];
  int st;
  // a length that reaches past the nearest neighbour cannot be served by one halo: nothing is written, the caller
  // computes the entry from the eigenvectors
  if (part && lp->dispStop[id] > lp->localL[dir]) return -1;
  for (int k = lp->dispStart[id]; k <= lp->dispStop[id]; k++) {
    void *dst = static_cast<char *>(slot0) + slotBytes * (size_t)(k - lp->dispStart[id]);
    const void *src = src0 + slotBytes * (size_t)(k - lp->dispStart[jd]);
    void *grecv = nullptr;
    if (part) {
      // dst "-": the source sites x - k mu below my block are the backward neighbour's top k layers, so every rank sends its
      // top layers forward; dst "+": bottom layers backward
      const int high = sign == MUGIQ_HIP_DISP_SIGN_MINUS ? 1 : 0;
      const size_t bytes = (size_t)32 * k * faceCB * lp->loopBytes();
      void *gsend = nullptr;
      if ((st = scratch_alloc(lp, &gsend, bytes, false))) return st;
      if ((st = scratch_alloc(lp, &grecv, bytes, false))) return st;
      if ((st = mugiq_hip_pack_loop_layers(gsend, src, lp->localL, dir, high, k, lp->loopPrecision, lp->stream))) return st;
      st = lp->comm.sendrecv(lp->comm.ctx, gsend, grecv, bytes, dir, high ? +1 : -1, lp->stream);
      if (st) return set_error(MUGIQ_HIP_ERROR_HIP, "halo sendrecv callback failed with status %d", st);
    }
    if ((st = mugiq_hip_reflect_displaced_loop(dst, src, grecv, lp->localL, dir, sign, k, lp->commDim, lp->loopPrecision, lp->stream)))
      return st;
  }
  return MUGIQ_HIP_SUCCESS;
}

// hand the current entry's buffers back to the pool (the stream has been synchronised by the caller)
static void free_scratch(MugiqHipLoop *lp) {
  for (void *p : lp->scratch)
    for (auto &b : lp->pool)
      if (b.ptr == p) b.inUse = false;
  lp->scratch.clear();
}
static void destroy_pool(MugiqHipLoop *lp) {
  for (auto &b : lp->pool) (void)hipFree(b.ptr);
  lp->pool.clear();
  lp->scratch.clear();
}

// the position-space slots of the reflected entries a compute left out (momentum-space reflection): produced on first request
static int materialise_reflected(MugiqHipLoop *lp) {
  if (!lp->posReflectPending) return MUGIQ_HIP_SUCCESS;
  const size_t cb = lp->loopBytes();
  int st = MUGIQ_HIP_SUCCESS;
  for (int id = 0; id < lp->nDispEntries && !st; id++) {
    if (lp->derivedFrom[id] < 0) continue;
    void *slot0 = static_cast<char *>(lp->dataPos_d) + (size_t)lp->nElemPosLocPerLoop * lp->nLoopOffset[id] * cb;
    st = entry_reflected(lp, id, lp->derivedFrom[id], slot0);
    free_scratch(lp);
  }
  hipError_t e = hipStreamSynchronize(lp->stream);
  if (!st && e != hipSuccess) st = set_error(MUGIQ_HIP_ERROR_HIP, "dataPos: %s", hipGetErrorString(e));
  if (!st) lp->posReflectPending = false;
  return st;
}

// can the fused reorder + x step take this lattice?  (else: reorder, then the three separable steps)
static bool fused_projection_applies(const MugiqHipLoop *lp) {
  std::vector<int> px;
  for (int n = 0; n < lp->Nmom; n++)
    if (std::find(px.begin(), px.end(), lp->momMatrix[3 * n]) == px.end()) px.push_back(lp->momMatrix[3 * n]);
  return eo_dft_x_time_chunk(lp->loopPrecision, lp->localL, (int)px.size()) >= 1 && lp->localL[2] <= 65535 && lp->nData <= 65535;
}

// Loop_Mugiq::performMomentumProjection  lib/loop_mugiq.cpp:322-434
static int momentum_projection(MugiqHipLoop *lp) {
  if (lp->momProjDone) return set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "performMomentumProjection: Not supposed to be called more than once!!");
  int st;
  std::vector<int> activeSlots;  // momentum-space reflection: the loop slots that go through the projection
  const int phDev = phase_begin(lp, MUGIQ_HIP_PHASE_MOMENTUM_PROJECTION, -1, lp->stream);
  if (lp->calcType == MUGIQ_HIP_LOOP_CALC_TYPE_BASIC_KERNEL) {
    // the reference's sequence: reorder (:343-344), then one dense product with the phase matrix of createPhaseMatrixGPU (:363-378)
    if (!lp->dataPosMP_d && (st = dev_alloc(lp, &lp->dataPosMP_d, (size_t)lp->nElemPosLoc * lp->loopBytes(), true))) return st;
    if ((st = mugiq_hip_convert_idx_order_map_gamma(lp->dataPosMP_d, lp->dataPos_d, lp->nData, lp->nLoop, 2, lp->volumeCB,
                                                    lp->localL, lp->loopPrecision, lp->stream)))
      return st;
    if ((st = mugiq_hip_momentum_projection(lp->dataMom_d, lp->dataPosMP_d, lp->phaseMatrix_d, lp->locT, lp->nData, lp->locV3,
                                            lp->Nmom, lp->loopPrecision, nullptr, 0, lp->stream)))
      return st;
  } else {
    // OPT: reorder + gamma5 map + sum over x in one kernel, then the y and z sums (csrc/momproj.hip); no reordered copy
    int coord[4] = {0, 0, 0, 0};
    if (lp->haveComm)
      for (int d = 0; d < 4; d++) coord[d] = lp->comm.coord[d];
    if (lp->momReflect) {
      // only the slots computed from the eigenvectors are transformed; the reflected ones follow on the gathered array below
      for (int id = -1; id < lp->nDispEntries; id++) {
        if (id >= 0 && lp->derivedFrom[id] >= 0) continue;
        const int first = id < 0 ? 0 : lp->nLoopOffset[id], cnt = id < 0 ? 1 : lp->nLoopPerEntry[id];
        for (int i = 0; i < cnt; i++) activeSlots.push_back(first + i);
      }
      // (the rows of the reflected slots in dataMom_d stay zero -- allocated zeroed, never written -- until they are derived)
      if ((st = mugiq_hip_convert_and_project_slots(lp->dataMom_d, lp->dataPos_d, lp->nLoop, activeSlots.data(), (int)activeSlots.size(),
                                                    lp->momMatrix.data(), lp->Nmom, lp->FTSign, lp->localL, lp->totalL, coord,
                                                    lp->loopPrecision, nullptr, 0, lp->stream)))
        return st;
    } else if (fused_projection_applies(lp)) {
      if ((st = mugiq_hip_convert_and_project(lp->dataMom_d, lp->dataPos_d, lp->nData, lp->nLoop, lp->momMatrix.data(), lp->Nmom, lp->FTSign,
                                              lp->localL, lp->totalL, coord, lp->loopPrecision, nullptr, 0, lp->stream)))
        return st;
    } else {  // a lattice whose (x, t) rows do not fit the LDS tile: reorder, then the three separable steps
      if (!lp->dataPosMP_d && (st = dev_alloc(lp, &lp->dataPosMP_d, (size_t)lp->nElemPosLoc * lp->loopBytes(), true))) return st;
      if ((st = mugiq_hip_convert_idx_order_map_gamma(lp->dataPosMP_d, lp->dataPos_d, lp->nData, lp->nLoop, 2, lp->volumeCB,
                                                      lp->localL, lp->loopPrecision, lp->stream)))
        return st;
      if ((st = mugiq_hip_momentum_projection_separable(lp->dataMom_d, lp->dataPosMP_d, lp->momMatrix.data(), lp->Nmom, lp->FTSign,
                                                        lp->localL, lp->totalL, coord, lp->locT, lp->nData, lp->loopPrecision, nullptr,
                                                        0, lp->stream)))
        return st;
    }
  }
  const size_t locBytes = (size_t)lp->nElemMomLoc * lp->loopBytes();
  phase_end(lp, phDev, lp->stream);
  const int phCopy = phase_begin(lp, MUGIQ_HIP_PHASE_MOMENTUM_COPY, -1, lp->stream, (double)locBytes);
  MUGIQ_CHECK_HIP(hipMemcpyAsync(lp->dataMom_h, lp->dataMom_d, locBytes, hipMemcpyDeviceToHost, lp->stream));  // :386
  phase_end(lp, phCopy, lp->stream);
  MUGIQ_CHECK_HIP(hipStreamSynchronize(lp->stream));
  const auto tRed0 = std::chrono::steady_clock::now();
  if (lp->haveComm && lp->comm.size > 1) {
    const size_t nReal = 2 * (size_t)lp->nElemMomLoc;
    if ((st = lp->comm.reduce_space(lp->comm.ctx, lp->dataMom_h, lp->dataMom, nReal, lp->loopPrecision)))  // :406
      return set_error(MUGIQ_HIP_ERROR_HIP, "reduce_space callback failed with status %d", st);
    if ((st = lp->comm.gather_time(lp->comm.ctx, lp->dataMom, lp->dataMom_bcast, nReal, lp->loopPrecision)))  // :420-422
      return set_error(MUGIQ_HIP_ERROR_HIP, "gather_time callback failed with status %d", st);
    if ((st = lp->comm.bcast(lp->comm.ctx, lp->dataMom_bcast, 2 * (size_t)lp->nElemMomTot, lp->loopPrecision)))  // :424
      return set_error(MUGIQ_HIP_ERROR_HIP, "bcast callback failed with status %d", st);
  }  // (one process: dataMom and dataMom_bcast alias dataMom_h)
  phase_host(lp, MUGIQ_HIP_PHASE_MOMENTUM_REDUCE, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tRed0).count(),
             2.0 * (double)lp->nElemMomLoc * lp->loopPrecision);
  if (lp->momReflect) {
    // the reflected entries, on the gathered array (every rank holds it after the broadcast): csrc/reflect_mom.cpp
    const auto tRef0 = std::chrono::steady_clock::now();
    struct Task {
      int dst, src, dir, sign, k;
    };
    std::vector<Task> tasks;
    for (int id = 0; id < lp->nDispEntries; id++) {
      const int jd = lp->derivedFrom[id];
      if (jd < 0) continue;
      for (int k = lp->dispStart[id]; k <= lp->dispStop[id]; k++)
        tasks.push_back({lp->nLoopOffset[id] + k - lp->dispStart[id], lp->nLoopOffset[jd] + k - lp->dispStart[jd], lp->dispDir[id], lp->dispSign[id], k});
    }
    // the slots are independent (distinct destinations, sources only read): one host thread per slot, up to the cores there are
    // (a slot is ~50k complex numbers scattered over the 20 MB array: 0.2 ms each when taken one after the other)
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t nThreads = std::min<size_t>(std::min<size_t>(tasks.size(), hw), 16);
    std::vector<int> status(nThreads, 0);
    std::vector<std::string> message(nThreads);
    auto work = [&](size_t w) {
      for (size_t i = w; i < tasks.size() && !status[w]; i += nThreads) {
        const Task &t = tasks[i];
        status[w] = mugiq_hip_reflect_momentum_space(lp->dataMom_bcast, lp->loopPrecision, lp->Nmom, lp->momMatrix.data(), lp->FTSign, lp->totalL,
                                                     lp->nLoop, lp->locT, lp->totT, t.dst, t.src, t.dir, t.sign, t.k);
        if (status[w]) message[w] = mugiq_hip_last_error();  // (the error text is per thread)
      }
    };
    std::vector<std::thread> pool;
    for (size_t w = 1; w < nThreads; w++) pool.emplace_back(work, w);
    if (nThreads) work(0);
    for (auto &th : pool) th.join();
    for (size_t w = 0; w < nThreads; w++)
      if (status[w]) return set_error(status[w], "%s", message[w].c_str());
    phase_host(lp, MUGIQ_HIP_PHASE_MOMENTUM_REFLECT, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tRef0).count());
  }
  lp->momProjDone = true;
  return MUGIQ_HIP_SUCCESS;
}

}  // namespace mugiq

extern "C" {

int mugiq_hip_parse_displacement(const char *disp_str, int *dir_out, int *sign_out) {
  MUGIQ_REQUIRE(dir_out && sign_out, "mugiq_hip_parse_displacement: NULL output");
  return parse_displacement(disp_str, dir_out, sign_out);
}

// tests/loop.cpp:607-705 (ParseDispEntry with ';' and ':', ParseDispLimits with ',')
int mugiq_hip_parse_displace_entry_string(const char *entry_string, int max_entries, char *disp_str_out, int *disp_start_out,
                                          int *disp_stop_out) {
  if (!entry_string || !disp_str_out || !disp_start_out || !disp_stop_out)
    return -set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "setLoopParam: NULL argument");
  std::string all(entry_string);
  if (all.empty()) return -set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "Got option '--loop-do-nonlocal yes' but option --displace-entry-string is not set!");
  int n = 0;
  size_t pos = 0;
  while (true) {
    size_t semi = all.find(';', pos);
    std::string entry = all.substr(pos, semi == std::string::npos ? std::string::npos : semi - pos);
    size_t colon = entry.find(':');
    if (colon == std::string::npos || entry.find(':', colon + 1) != std::string::npos)
      return -set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT,
                        "Displacement entry %d has the Wrong format. Example of good entries: +z:1,8 , +x:3", n);
    std::string dstr = entry.substr(0, colon), lim = entry.substr(colon + 1);
    std::vector<int> lims;
    size_t lp = 0;
    while (true) {
      size_t comma = lim.find(',', lp);
      std::string tok = lim.substr(lp, comma == std::string::npos ? std::string::npos : comma - lp);
      char *end = nullptr;
      long v = strtol(tok.c_str(), &end, 10);
      if (tok.empty() || end == tok.c_str())
        return -set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "Wrong format of displacement entry %d. Example of good entries: +z:1,8 , +x:3", n);
      lims.push_back((int)v);
      if (comma == std::string::npos) break;
      lp = comma + 1;
    }
    if (lims.empty() || lims.size() > 2)
      return -set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "Wrong format of displacement entry %d. Example of good entries: +z:1,8 , +x:3", n);
    if (n >= max_entries) return -set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "more than %d displacement entries", max_entries);
    if (dstr.size() > 3) return -set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "displacement string '%s' too long", dstr.c_str());
    strncpy(disp_str_out + 4 * n, dstr.c_str(), 4);
    disp_str_out[4 * n + 3] = '\0';
    disp_start_out[n] = lims[0];
    disp_stop_out[n] = lims.size() == 2 ? lims[1] : lims[0];
    n++;
    if (semi == std::string::npos) break;
    pos = semi + 1;
  }
  return n;
}

int mugiq_hip_loop_create(MugiqHipLoop **out, const MugiqHipLoopParam *p, const MugiqHipSpinorField *eVecs_h,
                          const double *eVals_sigma_h, int nEv, const MugiqHipComm *comm, void *stream) {
  const char *who = "Loop_Mugiq";
  MUGIQ_REQUIRE(out && p && eVecs_h && eVals_sigma_h, "%s: NULL argument", who);
  MUGIQ_REQUIRE(nEv >= 1, "%s: nEv = %d must be >= 1", who, nEv);
  *out = nullptr;
  int st;
  for (int n = 0; n < nEv; n++) {
    if ((st = validate_spinor(&eVecs_h[n], who, "eVecs"))) return st;
    MUGIQ_REQUIRE(same_geometry(eVecs_h[n], eVecs_h[0]), "%s: eigenvector %d differs in precision, order or geometry from eVecs[0]", who, n);
    MUGIQ_REQUIRE(eVals_sigma_h[n] != 0.0, "%s: eVals_sigma[%d] is zero", who, n);
  }
  MugiqHipLoop *lp = new MugiqHipLoop_s();
  auto fail = [&](int code) {
    mugiq_hip_loop_destroy(lp);
    return code;
  };
  lp->stream = static_cast<hipStream_t>(stream);
  lp->eVecs.assign(eVecs_h, eVecs_h + nEv);
  lp->sigma.assign(eVals_sigma_h, eVals_sigma_h + nEv);
  lp->nEv = nEv;
  lp->precision = eVecs_h[0].precision;
  lp->loopPrecision = p->loopPrecision ? p->loopPrecision : lp->precision;
  if (!(lp->loopPrecision == lp->precision || (lp->loopPrecision == 8 && lp->precision == 4))) {
    set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "%s: loopPrecision %d with eigenvector precision %d is not supported", who,
              lp->loopPrecision, lp->precision);
    return fail(MUGIQ_HIP_ERROR_INVALID_ARGUMENT);
  }
  lp->order = eVecs_h[0].field_order;
  lp->volumeCB = eVecs_h[0].volumeCB;
  if (comm) {
    lp->comm = *comm;
    lp->haveComm = true;
    long long prod = 1;
    bool anyPart = false;
    for (int d = 0; d < 4; d++) {
      if (comm->grid[d] < 1 || comm->coord[d] < 0 || comm->coord[d] >= comm->grid[d]) {
        set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "%s: invalid comm grid/coord in dim %d", who, d);
        return fail(MUGIQ_HIP_ERROR_INVALID_ARGUMENT);
      }
      lp->commDim[d] = comm_partitioned(comm, d);
      anyPart = anyPart || lp->commDim[d];
      prod *= comm->grid[d];
    }
    if (prod != comm->size || (comm->size > 1 && (!comm->sendrecv || !comm->reduce_space || !comm->gather_time || !comm->bcast)) ||
        (anyPart && !comm->sendrecv)) {
      set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "%s: comm grid does not match comm size %d, or a callback is NULL", who, comm->size);
      return fail(MUGIQ_HIP_ERROR_INVALID_ARGUMENT);
    }
  }
  // ---- LoopComputeParam constructor, include/loop_mugiq.h:185-261
  lp->Nmom = p->Nmom;
  lp->FTSign = p->FTSign;
  lp->calcType = p->calcType;
  lp->doMomProj = p->doMomProj != 0;
  lp->doNonLocal = p->doNonLocal != 0;
  lp->writeMom = p->writeMomSpaceHDF5 != 0;
  lp->writePos = p->writePosSpaceHDF5 != 0;
  lp->fnameMom = p->fname_mom_h5 ? p->fname_mom_h5 : "";
  lp->fnamePos = p->fname_pos_h5 ? p->fname_pos_h5 : "";
  for (int i = 0; i < 4; i++) {
    lp->localL[i] = eVecs_h[0].X[i];
    lp->totalL[i] = lp->localL[i] * (lp->haveComm ? lp->comm.grid[i] : 1);
    lp->locV4 *= lp->localL[i];
    if (i < 3) {
      lp->locV3 *= lp->localL[i];
      lp->totV3 *= lp->totalL[i];
    }
  }
  lp->locT = lp->localL[3];
  lp->totT = lp->totalL[3];
  if (lp->doMomProj) {
    if (p->Nmom < 1 || !p->momMatrix || (p->FTSign != 1 && p->FTSign != -1)) {
      set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "%s: doMomProj needs Nmom >= 1, a momentum matrix and FTSign = +-1 (Loop FT sign is undefined/unsupported)", who);
      return fail(MUGIQ_HIP_ERROR_INVALID_ARGUMENT);
    }
    lp->momMatrix.assign(p->momMatrix, p->momMatrix + 3 * (size_t)p->Nmom);
  }
  if (lp->doNonLocal) {
    lp->nDispEntries = p->nDispEntries;
    if (p->nDispEntries < 0 || (p->nDispEntries > 0 && (!p->disp_str || !p->disp_start || !p->disp_stop))) {
      set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "Displacement string length not compatible with displacement limits length");
      return fail(MUGIQ_HIP_ERROR_INVALID_ARGUMENT);
    }
    for (int id = 0; id < lp->nDispEntries; id++) {
      lp->dispEntry.push_back(p->disp_entry && p->disp_entry[id] ? p->disp_entry[id] : "");
      lp->dispString.push_back(p->disp_str[id] ? p->disp_str[id] : "");
      int a = p->disp_start[id], b = p->disp_stop[id];
      if (a > b) std::swap(a, b);  // "Stop length is smaller than Start length ... Will switch lengths!"  :234-239
      if (a < 1) {
        set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "%s: displacement lengths must be >= 1 (entry %d: %d)", who, id, a);
        return fail(MUGIQ_HIP_ERROR_INVALID_ARGUMENT);
      }
      lp->dispStart.push_back(a);
      lp->dispStop.push_back(b);
      lp->nLoopPerEntry.push_back(b - a + 1);
      lp->nLoop += b - a + 1;
      int osum = 1;  // start with ultra-local
      for (int is = 0; is < id; is++) osum += lp->nLoopPerEntry[is];
      lp->nLoopOffset.push_back(osum);
      int dir, sign;
      if ((st = parse_displacement(lp->dispString[id].c_str(), &dir, &sign))) return fail(st);  // Displace::setupDisplacement
      lp->dispDir.push_back(dir);
      lp->dispSign.push_back(sign);
      lp->derivedFrom.push_back(-1);
    }
    lp->nLoop += 1;  // Don't forget ultra-local case!!
    if (lp->nDispEntries > 0) {
      if (!p->gauge) {
        set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "%s: doNonLocal needs the (extended) gauge field for the displacements", who);
        return fail(MUGIQ_HIP_ERROR_INVALID_ARGUMENT);
      }
      lp->gauge = *p->gauge;
      lp->haveGauge = true;
      for (int d = 0; d < 4; d++)
        if (lp->commDim[d] && lp->gauge.R[d] < 1) {
          set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "%s: dim %d is partitioned but the gauge border R[%d] = %d (the reference uses 2, lib/displace.cpp:16)", who, d, d, lp->gauge.R[d]);
          return fail(MUGIQ_HIP_ERROR_INVALID_ARGUMENT);
        }
    }
  } else {
    lp->nDispEntries = 0;
    lp->nLoop = 1;
  }
  lp->nData = lp->nLoop * 16;
  // ---- allocateDataMemory, lib/loop_mugiq.cpp:101-158
  lp->nElemMomTotPerLoop = 16LL * lp->Nmom * lp->totT;
  lp->nElemMomLocPerLoop = 16LL * lp->Nmom * lp->locT;
  lp->nElemPosLocPerLoop = 16LL * lp->locV4;
  lp->nElemMomTot = lp->nElemMomTotPerLoop * lp->nLoop;
  lp->nElemMomLoc = lp->nElemMomLocPerLoop * lp->nLoop;
  lp->nElemPosLoc = lp->nElemPosLocPerLoop * lp->nLoop;
  lp->nElemPhMat = (long long)lp->Nmom * lp->locV3;
  const size_t cb = lp->loopBytes();
  if ((st = dev_alloc(lp, &lp->dataPos_d, (size_t)lp->nElemPosLoc * cb, true))) return fail(st);
  if (lp->doMomProj) {
    // the device -> host landing buffer is pinned (a pageable copy of ~20 MB costs several ms, more than the projection)
    if (hipHostMalloc(&lp->dataMom_h, (size_t)lp->nElemMomLoc * cb, hipHostMallocDefault) != hipSuccess) lp->dataMom_h = nullptr;
    else memset(lp->dataMom_h, 0, (size_t)lp->nElemMomLoc * cb);
    if (lp->haveComm && lp->comm.size > 1) {
      lp->dataMom_bcast = calloc((size_t)lp->nElemMomTot, cb);
      lp->dataMom = calloc((size_t)lp->nElemMomLoc, cb);
    } else {
      // one process: the reduced (dataMom) and the gathered + broadcast (dataMom_bcast) arrays ARE the local one -- aliases
      // instead of two more copies through freshly mapped pages (6 ms for 19 MB)
      lp->dataMom = lp->dataMom_h;
      lp->dataMom_bcast = lp->dataMom_h;
    }
    if (!lp->dataMom_bcast || !lp->dataMom_h || !lp->dataMom) {
      set_error(MUGIQ_HIP_ERROR_HIP, "%s: Could not allocate host buffers dataMom*", who);
      return fail(MUGIQ_HIP_ERROR_HIP);
    }
    if ((st = dev_alloc(lp, &lp->phaseMatrix_d, (size_t)lp->nElemPhMat * cb, true))) return fail(st);
    if ((st = dev_alloc(lp, &lp->dataMom_d, (size_t)lp->nElemMomLoc * cb, true))) return fail(st);
    // dataPosMP_d (the reordered copy of the loop data, lib/loop_mugiq.cpp:143) is allocated on first use: only the BASIC
    // plan forms it
  }
  // copyGammaToConstMem :162-167, createPhaseMatrix :171-178
  if ((st = mugiq_hip_copy_gamma_coeff_to_symbol(lp->precision))) return fail(st);
  if (lp->doMomProj) {
    if ((st = mugiq_hip_copy_gamma_map_to_symbol(lp->loopPrecision))) return fail(st);
    int cc[4] = {0, 0, 0, 0};
    if (lp->haveComm)
      for (int d = 0; d < 4; d++) cc[d] = lp->comm.coord[d];
    if ((st = mugiq_hip_create_phase_matrix(lp->phaseMatrix_d, lp->momMatrix.data(), lp->locV3, lp->Nmom, lp->FTSign, lp->localL,
                                            lp->totalL, cc, lp->loopPrecision, lp->stream)))
      return fail(st);
  }
  // the buffers the OPT plan will hold for the halos it posts ahead (the large allocations of a partitioned run)
  if (lp->haveGauge && (st = reserve_plan_buffers(lp))) return fail(st);
  *out = lp;
  return MUGIQ_HIP_SUCCESS;
}

// Loop_Mugiq with eigsolve->useMGenv && eigsolve->computeCoarse (lib/loop_mugiq.cpp:42,482): the eigenvectors live on
// the coarsest grid and are prolonged through the MG transfer operators before they are contracted
// (prolongateEvec, lib/loop_mugiq.cpp:277-319: transfer[lev-1]->P for lev = nCoarseLevels .. 2, then transfer[0]->P).
int mugiq_hip_loop_create_coarse_levels(MugiqHipLoop **out, const MugiqHipLoopParam *p, const MugiqHipCoarseField *coarseEvecs_h,
                                        const double *eVals_sigma_h, int nEv, const MugiqHipTransfer *transfers_h, int nCoarseLevels,
                                        int fineFieldOrder, const MugiqHipComm *comm, void *stream) {
  const char *who = "Loop_Mugiq(coarse)";
  MUGIQ_REQUIRE(out && p && coarseEvecs_h && eVals_sigma_h && transfers_h, "%s: NULL argument", who);
  MUGIQ_REQUIRE(nEv >= 1, "%s: nEv = %d must be >= 1", who, nEv);
  MUGIQ_REQUIRE(nCoarseLevels >= 1 && nCoarseLevels <= 4, "%s: nCoarseLevels = %d must be in [1, 4] (QUDA_MAX_MG_LEVEL - 1)", who, nCoarseLevels);
  // the reference insists on FLOAT2 for the MG-coarse path (lib/loop_mugiq.cpp:283, lib/interface_mugiq.cpp:226-230)
  MUGIQ_REQUIRE(fineFieldOrder == 2, "%s: Vector prolongation requires fieldOrder = FLOAT2", who);
  const MugiqHipTransfer *transfer = &transfers_h[0];
  MUGIQ_REQUIRE(transfer->V && (transfer->precision == 4 || transfer->precision == 8), "%s: Transfer operator for finest level does not exist!", who);
  for (int l = 1; l < nCoarseLevels; l++) {
    MUGIQ_REQUIRE(transfers_h[l].V, "%s: Transfer operator for level %d does not exist!", who, l + 1);  // lib/loop_mugiq.cpp:309
    MUGIQ_REQUIRE(transfers_h[l].precision == transfer->precision, "%s: transfer level %d differs in precision", who, l);
    for (int d = 0; d < 4; d++)
      MUGIQ_REQUIRE(transfers_h[l].X[d] * transfers_h[l - 1].geoBlockSize[d] == transfers_h[l - 1].X[d],
                    "%s: transfer level %d: X[%d] = %d is not level %d's X / geo_block_size", who, l, d, transfers_h[l].X[d], l - 1);
  }
  *out = nullptr;
  long long vol = 1;
  for (int d = 0; d < 4; d++) {
    MUGIQ_REQUIRE(transfer->X[d] > 0 && (transfer->X[d] & 1) == 0, "%s: fine X[%d] = %d must be positive and even", who, d, transfer->X[d]);
    vol *= transfer->X[d];
  }
  const int volumeCB = (int)(vol / 2);
  const bool needFine = (p->doNonLocal && p->nDispEntries > 0) || p->calcType == MUGIQ_HIP_LOOP_CALC_TYPE_BASIC_KERNEL;
  const size_t fieldBytes = (size_t)24 * volumeCB * 2 * (size_t)transfer->precision;
  void *store = nullptr;
  if (needFine) MUGIQ_CHECK_HIP(hipMalloc(&store, fieldBytes * (size_t)nEv));
  std::vector<MugiqHipSpinorField> fine(nEv);
  for (int n = 0; n < nEv; n++) {
    MugiqHipSpinorField f{};
    // without fine storage the descriptors only carry the geometry: the ultra-local loop runs through the fused
    // prolong-contract kernel and never dereferences them
    f.data = needFine ? static_cast<char *>(store) + fieldBytes * (size_t)n : reinterpret_cast<void *>(uintptr_t(16));
    f.precision = transfer->precision;
    f.field_order = fineFieldOrder;
    f.nParity = 2;
    f.volumeCB = volumeCB;
    f.stride = volumeCB;
    f.parity_offset = (int64_t)12 * volumeCB;
    for (int d = 0; d < 4; d++) f.X[d] = transfer->X[d];
    fine[n] = f;
  }
  MugiqHipLoop *lp = nullptr;
  int st = mugiq_hip_loop_create(&lp, p, fine.data(), eVals_sigma_h, nEv, comm, stream);
  if (st) {
    if (store) (void)hipFree(store);
    return st;
  }
  lp->coarseMode = true;
  lp->transfer = *transfer;
  lp->fineStore = store;
  // levelVecs[l] = eigenvectors on level l + 1; the input sits on the coarsest one, the others are owned temporaries
  // (tmpCSF[1 .. nCoarseLevels-1] of the reference, allocated once instead of per eigenvector and call)
  lp->levelVecs.resize(nCoarseLevels);
  lp->levelStore.assign(nCoarseLevels, nullptr);
  lp->levelVecs[nCoarseLevels - 1].assign(coarseEvecs_h, coarseEvecs_h + nEv);
  for (int l = nCoarseLevels - 2; l >= 0; l--) {
    const MugiqHipTransfer &T = transfers_h[l + 1];  // between level l+1 (finer side, dims T.X) and level l+2
    lp->upper.insert(lp->upper.begin(), T);
    MugiqHipCoarseField f{};
    f.precision = T.precision;
    f.nSpin = 2;
    f.nColor = transfers_h[l].nVec;
    long long v = 1;
    for (int d = 0; d < 4; d++) {
      f.X[d] = T.X[d];
      v *= T.X[d];
    }
    f.volumeCB = (int)(v / 2);
    f.stride = f.volumeCB;
    f.parity_offset = (int64_t)2 * f.nColor * f.stride;
    const size_t bytes = (size_t)2 * f.parity_offset * 2 * (size_t)T.precision;
    if (hipMalloc(&lp->levelStore[l], bytes * (size_t)nEv) != hipSuccess) {
      mugiq_hip_loop_destroy(lp);
      return set_error(MUGIQ_HIP_ERROR_HIP, "%s: could not allocate the eigenvectors of coarse level %d", who, l + 1);
    }
    for (int n = 0; n < nEv; n++) {
      f.data = static_cast<char *>(lp->levelStore[l]) + bytes * (size_t)n;
      lp->levelVecs[l].push_back(f);
    }
  }
  lp->coarseVecs = lp->levelVecs[0];
  *out = lp;
  return MUGIQ_HIP_SUCCESS;
}

int mugiq_hip_loop_create_coarse(MugiqHipLoop **out, const MugiqHipLoopParam *p, const MugiqHipCoarseField *coarseEvecs_h,
                                 const double *eVals_sigma_h, int nEv, const MugiqHipTransfer *transfer, int fineFieldOrder,
                                 const MugiqHipComm *comm, void *stream) {
  return mugiq_hip_loop_create_coarse_levels(out, p, coarseEvecs_h, eVals_sigma_h, nEv, transfer, 1, fineFieldOrder, comm, stream);
}

int mugiq_hip_loop_compute(MugiqHipLoop *lp) {
  MUGIQ_REQUIRE(lp != nullptr, "computeCoarseLoop: NULL loop handle");
  int st = MUGIQ_HIP_SUCCESS;
  const size_t cb = lp->loopBytes();
  const bool basic = lp->calcType == MUGIQ_HIP_LOOP_CALC_TYPE_BASIC_KERNEL;
  lp->phases.clear();
  lp->eventsUsed = 0;
  lp->carryUltra = lp->ultraCarried = false;
  lp->ultraCarrier = -1;
  const auto tWall0 = std::chrono::steady_clock::now();
  if (lp->coarseMode && lp->levelVecs.size() > 1) {
    // coarsest level -> level 1 through the upper transfer operators (lib/loop_mugiq.cpp:306-311), all eigenvectors per launch
    const int ph = phase_begin(lp, MUGIQ_HIP_PHASE_PROLONGATION, -1, lp->stream);
    for (int l = (int)lp->levelVecs.size() - 1; l >= 1; l--)
      if ((st = mugiq_hip_prolongate_coarse_batched(lp->levelVecs[l - 1].data(), lp->levelVecs[l].data(), lp->nEv, &lp->upper[l - 1], lp->stream)))
        return st;
    phase_end(lp, ph, lp->stream);
  }
  if (lp->coarseMode && lp->fineStore) {
    // prolongateEvec for every eigenvector, once (the reference repeats it per displacement entry, lib/loop_mugiq.cpp:482)
    const int ph = phase_begin(lp, MUGIQ_HIP_PHASE_PROLONGATION, -1, lp->stream);
    if ((st = mugiq_hip_prolongate_batched(lp->eVecs.data(), lp->coarseVecs.data(), lp->nEv, &lp->transfer, lp->stream))) return st;
    phase_end(lp, ph, lp->stream);
  }
  // ---- plan (OPT): which entries are reflected from which, and in what order things run.  The slots are independent,
  // so the order of lib/loop_mugiq.cpp:455 is kept for BASIC only; OPT posts the eigenvector halos of all partitioned
  // entries first, runs the ultra-local loop and the entries of unpartitioned directions while they travel, then the
  // partitioned entries (interior tiles before the halo is waited for), and the reflected entries last.
  std::vector<int> order;
  order.push_back(-1);
  int earlyEntry = -2;        // OPT plan with halos to post: the entry that runs before they are packed (-2: none)
  bool postHalos = false;
  bool grouped = false;
  std::vector<char> aheadFlags;
  if (basic) {
    for (int id = 0; id < lp->nDispEntries; id++) {
      lp->derivedFrom[id] = -1;
      order.push_back(id);
    }
  } else {
    lp->halo.resize(lp->nDispEntries);
    std::vector<char> ahead;
    if ((st = plan_opt(lp, ahead))) return st;
    bool any = false;
    for (int id = 0; id < lp->nDispEntries; id++) {
      lp->halo[id].posted = false;
      any = any || ahead[id];
    }
    // momentum-space output only needs the reflected entries in momentum space (csrc/reflect_mom.cpp): when the momentum list
    // holds -p for every p they are left out of position space, of the reorder and of the Fourier kernels, and derived on the
    // gathered array; dataPos materialises them on first request (MUGIQ_HIP_REFLECT_MOM=0: always in position space)
    lp->momReflect = false;
    if (lp->doMomProj && !lp->momProjDone) {
      bool anyDerived = false;
      for (int id = 0; id < lp->nDispEntries; id++) anyDerived = anyDerived || lp->derivedFrom[id] >= 0;
      std::vector<int> neg;
      bool on = true;
      if (const char *e = getenv("MUGIQ_HIP_REFLECT_MOM")) on = atoi(e) != 0;
      lp->momReflect = on && anyDerived && fused_projection_applies(lp) && momenta_negation_table(lp->momMatrix.data(), lp->Nmom, neg);
    }
    grouped = lp->haveComm && lp->comm.group_begin && lp->comm.group_end;
    for (int pass = 0; pass < 3; pass++)
      for (int id = 0; id < lp->nDispEntries; id++) {
        const bool derived = lp->derivedFrom[id] >= 0, part = lp->commDim[lp->dispDir[id]] != 0;
        if ((pass == 0 && !derived && !part) || (pass == 1 && !derived && part) || (pass == 2 && derived)) order.push_back(id);
      }
    // the ultra-local loop may ride along with a displaced entry (MUGIQ_HIP_CARRY_ULTRALOCAL=0: never): it then moves to the end
    // of the order and is skipped if some entry has taken it along
    lp->carryUltra = lp->nDispEntries > 0 && !lp->coarseMode;
    if (const char *e = getenv("MUGIQ_HIP_CARRY_ULTRALOCAL")) lp->carryUltra = lp->carryUltra && atoi(e) != 0;
    lp->ultraCarried = false;
    if (lp->carryUltra) {
      order.erase(order.begin());
      order.push_back(-1);
    }
    earlyEntry = -2;  // (the halos are posted further down, once the order is known)
    if (any) {
      if ((st = ensure_comm_stream(lp))) return st;
      // the halo stream starts behind what the compute stream holds so far (e.g. the prolongation that writes the eigenvectors)
      MUGIQ_CHECK_HIP(hipEventRecord(lp->evPacked, lp->stream));
      MUGIQ_CHECK_HIP(hipStreamWaitEvent(lp->commStream, lp->evPacked, 0));
      // One entry that needs no halo goes FIRST, before the halos are packed: a pack kernel launched ahead of it fills the
      // device and the entry's kernels queue up behind it (measured: the compute stream made no progress during the 12 ms of
      // packing even with the packs on the halo stream); launched behind a tiled kernel that is already resident -- one
      // workgroup per CU, LDS-bound occupancy -- the packs and the transfer run in its shadow instead.
      for (int id : order)
        if (id >= 0 && lp->derivedFrom[id] < 0 && !lp->commDim[lp->dispDir[id]]) {
          earlyEntry = id;
          break;
        }
    }
    postHalos = any;
    aheadFlags = ahead;
  }
  lp->posReflectPending = false;
  std::vector<int> pendingBoundary;  // entries whose interior tiles are out and whose boundary tiles wait for their halo
  auto run_boundaries = [&]() -> int {
    for (int id : pendingBoundary) {
      void *slot0 = static_cast<char *>(lp->dataPos_d) + (size_t)lp->nElemPosLocPerLoop * lp->nLoopOffset[id] * cb;
      if ((st = entry_fused(lp, id, slot0, 2))) return st;
      free_scratch(lp);
    }
    pendingBoundary.clear();
    return MUGIQ_HIP_SUCCESS;
  };
  auto run_one = [&](int id, bool holdScratch = false) -> int {
    if (id == -1 && !basic && lp->carryUltra && lp->ultraCarried) return MUGIQ_HIP_SUCCESS;  // produced by a displaced entry's pass
    if (id >= 0 && !basic && lp->momReflect && lp->derivedFrom[id] >= 0) {   // derived in momentum space; position space on request
      lp->posReflectPending = true;
      return MUGIQ_HIP_SUCCESS;
    }
    long long bufOffset;
    size_t bufByteSize;
    if (id != -1) {  // :465-474
      bufOffset = lp->nElemPosLocPerLoop * lp->nLoopOffset[id];
      bufByteSize = cb * (size_t)lp->nElemPosLocPerLoop * lp->nLoopPerEntry[id];
    } else {
      bufOffset = 0;
      bufByteSize = cb * (size_t)lp->nElemPosLocPerLoop;
    }
    void *slot0 = static_cast<char *>(lp->dataPos_d) + (size_t)bufOffset * cb;
    // cudaMemset :476 -- needed where kernels accumulate into the slots: the ultra-local loop, the BASIC plan, and an OPT
    // entry that falls back to the step-by-step sequence (length beyond the neighbour).  Reflected entries and the fused
    // displaced contraction write every site of their slots (MUGIQ_HIP_REGION_OVERWRITE).
    const bool stepByStep = id >= 0 && lp->commDim[lp->dispDir[id]] && lp->dispStop[id] > lp->localL[lp->dispDir[id]];
    if (id == -1 || basic || (lp->derivedFrom[id] < 0 && stepByStep)) MUGIQ_CHECK_HIP(hipMemsetAsync(slot0, 0, bufByteSize, lp->stream));
    const bool reflected = id >= 0 && !basic && lp->derivedFrom[id] >= 0;
    const bool split = id >= 0 && !basic && !reflected && !stepByStep && lp->commDim[lp->dispDir[id]];  // entry_fused opens its own phases
    const int ph = split ? -1
                         : phase_begin(lp, id < 0 ? MUGIQ_HIP_PHASE_ULTRA_LOCAL
                                                  : reflected ? MUGIQ_HIP_PHASE_ENTRY_REFLECTED
                                                              : (basic || stepByStep) ? MUGIQ_HIP_PHASE_ENTRY_STEPWISE : MUGIQ_HIP_PHASE_ENTRY_FUSED,
                                       id, lp->stream);
    if (id == -1 && lp->coarseMode && !lp->fineStore) {
      // MG ultra-local loop without materialising the fine vectors
      st = mugiq_hip_prolongate_contract_batched(slot0, lp->loopPrecision, lp->coarseVecs.data(), lp->sigma.data(), lp->nEv,
                                                 &lp->transfer, lp->stream);
    } else if (id == -1) {
      if (basic) {
        for (int n = 0; n < lp->nEv && !st; n++)  // :501-502
          st = mugiq_hip_perform_loop_contraction_batched_mixed(slot0, lp->loopPrecision, &lp->eVecs[n], &lp->eVecs[n], &lp->sigma[n], 1,
                                                                lp->stream);
      } else {
        st = mugiq_hip_perform_loop_contraction_batched_mixed(slot0, lp->loopPrecision, lp->eVecs.data(), lp->eVecs.data(),
                                                              lp->sigma.data(), lp->nEv, lp->stream);
      }
    } else {
      if (basic) st = entry_basic(lp, id, slot0);
      else if (lp->derivedFrom[id] >= 0) st = entry_reflected(lp, id, lp->derivedFrom[id], slot0);
      else if (split && lp->halo[id].posted) {
        st = entry_fused(lp, id, slot0, 1);  // interior tiles now; the boundary tiles once every entry's interior is through
        pendingBoundary.push_back(id);
      } else st = entry_fused(lp, id, slot0);
      // No host synchronisation between entries: every user of this entry's scratch is ordered on lp->stream (the halo
      // stream's part was waited for by the boundary kernels), so the next entry may take the buffers over at once.
      // The exception is the entry that runs BEFORE the halos are posted: the pack and halo streams start from an event
      // recorded ahead of it, so a buffer it hands back could be given to prepare_halo as gsend / grecv and be written by the
      // pack kernels while this entry's kernels still read it.  Its scratch stays out of the pool until the compute ends.
      if (holdScratch) {
        for (void *q : lp->scratch) lp->held.push_back(q);
        lp->scratch.clear();
      } else {
        free_scratch(lp);
      }
    }
    phase_end(lp, ph, lp->stream);
    return st;
  };
  // the halos of the plan: link fields, packed face layers, one transfer group on the halo stream
  int phPack = -1, phHalo = -1, maxBlocks = 0;
  bool halosPrepared = false;
  auto prepare_halos = [&]() -> int {
    for (int id = 0; id < lp->nDispEntries; id++)
      if (aheadFlags[id] && (st = prepare_halo(lp, id))) return st;
    double haloBytes = 0;
    for (int id = 0; id < lp->nDispEntries; id++)
      if (lp->halo[id].posted) {
        haloBytes += (double)halo_bytes(lp, id);
        maxBlocks = std::max(maxBlocks, lp->halo[id].nBlocks);
      }
    // the pack stream starts where the halo stream starts (behind what the compute stream held when the compute began)
    MUGIQ_CHECK_HIP(hipStreamWaitEvent(lp->packStream, lp->evPacked, 0));
    phPack = phase_begin(lp, MUGIQ_HIP_PHASE_HALO_PREPARE, -1, lp->packStream, haloBytes);
    phHalo = phase_begin(lp, MUGIQ_HIP_PHASE_HALO_TRANSFER, -1, lp->commStream, haloBytes);
    halosPrepared = true;
    return MUGIQ_HIP_SUCCESS;
  };
  auto send_halos = [&](int bFirst, int whichFirst) -> int {
    for (int b = bFirst; b < maxBlocks; b++)
      if ((st = send_halo_block(lp, b, grouped, b == bFirst ? whichFirst : -1))) return st;
    phase_end(lp, phPack, lp->packStream);
    phase_end(lp, phHalo, lp->commStream);
    return MUGIQ_HIP_SUCCESS;
  };
  // Who packs?  Pack kernels beside a tile kernel that fills every CU's registers and LDS do not overlap with it, they take turns
  // (configs[2] per-GPU job: 12.6 ms of packing made the first entry 10 ms longer).  Where the entry that runs first is a mu = x
  // entry on the row tile of csrc/fused_mfma.hip and the partitioned axes are z / t, that entry writes the face layers itself -- every
  // raw eigenvector passes through its registers anyway -- and the pack kernels read nothing a second time.  The first block of a
  // halo that really travels still goes out ahead, packed by its own kernel: the link must not wait for the entry to end.
  // MUGIQ_HIP_PACK_IN_ENTRY = 0: pack kernels for everything.
  std::vector<EntryPackTarget> packTargets;
  lp->halosPackedInEntry = 0;
  if (postHalos && earlyEntry >= 0 && lp->dispDir[earlyEntry] == 0 && lp->loopPrecision == lp->precision) {
    std::vector<int> kv;
    for (int k = lp->dispStart[earlyEntry]; k <= lp->dispStop[earlyEntry]; k++) kv.push_back(k);
    const int room = entry_pack_capacity(lp->eVecs[0], kv.data(), (int)kv.size());
    if (room > 0 && !(st = prepare_halos())) {
      for (int id = 0; id < lp->nDispEntries && (int)packTargets.size() < room; id++) {
        MugiqHipLoop::HaloPost &h = lp->halo[id];
        if (!h.posted || lp->dispDir[id] < 2) continue;
        const int from = h.selfAlias ? 0 : h.blockN;  // (a halo that travels in ONE block is packed by its kernel, ahead of the entry)
        if (from >= lp->nEv) continue;
        h.entryPacksFrom = from;
        packTargets.push_back(EntryPackTarget{h.gsend, lp->dispDir[id], lp->dispSign[id] == MUGIQ_HIP_DISP_SIGN_PLUS ? 0 : 1, lp->dispStop[id], from});
      }
      if (packTargets.empty()) {  // nothing for the entry to do: the order of round 3 (entry first, pack kernels in its shadow)
        st = run_one(earlyEntry, true);
        if (!st) st = send_halos(0, -1);
      } else {
        st = send_halo_block(lp, 0, grouped, 0);  // first blocks that are packed by their kernels: on their way before the entry starts
        if (!st) {
          set_entry_pack_hint(packTargets.data(), (int)packTargets.size());
          st = run_one(earlyEntry, true);
          const bool taken = entry_pack_taken();
          set_entry_pack_hint(nullptr, 0);
          if (!st && taken) lp->halosPackedInEntry = (int)packTargets.size();
          if (!st && !taken) st = set_error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "computeCoarseLoop: entry %d was to write the face layers of the posted halos and did not (internal)", earlyEntry);
        }
        if (!st) {
          MUGIQ_CHECK_HIP(hipEventRecord(lp->evEntryPacked, lp->stream));
          MUGIQ_CHECK_HIP(hipStreamWaitEvent(lp->packStream, lp->evEntryPacked, 0));
          st = send_halo_block(lp, 0, grouped, 1);
        }
        if (!st) st = send_halos(1, -1);
      }
    }
  }
  if (!halosPrepared && !st) {
    if (earlyEntry >= 0) st = run_one(earlyEntry, postHalos);
    if (!st && postHalos) st = prepare_halos();
    if (!st && postHalos) st = send_halos(0, -1);
  }
  for (int id : order) {
    if (st) break;
    if (id == earlyEntry) continue;
    // reflected entries and the ultra-local loop come after the computed ones: the boundary tiles go before them (a reflected
    // entry reads the complete slots of its source)
    if (!pendingBoundary.empty() && (id < 0 || lp->derivedFrom[id] >= 0)) st = run_boundaries();
    if (!st) st = run_one(id);
  }
  if (!st && !pendingBoundary.empty()) st = run_boundaries();
  {
    hipError_t e = hipStreamSynchronize(lp->stream);
    if (!st && e != hipSuccess) st = set_error(MUGIQ_HIP_ERROR_HIP, "computeCoarseLoop: %s", hipGetErrorString(e));
  }
  // hand the buffers of the posted halos back (their transfers were waited for by the entries that used them; after an
  // error drain the halo stream first)
  if (!lp->held.empty()) {
    if (st && lp->packStream) (void)hipStreamSynchronize(lp->packStream);
    if (st && lp->commStream) (void)hipStreamSynchronize(lp->commStream);
    for (void *p : lp->held)
      for (auto &b : lp->pool)
        if (b.ptr == p) b.inUse = false;
    lp->held.clear();
  }
  if (st) return st;
  lp->dataPosCopied = false;
  if (lp->doMomProj && (st = momentum_projection(lp))) return st;  // :517-520
  MUGIQ_CHECK_HIP(hipStreamSynchronize(lp->stream));
  if (lp->profiling) {
    if (lp->packStream) MUGIQ_CHECK_HIP(hipStreamSynchronize(lp->packStream));  // HALO_PREPARE's end event is recorded there
    if (lp->commStream) MUGIQ_CHECK_HIP(hipStreamSynchronize(lp->commStream));
    phases_resolve(lp);
    phase_host(lp, MUGIQ_HIP_PHASE_TOTAL_WALL, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tWall0).count());
  }
  lp->computed = true;
  return MUGIQ_HIP_SUCCESS;
}

int mugiq_hip_loop_set_profiling(MugiqHipLoop *lp, int on) {
  MUGIQ_REQUIRE(lp != nullptr, "mugiq_hip_loop_set_profiling: NULL loop handle");
  lp->profiling = on != 0;
  return MUGIQ_HIP_SUCCESS;
}

int mugiq_hip_loop_get_phases(const MugiqHipLoop *lp, MugiqHipLoopPhase *out, int max_phases) {
  if (!lp) return -MUGIQ_HIP_ERROR_INVALID_ARGUMENT;
  const int n = (int)lp->phases.size();
  for (int i = 0; i < n && i < max_phases && out; i++) {
    out[i].kind = lp->phases[i].kind;
    out[i].entry = lp->phases[i].entry;
    out[i].ms = lp->phases[i].ms;
    out[i].bytes = lp->phases[i].bytes;
  }
  return n;
}

int mugiq_hip_loop_get_info(const MugiqHipLoop *lp, MugiqHipLoopInfo *info) {
  MUGIQ_REQUIRE(lp && info, "mugiq_hip_loop_get_info: NULL argument");
  info->nDispEntries = lp->nDispEntries;
  info->nLoop = lp->nLoop;
  info->nData = lp->nData;
  info->Nmom = lp->Nmom;
  info->precision = lp->precision;
  info->loopPrecision = lp->loopPrecision;
  info->field_order = lp->order;
  for (int d = 0; d < 4; d++) {
    info->localL[d] = lp->localL[d];
    info->totalL[d] = lp->totalL[d];
  }
  info->locT = lp->locT;
  info->totT = lp->totT;
  info->locV4 = lp->locV4;
  info->locV3 = lp->locV3;
  info->totV3 = lp->totV3;
  info->nElemPosLocPerLoop = lp->nElemPosLocPerLoop;
  info->nElemMomLocPerLoop = lp->nElemMomLocPerLoop;
  info->nElemMomTotPerLoop = lp->nElemMomTotPerLoop;
  info->nElemPosLoc = lp->nElemPosLoc;
  info->nElemMomLoc = lp->nElemMomLoc;
  info->nElemMomTot = lp->nElemMomTot;
  info->nElemPhMat = lp->nElemPhMat;
  return MUGIQ_HIP_SUCCESS;
}

int mugiq_hip_loop_get_entry(const MugiqHipLoop *lp, int id, int out6[6]) {
  MUGIQ_REQUIRE(lp && out6, "mugiq_hip_loop_get_entry: NULL argument");
  MUGIQ_REQUIRE(id >= 0 && id < lp->nDispEntries, "mugiq_hip_loop_get_entry: entry %d out of range [0,%d)", id, lp->nDispEntries);
  out6[0] = lp->dispDir[id];
  out6[1] = lp->dispSign[id];
  out6[2] = lp->dispStart[id];
  out6[3] = lp->dispStop[id];
  out6[4] = lp->nLoopPerEntry[id];
  out6[5] = lp->nLoopOffset[id];
  return MUGIQ_HIP_SUCCESS;
}

int mugiq_hip_loop_ultra_local_carrier(const MugiqHipLoop *lp) { return (lp && lp->computed) ? lp->ultraCarrier : -1; }

int mugiq_hip_loop_halos_packed_in_entry(const MugiqHipLoop *lp) { return (lp && lp->computed) ? lp->halosPackedInEntry : -1; }

int mugiq_hip_loop_entry_derived_from(const MugiqHipLoop *lp, int id) {
  if (!lp || id < 0 || id >= lp->nDispEntries) return -2;
  return lp->derivedFrom[id];
}

const void *mugiq_hip_loop_data_pos_d(const MugiqHipLoop *lp) {
  if (!lp) return nullptr;
  if (materialise_reflected(const_cast<MugiqHipLoop *>(lp))) return nullptr;  // (see mugiq_hip.h: slots left out by the last compute)
  return lp->dataPos_d;
}

const void *mugiq_hip_loop_data_pos_h(MugiqHipLoop *lp) {
  if (!lp) return nullptr;
  const size_t bytes = (size_t)lp->nElemPosLoc * lp->loopBytes();
  if (!lp->dataPos) {  // lib/loop_mugiq.cpp:116; page-locked so that the copy runs at the link rate (pageable memory: a fraction of it)
    if (hipHostMalloc(&lp->dataPos, bytes, hipHostMallocDefault) == hipSuccess) lp->dataPosPinned = true;
    else {
      (void)hipGetLastError();
      lp->dataPos = calloc((size_t)lp->nElemPosLoc, lp->loopBytes());
    }
  }
  if (!lp->dataPos) return nullptr;
  if (materialise_reflected(lp)) return nullptr;
  if (!lp->dataPosCopied) {
    if (hipMemcpy(lp->dataPos, lp->dataPos_d, bytes, hipMemcpyDeviceToHost) != hipSuccess) return nullptr;  // :512
    lp->dataPosCopied = true;
  }
  return lp->dataPos;
}

const void *mugiq_hip_loop_data_mom_bcast_h(const MugiqHipLoop *lp) { return (lp && lp->momProjDone) ? lp->dataMom_bcast : nullptr; }

int mugiq_hip_write_loops_hdf5_mom(const char *filename, const void *dataMom_bcast_h, int precision, int Nmom,
                                   const int *momMatrix, int nDispEntries, const char *const *disp_str, const int *disp_start,
                                   const int *disp_stop, int locT, int totT) {
  const char *who = "writeLoopsHDF5_Mom";
  MUGIQ_REQUIRE(filename && filename[0] && dataMom_bcast_h && momMatrix, "%s: NULL / empty argument", who);
  MUGIQ_REQUIRE(precision == 4 || precision == 8, "%s: Precision not supported!", who);
  MUGIQ_REQUIRE(Nmom >= 1 && locT >= 1 && totT >= locT && totT % locT == 0, "%s: invalid sizes Nmom=%d locT=%d totT=%d", who, Nmom, locT, totT);
  MUGIQ_REQUIRE(nDispEntries >= 0 && (nDispEntries == 0 || (disp_str && disp_start && disp_stop)), "%s: invalid displacement entries", who);
  std::vector<std::string> ds;
  std::vector<int> a, b;
  int nLoop = 1;
  for (int i = 0; i < nDispEntries; i++) {
    MUGIQ_REQUIRE(disp_str[i] && disp_start[i] >= 1 && disp_start[i] <= disp_stop[i], "%s: invalid displacement entry %d", who, i);
    ds.push_back(disp_str[i]);
    a.push_back(disp_start[i]);
    b.push_back(disp_stop[i]);
    nLoop += disp_stop[i] - disp_start[i] + 1;
  }
  return write_loops_hdf5_mom(filename, dataMom_bcast_h, precision, Nmom, momMatrix, nDispEntries, ds, a, b, nLoop, locT, totT);
}

// Loop_Mugiq::writeLoopsHDF5  lib/loop_mugiq.cpp:668-693
int mugiq_hip_loop_write_hdf5(MugiqHipLoop *lp) {
  MUGIQ_REQUIRE(lp != nullptr, "writeLoopsHDF5: NULL loop handle");
  MUGIQ_REQUIRE(lp->computed, "writeLoopsHDF5: computeCoarseLoop has not been called");
  if (lp->doMomProj) {
    if (!lp->writeMom) {
      fprintf(stderr, "writeLoopsHDF5: Performed momentum projection, but got writeDatMom = FALSE.\n"
                      "writeLoopsHDF5: Will proceed to write momentum-space loop data\n");
      lp->writeMom = true;
    }
    MUGIQ_REQUIRE(!lp->fnameMom.empty(), "Got --loop-write-mom-space yes but no filename was given. Set option --loop-mom-space-filename");
    if (!lp->haveComm || lp->comm.rank == 0) {  // dataMom_bcast is replicated; one writer produces the identical file
      int st = write_loops_hdf5_mom(lp->fnameMom.c_str(), lp->dataMom_bcast, lp->loopPrecision, lp->Nmom, lp->momMatrix.data(),
                                    lp->nDispEntries, lp->dispString, lp->dispStart, lp->dispStop, lp->nLoop, lp->locT, lp->totT);
      if (st) return st;
    }
  } else if (!lp->writePos) {
    fprintf(stderr, "writeLoopsHDF5: Did not perform momentum projection, but got writeDatPos = FALSE.\n"
                    "writeLoopsHDF5: Will proceed to write position-space loop data\n");
    lp->writePos = true;
  }
  if (lp->writePos)  // Loop_Mugiq::writeLoopsHDF5_Pos is an errorQuda in the reference too (:660-663)
    return set_error(MUGIQ_HIP_ERROR_UNSUPPORTED, "writeLoopsHDF5_Pos: Not supported yet!");
  return MUGIQ_HIP_SUCCESS;
}

int mugiq_hip_loop_destroy(MugiqHipLoop *lp) {  // freeDataMemory, lib/loop_mugiq.cpp:182-229
  if (!lp) return MUGIQ_HIP_SUCCESS;
  destroy_pool(lp);
  if (lp->dataMom_bcast != lp->dataMom_h) free(lp->dataMom_bcast);
  if (lp->dataMom_h) (void)hipHostFree(lp->dataMom_h);
  if (lp->dataMom != lp->dataMom_h) free(lp->dataMom);
  if (lp->dataPosPinned) (void)hipHostFree(lp->dataPos);
  else free(lp->dataPos);
  if (lp->fineStore) (void)hipFree(lp->fineStore);
  for (void *q : lp->levelStore)
    if (q) (void)hipFree(q);
  for (auto &h : lp->halo) {
    if (h.evPacked) (void)hipEventDestroy(h.evPacked);
    if (h.evHalo) (void)hipEventDestroy(h.evHalo);
    for (hipEvent_t e : h.evPackedBlk) (void)hipEventDestroy(e);
    for (hipEvent_t e : h.evBlock) (void)hipEventDestroy(e);
  }
  if (lp->packStream) (void)hipStreamDestroy(lp->packStream);
  for (hipEvent_t e : lp->events) (void)hipEventDestroy(e);
  if (lp->evPacked) (void)hipEventDestroy(lp->evPacked);
  if (lp->evHalo) (void)hipEventDestroy(lp->evHalo);
  if (lp->evEntryPacked) (void)hipEventDestroy(lp->evEntryPacked);
  if (lp->commStream) (void)hipStreamDestroy(lp->commStream);
  if (lp->dataPos_d) (void)hipFree(lp->dataPos_d);
  if (lp->dataPosMP_d) (void)hipFree(lp->dataPosMP_d);
  if (lp->dataMom_d) (void)hipFree(lp->dataMom_d);
  if (lp->phaseMatrix_d) (void)hipFree(lp->phaseMatrix_d);
  delete lp;
  return MUGIQ_HIP_SUCCESS;
}

}  // extern "C"
