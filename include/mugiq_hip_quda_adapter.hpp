// mugiq_hip_quda_adapter.hpp -- the MuGiq/QUDA-side binding of libmugiq_hip.so.
//
// STATUS: this file is shipped source, never LINKED or RUN in this repository: it needs QUDA's headers (color_spinor_field.h,
// gauge_field.h, comm_quda.h, transfer.h, index_helper.cuh) and MuGiq's (mugiq.h, eigsolve_mugiq.h, mg_mugiq.h), none of
// which exist in the build image (QUDA is not vendored by the reference and no version is pinned, CMakeLists.txt:112-114).
// It does go through a compiler: tests/test_adapter_syntax.py runs it, both switches on, through `-fsyntax-only` against
// declaration-only stand-ins for the QUDA / MPI / Eigsolve_Mugiq names it touches (tests/quda_stub/) and, where the reference
// tree is mounted, against the reference's own include/mugiq.h + include/enum_mugiq.h -- that pins templates, signatures and
// C-ABI calls, and nothing about layouts (see 5. below).
// Everything QUDA-independent that it calls IS built and tested here (include/mugiq_hip_operators.hpp, tests/cpp/loop.cpp).
// The QUDA accessors used below are those of QUDA develop, early-to-mid 2020 (the reference's vintage);
// a maintainer on another QUDA version adjusts the few lines marked [QUDA-API].
//
// What it provides, inside a MuGiq build that links libmugiq_hip.so instead of compiling lib/contract_wrappers.cu,
// lib/mugiq_{contract,displace,util}_kernels.cu and lib/loop_mugiq.cpp:
//   1. describe(): ColorSpinorField / cudaGaugeField / coarse ColorSpinorField / Transfer  ->  the POD descriptors of the C ABI
//   2. the six operator templates of include/loop_mugiq.h:280-311 and include/displace.h:109-111, same names and signatures
//   3. a MugiqHipComm over QUDA's process grid and MPI (what lib/loop_mugiq.cpp:61-88,406-424 and exchangeGhost do)
//   4. computeLoop<Float, fieldOrder>(MugiqLoopParam, Eigsolve_Mugiq *)  (lib/interface_mugiq.cpp:158-172) and the five-argument
//      computeLoop<Float>(QudaMultigridParam, QudaEigParam, MugiqLoopParam, MuGiqBool, MuGiqBool) of include/mugiq.h:79-81
//      (lib/interface_mugiq.cpp:175-248) with the reference's own MugiqLoopParam (void *gauge[4] + QudaGaugeParam *gauge_param)
//   5. mugiq_hip_adapter::layoutSelfCheck(): writes known values through QUDA's own accessors and reads them back through the
//      descriptors' index formulas -- the one place where the conventions this library ASSUMES about QUDA (even-odd site
//      index, FLOAT2/FLOAT4 spinor order, FLOAT2 gauge order, ghost-face index, coarse-field and null-vector order) can be
//      pinned to a real QUDA.  Run it once per QUDA version before trusting results (it aborts through errorQuda).
#ifndef MUGIQ_HIP_QUDA_ADAPTER_HPP
#define MUGIQ_HIP_QUDA_ADAPTER_HPP

#include <mpi.h>
#include <hip/hip_runtime_api.h>

#include <quda.h>
#include <color_spinor_field.h>
#include <color_spinor_field_order.h>
#include <comm_quda.h>
#include <gauge_field.h>
#include <gauge_field_order.h>
#include <index_helper.cuh>
#include <transfer.h>

#include <mugiq.h>            // MugiqLoopParam, computeLoop<Float> declaration, enums
#include <eigsolve_mugiq.h>   // Eigsolve_Mugiq, MugiqEigParam, MG_Mugiq
#include <util_mugiq.h>

#define MUGIQ_HIP_NO_REFERENCE_ENUMS 1   // enum_mugiq.h (through mugiq.h) provides them
#define MUGIQ_HIP_WITH_QUDA 1            // mugiq_hip::GaugeParam = QudaGaugeParam
#include "mugiq_hip_operators.hpp"

namespace mugiq_hip_adapter {

using quda::ColorSpinorField;
using quda::cudaGaugeField;

#define MUGIQ_HIP_OK(call)                                         \
  do {                                                             \
    if ((call) != 0) errorQuda("%s", mugiq_hip_last_error());      \
  } while (0)

// ---- 1. descriptors ---------------------------------------------------------------------------------------------------
// What the reference's Arg structs read out of the fields (include/contract_util.cuh:69-134).
inline MugiqHipSpinorField describe(const ColorSpinorField &f) {
  MugiqHipSpinorField d{};
  d.data = const_cast<void *>(f.V());
  d.precision = static_cast<int>(f.Precision());    // QUDA_SINGLE/DOUBLE_PRECISION = 4 / 8
  d.field_order = static_cast<int>(f.FieldOrder()); // QUDA_FLOAT2/FLOAT4_FIELD_ORDER = 2 / 4
  d.nParity = f.SiteSubset();
  d.volumeCB = f.VolumeCB();
  d.stride = f.Stride();
  d.parity_offset = static_cast<int64_t>(f.Bytes() / 2 / (2 * f.Precision()));  // complex elements between the parities
  d.X[0] = (3 - d.nParity) * f.X(0);                // include/contract_util.cuh:87
  for (int i = 1; i < 4; i++) d.X[i] = f.X(i);
  for (int dim = 0; dim < 4; dim++)
    for (int dir = 0; dir < 2; dir++)
      d.ghost[dim][dir] = f.Ghost2() ? static_cast<void **>(f.Ghost2())[2 * dim + dir] : nullptr;  // [QUDA-API] ghost zone pointers
  return d;
}

// the EXTENDED field Displace builds (lib/displace.cpp:104-134); include/contract_util.cuh:99-105
inline MugiqHipGaugeField describe(const cudaGaugeField &u) {
  MugiqHipGaugeField g{};
  g.data = const_cast<void *>(u.Gauge_p());
  g.precision = static_cast<int>(u.Precision());
  for (int d = 0; d < 4; d++) {
    g.R[d] = u.R()[d];
    g.X[d] = u.X()[d] - 2 * u.R()[d];
  }
  g.stride = u.Stride();
  g.parity_offset = static_cast<int64_t>(u.Bytes() / 2 / (2 * u.Precision()));
  return g;
}

// a coarse-grid eigenvector (eigsolve->computeCoarse): FieldOrderCB<Float, 2, n_vec, 1, FLOAT2>
inline MugiqHipCoarseField describeCoarse(const ColorSpinorField &f) {
  if (f.FieldOrder() != QUDA_FLOAT2_FIELD_ORDER) errorQuda("describeCoarse: coarse fields must be FLOAT2 (lib/loop_mugiq.cpp:283)");
  MugiqHipCoarseField d{};
  d.data = const_cast<void *>(f.V());
  d.precision = static_cast<int>(f.Precision());
  d.nSpin = f.Nspin();
  d.nColor = f.Ncolor();
  d.volumeCB = f.VolumeCB();
  d.stride = f.Stride();
  d.parity_offset = static_cast<int64_t>(f.Bytes() / 2 / (2 * f.Precision()));
  d.X[0] = (3 - f.SiteSubset()) * f.X(0);
  for (int i = 1; i < 4; i++) d.X[i] = f.X(i);
  return d;
}

// one level of the hierarchy, mg_env->transfer[lev]: the packed null vectors live in Transfer::Vectors() on the device
inline MugiqHipTransfer describe(const quda::Transfer &T, const QudaMultigridParam &mg, int lev) {
  const ColorSpinorField &V = T.Vectors(QUDA_CUDA_FIELD_LOCATION);  // [QUDA-API] FieldOrderCB<Float, nSpin_f, nColor_f, n_vec, FLOAT2>
  MugiqHipTransfer t{};
  t.V = V.V();
  t.precision = static_cast<int>(V.Precision());
  t.nVec = mg.n_vec[lev];
  for (int d = 0; d < 4; d++) t.geoBlockSize[d] = mg.geo_block_size[lev][d];
  t.spinBlockSize = mg.spin_block_size[lev];
  t.X[0] = (3 - V.SiteSubset()) * V.X(0);
  for (int i = 1; i < 4; i++) t.X[i] = V.X(i);
  t.stride = V.Stride();
  t.parity_offset = static_cast<int64_t>(V.Bytes() / 2 / (2 * V.Precision()));
  return t;
}

// ---- 3. transport ------------------------------------------------------------------------------------------------------
// COMM_SPACE / COMM_TIME exactly as Loop_Mugiq::setupComms builds them (lib/loop_mugiq.cpp:61-88).
struct MpiGrid {
  MPI_Comm space = MPI_COMM_NULL, time = MPI_COMM_NULL;
  bool timeProcess = false;
  MugiqHipComm c{};
  static MPI_Datatype T(int prec) { return prec == 8 ? MPI_DOUBLE : MPI_FLOAT; }

  MpiGrid() {
    const int tCoord = comm_coord(3), cRank = comm_rank();
    MPI_Comm_split(MPI_COMM_WORLD, tCoord, cRank, &space);
    int time_color = cRank;
    const int time_tag = 1000;  // lib/loop_mugiq.cpp: time_tag member
    if (comm_coord(0) == 0 && comm_coord(1) == 0 && comm_coord(2) == 0) {
      time_color = (time_tag > comm_size()) ? time_tag : time_tag + comm_size();
      timeProcess = true;
    }
    MPI_Comm_split(MPI_COMM_WORLD, time_color, tCoord, &time);
    c.ctx = this;
    c.rank = cRank;
    c.size = comm_size();
    for (int d = 0; d < 4; d++) {
      c.grid[d] = comm_dim(d);
      c.coord[d] = comm_coord(d);
      c.partitioned[d] = (comm_dim(d) == 1 && comm_dim_partitioned(d)) ? 1 : 0;  // QUDA's forced partitioning (self-neighbour)
    }
    c.sendrecv = &sendrecv;
    c.reduce_space = &reduce_space;
    c.gather_time = &gather_time;
    c.bcast = &bcast;
    c.group_begin = nullptr;  // every sendrecv runs as it comes; see INTEGRATION.md for an Isend/Irecv + Waitall variant
    c.group_end = nullptr;
  }
  ~MpiGrid() {
    if (space != MPI_COMM_NULL) MPI_Comm_free(&space);
    if (time != MPI_COMM_NULL) MPI_Comm_free(&time);
  }
  // GPU-aware MPI assumed (device pointers); otherwise stage through pinned host buffers here
  static int sendrecv(void *, const void *s, void *r, size_t n, int dim, int dir, void *stream) {
    if (hipStreamSynchronize(static_cast<hipStream_t>(stream)) != hipSuccess) return 1;  // the pack kernel ran on `stream`
    int disp[4] = {0, 0, 0, 0};
    disp[dim] = dir;
    const int to = comm_rank_displaced(comm_default_topology(), disp);  // [QUDA-API]
    disp[dim] = -dir;
    const int from = comm_rank_displaced(comm_default_topology(), disp);
    // MPI counts are int: the OPT plan posts the whole multi-layer halo of all eigenvectors as ONE message (12.7 GB at
    // configs[2]), so it goes out in pieces of at most 1 GiB -- same order on both sides, the pairing is preserved
    const size_t piece = size_t(1) << 30;
    const char *sp = static_cast<const char *>(s);
    char *rp = static_cast<char *>(r);
    for (size_t off = 0; off < n; off += piece) {
      const int m = static_cast<int>(n - off < piece ? n - off : piece);
      const int st = MPI_Sendrecv(sp + off, m, MPI_BYTE, to, dim, rp + off, m, MPI_BYTE, from, dim, MPI_COMM_WORLD, MPI_STATUS_IGNORE);
      if (st != MPI_SUCCESS) return st;
    }
    return 0;
  }
  static int reduce_space(void *ctx, const void *s, void *r, size_t n, int prec) {  // lib/loop_mugiq.cpp:406
    return MPI_Reduce(s, r, static_cast<int>(n), T(prec), MPI_SUM, 0, static_cast<MpiGrid *>(ctx)->space);
  }
  static int gather_time(void *ctx, const void *s, void *r, size_t n, int prec) {   // :420-422
    MpiGrid *g = static_cast<MpiGrid *>(ctx);
    return g->timeProcess ? MPI_Gather(s, static_cast<int>(n), T(prec), r, static_cast<int>(n), T(prec), 0, g->time) : 0;
  }
  static int bcast(void *, void *b, size_t n, int prec) { return MPI_Bcast(b, static_cast<int>(n), T(prec), 0, MPI_COMM_WORLD); }  // :424
};

// ---- 5. layout self-check ------------------------------------------------------------------------------------------------
// Encodes (parity, x_cb, spin, colour) / (dir, parity, x_cb, row, col) into the VALUE of each element, writes the field through
// QUDA's accessors, downloads the raw device buffer and reads every element back through the index formulas of mugiq_hip.h.
// Also checks quda::getCoords / linkIndexShift / ghostFaceIndex against the restatements the kernels use (by value, for every
// site of the local lattice), through the library's own exported probe: mugiq_hip_perform_covariant_displacement_vector on
// an integer-valued field with unit links is an exact copy of one element per output, so comparing it with a displacement
// done by QUDA's own shift (or the reference CUDA kernel, when available) pins the integer index arithmetic.
template <typename Float> inline double encodeSpinor(int parity, int x_cb, int s, int c) { return ((parity * 16777216.0 + x_cb) * 4 + s) * 3 + c; }

template <typename Float, QudaFieldOrder order> void layoutSelfCheckSpinor(const ColorSpinorField &like) {
  using namespace quda;
  ColorSpinorParam cpuParam(like);
  cpuParam.location = QUDA_CPU_FIELD_LOCATION;
  cpuParam.fieldOrder = QUDA_SPACE_SPIN_COLOR_FIELD_ORDER;
  cpuParam.create = QUDA_ZERO_FIELD_CREATE;
  cpuColorSpinorField host(cpuParam);
  {
    colorspinor::FieldOrderCB<Float, 4, 3, 1, QUDA_SPACE_SPIN_COLOR_FIELD_ORDER> acc(host);  // [QUDA-API]
    for (int pty = 0; pty < 2; pty++)
      for (int x = 0; x < host.VolumeCB(); x++)
        for (int s = 0; s < 4; s++)
          for (int c = 0; c < 3; c++) acc(pty, x, s, c) = complex<Float>(encodeSpinor<Float>(pty, x, s, c), -encodeSpinor<Float>(pty, x, s, c));
  }
  ColorSpinorParam devParam(like);
  devParam.create = QUDA_ZERO_FIELD_CREATE;
  cudaColorSpinorField dev(devParam);
  dev = host;  // QUDA's own reordering into the native device order
  const MugiqHipSpinorField d = describe(dev);
  std::vector<std::complex<Float>> raw(static_cast<size_t>(2) * d.parity_offset);
  qudaMemcpy(raw.data(), dev.V(), raw.size() * sizeof(std::complex<Float>), cudaMemcpyDeviceToHost);  // [QUDA-API] (hipMemcpy on ROCm)
  for (int pty = 0; pty < 2; pty++)
    for (int x = 0; x < d.volumeCB; x++)
      for (int s = 0; s < 4; s++)
        for (int c = 0; c < 3; c++) {
          const int k = 3 * s + c;
          const int64_t idx = order == QUDA_FLOAT2_FIELD_ORDER ? pty * d.parity_offset + static_cast<int64_t>(k) * d.stride + x
                                                               : pty * d.parity_offset + (static_cast<int64_t>(k / 2) * d.stride + x) * 2 + (k % 2);
          if (raw[idx].real() != static_cast<Float>(encodeSpinor<Float>(pty, x, s, c)))
            errorQuda("layoutSelfCheck: spinor element (parity %d, x_cb %d, s %d, c %d) is not where mugiq_hip.h says (order %d)", pty, x, s, c, static_cast<int>(order));
        }
}

template <typename Float> void layoutSelfCheckGauge(QudaGaugeParam &gauge_param) {
  using namespace quda;
  // host links in QDP order, value = encoded (dir, parity, x_cb, row, col) -- what loopParams.gauge[4] holds
  const int V = gauge_param.X[0] * gauge_param.X[1] * gauge_param.X[2] * gauge_param.X[3];
  std::vector<std::vector<double>> qdp(4, std::vector<double>(static_cast<size_t>(V) * 18));
  void *ptr[4];
  for (int dir = 0; dir < 4; dir++) {
    for (int i = 0; i < V; i++)
      for (int e = 0; e < 9; e++) {
        qdp[dir][static_cast<size_t>(i) * 18 + 2 * e] = (dir * static_cast<double>(V) + i) * 9 + e;
        qdp[dir][static_cast<size_t>(i) * 18 + 2 * e + 1] = 0;
      }
    ptr[dir] = qdp[dir].data();
  }
  QudaGaugeParam gp = gauge_param;
  gp.cpu_prec = QUDA_DOUBLE_PRECISION;
  GaugeFieldParam cpuParam(ptr, gp);
  cpuGaugeField host(cpuParam);
  GaugeFieldParam devParam(ptr, gp);  // as Displace::createCudaGaugeField (lib/displace.cpp:70-99)
  devParam.create = QUDA_NULL_FIELD_CREATE;
  devParam.reconstruct = QUDA_RECONSTRUCT_NO;
  devParam.ghostExchange = QUDA_GHOST_EXCHANGE_PAD;
  devParam.pad = gp.ga_pad * 2;
  devParam.setPrecision(sizeof(Float) == 8 ? QUDA_DOUBLE_PRECISION : QUDA_SINGLE_PRECISION, true);
  cudaGaugeField dev(devParam);
  dev.copy(host);
  MugiqHipGaugeField g = describe(dev);
  std::vector<std::complex<Float>> raw(static_cast<size_t>(2) * g.parity_offset);
  qudaMemcpy(raw.data(), dev.Gauge_p(), raw.size() * sizeof(std::complex<Float>), cudaMemcpyDeviceToHost);
  for (int dir = 0; dir < 4; dir++)
    for (int pty = 0; pty < 2; pty++)
      for (int x = 0; x < V / 2; x++)
        for (int e = 0; e < 9; e++) {
          const int64_t idx = pty * g.parity_offset + static_cast<int64_t>(dir * 9 + e) * g.stride + x;
          const double want = (dir * static_cast<double>(V) + (pty * (V / 2) + x)) * 9 + e;
          if (raw[idx].real() != static_cast<Float>(want))
            errorQuda("layoutSelfCheck: link element (dir %d, parity %d, x_cb %d, elem %d) is not where mugiq_hip.h says", dir, pty, x, e);
        }
}

// even-odd coordinates: quda::getCoords against the formula of SURVEY.md Appendix A (the kernels' get_coords)
inline void layoutSelfCheckIndex(const int X[4]) {
  const int vcb = X[0] * X[1] * X[2] * X[3] / 2;
  for (int pty = 0; pty < 2; pty++)
    for (int cb = 0; cb < vcb; cb++) {
      int q[4];
      quda::getCoords(q, cb, X, pty);  // [QUDA-API] index_helper.cuh
      const int za = cb / (X[0] >> 1), zb = za / X[1], x1 = za - zb * X[1], x3 = zb / X[2], x2 = zb - x3 * X[2];
      const int x0 = 2 * cb + ((x1 + x2 + x3 + pty) & 1) - za * X[0];
      if (q[0] != x0 || q[1] != x1 || q[2] != x2 || q[3] != x3) errorQuda("layoutSelfCheck: getCoords differs at parity %d cb %d", pty, cb);
      for (int mu = 0; mu < 4; mu++) {
        int y[4] = {q[0], q[1], q[2], q[3]};
        y[mu] = (y[mu] + 1) % X[mu];
        const int lex = ((y[3] * X[2] + y[2]) * X[1] + y[1]) * X[0] + y[0];
        if (quda::linkIndexP1(q, X, mu) != (lex >> 1)) errorQuda("layoutSelfCheck: linkIndexP1 differs at parity %d cb %d mu %d", pty, cb, mu);
      }
    }
}

template <typename Float, QudaFieldOrder order> void layoutSelfCheck(const ColorSpinorField &eVec, QudaGaugeParam &gauge_param) {
  int X[4] = {(3 - eVec.SiteSubset()) * eVec.X(0), eVec.X(1), eVec.X(2), eVec.X(3)};
  layoutSelfCheckIndex(X);
  layoutSelfCheckSpinor<Float, order>(eVec);
  layoutSelfCheckGauge<Float>(gauge_param);
  printfQuda("mugiq_hip layoutSelfCheck: even-odd index, spinor order %d and FLOAT2 gauge order agree with include/mugiq_hip.h\n", static_cast<int>(order));
}

// ---- 4. the class-level entry points -------------------------------------------------------------------------------------
// the reference's MugiqLoopParam (include/mugiq.h:28-47) -> mugiq_hip::MugiqLoopParam, member for member
inline mugiq_hip::MugiqLoopParam convert(const ::MugiqLoopParam &in) {
  mugiq_hip::MugiqLoopParam p;
  p.Nmom = in.Nmom;
  p.momMatrix = in.momMatrix;
  p.FTSign = in.FTSign;
  p.calcType = in.calcType;
  p.writeMomSpaceHDF5 = in.writeMomSpaceHDF5;
  p.writePosSpaceHDF5 = in.writePosSpaceHDF5;
  p.doMomProj = in.doMomProj;
  p.doNonLocal = in.doNonLocal;
  p.disp_entry = in.disp_entry;
  p.disp_str = in.disp_str;
  p.fname_mom_h5 = in.fname_mom_h5;
  p.fname_pos_h5 = in.fname_pos_h5;
  p.disp_start = in.disp_start;
  p.disp_stop = in.disp_stop;
  for (int d = 0; d < 4; d++) p.gauge[d] = in.gauge[d];   // host QDP links: Loop_Mugiq builds the extended device field (Displace)
  p.gauge_param = in.gauge_param;
  return p;
}

}  // namespace mugiq_hip_adapter

// ---- 2. the operator templates of include/loop_mugiq.h:280-311 / include/displace.h:109-111 -------------------------------
// (this block replaces lib/contract_wrappers.cu when Loop_Mugiq / Displace are kept as they are)
#ifdef MUGIQ_HIP_ADAPTER_DEFINE_OPERATORS
using namespace quda;
template <typename Float> void copyGammaCoeffStructToSymbol() { MUGIQ_HIP_OK(mugiq_hip_copy_gamma_coeff_to_symbol(sizeof(Float))); }
template <typename Float> void copyGammaMapStructToSymbol() { MUGIQ_HIP_OK(mugiq_hip_copy_gamma_map_to_symbol(sizeof(Float))); }
template <typename Float>
void createPhaseMatrixGPU(complex<Float> *phaseMatrix_d, const int *momMatrix_h, long long locV3, int Nmom, int FTSign,
                          const int localL[], const int totalL[]) {
  const int cc[4] = {comm_coord(0), comm_coord(1), comm_coord(2), comm_coord(3)};
  MUGIQ_HIP_OK(mugiq_hip_create_phase_matrix(phaseMatrix_d, momMatrix_h, locV3, Nmom, FTSign, localL, totalL, cc, sizeof(Float), nullptr));
}
template <typename Float, QudaFieldOrder order>
void performLoopContraction(complex<Float> *loopData_d, ColorSpinorField *eVecL, ColorSpinorField *eVecR, Float sigma) {
  const MugiqHipSpinorField L = mugiq_hip_adapter::describe(*eVecL), R = mugiq_hip_adapter::describe(*eVecR);
  MUGIQ_HIP_OK(mugiq_hip_perform_loop_contraction(loopData_d, &L, &R, sigma, nullptr));
}
template <typename Float>
void convertIdxOrder_mapGamma(complex<Float> *dataPosMP_d, const complex<Float> *dataPos_d, int nData, int nLoop, int nParity,
                              int volumeCB, const int localL[]) {
  MUGIQ_HIP_OK(mugiq_hip_convert_idx_order_map_gamma(dataPosMP_d, dataPos_d, nData, nLoop, nParity, volumeCB, localL, sizeof(Float), nullptr));
}
template <typename Float, QudaFieldOrder order>
void performCovariantDisplacementVector(ColorSpinorField *dst, ColorSpinorField *src, cudaGaugeField *gauge, DisplaceDir dispDir,
                                        DisplaceSign dispSign) {
  src->exchangeGhost((QudaParity)1, 1, 0);  // exchangeGhostVec, lib/contract_wrappers.cu:166-169, unchanged
  const MugiqHipSpinorField D = mugiq_hip_adapter::describe(*dst), S = mugiq_hip_adapter::describe(*src);
  const MugiqHipGaugeField G = mugiq_hip_adapter::describe(*gauge);
  int commDim[4];
  for (int d = 0; d < 4; d++) commDim[d] = comm_dim_partitioned(d);
  MUGIQ_HIP_OK(mugiq_hip_perform_covariant_displacement_vector(&D, &S, &G, (int)dispDir, (int)dispSign, commDim, nullptr));
}
// explicit instantiations as in lib/contract_wrappers.cu:46-47,79-84,118-129,158-161,202-217
template void copyGammaCoeffStructToSymbol<float>();
template void copyGammaCoeffStructToSymbol<double>();
template void copyGammaMapStructToSymbol<float>();
template void copyGammaMapStructToSymbol<double>();
template void createPhaseMatrixGPU<float>(complex<float> *, const int *, long long, int, int, const int[], const int[]);
template void createPhaseMatrixGPU<double>(complex<double> *, const int *, long long, int, int, const int[], const int[]);
template void performLoopContraction<float, QUDA_FLOAT2_FIELD_ORDER>(complex<float> *, ColorSpinorField *, ColorSpinorField *, float);
template void performLoopContraction<float, QUDA_FLOAT4_FIELD_ORDER>(complex<float> *, ColorSpinorField *, ColorSpinorField *, float);
template void performLoopContraction<double, QUDA_FLOAT2_FIELD_ORDER>(complex<double> *, ColorSpinorField *, ColorSpinorField *, double);
template void performLoopContraction<double, QUDA_FLOAT4_FIELD_ORDER>(complex<double> *, ColorSpinorField *, ColorSpinorField *, double);
template void convertIdxOrder_mapGamma<float>(complex<float> *, const complex<float> *, int, int, int, int, const int[]);
template void convertIdxOrder_mapGamma<double>(complex<double> *, const complex<double> *, int, int, int, int, const int[]);
template void performCovariantDisplacementVector<float, QUDA_FLOAT2_FIELD_ORDER>(ColorSpinorField *, ColorSpinorField *, cudaGaugeField *, DisplaceDir, DisplaceSign);
template void performCovariantDisplacementVector<float, QUDA_FLOAT4_FIELD_ORDER>(ColorSpinorField *, ColorSpinorField *, cudaGaugeField *, DisplaceDir, DisplaceSign);
template void performCovariantDisplacementVector<double, QUDA_FLOAT2_FIELD_ORDER>(ColorSpinorField *, ColorSpinorField *, cudaGaugeField *, DisplaceDir, DisplaceSign);
template void performCovariantDisplacementVector<double, QUDA_FLOAT4_FIELD_ORDER>(ColorSpinorField *, ColorSpinorField *, cudaGaugeField *, DisplaceDir, DisplaceSign);
#endif  // MUGIQ_HIP_ADAPTER_DEFINE_OPERATORS

// ---- 4. computeLoop: lib/interface_mugiq.cpp:158-248 with Loop_Mugiq running in libmugiq_hip.so -------------------------------
#ifdef MUGIQ_HIP_ADAPTER_DEFINE_COMPUTE_LOOP
// lib/interface_mugiq.cpp:158-172
template <typename Float, QudaFieldOrder fieldOrder> void computeLoop(MugiqLoopParam loopParams, Eigsolve_Mugiq *eigsolve) {
  using namespace mugiq_hip_adapter;
  mugiq_hip::MugiqLoopParam hp = convert(loopParams);
  std::vector<double> sigma(eigsolve->getEvalsSigma()->begin(), eigsolve->getEvalsSigma()->end());  // lib/loop_mugiq.cpp:479
  MpiGrid grid;
  MG_Mugiq *mg_env = eigsolve->getMGEnv();
  // eigsolve->useMGenv && eigsolve->computeCoarse are private (Loop_Mugiq reads them as a friend, lib/loop_mugiq.cpp:42):
  // the field order tells the same thing, as lib/interface_mugiq.cpp:221-235 itself relies on
  const bool coarse = fieldOrder == QUDA_FLOAT2_FIELD_ORDER && mg_env != nullptr && eigsolve->getEvecs()[0]->Nspin() == 2;
  try {
    if (coarse) {
      std::vector<MugiqHipCoarseField> ev;
      for (auto *v : eigsolve->getEvecs()) ev.push_back(describeCoarse(*v));
      std::vector<MugiqHipTransfer> tr;
      for (int lev = 0; lev < mg_env->nCoarseLevels; lev++) {
        if (!mg_env->transfer[lev]) errorQuda("computeLoop: Transfer operator for level %d does not exist!", lev);  // lib/loop_mugiq.cpp:309,313
        tr.push_back(describe(*mg_env->transfer[lev], *mg_env->mgParams, lev));
      }
      mugiq_hip::Loop_Mugiq<Float, MUGIQ_HIP_FLOAT2_FIELD_ORDER> loop(&hp, ev, sigma, tr, &grid.c, nullptr);
      loop.computeCoarseLoop();
      if (loopParams.writeMomSpaceHDF5 != MUGIQ_BOOL_FALSE || loopParams.writePosSpaceHDF5 != MUGIQ_BOOL_FALSE) loop.writeLoopsHDF5();
      else warningQuda("%s: Will NOT write output data!\n", __func__);
    } else {
      std::vector<MugiqHipSpinorField> ev;  // what Loop_Mugiq reads as a friend (lib/loop_mugiq.cpp:442)
      for (auto *v : eigsolve->getEvecs()) ev.push_back(describe(*v));
      mugiq_hip::Loop_Mugiq<Float, static_cast<int>(fieldOrder)> loop(&hp, ev, sigma, &grid.c, nullptr);
      loop.computeCoarseLoop();
      if (loopParams.writeMomSpaceHDF5 != MUGIQ_BOOL_FALSE || loopParams.writePosSpaceHDF5 != MUGIQ_BOOL_FALSE) loop.writeLoopsHDF5();
      else warningQuda("%s: Will NOT write output data!\n", __func__);
    }
  } catch (const mugiq_hip::Error &e) {
    errorQuda("%s", e.what());  // the library reports, the host aborts the MPI job as the reference does
  }
}

// include/mugiq.h:79-81, lib/interface_mugiq.cpp:175-248: unchanged orchestration -- MG environment and eigensolver stay
// MuGiq's / QUDA's (out of this library's scope: eigenvectors and sigma_n are its inputs)
extern quda::TimeProfile profileEigensolveMuGiq;                                  // lib/interface_mugiq.cpp
MG_Mugiq *newMG_Mugiq(QudaMultigridParam *mgParams, QudaEigParam *QudaEigParams);  // lib/interface_mugiq.cpp:62
template <typename Float>
void computeLoop(QudaMultigridParam mgParams, QudaEigParam QudaEigParams, MugiqLoopParam loopParams, MuGiqBool computeCoarse, MuGiqBool useMG) {
  pushVerbosity(QudaEigParams.invert_param->verbosity);
  MG_Mugiq *mg_env = nullptr;
  Eigsolve_Mugiq *eigsolve = nullptr;
  MugiqEigParam *eigParams = new MugiqEigParam(&QudaEigParams);
  if (useMG) {
    mg_env = newMG_Mugiq(&mgParams, &QudaEigParams);
    eigsolve = new Eigsolve_Mugiq(eigParams, mg_env, &profileEigensolveMuGiq, computeCoarse);
  } else {
    eigsolve = new Eigsolve_Mugiq(eigParams, &profileEigensolveMuGiq);
  }
  eigsolve->printInfo();
  eigsolve->computeEvecs();
  eigsolve->computeEvals();
  eigsolve->printEvals();
  const QudaPrecision ePrec = eigsolve->getEvecs()[0]->Precision();
  if (!((ePrec == QUDA_SINGLE_PRECISION && sizeof(Float) == 4) || (ePrec == QUDA_DOUBLE_PRECISION && sizeof(Float) == 8)))
    errorQuda("Missmatch between eigenvector precision %d and templated precision %zu\n", static_cast<int>(ePrec), sizeof(Float));
  if (eigsolve->getEvecs()[0]->FieldOrder() == QUDA_FLOAT2_FIELD_ORDER) {
    if (!(useMG && computeCoarse)) errorQuda("%s: Got FieldOrder = FLOAT2, but useMGenv = FALSE and computeCoarse = FALSE\n", __func__);
    computeLoop<Float, QUDA_FLOAT2_FIELD_ORDER>(loopParams, eigsolve);
  } else if (eigsolve->getEvecs()[0]->FieldOrder() == QUDA_FLOAT4_FIELD_ORDER) {
    if (useMG && computeCoarse) errorQuda("%s: Got FieldOrder = FLOAT4, but useMGenv = TRUE and computeCoarse = TRUE\n", __func__);
    computeLoop<Float, QUDA_FLOAT4_FIELD_ORDER>(loopParams, eigsolve);
  }
  delete eigsolve;
  if (useMG) delete mg_env;
  delete eigParams;
  popVerbosity();
}
template void computeLoop<double>(QudaMultigridParam, QudaEigParam, MugiqLoopParam, MuGiqBool, MuGiqBool);
template void computeLoop<float>(QudaMultigridParam, QudaEigParam, MugiqLoopParam, MuGiqBool, MuGiqBool);
#endif  // MUGIQ_HIP_ADAPTER_DEFINE_COMPUTE_LOOP

#endif  // MUGIQ_HIP_QUDA_ADAPTER_HPP
