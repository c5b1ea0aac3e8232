// mugiq_hip_operators.hpp -- C++ host-side mirror of MuGiq's operator API over the C ABI of libmugiq_hip.so.
//
// Same template names, argument order and meaning as the reference's wrappers
// (lib/contract_wrappers.cu; declared at include/loop_mugiq.h:280-311 and include/displace.h:109-111), with
// MugiqHipSpinorField / MugiqHipGaugeField standing in for quda::ColorSpinorField / cudaGaugeField, plus the
// enums (include/enum_mugiq.h), MugiqLoopParam (include/mugiq.h:28-47) and the public surface of
// Loop_Mugiq<Float,order> (include/loop_mugiq.h:123-134).  Where the reference aborts through errorQuda, these
// throw mugiq_hip::Error carrying the same message.  Header-only; link with -lmugiq_hip.
//
// Inside a MuGiq/QUDA build include mugiq_hip_quda_adapter.hpp instead: it defines MUGIQ_HIP_NO_REFERENCE_ENUMS (MuGiq's own
// enum_mugiq.h provides the enums) and MUGIQ_HIP_WITH_QUDA, and produces the descriptors from QUDA fields.
#ifndef MUGIQ_HIP_OPERATORS_HPP
#define MUGIQ_HIP_OPERATORS_HPP

#include <climits>
#include <complex>
#include <cstdio>
#include <stdexcept>
#include <string>
#include <vector>

#include "mugiq_hip.h"

#ifndef MUGIQ_HIP_NO_REFERENCE_ENUMS
#include "mugiq_hip_enums.hpp"  // MuGiq's enums (inside a MuGiq build its own enum_mugiq.h provides them)
#endif

namespace mugiq_hip {

constexpr int FLOAT2_FIELD_ORDER = MUGIQ_HIP_FLOAT2_FIELD_ORDER;  // QUDA_FLOAT2_FIELD_ORDER
constexpr int FLOAT4_FIELD_ORDER = MUGIQ_HIP_FLOAT4_FIELD_ORDER;  // QUDA_FLOAT4_FIELD_ORDER

struct Error : std::runtime_error {
  int status;
  Error(int st, const std::string &msg) : std::runtime_error(msg), status(st) {}
};
inline void check(int status) {
  if (status != 0) throw Error(status, mugiq_hip_last_error());
}

using ColorSpinorField = MugiqHipSpinorField;
using GaugeField = MugiqHipGaugeField;

template <typename Float> constexpr int precisionOf() {
  static_assert(sizeof(Float) == 4 || sizeof(Float) == 8, "Float must be float or double");
  return (int)sizeof(Float);
}
template <typename Float, int order> inline void checkField(const ColorSpinorField *f, const char *who) {
  if (f->precision != precisionOf<Float>() || f->field_order != order)
    throw Error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, std::string(who) + ": field precision/order does not match the template arguments");
}

// lib/contract_wrappers.cu:6-19, 26-43
template <typename Float> inline void copyGammaCoeffStructToSymbol() { check(mugiq_hip_copy_gamma_coeff_to_symbol(precisionOf<Float>())); }
template <typename Float> inline void copyGammaMapStructToSymbol() { check(mugiq_hip_copy_gamma_map_to_symbol(precisionOf<Float>())); }

// lib/contract_wrappers.cu:50-77 (commCoord replaces QUDA's comm_coord(); NULL = single process)
template <typename Float>
inline void createPhaseMatrixGPU(std::complex<Float> *phaseMatrix_d, const int *momMatrix_h, long long locV3, int Nmom, int FTSign,
                                 const int localL[], const int totalL[], const int commCoord[] = nullptr, void *stream = nullptr) {
  check(mugiq_hip_create_phase_matrix(phaseMatrix_d, momMatrix_h, locV3, Nmom, FTSign, localL, totalL, commCoord,
                                      precisionOf<Float>(), stream));
}

// lib/contract_wrappers.cu:88-115
template <typename Float, int fieldOrder>
inline void performLoopContraction(std::complex<Float> *loopData_d, ColorSpinorField *eVecL, ColorSpinorField *eVecR, Float sigma,
                                   void *stream = nullptr) {
  checkField<Float, fieldOrder>(eVecL, "performLoopContraction");
  checkField<Float, fieldOrder>(eVecR, "performLoopContraction");
  check(mugiq_hip_perform_loop_contraction(loopData_d, eVecL, eVecR, (double)sigma, stream));
}

// the eigenvector loop of lib/loop_mugiq.cpp:478-503 in one launch (new)
template <typename Float, int fieldOrder>
inline void performLoopContractionBatched(std::complex<Float> *loopData_d, const std::vector<ColorSpinorField> &eVecL,
                                          const std::vector<ColorSpinorField> &eVecR, const std::vector<double> &sigma,
                                          void *stream = nullptr) {
  if (eVecL.empty() || eVecL.size() != eVecR.size() || eVecL.size() != sigma.size())
    throw Error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "performLoopContractionBatched: size mismatch");
  checkField<Float, fieldOrder>(&eVecL[0], "performLoopContractionBatched");
  check(mugiq_hip_perform_loop_contraction_batched(loopData_d, eVecL.data(), eVecR.data(), sigma.data(), (int)eVecL.size(), stream));
}

// lib/contract_wrappers.cu:133-156
template <typename Float>
inline void convertIdxOrder_mapGamma(std::complex<Float> *dataPosMP_d, const std::complex<Float> *dataPos_d, int nData, int nLoop,
                                     int nParity, int volumeCB, const int localL[], void *stream = nullptr) {
  check(mugiq_hip_convert_idx_order_map_gamma(dataPosMP_d, dataPos_d, nData, nLoop, nParity, volumeCB, localL, precisionOf<Float>(),
                                              stream));
}

// lib/contract_wrappers.cu:171-198 (the halo exchange of :166-169 is the caller's: fill src->ghost first)
template <typename Float, int order>
inline void performCovariantDisplacementVector(ColorSpinorField *dst, ColorSpinorField *src, GaugeField *gauge, DisplaceDir dispDir,
                                               DisplaceSign dispSign, const int commDim[4] = nullptr, void *stream = nullptr) {
  checkField<Float, order>(dst, "performCovariantDisplacementVector");
  checkField<Float, order>(src, "performCovariantDisplacementVector");
  check(mugiq_hip_perform_covariant_displacement_vector(dst, src, gauge, (int)dispDir, (int)dispSign, commDim, stream));
}

// the cublas{Z,C}gemm of lib/loop_mugiq.cpp:363-378
template <typename Float>
inline void momentumProjectionGemm(std::complex<Float> *dataMom_d, const std::complex<Float> *dataPosMP_d,
                                   const std::complex<Float> *phaseMatrix_d, int locT, int nData, long long locV3, int Nmom,
                                   void *stream = nullptr) {
  check(mugiq_hip_momentum_projection(dataMom_d, dataPosMP_d, phaseMatrix_d, locT, nData, locV3, Nmom, precisionOf<Float>(), nullptr,
                                      0, stream));
}

// include/gamma.h:11-20
inline std::string GammaName(int m) {
  const char *s = mugiq_hip_gamma_name(m);
  if (!s) throw std::out_of_range("GammaName");
  return s;
}

// The members of QudaGaugeParam this path reads (lib/displace.cpp:70-99: local lattice, host and device precision).
// Inside a QUDA build (MUGIQ_HIP_WITH_QUDA, see mugiq_hip_quda_adapter.hpp) it IS QudaGaugeParam: same member names.
#ifdef MUGIQ_HIP_WITH_QUDA
using GaugeParam = QudaGaugeParam;
#else
struct GaugeParam {
  int X[4];       // local lattice dimensions
  int cpu_prec;   // precision of the host links: 4 | 8 (QUDA_SINGLE/DOUBLE_PRECISION have these values)
  int cuda_prec;  // precision of the device field: must equal sizeof(Float) of the loop (lib/displace.cpp:84-87)
};
#endif

// include/mugiq.h:28-47, member for member: `gauge` are the four host arrays of the LOCAL lattice in QDP order
// (gauge[dir][(parity*V/2 + x_cb)*18 + (row*3+col)*2 + re/im], tests/loop.cpp:88,106,902-918) and `gauge_param` their
// description; Loop_Mugiq builds the border-extended device field from them as Displace does (lib/displace.cpp:70-134).
// gauge_ext (not in the reference) hands over an extended device field that already exists instead.
struct MugiqLoopParam {
  int Nmom = 0;
  std::vector<std::vector<int>> momMatrix;  // [Nmom][3]
  LoopFTSign FTSign = LOOP_FT_SIGN_PLUS;
  LoopCalcType calcType = LOOP_CALC_TYPE_OPT_KERNEL;
  MuGiqBool writeMomSpaceHDF5 = MUGIQ_BOOL_FALSE;
  MuGiqBool writePosSpaceHDF5 = MUGIQ_BOOL_FALSE;
  MuGiqBool doMomProj = MUGIQ_BOOL_FALSE;
  MuGiqBool doNonLocal = MUGIQ_BOOL_FALSE;
  std::vector<std::string> disp_entry;
  std::vector<std::string> disp_str;
  std::string fname_mom_h5;
  std::string fname_pos_h5;
  std::vector<int> disp_start;
  std::vector<int> disp_stop;
  void *gauge[4] = {nullptr, nullptr, nullptr, nullptr};
  GaugeParam *gauge_param = nullptr;
  const GaugeField *gauge_ext = nullptr;
  int loopPrecision = 0;  // not in the reference: 8 over fp32 eigenvectors = mixed precision
};

// tests/loop.cpp:656-705
inline void setDisplaceEntryString(MugiqLoopParam &p, const std::string &entries) {
  const int maxE = 64;
  std::vector<char> strs(4 * maxE);
  std::vector<int> a(maxE), b(maxE);
  int n = mugiq_hip_parse_displace_entry_string(entries.c_str(), maxE, strs.data(), a.data(), b.data());
  if (n < 0) check(-n);
  p.disp_entry.clear();
  p.disp_str.clear();
  p.disp_start.assign(a.begin(), a.begin() + n);
  p.disp_stop.assign(b.begin(), b.begin() + n);
  size_t pos = 0;
  for (int i = 0; i < n; i++) {
    p.disp_str.emplace_back(&strs[4 * i]);
    size_t semi = entries.find(';', pos);
    p.disp_entry.push_back(entries.substr(pos, semi == std::string::npos ? std::string::npos : semi - pos));
    pos = semi == std::string::npos ? entries.size() : semi + 1;
  }
  p.doNonLocal = MUGIQ_BOOL_TRUE;
}

// lib/contract_wrappers.cu:166-169: depth-1 ghost zones of every partitioned dimension, both directions
inline void exchangeGhostVec(ColorSpinorField *x, const MugiqHipComm *comm = nullptr, void *stream = nullptr) {
  check(mugiq_hip_exchange_ghost_vec(x, comm, stream));
}

// include/displace.h:13-80, lib/displace.cpp: the displacement state machine Loop_Mugiq drives (its friend in the reference;
// public here so that the reference's loop nest, lib/loop_mugiq.cpp:455-509, can be written against it call for call).
// `comm` stands for QUDA's communicator (commDimPartitioned), NULL = one process.
template <typename Float, int fieldOrder> class Displace {
  const std::vector<std::string> DisplaceFlagArray{"+x", "-x", "+y", "-y", "+z", "-z", "+t", "-t"};  // include/displace.h:21
  const char *DisplaceDirArray[4] = {"x", "y", "z", "t"};
  const char *DisplaceSignArray[2] = {"-", "+"};
  std::string dispString;
  DisplaceFlag dispFlag = DispFlagNone;
  DisplaceDir dispDir = DispDirNone;
  DisplaceSign dispSign = DispSignNone;
  GaugeField ownGauge_{};            // gaugeField when built here from loopParams.gauge[4]
  const GaugeField *gaugeField = nullptr;
  ColorSpinorField auxDispVec{};
  const MugiqHipComm *comm_;
  void *stream_;
  int commDim_[4], exRng[4];

public:
  Displace(MugiqLoopParam *lp, const ColorSpinorField *csf, int coarsePrec = 0, const MugiqHipComm *comm = nullptr, void *stream = nullptr)
      : comm_(comm), stream_(stream) {
    checkField<Float, fieldOrder>(csf, "Displace");
    for (int d = 0; d < 4; d++) {
      commDim_[d] = (comm && (comm->grid[d] > 1 || comm->partitioned[d])) ? 1 : 0;  // comm_dim_partitioned(d), forced included
      exRng[d] = 2 * commDim_[d];  // lib/displace.cpp:16
    }
    if (lp->gauge_ext) gaugeField = lp->gauge_ext;
    else {
      if (!lp->gauge[0] || !lp->gauge_param) throw Error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "Displace: loopParams holds no gauge field");
      const GaugeParam &gp = *lp->gauge_param;
      if ((int)gp.cuda_prec != precisionOf<Float>())
        throw Error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "createCudaGaugeField: Incompatible precision settings between Displace template and gauge field parameters");
      int X[4];
      for (int d = 0; d < 4; d++) X[d] = gp.X[d];
      check(mugiq_hip_alloc_extended_gauge(&ownGauge_, X, exRng, precisionOf<Float>()));
      const void *links[4] = {lp->gauge[0], lp->gauge[1], lp->gauge[2], lp->gauge[3]};
      const int st = mugiq_hip_create_extended_gauge(&ownGauge_, links, (int)gp.cpu_prec, comm, stream);
      if (st) {
        mugiq_hip_free_extended_gauge(&ownGauge_);
        check(st);
      }
      gaugeField = &ownGauge_;
    }
    check(mugiq_hip_alloc_spinor_like(&auxDispVec, csf, coarsePrec, nullptr));  // QUDA_ZERO_FIELD_CREATE, lib/displace.cpp:26-30
  }
  ~Displace() {
    mugiq_hip_free_spinor(&auxDispVec);
    if (ownGauge_.data) mugiq_hip_free_extended_gauge(&ownGauge_);
  }
  Displace(const Displace &) = delete;
  Displace &operator=(const Displace &) = delete;

  // lib/displace.cpp:137-152
  DisplaceFlag WhichDisplaceFlag() const {
    for (int i = 0; i < (int)DisplaceFlagArray.size(); i++)
      if (dispString == DisplaceFlagArray[i]) return static_cast<DisplaceFlag>(i);
    throw Error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "WhichDisplaceFlag: Cannot parse given displacement string = " + dispString + ".");
  }
  DisplaceDir WhichDisplaceDir() const { return static_cast<DisplaceDir>((int)dispFlag / 2); }                          // :155-180
  DisplaceSign WhichDisplaceSign() const { return ((int)dispFlag % 2 == 0) ? DispSignPlus : DispSignMinus; }           // :182-203
  // lib/displace.cpp:206-223
  void setupDisplacement(const std::string &dStr) {
    dispString = dStr;
    dispFlag = WhichDisplaceFlag();
    dispDir = WhichDisplaceDir();
    dispSign = WhichDisplaceSign();
    int d = -1, sg = -1;
    check(mugiq_hip_parse_displacement(dStr.c_str(), &d, &sg));  // the library's table must agree
    if (d != (int)dispDir || sg != (int)dispSign) throw Error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "setupDisplacement: Got invalid dispDir and/or dispSign.");
  }
  DisplaceDir dir() const { return dispDir; }
  DisplaceSign sign() const { return dispSign; }
  const GaugeField *gauge() const { return gaugeField; }
  // lib/displace.cpp:40-52
  void resetAuxDispVec(const ColorSpinorField *fineEvec) { check(mugiq_hip_copy_spinor(&auxDispVec, fineEvec, stream_)); }
  void swapAuxDispVec(ColorSpinorField *displacedEvec) { check(mugiq_hip_copy_spinor(displacedEvec, &auxDispVec, stream_)); }
  // lib/displace.cpp:55-67 (+ the exchangeGhostVec of lib/contract_wrappers.cu:178): displacedEvec <- D_{dir,sign} displacedEvec
  void doVectorDisplacement(DisplaceType dispType, ColorSpinorField *displacedEvec, int /*idisp*/) {
    if (dispType != DISPLACE_TYPE_COVARIANT) throw Error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "Unsupported Displacement type " + std::to_string((int)dispType));
    if (dispDir == DispDirNone) throw Error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "doVectorDisplacement: Got invalid dispDir and/or dispSign.");
    check(mugiq_hip_zero_spinor(&auxDispVec, stream_));
    exchangeGhostVec(displacedEvec, comm_, stream_);
    ColorSpinorField aux = auxDispVec;
    performCovariantDisplacementVector<Float, fieldOrder>(&aux, displacedEvec, const_cast<GaugeField *>(gaugeField), dispDir, dispSign, commDim_, stream_);
    swapAuxDispVec(displacedEvec);
  }
};

// include/loop_mugiq.h:123-134: Loop_Mugiq(loopParams, eigsolve); computeCoarseLoop(); writeLoopsHDF5()
// `eVecs` / `eVals_sigma` are what the reference reads from Eigsolve_Mugiq as a friend (lib/loop_mugiq.cpp:442,479).
template <typename Float, int fieldOrder> class Loop_Mugiq {
  MugiqHipLoop *h_ = nullptr;
  GaugeField ownGauge_{};  // Displace::gaugeField when built here from loopParams.gauge[4]

  // everything the C parameter block points into, alive for the duration of the create call
  struct CParam {
    std::vector<int> mom;
    std::vector<const char *> ent, str;
    MugiqHipLoopParam p{};
  };
  void fill(CParam &c, MugiqLoopParam *lp, const MugiqHipComm *comm, void *stream) {
    for (auto &m : lp->momMatrix) {
      if (m.size() != 3) throw Error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "Loop_Mugiq: momMatrix rows must have 3 entries");
      c.mom.insert(c.mom.end(), m.begin(), m.end());
    }
    for (auto &s : lp->disp_entry) c.ent.push_back(s.c_str());
    for (auto &s : lp->disp_str) c.str.push_back(s.c_str());
    c.ent.resize(c.str.size(), "");
    if (lp->disp_str.size() != lp->disp_start.size() || lp->disp_str.size() != lp->disp_stop.size())
      throw Error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "Displacement string length not compatible with displacement limits length");
    MugiqHipLoopParam &p = c.p;
    p.Nmom = lp->Nmom ? lp->Nmom : (int)lp->momMatrix.size();
    p.momMatrix = c.mom.empty() ? nullptr : c.mom.data();
    p.FTSign = (int)lp->FTSign;
    p.calcType = (int)lp->calcType;
    p.writeMomSpaceHDF5 = lp->writeMomSpaceHDF5 == MUGIQ_BOOL_TRUE;
    p.writePosSpaceHDF5 = lp->writePosSpaceHDF5 == MUGIQ_BOOL_TRUE;
    p.doMomProj = lp->doMomProj == MUGIQ_BOOL_TRUE;
    p.doNonLocal = lp->doNonLocal == MUGIQ_BOOL_TRUE;
    p.nDispEntries = (int)c.str.size();
    p.disp_entry = c.ent.data();
    p.disp_str = c.str.data();
    p.disp_start = lp->disp_start.data();
    p.disp_stop = lp->disp_stop.data();
    p.fname_mom_h5 = lp->fname_mom_h5.c_str();
    p.fname_pos_h5 = lp->fname_pos_h5.c_str();
    p.loopPrecision = lp->loopPrecision;
    p.gauge = lp->gauge_ext;
    if (p.doNonLocal && p.nDispEntries > 0 && !p.gauge && lp->gauge[0] && lp->gauge_param) {
      // Displace::Displace + createExtendedCudaGaugeField (lib/displace.cpp:6-37,70-134)
      const GaugeParam &gp = *lp->gauge_param;
      if ((int)gp.cuda_prec != precisionOf<Float>())
        throw Error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "createCudaGaugeField: Incompatible precision settings between Displace template and gauge field parameters");
      int X[4], R[4];
      for (int d = 0; d < 4; d++) {
        X[d] = gp.X[d];
        R[d] = 2 * ((comm && (comm->grid[d] > 1 || comm->partitioned[d])) ? 1 : 0);  // exRng[i] = 2 * redundantComms-or-commDimPartitioned, lib/displace.cpp:16
      }
      check(mugiq_hip_alloc_extended_gauge(&ownGauge_, X, R, precisionOf<Float>()));
      const void *links[4] = {lp->gauge[0], lp->gauge[1], lp->gauge[2], lp->gauge[3]};
      const int st = mugiq_hip_create_extended_gauge(&ownGauge_, links, (int)gp.cpu_prec, comm, stream);
      if (st) {
        mugiq_hip_free_extended_gauge(&ownGauge_);
        check(st);
      }
      p.gauge = &ownGauge_;
    }
  }

public:
  Loop_Mugiq(MugiqLoopParam *lp, const std::vector<ColorSpinorField> &eVecs, const std::vector<double> &eVals_sigma,
             const MugiqHipComm *comm = nullptr, void *stream = nullptr) {
    if (eVecs.empty() || eVecs.size() != eVals_sigma.size()) throw Error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "Loop_Mugiq: eVecs / eVals_sigma size mismatch");
    checkField<Float, fieldOrder>(&eVecs[0], "Loop_Mugiq");
    CParam c;
    fill(c, lp, comm, stream);
    const int st = mugiq_hip_loop_create(&h_, &c.p, eVecs.data(), eVals_sigma.data(), (int)eVecs.size(), comm, stream);
    if (st) {
      mugiq_hip_free_extended_gauge(&ownGauge_);
      check(st);
    }
  }
  // eigsolve->useMGenv && eigsolve->computeCoarse (lib/loop_mugiq.cpp:42,277-319,482): coarse eigenvectors on the coarsest
  // level of the hierarchy + mg_env->transfer[0 .. nCoarseLevels)
  Loop_Mugiq(MugiqLoopParam *lp, const std::vector<MugiqHipCoarseField> &coarseEvecs, const std::vector<double> &eVals_sigma,
             const std::vector<MugiqHipTransfer> &transfers, const MugiqHipComm *comm = nullptr, void *stream = nullptr) {
    if (coarseEvecs.empty() || coarseEvecs.size() != eVals_sigma.size() || transfers.empty())
      throw Error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "Loop_Mugiq: coarse eVecs / eVals_sigma / transfer size mismatch");
    if (fieldOrder != FLOAT2_FIELD_ORDER) throw Error(MUGIQ_HIP_ERROR_INVALID_ARGUMENT, "prolongateEvec: Vector prolongation requires fieldOrder = FLOAT2");
    CParam c;
    fill(c, lp, comm, stream);
    const int st = mugiq_hip_loop_create_coarse_levels(&h_, &c.p, coarseEvecs.data(), eVals_sigma.data(), (int)coarseEvecs.size(),
                                                       transfers.data(), (int)transfers.size(), fieldOrder, comm, stream);
    if (st) {
      mugiq_hip_free_extended_gauge(&ownGauge_);
      check(st);
    }
  }
  Loop_Mugiq(const Loop_Mugiq &) = delete;
  Loop_Mugiq &operator=(const Loop_Mugiq &) = delete;
  ~Loop_Mugiq() {
    mugiq_hip_loop_destroy(h_);
    mugiq_hip_free_extended_gauge(&ownGauge_);
  }

  void computeCoarseLoop() { check(mugiq_hip_loop_compute(h_)); }  // lib/loop_mugiq.cpp:439-525
  void writeLoopsHDF5() { check(mugiq_hip_loop_write_hdf5(h_)); }  // lib/loop_mugiq.cpp:668-693

  MugiqHipLoopInfo info() const {
    MugiqHipLoopInfo i;
    check(mugiq_hip_loop_get_info(h_, &i));
    return i;
  }
  MugiqHipLoop *handle() { return h_; }
  const std::complex<Float> *dataPos_d() const { return static_cast<const std::complex<Float> *>(mugiq_hip_loop_data_pos_d(h_)); }
  const std::complex<Float> *dataPos() { return static_cast<const std::complex<Float> *>(mugiq_hip_loop_data_pos_h(h_)); }
  const std::complex<Float> *dataMom_bcast() const { return static_cast<const std::complex<Float> *>(mugiq_hip_loop_data_mom_bcast_h(h_)); }
};

// lib/interface_mugiq.cpp:158-172: computeLoop<Float,fieldOrder>(loopParams, eigsolve)
template <typename Float, int fieldOrder>
inline void computeLoop(MugiqLoopParam loopParams, const std::vector<ColorSpinorField> &eVecs, const std::vector<double> &eVals_sigma,
                        const MugiqHipComm *comm = nullptr, void *stream = nullptr) {
  Loop_Mugiq<Float, fieldOrder> loop(&loopParams, eVecs, eVals_sigma, comm, stream);
  loop.computeCoarseLoop();
  if (loopParams.writeMomSpaceHDF5 != MUGIQ_BOOL_FALSE || loopParams.writePosSpaceHDF5 != MUGIQ_BOOL_FALSE) loop.writeLoopsHDF5();
  else fprintf(stderr, "computeLoop: Will NOT write output data!\n");  // warningQuda, lib/interface_mugiq.cpp:167
}

}  // namespace mugiq_hip

#endif  // MUGIQ_HIP_OPERATORS_HPP
