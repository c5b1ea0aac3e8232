// STUB (tests/quda_stub/README.md): MugiqLoopParam with the member names include/mugiq_hip_operators.hpp mirrors, and the
// declaration of the five-argument computeLoop.  Where /root/reference is mounted the syntax test uses the reference's own
// include/mugiq.h instead of this one (it needs nothing of QUDA beyond three parameter-struct names).
#pragma once
#include <quda.h>
#include <enum_mugiq.h>
#include <string>
#include <vector>
typedef struct MugiqLoopParam_s {
  int Nmom;
  std::vector<std::vector<int>> momMatrix;
  LoopFTSign FTSign;
  LoopCalcType calcType;
  MuGiqBool writeMomSpaceHDF5, writePosSpaceHDF5, doMomProj, doNonLocal;
  std::vector<std::string> disp_entry, disp_str;
  std::string fname_mom_h5, fname_pos_h5;
  std::vector<int> disp_start, disp_stop;
  void *gauge[4];
  QudaGaugeParam *gauge_param;
} MugiqLoopParam;
template <typename Float>
void computeLoop(QudaMultigridParam mgParams, QudaEigParam eigParams, MugiqLoopParam loopParams, MuGiqBool computeCoarse, MuGiqBool useMG);
