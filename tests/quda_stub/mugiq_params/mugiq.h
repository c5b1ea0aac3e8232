// STUB (tests/quda_stub/README.md): stands where MuGiq's include/mugiq.h stands in a real build -- a MugiqLoopParam with the
// member names include/mugiq_hip_operators.hpp mirrors (declaration order is irrelevant to the adapter, which reads them by name)
// and the declaration of the five-argument computeLoop.  Where /root/reference is mounted the syntax test uses the reference's own
// header instead of this one (it needs nothing of QUDA beyond three parameter-struct names).
#pragma once
#include <string>
#include <vector>

#include <quda.h>
#include <enum_mugiq.h>

struct MugiqLoopParam_s {
  // what to compute
  MuGiqBool doMomProj;
  MuGiqBool doNonLocal;
  LoopCalcType calcType;
  // Fourier transform
  LoopFTSign FTSign;
  int Nmom;
  std::vector<std::vector<int>> momMatrix;
  // displacement entries, parsed
  std::vector<std::string> disp_entry;
  std::vector<std::string> disp_str;
  std::vector<int> disp_start;
  std::vector<int> disp_stop;
  // output
  MuGiqBool writeMomSpaceHDF5;
  MuGiqBool writePosSpaceHDF5;
  std::string fname_mom_h5;
  std::string fname_pos_h5;
  // host links
  QudaGaugeParam *gauge_param;
  void *gauge[4];
};
typedef struct MugiqLoopParam_s MugiqLoopParam;

template <typename Float>
void computeLoop(QudaMultigridParam mgParams, QudaEigParam eigParams, MugiqLoopParam loopParams, MuGiqBool computeCoarse, MuGiqBool useMG);
