// STUB (tests/quda_stub/README.md): the accessors of MuGiq's Eigsolve_Mugiq / MG_Mugiq / MugiqEigParam that
// include/mugiq_hip_quda_adapter.hpp calls (reference: include/eigsolve_mugiq.h:104-184, include/mg_mugiq.h:12-30).
// Declarations only.
#pragma once
#include <vector>
#include <quda.h>
#include <color_spinor_field.h>
#include <transfer.h>
#include <enum_mugiq.h>
using namespace quda;
struct MugiqEigParam {
  explicit MugiqEigParam(QudaEigParam *);
};
struct MG_Mugiq {
  QudaMultigridParam *mgParams;
  Transfer *transfer[QUDA_MAX_MG_LEVEL - 1];
  int nCoarseLevels;
};
class Eigsolve_Mugiq {
public:
  Eigsolve_Mugiq(MugiqEigParam *, MG_Mugiq *, TimeProfile *, MuGiqBool computeCoarse);
  Eigsolve_Mugiq(MugiqEigParam *, TimeProfile *);
  void printInfo();
  void computeEvecs();
  void computeEvals();
  void printEvals();
  std::vector<ColorSpinorField *> &getEvecs();
  std::vector<double> *getEvalsSigma();
  MG_Mugiq *getMGEnv();
};
