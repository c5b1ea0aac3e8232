// STUB (tests/quda_stub/README.md): accessor names of quda::cudaGaugeField the adapter calls.  Declarations only.
#pragma once
#include <quda.h>
namespace quda {
struct GaugeFieldParam {
  QudaFieldCreate create;
  QudaReconstructType reconstruct;
  QudaGhostExchange ghostExchange;
  int pad;
  GaugeFieldParam(void *const h_gauge[], const QudaGaugeParam &);
  void setPrecision(QudaPrecision, bool force_native = false);
};
class GaugeField {
public:
  QudaPrecision Precision() const;
  const int *R() const;
  const int *X() const;
  int Stride() const;
  size_t Bytes() const;
};
class cpuGaugeField : public GaugeField {
public:
  explicit cpuGaugeField(const GaugeFieldParam &);
};
class cudaGaugeField : public GaugeField {
public:
  explicit cudaGaugeField(const GaugeFieldParam &);
  const void *Gauge_p() const;
  void copy(const GaugeField &);
};
}  // namespace quda
