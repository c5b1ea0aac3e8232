// STUB (tests/quda_stub/README.md): accessor names of quda::ColorSpinorField the adapter calls.  Declarations only.
#pragma once
#include <complex>
#include <quda.h>
namespace quda {
template <typename T> using complex = std::complex<T>;
class TimeProfile;
class ColorSpinorField;
struct ColorSpinorParam {
  QudaFieldLocation location;
  QudaFieldOrder fieldOrder;
  QudaFieldCreate create;
  explicit ColorSpinorParam(const ColorSpinorField &);
};
class ColorSpinorField {
public:
  const void *V() const;
  void *V();
  QudaPrecision Precision() const;
  QudaFieldOrder FieldOrder() const;
  int SiteSubset() const;
  int VolumeCB() const;
  int Stride() const;
  size_t Bytes() const;
  int X(int) const;
  int Nspin() const;
  int Ncolor() const;
  void *Ghost2() const;
  void exchangeGhost(QudaParity, int nFace, int dagger) const;
};
class cpuColorSpinorField : public ColorSpinorField {
public:
  explicit cpuColorSpinorField(const ColorSpinorParam &);
};
class cudaColorSpinorField : public ColorSpinorField {
public:
  explicit cudaColorSpinorField(const ColorSpinorParam &);
  cudaColorSpinorField &operator=(const cpuColorSpinorField &);
};
}  // namespace quda
