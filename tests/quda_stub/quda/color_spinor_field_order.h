// STUB (tests/quda_stub/README.md).  Declarations only.
#pragma once
#include <color_spinor_field.h>
namespace quda {
namespace colorspinor {
template <typename Float, int nSpin, int nColor, int nVec, QudaFieldOrder order> struct FieldOrderCB {
  explicit FieldOrderCB(const ColorSpinorField &);
  complex<Float> &operator()(int parity, int x_cb, int s, int c);
};
}  // namespace colorspinor
}  // namespace quda
