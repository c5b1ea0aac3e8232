// STUB (tests/quda_stub/README.md).  Declarations only.
#pragma once
#include <color_spinor_field.h>
namespace quda {
class Transfer {
public:
  const ColorSpinorField &Vectors(QudaFieldLocation) const;
};
}  // namespace quda
