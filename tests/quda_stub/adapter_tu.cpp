// Translation unit of tests/test_adapter_syntax.py: the QUDA/MuGiq-side binding with both of its switches on, plus the layout
// self-check templates instantiated, so that every line of include/mugiq_hip_quda_adapter.hpp is seen by a compiler.
#define MUGIQ_HIP_ADAPTER_DEFINE_OPERATORS 1
#define MUGIQ_HIP_ADAPTER_DEFINE_COMPUTE_LOOP 1
#include "mugiq_hip_quda_adapter.hpp"
template void mugiq_hip_adapter::layoutSelfCheck<double, QUDA_FLOAT2_FIELD_ORDER>(const quda::ColorSpinorField &, QudaGaugeParam &);
template void mugiq_hip_adapter::layoutSelfCheck<float, QUDA_FLOAT4_FIELD_ORDER>(const quda::ColorSpinorField &, QudaGaugeParam &);
