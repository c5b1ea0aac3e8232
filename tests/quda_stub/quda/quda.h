// STUB (tests/quda_stub/README.md): names of QUDA's C interface that include/mugiq_hip_quda_adapter.hpp touches.  Declarations
// only -- no layout, no behaviour; it exists so that the adapter can go through `hipcc -fsyntax-only`.
#pragma once
#include <cstddef>
#include <cstdio>
#define QUDA_MAX_MG_LEVEL 5
typedef enum { QUDA_SINGLE_PRECISION = 4, QUDA_DOUBLE_PRECISION = 8 } QudaPrecision;
typedef enum { QUDA_FLOAT2_FIELD_ORDER = 2, QUDA_FLOAT4_FIELD_ORDER = 4, QUDA_SPACE_SPIN_COLOR_FIELD_ORDER = 9 } QudaFieldOrder;
typedef enum { QUDA_EVEN_PARITY = 0, QUDA_ODD_PARITY = 1, QUDA_INVALID_PARITY = -1 } QudaParity;
typedef enum { QUDA_CPU_FIELD_LOCATION = 1, QUDA_CUDA_FIELD_LOCATION = 2 } QudaFieldLocation;
typedef enum { QUDA_NULL_FIELD_CREATE, QUDA_ZERO_FIELD_CREATE } QudaFieldCreate;
typedef enum { QUDA_RECONSTRUCT_NO = 18 } QudaReconstructType;
typedef enum { QUDA_GHOST_EXCHANGE_NO, QUDA_GHOST_EXCHANGE_PAD, QUDA_GHOST_EXCHANGE_EXTENDED } QudaGhostExchange;
typedef enum { QUDA_SILENT, QUDA_SUMMARIZE, QUDA_VERBOSE } QudaVerbosity;
typedef enum { QUDA_BOOLEAN_NO = 0, QUDA_BOOLEAN_YES = 1 } QudaBoolean;
typedef struct QudaGaugeParam_s {
  int X[4];
  QudaPrecision cpu_prec, cuda_prec;
  int ga_pad;
} QudaGaugeParam;
typedef struct QudaInvertParam_s {
  QudaVerbosity verbosity;
} QudaInvertParam;
typedef struct QudaEigParam_s {
  QudaInvertParam *invert_param;
  int nEv, nKr;
  double tol;
  QudaBoolean use_poly_acc;
} QudaEigParam;
typedef struct QudaMultigridParam_s {
  int n_level;
  int geo_block_size[QUDA_MAX_MG_LEVEL][4];
  int spin_block_size[QUDA_MAX_MG_LEVEL];
  int n_vec[QUDA_MAX_MG_LEVEL];
} QudaMultigridParam;
// util_quda.h
void stubErrorQuda(const char *, ...);
#define errorQuda(...) stubErrorQuda(__VA_ARGS__)
#define warningQuda(...) stubErrorQuda(__VA_ARGS__)
#define printfQuda(...) stubErrorQuda(__VA_ARGS__)
void pushVerbosity(QudaVerbosity);
void popVerbosity();
enum cudaMemcpyKind { cudaMemcpyDeviceToHost = 2 };
void qudaMemcpy(void *, const void *, size_t, cudaMemcpyKind);
