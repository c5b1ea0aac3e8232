// STUB (tests/quda_stub/README.md): the enums of MuGiq's include/enum_mugiq.h the adapter and mugiq_hip_operators.hpp name,
// with the values mugiq_hip_operators.hpp itself documents.  Where /root/reference is mounted the syntax test uses the
// reference's own header instead of this one.
#pragma once
#include <limits.h>
#define MUGIQ_INVALID_ENUM INT_MIN
typedef enum { LOOP_FT_SIGN_MINUS = -1, LOOP_FT_SIGN_PLUS = 1, LOOP_FT_SIGN_INVALID = MUGIQ_INVALID_ENUM } LoopFTSign;
typedef enum { LOOP_CALC_TYPE_BLAS, LOOP_CALC_TYPE_OPT_KERNEL, LOOP_CALC_TYPE_BASIC_KERNEL, LOOP_CALC_TYPE_INVALID = MUGIQ_INVALID_ENUM } LoopCalcType;
typedef enum { DISPLACE_TYPE_COVARIANT = 0, DISPLACE_TYPE_INVALID = MUGIQ_INVALID_ENUM } DisplaceType;
typedef enum { MUGIQ_BOOL_FALSE = 0, MUGIQ_BOOL_TRUE = 1, MUGIQ_BOOL_INVALID = MUGIQ_INVALID_ENUM } MuGiqBool;
typedef enum {
  DispFlag_X = 0, DispFlag_x = 1, DispFlag_Y = 2, DispFlag_y = 3, DispFlag_Z = 4, DispFlag_z = 5, DispFlag_T = 6, DispFlag_t = 7,
  DispFlagNone = MUGIQ_INVALID_ENUM
} DisplaceFlag;
typedef enum { DispDir_x = 0, DispDir_y = 1, DispDir_z = 2, DispDir_t = 3, DispDirNone = MUGIQ_INVALID_ENUM } DisplaceDir;
typedef enum { DispSignMinus = 0, DispSignPlus = 1, DispSignNone = MUGIQ_INVALID_ENUM } DisplaceSign;
