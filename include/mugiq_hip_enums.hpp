// The enums of MuGiq's include/enum_mugiq.h that the loop path names, with the reference's names and values (source
// compatibility: code written against the reference compiles against mugiq_hip_operators.hpp unchanged).  Inside a MuGiq / QUDA
// build the reference's own header provides them and this file is not included (MUGIQ_HIP_NO_REFERENCE_ENUMS).
#ifndef MUGIQ_HIP_ENUMS_HPP
#define MUGIQ_HIP_ENUMS_HPP
#include <climits>
#define MUGIQ_INVALID_ENUM INT_MIN
// include/enum_mugiq.h:28-97 (values identical)
typedef enum LoopFTSign_s { LOOP_FT_SIGN_MINUS = -1, LOOP_FT_SIGN_PLUS = 1, LOOP_FT_SIGN_INVALID = MUGIQ_INVALID_ENUM } LoopFTSign;
typedef enum LoopCalcType_s {
  LOOP_CALC_TYPE_BLAS,
  LOOP_CALC_TYPE_OPT_KERNEL,
  LOOP_CALC_TYPE_BASIC_KERNEL,
  LOOP_CALC_TYPE_INVALID = MUGIQ_INVALID_ENUM
} LoopCalcType;
typedef enum DisplaceType_s { DISPLACE_TYPE_COVARIANT = 0, DISPLACE_TYPE_INVALID = MUGIQ_INVALID_ENUM } DisplaceType;
typedef enum MuGiqBool_s { MUGIQ_BOOL_FALSE = 0, MUGIQ_BOOL_TRUE = 1, MUGIQ_BOOL_INVALID = MUGIQ_INVALID_ENUM } MuGiqBool;
typedef enum DisplaceFlag_s {
  DispFlagNone = MUGIQ_INVALID_ENUM,
  DispFlag_X = 0, DispFlag_x = 1, DispFlag_Y = 2, DispFlag_y = 3, DispFlag_Z = 4, DispFlag_z = 5, DispFlag_T = 6, DispFlag_t = 7
} DisplaceFlag;
typedef enum DisplaceDir_s { DispDirNone = MUGIQ_INVALID_ENUM, DispDir_x = 0, DispDir_y = 1, DispDir_z = 2, DispDir_t = 3 } DisplaceDir;
typedef enum DisplaceSign_s { DispSignNone = MUGIQ_INVALID_ENUM, DispSignMinus = 0, DispSignPlus = 1 } DisplaceSign;
typedef enum MuGiqBoundaryDirection_s {
  MUGIQ_BOUNDARY_BACKWARD = 0,
  MUGIQ_BOUNDARY_FORWARD = 1,
  MUGIQ_BOUNDARY_INVALID = MUGIQ_INVALID_ENUM
} MuGiqBoundaryDirection;
#endif  // MUGIQ_HIP_ENUMS_HPP
