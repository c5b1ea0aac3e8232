// STUB (tests/quda_stub/README.md): stands where MuGiq's include/enum_mugiq.h stands in a real build.  The enums themselves are the
// library's own copy (include/mugiq_hip_enums.hpp: reference names and values); where /root/reference is mounted the syntax test
// uses the reference's header instead of this one.
#pragma once
#include "mugiq_hip_enums.hpp"
