// STUB (tests/quda_stub/README.md): nothing of it is used by name.
#pragma once
#include <gauge_field.h>
