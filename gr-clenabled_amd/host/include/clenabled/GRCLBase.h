// What callers of the reference take from include/clenabled/GRCLBase.h: the item-type, device-class and device-selector
// codes (:57-70).  The GRCLBase CLASS (OpenCL context / queue / run-time kernel compilation, :77-141) has no counterpart
// here by design: the blocks sit on the C ABI of include/mi355_clenabled.h (mi355_ctx_*), see DESIGN.md section 1.
#pragma once
#include "api.h"
#include "clSComplex.h"

constexpr int DTYPE_COMPLEX = 1, DTYPE_FLOAT = 2, DTYPE_INT = 3, DTYPE_SHORT = 4, DTYPE_BYTE = 5, DTYPE_PACKEDXY = 6;
constexpr int OCLTYPE_GPU = 1, OCLTYPE_ACCELERATOR = 2, OCLTYPE_CPU = 3, OCLTYPE_ANY = 4;
constexpr int OCLDEVICESELECTOR_FIRST = 1, OCLDEVICESELECTOR_SPECIFIC = 2;
