// compat/opencl_host.h -- forwarding header: with this directory ahead of the
// reference's include/ on the include path, reference src/render.cc compiles
// UNMODIFIED against the HIP render host (`#include "opencl_host.h"` lands here,
// `OpenCLHost` is an alias of `HipHost`).  See INTEGRATION.md section 1.
#pragma once
#include "../hip_host.h"
