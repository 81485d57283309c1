// earl_rt.h -- what the per-env device functions of the tabletop path (philox.h, tabletop_device.h, tabletop_step.h) are compiled against.
// gfx950 build (hipcc): the HIP runtime header.  Host build of the SAME functions (g++ -DEARL_HOST_BUILD, csrc/tabletop_host.cpp ->
// libearl_host.so, the `_cpu` entry points of SURVEY 8(b) / BASELINE configs[0]): host_shim.h, which spells the handful of HIP words those
// functions use.  This is the only switch; the arithmetic below it is one source.
#pragma once
#ifdef EARL_HOST_BUILD
#include "host_shim.h"
#else
#include <hip/hip_runtime.h>
#endif
