// host_shim.h -- the HIP vocabulary of csrc/tabletop_device.h / tabletop_step.h / philox.h for a plain C++ host compile (g++, no HIP
// headers, no GPU): qualifiers vanish, the two vector types become structs, __umulhi becomes a 64-bit multiply.  Included only through
// earl_rt.h under -DEARL_HOST_BUILD (csrc/tabletop_host.cpp).
#pragma once
#include <cmath>
#include <cstdint>

#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))

struct alignas(16) double2 { double x, y; };
struct alignas(16) float4 { float x, y, z, w; };

static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32); }
using std::exp;
using std::fma;
using std::sqrt;
