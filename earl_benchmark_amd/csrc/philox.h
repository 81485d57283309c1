// philox.h -- Philox4x32-10 counter-based RNG shared by the env kernels (draw layout: see the callers and the oracle)
#pragma once
#include "earl_rt.h"

#include <cstdint>

namespace earl {

struct U4 { uint32_t x, y, z, w; };
__device__ __forceinline__ U4 philox4x32_10(U4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(0xD2511F53u, c.x), lo0 = 0xD2511F53u * c.x;
    const uint32_t hi1 = __umulhi(0xCD9E8D57u, c.z), lo1 = 0xCD9E8D57u * c.z;
    c = U4{hi1 ^ c.y ^ k0, lo1, hi0 ^ c.w ^ k1, lo0};
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}
// 53-bit uniform in [0, 1) from two words
__device__ __forceinline__ double u01(uint32_t lo, uint32_t hi) {
  return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}

}  // namespace earl
