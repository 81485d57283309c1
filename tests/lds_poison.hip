// lds_poison.hip -- fill every CU's LDS with a bit pattern (NaNs by default) so that a later kernel that reads LDS it never wrote shows up: LDS keeps its
// contents between kernels, and an uninitialised slot that is multiplied by a zero weight is harmless only while the garbage is finite.
// Test infrastructure (tests/test_lds_hygiene_gpu.py builds and loads it).
#include <hip/hip_runtime.h>
#include <cstdint>

__global__ __launch_bounds__(1024) void poison_kernel(uint64_t pattern, uint64_t* sink) {
  extern __shared__ uint64_t lds[];
  const int n = 160 * 1024 / 8;
  for (int i = threadIdx.x; i < n; i += blockDim.x) lds[i] = pattern;
  __syncthreads();
  // keep the workgroup resident for a moment so that the launch spreads over all CUs (one 160 KB workgroup per CU at a time)
  uint64_t acc = 0;
  for (int r = 0; r < 64; ++r)
    for (int i = threadIdx.x; i < n; i += blockDim.x) acc += lds[i] ^ (uint64_t)r;
  if (acc == 0x1234567ull) sink[0] = acc;
}

extern "C" int lds_poison(uint64_t pattern, int workgroups, void* sink, void* stream) {
  if (hipFuncSetAttribute((const void*)poison_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2;
  poison_kernel<<<workgroups, 1024, 160 * 1024, (hipStream_t)stream>>>(pattern, (uint64_t*)sink);
  return (int)hipGetLastError();
}
