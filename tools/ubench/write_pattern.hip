// write_pattern.hip -- what HBM write bandwidth do the OUTPUT ADDRESS PATTERNS of the fused rollout reach, with no compute at all?
//   linear      : workgroup w streams its own contiguous slab (what a fill does)
//   time-major  : for every step t, workgroup w writes the 3 KB of its 64 envs at (t * N + 64 w) * 48 B  (obs [T, N, 12] f32)
//   time-major x4 : the same with 256 envs (12 KB) per workgroup and step
// Build: hipcc --offload-arch=gfx950 -O3 -o write_pattern write_pattern.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { if ((x) != hipSuccess) { printf("HIP error at %s:%d\n", __FILE__, __LINE__); exit(1); } } while (0)

__global__ void linear_k(float4* out, size_t per_wg_vec, int T) {
  float4* p = out + (size_t)blockIdx.x * per_wg_vec;
  const float4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  for (size_t i = threadIdx.x; i < per_wg_vec; i += blockDim.x) p[i] = v;
}
// tile = envs per workgroup; each env row is 3 float4
__global__ void time_major_k(float4* out, int N, int T, int tile) {
  const float4 v = {1.f, 2.f, 3.f, (float)blockIdx.x};
  const int vec_per_step = tile * 3;
  for (int t = 0; t < T; ++t) {
    float4* p = out + ((size_t)t * N + (size_t)blockIdx.x * tile) * 3;
    for (int i = threadIdx.x; i < vec_per_step; i += blockDim.x) p[i] = v;
  }
}

int main(int argc, char** argv) {
  const int T = 200;
  for (int N : {16384, 65536, 262144, 1048576}) {
    const size_t bytes = (size_t)N * T * 48;
    float4* d;
    if (hipMalloc(&d, bytes) != hipSuccess) { printf("N=%d: alloc failed\n", N); continue; }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](auto&& launch, const char* name) {
      for (int i = 0; i < 2; ++i) launch();
      CK(hipDeviceSynchronize());
      const int reps = N >= 262144 ? 5 : 20;
      CK(hipEventRecord(e0));
      for (int i = 0; i < reps; ++i) launch();
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("N=%8d %-16s %8.1f us  %6.2f TB/s\n", N, name, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12);
    };
    const int wg64 = N / 64, wg256 = N / 256;
    timeit([&] { linear_k<<<wg64, 256>>>(d, bytes / 16 / wg64, T); }, "linear");
    timeit([&] { time_major_k<<<wg64, 192>>>(d, N, T, 64); }, "time-major 64");
    timeit([&] { time_major_k<<<wg256, 256>>>(d, N, T, 256); }, "time-major 256");
    timeit([&] { time_major_k<<<N / 1024, 512>>>(d, N, T, 1024); }, "time-major 1024");
    CK(hipFree(d));
  }
  return 0;
}
