// traffic_twin.hip -- the fused tabletop rollout's HBM traffic with NO env arithmetic and NO step-to-step dependence: what this MI355X sustains for exactly that
// address pattern and read:write mix (per env-step: 12 B of actions read; obs row 48 B, reward 4 B, done 1 B, success 1 B written; outputs time-major [T, N, .]).
// One workgroup = 64 envs (the product's tile) or 256; every step's 3,456 B go out as non-temporal 16-byte stores, actions come in as 16-byte loads.
// A yardstick for DESIGN.md section 4, not product.  Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libtwin.so traffic_twin.hip
#include <hip/hip_runtime.h>
#include <cstdint>

typedef float v4f __attribute__((ext_vector_type(4)));

template <int TILE, bool NT>
__global__ __launch_bounds__(256) void twin_kernel(const v4f* __restrict__ act, v4f* __restrict__ obs, v4f* __restrict__ rew, v4f* __restrict__ done,
                                                   v4f* __restrict__ suc, int N, int T, int xcd_remap) {
  int b = blockIdx.x;
  if (xcd_remap) b = (b % 8) * (gridDim.x / 8) + b / 8;
  const size_t e0 = (size_t)b * TILE;
  const int tid = threadIdx.x;
  constexpr int OV = TILE * 3, AV = TILE * 3 / 4, RV = TILE / 4, FV = TILE / 16;      // 16-byte vectors per step: obs, actions, reward, flags
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0; t < T; ++t) {
    const size_t row = (size_t)t * N + e0;
    for (int i = tid; i < AV; i += 256) { v4f a = act[row * 3 / 4 + i]; acc += a; }
    const v4f v = {acc.x, 2.f, 3.f, (float)t};
    for (int i = tid; i < OV; i += 256) { if (NT) __builtin_nontemporal_store(v, obs + row * 3 + i); else obs[row * 3 + i] = v; }
    for (int i = tid; i < RV; i += 256) { if (NT) __builtin_nontemporal_store(v, rew + row / 4 + i); else rew[row / 4 + i] = v; }
    if (tid >= 64 && tid < 64 + FV) { if (NT) __builtin_nontemporal_store(v, done + row / 16 + tid - 64); else done[row / 16 + tid - 64] = v; }
    if (tid >= 128 && tid < 128 + FV) { if (NT) __builtin_nontemporal_store(v, suc + row / 16 + tid - 128); else suc[row / 16 + tid - 128] = v; }
  }
}

// control: the same bytes (12 read : 54 written per env-step) as ONE contiguous slab per workgroup -- what a plain streaming kernel gets for this mix
__global__ __launch_bounds__(256) void linear_kernel(const v4f* __restrict__ act, v4f* __restrict__ out, int N, int T) {
  const size_t wv = (size_t)T * 64 * 54 / 16, rv = (size_t)T * 64 * 12 / 16;       // 16-byte vectors per workgroup: written, read
  v4f* o = out + (size_t)blockIdx.x * wv;
  const v4f* a = act + (size_t)blockIdx.x * rv;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = threadIdx.x; i < rv; i += 256) acc += a[i];
  for (size_t i = threadIdx.x; i < wv; i += 256) __builtin_nontemporal_store(acc, o + i);
}

extern "C" int twin_launch(const void* act, void* obs, void* rew, void* done, void* suc, int N, int T, int tile, int nt, int xcd_remap, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const v4f* a = (const v4f*)act;
  v4f *o = (v4f*)obs, *r = (v4f*)rew, *d = (v4f*)done, *u = (v4f*)suc;
  if (tile == 64 && nt) twin_kernel<64, true><<<N / 64, 256, 0, s>>>(a, o, r, d, u, N, T, xcd_remap);
  else if (tile == 64) twin_kernel<64, false><<<N / 64, 256, 0, s>>>(a, o, r, d, u, N, T, xcd_remap);
  else if (tile == 256 && nt) twin_kernel<256, true><<<N / 256, 256, 0, s>>>(a, o, r, d, u, N, T, xcd_remap);
  else if (tile == 256) twin_kernel<256, false><<<N / 256, 256, 0, s>>>(a, o, r, d, u, N, T, xcd_remap);
  else if (tile == 1024 && nt) twin_kernel<1024, true><<<N / 1024, 256, 0, s>>>(a, o, r, d, u, N, T, xcd_remap);
  else if (tile == -1) linear_kernel<<<N / 64, 256, 0, s>>>(a, o, N, T);
  else return -1;
  return (int)hipGetLastError();
}
