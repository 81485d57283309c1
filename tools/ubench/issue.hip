// Single-wave issue/latency microbenchmark (gfx950): cycles per instruction for dependent chains and for
// ILP = 2 / 4 independent chains of fp64 add / max / fma, f32 add, cndmask, SALU.  One wave on one CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 256
template <int MODE, int ILP>
__global__ void k(unsigned long long* out, double seed) {
  double x[4] = {seed, seed + 1, seed + 2, seed + 3};
  float y[4] = {(float)seed, 1.f, 2.f, 3.f};
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < ILP; ++j) {
      if (MODE == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[j]) : "v"(seed));
      if (MODE == 1) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x[j]) : "v"(seed));
      if (MODE == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x[j]) : "v"(seed));
      if (MODE == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(y[j]) : "v"(y[3]));
      if (MODE == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(y[j]) : "v"(y[3]) : );
      if (MODE == 5) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(y[j]) : "v"(x[j]));
      if (MODE == 6) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[j]) : "v"(seed));
    }
  }
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if (threadIdx.x == 0) out[0] = t1 - t0;
  if (x[0] + x[1] + x[2] + x[3] + y[0] + y[1] + y[2] == 12345.678) out[1] = 1;
}
template <int MODE, int ILP>
void run(const char* name, unsigned long long* d) {
  unsigned long long h = 0;
  for (int r = 0; r < 3; ++r) { k<MODE, ILP><<<1, 64>>>(d, 1.0); hipDeviceSynchronize(); }
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("%-12s ILP=%d : %6.2f cycles/instr (%.2f per group)\n", name, ILP, (double)h / (N * ILP), (double)h / N);
}
int main() {
  unsigned long long* d; hipMalloc(&d, 64);
  run<0,1>("v_add_f64", d); run<0,2>("v_add_f64", d); run<0,4>("v_add_f64", d);
  run<1,1>("v_max_f64", d); run<1,2>("v_max_f64", d); run<1,4>("v_max_f64", d);
  run<2,1>("v_fma_f64", d); run<2,2>("v_fma_f64", d); run<2,4>("v_fma_f64", d);
  run<6,1>("v_mul_f64", d); run<6,4>("v_mul_f64", d);
  run<3,1>("v_add_f32", d); run<3,2>("v_add_f32", d); run<3,4>("v_add_f32", d);
  run<4,1>("v_cndmask", d); run<4,4>("v_cndmask", d);
  run<5,1>("v_cvt_f32_f64", d); run<5,4>("v_cvt_f32_f64", d);
  return 0;
}
