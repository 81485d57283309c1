// CU-level issue capacity (gfx950): W waves in ONE workgroup (one CU), each running independent chains (ILP 4) of one
// instruction type; reports aggregate wave-instructions per cycle (s_memtime ticks) of the CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N 512
template <int MODE>
__global__ void k(unsigned long long* out, double seed) {
  double x[4] = {seed, seed + 1, seed + 2, seed + 3};
  float y[4] = {(float)seed, 1.f, 2.f, 3.f};
  __syncthreads();
  unsigned long long t0, t1;
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MODE == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[j]) : "v"(seed));
      if (MODE == 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(y[j]) : "v"(y[3]));
      if (MODE == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x[j]) : "v"(seed));
      if (MODE == 3) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
      if (MODE == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(y[j]) : "v"(y[3]));
      if (MODE == 5) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(y[j]) : "v"(y[3]) : "s20", "s21");
      if (MODE == 6) asm volatile("v_cmp_lt_f64 vcc, %0, %1" :: "v"(x[j]), "v"(seed) : "vcc");
      if (MODE == 7) asm volatile("v_cmp_lt_f64_e64 s[20:21], %0, %1" :: "v"(x[j]), "v"(seed) : "s20", "s21");
      if (MODE == 8) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x[j]) : "v"(seed));
      if (MODE == 9) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(y[j]) : "v"(x[j]));
      if (MODE == 10) asm volatile("v_mov_b32 %0, %1" : "=v"(y[j]) : "v"(y[3]));
      if (MODE == 11) asm volatile("s_nop 1\n v_permlane32_swap_b32 %0, %1" : "+v"(y[j]), "+v"(y[(j + 1) & 3]));
      if (MODE == 12) asm volatile("s_and_b64 s[20:21], s[22:23], s[24:25]" ::: "s20", "s21");
      if (MODE == 13) asm volatile("v_readfirstlane_b32 s20, %0" :: "v"(y[j]) : "s20");
    }
  }
  asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
  if (x[0] + x[1] + x[2] + x[3] + y[0] + y[1] + y[2] == 12345.678) out[63] = 1;
}
template <int MODE>
void run(const char* name, unsigned long long* d) {
  for (int W : {1, 4, 8, 16}) {
    unsigned long long h[16] = {0};
    for (int r = 0; r < 3; ++r) { k<MODE><<<1, 64 * W>>>(d, 1.0); hipDeviceSynchronize(); }
    hipMemcpy(h, d, 8 * W, hipMemcpyDeviceToHost);
    unsigned long long mx = 0; for (int w = 0; w < W; ++w) mx = h[w] > mx ? h[w] : mx;
    printf("%-10s waves/CU=%2d : per-wave %5.2f ticks/instr, CU aggregate %5.2f instr/tick\n", name, W, (double)mx / (N * 4), (double)W * N * 4 / mx);
  }
}
int main() {
  unsigned long long* d; hipMalloc(&d, 1024);
  run<4>("cndmask vcc", d); run<5>("cndmask sgpr", d); run<6>("v_cmp->vcc", d); run<7>("v_cmp->sgpr", d); run<8>("v_max_f64", d);
  run<9>("cvt_f32_f64", d); run<10>("v_mov_b32", d); run<11>("nop+permswap", d); run<12>("s_and_b64", d); run<13>("readfirstlane", d);
  return 0;
}
