// mfma_f64_hessian.hip -- VERDICT r04 item 4b: ONE measured experiment on the fp64 matrix pipe.  The steppers' constraint Hessian H = M + J' D J (nv = 23 padded to 32, K rows) for the
// two envs of a wave, built (a) the way the kernels build it today in its generic form -- lane = column, rows of J read from LDS, fp64 FMAs on the vector ALU -- and (b) as
// v_mfma_f64_16x16x4_f64 tiles (three 16 x 16 tiles per env by symmetry, K / 4 steps each, operands read from LDS in the instruction's lane map: A[i = lane & 15][k = lane >> 4],
// B[k = lane >> 4][j = lane & 15], C/D col = lane & 15, row = (lane >> 4) + 4 reg).  Also the raw issue rate / dependent latency of the instruction.  Cycles from s_memtime
// (wall clock of the wave), one wave per SIMD, every CU busy.  Driver: tools/bench_mfma_hessian.py.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double double4_ __attribute__((ext_vector_type(4)));
constexpr int NV = 23, NP = 32, KMAX = 44, EPW = 2, WPB = 4;      // 19.8 KB per env, eight envs per workgroup: the steppers' footprint

struct EnvLds {
  double J[KMAX][NP];      // row k, column l (columns 23..31 zero)
  double D[KMAX];
  double H[NP][NP];        // holds M on entry, M + J' D J on exit
};

__device__ __forceinline__ void wfence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

// mode 0: VALU, lane = column (32 lanes per env); mode 1: MFMA tiles; mode 2: MFMA issue rate (independent accumulators); mode 3: MFMA dependent chain
extern "C" __global__ __launch_bounds__(64 * WPB) void hessian_kernel(const double* __restrict__ Jg, const double* __restrict__ Dg, const double* __restrict__ Mg, double* __restrict__ Hg,
                                                                      unsigned long long* __restrict__ cycles, const int K, const int reps, const int mode) {
  __shared__ EnvLds sh[EPW * WPB];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, sub = lane & 31, grp = lane >> 5;
  const int env = (blockIdx.x * WPB + wave) * EPW + grp;
  EnvLds& s = sh[wave * EPW + grp];
  for (int k = 0; k < K; ++k) s.J[k][sub] = Jg[((size_t)env * KMAX + k) * NP + sub];
  for (int k = sub; k < K; k += 32) s.D[k] = Dg[(size_t)env * KMAX + k];
  for (int i = 0; i < NP; ++i) s.H[i][sub] = Mg[((size_t)env * NP + i) * NP + sub];
  wfence();
  const unsigned long long t0 = __builtin_readcyclecounter();
  double sink = 0;
  for (int r = 0; r < reps; ++r) {
    if (mode == 0) {
      double h[NP];
#pragma unroll
      for (int i = 0; i < NP; ++i) h[i] = s.H[i][sub];
      for (int k = 0; k < K; ++k) {
        const double w = s.D[k] * s.J[k][sub];
#pragma unroll
        for (int i = 0; i < NV; ++i) h[i] = fma(w, s.J[k][i], h[i]);
      }
#pragma unroll
      for (int i = 0; i < NP; ++i) s.H[i][sub] = h[i];
    } else if (mode == 1) {
      const int i16 = lane & 15, kk = lane >> 4;
      for (int e = 0; e < EPW; ++e) {                       // the whole wave works on one env's matrices at a time
        EnvLds& se = sh[wave * EPW + e];
        double4_ t00, t10, t11;
#pragma unroll
        for (int v = 0; v < 4; ++v) {                       // C = M: row = kk + 4 v, col = i16
          t00[v] = se.H[kk + 4 * v][i16]; t10[v] = se.H[16 + kk + 4 * v][i16]; t11[v] = se.H[16 + kk + 4 * v][16 + i16];
        }
        for (int k0 = 0; k0 < K; k0 += 4) {
          const double d = se.D[k0 + kk], j0 = se.J[k0 + kk][i16], j1 = se.J[k0 + kk][16 + i16];
          const double a0 = d * j0, a1 = d * j1;
          t00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, j0, t00, 0, 0, 0);
          t10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, j0, t10, 0, 0, 0);
          t11 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, j1, t11, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          se.H[kk + 4 * v][i16] = t00[v]; se.H[16 + kk + 4 * v][i16] = t10[v]; se.H[i16][16 + kk + 4 * v] = t10[v]; se.H[16 + kk + 4 * v][16 + i16] = t11[v];
        }
      }
    } else if (mode == 2) {                                 // issue rate: eight independent accumulators
      double4_ acc[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] = double4_{0, 0, 0, 0};
      const double a = s.J[0][sub], b = s.J[1][sub];
      for (int k0 = 0; k0 < K; ++k0) {
#pragma unroll
        for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[q], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 8; ++q) sink += acc[q][0] + acc[q][3];
    } else {                                                // dependent chain
      double4_ acc = double4_{0, 0, 0, 0};
      const double a = s.J[0][sub], b = s.J[1][sub];
      for (int k0 = 0; k0 < 8 * K; ++k0) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
      sink += acc[0] + acc[3];
    }
    wfence();
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 0) cycles[blockIdx.x * WPB + wave] = t1 - t0;
  for (int i = 0; i < NP; ++i) Hg[((size_t)env * NP + i) * NP + sub] = s.H[i][sub] + (sink == 12345.678 ? 1.0 : 0.0);
}

extern "C" int run_hessian(const double* J, const double* D, const double* M, double* H, unsigned long long* cycles, int nwg, int K, int reps, int mode) {
  hessian_kernel<<<nwg, 64 * WPB>>>(J, D, M, H, cycles, K, reps, mode);
  return (int)hipDeviceSynchronize();
}
