// tabletop_rollout_ws.h -- wave-specialised fused rollout kernel (the bench's dominant kernel).
//
// Why: a T-step rollout is a bit-exact fp64 recurrence per env, so time cannot be parallelised; at
// N = 4096 there are only 64 wavefronts and the launch is bound by ONE wave's per-step latency.  The
// plain one-lane-per-env loop (rollout_kernel) puts loads, ~100 instructions of formatting and six
// partially-coalesced stores on that critical path, and because vmcnt retires in order every wait for
// an action load also waits for all older stores.  Here a 64-env workgroup splits the step by ROLE:
//
//   wave 0        COMPUTE  the recurrence only (~30 VALU/step): reads rescaled actions from LDS, writes
//                          the 4 f32 coordinates + attached flag into an LDS "row image" of the obs rows
//   wave 1        LOADER   streams raw actions HBM -> VGPR ring (LEAD chunks ahead, coalesced dword loads, no
//                          stores in this wave so its vmcnt waits never queue behind stores), transposes them
//                          through LDS and rescales them to fp64 -> LDS
//   waves 2..     STORERS  copy finished row images LDS -> HBM as fully coalesced float4 (the goal part of
//                          each row is constant and pre-filled), evaluate success/reward, pack flags
//
// One workgroup barrier per chunk of K steps; 3-deep action ring and 2-deep row-image ring in LDS.
// Semantics are exactly those of rollout_kernel<1> without lifelong / auto-reset (the host picks the
// kernel); outputs are bit-identical (tests/test_tabletop_gpu.py::test_rollout_kernels_agree).
#pragma once
#include <type_traits>

#include "tabletop_device.h"

namespace earl {

struct WsArgs {
  int32_t n, T, horizon, wide;
  const float* __restrict__ act;      // [T, n, 3]
  double* __restrict__ qpos;          // [n, 4]
  int8_t* __restrict__ attached;      // [n]
  const int32_t* __restrict__ goal_idx;
  const double* __restrict__ goal_table;
  int32_t* __restrict__ steps_since_reset;
  float* __restrict__ obs;            // [T, n, 12]
  float* __restrict__ reward;         // [T, n]
  uint8_t* __restrict__ done;         // [T, n]
  uint8_t* __restrict__ success;      // [T, n]
  Thresholds th;
};

// Lanes of ONE wave exchange data through LDS (loader staging).  Per thread the write and read addresses never
// alias, so without a fence hipcc reorders them freely (it did: seen in the ISA); a wavefront-scope fence costs no
// instruction -- the LDS queue of a wave is already in order -- it only pins the compiler's order.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// fast clip for finite-or-inf values; NaN handling is done by the EXACT path
__device__ __forceinline__ double clip_fast(double x) { return fmin(fmax(x, -2.8), 2.8); }

template <bool EXACT>
__device__ __forceinline__ void ws_step(double& fx, double& fy, double& ox, double& oy, bool& att, double a0,
                                        double a1, bool grip, const Thresholds& th) {
  // move :140-174 with the grasp test folded into mask logic; EXACT = np.clip NaN propagation (compare+select)
  const double dx = fx - ox, dy = fy - oy;
  const bool near = fma(dy, dy, dx * dx) < th.grasp_d2;
  att = grip && (att || near);
  double nfx, nfy;
  if constexpr (EXACT) {
    nfx = clipd(fx + a0, -2.8, 2.8);
    nfy = clipd(fy + a1, -2.8, 2.8);
  } else {
    nfx = clip_fast(fx + a0);
    nfy = clip_fast(fy + a1);
  }
  const double ddx = nfx - fx, ddy = nfy - fy;
  double nox, noy;
  if constexpr (EXACT) {
    nox = clipd(ox + ddx, -2.8, 2.8);
    noy = clipd(oy + ddy, -2.8, 2.8);
  } else {
    nox = clip_fast(ox + ddx);
    noy = clip_fast(oy + ddy);
  }
  ox = att ? nox : ox;
  oy = att ? noy : oy;
  fx = nfx;
  fy = nfy;
}

template <int RT, int NS, int K, int LEAD>
__global__ __launch_bounds__(64 * (2 + NS)) void rollout_ws_kernel(const WsArgs a) {
  constexpr int E = 64;
  __shared__ double2 A[3][K][E];      // rescaled (a0, a1)
  __shared__ uint8_t G[3][K][E];      // bit0: rescaled grip > 0, bit1: a0 or a1 is NaN
  __shared__ int slow_flag[3];        // chunk contains a NaN action -> exact-NaN path
  __shared__ float4 R[2][K][E * 3];   // row images: 64 obs rows of 48 B per step
  __shared__ float S[K][E * 3];       // loader-private staging: raw actions of one chunk (coalesced -> per env)

  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int n = a.n, T = a.T;
  const int i0 = blockIdx.x * E;
  const int i = i0 + lane;
  const int valid = min(E, n - i0);   // live envs of this workgroup
  const bool live = lane < valid;
  const int nch = (T + K - 1) / K;

  // ---- every thread: pre-fill the constant goal part of all row images (parts 1 and 2 of each row)
  {
    constexpr int per_buf = K * E;
    for (int idx = threadIdx.x; idx < 2 * per_buf; idx += 64 * (2 + NS)) {
      const int e = idx % E;
      float g[6] = {0, 0, 0, 0, 0, 0};
      if (e < valid) load_goal<1>(a.goal_table, a.goal_idx[i0 + e], g);
      float4* row = &R[0][0][0] + (size_t)(idx / E) * (E * 3) + e * 3;
      row[1] = float4{-1.0f, -1.0f, g[0], g[1]};
      row[2] = float4{g[2], g[3], g[4], g[5]};
    }
  }

  if (wave == 0) {
    // ================================================================= COMPUTE
    double fx = 0, fy = 0, ox = 0, oy = 0;
    bool att = false;
    if (live) {
      const double2* q2 = reinterpret_cast<const double2*>(a.qpos + (size_t)i * 4);
      const double2 u = q2[0], v = q2[1];
      fx = u.x; fy = u.y; ox = v.x; oy = v.y;
      att = a.attached[i] >= 0;
    }
    bool slow = __any((fx != fx) || (fy != fy) || (ox != ox) || (oy != oy));   // sticky, wave-uniform
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
      const int ab = c % 3, rb = c & 1;
      slow = slow || (slow_flag[ab] != 0);
      double2 av[K];
      uint8_t gv[K];
#pragma unroll
      for (int k = 0; k < K; ++k) { av[k] = A[ab][k][lane]; gv[k] = G[ab][k][lane]; }
      auto run = [&](auto exact) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          if (c * K + k >= T) break;      // tail chunk (wave-uniform)
          ws_step<decltype(exact)::value>(fx, fy, ox, oy, att, av[k].x, av[k].y, (gv[k] & 1) != 0, a.th);
          float4* row = &R[rb][k][lane * 3];
          row[0] = float4{(float)fx, (float)fy, (float)ox, (float)oy};
          const float flag = att ? 0.0f : -1.0f;
          *reinterpret_cast<float2*>(&row[1]) = float2{flag, flag};
        }
      };
      if (__builtin_amdgcn_readfirstlane(slow ? 1 : 0)) run(std::true_type{});
      else run(std::false_type{});
      __syncthreads();
    }
    if (live) {
      double2* q2 = reinterpret_cast<double2*>(a.qpos + (size_t)i * 4);
      q2[0] = double2{fx, fy};
      q2[1] = double2{ox, oy};
      a.attached[i] = att ? 0 : -1;
      a.steps_since_reset[i] += T;
    }
  } else if (wave == 1) {
    // ================================================================= LOADER
    // The actions of one step for this workgroup are 192 consecutive floats.  Lane l loads floats l, l+64, l+128
    // (three fully coalesced 256-B wave loads per step, each into its own VGPR so the ring below needs no
    // register shuffling), LEAD chunks ahead of their use.  Loads are UNCONDITIONAL with clamped indices: a load
    // under a branch, or a multi-dword load whose lanes are later split, makes hipcc wait vmcnt(0) right behind
    // it, which serialises the prefetch ring (seen in the ISA of earlier versions).
    float raw[LEAD][K][3];
    const int last = max(valid * 3 - 1, 0);
    const int e0 = min(lane, last), e1 = min(lane + 64, last), e2 = min(lane + 128, last);
    auto issue = [&](int d, int j) {   // raw[d] <- actions of chunk j
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const int t = min(j * K + k, T - 1);
        const float* p = a.act + ((size_t)t * n + i0) * 3;
        raw[d][k][0] = p[e0]; raw[d][k][1] = p[e1]; raw[d][k][2] = p[e2];
      }
    };
    auto process = [&](int d, int j) { // transpose through LDS, rescale, publish chunk j in slot j % 3
      const int ab = j % 3;
      bool any_nan = false;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        float* sk = &S[k][0];
        sk[lane] = raw[d][k][0]; sk[lane + 64] = raw[d][k][1]; sk[lane + 128] = raw[d][k][2];
      }
      wave_lds_fence();
#pragma unroll
      for (int k = 0; k < K; ++k) {
        const float* sk = &S[k][0];   // same wave wrote it: LDS operations of one wave execute in order
        const double a0 = rescale_action(sk[lane * 3]), a1 = rescale_action(sk[lane * 3 + 1]);
        const double a2 = rescale_action(sk[lane * 3 + 2]);
        const bool nan = (a0 != a0) || (a1 != a1);
        any_nan = any_nan || nan;
        if (j < nch) {
          A[ab][k][lane] = double2{a0, a1};
          G[ab][k][lane] = (uint8_t)((a2 > 0 ? 1 : 0) | (nan ? 2 : 0));
        }
      }
      const bool wave_nan = __any(any_nan);
      if (lane == 0 && j < nch) slow_flag[ab] = wave_nan ? 1 : 0;
      wave_lds_fence();   // the next chunk's staging writes must stay behind these reads
    };
#pragma unroll
    for (int d = 0; d < LEAD; ++d) issue(d, d);
    for (int j0 = 0; j0 < nch + 2; j0 += LEAD) {
#pragma unroll
      for (int d = 0; d < LEAD; ++d) {
        const int j = j0 + d;
        process(d, j);
        issue(d, j + LEAD);
        if (j >= 1 && j < nch + 2) __syncthreads();
      }
    }
  } else {
    // ================================================================= STORERS
    const int s = wave - 2;
    float g[6] = {0, 0, 0, 0, 0, 0};
    int steps0 = 0;
    if (live) {
      load_goal<1>(a.goal_table, a.goal_idx[i], g);
      steps0 = a.steps_since_reset[i];
    }
    // steps0 must be in a register before the compute wave can possibly update steps_since_reset
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const bool packed_flags = (valid == E) && ((n & 3) == 0);
    auto store_chunk = [&](int c) {
      const int rb = c & 1;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if ((k % NS) != s) continue;     // storers interleave over the steps of a chunk
        const int t = c * K + k;
        if (t >= T) continue;
        const size_t row0 = (size_t)t * n + i0;
        // (1) the 64 obs rows of this step: 192 float4, contiguous in HBM and in LDS
        float4* dst = reinterpret_cast<float4*>(a.obs + row0 * 12);
        const float4* src = &R[rb][k][0];
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          const int idx = lane + 64 * m;
          if (idx < valid * 3) dst[idx] = src[idx];
        }
        // (2) reward / success / done of env `lane`
        const float4 p = src[lane * 3];
        float o[12] = {p.x, p.y, p.z, p.w, 0.f, 0.f, g[0], g[1], g[2], g[3], g[4], g[5]};
        const bool succ = success1(o, a.wide, a.th);
        float rew;
        if constexpr (RT == EARL_REWARD_SPARSE) rew = succ ? 1.0f : 0.0f;
        else rew = (float)dense1(o);
        const bool dn = steps0 + t + 1 >= a.horizon;
        if (live) a.reward[row0 + lane] = rew;
        if (packed_flags) {   // 64 flag bytes of a step = 16 dwords: expand ballot nibbles to bytes
          const unsigned long long ms = __ballot(succ), md = __ballot(dn);
          if (lane < 16) {
            const uint32_t ns = (uint32_t)(ms >> (4 * lane)) & 0xFu, nd = (uint32_t)(md >> (4 * lane)) & 0xFu;
            reinterpret_cast<uint32_t*>(a.success + row0)[lane] = (ns * 0x00204081u) & 0x01010101u;
            reinterpret_cast<uint32_t*>(a.done + row0)[lane] = (nd * 0x00204081u) & 0x01010101u;
          }
        } else if (live) {
          a.success[row0 + lane] = succ;
          a.done[row0 + lane] = dn;
        }
      }
    };
    __syncthreads();
    for (int c = 0; c < nch; ++c) {
      if (c >= 1) store_chunk(c - 1);
      __syncthreads();
    }
    store_chunk(nch - 1);
  }
}

}  // namespace earl
