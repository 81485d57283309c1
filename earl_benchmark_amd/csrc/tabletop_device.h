// tabletop_device.h -- per-env device functions of the tabletop hot path (gfx950).
//
// One LANE owns one env: the whole state is 4 (or 8) doubles + a few ints, so it lives in VGPRs;
// 64 envs per wavefront, no cross-lane traffic in the arithmetic.  Everything here must be compiled
// with -ffp-contract=off: the reference's fp64 expressions are separately rounded and the only
// fused operations are the explicit fma() calls below (see oracle/tabletop_oracle.c for the probes).
//
// Reference being restated: /root/reference/earl_benchmark/envs/tabletop_manipulation.py (cited per
// function), envs/tabletop_manipulation_3obj.py, wrappers/persistent_state_wrapper.py,
// wrappers/lifelong_wrapper.py.
#pragma once
#include "philox.h"
#include "earl_rt.h"
#include <stdint.h>

#include "../../include/earl_tabletop.h"

namespace earl {

// Exact thresholds that replace correctly-rounded square roots in comparisons.  sqrt is monotone, so
//   sqrt(d2) <  c   <=>  d2 <  min{ s : sqrt(s) >= c }      (fp64; grasp radius 0.4, valid-init radius 1)
//   (double)sqrtf(s) <= c  <=>  s <= max{ s : (double)sqrtf(s) <= c }   (fp32; success radii 0.2 / 0.4)
// The host computes them once with IEEE sqrt (tabletop.hip: compute_thresholds) and passes them by value.
struct Thresholds {
  double grasp_d2;   // sqrt(d2) < 0.4   (tabletop_manipulation.py:150)
  double valid_d2;   // sqrt(d2) < 1     (:90, :94)
  float succ_s;      // (double)sqrtf(s) <= 0.2  (:202, :204; numpy-1.22 promotion)
  float succ3_s;     // (double)sqrtf(s) <= 0.4  (tabletop_manipulation_3obj.py:159)
};

template <int NOBJ>
struct Dims {
  static constexpr int NQ = 2 + 2 * NOBJ;   // qpos entries kept (the dummy joint is dropped, :57)
  static constexpr int NG = NQ + 2;         // goal row: qpos-shaped + the two attached-flag slots
  static constexpr int NOBS = NQ + 2 + NG;  // 12 / 20
};

// draw layout (shared with the oracle): ctr = {draw, global env id, counter lo, counter hi}, key = seed
__device__ __forceinline__ U4 draw_block(const earl_tabletop_cfg& cfg, uint64_t counter, int env, uint32_t draw) {
  return philox4x32_10(U4{draw, (uint32_t)(cfg.env_offset + env), (uint32_t)counter, (uint32_t)(counter >> 32)},
                       (uint32_t)cfg.seed, (uint32_t)(cfg.seed >> 32));
}
__device__ __forceinline__ int sample_goal(const earl_tabletop_cfg& cfg, uint64_t counter, int env,
                                           const int32_t* __restrict__ next_goal_idx) {
  if (next_goal_idx) return next_goal_idx[env];
  const U4 b = draw_block(cfg, counter, env, 0);
  return (int)__umulhi(b.x, (uint32_t)cfg.n_sample_goals);
}

// ---------------------------------------------------------------- arithmetic helpers
// np.clip semantics (NaN propagates): compare+select, not fmin/fmax
__device__ __forceinline__ double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

// step :130-132 -- the clip to [-1,1] is exact in f32 (the bounds are representable), then promote
__device__ __forceinline__ double rescale_action(float a) {
  const float c = a < -1.0f ? -1.0f : (a > 1.0f ? 1.0f : a);
  return -0.2 + (((double)c + 1.) * 0.5) * (0.2 - -0.2);
}

// numpy float32 dot: float-rounded products accumulated in double, rounded to float once
template <int N>
__device__ __forceinline__ float sqnorm_f32(const float (&d)[N]) {
  double acc = 0.0;
#pragma unroll
  for (int i = 0; i < N; ++i) acc += (double)(d[i] * d[i]);
  return (float)acc;
}

// ---------------------------------------------------------------- env state in registers
template <int NOBJ>
struct Env {
  double q[Dims<NOBJ>::NQ];
  int attached;  // -1 free, k = holding object k
};

// move :140-174 (3obj :96-134)
template <int NOBJ>
__device__ __forceinline__ void move(Env<NOBJ>& e, double a0, double a1, double a2, const Thresholds& th) {
  const double fx = e.q[0], fy = e.q[1];
  if (a2 > 0) {
    if (e.attached < 0) {
      if constexpr (NOBJ == 1) {
        const double dx = fx - e.q[2], dy = fy - e.q[3];
        if (fma(dy, dy, dx * dx) < th.grasp_d2) e.attached = 0;  // dist < 0.4 (:150)
      } else {
        double best = __builtin_inf();  // closest object inside the radius wins; ties keep the first (:146-152)
#pragma unroll
        for (int k = 0; k < NOBJ; ++k) {
          const double dx = fx - e.q[2 + 2 * k], dy = fy - e.q[3 + 2 * k];
          const double dist = sqrt(fma(dy, dy, dx * dx));  // correctly rounded fp64 sqrt: dist values are compared
          if (dist < 0.4 && dist < best) { e.attached = k; best = dist; }
        }
      }
    }
  } else {
    e.attached = -1;  // :153-154
  }
  const double nfx = clipd(fx + a0, -2.8, 2.8), nfy = clipd(fy + a1, -2.8, 2.8);  // :156-157
  const double ddx = nfx - fx, ddy = nfy - fy;  // the CLIPPED gripper delta (:162)
  if constexpr (NOBJ == 1) {
    if (e.attached >= 0) {
      e.q[2] = clipd(e.q[2] + ddx, -2.8, 2.8);
      e.q[3] = clipd(e.q[3] + ddy, -2.8, 2.8);
    }
  } else {
#pragma unroll
    for (int k = 0; k < NOBJ; ++k)  // static indexing keeps q[] in registers
      if (e.attached == k) {
        e.q[2 + 2 * k] = clipd(e.q[2 + 2 * k] + ddx, -2.8, 2.8);
        e.q[3 + 2 * k] = clipd(e.q[3 + 2 * k] + ddy, -2.8, 2.8);
      }
  }
  e.q[0] = nfx;
  e.q[1] = nfy;
}

// _get_obs :55-60 -- f32 rounding of the state, flag pair, goal row
template <int NOBJ>
__device__ __forceinline__ void make_obs(const Env<NOBJ>& e, const float (&g)[Dims<NOBJ>::NG], float (&o)[Dims<NOBJ>::NOBS]) {
  constexpr int NQ = Dims<NOBJ>::NQ;
#pragma unroll
  for (int i = 0; i < NQ; ++i) o[i] = (float)e.q[i];
  const float flag = e.attached < 0 ? -1.0f : 0.5f * (float)e.attached;  // (-1,-1) | (0,0),(.5,.5),(1,1)
  o[NQ] = flag;
  o[NQ + 1] = flag;
#pragma unroll
  for (int i = 0; i < Dims<NOBJ>::NG; ++i) o[NQ + 2 + i] = g[i];
}

template <int NOBJ>
__device__ __forceinline__ void load_goal(const double* __restrict__ goal_table, int idx, float (&g)[Dims<NOBJ>::NG]) {
  const double* row = goal_table + (size_t)idx * Dims<NOBJ>::NG;
#pragma unroll
  for (int i = 0; i < Dims<NOBJ>::NG; ++i) g[i] = (float)row[i];
}

// is_successful :197-204 / compute_reward :176-191 on an f32 observation (1 object)
__device__ __forceinline__ bool success1(const float (&o)[12], int wide, const Thresholds& th) {
  if (wide) {
    const float d[2] = {o[2] - o[8], o[3] - o[9]};
    return sqnorm_f32(d) <= th.succ_s;
  }
  const float d[4] = {o[0] - o[6], o[1] - o[7], o[2] - o[8], o[3] - o[9]};
  return sqnorm_f32(d) <= th.succ_s;
}
__device__ __forceinline__ double dense1(const float (&o)[12]) {
  const float d[2] = {o[2] - o[8], o[3] - o[9]};
  const float n1 = sqrtf(sqnorm_f32(d));
  double reward = (double)(-n1);
  const float n1sq = n1 * n1;                       // np.float32 ** 2 stays float32
  reward += 2. * exp((double)(-n1sq) / 0.01);       // float32 / python float -> float64 (numpy 1.22)
  const float e[2] = {o[0] - o[2], o[1] - o[3]};
  const double g = 0.5 * (double)sqrtf(sqnorm_f32(e));
  reward += -g;
  reward += 0.5 * exp(-(g * g) / 0.01);
  return reward;
}
// 3obj is_successful :153-159 / compute_reward :136-151
__device__ __forceinline__ bool success3(const float (&o)[20], const Thresholds& th) {
  float d[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) d[i] = o[i] - o[10 + i];
  return sqnorm_f32(d) <= th.succ3_s;
}
__device__ __forceinline__ double dense3(const float (&o)[20]) {
  float d[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) d[i] = o[2 + i] - o[12 + i];
  double reward = (double)(-sqrtf(sqnorm_f32(d)));
#pragma unroll
  for (int k = 1; k < 4; ++k) {
    const float e[2] = {o[2 * k] - o[2 * k + 10], o[2 * k + 1] - o[2 * k + 11]};
    const float n = sqrtf(sqnorm_f32(e));
    const float nsq = n * n;
    reward += 2. * exp((double)(-nsq) / 0.01);
  }
  return reward;
}

template <int NOBJ>
__device__ __forceinline__ void reward_success(const float (&o)[Dims<NOBJ>::NOBS], int reward_type, int wide,
                                               const Thresholds& th, double& reward, bool& succ) {
  if constexpr (NOBJ == 1) {
    succ = success1(o, wide, th);
    reward = reward_type == EARL_REWARD_SPARSE ? (succ ? 1.0 : 0.0) : dense1(o);
  } else {
    succ = success3(o, th);
    reward = reward_type == EARL_REWARD_SPARSE ? (succ ? 1.0 : 0.0) : dense3(o);
  }
}

// is_valid_init :89-97 (always against the MODULE's goal_states :12-16)
__device__ __forceinline__ bool valid_init(const double (&s)[4], const Thresholds& th) {
  const double gx[4] = {-2.5, -2.5, 0.0, 0.0}, gy[4] = {-1.0, 1.0, 2.0, -2.0};
  double dx = s[0] - s[2], dy = s[1] - s[3];
  bool ok = !(fma(dy, dy, dx * dx) < th.valid_d2);
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    dx = s[2] - gx[g];
    dy = s[3] - gy[g];
    ok = ok && !(fma(dy, dy, dx * dx) < th.valid_d2);
  }
  return ok;
}

// reset :105-126 (3obj :60-82), state part.  Returns the new goal row.
template <int NOBJ>
__device__ __forceinline__ int reset_env(Env<NOBJ>& e, const earl_tabletop_cfg& cfg, uint64_t counter, int env,
                                         const double* __restrict__ goal_table,
                                         const int32_t* __restrict__ next_goal_idx, const Thresholds& th) {
  e.attached = -1;
  int gi;
  if constexpr (NOBJ == 1) {
    if (cfg.reset_at_goal) {
      gi = sample_goal(cfg, counter, env, next_goal_idx);
      const double* row = goal_table + (size_t)gi * 6;
#pragma unroll
      for (int k = 0; k < 4; ++k) e.q[k] = row[k];
      return gi;
    }
    if (cfg.wide_init) {
      for (uint32_t k = 0;; ++k) {  // rejection sampling, acceptance ~0.5
        const U4 a = draw_block(cfg, counter, env, 1 + 2 * k), b = draw_block(cfg, counter, env, 2 + 2 * k);
        const double s[4] = {-2.5 + 5.0 * u01(a.x, a.y), -2.5 + 5.0 * u01(a.z, a.w), -2.5 + 5.0 * u01(b.x, b.y),
                             -2.5 + 5.0 * u01(b.z, b.w)};
#pragma unroll
        for (int j = 0; j < 4; ++j) e.q[j] = s[j];
        if (valid_init(s, th) || k >= 1023) break;
      }
    } else {
      e.q[0] = 0.0; e.q[1] = 0.0; e.q[2] = 2.5; e.q[3] = 0.0;  // initial_states[0] :11
    }
    gi = sample_goal(cfg, counter, env, next_goal_idx);
  } else {
    if (cfg.reset_at_goal) {  // 3obj reset :64-69: goal + U(-0.3, 0.3)^8 (np.random.uniform = low + (high - low) u)
      gi = sample_goal(cfg, counter, env, nullptr);
      const double* row = goal_table + (size_t)gi * 10;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const U4 b = draw_block(cfg, counter, env, 1 + j);
        e.q[2 * j] = row[2 * j] + (-0.3 + 0.6 * u01(b.x, b.y));
        e.q[2 * j + 1] = row[2 * j + 1] + (-0.3 + 0.6 * u01(b.z, b.w));
      }
      return gi;
    }
    const double init[8] = {0.0, 0.0, 2.5, 0.0, 2.5, -1.0, 2.5, 1.0};  // 3obj initial_states :11
#pragma unroll
    for (int k = 0; k < 8; ++k) e.q[k] = init[k];
    gi = sample_goal(cfg, counter, env, nullptr);
  }
  return gi;
}

}  // namespace earl
