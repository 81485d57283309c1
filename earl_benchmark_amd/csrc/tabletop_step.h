// tabletop_step.h -- one wrapped env step / reset / observation of ONE env on register state, shared by the gfx950 kernels (tabletop.hip:
// one lane per env) and by the host build of the same functions (tabletop_host.cpp: one OpenMP iteration per env).
// Reference: wrappers/persistent_state_wrapper.py:17-31, wrappers/lifelong_wrapper.py:30-44 around
// envs/tabletop_manipulation.py:128-138 (tabletop_device.h holds move / obs / reward / reset).
#pragma once
#include "tabletop_device.h"

namespace earl {

struct KArgs {
  earl_tabletop_cfg cfg;
  earl_tabletop_state st;
  earl_tabletop_out out;
  const float* act;
  const int32_t* next_goal_idx;
  const uint8_t* mask;
  float* obs_only;
  int32_t T;
  Thresholds th;
};

// ------------------------------------------------------------------------------------------------
// per-lane state I/O
// ------------------------------------------------------------------------------------------------
template <int NOBJ>
struct Lane {
  Env<NOBJ> e;
  int goal_idx;
  int steps;
  int sgc;      // steps since goal change (lifelong)
  double lret;  // lifelong return
  int resets;   // resets performed inside this launch (auto_reset)
};

template <int NOBJ>
__device__ __forceinline__ void load_lane(const KArgs& a, int i, Lane<NOBJ>& L) {
  constexpr int NQ = Dims<NOBJ>::NQ;
  const double2* q2 = reinterpret_cast<const double2*>(a.st.qpos + (size_t)i * NQ);
#pragma unroll
  for (int k = 0; k < NQ / 2; ++k) {
    const double2 v = q2[k];
    L.e.q[2 * k] = v.x;
    L.e.q[2 * k + 1] = v.y;
  }
  L.e.attached = a.st.attached[i];
  L.goal_idx = a.st.goal_idx[i];
  L.steps = a.st.steps_since_reset[i];
  L.resets = 0;
  if (a.cfg.goal_change_frequency > 0) {
    L.sgc = a.st.steps_since_goal_change[i];
    L.lret = a.st.lifelong_return[i];
  } else {
    L.sgc = 0;
    L.lret = 0.0;
  }
}

template <int NOBJ>
__device__ __forceinline__ void store_lane(const KArgs& a, int i, const Lane<NOBJ>& L) {
  constexpr int NQ = Dims<NOBJ>::NQ;
  double2* q2 = reinterpret_cast<double2*>(a.st.qpos + (size_t)i * NQ);
#pragma unroll
  for (int k = 0; k < NQ / 2; ++k) q2[k] = double2{L.e.q[2 * k], L.e.q[2 * k + 1]};
  a.st.attached[i] = (int8_t)L.e.attached;
  a.st.steps_since_reset[i] = L.steps;
  if (a.cfg.goal_change_frequency > 0) {
    a.st.steps_since_goal_change[i] = L.sgc;
    a.st.lifelong_return[i] = L.lret;
    a.st.goal_idx[i] = L.goal_idx;
  }
  if (L.resets) {
    a.st.goal_idx[i] = L.goal_idx;
    a.st.num_interventions[i] += L.resets;
  }
}

template <int NOBJ>
__device__ __forceinline__ void store_obs(float* __restrict__ dst, const float (&o)[Dims<NOBJ>::NOBS]) {
  float4* d4 = reinterpret_cast<float4*>(dst);  // rows are 48 B / 80 B: 16-byte aligned
#pragma unroll
  for (int k = 0; k < Dims<NOBJ>::NOBS / 4; ++k) d4[k] = float4{o[4 * k], o[4 * k + 1], o[4 * k + 2], o[4 * k + 3]};
}

// One wrapped step on register state.  `counter` is the Philox counter of THIS step.
// GENERAL = lifelong goal switching and auto-reset compiled in (they drag the Philox generator into the loop).
template <int NOBJ, bool GENERAL>
__device__ __forceinline__ void wrapped_step(const KArgs& a, int i, uint64_t counter, Lane<NOBJ>& L,
                                             float (&g)[Dims<NOBJ>::NG], float a0, float a1, float a2,
                                             float (&o)[Dims<NOBJ>::NOBS], float& reward, bool& done, bool& succ, double* r64 = nullptr) {
  move<NOBJ>(L.e, rescale_action(a0), rescale_action(a1), rescale_action(a2), a.th);
  make_obs<NOBJ>(L.e, g, o);
  double r;
  reward_success<NOBJ>(o, a.cfg.reward_type, a.cfg.wide_init, a.th, r, succ);
  reward = (float)r;
  if (r64) *r64 = r;
  L.steps += 1;                       // persistent_state_wrapper.py:25-26
  done = L.steps >= a.cfg.horizon;    // :28-29
  if constexpr (NOBJ == 1 && GENERAL) {
    if (a.cfg.goal_change_frequency > 0) {  // lifelong_wrapper.py:30-44
      L.sgc += 1;
      L.lret += r;
      if (L.sgc >= a.cfg.goal_change_frequency) {
        L.sgc = 0;
        L.goal_idx = sample_goal(a.cfg, counter, i, a.next_goal_idx);
        load_goal<NOBJ>(a.st.goal_table, L.goal_idx, g);
#pragma unroll
        for (int k = 0; k < Dims<NOBJ>::NG; ++k) o[Dims<NOBJ>::NQ + 2 + k] = g[k];  // obs re-read with the new goal
      }
    }
  }
  if constexpr (GENERAL)
  if (done && a.cfg.auto_reset) {  // batched-only extension; the outputs above stay the terminal ones
    L.goal_idx = reset_env<NOBJ>(L.e, a.cfg, counter, i, a.st.goal_table, a.next_goal_idx, a.th);
    load_goal<NOBJ>(a.st.goal_table, L.goal_idx, g);
    L.steps = 0;
    L.sgc = 0;
    L.resets += 1;
  }
}


// ------------------------------------------------------------------------------------------------
// per-env bodies (the kernels and the host loops are `for every env i: body(a, i)`)
// ------------------------------------------------------------------------------------------------
template <int NOBJ, bool GENERAL>
__device__ __forceinline__ void step_body(const KArgs& a, int i) {
  Lane<NOBJ> L;
  load_lane<NOBJ>(a, i, L);
  float g[Dims<NOBJ>::NG];
  load_goal<NOBJ>(a.st.goal_table, L.goal_idx, g);
  const float* ap = a.act + (size_t)i * 3;
  float o[Dims<NOBJ>::NOBS];
  float reward;
  bool done, succ;
  double r64;
  const uint64_t counter = a.cfg.counter + (a.st.counter_base ? *a.st.counter_base : 0ull);      // (counter_base: captured step loops, include/earl_tabletop.h)
  wrapped_step<NOBJ, GENERAL>(a, i, counter, L, g, ap[0], ap[1], ap[2], o, reward, done, succ, &r64);
  if (a.out.obs) store_obs<NOBJ>(a.out.obs + (size_t)i * Dims<NOBJ>::NOBS, o);
  if (a.out.reward) a.out.reward[i] = reward;
  if (a.out.reward_f64) a.out.reward_f64[i] = r64;
  if (a.out.done) a.out.done[i] = done;
  if (a.out.success) a.out.success[i] = succ;
  store_lane<NOBJ>(a, i, L);
}

template <int NOBJ, bool GENERAL>
__device__ __forceinline__ void rollout_body(const KArgs& a, int i) {
  const int n = a.cfg.n;
  Lane<NOBJ> L;
  load_lane<NOBJ>(a, i, L);
  float g[Dims<NOBJ>::NG];
  load_goal<NOBJ>(a.st.goal_table, L.goal_idx, g);
  constexpr int PF = 8;  // actions are state-independent: fetch PF steps ahead to keep loads in flight
  for (int t0 = 0; t0 < a.T; t0 += PF) {
    float av[PF][3];
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const int t = t0 + k < a.T ? t0 + k : a.T - 1;
      const float* ap = a.act + ((size_t)t * n + i) * 3;
      av[k][0] = ap[0];
      av[k][1] = ap[1];
      av[k][2] = ap[2];
    }
#pragma unroll
    for (int k = 0; k < PF; ++k) {
      const int t = t0 + k;
      if (t < a.T) {
        float o[Dims<NOBJ>::NOBS];
        float reward;
        bool done, succ;
        wrapped_step<NOBJ, GENERAL>(a, i, a.cfg.counter + (uint64_t)t, L, g, av[k][0], av[k][1], av[k][2], o, reward, done, succ);
        const size_t row = (size_t)t * n + i;
        if (a.out.obs) store_obs<NOBJ>(a.out.obs + row * Dims<NOBJ>::NOBS, o);
        if (a.out.reward) a.out.reward[row] = reward;
        if (a.out.done) a.out.done[row] = done;
        if (a.out.success) a.out.success[row] = succ;
      }
    }
  }
  store_lane<NOBJ>(a, i, L);
}

template <int NOBJ>
__device__ __forceinline__ void reset_body(const KArgs& a, int i) {
  Lane<NOBJ> L;
  load_lane<NOBJ>(a, i, L);
  if (!a.mask || a.mask[i]) {
    L.goal_idx = reset_env<NOBJ>(L.e, a.cfg, a.cfg.counter, i, a.st.goal_table, a.next_goal_idx, a.th);
    L.steps = 0;   // persistent_state_wrapper.py:18-19
    L.sgc = 0;     // lifelong_wrapper.py:26-27
    L.resets = 1;
    store_lane<NOBJ>(a, i, L);
  }
  if (a.obs_only) {
    float g[Dims<NOBJ>::NG], o[Dims<NOBJ>::NOBS];
    load_goal<NOBJ>(a.st.goal_table, L.goal_idx, g);
    make_obs<NOBJ>(L.e, g, o);
    store_obs<NOBJ>(a.obs_only + (size_t)i * Dims<NOBJ>::NOBS, o);
  }
}

__device__ __forceinline__ void observe_body(const KArgs& a, int i) {
  Lane<1> L;
  load_lane<1>(a, i, L);
  float g[6], o[12];
  load_goal<1>(a.st.goal_table, L.goal_idx, g);
  make_obs<1>(L.e, g, o);
  double r;
  bool succ;
  reward_success<1>(o, a.cfg.reward_type, a.cfg.wide_init, a.th, r, succ);
  if (a.out.obs) store_obs<1>(a.out.obs + (size_t)i * 12, o);
  if (a.out.reward) a.out.reward[i] = (float)r;
  if (a.out.success) a.out.success[i] = succ;
  if (a.out.done) a.out.done[i] = L.steps >= a.cfg.horizon;
}

template <int NOBJ>
__device__ __forceinline__ void reward_body(int i, const float* __restrict__ obs, int reward_type, int wide, float* __restrict__ reward,
                                            uint8_t* __restrict__ success, const Thresholds& th) {
  constexpr int NOBS = Dims<NOBJ>::NOBS;
  float o[NOBS];
  const float4* s4 = reinterpret_cast<const float4*>(obs + (size_t)i * NOBS);
#pragma unroll
  for (int k = 0; k < NOBS / 4; ++k) {
    const float4 v = s4[k];
    o[4 * k] = v.x; o[4 * k + 1] = v.y; o[4 * k + 2] = v.z; o[4 * k + 3] = v.w;
  }
  double r;
  bool succ;
  reward_success<NOBJ>(o, reward_type, wide, th, r, succ);
  if (reward) reward[i] = (float)r;
  if (success) success[i] = succ;
}

__device__ __forceinline__ void valid_init_body(int i, const double* __restrict__ cand, uint8_t* __restrict__ valid, const Thresholds& th) {
  const double2* c2 = reinterpret_cast<const double2*>(cand + (size_t)i * 4);
  const double2 u = c2[0], v = c2[1];
  const double s[4] = {u.x, u.y, v.x, v.y};
  valid[i] = valid_init(s, th);
}

}  // namespace earl
