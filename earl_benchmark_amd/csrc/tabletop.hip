// tabletop.hip -- HIP kernels (gfx950) + the C ABI of include/earl_tabletop.h.
//
// Kernels (all one lane per env, 256-thread workgroups = 4 wavefronts of 64 envs):
//   step_kernel<NOBJ>     one wrapped env step: act -> move -> obs -> reward/success -> horizon -> lifelong
//   rollout_kernel<NOBJ>  T steps per launch, state held in VGPRs, only act in / obs,reward,flags out per step
//   reset_kernel<NOBJ>    masked reset + observation of every env
//   observe/reward/valid_init kernels: the pure functions of the reference API
// HBM-bound streaming work: no MFMA, nothing GEMM-shaped.  See DESIGN.md for bytes/env-step and rooflines.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "tabletop_device.h"
#include "tabletop_rollout_ws.h"

using namespace earl;

namespace {

constexpr int kBlock = 256;

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
// kernels
// ------------------------------------------------------------------------------------------------
template <int NOBJ, bool GENERAL>
__global__ __launch_bounds__(kBlock) void step_kernel(const KArgs a) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= a.cfg.n) return;
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
__global__ __launch_bounds__(kBlock) void rollout_kernel(const KArgs a) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const int n = a.cfg.n;
  if (i >= n) return;
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
__global__ __launch_bounds__(kBlock) void reset_kernel(const KArgs a) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= a.cfg.n) return;
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

__global__ __launch_bounds__(kBlock) void observe_kernel(const KArgs a) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= a.cfg.n) return;
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
__global__ __launch_bounds__(kBlock) void reward_kernel(int n, const float* __restrict__ obs, int reward_type, int wide,
                                                        float* __restrict__ reward, uint8_t* __restrict__ success,
                                                        const Thresholds th) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
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

__global__ __launch_bounds__(kBlock) void valid_init_kernel(int n, const double* __restrict__ cand,
                                                            uint8_t* __restrict__ valid, const Thresholds th) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  const double2* c2 = reinterpret_cast<const double2*>(cand + (size_t)i * 4);
  const double2 u = c2[0], v = c2[1];
  const double s[4] = {u.x, u.y, v.x, v.y};
  valid[i] = valid_init(s, th);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
thread_local char g_err[512] = "";
int g_rollout_lds_pad = 0;  // extra dynamic LDS per workgroup of the wave-specialised kernel (limits co-residency; tuning)
int g_rollout_wgs_per_cu = 1;  // (tuning) workgroups per CU the episode groups of a multi-episode launch may fill
int g_rollout_impl = 0;  // 0 = auto (wave-specialised when applicable), 1 = force the plain one-lane-per-env kernel

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

// smallest double s with sqrt(s) >= c  (so that  sqrt(d2) < c  <=>  d2 < s)
double lt_threshold_f64(double c) {
  double s = c * c;
  while (std::sqrt(s) >= c) s = std::nextafter(s, 0.0);
  while (std::sqrt(s) < c) s = std::nextafter(s, INFINITY);
  return s;
}
// largest float s with (double)sqrtf(s) <= c  (so that  (double)sqrtf(x) <= c  <=>  x <= s)
float le_threshold_f32(double c) {
  float s = (float)(c * c);
  while ((double)std::sqrt(s) <= c) s = std::nextafterf(s, INFINITY);
  while ((double)std::sqrt(s) > c) s = std::nextafterf(s, 0.0f);
  return s;
}
const Thresholds& thresholds() {
  static const Thresholds th = {lt_threshold_f64(0.4), lt_threshold_f64(1.0), le_threshold_f32(0.2), le_threshold_f32(0.4)};
  return th;
}

// smallest float x with rescale_action(x) > 0: the grip test `rescaled a[2] > 0` (:144) on the RAW action.
// rescale is monotone, so a bisection over the (ordered) non-negative float bit patterns finds it exactly.
float grip_threshold() {
  static const float thr = [] {
    auto rescaled_positive = [](float x) {
      const double c = x < -1.0f ? -1.0 : (x > 1.0f ? 1.0 : (double)x);
      volatile double v = -0.2 + ((c + 1.) * 0.5) * (0.2 - -0.2);
      return v > 0;
    };
    uint32_t lo = 0u, hi = 0x3f800000u;  // +0.0f (not positive) .. 1.0f (positive)
    while (hi - lo > 1) {
      const uint32_t mid = lo + (hi - lo) / 2;
      float f;
      memcpy(&f, &mid, 4);
      if (rescaled_positive(f)) hi = mid; else lo = mid;
    }
    float f;
    memcpy(&f, &hi, 4);
    return f;
  }();
  return thr;
}

int check_common(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int nobj) {
  if (!cfg || !st) return fail(EARL_ERR_ARG, "cfg/state is NULL");
  if (cfg->n < 0) return fail(EARL_ERR_ARG, "n = %d < 0", cfg->n);
  if (!st->qpos || !st->attached || !st->goal_idx || !st->goal_table || !st->steps_since_reset || !st->num_interventions)
    return fail(EARL_ERR_ARG, "state has a NULL array");
  if (cfg->n_goals < 1 || cfg->n_sample_goals < 1 || cfg->n_sample_goals > cfg->n_goals)
    return fail(EARL_ERR_ARG, "bad goal table sizes n_goals=%d n_sample_goals=%d", cfg->n_goals, cfg->n_sample_goals);
  if (cfg->reward_type != EARL_REWARD_SPARSE && cfg->reward_type != EARL_REWARD_DENSE)
    return fail(EARL_ERR_ARG, "reward_type = %d", cfg->reward_type);
  if (cfg->goal_change_frequency < 0 || cfg->horizon < 0) return fail(EARL_ERR_ARG, "negative horizon/frequency");
  if (cfg->goal_change_frequency > 0 && (!st->steps_since_goal_change || !st->lifelong_return))
    return fail(EARL_ERR_ARG, "lifelong mode needs steps_since_goal_change and lifelong_return");
  if (nobj == 3 && (cfg->wide_init || cfg->goal_change_frequency))
    return fail(EARL_ERR_ARG, "3-object variant: wide_init / lifelong do not exist in the reference class");
  return EARL_OK;
}

int launched(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(EARL_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
  return EARL_OK;
}

// compute units of the current device (MI355X: 256), asked once per device: the launch geometry of the fused rollout (episode groups side by
// side, 16-step chunks while every workgroup has a CU to itself) follows the chip, not a constant
int cu_count() {
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cus[dev] == 0) {
    int v = 0;
    cus[dev] = (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
  }
  return cus[dev];
}

inline dim3 grid_for(int n) { return dim3((unsigned)((n + kBlock - 1) / kBlock)); }

template <int NOBJ>
int do_step(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const float* act, const int32_t* ngi,
            const earl_tabletop_out* out, earl_stream_t stream) {
  if (int rc = check_common(cfg, st, NOBJ)) return rc;
  if (!act || !out) return fail(EARL_ERR_ARG, "act/out is NULL");
  if (cfg->n == 0) return EARL_OK;
  KArgs a{*cfg, *st, *out, act, ngi, nullptr, nullptr, 1, thresholds()};
  if (cfg->goal_change_frequency > 0 || cfg->auto_reset)
    step_kernel<NOBJ, true><<<grid_for(cfg->n), kBlock, 0, (hipStream_t)stream>>>(a);
  else
    step_kernel<NOBJ, false><<<grid_for(cfg->n), kBlock, 0, (hipStream_t)stream>>>(a);
  return launched("step_kernel");
}

template <int NOBJ>
int do_reset(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const uint8_t* mask, const int32_t* ngi,
             float* obs, earl_stream_t stream);

// reset_first: perform the reset of all envs (counter cfg->counter) before the T steps (counters cfg->counter+1 ..)
template <int NOBJ>
int do_rollout(const earl_tabletop_cfg* cfg_in, const earl_tabletop_state* st, int32_t T, const float* act,
               const earl_tabletop_out* out, earl_stream_t stream, bool reset_first = false, int32_t episodes = 1, long long act_ep_stride = 0) {
  const earl_tabletop_cfg* cfg = cfg_in;
  const int32_t Tep = T;                     // steps per episode; the fused kernel walks episodes * Tep steps
  if (episodes > 1) T = episodes * Tep;
  if (int rc = check_common(cfg, st, NOBJ)) return rc;
  if (!act || !out) return fail(EARL_ERR_ARG, "act/out is NULL");
  if (T < 0) return fail(EARL_ERR_ARG, "T = %d < 0", T);
  if (cfg->n == 0) return EARL_OK;
  if (T == 0) return reset_first ? do_reset<NOBJ>(cfg, st, nullptr, nullptr, nullptr, stream) : EARL_OK;
  const bool general = cfg->goal_change_frequency > 0 || cfg->auto_reset;
  if constexpr (NOBJ == 1) {
    // the common case (no lifelong switching, no auto-reset, all outputs requested): wave-specialised kernel
    if (!general && out->obs && out->reward && out->done && out->success && g_rollout_impl != 1) {
      WsArgs w{cfg->n, T, cfg->horizon, cfg->wide_init, act, st->qpos, st->attached, st->goal_idx, st->goal_table,
               st->steps_since_reset, out->obs, out->reward, out->done, out->success, thresholds(), grip_threshold(),
               reset_first ? 1 : 0, *cfg, st->goal_idx, st->num_interventions, episodes, Tep, act_ep_stride, 0, 1, episodes};
      dim3 grid((unsigned)((cfg->n + 63) / 64));
      w.wgs = (int)grid.x;
      // Evaluation episodes are independent of one another (each starts with reset()): when the env batch does not fill the chip -- 4096 envs are
      // 64 workgroups on 256 CUs -- several of them run side by side, each group of episodes on its own workgroups (WsArgs::ep_groups).  Same
      // outputs, same final state as the sequence.  impl 38 forces the sequence (one group) for comparison.
      const unsigned cus = (unsigned)cu_count();
      if (episodes > 1 && reset_first && g_rollout_impl != 38 && grid.x * 2 <= cus * (unsigned)g_rollout_wgs_per_cu) {
        const int slots = (int)(cus * (unsigned)g_rollout_wgs_per_cu / grid.x);
        const int P = slots < episodes ? slots : episodes;     // (eight groups, two workgroups per CU: no faster -- HBM-bound)
        w.ep_per_group = (episodes + P - 1) / P;
        w.ep_groups = (episodes + w.ep_per_group - 1) / w.ep_per_group;
        grid.x *= (unsigned)w.ep_groups;
      }
      const hipStream_t hs = (hipStream_t)stream;
#define EARL_WS(RT, NC, NL, NS, K, LEAD) \
  rollout_ws_kernel<RT, NC, NL, NS, K, LEAD><<<grid, 64 * (((NC) == 3 ? 2 : (NC)) + NL + NS), g_rollout_lds_pad, hs>>>(w)
#define EARL_WSM(RT, NC, NL, NS, K, LEAD) \
  rollout_ws_kernel<RT, NC, NL, NS, K, LEAD, false, true><<<grid, 64 * (((NC) == 3 ? 2 : (NC)) + NL + NS), g_rollout_lds_pad, hs>>>(w)
      if (cfg->reward_type == EARL_REWARD_SPARSE) {
        switch (g_rollout_impl) {   // tuning variants (tools/archive/tune_rollout.py); 0 = the shipped configuration
          case 2: EARL_WS(EARL_REWARD_SPARSE, 1, 2, 4, 4, 6); break;
          case 3: EARL_WS(EARL_REWARD_SPARSE, 1, 2, 4, 8, 3); break;
          case 4: EARL_WS(EARL_REWARD_SPARSE, 2, 2, 4, 4, 6); break;
          case 5: EARL_WS(EARL_REWARD_SPARSE, 2, 2, 4, 8, 3); break;
          case 6: EARL_WS(EARL_REWARD_SPARSE, 2, 4, 8, 8, 3); break;
          case 7: EARL_WS(EARL_REWARD_SPARSE, 2, 2, 8, 8, 3); break;
          case 10: EARL_WS(EARL_REWARD_SPARSE, 2, 2, 4, 8, 3); break;
          case 11: EARL_WS(EARL_REWARD_SPARSE, 2, 4, 4, 8, 3); break;
          case 12: EARL_WS(EARL_REWARD_SPARSE, 2, 4, 4, 4, 6); break;
          case 13: EARL_WS(EARL_REWARD_SPARSE, 1, 4, 4, 8, 3); break;
          case 14: EARL_WS(EARL_REWARD_SPARSE, 2, 2, 2, 4, 6); break;
          case 8: EARL_WS(EARL_REWARD_SPARSE, 1, 1, 1, 4, 6); break;
          case 20: EARL_WS(EARL_REWARD_SPARSE, 3, 4, 4, 8, 3); break;    // x / y in adjacent lanes (DPP), VGPR-only masks
          case 21: EARL_WS(EARL_REWARD_SPARSE, 3, 4, 8, 8, 3); break;
          case 22: EARL_WS(EARL_REWARD_SPARSE, 3, 2, 8, 8, 3); break;    // = shipped for grids of up to 256 workgroups
          case 23: EARL_WS(EARL_REWARD_SPARSE, 2, 4, 8, 8, 3); break;
          case 30: EARL_WS(EARL_REWARD_SPARSE, 3, 1, 8, 8, 3); break;
          case 31: EARL_WS(EARL_REWARD_SPARSE, 3, 2, 4, 8, 3); break;
          case 32: EARL_WS(EARL_REWARD_SPARSE, 3, 2, 4, 4, 6); break;
          case 33: EARL_WS(EARL_REWARD_SPARSE, 3, 2, 8, 8, 2); break;    // = shipped for larger grids
          case 34: EARL_WS(EARL_REWARD_SPARSE, 3, 2, 8, 8, 4); break;
          case 29:                                                                                                     // stamps
            if (episodes > 1 && Tep >= 32) rollout_ws_kernel<EARL_REWARD_SPARSE, 3, 2, 8, 16, 2, true, true><<<grid, 64 * 12, 0, hs>>>(w);
            else if (episodes > 1) rollout_ws_kernel<EARL_REWARD_SPARSE, 3, 2, 8, 8, 3, true, true><<<grid, 64 * 12, 0, hs>>>(w);
            else rollout_ws_kernel<EARL_REWARD_SPARSE, 3, 2, 8, 8, 3, true><<<grid, 64 * 12, 0, hs>>>(w);
            break;
          case 9: rollout_ws_kernel<EARL_REWARD_SPARSE, 2, 4, 4, 8, 3, true><<<grid, 64 * 10, 0, hs>>>(w); break;  // stamps
          case 19: rollout_ws_kernel<EARL_REWARD_SPARSE, 1, 4, 4, 8, 3, true><<<grid, 64 * 9, 0, hs>>>(w); break;  // stamps
#define EARL_WSX(K, LEAD, NT) rollout_ws_kernel<EARL_REWARD_SPARSE, 3, 2, 8, K, LEAD, false, true, NT><<<grid, 64 * 12, g_rollout_lds_pad, hs>>>(w)
          // experiments on the multi-episode launch with per-episode actions (tools/archive/own_actions_experiment.py): prefetch depth, chunk length, nt loads
          case 40: if (episodes > 1) { EARL_WSX(16, 3, false); break; } [[fallthrough]];
          case 41: if (episodes > 1) { EARL_WSX(8, 3, false); break; } [[fallthrough]];
          case 42: if (episodes > 1) { EARL_WSX(8, 4, false); break; } [[fallthrough]];
          case 43: if (episodes > 1) { EARL_WSX(8, 6, false); break; } [[fallthrough]];
          case 44: if (episodes > 1) { EARL_WSX(16, 2, true); break; } [[fallthrough]];
          case 45: if (episodes > 1) { EARL_WSX(8, 3, true); break; } [[fallthrough]];
          case 46: if (episodes > 1) { EARL_WSX(8, 2, false); break; } [[fallthrough]];
#undef EARL_WSX
          case 36:
          case 38:
          default:
            // 2 compute waves with x / y in adjacent lanes (DPP, VGPR-only masks) + 2 loaders + 8 storers, 8-step chunks:
            // fastest of the variants above at N = 4096 and not slower at any larger N measured (tools/archive/tune_rollout.py;
            // profiles/r01_tune_rollout.txt).  Variant 11 is the previous default (lane-half layout, 4 storers).
            // Two loaders, not four: fewer waves compete with the compute waves for issue slots (4 loaders: 30.4 us, 2: 29.0 us
            // at N = 4096, T = 200).  Large grids (more than one workgroup per CU) prefer shorter loader trips (LEAD 2):
            // 69.5 vs 65.0 G env-steps/s at N = 2^20.
            if (episodes > 1) {          // several evaluation episodes per launch: the MULTI instantiation of the shipped configuration
              // 16-step chunks (two 8-step granules; an episode may end between them) when every workgroup has a CU to itself: the per-chunk costs
              // -- barrier, action fetch, loop -- are paid half as often (105 -> 99 ns per step at N = 4096); 152 KB of LDS, so larger grids keep
              // the 8-step chunks (77 KB, two workgroups per CU).  impl 36 forces the 8-step form for comparison.
              // ... but only while ONE episode is in flight per env (latency-bound).  With several episode groups side by side the launch is
              // HBM-bound and the 8-step chunks win (own actions per episode: 280-290 against 245-260 us per 28-episode launch).
              if (grid.x <= cus && Tep >= 32 && g_rollout_impl != 36 && w.ep_groups == 1) EARL_WSM(EARL_REWARD_SPARSE, 3, 2, 8, 16, 2);
              else if (grid.x <= cus) EARL_WSM(EARL_REWARD_SPARSE, 3, 2, 8, 8, 3);
              else EARL_WSM(EARL_REWARD_SPARSE, 3, 2, 8, 8, 2);
            } else if (grid.x <= cus) EARL_WS(EARL_REWARD_SPARSE, 3, 2, 8, 8, 3);     // (one episode per launch: 16-step chunks lengthen the pipeline's fill by
                                                                                      // as much as they save over 200 steps: 27.9 against 27.3 us)
            else EARL_WS(EARL_REWARD_SPARSE, 3, 2, 8, 8, 2);
            break;
        }
      } else {
        if (episodes > 1) {
          if (grid.x <= cus && Tep >= 32 && w.ep_groups == 1) EARL_WSM(EARL_REWARD_DENSE, 3, 2, 8, 16, 2);
          else if (grid.x <= cus) EARL_WSM(EARL_REWARD_DENSE, 3, 2, 8, 8, 3);
          else EARL_WSM(EARL_REWARD_DENSE, 3, 2, 8, 8, 2);
        } else if (grid.x <= cus) EARL_WS(EARL_REWARD_DENSE, 3, 2, 8, 8, 3);
        else EARL_WS(EARL_REWARD_DENSE, 3, 2, 8, 8, 2);
      }
#undef EARL_WS
#undef EARL_WSM
      return launched("rollout_ws_kernel");
    }
  }
  earl_tabletop_cfg c2 = *cfg;
  if (reset_first) {   // no fused variant of the general kernel: reset launch, then the steps with the next counters
    if (int rc = do_reset<NOBJ>(cfg, st, nullptr, nullptr, nullptr, stream)) return rc;
    c2.counter += 1;
  }
  KArgs a{c2, *st, *out, act, nullptr, nullptr, nullptr, T, thresholds()};
  if (general)
    rollout_kernel<NOBJ, true><<<grid_for(cfg->n), kBlock, 0, (hipStream_t)stream>>>(a);
  else
    rollout_kernel<NOBJ, false><<<grid_for(cfg->n), kBlock, 0, (hipStream_t)stream>>>(a);
  return launched("rollout_kernel");
}

template <int NOBJ>
int do_reset(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const uint8_t* mask, const int32_t* ngi,
             float* obs, earl_stream_t stream) {
  if (int rc = check_common(cfg, st, NOBJ)) return rc;
  if (cfg->n == 0) return EARL_OK;
  KArgs a{*cfg, *st, earl_tabletop_out{nullptr, nullptr, nullptr, nullptr, nullptr}, nullptr, ngi, mask, obs, 0, thresholds()};
  reset_kernel<NOBJ><<<grid_for(cfg->n), kBlock, 0, (hipStream_t)stream>>>(a);
  return launched("reset_kernel");
}

}  // namespace

extern "C" {

int earl_tabletop_step(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const float* act,
                       const int32_t* next_goal_idx, const earl_tabletop_out* out, earl_stream_t stream) {
  return do_step<1>(cfg, st, act, next_goal_idx, out, stream);
}
int earl_tabletop_rollout(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int32_t T, const float* act,
                          const earl_tabletop_out* out, earl_stream_t stream) {
  return do_rollout<1>(cfg, st, T, act, out, stream);
}
int earl_tabletop_reset(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const uint8_t* mask,
                        const int32_t* next_goal_idx, float* obs, earl_stream_t stream) {
  return do_reset<1>(cfg, st, mask, next_goal_idx, obs, stream);
}
int earl_tabletop_reset_rollout(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int32_t T, const float* act,
                                const earl_tabletop_out* out, earl_stream_t stream) {
  return do_rollout<1>(cfg, st, T, act, out, stream, true);
}
int earl_tabletop_eval_episodes(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int32_t episodes, int32_t T, const float* act,
                                int64_t act_episode_stride, const earl_tabletop_out* out, earl_stream_t stream) {
  if (episodes < 0) return fail(EARL_ERR_ARG, "episodes = %d < 0", episodes);
  if (act_episode_stride < 0) return fail(EARL_ERR_ARG, "negative action stride");
  if (int rc = check_common(cfg, st, 1)) return rc;
  if (!act || !out) return fail(EARL_ERR_ARG, "act/out is NULL");
  if (T < 0) return fail(EARL_ERR_ARG, "T = %d < 0", T);
  if (episodes == 0 || cfg->n == 0) return EARL_OK;
  const bool general = cfg->goal_change_frequency > 0 || cfg->auto_reset;
  // one launch walks all episodes when the wave-specialised kernel applies and episodes end on its chunk boundaries (8 steps)
  const bool fused = episodes > 1 && !general && out->obs && out->reward && out->done && out->success && (g_rollout_impl == 0 || g_rollout_impl == 29 || g_rollout_impl == 36 || g_rollout_impl == 38 || (g_rollout_impl >= 40 && g_rollout_impl <= 46)) &&
                     T % 8 == 0 && T >= 16 && (long long)episodes * T < (1 << 24);
  if (fused || episodes == 1) return do_rollout<1>(cfg, st, T, act, out, stream, true, episodes, (long long)act_episode_stride);
  for (int32_t e = 0; e < episodes; ++e) {          // otherwise: the same thing as `episodes` launches
    earl_tabletop_cfg c = *cfg;
    c.counter += (uint64_t)e * (uint64_t)(T + 1);
    const size_t rows = (size_t)e * (size_t)T * (size_t)cfg->n;
    earl_tabletop_out o{out->obs ? out->obs + rows * 12 : nullptr, out->reward ? out->reward + rows : nullptr, out->done ? out->done + rows : nullptr,
                        out->success ? out->success + rows : nullptr, nullptr};
    if (int rc = do_rollout<1>(&c, st, T, act + (size_t)e * (size_t)act_episode_stride, &o, stream, true)) return rc;
  }
  return EARL_OK;
}
int earl_tabletop_observe(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const earl_tabletop_out* out,
                          earl_stream_t stream) {
  if (int rc = check_common(cfg, st, 1)) return rc;
  if (!out) return fail(EARL_ERR_ARG, "out is NULL");
  if (cfg->n == 0) return EARL_OK;
  KArgs a{*cfg, *st, *out, nullptr, nullptr, nullptr, nullptr, 0, thresholds()};
  observe_kernel<<<grid_for(cfg->n), kBlock, 0, (hipStream_t)stream>>>(a);
  return launched("observe_kernel");
}
int earl_tabletop_reward(int32_t n, const float* obs, int32_t reward_type, int32_t wide_init, float* reward,
                         uint8_t* success, earl_stream_t stream) {
  if (n < 0 || !obs) return fail(EARL_ERR_ARG, "bad n/obs");
  if (reward_type != EARL_REWARD_SPARSE && reward_type != EARL_REWARD_DENSE) return fail(EARL_ERR_ARG, "reward_type = %d", reward_type);
  if (n == 0) return EARL_OK;
  reward_kernel<1><<<grid_for(n), kBlock, 0, (hipStream_t)stream>>>(n, obs, reward_type, wide_init, reward, success, thresholds());
  return launched("reward_kernel");
}
int earl_tabletop_valid_init(int32_t n, const double* cand, uint8_t* valid, earl_stream_t stream) {
  if (n < 0 || !cand || !valid) return fail(EARL_ERR_ARG, "bad n/cand/valid");
  if (n == 0) return EARL_OK;
  valid_init_kernel<<<grid_for(n), kBlock, 0, (hipStream_t)stream>>>(n, cand, valid, thresholds());
  return launched("valid_init_kernel");
}

int earl_tabletop3_step(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const float* act,
                        const earl_tabletop_out* out, earl_stream_t stream) {
  return do_step<3>(cfg, st, act, nullptr, out, stream);
}
int earl_tabletop3_rollout(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int32_t T, const float* act,
                           const earl_tabletop_out* out, earl_stream_t stream) {
  return do_rollout<3>(cfg, st, T, act, out, stream);
}
int earl_tabletop3_reset(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const uint8_t* mask, float* obs,
                         earl_stream_t stream) {
  return do_reset<3>(cfg, st, mask, nullptr, obs, stream);
}
int earl_tabletop3_reward(int32_t n, const float* obs, int32_t reward_type, float* reward, uint8_t* success,
                          earl_stream_t stream) {
  if (n < 0 || !obs) return fail(EARL_ERR_ARG, "bad n/obs");
  if (reward_type != EARL_REWARD_SPARSE && reward_type != EARL_REWARD_DENSE) return fail(EARL_ERR_ARG, "reward_type = %d", reward_type);
  if (n == 0) return EARL_OK;
  reward_kernel<3><<<grid_for(n), kBlock, 0, (hipStream_t)stream>>>(n, obs, reward_type, 0, reward, success, thresholds());
  return launched("reward_kernel3");
}

/* test/bench hook: choose the rollout kernel (0 auto, 1 plain); returns the previous value */
int earl_debug_set_rollout_impl(int impl) {
  const int prev = g_rollout_impl + 1000 * (g_rollout_lds_pad / 1024);
  g_rollout_lds_pad = (impl / 1000) * 1024;        // thousands digit and up: KiB of LDS padding per workgroup
  g_rollout_impl = impl % 1000;
  return prev;
}

/* tuning hook: workgroups per CU that the episode groups of earl_tabletop_eval_episodes may occupy (1 = shipped); returns the previous value */
int earl_debug_set_rollout_wgs_per_cu(int k) {
  const int prev = g_rollout_wgs_per_cu;
  if (k >= 1 && k <= 4) g_rollout_wgs_per_cu = k;
  return prev;
}

/* diagnostic: copy the per-workgroup s_memtime sums of the PROF rollout variant (impl 9) to host memory */
int earl_debug_read_ws_profile(uint64_t* out, int32_t n_words) {
  if (!out || n_words < 0 || n_words > 64 * 16) return fail(EARL_ERR_ARG, "bad profile buffer");
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(earl::g_ws_prof), (size_t)n_words * 8) != hipSuccess)
    return fail(EARL_ERR_LAUNCH, "hipMemcpyFromSymbol failed");
  return EARL_OK;
}

const char* earl_version(void) { return "earl-hip 0.1 (gfx950)"; }
const char* earl_last_error(void) { return g_err; }
int earl_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

}  // extern "C"
