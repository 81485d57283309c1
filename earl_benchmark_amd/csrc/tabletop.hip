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
#include "tabletop_hostside.h"
#include "tabletop_rollout_ws.h"
#include "tabletop_step.h"

using namespace earl;
using namespace earl::hostside;

namespace {

constexpr int kBlock = 256;

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
template <int NOBJ, bool GENERAL>
__global__ __launch_bounds__(kBlock) void step_kernel(const KArgs a) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < a.cfg.n) step_body<NOBJ, GENERAL>(a, i);
}

template <int NOBJ, bool GENERAL>
__global__ __launch_bounds__(kBlock) void rollout_kernel(const KArgs a) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < a.cfg.n) rollout_body<NOBJ, GENERAL>(a, i);
}

template <int NOBJ>
__global__ __launch_bounds__(kBlock) void reset_kernel(const KArgs a) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < a.cfg.n) reset_body<NOBJ>(a, i);
}

__global__ __launch_bounds__(kBlock) void observe_kernel(const KArgs a) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < a.cfg.n) observe_body(a, i);
}

template <int NOBJ>
__global__ __launch_bounds__(kBlock) void reward_kernel(int n, const float* __restrict__ obs, int reward_type, int wide,
                                                        float* __restrict__ reward, uint8_t* __restrict__ success,
                                                        const Thresholds th) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) reward_body<NOBJ>(i, obs, reward_type, wide, reward, success, th);
}

__global__ __launch_bounds__(kBlock) void valid_init_kernel(int n, const double* __restrict__ cand,
                                                            uint8_t* __restrict__ valid, const Thresholds th) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < n) valid_init_body(i, cand, valid, th);
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
int g_rollout_lds_pad = 0;  // extra dynamic LDS per workgroup of the wave-specialised kernel (limits co-residency; tuning)
int g_rollout_wgs_per_cu = 1;  // (tuning) workgroups per CU the episode groups of a multi-episode launch may fill
int g_rollout_impl = 0;  // 0 = auto (wave-specialised when applicable), 1 = force the plain one-lane-per-env kernel

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
          // variants the parity tests cross-check against the shipped one (tests/test_tabletop_gpu.py): other lane layouts of the same arithmetic
          case 11: EARL_WS(EARL_REWARD_SPARSE, 2, 4, 4, 8, 3); break;    // x / y in the two lane halves (v_permlane32_swap); round 1's default
          case 13: EARL_WS(EARL_REWARD_SPARSE, 1, 4, 4, 8, 3); break;    // one lane per env, VGPR-only masks
          case 20: EARL_WS(EARL_REWARD_SPARSE, 3, 4, 4, 8, 3); break;    // x / y in adjacent lanes (DPP), VGPR-only masks, other role counts
          case 22: EARL_WS(EARL_REWARD_SPARSE, 3, 2, 8, 8, 3); break;    // = shipped for grids of up to 256 workgroups
#ifdef EARL_WS_EXPERIMENTS   // tuning variants and the cycle-stamped builds: only in tools/build_ws_variant.sh's libraries (tools/ubench), not in libearl_hip.so
          case 2: EARL_WS(EARL_REWARD_SPARSE, 1, 2, 4, 4, 6); break;
          case 3: EARL_WS(EARL_REWARD_SPARSE, 1, 2, 4, 8, 3); break;
          case 4: EARL_WS(EARL_REWARD_SPARSE, 2, 2, 4, 4, 6); break;
          case 5: EARL_WS(EARL_REWARD_SPARSE, 2, 2, 4, 8, 3); break;
          case 6: EARL_WS(EARL_REWARD_SPARSE, 2, 4, 8, 8, 3); break;
          case 7: EARL_WS(EARL_REWARD_SPARSE, 2, 2, 8, 8, 3); break;
          case 10: EARL_WS(EARL_REWARD_SPARSE, 2, 2, 4, 8, 3); break;
          case 12: EARL_WS(EARL_REWARD_SPARSE, 2, 4, 4, 4, 6); break;
          case 14: EARL_WS(EARL_REWARD_SPARSE, 2, 2, 2, 4, 6); break;
          case 8: EARL_WS(EARL_REWARD_SPARSE, 1, 1, 1, 4, 6); break;
          case 21: EARL_WS(EARL_REWARD_SPARSE, 3, 4, 8, 8, 3); break;
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
#endif
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
