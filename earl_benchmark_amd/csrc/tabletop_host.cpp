// tabletop_host.cpp -- the `_cpu` entry points of include/earl_tabletop.h (BASELINE configs[0]: "1 env, CPU ... plumbing, no GPU"):
// the SAME per-env functions the gfx950 kernels run (tabletop_device.h, tabletop_step.h, philox.h), compiled for the host by g++
// (-DEARL_HOST_BUILD: earl_rt.h -> host_shim.h) with -ffp-contract=off, one OpenMP iteration per env where a kernel has one lane per env.
// Host pointers, no stream, no HIP runtime.  This library is loaded only when a caller ASKS for device='cpu'; nothing falls back to it,
// and nothing here touches oracle/ (the oracle is the checker of both builds).
#include <cstdint>

#include "tabletop_hostside.h"
#include "tabletop_step.h"

#ifdef _OPENMP
#include <omp.h>
#endif

using namespace earl;
using namespace earl::hostside;

namespace {

// envs per OpenMP chunk.  A thread walks its envs one after the other, each through all T steps (the kernels' loop order: state in registers), so the
// rows [t, i] a chunk writes must stay cache-resident across its envs: 16 envs x 200 steps x 66 B = 211 KB
constexpr int kChunk = 16;

template <class F>
inline void for_each_env(int n, F&& f) {
#pragma omp parallel for schedule(static, kChunk) if (n >= 4 * kChunk)
  for (int i = 0; i < n; ++i) f(i);
}

// The fused rollout on the host: a TILE of kTile consecutive envs keeps its state in a small array (the kernels keep it in registers) and walks the T steps time-major,
// so that every step writes kTile consecutive rows of each output -- the kernels' row order, contiguous in memory -- instead of one row per env per step scattered
// T * n * 66 bytes apart (rollout_body's loop order, made for a lane that owns one env).  Same wrapped_step, same counters: the same bits.
constexpr int kTile = 64;
template <int NOBJ, bool GENERAL>
void rollout_tiles(const KArgs& a) {
  const int n = a.cfg.n, tiles = (n + kTile - 1) / kTile;
#pragma omp parallel for schedule(static) if (tiles >= 4)
  for (int tile = 0; tile < tiles; ++tile) {
    const int i0 = tile * kTile, m = n - i0 < kTile ? n - i0 : kTile;
    Lane<NOBJ> L[kTile];
    float g[kTile][Dims<NOBJ>::NG];
    for (int k = 0; k < m; ++k) {
      load_lane<NOBJ>(a, i0 + k, L[k]);
      load_goal<NOBJ>(a.st.goal_table, L[k].goal_idx, g[k]);
    }
    for (int t = 0; t < a.T; ++t) {
      const float* ap = a.act + ((size_t)t * n + i0) * 3;
      const size_t row0 = (size_t)t * n + i0;
      for (int k = 0; k < m; ++k) {
        float o[Dims<NOBJ>::NOBS];
        float reward;
        bool done, succ;
        wrapped_step<NOBJ, GENERAL>(a, i0 + k, a.cfg.counter + (uint64_t)t, L[k], g[k], ap[3 * k], ap[3 * k + 1], ap[3 * k + 2], o, reward, done, succ);
        const size_t row = row0 + k;
        if (a.out.obs) store_obs<NOBJ>(a.out.obs + row * Dims<NOBJ>::NOBS, o);
        if (a.out.reward) a.out.reward[row] = reward;
        if (a.out.done) a.out.done[row] = done;
        if (a.out.success) a.out.success[row] = succ;
      }
    }
    for (int k = 0; k < m; ++k) store_lane<NOBJ>(a, i0 + k, L[k]);
  }
}

template <int NOBJ>
int do_step(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const float* act, const int32_t* ngi, const earl_tabletop_out* out) {
  if (int rc = check_common(cfg, st, NOBJ)) return rc;
  if (!act || !out) return fail(EARL_ERR_ARG, "act/out is NULL");
  if (cfg->n == 0) return EARL_OK;
  const KArgs a{*cfg, *st, *out, act, ngi, nullptr, nullptr, 1, thresholds()};
  if (cfg->goal_change_frequency > 0 || cfg->auto_reset) for_each_env(cfg->n, [&](int i) { step_body<NOBJ, true>(a, i); });
  else for_each_env(cfg->n, [&](int i) { step_body<NOBJ, false>(a, i); });
  return EARL_OK;
}

template <int NOBJ>
int do_reset(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const uint8_t* mask, const int32_t* ngi, float* obs) {
  if (int rc = check_common(cfg, st, NOBJ)) return rc;
  if (cfg->n == 0) return EARL_OK;
  const KArgs a{*cfg, *st, earl_tabletop_out{nullptr, nullptr, nullptr, nullptr, nullptr}, nullptr, ngi, mask, obs, 0, thresholds()};
  for_each_env(cfg->n, [&](int i) { reset_body<NOBJ>(a, i); });
  return EARL_OK;
}

template <int NOBJ>
int do_rollout(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int32_t T, const float* act, const earl_tabletop_out* out, bool reset_first) {
  if (int rc = check_common(cfg, st, NOBJ)) return rc;
  if (!act || !out) return fail(EARL_ERR_ARG, "act/out is NULL");
  if (T < 0) return fail(EARL_ERR_ARG, "T = %d < 0", T);
  if (cfg->n == 0) return EARL_OK;
  earl_tabletop_cfg c2 = *cfg;
  if (reset_first) {                       // reset with counter cfg->counter, the steps with the next ones (earl_tabletop_reset_rollout)
    if (int rc = do_reset<NOBJ>(cfg, st, nullptr, nullptr, nullptr)) return rc;
    c2.counter += 1;
  }
  if (T == 0) return EARL_OK;
  const KArgs a{c2, *st, *out, act, nullptr, nullptr, nullptr, T, thresholds()};
  if (cfg->goal_change_frequency > 0 || cfg->auto_reset) rollout_tiles<NOBJ, true>(a);
  else rollout_tiles<NOBJ, false>(a);
  return EARL_OK;
}

}  // namespace

extern "C" {

int earl_tabletop_step_cpu(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const float* act, const int32_t* next_goal_idx,
                           const earl_tabletop_out* out) {
  return do_step<1>(cfg, st, act, next_goal_idx, out);
}
int earl_tabletop_rollout_cpu(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int32_t T, const float* act, const earl_tabletop_out* out) {
  return do_rollout<1>(cfg, st, T, act, out, false);
}
int earl_tabletop_reset_rollout_cpu(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int32_t T, const float* act, const earl_tabletop_out* out) {
  return do_rollout<1>(cfg, st, T, act, out, true);
}
int earl_tabletop_eval_episodes_cpu(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int32_t episodes, int32_t T, const float* act,
                                    int64_t act_episode_stride, const earl_tabletop_out* out) {
  if (episodes < 0) return fail(EARL_ERR_ARG, "episodes = %d < 0", episodes);
  if (act_episode_stride < 0) return fail(EARL_ERR_ARG, "negative action stride");
  if (int rc = check_common(cfg, st, 1)) return rc;
  if (!act || !out) return fail(EARL_ERR_ARG, "act/out is NULL");
  if (T < 0) return fail(EARL_ERR_ARG, "T = %d < 0", T);
  for (int32_t e = 0; e < episodes; ++e) {
    earl_tabletop_cfg c = *cfg;
    c.counter += (uint64_t)e * (uint64_t)(T + 1);
    const size_t rows = (size_t)e * (size_t)T * (size_t)cfg->n;
    const earl_tabletop_out o{out->obs ? out->obs + rows * 12 : nullptr, out->reward ? out->reward + rows : nullptr, out->done ? out->done + rows : nullptr,
                              out->success ? out->success + rows : nullptr, nullptr};
    if (int rc = do_rollout<1>(&c, st, T, act + (size_t)e * (size_t)act_episode_stride, &o, true)) return rc;
  }
  return EARL_OK;
}
int earl_tabletop_reset_cpu(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const uint8_t* mask, const int32_t* next_goal_idx, float* obs) {
  return do_reset<1>(cfg, st, mask, next_goal_idx, obs);
}
int earl_tabletop_observe_cpu(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const earl_tabletop_out* out) {
  if (int rc = check_common(cfg, st, 1)) return rc;
  if (!out) return fail(EARL_ERR_ARG, "out is NULL");
  const KArgs a{*cfg, *st, *out, nullptr, nullptr, nullptr, nullptr, 0, thresholds()};
  for_each_env(cfg->n, [&](int i) { observe_body(a, i); });
  return EARL_OK;
}
int earl_tabletop_reward_cpu(int32_t n, const float* obs, int32_t reward_type, int32_t wide_init, float* reward, uint8_t* success) {
  if (n < 0 || !obs) return fail(EARL_ERR_ARG, "bad n/obs");
  if (reward_type != EARL_REWARD_SPARSE && reward_type != EARL_REWARD_DENSE) return fail(EARL_ERR_ARG, "reward_type = %d", reward_type);
  const Thresholds th = thresholds();
  for_each_env(n, [&](int i) { reward_body<1>(i, obs, reward_type, wide_init, reward, success, th); });
  return EARL_OK;
}
int earl_tabletop_valid_init_cpu(int32_t n, const double* cand, uint8_t* valid) {
  if (n < 0 || !cand || !valid) return fail(EARL_ERR_ARG, "bad n/cand/valid");
  const Thresholds th = thresholds();
  for_each_env(n, [&](int i) { valid_init_body(i, cand, valid, th); });
  return EARL_OK;
}

int earl_tabletop3_step_cpu(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const float* act, const earl_tabletop_out* out) {
  return do_step<3>(cfg, st, act, nullptr, out);
}
int earl_tabletop3_rollout_cpu(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int32_t T, const float* act, const earl_tabletop_out* out) {
  return do_rollout<3>(cfg, st, T, act, out, false);
}
int earl_tabletop3_reset_cpu(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const uint8_t* mask, float* obs) {
  return do_reset<3>(cfg, st, mask, nullptr, obs);
}
int earl_tabletop3_reward_cpu(int32_t n, const float* obs, int32_t reward_type, float* reward, uint8_t* success) {
  if (n < 0 || !obs) return fail(EARL_ERR_ARG, "bad n/obs");
  if (reward_type != EARL_REWARD_SPARSE && reward_type != EARL_REWARD_DENSE) return fail(EARL_ERR_ARG, "reward_type = %d", reward_type);
  const Thresholds th = thresholds();
  for_each_env(n, [&](int i) { reward_body<3>(i, obs, reward_type, 0, reward, success, th); });
  return EARL_OK;
}

int earl_host_set_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}
const char* earl_host_version(void) { return "earl-host 0.1 (csrc/tabletop_device.h compiled for the host)"; }
const char* earl_host_last_error(void) { return g_err; }

}  // extern "C"
