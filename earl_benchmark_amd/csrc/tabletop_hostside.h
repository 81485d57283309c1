// tabletop_hostside.h -- host-side helpers of the tabletop entry points shared by libearl_hip.so (tabletop.hip) and libearl_host.so
// (tabletop_host.cpp): the exact comparison thresholds (IEEE sqrt on the host), argument validation, the thread-local error message.
#pragma once
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "tabletop_device.h"

namespace earl {
namespace hostside {

inline thread_local char g_err[512] = "";

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}

// smallest double s with sqrt(s) >= c  (so that  sqrt(d2) < c  <=>  d2 < s)
inline double lt_threshold_f64(double c) {
  double s = c * c;
  while (std::sqrt(s) >= c) s = std::nextafter(s, 0.0);
  while (std::sqrt(s) < c) s = std::nextafter(s, INFINITY);
  return s;
}
// largest float s with (double)sqrtf(s) <= c  (so that  (double)sqrtf(x) <= c  <=>  x <= s)
inline float le_threshold_f32(double c) {
  float s = (float)(c * c);
  while ((double)std::sqrt(s) <= c) s = std::nextafterf(s, INFINITY);
  while ((double)std::sqrt(s) > c) s = std::nextafterf(s, 0.0f);
  return s;
}
inline const Thresholds& thresholds() {
  static const Thresholds th = {lt_threshold_f64(0.4), lt_threshold_f64(1.0), le_threshold_f32(0.2), le_threshold_f32(0.4)};
  return th;
}

// smallest float x with rescale_action(x) > 0: the grip test `rescaled a[2] > 0` (:144) on the RAW action.
// rescale is monotone, so a bisection over the (ordered) non-negative float bit patterns finds it exactly.
inline float grip_threshold() {
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

inline int check_common(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int nobj) {
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


}  // namespace hostside
}  // namespace earl
