/* tabletop_oracle.c -- CPU restatement of the reference's tabletop step()/reset() path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may build, load or call this file.  The product (earl_benchmark_amd) never
 * links or imports it and has no CPU fallback.
 *
 * What it restates (paths under /root/reference/earl_benchmark/):
 *   envs/tabletop_manipulation.py       step :128-138, move :140-174, _get_obs :55-60,
 *                                       compute_reward :176-191, is_successful :197-204,
 *                                       reset :105-126, is_valid_init :89-97, get_next_goal :62-76
 *   envs/tabletop_manipulation_3obj.py  step :84-94, move :96-134, compute_reward :136-151,
 *                                       is_successful :153-159, reset :60-82
 *   wrappers/persistent_state_wrapper.py reset :17-20, step :22-31
 *   wrappers/lifelong_wrapper.py         step :30-44
 *
 * Pinned (tests/test_oracle.py) against golden vectors recorded from the reference's own classes
 * (tests/golden/make_golden.py) and against the 2,534 demonstration transitions the reference ships.
 *
 * Arithmetic notes (each one probed on numpy 2.2.6 / OpenBLAS 0.3.29 in the build container):
 *   - np.linalg.norm(x) = sqrt(x.dot(x)).  For float64 x the BLAS ddot tail loop is FMA-contracted:
 *     dot = fma(x1, x1, x0*x0) (200000/200000 random pairs agree; the un-fused form fails on 7.9 %).
 *   - For float32 x, sdot accumulates the float-rounded products in a double and rounds the sum to
 *     float once (200000/200000 agree; a pure-f32 or exact-product sum does not).
 *   - The reference pins numpy==1.22.2, where `np.float32 <= 0.2` and `np.float32 * 2.0` promote to
 *     float64.  We follow 1.22: success = (double)norm_f32 <= 0.2, dense reward evaluated in double on the
 *     f32-rounded norms.  (numpy 2 differs only when norm_f32 == float32(0.2) exactly, and in the
 *     last bits of the dense reward; the golden generator lists such rows: none.)
 *   - Build with -ffp-contract=off: `lb + (a + 1.) * 0.5 * (ub - lb)` and `fist + a` are separately
 *     rounded in the reference.  The only FMAs are the explicit fma() calls.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#include "../include/earl_tabletop.h"

/* ------------------------------------------------------------------ Philox4x32-10 (Salmon et al. 2011) */
static void philox4x32_10(const uint32_t ctr_in[4], const uint32_t key_in[2], uint32_t out[4]) {
  uint32_t c0 = ctr_in[0], c1 = ctr_in[1], c2 = ctr_in[2], c3 = ctr_in[3];
  uint32_t k0 = key_in[0], k1 = key_in[1];
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void oracle_philox4x32_10(const uint32_t* ctr, const uint32_t* key, uint32_t* out) { philox4x32_10(ctr, key, out); }

/* The draw layout shared with the HIP kernels (include/earl_tabletop.h: seed = key, counter):
 *   block(draw) = philox(ctr = {draw, global_env_id, counter_lo, counter_hi}, key = {seed_lo, seed_hi})
 *   draw 0            -> goal sample: row = (block[0] * n_sample_goals) >> 32
 *   draws 1+2k, 2+2k  -> k-th wide-init candidate: 4 doubles u = (hi32:lo32 >> 11) * 2^-53, x = -2.5 + 5 u
 *                        (numpy's Generator/RandomState.uniform: low + (high - low) * next_double) */
static void draw_block(const earl_tabletop_cfg* cfg, int32_t env, uint32_t draw, uint32_t out[4]) {
  uint32_t ctr[4] = {draw, (uint32_t)(cfg->env_offset + env), (uint32_t)cfg->counter, (uint32_t)(cfg->counter >> 32)};
  uint32_t key[2] = {(uint32_t)cfg->seed, (uint32_t)(cfg->seed >> 32)};
  philox4x32_10(ctr, key, out);
}
static double u01(uint32_t lo, uint32_t hi) {
  return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0);
}

/* ------------------------------------------------------------------ numpy norms */
static double norm2_f64(double x0, double x1) { return sqrt(fma(x1, x1, x0 * x0)); }
static float norm_f32(const float* x, int n) {
  double dot = 0.0;
  for (int i = 0; i < n; ++i) dot += (double)(x[i] * x[i]); /* product rounded to float first */
  return sqrtf((float)dot);
}
static double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

/* goal_states of the module (tabletop_manipulation.py:12-16), used by is_valid_init with the MODULE table */
static const double kGoalStatesXY[4][2] = {{-2.5, -1.0}, {-2.5, 1.0}, {0.0, 2.0}, {0.0, -2.0}};

/* ------------------------------------------------------------------ tabletop_manipulation.py */
/* step :128-132 -- clip to [-1,1] (np.clip promotes the f32 action to f64), rescale to [-0.2,0.2] */
static double rescale_action(float a) {
  double x = clipd((double)a, -1.0, 1.0);
  return -0.2 + ((x + 1.) * 0.5) * (0.2 - -0.2);
}

/* move :140-174 for NOBJ objects (1 for the loader's env, 3 for the _3obj variant :96-134) */
static void move(double* qpos, int8_t* attached, const double a[3], int nobj) {
  const double fx = qpos[0], fy = qpos[1];
  if (a[2] > 0) {
    if (*attached < 0) {
      double best = INFINITY; /* closest object inside the threshold wins :146-152 */
      for (int k = 0; k < nobj; ++k) {
        double dist = norm2_f64(fx - qpos[2 + 2 * k], fy - qpos[3 + 2 * k]);
        if (dist < 0.4 && dist < best) { *attached = (int8_t)k; best = dist; }
      }
    }
  } else {
    *attached = -1; /* :153-154 */
  }
  double nfx = clipd(fx + a[0], -2.8, 2.8), nfy = clipd(fy + a[1], -2.8, 2.8); /* :156-157 */
  if (*attached >= 0) { /* :158-163: the object moves by the CLIPPED gripper delta, then is clipped */
    int k = *attached;
    qpos[2 + 2 * k] = clipd(qpos[2 + 2 * k] + (nfx - fx), -2.8, 2.8);
    qpos[3 + 2 * k] = clipd(qpos[3 + 2 * k] + (nfy - fy), -2.8, 2.8);
  }
  qpos[0] = nfx; qpos[1] = nfy;
}

/* _get_obs :55-60 -> obs[2*(1+nobj) + 2 + goal_dim], goal_dim = 2*(1+nobj)+2 */
static void get_obs(const double* qpos, int8_t attached, const double* goal, int nobj, float* obs) {
  int nq = 2 + 2 * nobj;
  for (int i = 0; i < nq; ++i) obs[i] = (float)qpos[i];
  /* attached_object tuple: (-1,-1) free; 1-object env (0,0); 3obj keys (0,0),(0.5,0.5),(1,1) (3obj :30-34) */
  float flag = attached < 0 ? -1.0f : 0.5f * (float)attached;
  obs[nq] = flag; obs[nq + 1] = flag;
  for (int i = 0; i < nq + 2; ++i) obs[nq + 2 + i] = (float)goal[i];
}

/* is_successful :197-204 (numpy 1.22 promotion: f32 norm compared in double) */
static int is_successful(const float* obs, int wide) {
  float d[4];
  if (wide) {
    d[0] = obs[2] - obs[8]; d[1] = obs[3] - obs[9];
    return (double)norm_f32(d, 2) <= 0.2;
  }
  for (int i = 0; i < 4; ++i) d[i] = obs[i] - obs[6 + i];
  return (double)norm_f32(d, 4) <= 0.2;
}

/* compute_reward :176-191 */
static double compute_reward(const float* obs, int reward_type, int wide) {
  if (reward_type == EARL_REWARD_SPARSE) return is_successful(obs, wide) ? 1.0 : 0.0;
  float d[2] = {obs[2] - obs[8], obs[3] - obs[9]};
  float n1 = norm_f32(d, 2);
  double reward = (double)(-n1);
  float n1sq = n1 * n1;                           /* np.float32 ** 2 stays float32 */
  reward += 2. * exp((double)(-n1sq) / 0.01);     /* float32 / python float -> float64 (numpy 1.22) */
  float e[2] = {obs[0] - obs[2], obs[1] - obs[3]};
  double grip_to_object = 0.5 * (double)norm_f32(e, 2);
  reward += -grip_to_object;
  reward += 0.5 * exp(-(grip_to_object * grip_to_object) / 0.01);
  return reward;
}

/* 3obj is_successful :153-159 and compute_reward :136-151 */
static int is_successful3(const float* obs) {
  float d[8];
  for (int i = 0; i < 8; ++i) d[i] = obs[i] - obs[10 + i];
  return (double)norm_f32(d, 8) <= 0.4;
}
static double compute_reward3(const float* obs, int reward_type) {
  if (reward_type == EARL_REWARD_SPARSE) return is_successful3(obs) ? 1.0 : 0.0;
  float d[6];
  for (int i = 0; i < 6; ++i) d[i] = obs[2 + i] - obs[12 + i];
  double reward = (double)(-norm_f32(d, 6));
  for (int k = 1; k < 4; ++k) {
    float e[2] = {obs[2 * k] - obs[2 * k + 10], obs[2 * k + 1] - obs[2 * k + 11]};
    float n = norm_f32(e, 2);
    float nsq = n * n;
    reward += 2. * exp((double)(-nsq) / 0.01);
  }
  return reward;
}

/* is_valid_init :89-97 */
static int is_valid_init(const double* s) {
  if (norm2_f64(s[0] - s[2], s[1] - s[3]) < 1) return 0;
  for (int g = 0; g < 4; ++g)
    if (norm2_f64(s[2] - kGoalStatesXY[g][0], s[3] - kGoalStatesXY[g][1]) < 1) return 0;
  return 1;
}

static int sample_goal(const earl_tabletop_cfg* cfg, int32_t env, const int32_t* next_goal_idx) {
  if (next_goal_idx) return next_goal_idx[env];
  uint32_t b[4];
  draw_block(cfg, env, 0, b);
  return (int)(((uint64_t)b[0] * (uint32_t)cfg->n_sample_goals) >> 32);
}

/* reset :105-126 (state part; the caller emits obs) */
static void reset_env(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int32_t i, const int32_t* next_goal_idx) {
  double* q = st->qpos + 4 * (size_t)i;
  st->attached[i] = -1;
  if (cfg->reset_at_goal) {
    st->goal_idx[i] = sample_goal(cfg, i, next_goal_idx);
    const double* g = st->goal_table + 6 * (size_t)st->goal_idx[i];
    for (int k = 0; k < 4; ++k) q[k] = g[k];
  } else {
    if (cfg->wide_init) {
      for (uint32_t k = 0;; ++k) {
        uint32_t a[4], b[4];
        draw_block(cfg, i, 1 + 2 * k, a);
        draw_block(cfg, i, 2 + 2 * k, b);
        q[0] = -2.5 + 5.0 * u01(a[0], a[1]); q[1] = -2.5 + 5.0 * u01(a[2], a[3]);
        q[2] = -2.5 + 5.0 * u01(b[0], b[1]); q[3] = -2.5 + 5.0 * u01(b[2], b[3]);
        if (is_valid_init(q) || k >= 1023) break;
      }
    } else {
      q[0] = 0.0; q[1] = 0.0; q[2] = 2.5; q[3] = 0.0; /* initial_states[0] :11 */
    }
    st->goal_idx[i] = sample_goal(cfg, i, next_goal_idx);
  }
  st->steps_since_reset[i] = 0;       /* persistent_state_wrapper.py:18-19 */
  st->num_interventions[i] += 1;
  if (st->steps_since_goal_change) st->steps_since_goal_change[i] = 0; /* lifelong_wrapper.py:26-27 */
}

/* One wrapped step of env i; counter semantics documented in include/earl_tabletop.h */
static void step_env(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int32_t i, const float* act,
                     const int32_t* next_goal_idx, float* obs, float* reward, uint8_t* done, uint8_t* success) {
  double* q = st->qpos + 4 * (size_t)i;
  double a[3] = {rescale_action(act[0]), rescale_action(act[1]), rescale_action(act[2])};
  move(q, &st->attached[i], a, 1);
  float o[12];
  get_obs(q, st->attached[i], st->goal_table + 6 * (size_t)st->goal_idx[i], 1, o);
  double r = compute_reward(o, cfg->reward_type, cfg->wide_init);
  int succ = is_successful(o, cfg->wide_init);
  st->steps_since_reset[i] += 1;                                   /* persistent_state_wrapper.py:25-26 */
  int d = st->steps_since_reset[i] >= cfg->horizon;                /* :28-29 */
  if (cfg->goal_change_frequency > 0) {                            /* lifelong_wrapper.py:30-44 */
    st->steps_since_goal_change[i] += 1;
    st->lifelong_return[i] += r;
    if (st->steps_since_goal_change[i] >= cfg->goal_change_frequency) {
      st->steps_since_goal_change[i] = 0;
      st->goal_idx[i] = sample_goal(cfg, i, next_goal_idx);
      get_obs(q, st->attached[i], st->goal_table + 6 * (size_t)st->goal_idx[i], 1, o); /* obs re-read with the new goal */
    }
  }
  if (obs) memcpy(obs, o, sizeof o);
  if (reward) *reward = (float)r;
  if (done) *done = (uint8_t)d;
  if (success) *success = (uint8_t)succ;
  if (d && cfg->auto_reset) reset_env(cfg, st, i, next_goal_idx);
}

/* ------------------------------------------------------------------ batched entry points (host pointers) */
int oracle_tabletop_step(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const float* act,
                         const int32_t* next_goal_idx, const earl_tabletop_out* out) {
#pragma omp parallel for schedule(static)
  for (int32_t i = 0; i < cfg->n; ++i)
    step_env(cfg, st, i, act + 3 * (size_t)i, next_goal_idx, out->obs ? out->obs + 12 * (size_t)i : 0,
             out->reward ? out->reward + i : 0, out->done ? out->done + i : 0, out->success ? out->success + i : 0);
  return EARL_OK;
}

int oracle_tabletop_rollout(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int32_t T, const float* act,
                            const earl_tabletop_out* out) {
  /* envs are independent: each thread owns a block of envs and walks time in the outer loop, so its reads of act and
     its writes of obs/reward/flags are contiguous runs per step (cache/TLB friendly for the CPU-baseline timing) */
  const size_t n = (size_t)cfg->n;
  const int32_t blk = 64;
#pragma omp parallel for schedule(static)
  for (int32_t i0 = 0; i0 < cfg->n; i0 += blk) {
    const int32_t i1 = i0 + blk < cfg->n ? i0 + blk : cfg->n;
    earl_tabletop_cfg c = *cfg;
    for (int32_t t = 0; t < T; ++t) {
      c.counter = cfg->counter + (uint64_t)t;
      for (int32_t i = i0; i < i1; ++i) {
        size_t row = (size_t)t * n + (size_t)i;
        step_env(&c, st, i, act + 3 * row, 0, out->obs ? out->obs + 12 * row : 0, out->reward ? out->reward + row : 0,
                 out->done ? out->done + row : 0, out->success ? out->success + row : 0);
      }
    }
  }
  return EARL_OK;
}

int oracle_tabletop_observe(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const earl_tabletop_out* out) {
  for (int32_t i = 0; i < cfg->n; ++i) {
    float o[12];
    get_obs(st->qpos + 4 * (size_t)i, st->attached[i], st->goal_table + 6 * (size_t)st->goal_idx[i], 1, o);
    if (out->obs) memcpy(out->obs + 12 * (size_t)i, o, sizeof o);
    if (out->reward) out->reward[i] = (float)compute_reward(o, cfg->reward_type, cfg->wide_init);
    if (out->success) out->success[i] = (uint8_t)is_successful(o, cfg->wide_init);
    if (out->done) out->done[i] = (uint8_t)(st->steps_since_reset[i] >= cfg->horizon);
  }
  return EARL_OK;
}

int oracle_tabletop_reset(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const uint8_t* mask,
                          const int32_t* next_goal_idx, float* obs) {
  for (int32_t i = 0; i < cfg->n; ++i) {
    if (!mask || mask[i]) reset_env(cfg, st, i, next_goal_idx);
    if (obs) get_obs(st->qpos + 4 * (size_t)i, st->attached[i], st->goal_table + 6 * (size_t)st->goal_idx[i], 1, obs + 12 * (size_t)i);
  }
  return EARL_OK;
}

int oracle_tabletop_reward(int32_t n, const float* obs, int32_t reward_type, int32_t wide_init, float* reward,
                           double* reward_f64, uint8_t* success) {
  for (int32_t i = 0; i < n; ++i) {
    double r = compute_reward(obs + 12 * (size_t)i, reward_type, wide_init);
    if (reward) reward[i] = (float)r;
    if (reward_f64) reward_f64[i] = r;
    if (success) success[i] = (uint8_t)is_successful(obs + 12 * (size_t)i, wide_init);
  }
  return EARL_OK;
}

int oracle_tabletop_valid_init(int32_t n, const double* cand, uint8_t* valid) {
  for (int32_t i = 0; i < n; ++i) valid[i] = (uint8_t)is_valid_init(cand + 4 * (size_t)i);
  return EARL_OK;
}

/* ------------------------------------------------------------------ 3-object variant */
static void step_env3(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int32_t i, const float* act,
                      float* obs, float* reward, uint8_t* done, uint8_t* success) {
  double* q = st->qpos + 8 * (size_t)i;
  double a[3] = {rescale_action(act[0]), rescale_action(act[1]), rescale_action(act[2])};
  move(q, &st->attached[i], a, 3);
  float o[20];
  get_obs(q, st->attached[i], st->goal_table + 10 * (size_t)st->goal_idx[i], 3, o);
  double r = compute_reward3(o, cfg->reward_type);
  st->steps_since_reset[i] += 1;
  int d = st->steps_since_reset[i] >= cfg->horizon;
  if (obs) memcpy(obs, o, sizeof o);
  if (reward) *reward = (float)r;
  if (done) *done = (uint8_t)d;
  if (success) *success = (uint8_t)is_successful3(o);
}

static void reset_env3(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int32_t i) {
  static const double init[8] = {0.0, 0.0, 2.5, 0.0, 2.5, -1.0, 2.5, 1.0}; /* 3obj initial_states :11 */
  double* q = st->qpos + 8 * (size_t)i;
  st->attached[i] = -1;
  if (cfg->reset_at_goal) { /* :64-69: qpos[:8] = goal[:8] + np.random.uniform(-0.3, 0.3, size=8) */
    st->goal_idx[i] = sample_goal(cfg, i, 0);
    const double* g = st->goal_table + 10 * (size_t)st->goal_idx[i];
    for (int j = 0; j < 4; ++j) {
      uint32_t b[4];
      draw_block(cfg, i, 1 + (uint32_t)j, b);
      q[2 * j] = g[2 * j] + (-0.3 + 0.6 * u01(b[0], b[1]));
      q[2 * j + 1] = g[2 * j + 1] + (-0.3 + 0.6 * u01(b[2], b[3]));
    }
  } else {
    for (int k = 0; k < 8; ++k) q[k] = init[k];
    st->goal_idx[i] = sample_goal(cfg, i, 0); /* np.random.randint(len(goal_list)) :52-56 */
  }
  st->steps_since_reset[i] = 0;
  st->num_interventions[i] += 1;
}

int oracle_tabletop3_step(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const float* act,
                          const earl_tabletop_out* out) {
#pragma omp parallel for schedule(static)
  for (int32_t i = 0; i < cfg->n; ++i)
    step_env3(cfg, st, i, act + 3 * (size_t)i, out->obs ? out->obs + 20 * (size_t)i : 0, out->reward ? out->reward + i : 0,
              out->done ? out->done + i : 0, out->success ? out->success + i : 0);
  return EARL_OK;
}

int oracle_tabletop3_rollout(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int32_t T, const float* act,
                             const earl_tabletop_out* out) {
  const size_t n = (size_t)cfg->n;
#pragma omp parallel for schedule(static)
  for (int32_t i = 0; i < cfg->n; ++i)
    for (int32_t t = 0; t < T; ++t) {
      size_t row = (size_t)t * n + (size_t)i;
      step_env3(cfg, st, i, act + 3 * row, out->obs ? out->obs + 20 * row : 0, out->reward ? out->reward + row : 0,
                out->done ? out->done + row : 0, out->success ? out->success + row : 0);
    }
  return EARL_OK;
}

int oracle_tabletop3_reset(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const uint8_t* mask, float* obs) {
  for (int32_t i = 0; i < cfg->n; ++i) {
    if (!mask || mask[i]) reset_env3(cfg, st, i);
    if (obs) get_obs(st->qpos + 8 * (size_t)i, st->attached[i], st->goal_table + 10 * (size_t)st->goal_idx[i], 3, obs + 20 * (size_t)i);
  }
  return EARL_OK;
}

int oracle_tabletop3_reward(int32_t n, const float* obs, int32_t reward_type, float* reward, double* reward_f64,
                            uint8_t* success) {
  for (int32_t i = 0; i < n; ++i) {
    double r = compute_reward3(obs + 20 * (size_t)i, reward_type);
    if (reward) reward[i] = (float)r;
    if (reward_f64) reward_f64[i] = r;
    if (success) success[i] = (uint8_t)is_successful3(obs + 20 * (size_t)i);
  }
  return EARL_OK;
}

const char* oracle_version(void) { return "earl-tabletop-oracle 1"; }

#ifdef _OPENMP
#include <omp.h>
int oracle_set_threads(int n) { if (n > 0) omp_set_num_threads(n); return omp_get_max_threads(); }
#else
int oracle_set_threads(int n) { (void)n; return 1; }
#endif
