/* earl_tabletop.h -- C ABI of the MI355X-native tabletop_manipulation step()/reset() hot path.
 *
 * The reference (architsharma97/earl_benchmark) has NO FFI for this path: the env is an ordinary
 * Python object.  Each entry point below therefore cites the Python call it
 * replaces; INTEGRATION.md shows the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (HBM) owned by the caller; the library allocates nothing,
 *     never synchronises, and enqueues on the caller's HIP stream (earl_stream_t == hipStream_t);
 *   - one row per env; a "shard" is a contiguous range of `n` envs whose global ids are
 *     env_offset .. env_offset+n-1 (RNG streams are keyed by the GLOBAL id, so results do not
 *     depend on how envs are sharded over GPUs);
 *   - functions return EARL_OK (0) or a negative EARL_ERR_* code; they never throw.
 *     earl_last_error() returns a thread-local message for the last failure;
 *   - distinct state buffers may be driven concurrently from different threads/streams.
 *
 * Numerics (tested bit-exact against oracle/ and the golden vectors in tests/golden/):
 *   state is fp64, observations are the fp32 rounding of the state, discrete outputs (attached
 *   flag, done, success, sparse reward) are bit-exact, the dense reward is within 1e-6 relative.
 */
#ifndef EARL_TABLETOP_H
#define EARL_TABLETOP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* earl_stream_t; /* a hipStream_t; NULL = the default stream */

enum { EARL_OK = 0, EARL_ERR_ARG = -1, EARL_ERR_LAUNCH = -2, EARL_ERR_NODEVICE = -3 };
enum { EARL_REWARD_SPARSE = 0, EARL_REWARD_DENSE = 1 };

#define EARL_TABLETOP_OBS_DIM 12  /* qpos[4], attached flag x2, goal[6]: tabletop_manipulation.py:55-60 */
#define EARL_TABLETOP_ACT_DIM 3   /* dx, dy, grip:                     tabletop_manipulation.py:128-132 */
#define EARL_TABLETOP3_OBS_DIM 20 /* qpos[8], flag x2, goal[10]:       tabletop_manipulation_3obj.py:45-50 */

/* Static configuration of one batched env (constructor arguments of the reference's classes). */
typedef struct earl_tabletop_cfg {
  int32_t n;             /* envs in this shard */
  int32_t env_offset;    /* global id of row 0 */
  int32_t reward_type;   /* EARL_REWARD_*:      TabletopManipulation(reward_type=...)  tabletop_manipulation.py:26 */
  int32_t wide_init;     /* wide_init_distr: object-only success test (:201-202) + rejection-sampled reset (:114-117) */
  int32_t reset_at_goal; /* reset_at_goal (:109-111) */
  int32_t horizon;       /* PersistentStateWrapper(episode_horizon)  wrappers/persistent_state_wrapper.py:10-13 */
  int32_t goal_change_frequency; /* LifelongWrapper(goal_change_frequency), 0 = not lifelong  wrappers/lifelong_wrapper.py:18-23 */
  int32_t auto_reset;    /* batched-only extension: 1 = an env whose done fired is reset in the same launch
                            (outputs of that step are still the terminal ones); 0 = reference behaviour */
  int32_t n_goals;       /* rows of state.goal_table */
  int32_t n_sample_goals;/* get_next_goal() draws uniformly from rows 0..n_sample_goals-1 (:62-76; 4 tasks) */
  uint64_t seed;         /* Philox4x32-10 key */
  uint64_t counter;      /* Philox counter word: the caller passes a fresh value on every call that may draw */
} earl_tabletop_cfg;

/* Persistent per-env state (all arrays have cfg.n rows). */
typedef struct earl_tabletop_state {
  double* qpos;                     /* [n,4] gripper x,y then mug x,y  (sim.data.qpos[:4], fp64 like MuJoCo) */
  int8_t* attached;                 /* [n]  -1 free, 0 holding the mug (attached_object, :42) */
  int32_t* goal_idx;                /* [n]  row of goal_table currently set as self.goal */
  const double* goal_table;         /* [n_goals,6] rows 0..3 = goal_states-derived goals (:12-16, :62-76); the
                                       caller may append rows for reset_goal(goal) with arbitrary goals (:78-81) */
  int32_t* steps_since_reset;       /* [n]  PersistentStateWrapper._steps_since_reset */
  int32_t* num_interventions;       /* [n]  PersistentStateWrapper._num_interventions */
  int32_t* steps_since_goal_change; /* [n]  LifelongWrapper, may be NULL when goal_change_frequency == 0 */
  double* lifelong_return;          /* [n]  LifelongWrapper._lifelong_return, may be NULL likewise */
  const uint64_t* counter_base;     /* may be NULL.  Device word ADDED to cfg.counter by earl_tabletop_step / earl_tabletop3_step (only): the Philox counter of a
                                       step launch captured into a HIP graph is a kernel argument frozen at capture time; with the base in device memory the
                                       captured launch of step t (cfg.counter = t) draws with base + t, and the host refreshes the one word before each replay
                                       -- lifelong goal switching and auto-reset inside a captured step loop (envs/tabletop.py StepGraph) */
} earl_tabletop_state;

/* Outputs of one step (rows = envs) or of one rollout (rows = [T, n]). */
typedef struct earl_tabletop_out {
  float* obs;       /* [.., 12] */
  float* reward;    /* [..] */
  uint8_t* done;    /* [..] 0/1: horizon reached (the env itself never terminates, :137) */
  uint8_t* success; /* [..] 0/1: is_successful(next_obs) (:197-204) */
  double* reward_f64; /* [n] may be NULL; written by earl_tabletop_step / earl_tabletop3_step only: the reward before it is rounded to
                         float32 (the reference's compute_reward returns a Python float under its pinned numpy 1.22, :176-191) */
} earl_tabletop_out;

/* Lifelong(PersistentStateWrapper(TabletopManipulation)).step(action) for every env of the shard.
 * Replaces: TabletopManipulation.step/move/_get_obs/compute_reward/is_successful
 *           (envs/tabletop_manipulation.py:128-204), PersistentStateWrapper.step
 *           (wrappers/persistent_state_wrapper.py:22-31), LifelongWrapper.step
 *           (wrappers/lifelong_wrapper.py:30-44).
 * act [n,3] fp32.  next_goal_idx: NULL, or [n] goal rows to use instead of the RNG whenever this call
 * resamples a goal (lifelong switch / auto-reset) -- the injection hook parity tests use, like
 * reset_goal(goal) in the reference. */
int earl_tabletop_step(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const float* act,
                       const int32_t* next_goal_idx, const earl_tabletop_out* out, earl_stream_t stream);

/* T consecutive steps in ONE launch: state stays in registers, act [T,n,3], outputs [T,n,..].
 * Equivalent to T calls of earl_tabletop_step with counter, counter+1, ... (bit-identical outputs). */
int earl_tabletop_rollout(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int32_t T,
                          const float* act, const earl_tabletop_out* out, earl_stream_t stream);

/* One evaluation episode per env in ONE launch: earl_tabletop_reset (all envs, Philox counter cfg->counter) followed by
 * earl_tabletop_rollout of T steps (counters cfg->counter + 1 ...).  Bit-identical to the two calls; the caller advances
 * its counter by T + 1.  (With lifelong switching / auto-reset enabled it is executed as the two launches.) */
int earl_tabletop_reset_rollout(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int32_t T,
                                const float* act, const earl_tabletop_out* out, earl_stream_t stream);

/* `episodes` evaluation episodes per env, back to back: episode e = earl_tabletop_reset_rollout with Philox counter
 * cfg->counter + e * (T + 1), its outputs at rows [e, T, n, ..] of `out` (obs [episodes, T, n, 12], ...), its actions at
 * act + e * act_episode_stride floats (act_episode_stride = T * n * 3: a contiguous [episodes, T, n, 3] array; 0: every episode replays
 * the same [T, n, 3] actions).  Bit-identical to that sequence of calls; the caller advances its counter by episodes * (T + 1).
 * This is the reference's evaluation loop (`for _ in range(num_eval_episodes): obs = env.reset(); while not done: env.step(...)`
 * over PersistentStateWrapper, persistent_state_wrapper.py:17-31) for the whole batch.  When the wave-specialised kernel applies
 * (no lifelong switching / auto-reset, all four outputs, T a multiple of 8) ALL episodes run in ONE launch: the launch's fixed cost
 * (prologue, pipeline fill and drain) is paid once instead of once per episode.  The episodes being independent of one another (each starts
 * with the reset), a batch that leaves CUs idle (up to 8192 envs) has several of them IN FLIGHT at a time, each group of episodes on its own
 * workgroups; outputs and the state left behind are those of the sequence, bit for bit. */
int earl_tabletop_eval_episodes(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int32_t episodes, int32_t T, const float* act,
                                int64_t act_episode_stride, const earl_tabletop_out* out, earl_stream_t stream);

/* PersistentStateWrapper.reset() + TabletopManipulation.reset() for the envs with mask[i] != 0
 * (mask == NULL: all).  Replaces wrappers/persistent_state_wrapper.py:17-20 and
 * envs/tabletop_manipulation.py:105-126 (incl. is_valid_init :89-97, get_next_goal :62-76).
 * obs (may be NULL): [n,12] current observation of EVERY env after the reset. */
int earl_tabletop_reset(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const uint8_t* mask,
                        const int32_t* next_goal_idx, float* obs, earl_stream_t stream);

/* _get_obs() / is_successful() / compute_reward(_get_obs()) of the current state; any output may be NULL. */
int earl_tabletop_observe(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st,
                          const earl_tabletop_out* out, earl_stream_t stream);

/* compute_reward(obs) and is_successful(obs) on caller-supplied observations obs [n,12]
 * (envs/tabletop_manipulation.py:176-204).  reward / success may be NULL. */
int earl_tabletop_reward(int32_t n, const float* obs, int32_t reward_type, int32_t wide_init,
                         float* reward, uint8_t* success, earl_stream_t stream);

/* is_valid_init(state, goal_states) (:89-97) on candidates cand [n,4] -> valid [n] 0/1. */
int earl_tabletop_valid_init(int32_t n, const double* cand, uint8_t* valid, earl_stream_t stream);

/* ---- 3-object variant (envs/tabletop_manipulation_3obj.py; not wired into the reference's loader) ----
 * qpos [n,8], attached in {-1,0,1,2} (object index; the reference encodes it as (0,0)/(.5,.5)/(1,1)),
 * goal_table [n_goals,10], obs [n,20].  Same structs; reset_at_goal = goal + U(-0.3,0.3)^8 (:64-69); wide_init and
 * goal_change_frequency must be 0 (the reference class has neither). */
int earl_tabletop3_step(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const float* act,
                        const earl_tabletop_out* out, earl_stream_t stream);
int earl_tabletop3_rollout(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int32_t T,
                           const float* act, const earl_tabletop_out* out, earl_stream_t stream);
int earl_tabletop3_reset(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const uint8_t* mask,
                         float* obs, earl_stream_t stream);
int earl_tabletop3_reward(int32_t n, const float* obs, int32_t reward_type, float* reward, uint8_t* success,
                          earl_stream_t stream);

/* ---- host build: the `_cpu` entry points (BASELINE.json configs[0] "1 env, CPU ... plumbing, no GPU") ----
 * csrc/libearl_host.so = the SAME per-env functions the gfx950 kernels run (csrc/tabletop_device.h, tabletop_step.h, philox.h), compiled for the host by
 * g++ with -ffp-contract=off; one OpenMP iteration per env where a kernel has one lane per env.  Same structs, same arguments minus the stream, HOST
 * pointers, synchronous.  Outputs are bit-identical to the device entry points for every discrete output, the fp64 state and the f32 observations; the
 * dense reward differs by the device's exp() (1e-6, like device vs oracle).  Replaces the same reference calls as the twin of each name, for ONE env or a
 * batch: envs/tabletop_manipulation.py:128-138 (step), :105-126 (reset), wrappers/persistent_state_wrapper.py:17-31, wrappers/lifelong_wrapper.py:30-44.
 * Nothing in the library falls back to these: a caller asks for them (Python: EARLEnvs(..., device='cpu')).
 * earl_host_last_error() holds the message of the last failure of THESE calls. */
int earl_tabletop_step_cpu(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const float* act, const int32_t* next_goal_idx,
                           const earl_tabletop_out* out);
int earl_tabletop_rollout_cpu(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int32_t T, const float* act, const earl_tabletop_out* out);
int earl_tabletop_reset_rollout_cpu(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int32_t T, const float* act, const earl_tabletop_out* out);
int earl_tabletop_eval_episodes_cpu(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int32_t episodes, int32_t T, const float* act,
                                    int64_t act_episode_stride, const earl_tabletop_out* out);
int earl_tabletop_reset_cpu(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const uint8_t* mask, const int32_t* next_goal_idx, float* obs);
int earl_tabletop_observe_cpu(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const earl_tabletop_out* out);
int earl_tabletop_reward_cpu(int32_t n, const float* obs, int32_t reward_type, int32_t wide_init, float* reward, uint8_t* success);
int earl_tabletop_valid_init_cpu(int32_t n, const double* cand, uint8_t* valid);
int earl_tabletop3_step_cpu(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const float* act, const earl_tabletop_out* out);
int earl_tabletop3_rollout_cpu(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, int32_t T, const float* act, const earl_tabletop_out* out);
int earl_tabletop3_reset_cpu(const earl_tabletop_cfg* cfg, const earl_tabletop_state* st, const uint8_t* mask, float* obs);
int earl_tabletop3_reward_cpu(int32_t n, const float* obs, int32_t reward_type, float* reward, uint8_t* success);
int earl_host_set_threads(int n);        /* OpenMP threads of the calls above (n <= 0: query); returns the count in force */
const char* earl_host_version(void);
const char* earl_host_last_error(void);

/* ---- library ---- */
/* Test/bench hook: which kernel earl_tabletop_rollout uses. 0 = automatic (the wave-specialised kernel whenever
 * lifelong switching and auto-reset are off and all four outputs are requested), 1 = always the plain
 * one-lane-per-env kernel.  Both produce bit-identical outputs.  Returns the previous setting. */
int earl_debug_set_rollout_impl(int impl);
int earl_debug_set_rollout_wgs_per_cu(int k);
/* Diagnostic: per-workgroup cycle sums of the instrumented rollout variant (impl 9); blocks until the copy is done. */
int earl_debug_read_ws_profile(uint64_t* out, int32_t n_words);
const char* earl_version(void);
const char* earl_last_error(void);
int earl_device_count(void); /* number of HIP devices visible, <= 0 when there is none */

#ifdef __cplusplus
}
#endif
#endif /* EARL_TABLETOP_H */
