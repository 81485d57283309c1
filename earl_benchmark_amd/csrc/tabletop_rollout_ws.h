// tabletop_rollout_ws.h -- wave-specialised fused rollout kernel (the bench's dominant kernel).
//
// Why: a T-step rollout is a bit-exact fp64 recurrence per env, so time cannot be parallelised; at
// N = 4096 there are only 64 wavefronts and the launch is bound by ONE wave's per-step latency.  The
// plain one-lane-per-env loop (rollout_kernel) puts loads, ~100 instructions of formatting and six
// partially-coalesced stores on that critical path, and because vmcnt retires in order every wait for
// an action load also waits for all older stores.  Here a 64-env workgroup splits the step by ROLE:
//
//   wave 0         COMPUTE  the recurrence only: reads rescaled actions from LDS, writes the 4 f32
//                           coordinates + attached flag into an LDS "row image" of the obs rows
//   waves 1..NL    LOADERS  stream raw actions HBM -> VGPRs a whole TRIP (LEAD chunks) ahead with coalesced
//                           dword loads (no stores in these waves, so their vmcnt waits never queue behind
//                           stores), transpose them through LDS and rescale them to fp64 -> LDS ring
//   waves NL+1..   STORERS  copy finished row images LDS -> HBM as fully coalesced float4 (the goal part of
//                           each row is constant and pre-filled), evaluate success/reward, pack flags
//
// One workgroup barrier per chunk of K steps; 3-deep action ring and 2-deep row-image ring in LDS.
// Semantics are exactly those of rollout_kernel<1> without lifelong / auto-reset (the host picks the
// kernel); outputs are bit-identical (tests/test_tabletop_gpu.py::test_rollout_kernels_agree).
//
// Measured (profiles/, tools/prof_ws.py): the first version of this kernel spent 318 cycles per step in
// the compute wave although it issues < 40 instructions -- the attach test (sub, mul, fma, cmp, mask,
// select) sat on the loop-carried path of the object position.  ws_step() below removes it from that
// path: whenever the test matters the object has not moved during the previous step, so the test can
// use the object position from ONE STEP EARLIER, which is ready long before.
#pragma once
#include <type_traits>

#include "tabletop_device.h"

#ifndef EARL_WSM_OFF
#define EARL_WSM_OFF 0      // measurement only: bit 0 loaders, 1 storers, 2 compute waves ignore the episode structure of a multi-episode launch (WRONG results)
#endif

// Memory-policy switches of the HBM-bound regime (round 3; measured with every evaluation episode reading its OWN actions, i.e. all 66 B per
// env-step crossing the HBM interface: 4096 envs x 200 steps x 28 episodes, four in flight, one MI355X, three repetitions on one box,
// gpurun_out/own_actions_experiment3.txt; tools/build_ws_variant.sh builds tools/ubench/libearl_ws_<tag>.so with other settings):
//   EARL_WS_SMALL_NT   (on)  reward / done / success rows are stored non-temporally like the observation rows: 312-326 -> 285-290 us per launch
//                            with 16-step chunks, 288-298 -> 254-260 us with 8-step chunks.  As ordinary stores those 6 of the 54 written
//                            bytes per env-step sat in the L2s as partial dirty lines beside the action reads.
//   EARL_WS_XCD_REMAP  (on)  hardware workgroup b works on env block (b % 8) * (grid / 8) + b / 8: dispatch is round-robin over the 8 XCDs, so
//                            the workgroups of one XCD now cover CONSECUTIVE env blocks and the two 64-byte halves of a done / success line
//                            (64 envs each) meet in one L2: 254-260 -> 245-249 us (6.1-6.2 TB/s).
//   EARL_WS_LD_AUX=<bits>    (off) actions through raw buffer loads with cache-policy bits (1 sc0, 2 nt, 16 sc1): sc1 no gain; nt loads help
//                            only while the small stores are ordinary (249-267 us) and cost 25 % once they are non-temporal (316-323 us)
//   EARL_WS_ST_AUX=<bits>    (off) observation rows through raw buffer stores with these bits instead of the nt builtin: nt | sc1 = nt; sc1 alone slower
#if !defined(EARL_WS_NO_SMALL_NT) && !defined(EARL_WS_SMALL_NT)
#define EARL_WS_SMALL_NT 1
#endif
#if !defined(EARL_WS_NO_XCD_REMAP) && !defined(EARL_WS_XCD_REMAP)
#define EARL_WS_XCD_REMAP 1
#endif
namespace earl {

typedef int ws_v4i __attribute__((ext_vector_type(4)));
typedef float ws_v4f __attribute__((ext_vector_type(4)));
// buffer resource over [p, p + 4 GiB) for the raw buffer loads / stores of the experiments (wave-uniform base: the descriptor lives in SGPRs)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t ws_rsrc(const void* p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, -1, 0x00020000);
}

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
  float grip_x;                       // smallest float x with rescale_action(x) > 0 (host: exact search)
  // fused reset: reset_first != 0 performs earl_tabletop_reset (all envs, counter cfg.counter) in the prologue and
  // runs the steps with counters cfg.counter + 1 ..; cfg carries seed / env_offset / reset mode / n_sample_goals
  int32_t reset_first;
  earl_tabletop_cfg cfg;
  int32_t* __restrict__ goal_idx_w;         // same array as goal_idx (written by the fused reset)
  int32_t* __restrict__ num_interventions;
  // several evaluation episodes in ONE launch (earl_tabletop_eval_episodes; NC == 3 kernels only): T = episodes * Tep steps, every
  // episode starts with a reset (counter cfg.counter + ep * (Tep + 1), exactly what `episodes` fused reset+rollout launches would use);
  // Tep % K == 0 and Tep >= 2 K (host-checked).  The outputs are [episodes, Tep, n, ..] = linear in the global step index; the
  // actions of episode ep start at act + ep * act_ep_stride floats (0: every episode replays the same [Tep, n, 3] actions).
  int32_t episodes, Tep;
  long long act_ep_stride;
  // ... and, the evaluation episodes of a launch being independent of one another (each starts with reset()), SEVERAL OF THEM IN FLIGHT: the grid is
  // ep_groups x wgs workgroups, group g walks episodes [g ep_per_group, (g + 1) ep_per_group) of the launch on its own wgs workgroups (its own
  // rows of the outputs and of the actions, the Philox counters those episodes have in the sequence); the group of the LAST episode leaves the
  // env state behind, as the sequence would.  ep_groups = 1: the plain sequence.
  int32_t wgs, ep_groups, ep_per_group;
};

// Philox counter of the reset that starts episode `ep` of a launch
__device__ __forceinline__ uint64_t ws_ep_counter(const WsArgs& a, int ep) { return a.cfg.counter + (uint64_t)ep * (uint64_t)(a.Tep + 1); }
__device__ __forceinline__ int ws_goal_row_ep(const WsArgs& a, int env, int ep) {
  if (a.reset_first) return sample_goal(a.cfg, ws_ep_counter(a, ep), env, nullptr);
  return a.goal_idx[env];
}

// goal row of env `e` for this launch: the stored one, or the one the fused reset samples (every wave that needs it
// recomputes the same Philox draw instead of waiting for another wave to publish it)
__device__ __forceinline__ int ws_goal_row(const WsArgs& a, int env) {
  if (a.reset_first) return sample_goal(a.cfg, a.cfg.counter, env, nullptr);
  return a.goal_idx[env];
}

// Lanes of ONE wave exchange data through LDS (loader staging).  Per thread the write and read addresses never
// alias, so without a fence hipcc reorders them freely (it did: seen in the ISA); a wavefront-scope fence costs no
// instruction -- the LDS queue of a wave is already in order -- it only pins the compiler's order.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// fast clip for values that are not NaN (v_max_f64 / v_min_f64); NaN handling is done by the EXACT path
__device__ __forceinline__ double clip_fast(double x) { return fmin(fmax(x, -2.8), 2.8); }

// One step of move :140-174 on register state.
//   EXACT : np.clip NaN propagation and no range assumptions (compare+select, the textbook order).
//   fast  : requires no NaN anywhere and |object| <= 2.8 (true after any clipped move; checked at kernel start):
//           - the attach test reads (oxl, oyl) = the object position at the START OF THE PREVIOUS STEP.  If the mug
//             was not attached during the previous step it did not move, so this is its current position; if it was
//             attached, `att` is already true and the test result is ignored.  This takes the test off the
//             loop-carried dependency of the object position (3 dependent fp64 ops per step remain).
//           - a free object is "moved" by -0.0 and clipped: x + (-0.0) == x bit for bit, clip is the identity in range.
template <bool EXACT>
__device__ __forceinline__ void ws_step(double& fx, double& fy, double& ox, double& oy, double& oxl, double& oyl,
                                        bool& att, double a0, double a1, bool grip, const Thresholds& th) {
  if constexpr (EXACT) {
    const double dx = fx - ox, dy = fy - oy;
    const bool near = fma(dy, dy, dx * dx) < th.grasp_d2;
    att = grip && (att || near);
    const double nfx = clipd(fx + a0, -2.8, 2.8), nfy = clipd(fy + a1, -2.8, 2.8);
    if (att) {
      ox = clipd(ox + (nfx - fx), -2.8, 2.8);
      oy = clipd(oy + (nfy - fy), -2.8, 2.8);
    }
    fx = nfx;
    fy = nfy;
    oxl = ox;
    oyl = oy;
  } else {
    const double dx = fx - oxl, dy = fy - oyl;
    const bool near = fma(dy, dy, dx * dx) < th.grasp_d2;
    att = grip && (att || near);
    oxl = ox;
    oyl = oy;
    const double nfx = clip_fast(fx + a0), nfy = clip_fast(fy + a1);
    const double ddx = att ? nfx - fx : -0.0, ddy = att ? nfy - fy : -0.0;
    ox = clip_fast(ox + ddx);
    oy = clip_fast(oy + ddy);
    fx = nfx;
    fy = nfy;
  }
}

// Fast step in the form the hardware likes.  Measured on MI355X (tools/ubench/cu_issue*.hip), ticks per instruction of
// ONE wave / instructions per tick a whole CU sustains:   VALU with VGPR or inline operands 5.3 / 1.5-1.8;
// ANY instruction that reads or writes an SGPR (SALU, v_cmp, v_cndmask with a mask, v_readfirstlane, a VALU with a
// scalar source operand such as the clip bounds) 8.3 / 1.0 -- the scalar side is one shared pipe per CU;
// v_cvt_f32_f64 8.3; v_permlane32_swap 20.  So this version keeps every operand in VGPRs: predicates are int lane
// masks built from sign bits (d2 < thr  <=>  sign(d2 - thr), exact for non-NaN), selections are and / bfi on the halves
// of the doubles, the clip bounds and the threshold are pinned in VGPRs.  Same arithmetic, bit for bit.
struct WsConst { double lo, hi, thr; int nz; };   // clip bounds, attach threshold, high word of -0.0
__device__ __forceinline__ void ws_step_v(double& fx, double& fy, double& ox, double& oy, double& oxl, double& oyl,
                                          int& att, double a0, double a1, int grip, const WsConst& k) {
  const double dx = fx - oxl, dy = fy - oyl;
  const int near = __double2hiint(fma(dy, dy, dx * dx) - k.thr) >> 31;   // all ones iff d2 < thr
  att = grip & (att | near);
  oxl = ox;
  oyl = oy;
  const double nfx = fmin(fmax(fx + a0, k.lo), k.hi), nfy = fmin(fmax(fy + a1, k.lo), k.hi);
  const double ddx = nfx - fx, ddy = nfy - fy;
  // att ? dd : -0.0, on the two halves of the double
  const double sx = __hiloint2double((att & __double2hiint(ddx)) | (~att & k.nz), att & __double2loint(ddx));
  const double sy = __hiloint2double((att & __double2hiint(ddy)) | (~att & k.nz), att & __double2loint(ddy));
  ox = fmin(fmax(ox + sx, k.lo), k.hi);
  oy = fmin(fmax(oy + sy, k.lo), k.hi);
  fx = nfx;
  fy = nfy;
}

// The same step with the two coordinates of an env in two LANES (l: x, l+32: y) of a 32-env wave: every vector
// instruction advances x and y at once, which halves the instructions a wave issues per step -- and a lone wave
// issues one VALU per ~5.6 cycles (10.3 if it depends on the previous one; tools/ubench/issue.hip), fp32 and fp64
// alike, so the instruction count IS the step latency.  Only the attach test couples the halves: the Y half needs
// dx*dx (two v_permlane32_swap), and its compare mask is copied to the X half with scalar ops.
//   f, o, ol: this lane's coordinate of gripper / mug / mug-one-step-ago; att64, grip64: lane masks, both halves equal.
__device__ __forceinline__ double both_halves_from_lower(double v) {
  const unsigned lo = __double2loint(v), hi = __double2hiint(v);
  const auto r0 = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
  const auto r1 = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
  return __hiloint2double(r1[0], r0[0]);
}
template <bool EXACT>
__device__ __forceinline__ void ws_step_xy(double& f, double& o, double& ol, unsigned long long& att64, double a,
                                           unsigned long long grip64, const Thresholds& th) {
  const double d = f - (EXACT ? o : ol);
  const double d2 = fma(d, d, both_halves_from_lower(d * d));   // meaningful in the Y half: fma(dy, dy, dx*dx)
  const unsigned near_y = (unsigned)(__ballot(d2 < th.grasp_d2) >> 32);
  const unsigned long long near64 = ((unsigned long long)near_y << 32) | near_y;
  att64 = grip64 & (att64 | near64);
  const bool att = __builtin_amdgcn_inverse_ballot_w64(att64);
  if constexpr (EXACT) {
    const double nf = clipd(f + a, -2.8, 2.8);
    if (att) o = clipd(o + (nf - f), -2.8, 2.8);
    f = nf;
    ol = o;
  } else {
    ol = o;
    const double nf = clip_fast(f + a);
    const double dd = att ? nf - f : -0.0;
    o = clip_fast(o + dd);
    f = nf;
  }
}

// Fast path of ws_step_xy, software-pipelined by one step: `nm` is the attach-test mask of the step being executed,
// computed during the PREVIOUS step from (gripper after that step, mug before that step) -- by the argument above
// that is the mug position whenever the test matters -- so the 10-instruction test chain (sub, mul, 2x swap, fma,
// cmp, mask ops) runs beside the object update and the output conversion instead of in front of them.
__device__ __forceinline__ unsigned long long ws_near_mask_xy(double f, double o, const Thresholds& th) {
  const double d = f - o;
  const double d2 = fma(d, d, both_halves_from_lower(d * d));   // meaningful in the Y half: fma(dy, dy, dx*dx)
  const unsigned near_y = (unsigned)(__ballot(d2 < th.grasp_d2) >> 32);
  return ((unsigned long long)near_y << 32) | near_y;
}
__device__ __forceinline__ void ws_step_xy_pipe(double& f, double& o, unsigned long long& nm, unsigned long long& att64,
                                                double a, unsigned long long grip64, const Thresholds& th) {
  att64 = grip64 & (att64 | nm);
  const bool att = __builtin_amdgcn_inverse_ballot_w64(att64);
  const double nf = clip_fast(f + a);
  nm = ws_near_mask_xy(nf, o, th);          // for the NEXT step: o is still the mug position before this step's move
  const double dd = att ? nf - f : -0.0;
  o = clip_fast(o + dd);
  f = nf;
}

// Third layout (NC == 3): x and y of an env in ADJACENT lanes (2e, 2e+1) of a 32-env wave.  The cross-lane traffic of the attach
// test is then a DPP quad permutation (v_mov_b32_dpp, an ordinary VALU slot) instead of v_permlane32_swap (20 cycles each),
// and -- as in ws_step_v -- every predicate stays in a VGPR lane mask (sign bits, and / bfi), so no instruction of the step
// touches an SGPR (8.3 cycles and one shared scalar pipe per CU): no ballot, no inverse ballot, no v_cndmask with a mask.
// 25 instructions per step in the ISA (31 for the lane-half layout): add max min | sub mul dpp dpp fma sub ashr dpp | or and |
// sub bfi and | add max min | cvt cvt bitop | address, ds_write2, ds_write.
//   quad_perm [0,0,2,2] (0xA0): both lanes of a pair read the EVEN (x) lane;  [1,1,3,3] (0xF5): both read the ODD (y) lane.
//   old = src in every update_dpp: all lanes are written, so no zero-initialised destination (saves a v_mov per DPP).
__device__ __forceinline__ double dpp_from_x(double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0xA0, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0xA0, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double dpp_swap_pair(double v) {   // quad_perm [1,0,3,2]
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0xB1, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0xB1, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}
// attach-test mask (all ones / zero, both lanes of the pair) of gripper coordinate f against mug coordinate o
__device__ __forceinline__ int ws_near_mask_adj(double f, double o, const WsConst& k) {
  const double d = f - o;
  const double d2 = fma(d, d, dpp_from_x(d * d));          // meaningful in the y lane: fma(dy, dy, dx*dx)
  const int m = __double2hiint(d2 - k.thr) >> 31;          // all ones iff d2 < thr (exact for non-NaN)
  return __builtin_amdgcn_update_dpp(m, m, 0xF5, 0xF, 0xF, false);
}
// fast step, software-pipelined like ws_step_xy_pipe: `nm` is the mask of THIS step's test, computed during the previous step
__device__ __forceinline__ void ws_step_adj(double& f, double& o, int& nm, int& attm, double a, int grip, const WsConst& k) {
  attm = grip & (attm | nm);
  const double nf = fmin(fmax(f + a, k.lo), k.hi);
  nm = ws_near_mask_adj(nf, o, k);                         // for the NEXT step: o is still the mug position before this step's move
  const double dd = nf - f;
  // attm ? dd : -0.0 on the two halves of the double (v_bfi_b32, v_and_b32)
  const double sd = __hiloint2double((attm & __double2hiint(dd)) | (~attm & k.nz), attm & __double2loint(dd));
  o = fmin(fmax(o + sd, k.lo), k.hi);
  f = nf;
}
// exact step (NaN actions / states: np.clip propagates NaN): both lanes of a pair gather the partner's coordinate and run the
// one-lane-per-env reference step redundantly; each keeps its own coordinate
__device__ __forceinline__ void ws_step_adj_exact(double& f, double& o, int& attm, double a, int grip, int c, const Thresholds& th) {
  const double pf = dpp_swap_pair(f), po = dpp_swap_pair(o), pa = dpp_swap_pair(a);
  double fx = c ? pf : f, fy = c ? f : pf, ox = c ? po : o, oy = c ? o : po, oxl = 0, oyl = 0;
  bool att = attm != 0;
  ws_step<true>(fx, fy, ox, oy, oxl, oyl, att, c ? pa : a, c ? a : pa, grip != 0, th);
  f = c ? fy : fx;
  o = c ? oy : ox;
  attm = att ? -1 : 0;
}

// PROF = diagnostic build only (tools/prof_ws.py): s_memtime stamps per role, summed per workgroup into
// g_ws_prof; never used by the shipped configuration and its timings are not quoted.
__device__ unsigned long long g_ws_prof[64 * 16];      // rows are indexed by the ENV block (`bid`, after the XCD-aware remap), so rows 0 .. wgs - 1 are episode group 0
                                                        // (indexed by the hardware blockIdx they mixed the groups)
__device__ __forceinline__ unsigned long long ws_clock() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
#define WS_STAMP(acc)                                  \
  if constexpr (PROF) {                                \
    const unsigned long long t_ = ws_clock();          \
    acc += t_ - p_x;                                   \
    p_x = t_;                                          \
  }

// RT reward type; NC compute waves: 1 = one lane per env (64 envs), 2 = two 32-env waves with x / y in the two lane
// halves; NL loader waves (steps of a chunk are dealt round-robin to them), NS storer waves (likewise), K steps per
// chunk, LEAD chunks per loader trip.  K % NL == 0.
// MULTI: several evaluation episodes per launch (WsArgs.episodes > 1; NC == 3 only) -- a separate instantiation, so the one-episode kernel
// carries none of the episode bookkeeping.
template <int RT, int NC, int NL, int NS, int K, int LEAD, bool PROF = false, bool MULTI = false, bool LDNT = false>
__global__ __launch_bounds__(64 * ((NC == 3 ? 2 : NC) + NL + NS)) void rollout_ws_kernel(const WsArgs a_in) {
  constexpr int E = 64;
  // several episodes in flight (WsArgs::ep_groups): this workgroup's group sees a launch of its OWN episodes -- shifted counter, rows, actions
  WsArgs a = a_in;
#ifdef EARL_WS_XCD_REMAP
  // XCD-aware block -> env-block map (MI355X: 8 XCDs, workgroups dealt round-robin; each XCD has its own 4 MiB L2)
  const int bid = (gridDim.x % 8 == 0) ? (int)((blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8) : (int)blockIdx.x;
#else
  const int bid = (int)blockIdx.x;
#endif
  int wg = bid;
  bool leaves_state = true;                  // (the group that walks the launch's last episode)
  int eps_before = 0;                        // episodes of the launch before this group's first
  if constexpr (MULTI) {
    if (a_in.ep_groups > 1) {
      const int grp = bid / a_in.wgs;
      wg = bid - grp * a_in.wgs;
      eps_before = grp * a_in.ep_per_group;
      const int ne = min(a_in.ep_per_group, a_in.episodes - eps_before);
      leaves_state = eps_before + ne == a_in.episodes;
      a.episodes = ne;
      a.T = ne * a_in.Tep;
      a.cfg.counter = a_in.cfg.counter + (uint64_t)eps_before * (uint64_t)(a_in.Tep + 1);
      const size_t row0 = (size_t)eps_before * (size_t)a_in.Tep * (size_t)a_in.n;
      a.act = a_in.act + (size_t)eps_before * (size_t)a_in.act_ep_stride;
      a.obs = a_in.obs + row0 * 12; a.reward = a_in.reward + row0; a.done = a_in.done + row0; a.success = a_in.success + row0;
    }
  }
  constexpr int NCW = NC == 3 ? 2 : NC;     // compute WAVES (NC == 3: two, x / y in adjacent lanes)
  constexpr int KL = K / NL;          // steps of a chunk handled by one loader
  static_assert(K % NL == 0, "K must be a multiple of NL");
  static_assert(!MULTI || NC == 3, "several episodes per launch: adjacent-lane compute waves only");
  static_assert(NC == 1 || NC == 2 || NC == 3, "NC is 1, 2 or 3");
  // rescaled (a0, a1).  NC == 1: [2e], [2e+1] of env e.  NC == 2: wave cw reads [64cw + lane]: a0 of its 32 envs, then a1
  __shared__ double A[3][K][2 * E];
  __shared__ uint8_t G[3][K][E];      // rescaled grip > 0
  __shared__ int slow_flag[3][NL];    // chunk contains a NaN action -> exact path (sticky in the compute wave)
  __shared__ float4 R[2][K][E * 3];   // row images: 64 obs rows of 48 B per step
  // loader-private staging: raw actions of SQ steps (coalesced order -> per env).  SQ = 1 for 8-step chunks keeps a
  // workgroup under 80 KiB of LDS, so two of them fit on a CU and overlap each other's pipeline fill and drain.
  constexpr int SQ = K >= 8 ? 1 : KL;                  // (four steps per round trip in the 16-step-chunk build: measured, no gain)
  __shared__ float S[NL][SQ][E * 3];

  // readfirstlane: tell hipcc the role index is wave-uniform (otherwise every role test becomes exec-mask code)
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int n = a.n, T = a.T;
  const int i0 = wg * E;
  const int i = i0 + lane;
  const int valid = min(E, n - i0);   // live envs of this workgroup
  const bool live = lane < valid;
  const int nch = (T + K - 1) / K;
  unsigned long long p_x = 0, p_t0 = 0;

  // ---- compute and storer threads: pre-fill the constant goal part of all row images (parts 1 and 2 of each row).  The
  // loaders skip this and issue their first loads at once (the launch's fixed cost is the chain prologue -> first loads
  // land -> two chunks processed -> first barrier).  A thread serves ONE env column (the stride is a multiple of 64), so the
  // goal row -- a Philox draw when the reset is fused -- is evaluated once per thread, not once per row.
  if (wave < NCW || wave >= NCW + NL) {
    const int tid = wave < NCW ? (int)threadIdx.x : (int)threadIdx.x - 64 * NL;
    const int e = tid % E;
    float g[6] = {0, 0, 0, 0, 0, 0};
    if (e < valid) load_goal<1>(a.goal_table, ws_goal_row(a, i0 + e), g);
    for (int rowi = tid / E; rowi < 2 * K; rowi += NCW + NS) {
      float4* row = &R[0][0][0] + (size_t)rowi * (E * 3) + e * 3;
      row[1] = float4{-1.0f, -1.0f, g[0], g[1]};
      row[2] = float4{g[2], g[3], g[4], g[5]};
    }
  }

  if (wave < NCW) {
   // the recurrence is the critical path: its wave wins issue arbitration against the loader/storer wave that shares
   // its SIMD (MI355X_MICROARCH.md, "two waves per SIMD": priority, then age)
   __builtin_amdgcn_s_setprio(3);
   if constexpr (NC == 3) {
    // ================================================================= COMPUTE, x / y in adjacent lanes, VGPR-only masks
    const int h = lane & 1, el = lane >> 1, e = wave * 32 + el;   // coordinate, env within the wave / workgroup
    const int ie = i0 + e;
    const bool alive = e < valid;
    double f = 0, o = 0;
    bool att0 = false;
    if (alive) {
      if (a.reset_first) {   // PersistentStateWrapper.reset + TabletopManipulation.reset, both lanes of the env alike
        Env<1> ev;
        const int gi = reset_env<1>(ev, a.cfg, a.cfg.counter, ie, a.goal_table, nullptr, a.th);
        f = h ? ev.q[1] : ev.q[0];         // (selects, not ev.q[h]: a dynamic index would put the array into scratch memory)
        o = h ? ev.q[3] : ev.q[2];
        if (h == 0 && leaves_state) {      // (episodes in flight: only the group of the launch's last episode touches the env state)
          a.goal_idx_w[ie] = gi;
          a.num_interventions[ie] += 1 + eps_before;
        }
      } else {
        f = a.qpos[(size_t)ie * 4 + h];
        o = a.qpos[(size_t)ie * 4 + 2 + h];
        att0 = a.attached[ie] >= 0;
      }
    }
    int attm = att0 ? -1 : 0;
    WsConst kc{-2.8, 2.8, a.th.grasp_d2, (int)0x80000000};
    asm volatile("" : "+v"(kc.lo), "+v"(kc.hi), "+v"(kc.thr), "+v"(kc.nz));   // keep them in VGPRs (a scalar operand costs 8.3 vs 5.3)
    int nm = ws_near_mask_adj(f, o, kc);                       // attach test of the first step (true positions)
    bool slow = __any(!(fabs(f) <= 1e300) || !(fabs(o) <= 2.8));   // NaN / object outside the arena -> exact path
    unsigned long long p_read = 0, p_comp = 0, p_bar = 0;
    if constexpr (PROF) p_t0 = ws_clock();
    __syncthreads();
    if constexpr (PROF) p_x = ws_clock();
    const unsigned long long p_first = p_x - p_t0;
    double av[K], nv[K];
    int gv[K], ng[K];           // grip as a lane mask: the loaders store 0xFF / 0x00, read back sign-extended
    int nslow = 0;
    auto fetch = [&](int c) {   // chunk c+1 is already published when chunk c starts: fetch it one chunk ahead
      const int ab = c % 3;
      nslow = 0;
#pragma unroll
      for (int w = 0; w < NL; ++w) nslow |= slow_flag[ab][w];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        nv[k] = A[ab][k][wave * 64 + lane];                    // [2e + h]: the loaders' double2 {a0, a1} of env e
        ng[k] = reinterpret_cast<const int8_t*>(&G[ab][k][0])[e];
      }
    };
    fetch(0);
    // several episodes per launch: episodes are whole numbers of GRANULES of GR = 8 steps (Tep % 8 == 0), a chunk is K / GR granules -- one for
    // the 8-step chunks, two for the 16-step ones, where an episode of e.g. 200 steps ends in the middle of every other chunk
    constexpr int GR = 8, GPC = K / GR;
    static_assert(!MULTI || (K % GR == 0 && GPC <= 2), "multi-episode launches: chunks of one or two 8-step granules");
    const int gpe = MULTI ? a.Tep / GR : 0x7fffffff;   // granules per episode
    int next_ep_g = gpe, ep = 0, last_gi = -1;
    // (wave-uniform) granule gidx starts the next evaluation episode of this launch: reset() of every env.  The draw is made at the head of the
    // chunk, where few registers are live (`pend`); a boundary at the chunk's second granule is then only a handful of moves inside the step loop.
    double f_pend = 0, o_pend = 0;
    bool pend = false;
    auto prepare_episode = [&](const int gidx) -> bool {
      if (MULTI && !(EARL_WSM_OFF & 4) && gidx == next_ep_g && gidx * GR < T) {    // (the last chunk of a launch may be half a chunk)
        ++ep;
        next_ep_g += gpe;
        Env<1> ev;
        last_gi = reset_env<1>(ev, a.cfg, ws_ep_counter(a, ep), ie, a.goal_table, nullptr, a.th);
        f_pend = h ? ev.q[1] : ev.q[0];
        o_pend = h ? ev.q[3] : ev.q[2];
        return true;
      }
      return false;
    };
    auto apply_episode = [&]() {
      f = f_pend; o = o_pend;
      attm = 0;
      nm = ws_near_mask_adj(f, o, kc);
    };
    for (int c = 0; c < nch; ++c) {
      const int rb = c & 1;
      if (prepare_episode(c * (MULTI ? GPC : 1))) apply_episode();
      if constexpr (MULTI && GPC == 2) pend = prepare_episode(c * GPC + 1);
      slow = slow || (nslow != 0);
#pragma unroll
      for (int k = 0; k < K; ++k) { av[k] = nv[k]; gv[k] = ng[k]; }
      if (c + 1 < nch) fetch(c + 1);
      if constexpr (PROF) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      WS_STAMP(p_read)
      auto run = [&](auto exact, auto full) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          if constexpr (!decltype(full)::value)
            if (c * K + k >= T) break;      // tail chunk only (wave-uniform)
          if constexpr (MULTI && GPC == 2) { if (k == GR && pend) apply_episode(); }      // an episode may start at the chunk's second granule
          if constexpr (decltype(exact)::value) ws_step_adj_exact(f, o, attm, av[k], gv[k], h, a.th);
          else ws_step_adj(f, o, nm, attm, av[k], gv[k], kc);
          float* rowf = reinterpret_cast<float*>(&R[rb][k][e * 3]);   // (fx, fy, ox, oy), (flag, flag, ..)
          rowf[h] = (float)f;
          rowf[2 + h] = (float)o;
          rowf[4 + h] = __int_as_float(~attm & (int)0xBF800000);     // attached ? 0.0f : -1.0f
        }
      };
      const bool whole = (c + 1) * K <= T;   // every step of the chunk exists: no per-step tail test
      if (__builtin_amdgcn_readfirstlane(slow ? 1 : 0)) {
        if (whole) run(std::true_type{}, std::true_type{}); else run(std::true_type{}, std::false_type{});
      } else {
        if (whole) run(std::false_type{}, std::true_type{}); else run(std::false_type{}, std::false_type{});
      }
      if constexpr (PROF) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      WS_STAMP(p_comp)
      __syncthreads();
      WS_STAMP(p_bar)
    }
    if constexpr (PROF) {
      if (lane == 0 && bid < 64 && wave == 0) {
        unsigned long long* op = g_ws_prof + bid * 16;
        op[0] = p_first; op[1] = p_read; op[2] = p_comp; op[3] = p_bar; op[4] = p_x - p_t0;
      }
      if (lane == 0 && bid < 64 && wave == 1) g_ws_prof[bid * 16 + 13] = p_bar;      // the second compute wave's barrier wait
    }
    if (alive && leaves_state) {
      a.qpos[(size_t)ie * 4 + h] = f;
      a.qpos[(size_t)ie * 4 + 2 + h] = o;
      if (h == 0) {
        a.attached[ie] = attm ? 0 : -1;
        a.steps_since_reset[ie] = (a.reset_first ? 0 : a.steps_since_reset[ie]) + (MULTI ? a.Tep : T);
        if (MULTI && ep > 0) {     // the later episodes of the launch: their resets counted, the last goal kept
          a.goal_idx_w[ie] = last_gi;
          a.num_interventions[ie] += ep;
        }
      }
    }
   } else if constexpr (NC == 2) {
    // ================================================================= COMPUTE, x / y in the two lane halves
    const int h = lane >> 5, el = lane & 31, e = wave * 32 + el;   // coordinate, env within the wave / workgroup
    const int ie = i0 + e;
    const bool alive = e < valid;
    double f = 0, o = 0;
    bool att0 = false;
    if (alive) {
      if (a.reset_first) {   // PersistentStateWrapper.reset + TabletopManipulation.reset, both lanes of the env alike
        Env<1> ev;
        const int gi = reset_env<1>(ev, a.cfg, a.cfg.counter, ie, a.goal_table, nullptr, a.th);
        f = h ? ev.q[1] : ev.q[0];         // (selects: a dynamic index puts the array into scratch memory)
        o = h ? ev.q[3] : ev.q[2];
        if (h == 0) {
          a.goal_idx_w[ie] = gi;
          a.num_interventions[ie] += 1;
        }
      } else {
        f = a.qpos[(size_t)ie * 4 + h];
        o = a.qpos[(size_t)ie * 4 + 2 + h];
        att0 = a.attached[ie] >= 0;
      }
    }
    double ol = o;
    unsigned long long att64 = __ballot(att0);
    unsigned long long nm = ws_near_mask_xy(f, o, a.th);   // attach test of the first step (true positions)
    bool slow = __any(!(fabs(f) <= 1e300) || !(fabs(o) <= 2.8));   // NaN / object outside the arena -> exact path
    unsigned long long p_read = 0, p_comp = 0, p_bar = 0;
    if constexpr (PROF) p_t0 = ws_clock();
    __syncthreads();
    if constexpr (PROF) p_x = ws_clock();
    const unsigned long long p_first = p_x - p_t0;
    double av[K], nv[K];
    uint8_t gv[K], ng[K];
    int nslow = 0;
    auto fetch = [&](int c) {   // chunk c+1 is already published when chunk c starts: fetch it one chunk ahead
      const int ab = c % 3;
      nslow = 0;
#pragma unroll
      for (int w = 0; w < NL; ++w) nslow |= slow_flag[ab][w];
#pragma unroll
      for (int k = 0; k < K; ++k) { nv[k] = A[ab][k][wave * 64 + lane]; ng[k] = G[ab][k][e]; }
    };
    fetch(0);
    for (int c = 0; c < nch; ++c) {
      const int rb = c & 1;
      slow = slow || (nslow != 0);
#pragma unroll
      for (int k = 0; k < K; ++k) { av[k] = nv[k]; gv[k] = ng[k]; }
      if (c + 1 < nch) fetch(c + 1);
      if constexpr (PROF) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      WS_STAMP(p_read)
      auto run = [&](auto exact, auto full) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          if constexpr (!decltype(full)::value)
            if (c * K + k >= T) break;      // tail chunk only (wave-uniform)
          if constexpr (decltype(exact)::value) ws_step_xy<true>(f, o, ol, att64, av[k], __ballot(gv[k] != 0), a.th);
          else ws_step_xy_pipe(f, o, nm, att64, av[k], __ballot(gv[k] != 0), a.th);
          float* rowf = reinterpret_cast<float*>(&R[rb][k][e * 3]);   // (fx, fy, ox, oy), (flag, flag, ..)
          rowf[h] = (float)f;
          rowf[2 + h] = (float)o;
          rowf[4 + h] = __builtin_amdgcn_inverse_ballot_w64(att64) ? 0.0f : -1.0f;
        }
      };
      const bool whole = (c + 1) * K <= T;   // every step of the chunk exists: no per-step tail test
      if (__builtin_amdgcn_readfirstlane(slow ? 1 : 0)) {
        if (whole) run(std::true_type{}, std::true_type{}); else run(std::true_type{}, std::false_type{});
      } else {
        if (whole) run(std::false_type{}, std::true_type{}); else run(std::false_type{}, std::false_type{});
      }
      if constexpr (PROF) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      WS_STAMP(p_comp)
      __syncthreads();
      WS_STAMP(p_bar)
    }
    if constexpr (PROF) {
      if (lane == 0 && bid < 64 && wave == 0) {
        unsigned long long* op = g_ws_prof + bid * 16;
        op[0] = p_first; op[1] = p_read; op[2] = p_comp; op[3] = p_bar; op[4] = p_x - p_t0;
      }
    }
    if (alive) {
      a.qpos[(size_t)ie * 4 + h] = f;
      a.qpos[(size_t)ie * 4 + 2 + h] = o;
      if (h == 0) {
        a.attached[ie] = (att64 >> lane) & 1 ? 0 : -1;
        a.steps_since_reset[ie] = (a.reset_first ? 0 : a.steps_since_reset[ie]) + T;
      }
    }
   } else {
    // ================================================================= COMPUTE
    double fx = 0, fy = 0, ox = 0, oy = 0;
    bool att = false;
    if (live) {
      if (a.reset_first) {
        Env<1> ev;
        const int gi = reset_env<1>(ev, a.cfg, a.cfg.counter, i, a.goal_table, nullptr, a.th);
        fx = ev.q[0]; fy = ev.q[1]; ox = ev.q[2]; oy = ev.q[3];
        a.goal_idx_w[i] = gi;
        a.num_interventions[i] += 1;
      } else {
        const double2* q2 = reinterpret_cast<const double2*>(a.qpos + (size_t)i * 4);
        const double2 u = q2[0], v = q2[1];
        fx = u.x; fy = u.y; ox = v.x; oy = v.y;
        att = a.attached[i] >= 0;
      }
    }
    double oxl = ox, oyl = oy;
    int attm = att ? -1 : 0;            // lane mask form of `att` for the VGPR-only fast step
    WsConst kc{-2.8, 2.8, a.th.grasp_d2, (int)0x80000000};
    asm volatile("" : "+v"(kc.lo), "+v"(kc.hi), "+v"(kc.thr), "+v"(kc.nz));   // keep them in VGPRs (a scalar operand costs 8.3 vs 5.3)
    // exact path if the state holds a NaN or an object outside the arena (sticky, wave-uniform)
    bool slow = __any(!(fabs(fx) <= 1e300) || !(fabs(fy) <= 1e300) || !(fabs(ox) <= 2.8) || !(fabs(oy) <= 2.8));
    unsigned long long p_read = 0, p_comp = 0, p_bar = 0;
    if constexpr (PROF) p_t0 = ws_clock();
    __syncthreads();
    if constexpr (PROF) p_x = ws_clock();
    const unsigned long long p_first = p_x - p_t0;
    // Chunk c+1 is already published when chunk c starts (the loaders run one barrier interval ahead and the ring
    // is 3 deep), so its actions are fetched from LDS while chunk c is being computed: no LDS latency per chunk.
    double2 av[K], nv[K];
    int gv[K], ng[K];     // grip as a lane mask: the loaders store 0xFF / 0x00, read back sign-extended
    int nslow = 0;
    auto fetch = [&](int c) {
      const int ab = c % 3;
      nslow = 0;
#pragma unroll
      for (int w = 0; w < NL; ++w) nslow |= slow_flag[ab][w];
#pragma unroll
      for (int k = 0; k < K; ++k) {
        nv[k] = reinterpret_cast<const double2*>(&A[ab][k][0])[lane];
        ng[k] = reinterpret_cast<const int8_t*>(&G[ab][k][0])[lane];
      }
    };
    fetch(0);
    for (int c = 0; c < nch; ++c) {
      const int rb = c & 1;
      slow = slow || (nslow != 0);
#pragma unroll
      for (int k = 0; k < K; ++k) { av[k] = nv[k]; gv[k] = ng[k]; }
      if (c + 1 < nch) fetch(c + 1);
      if constexpr (PROF) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      WS_STAMP(p_read)
      auto run = [&](auto exact, auto full) {
#pragma unroll
        for (int k = 0; k < K; ++k) {
          if constexpr (!decltype(full)::value)
            if (c * K + k >= T) break;      // tail chunk only (wave-uniform)
          if constexpr (decltype(exact)::value) {
            bool attb = attm != 0;
            ws_step<true>(fx, fy, ox, oy, oxl, oyl, attb, av[k].x, av[k].y, gv[k] != 0, a.th);
            attm = attb ? -1 : 0;
          } else {
            ws_step_v(fx, fy, ox, oy, oxl, oyl, attm, av[k].x, av[k].y, gv[k], kc);
          }
          float4* row = &R[rb][k][lane * 3];
          row[0] = float4{(float)fx, (float)fy, (float)ox, (float)oy};
          const float flag = __int_as_float(~attm & (int)0xBF800000);   // attached ? 0.0f : -1.0f
          *reinterpret_cast<float2*>(&row[1]) = float2{flag, flag};
        }
      };
      const bool whole = (c + 1) * K <= T;   // every step of the chunk exists: no per-step tail test
      if (__builtin_amdgcn_readfirstlane(slow ? 1 : 0)) {
        if (whole) run(std::true_type{}, std::true_type{}); else run(std::true_type{}, std::false_type{});
      } else {
        if (whole) run(std::false_type{}, std::true_type{}); else run(std::false_type{}, std::false_type{});
      }
      if constexpr (PROF) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      WS_STAMP(p_comp)
      __syncthreads();
      WS_STAMP(p_bar)
    }
    if constexpr (PROF) {
      if (lane == 0 && bid < 64) {
        unsigned long long* o = g_ws_prof + bid * 16;
        o[0] = p_first; o[1] = p_read; o[2] = p_comp; o[3] = p_bar; o[4] = p_x - p_t0;
      }
    }
    if (live) {
      double2* q2 = reinterpret_cast<double2*>(a.qpos + (size_t)i * 4);
      q2[0] = double2{fx, fy};
      q2[1] = double2{ox, oy};
      a.attached[i] = attm ? 0 : -1;
      a.steps_since_reset[i] = (a.reset_first ? 0 : a.steps_since_reset[i]) + T;
    }
   }
  } else if (wave < NCW + NL) {
    // ================================================================= LOADERS
    // The actions of one step for this workgroup are 192 consecutive floats.  Lane l loads floats l, l+64, l+128
    // (three coalesced 256-B wave loads, each into its own VGPR).  A loader owns the steps k = w, w+NL, .. of every
    // chunk and keeps TWO trips (LEAD chunks each) of raw actions in registers: at the start of a trip it first
    // touches the trip loaded one whole trip ago (long since landed), then issues the loads of the next one.
    // Loads are UNCONDITIONAL with clamped indices: a load under a branch makes hipcc wait vmcnt(0) right behind
    // it.  Clamped duplicates are never consumed (compute stops at T, storers mask dead lanes).
    const int w = wave - NCW;
    float rawA[LEAD][KL][3], rawB[LEAD][KL][3];
    const int last = max(valid * 3 - 1, 0);
    const int e0 = min(lane, last), e1 = min(lane + 64, last), e2 = min(lane + 128, last);
    const float* const base = a.act + (size_t)i0 * 3;
    const size_t step_stride = (size_t)n * 3;
    constexpr int GR = 8, GPC = MULTI ? K / GR : 1;      // (granules: see the compute waves)
    const int gpe = MULTI ? a.Tep / GR : 0x7fffffff;
    int l_ep_end = gpe, l_ep_first = 0;       // loader's episode cursor: GRANULES [l_ep_first, l_ep_end) belong to the current episode
    size_t l_ep_base = 0;                     // ... whose actions start at base + l_ep_base
    auto issue_trip = [&](float (&raw)[LEAD][KL][3], int r) {   // raw <- my steps of trip r
      // MULTI: chunk j of the launch = chunk j - l_ep_first of the episode the cursor stands in (wave-uniform scalars).  The row offsets of the whole
      // trip are worked out FIRST, branch-free (chunks come one after the other: at most one episode boundary per chunk); then the loads follow
      // back to back.  A loop or a branch between the loads of a trip makes the compiler drain them there, which put the loaders on the
      // critical path of the multi-episode kernel (118 ns per step against the single-episode kernel's 101).
      // (one 64-bit row pointer per chunk, then this loader's steps of the chunk by a constant stride: the loaders are scalar-ALU bound here)
      const float* rowp[LEAD][GPC];
      size_t off[LEAD][KL];
#pragma unroll
      for (int d = 0; d < LEAD; ++d) {
        if constexpr (MULTI && !(EARL_WSM_OFF & 1)) {
          const int j = min(r * LEAD + d, nch - 1);
#pragma unroll
          for (int hg = 0; hg < GPC; ++hg) {            // the granules of chunk j, one after the other: at most one episode boundary each
            const int gidx = min(j * GPC + hg, T / GR - 1);      // (the last chunk of a launch may be half a chunk: its second granule repeats the first)
            const int xm = -(int)(gidx >= l_ep_end);
            l_ep_end += gpe & xm; l_ep_first += gpe & xm;
            l_ep_base += (size_t)a.act_ep_stride & (size_t)(long long)xm;
            rowp[d][hg] = base + l_ep_base + (size_t)__builtin_amdgcn_readfirstlane((gidx - l_ep_first) * GR + w) * step_stride;
          }
        } else {
#pragma unroll
          for (int q = 0; q < KL; ++q)
            off[d][q] = (size_t)__builtin_amdgcn_readfirstlane(min((r * LEAD + d) * K + q * NL + w, (MULTI ? a.Tep : T) - 1)) * step_stride;
        }
      }
      const size_t qstride = (size_t)NL * step_stride;
      static_assert(!MULTI || GR % NL == 0, "a loader's steps of a granule: a whole number");
      constexpr int QG = MULTI ? GR / NL : KL;           // this loader's steps per granule (step q NL + w of the chunk lies in granule q / QG)
#pragma unroll
      for (int d = 0; d < LEAD; ++d) {
#pragma unroll
        for (int q = 0; q < KL; ++q) {
          const float* p;
          if constexpr (MULTI && !(EARL_WSM_OFF & 1)) p = rowp[d][q / QG] + (size_t)(q % QG) * qstride;
          else p = base + off[d][q];
#ifdef EARL_WS_LD_AUX
          if constexpr (true) {
            const __amdgpu_buffer_rsrc_t rs = ws_rsrc(p);
            raw[d][q][0] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, e0 * 4, 0, EARL_WS_LD_AUX));
            raw[d][q][1] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, e1 * 4, 0, EARL_WS_LD_AUX));
            raw[d][q][2] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, e2 * 4, 0, EARL_WS_LD_AUX));
          } else
#endif
          if constexpr (LDNT) {        // (experiment) non-temporal loads: the actions are read once
            raw[d][q][0] = __builtin_nontemporal_load(p + e0); raw[d][q][1] = __builtin_nontemporal_load(p + e1); raw[d][q][2] = __builtin_nontemporal_load(p + e2);
          } else {
            raw[d][q][0] = p[e0]; raw[d][q][1] = p[e1]; raw[d][q][2] = p[e2];
          }
        }
      }
    };
    unsigned long long p_proc = 0, p_bar = 0;
    auto process = [&](const float (&raw)[KL][3], int j) {  // transpose through LDS, rescale, publish chunk j
      const int ab = j % 3;
      bool any_nan = false;
#pragma unroll
      for (int q0 = 0; q0 < KL; q0 += SQ) {
#pragma unroll
        for (int q = q0; q < q0 + SQ; ++q) {
          float* sk = &S[w][q - q0][0];
          sk[lane] = raw[q][0]; sk[lane + 64] = raw[q][1]; sk[lane + 128] = raw[q][2];
        }
        wave_lds_fence();
#pragma unroll
        for (int q = q0; q < q0 + SQ; ++q) {
          const float* sk = &S[w][q - q0][0];   // same wave wrote it: LDS operations of one wave execute in order
          const float r0 = sk[lane * 3], r1 = sk[lane * 3 + 1], r2 = sk[lane * 3 + 2];
          const double a0 = rescale_action(r0), a1 = rescale_action(r1);
          any_nan = any_nan || (r0 != r0) || (r1 != r1);
          if (j < nch) {
            if constexpr (NC == 2) {
              A[ab][q * NL + w][(lane >> 5) * 64 + (lane & 31)] = a0;
              A[ab][q * NL + w][(lane >> 5) * 64 + 32 + (lane & 31)] = a1;
            } else {
              reinterpret_cast<double2*>(&A[ab][q * NL + w][0])[lane] = double2{a0, a1};
            }
            G[ab][q * NL + w][lane] = (uint8_t)(r2 >= a.grip_x ? 0xFF : 0);   // == rescale_action(r2) > 0 (NaN -> release)
          }
        }
        wave_lds_fence();   // the next staging writes must stay behind these reads
      }
      const bool wave_nan = __any(any_nan);
      if (lane == 0 && j < nch) slow_flag[ab][w] = wave_nan ? 1 : 0;
    };
    auto run_trip = [&](float (&cur)[LEAD][KL][3], float (&nxt)[LEAD][KL][3], int r) {
#pragma unroll
      for (int d = 0; d < LEAD; ++d) {
        const int j = r * LEAD + d;
        process(cur[d], j);
        if (d == 0) {                       // `cur` has been touched (its wait is behind us): now prefetch trip r+1
          __builtin_amdgcn_sched_barrier(0);
          issue_trip(nxt, r + 1);
          __builtin_amdgcn_sched_barrier(0);
        }
        WS_STAMP(p_proc)
        if (j >= 1 && j < nch + 2) __syncthreads();
        WS_STAMP(p_bar)
      }
    };
    if constexpr (PROF) p_t0 = p_x = ws_clock();
    issue_trip(rawA, 0);
    for (int r = 0; r * LEAD < nch + 2; r += 2) {
      run_trip(rawA, rawB, r);
      if ((r + 1) * LEAD < nch + 2) run_trip(rawB, rawA, r + 1);
    }
    if constexpr (PROF) {
      if (lane == 0 && bid < 64 && w == 0) {
        unsigned long long* o = g_ws_prof + bid * 16;
        o[5] = p_proc; o[6] = p_bar; o[7] = p_x - p_t0;
      }
      if (lane == 0 && bid < 64 && w == NL - 1 && NL > 1) {      // the last loader wave too
        unsigned long long* o = g_ws_prof + bid * 16;
        o[14] = p_proc; o[15] = p_bar;
      }
    }
  } else {
    // ================================================================= STORERS
    const int s = wave - NCW - NL;
    float g[6] = {0, 0, 0, 0, 0, 0};
    int t_done = 0x7fffffff;           // done fires from step index t_done on (steps_since_reset + t + 1 >= horizon)
    if (live) {
      load_goal<1>(a.goal_table, ws_goal_row(a, i), g);
      const long long td = (long long)a.horizon - 1 - (a.reset_first ? 0 : a.steps_since_reset[i]);
      t_done = td < 0 ? 0 : (td > 0x7fffffff ? 0x7fffffff : (int)td);
    }
    // several episodes per launch: chunk c belongs to episode c / cpe; `g`, and the step index `done` counts from, follow it (cursors,
    // no divisions: eight storer waves share the SIMDs with the compute waves)
    // (in GRANULES of 8 steps, see the compute waves: a 16-step chunk is two of them, and an episode may end between the two)
    constexpr int GR = 8, GPC = MULTI ? K / GR : 1;
    const int gpe = MULTI ? a.Tep / GR : 0x7fffffff;
    int g_ep = 0, t_ep0 = 0, g_end = gpe;      // episode `g` belongs to, its first global step, its end (granule index)
    float gn[6] = {0, 0, 0, 0, 0, 0};          // goal of the episode AFTER that one: loaded once per episode, an episode ahead of its use, at the head
    bool gn_stale = false;                     // of a chunk's stores (reload_next_goal)
    if (MULTI && live) load_goal<1>(a.goal_table, ws_goal_row_ep(a, i, 1), gn);
    auto reload_next_goal = [&]() {             // (uniform, once per episode) before any store of the chunk is issued
      if (MULTI && gn_stale) {
        gn_stale = false;
        if (live) load_goal<1>(a.goal_table, ws_goal_row_ep(a, i, g_ep + 1), gn);
      }
    };
    auto enter_episode = [&](int gidx) {        // (uniform) called before the steps of granule gidx are stored; gidx advances by one
      if (MULTI && !(EARL_WSM_OFF & 2) && gidx >= g_end) {
        // the next episode's goal is ALREADY in registers (gn): no global load between this chunk's stores -- vmcnt counts loads and stores alike, and a
        // load under this rare branch made the compiler drain the wave's stores in every chunk
        g_ep += 1; g_end += gpe; t_ep0 += a.Tep;
#pragma unroll
        for (int k = 0; k < 6; ++k) g[k] = gn[k];
        gn_stale = true;
      }
    };
    // ... and after the rows of granule gidx have been stored, their row images are next written for granule gidx + 2 GPC (same buffer, two
    // chunks on): if that one belongs to a later episode (an episode boundary b with gidx < b <= gidx + 2 GPC; g_end is the first boundary > gidx
    // after enter_episode(gidx); episodes are longer than that window) the constant goal part of those rows is rewritten (this storer: its own
    // steps of the granule, env column `lane`)
    auto refill_goal = [&](int gidx) {
      if (MULTI && !(EARL_WSM_OFF & 2) && g_end <= gidx + 2 * GPC && (gidx / GPC) + 2 < nch) {
        reload_next_goal();                      // (episodes of a few granules only: the boundary ahead can be the one after an episode that has just begun)
        const int k0 = (gidx % GPC) * GR;       // first step of this granule within its chunk
        for (int k = k0 + (s % GR); k < k0 + GR; k += NS) {
          if ((k % NS) != s) continue;
          float4* row = &R[(gidx / GPC) & 1][k][lane * 3];
          row[1] = float4{-1.0f, -1.0f, gn[0], gn[1]};    // (the flag words are rewritten by the compute wave every step)
          row[2] = float4{gn[2], gn[3], gn[4], gn[5]};
        }
      }
    };
    // steps_since_reset must be in a register before the compute wave can possibly update it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const bool full = (valid == E) && ((n & 3) == 0);   // whole workgroup live and flag rows dword-aligned
    unsigned long long p_st = 0, p_bar = 0;
    auto store_chunk = [&](int c) {
      const int rb = c & 1;
#pragma unroll
      for (int k = 0; k < K; ++k) {
        if constexpr (MULTI) {             // granule boundaries of the chunk: the previous granule's rows are done with, the next one's goal comes in
          if (k % GR == 0) {
            if (k > 0) refill_goal(c * GPC + k / GR - 1);
            enter_episode(c * GPC + k / GR);
          }
        }
        if ((k % NS) != s) continue;     // storers interleave over the steps of a chunk
        const int t = c * K + k;
        if (t >= T) continue;
        const size_t row0 = (size_t)t * n + i0;
        float4* dst = reinterpret_cast<float4*>(a.obs + row0 * 12);
        const float4* src = &R[rb][k][0];
        // reward / success / done of env `lane` need the coordinates of its own row
        const float4 p = src[lane * 3];
        const float o[12] = {p.x, p.y, p.z, p.w, 0.f, 0.f, g[0], g[1], g[2], g[3], g[4], g[5]};
        const bool succ = success1(o, a.wide, a.th);
        float rew;
        if constexpr (RT == EARL_REWARD_SPARSE) rew = succ ? 1.0f : 0.0f;
        else rew = (float)dense1(o);
        const bool dn = t - t_ep0 >= t_done;
        if (full) {
          // (1) the 64 obs rows of this step: 192 float4, contiguous in HBM and in LDS
          const float4 v0 = src[lane], v1 = src[lane + 64], v2 = src[lane + 128];
          dst[lane] = v0; dst[lane + 64] = v1; dst[lane + 128] = v2;
          a.reward[row0 + lane] = rew;
          // (2) 64 flag bytes of a step = 16 dwords: expand ballot nibbles to bytes
          const unsigned long long ms = __ballot(succ), md = __ballot(dn);
          if (lane < 16) {
            const uint32_t ns = (uint32_t)(ms >> (4 * lane)) & 0xFu, nd = (uint32_t)(md >> (4 * lane)) & 0xFu;
            reinterpret_cast<uint32_t*>(a.success + row0)[lane] = (ns * 0x00204081u) & 0x01010101u;
            reinterpret_cast<uint32_t*>(a.done + row0)[lane] = (nd * 0x00204081u) & 0x01010101u;
          }
        } else {
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            const int idx = lane + 64 * m;
            if (idx < valid * 3) dst[idx] = src[idx];
          }
          if (live) {
            a.reward[row0 + lane] = rew;
            a.success[row0 + lane] = succ;
            a.done[row0 + lane] = dn;
          }
        }
      }
      if constexpr (MULTI) refill_goal(c * GPC + GPC - 1);
    };
    // Fast form for whole chunks of a full workgroup (K % NS == 0): this storer owns steps s, s+NS, .. of the chunk.
    // All LDS reads of those steps are issued first (one round trip instead of two per step), then the HBM stores and
    // the success arithmetic run on registers.
    constexpr int Q = (K % NS == 0) ? K / NS : 0;
    auto store_chunk_fast = [&](int c) {
      const int rb = c & 1;
      struct Rows { float4 v0, v1, v2, p; };              // the row image of one step: by value, never an array (arrays of them ended up in scratch
                                                          // memory once the episode hooks stood between their definition and their use)
      auto load_rows = [&](const int q) -> Rows {
        const float4* src = &R[rb][s + NS * q][0];
        return Rows{src[lane], src[lane + 64], src[lane + 128], src[lane * 3]};
      };
      const Rows r0 = load_rows(0), r1 = load_rows(Q > 1 ? 1 : 0), r2 = load_rows(Q > 2 ? 2 : 0), r3 = load_rows(Q > 3 ? 3 : 0);
      auto step_q = [&](auto qc, const Rows rw) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value;
        if constexpr (MULTI) enter_episode(c * GPC + q);      // (NS == GR: this storer's q-th step lies in the chunk's q-th granule)
        const int t = c * K + s + NS * q;
        const size_t row0 = (size_t)t * n + i0;
        float4* dst = reinterpret_cast<float4*>(a.obs + row0 * 12);
        // NON-TEMPORAL stores: the observation rows (48 of the 54 bytes written per env step) are written once and never read back by the kernel;
        // as ordinary stores they went through the L2 write-back path and, with four episodes in flight, held the kernel at 5.1-5.7 TB/s;
        // streamed past it the launch runs at 7.3 TB/s algorithmic = 6.3 TB/s at the HBM interface (the actions hit the cache), which is what
        // MI355X_MICROARCH.md gives as the achievable HBM bandwidth.  (EARL_WS_TEMPORAL_STORES: the ordinary stores, for comparison.)
#ifdef EARL_WS_NO_STORES      // measurement only (WRONG results): the storers keep their LDS reads and arithmetic but write one row in 64 -- what the launch costs without its stores
        if ((t & 63) == 0) { dst[lane] = rw.v0; dst[lane + 64] = rw.v1; dst[lane + 128] = rw.v2; }
#elif defined(EARL_WS_ST_AUX)
        {
          const __amdgpu_buffer_rsrc_t rs = ws_rsrc(dst);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ws_v4i, ws_v4f{rw.v0.x, rw.v0.y, rw.v0.z, rw.v0.w}), rs, lane * 16, 0, EARL_WS_ST_AUX);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ws_v4i, ws_v4f{rw.v1.x, rw.v1.y, rw.v1.z, rw.v1.w}), rs, (lane + 64) * 16, 0, EARL_WS_ST_AUX);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(ws_v4i, ws_v4f{rw.v2.x, rw.v2.y, rw.v2.z, rw.v2.w}), rs, (lane + 128) * 16, 0, EARL_WS_ST_AUX);
        }
#elif !defined(EARL_WS_TEMPORAL_STORES)
        typedef float nt4 __attribute__((ext_vector_type(4)));
        nt4* d4 = reinterpret_cast<nt4*>(dst);
        __builtin_nontemporal_store(nt4{rw.v0.x, rw.v0.y, rw.v0.z, rw.v0.w}, d4 + lane);
        __builtin_nontemporal_store(nt4{rw.v1.x, rw.v1.y, rw.v1.z, rw.v1.w}, d4 + lane + 64);
        __builtin_nontemporal_store(nt4{rw.v2.x, rw.v2.y, rw.v2.z, rw.v2.w}, d4 + lane + 128);
#else
        dst[lane] = rw.v0; dst[lane + 64] = rw.v1; dst[lane + 128] = rw.v2;
#endif
        const float o[12] = {rw.p.x, rw.p.y, rw.p.z, rw.p.w, 0.f, 0.f, g[0], g[1], g[2], g[3], g[4], g[5]};
        const bool succ = success1(o, a.wide, a.th);
        float rew;
        if constexpr (RT == EARL_REWARD_SPARSE) rew = succ ? 1.0f : 0.0f;
        else rew = (float)dense1(o);
#ifdef EARL_WS_NO_STORES
        if ((t & 63) == 0) a.reward[row0 + lane] = rew;
#elif defined(EARL_WS_SMALL_NT)
        __builtin_nontemporal_store(rew, a.reward + row0 + lane);
#else
        a.reward[row0 + lane] = rew;
#endif
        const unsigned long long ms = __ballot(succ), md = __ballot(t - t_ep0 >= t_done);
#ifdef EARL_WS_NO_STORES
        if (lane < 16 && (t & 63) == 0) {
#else
        if (lane < 16) {
#endif
          const uint32_t ns = (uint32_t)(ms >> (4 * lane)) & 0xFu, nd = (uint32_t)(md >> (4 * lane)) & 0xFu;
#ifdef EARL_WS_SMALL_NT
          __builtin_nontemporal_store((ns * 0x00204081u) & 0x01010101u, reinterpret_cast<uint32_t*>(a.success + row0) + lane);
          __builtin_nontemporal_store((nd * 0x00204081u) & 0x01010101u, reinterpret_cast<uint32_t*>(a.done + row0) + lane);
#else
          reinterpret_cast<uint32_t*>(a.success + row0)[lane] = (ns * 0x00204081u) & 0x01010101u;
          reinterpret_cast<uint32_t*>(a.done + row0)[lane] = (nd * 0x00204081u) & 0x01010101u;
#endif
        }
        if constexpr (MULTI) refill_goal(c * GPC + q);         // (every row image of the chunk is in registers by now)
      };
      static_assert(Q <= 4, "steps of a chunk per storer");
      if constexpr (Q > 0) step_q(std::integral_constant<int, 0>{}, r0);
      if constexpr (Q > 1) step_q(std::integral_constant<int, 1>{}, r1);
      if constexpr (Q > 2) step_q(std::integral_constant<int, 2>{}, r2);
      if constexpr (Q > 3) step_q(std::integral_constant<int, 3>{}, r3);
    };
    auto store = [&](int c) {
      reload_next_goal();
      if (Q > 0 && (!MULTI || NS == GR) && full && (c + 1) * K <= T) store_chunk_fast(c);
      else store_chunk(c);
    };
    if constexpr (PROF) p_t0 = ws_clock();
    __syncthreads();
    if constexpr (PROF) p_x = ws_clock();
    for (int c = 0; c < nch; ++c) {
      if (c >= 1) store(c - 1);
      WS_STAMP(p_st)
      __syncthreads();
      WS_STAMP(p_bar)
    }
    store(nch - 1);
    if constexpr (PROF) {
      if (lane == 0 && bid < 64 && s == 0) {
        unsigned long long* o = g_ws_prof + bid * 16;
        o[8] = p_st; o[9] = p_bar; o[10] = ws_clock() - p_t0;
      }
      if (lane == 0 && bid == 0 && gridDim.x <= 32) {             // (small grids: every storer wave of workgroup 0, rows 32 .. 32 + NS)
        unsigned long long* o = g_ws_prof + (32 + s) * 16;
        o[8] = p_st; o[9] = p_bar;
      }
      if (lane == 0 && bid < 64 && s == NS - 1 && NS > 1) {      // the last storer wave too
        unsigned long long* o = g_ws_prof + bid * 16;
        o[11] = p_st; o[12] = p_bar;
      }
    }
  }
}
#undef WS_STAMP

}  // namespace earl
