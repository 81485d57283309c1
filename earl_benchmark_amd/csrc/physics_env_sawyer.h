// physics_env_sawyer.h -- the Sawyer door / peg env kernels: fused rollout (optionally time-sliced), reset / observe, reward and info kernels
// A section of csrc/physics.hip (included there, inside its anonymous namespace, after the stepper): split out in round 5 so that a change to one env's kernels
// recompiles only the translation units that hold them (csrc/Makefile lists the headers per unit).

// ------------------------------------------------------------------------------------------------ Sawyer env kernels
struct SawyerArgs {
  const void* m;
  const earl_collision_model* col;
  earl_sawyer_cfg cfg;
  earl_sawyer_state st;
  const float* action; int T;
  earl_sawyer_out out;
  const double* reset_qpos; const double* reset_qvel; const uint8_t* mask; double* reset_obs;
  int observe_only;
  int slice;                     // SLICED rollout: env steps per work item (0: one item = the whole rollout of a group)
};

// Work queue of the time-sliced rollout (earl_sawyer_state.sched: progress[G] then lock[G], zero on entry).  An env group's state is in HBM after every env
// step (the failure guard's "last stable state"), so ANY wave can take the group's next slice of env steps; a wave claims the unlocked group that has come
// LEAST far.  The groups whose envs are in contact -- the slow chains a statically scheduled launch waits for at the end of its second round -- are then
// re-claimed the moment they are released and run without a break from the start, while the fast groups share the other wave slots: the launch tends to
// total work / wave slots instead of (typical wave) + (slowest wave).  Results do not depend on the schedule: an env's arithmetic is its own.
// `home`: where this wave starts looking among groups that have come equally far (its own index in the launch x 2): at the start every group stands at 0, and
// a thousand waves going for group 0 at once would fight over every lock in turn
__device__ __forceinline__ int sched_claim(int32_t* sched, const int G, const int T, const int lane, const int home, int& t0) {
  int32_t* progress = sched;
  int32_t* lock = sched + G;
  for (;;) {
    unsigned long long best = ~0ull;
    for (int gi = lane; gi < G; gi += 64) {
      const int p = __hip_atomic_load(progress + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int l = __hip_atomic_load(lock + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const int rot = gi >= home ? gi - home : gi - home + G;                 // distance from `home`, going up and around
      const unsigned long long key = ((unsigned long long)(unsigned int)p << 32) | (unsigned int)rot;
      best = (l == 0 && p < T && key < best) ? key : best;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const unsigned long long o = __shfl_xor(best, off);
      best = o < best ? o : best;
    }
    if (best == ~0ull) return -1;                       // every unfinished group is in some wave's hands: nothing to do for this one
    const int rot_ = (int)(best & 0xFFFFFFFFull);
    const int gi = rot_ + home < G ? rot_ + home : rot_ + home - G;
    int ok = 0;
    if (lane == 0) ok = atomicCAS(lock + gi, 0, 1) == 0 ? 1 : 0;
    ok = __shfl(ok, 0);
    if (!ok) continue;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the rows the previous holder of this group wrote
    const int p = __hip_atomic_load(progress + gi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (p >= T) {                                       // (finished between the scan and the lock)
      if (lane == 0) __hip_atomic_store(lock + gi, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      continue;
    }
    t0 = p;
    return gi;
  }
}
__device__ __forceinline__ void sched_release(int32_t* sched, const int G, const int g, const int t1, const int lane) {
  fence();
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");     // state rows, output rows, goal rows of this slice -> visible to the next holder
  if (lane == 0) {
    __hip_atomic_store(sched + g, t1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(sched + G + g, 0, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// metaworld reward_utils.tolerance(x, bounds=(0, hi), margin, sigmoid='gaussian') [UPSTREAM, dm_control semantics; unpinned]
__device__ __forceinline__ double tolerance_gaussian(double x, double hi, double margin) {
#pragma clang fp contract(off)
  if (0.0 <= x && x <= hi) return 1.0;
  if (margin == 0) return 0.0;
  const double d = (x < 0.0 ? -x : x - hi) / margin;
  const double scale = sqrt(-2.0 * log(0.1));
  return exp(-0.5 * (d * scale) * (d * scale));
}

// reward + success of one observation row (sawyer_door.py:141-177)
__device__ __forceinline__ void door_reward(const earl_sawyer_cfg& cfg, const V3 tcp, const V3 obj, const V3 target, double& r, bool& ok, double* info = nullptr) {
#pragma clang fp contract(off)
  const V3 d = vsub(obj, target);
  const double obj_to_target = sqrt(d.x * d.x + d.y * d.y + d.z * d.z);     // np.linalg.norm in f64
  ok = obj_to_target <= cfg.success_radius;
  r = ok ? 1.0 : 0.0;
  if (cfg.reward_type != 0 || info) {
    const V3 e = vsub(tcp, obj);
    const V3 oi = vsub(ld3(cfg.obj_init_pos), target), hi = vsub(ld3(cfg.hand_init_pos), obj);
    const double in_place = tolerance_gaussian(obj_to_target, 0.05, sqrt(oi.x * oi.x + oi.y * oi.y + oi.z * oi.z));
    const double hand_in_place = tolerance_gaussian(sqrt(e.x * e.x + e.y * e.y + e.z * e.z), 0.25 * 0.05, sqrt(hi.x * hi.x + hi.y * hi.y + hi.z * hi.z) + 0.1);
    if (cfg.reward_type != 0) {
      r = 3 * hand_in_place + 6 * in_place;
      if (obj_to_target < 0.05) r = 10;
    }
    if (info) {
      // SawyerDoorV2.evaluate_state (sawyer_door.py:127-139); compute_reward returns [reward, obj_to_target, hand_in_place] (:171), so the dict's
      // 'in_place_reward' is the hand's term
      info[EARL_INFO_SUCCESS] = obj_to_target <= 0.08 ? 1.0 : 0.0;
      info[EARL_INFO_NEAR_OBJECT] = 0.0; info[EARL_INFO_GRASP_SUCCESS] = 1.0; info[EARL_INFO_GRASP_REWARD] = 1.0;
      info[EARL_INFO_IN_PLACE_REWARD] = hand_in_place; info[EARL_INFO_OBJ_TO_TARGET] = obj_to_target; info[EARL_INFO_UNSCALED_REWARD] = r;
      info[7] = 0.0;
    }
  }
}

// ---- metaworld reward_utils / SawyerXYZEnv._gripper_caging_reward [UPSTREAM metaworld, not in the reference tree; UNPINNED]:
// restated as in oracle/sawyer_oracle.py (tolerance_long_tail, rect_prism_tolerance, hamacher_product, gripper_caging_reward)
__device__ __forceinline__ double tol_long_tail(double x, double lo, double hi, double margin) {
#pragma clang fp contract(off)
  if (lo <= x && x <= hi) return 1.0;
  if (margin == 0) return 0.0;
  const double d = (x < lo ? lo - x : x - hi) / margin;
  const double scale = sqrt(1 / 0.1 - 1);
  return 1 / ((d * scale) * (d * scale) + 1);
}
__device__ __forceinline__ bool in_rng(double a, double b, double c) { return c >= b ? (b <= a && a <= c) : (c <= a && a <= b); }
__device__ __forceinline__ double rect_prism_tol(const V3 cur, const double* zero, const double* one) {
#pragma clang fp contract(off)
  if (in_rng(cur.x, zero[0], one[0]) && in_rng(cur.y, zero[1], one[1]) && in_rng(cur.z, zero[2], one[2]))
    return (cur.x - zero[0]) / (one[0] - zero[0]) * ((cur.y - zero[1]) / (one[1] - zero[1])) * ((cur.z - zero[2]) / (one[2] - zero[2]));
  return 1.0;
}
__device__ __forceinline__ double hamacher(double a, double b) {
#pragma clang fp contract(off)
  const double den = a + b - (a * b);
  return den > 0 ? (a * b) / den : 0.0;
}
// SawyerPegV2.compute_reward, reward_type 'dense' (sawyer_peg.py:231-299); head = obs[4:7] (site pegHead), tcp = obs[:3] (hand)
// dense = false: reward_type 'sparse' (sawyer_peg.py:284-285: object_grasped = 0 unless lifted); the terms are still worked out, for the info dict (terms[]:
// tcp_to_obj, obj_to_target (axis-scaled), object_grasped, in_place; may be NULL)
__device__ __forceinline__ double peg_dense_reward(const earl_sawyer_cfg& cfg, const V3 tcp, const double tcp_opened, const V3 head, const V3 grasp,
                                                   const V3 lpad, const V3 rpad, const V3 tcpc, const V3 target, const double* __restrict__ oi,
                                                   const double effort, const bool dense = true, double* terms = nullptr) {
#pragma clang fp contract(off)
  const V3 obj = grasp;                                   // obs[4:7] - pegHead + pegGrasp with obs[4:7] == pegHead
  const V3 e = vsub(obj, tcp);
  const double tcp_to_obj = sqrt(e.x * e.x + e.y * e.y + e.z * e.z);
  const V3 ht{(head.x - target.x) * 1.0, (head.y - target.y) * 2.0, (head.z - target.z) * 2.0};
  const double obj_to_target = sqrt(ht.x * ht.x + ht.y * ht.y + ht.z * ht.z);
  const V3 hi{(oi[3] - target.x) * 1.0, (oi[4] - target.y) * 2.0, (oi[5] - target.z) * 2.0};
  double in_place = tol_long_tail(obj_to_target, 0.0, 0.05, sqrt(hi.x * hi.x + hi.y * hi.y + hi.z * hi.z));
  const double box1 = rect_prism_tol(head, cfg.box_corners[0], cfg.box_corners[1]), box2 = rect_prism_tol(head, cfg.box_corners[2], cfg.box_corners[3]);
  in_place = hamacher(in_place, hamacher(box2, box1));
  const bool lifted = tcp_to_obj < 0.08 && tcp_opened > 0 && obj.z - 0.01 > oi[2];
  double grasped = 1.0;
  if (!lifted && !dense) grasped = 0.0;
  if (!lifted && dense) {
    // _gripper_caging_reward(action, obj, obj_radius 0.0075, pad_success_thresh 0.03, object_reach_radius 0.01, xz_thresh 0.005, high_density)
    const double pl = fabs(lpad.y - obj.y), pr = fabs(rpad.y - obj.y);
    const double ml = fabs(fabs(lpad.y - oi[1]) - 0.03), mr = fabs(fabs(rpad.y - oi[1]) - 0.03);
    const double caging_y = hamacher(tol_long_tail(pl, 0.0075, 0.03, ml), tol_long_tail(pr, 0.0075, 0.03, mr));
    const double ix = oi[0] - cfg.init_tcp[0], iz = oi[2] - cfg.init_tcp[2];
    const double dx = tcpc.x - obj.x, dz = tcpc.z - obj.z;
    const double caging_xz = tol_long_tail(sqrt(dx * dx + dz * dz), 0.0, 0.005, sqrt(ix * ix + iz * iz) - 0.005);
    const double closed = fmin(fmax(0.0, effort), 1.0) / 1.0;
    const double caging = hamacher(caging_y, caging_xz);
    const double gripping = caging > 0.97 ? closed : 0.0;
    grasped = (hamacher(caging, gripping) + caging) / 2;
  }
  double r = hamacher(grasped, in_place);
  if (lifted) r += 1.0 + 5 * in_place;
  if (obj_to_target <= 0.05) r = 10.0;
  if (terms) { terms[0] = tcp_to_obj; terms[1] = obj_to_target; terms[2] = grasped; terms[3] = in_place; }
  return r;
}

// any lane of this env's LPE-lane group (the whole wavefront calls it)

// obs[14] + reward + flags of one env from the kinematics in LDS (sawyer_door.py:86-94, :141-177); the whole group calls it
template <int NV>
__device__ __forceinline__ void sawyer_emit(Shared<NV>& s, const typename ModelOf<NV>::T& m, const earl_sawyer_cfg& cfg, const int sub, const bool live,
                                            const double* __restrict__ goal, double* __restrict__ obs, float* reward, uint8_t* success,
                                            const double* __restrict__ obj_init = nullptr, const double effort = 0.0, double* __restrict__ obs2 = nullptr,
                                            double* __restrict__ info = nullptr) {
#pragma clang fp contract(off)
  // (compiled into the peg model's kernels only: in the door kernel this code cost 35 more AGPR spills and 10 % of its throughput.  The door's info dict is a
  // function of the observation alone: earl_sawyer_door_info works it out from the emitted rows.)
  const bool peg_terms = NV >= 15 && cfg.obj_kind >= 1 && obj_init != nullptr && (cfg.reward_type != 0 || info != nullptr);
  const bool peg_dense = peg_terms && cfg.reward_type != 0;
  if (sub < (peg_terms ? 7 : 4)) {
    const int k = sub == 0 ? cfg.att_hand : (sub == 1 ? cfg.att_right : (sub == 2 ? cfg.att_left : (sub == 3 ? cfg.att_obj :
                  (sub == 4 ? cfg.att_grasp : (sub == 5 ? cfg.att_lpad : cfg.att_rpad)))));
    const V3 p = attachment<NV>(s, m, k);
    s.emit.att[sub][0] = p.x; s.emit.att[sub][1] = p.y; s.emit.att[sub][2] = p.z;
  }
  fence();
  if (sub < 14 && live && (obs || obs2)) {
    double v;
    if (sub < 3) v = s.emit.att[0][sub];
    else if (sub == 3) {
      const V3 d = vsub(ld3(s.emit.att[1]), ld3(s.emit.att[2]));
      v = fmin(fmax(sqrt(d.x * d.x + d.y * d.y + d.z * d.z) / 0.1, 0.0), 1.0);
    } else if (sub < 7) v = s.emit.att[3][sub - 4];
    else v = goal[sub - 7];
    if (obs) obs[sub] = v;
    if (obs2) obs2[sub] = v;
  }
  if (sub == 0 && live) {
    double r; bool ok;
    door_reward(cfg, ld3(s.emit.att[0]), ld3(s.emit.att[3]), ld3(goal + 4), r, ok);
    if constexpr (NV >= 15) if (peg_terms) {
      const V3 rr = ld3(s.emit.att[1]), ll = ld3(s.emit.att[2]), dg = vsub(rr, ll);
      const double opened = fmin(fmax(sqrt(dg.x * dg.x + dg.y * dg.y + dg.z * dg.z) / 0.1, 0.0), 1.0);      // obs[3]
      double terms[4];
      const double rd = peg_dense_reward(cfg, ld3(s.emit.att[0]), opened, ld3(s.emit.att[3]), ld3(s.emit.att[4]), ld3(s.emit.att[5]), ld3(s.emit.att[6]),
                                         scl(add(rr, ll), 0.5), ld3(goal + 4), obj_init, effort, peg_dense, terms);
      if (peg_dense) r = rd;
      if (info) {
        // SawyerPegV2.evaluate_state (sawyer_peg.py:165-184): tcp_to_obj to the pegGrasp site, obj = the observation's pegHead, TARGET_RADIUS 0.05
        const double headz = s.emit.att[3][2];
        info[EARL_INFO_SUCCESS] = terms[1] <= 0.05 ? 1.0 : 0.0;
        info[EARL_INFO_NEAR_OBJECT] = terms[0] <= 0.03 ? 1.0 : 0.0;
        info[EARL_INFO_GRASP_SUCCESS] = (terms[0] < 0.02 && opened > 0 && headz - 0.01 > obj_init[2]) ? 1.0 : 0.0;
        info[EARL_INFO_GRASP_REWARD] = terms[2]; info[EARL_INFO_IN_PLACE_REWARD] = terms[3]; info[EARL_INFO_OBJ_TO_TARGET] = terms[1];
        info[EARL_INFO_UNSCALED_REWARD] = r; info[7] = 0.0;
      }
    }
    if (reward) *reward = (float)r;
    if (success) *success = ok ? 1 : 0;
  }
  fence();
}

#ifndef EARL_WAVES_PER_EU
#define EARL_WAVES_PER_EU 1
#endif
// SLICED: the launch's work is a queue of (env group, slice of a.slice env steps) items (sched_claim above) taken by persistent waves, instead of one
// whole rollout of one group per wave
template <int NV, int LPE, bool SLICED = false>
__global__ __launch_bounds__(64 * Lim<NV>::WPB, EARL_WAVES_PER_EU) void sawyer_rollout_kernel(const SawyerArgs a) {
  static_assert(LPE >= 14, "the observation is written by 14 lanes");
  constexpr int EPW = 64 / LPE, WPB = Lim<NV>::WPB;
  __shared__ alignas(16) typename ModelOf<NV>::T m;
  __shared__ alignas(16) BlkTable<Lim<NV>::MB, Lim<NV>::KBT> bt;
  __shared__ alignas(16) Shared<NV> sh[EPW * WPB];
  stage_blocks(bt, a.col);
  stage_kb<NV>(bt, a.m, a.col);
  stage_model(m, a.m);
  const earl_sawyer_cfg& cfg = a.cfg;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), sub = lane % LPE, grp = lane / LPE, n = cfg.n;
#ifdef EARL_PHYS_PROF
  const unsigned long long wave_t0 = __builtin_readcyclecounter();
#endif
  Shared<NV>& s = sh[wave * EPW + grp];
  if constexpr (Lim<NV>::TS < Lim<NV>::NT) {            // the mass-matrix entries between the two trees are never written (K5): zero, once
    for (int k = sub; k < (int)(sizeof(s.M.v) / sizeof(double)); k += LPE) s.M.v[k] = 0.0;
  }
  const Q4 mq = ldq(cfg.mocap_quat);                     // as given, NOT normalised (include/earl_physics.h)
  const int gcf = a.st.steps_since_goal_change ? cfg.goal_change_frequency : 0;
  const float scale = (float)cfg.action_scale;
  const int G = (n + EPW - 1) / EPW;                     // env groups (one per wave at a time)
  for (;;) {
  int group = blockIdx.x * WPB + wave, t_begin = 0, t_end = a.T;
  if constexpr (SLICED) {
    group = sched_claim(a.st.sched, G, a.T, lane, (int)(((blockIdx.x * WPB + wave) * 2) % G), t_begin);
    if (group < 0) break;
    t_end = t_begin + a.slice < a.T ? t_begin + a.slice : a.T;
  }
  const int env_raw = group * EPW + grp;
  const bool live = env_raw < n;
  const int env = live ? env_raw : n - 1;
  load_state<NV>(s, m, a.st.qpos + (size_t)env * m.nq, a.st.qvel + (size_t)env * NV, sub);
  if (sub < 3) s.mocap[sub] = a.st.mocap_pos[(size_t)env * 3 + sub];
  fence();
  int steps = a.st.steps_since_reset ? a.st.steps_since_reset[env] : 0;
  int sgc = gcf > 0 ? a.st.steps_since_goal_change[env] : 0;
  RSTART();
  for (int t = t_begin; t < t_end; ++t) {
    const float4 act = *reinterpret_cast<const float4*>(a.action + ((size_t)t * n + env) * 4);
    // set_xyz_action [UPSTREAM]: clip, float32 product with the scale, float64 add, box clip
    const float cx = fminf(fmaxf(act.x, -1.f), 1.f) * scale, cy = fminf(fmaxf(act.y, -1.f), 1.f) * scale, cz = fminf(fmaxf(act.z, -1.f), 1.f) * scale;
    if (sub < 3) {                                      // lane k moves coordinate k
      const float ck = sub == 0 ? cx : (sub == 1 ? cy : cz);
      s.mocap[sub] = fmin(fmax(s.mocap[sub] + (double)ck, cfg.mocap_low[sub]), cfg.mocap_high[sub]);
    }
    fence();
    const double ctrl[EARL_MAXACT] = {(double)act.w, -(double)act.w, 0, 0};
    RSTAMP(12);
    for (int ts = 0; ts < cfg.frame_skip; ++ts) {
      // The lane's index is passed through an empty asm at the head of every timestep: the per-lane LDS addresses derived from it are then recomputed
      // inside the timestep (a few integer adds) instead of being hoisted out of the rollout loop, where dozens of them lived across the whole kernel
      // and went to scratch memory under the register cap -- every reload is a global-memory round trip on the timestep's critical path (scratch per
      // lane: eight-wave door build 296 -> 212 B, peg 36 -> 0 B).  (Doing the same to the block pointer hides that it is an LDS address: 640 B.)
      // (Small model only: the peg build, with 512 registers, loses 2 % to the recomputation although its last 36 B of scratch go too.)
      int sub_ = sub, grp_ = grp;
      if constexpr (NV <= 10) asm volatile("" : "+v"(sub_));
      else asm volatile("" : "+v"(grp_));               // (peg: the block's base address was what got spilled, and reloaded six times per timestep)
      __builtin_assume(sub_ >= 0 && sub_ < LPE);
      __builtin_assume(grp_ >= 0 && grp_ < EPW);
      substep<NV, LPE, true>(sh[wave * EPW + grp_], m, bt, a.col, sub_, grp_, mq, ctrl, ts > 0, nullptr, nullptr);   // (every env step starts cold: step() x T == rollout(T))
    }
    RSTAMP(13);
    const size_t row = (size_t)t * n + env;
    // failure guard (MuJoCo's mj_checkPos / mj_checkVel; metaworld's `except MujocoException` in SawyerXYZEnv.step [UPSTREAM]): an env whose
    // state went NaN or beyond EARL_BAD_VALUE is rolled back to its last stable state (the rows in HBM) and re-emits its last stable
    // observation with reward 0; its neighbours in the wavefront never see it (a group only reads its own LDS block)
    const bool bad_lane = (sub < NV && !(fabs(s.qp[sub]) < EARL_BAD_VALUE && fabs(s.qv[sub]) < EARL_BAD_VALUE)) || (sub < 4 && !(fabs(s.bq[sub]) < 2.0));
    const bool failed = group_any<LPE>(bad_lane, grp);
    sawyer_emit<NV>(s, m, cfg, sub, live && !failed, a.st.goal + (size_t)env * 7, a.out.obs + row * 14, a.out.reward ? a.out.reward + row : nullptr,
                    a.out.success ? a.out.success + row : nullptr, a.st.obj_init ? a.st.obj_init + (size_t)env * 6 : nullptr, (double)act.w, nullptr,
                    (NV >= 15 && a.out.info) ? a.out.info + row * EARL_SAWYER_INFO : nullptr);
    RSTAMP(14);
    if (sub == 0 && live && a.out.status) a.out.status[row] = failed ? EARL_STEP_DIVERGED : 0;
    if (!failed) {
      // this state is the env's last stable one from here on
      if (live) {
        store_state<NV>(s, m, a.st.qpos + (size_t)env * m.nq, a.st.qvel + (size_t)env * NV, sub);
        if (sub < 3) a.st.mocap_pos[(size_t)env * 3 + sub] = s.mocap[sub];
      }
      if constexpr (NV >= 15) {
        // the free body's orientation as load_state would read it back from the row just stored (re-normalised, the same expression): a rollout, its
        // time slices taken by different waves, and T single-step launches then walk through the same bits (like the minitaur kernel)
        if (m.ball_dof >= 0) {
          const double qn = renormalised_quat_entry<NV>(s, sub);
          fence();
          if (sub < 4) s.bq[sub] = qn;
          fence();
        }
      }
    } else {
      load_state<NV>(s, m, a.st.qpos + (size_t)env * m.nq, a.st.qvel + (size_t)env * NV, sub);
      if (sub < 3) s.mocap[sub] = a.st.mocap_pos[(size_t)env * 3 + sub];
      if (live) {
        const double* prev = t > 0 ? a.out.obs + ((size_t)(t - 1) * n + env) * 14 : (a.st.last_obs ? a.st.last_obs + (size_t)env * 14 : nullptr);
        if (sub < 14) a.out.obs[row * 14 + sub] = prev ? prev[sub] : __builtin_nan("");
        if (sub == 0) {
          if (a.out.reward) a.out.reward[row] = 0.f;
          if (a.out.success) a.out.success[row] = 0;
          if (a.st.fail_count) a.st.fail_count[env] += 1;
        }
        if (NV >= 15 && a.out.info && sub < EARL_SAWYER_INFO) a.out.info[row * EARL_SAWYER_INFO + sub] = 0.0;
      }
    }
    fence();
    ++steps;
    RSTAMP(15);
    if (sub == 0 && live && a.out.done) a.out.done[row] = (cfg.horizon > 0 && steps >= cfg.horizon) ? 1 : 0;
    // door with goal switching: slot 7 of EVERY row's info block is this kernel's to write -- 0, or 1 on a goal-switch row (below) -- so that earl_sawyer_door_info never reads
    // a marker the caller left behind (the Python side used to zero the column with a launch of its own before every call)
    if (NV < 15 && gcf > 0 && a.out.info && sub == 11 && live) a.out.info[row * EARL_SAWYER_INFO + 7] = 0.0;
    if (gcf > 0 && ++sgc >= gcf) {
      // LifelongWrapper.step (lifelong_wrapper.py:36-42): reset_goal() -> get_next_goal(), then the observation is re-read with the new
      // goal (same simulator state: only the goal block changes); the reward above used the old goal
      sgc = 0;
      if (cfg.n_goal_rows > 0 && cfg.goal_table && sub >= 7 && sub < 14 && live) {      // the lanes that wrote the goal block of this row
        const uint64_t ev = cfg.step_counter + (uint64_t)t;
        const earl::U4 b = earl::philox4x32_10(earl::U4{0xFFFEu, (uint32_t)(cfg.env_offset + env), (uint32_t)ev, (uint32_t)(ev >> 32)},
                                               (uint32_t)cfg.seed, (uint32_t)(cfg.seed >> 32));
        int grow = (int)(earl::u01(b.x, b.y) * (double)cfg.n_goal_rows);
        grow = grow < cfg.n_goal_rows ? grow : cfg.n_goal_rows - 1;
        const double gv = cfg.goal_table[(size_t)grow * 7 + (sub - 7)];
        if (NV < 15 && a.out.info && sub >= 11) {
          // the door's info dict is worked out after the launch from the emitted rows (earl_sawyer_door_info), whose goal block is about to change: this
          // row's info slots 0-2 carry the target the row's reward was computed with, slot 7 marks it (evaluate_state runs before reset_goal:
          // lifelong_wrapper.py:30-44; include/earl_physics.h)
          a.out.info[row * EARL_SAWYER_INFO + (sub - 11)] = a.st.goal[(size_t)env * 7 + (sub - 7)];
          if (sub == 11) a.out.info[row * EARL_SAWYER_INFO + 7] = 1.0;
        }
        a.st.goal[(size_t)env * 7 + (sub - 7)] = gv;
        a.out.obs[row * 14 + sub] = gv;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");      // the next step's observation reads the goal row back through global memory
    }
  }
  if (live) {
    // (qpos / qvel / mocap_pos were written back after the last stable step)
    if (t_end == a.T && a.st.last_obs && a.T > 0 && sub < 14) a.st.last_obs[(size_t)env * 14 + sub] = a.out.obs[((size_t)(a.T - 1) * n + env) * 14 + sub];
    if (sub == 0 && a.st.steps_since_reset) a.st.steps_since_reset[env] = steps;
    if (sub == 0 && gcf > 0) a.st.steps_since_goal_change[env] = sgc;
  }
  if constexpr (SLICED) sched_release(a.st.sched, G, group, t_end, lane);
  else break;
  }
#ifdef EARL_PHYS_PROF
  if (lane == 0 && blockIdx.x * WPB + wave < 4096) g_wave_cycles[blockIdx.x * WPB + wave] = __builtin_readcyclecounter() - wave_t0;
#endif
}

// reset (masked) / observe: both end with the kinematics of the current state and the observation
template <int NV, int LPE>
__global__ __launch_bounds__(64 * Lim<NV>::WPB) void sawyer_reset_kernel(const SawyerArgs a) {
  constexpr int EPW = 64 / LPE, WPB = Lim<NV>::WPB;
  __shared__ alignas(16) typename ModelOf<NV>::T m;
  __shared__ alignas(16) BlkTable<Lim<NV>::MB, Lim<NV>::KBT> bt;
  __shared__ alignas(16) Shared<NV> sh[EPW * WPB];
  stage_blocks(bt, a.col);
  stage_kb<NV>(bt, a.m, a.col);
  stage_model(m, a.m);
  const earl_sawyer_cfg& cfg = a.cfg;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), sub = lane % LPE, grp = lane / LPE;
  const int env_raw = (blockIdx.x * WPB + wave) * EPW + grp;
  const int env = env_raw < cfg.n ? env_raw : cfg.n - 1;
  Shared<NV>& s = sh[wave * EPW + grp];
  const bool resetting = !a.observe_only && env_raw < cfg.n && (!a.mask || a.mask[env]);
  const bool live = env_raw < cfg.n && (a.observe_only || resetting);
  if constexpr (Lim<NV>::TS < Lim<NV>::NT) {            // (the forward pass below reads the whole mass matrix; K5 leaves the entries between the trees alone)
    for (int k = sub; k < (int)(sizeof(s.M.v) / sizeof(double)); k += LPE) s.M.v[k] = 0.0;
  }
  if (sub < 3) s.mocap[sub] = a.st.mocap_pos[(size_t)env * 3 + sub];
  if (resetting) {
    const uint32_t gid = (uint32_t)(cfg.env_offset + env), c0 = (uint32_t)cfg.counter, c1 = (uint32_t)(cfg.counter >> 32);
    const uint32_t k0 = (uint32_t)cfg.seed, k1 = (uint32_t)(cfg.seed >> 32);
    load_state<NV>(s, m, a.reset_qpos, a.reset_qvel, sub);
    fence();
    if (cfg.obj_kind == 0) {
      const earl::U4 b = earl::philox4x32_10(earl::U4{0u, gid, c0, c1}, k0, k1);
      // np.random.uniform(lo, hi) = lo + (hi - lo) * u   (sawyer_door.py:116-118)
      double angle;
      {
#pragma clang fp contract(off)
        angle = cfg.obj_init_angle + (cfg.angle_noise[0] + (cfg.angle_noise[1] - cfg.angle_noise[0]) * earl::u01(b.x, b.y));
      }
      if (sub == cfg.obj_dof) { s.qp[sub] = angle; s.qv[sub] = 0.0; }
    } else {
      // sawyer_peg.py:199-212 / :221-223: xyz ~ U(obj_low, obj_high), redrawn while the xy distance to the hole block is < 0.1;
      // _set_obj_xyz [UPSTREAM]: qpos[9:12] <- xyz, qvel[9:15] <- 0 (the orientation is left as it is)
      double px = 0, py = 0, pz = 0;
      bool wide = false;
      if (cfg.obj_kind == 2 && cfg.n_wide > 0 && cfg.wide_table) {
        // wide_init (sawyer_peg.py:200-209): np.random.uniform() < 0.5 keeps the default draw below; otherwise a row of the wide table
        // (shifted by +0.1 in x: "- np.array([-0.1, 0, 0])") plus U(-0.02, 0.02)^3
#pragma clang fp contract(off)
        const earl::U4 c0_ = earl::philox4x32_10(earl::U4{0xFFF0u, gid, c0, c1}, k0, k1);
        const earl::U4 c1_ = earl::philox4x32_10(earl::U4{0xFFF1u, gid, c0, c1}, k0, k1);
        wide = !(earl::u01(c0_.x, c0_.y) < 0.5);
        int wr = (int)(earl::u01(c0_.z, c0_.w) * (double)cfg.n_wide);
        wr = wr < cfg.n_wide ? wr : cfg.n_wide - 1;
        const double lo = -cfg.wide_noise, hi = cfg.wide_noise;
        px = (cfg.wide_table[wr * 3 + 0] + cfg.wide_shift[0]) + (lo + (hi - lo) * earl::u01(c1_.x, c1_.y));
        py = (cfg.wide_table[wr * 3 + 1] + cfg.wide_shift[1]) + (lo + (hi - lo) * earl::u01(c1_.z, c1_.w));
        const earl::U4 c2_ = earl::philox4x32_10(earl::U4{0xFFF2u, gid, c0, c1}, k0, k1);
        pz = (cfg.wide_table[wr * 3 + 2] + cfg.wide_shift[2]) + (lo + (hi - lo) * earl::u01(c2_.x, c2_.y));
      }
      for (uint32_t attempt = 0; attempt < 16u && !wide; ++attempt) {
#pragma clang fp contract(off)
        const earl::U4 b0 = earl::philox4x32_10(earl::U4{2u * attempt, gid, c0, c1}, k0, k1);
        const earl::U4 b1 = earl::philox4x32_10(earl::U4{2u * attempt + 1u, gid, c0, c1}, k0, k1);
        px = cfg.obj_low[0] + (cfg.obj_high[0] - cfg.obj_low[0]) * earl::u01(b0.x, b0.y);
        py = cfg.obj_low[1] + (cfg.obj_high[1] - cfg.obj_low[1]) * earl::u01(b0.z, b0.w);
        pz = cfg.obj_low[2] + (cfg.obj_high[2] - cfg.obj_low[2]) * earl::u01(b1.x, b1.y);
        const double dx = px - cfg.obj_reject_xy[0], dy = py - cfg.obj_reject_xy[1];
        if (!(sqrt(dx * dx + dy * dy) < cfg.obj_reject_radius)) break;
      }
      const int k = sub - cfg.obj_dof;
      if (k >= 0 && k < 6 && sub < NV) {
        if (k < 3) s.qp[sub] = k == 0 ? px : (k == 1 ? py : pz);
        s.qv[sub] = 0.0;
      }
    }
    if (cfg.n_goal_rows > 0 && cfg.goal_table && sub < 7) {
      // get_next_goal with reset_at_goal (sawyer_peg.py:149-152): np.random.randint(0, rows) -> own Philox draw
      const earl::U4 b = earl::philox4x32_10(earl::U4{0xFFFFu, gid, c0, c1}, k0, k1);
      int row = (int)(earl::u01(b.x, b.y) * (double)cfg.n_goal_rows);
      row = row < cfg.n_goal_rows ? row : cfg.n_goal_rows - 1;
      a.st.goal[(size_t)env * 7 + sub] = cfg.goal_table[(size_t)row * 7 + sub];
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");    // the observation below reads the goal row back through global memory
    fence();
    store_state<NV>(s, m, a.st.qpos + (size_t)env * m.nq, a.st.qvel + (size_t)env * NV, sub);
    if (sub < 3) { s.mocap[sub] = cfg.hand_init_pos[sub]; a.st.mocap_pos[(size_t)env * 3 + sub] = cfg.hand_init_pos[sub]; }
    if (sub == 0 && a.st.steps_since_reset) a.st.steps_since_reset[env] = 0;
    if (sub == 0 && a.st.steps_since_goal_change) a.st.steps_since_goal_change[env] = 0;     // LifelongWrapper.reset (lifelong_wrapper.py:25-28)
  } else {
    load_state<NV>(s, m, a.st.qpos + (size_t)env * m.nq, a.st.qvel + (size_t)env * NV, sub);
  }
  fence();
  const bool keep = resetting && ((cfg.obj_kind >= 1 && a.st.obj_init) || a.st.last_obs);     // uniform enough: decided per lane, used per lane below
  if (!a.reset_obs && !__any(keep)) return;
  // set_state -> sim.forward(): kinematics of the state just written
  const Q4 mq = ldq(cfg.mocap_quat);                     // as given, NOT normalised (include/earl_physics.h)
  const double ctrl[EARL_MAXACT] = {0, 0, 0, 0};
  substep<NV, LPE, false>(s, m, bt, nullptr, sub, grp, mq, ctrl, false, nullptr, nullptr);
  sawyer_emit<NV>(s, m, cfg, sub, live, a.st.goal + (size_t)env * 7, a.reset_obs ? a.reset_obs + (size_t)env * 14 : nullptr, nullptr, nullptr, nullptr, 0.0,
                  (resetting && a.st.last_obs) ? a.st.last_obs + (size_t)env * 14 : nullptr);
  // reset_model keeps obj_init_pos and the pegHead site of the freshly placed peg for the dense reward (sawyer_peg.py:213-215)
  if (resetting && cfg.obj_kind >= 1 && a.st.obj_init && sub < 6) {
    double* oi = a.st.obj_init + (size_t)env * 6;
    oi[sub] = sub < 3 ? s.qp[cfg.obj_dof + sub] : s.emit.att[3][sub - 3];
  }
}

#ifndef EARL_PHYS_NOT_MAIN
// compute_reward / is_successful on given observations (sawyer_door.py:141-177), one lane per row
__global__ void sawyer_door_reward_kernel(const int n, const double* __restrict__ obs, const earl_sawyer_cfg cfg, float* __restrict__ reward,
                                          uint8_t* __restrict__ success) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* o = obs + (size_t)i * 14;
  double r; bool ok;
  door_reward(cfg, ld3(o), ld3(o + 4), ld3(o + 11), r, ok);
  if (reward) reward[i] = (float)r;
  if (success) success[i] = ok ? 1 : 0;
}
#endif

#ifndef EARL_PHYS_NOT_MAIN
// SawyerDoorV2.evaluate_state's info dict (sawyer_door.py:127-139) of given observation rows: every entry is a function of the observation (and of the
// reward type), so the rollout kernel need not carry it; one lane per row
__global__ void sawyer_door_info_kernel(const int n, const double* __restrict__ obs, const earl_sawyer_cfg cfg, const uint8_t* __restrict__ status, double* __restrict__ info) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* o = obs + (size_t)i * 14;
  double r, row[EARL_SAWYER_INFO]; bool ok;
  double* mine = info + (size_t)i * EARL_SAWYER_INFO;
  // a goal-switch row of a lifelong rollout (cfg.goal_change_frequency > 0: only then has the rollout kernel written the marker, for every row): the target its reward used
  // (the row's goal block holds the NEW goal).  Without goal switching the slot is output only.
  const V3 target = (cfg.goal_change_frequency > 0 && mine[7] == 1.0) ? ld3(mine) : ld3(o + 11);
  door_reward(cfg, ld3(o), ld3(o + 4), target, r, ok, row);
  const bool rolled_back = status && status[i] != 0;
#pragma unroll
  for (int k = 0; k < EARL_SAWYER_INFO; ++k) info[(size_t)i * EARL_SAWYER_INFO + k] = rolled_back ? 0.0 : row[k];
}
#endif

