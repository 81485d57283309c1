// physics_env_kitchen.h -- the kitchen env kernels: the small per-step kernels around the stepper and the fused rollout
// A section of csrc/physics.hip (included there, inside its anonymous namespace, after the stepper): split out in round 5 so that a change to one env's kernels
// recompiles only the translation units that hold them (csrc/Makefile lists the headers per unit).

// ------------------------------------------------------------------------------------------------ kitchen env step (include/earl_physics.h)
// small per-env kernels around the stepper; the numpy glue of the reference (action scaling, observation noise, reward) stays in csrc/glue.hip
struct KitchenArgs {
  earl_kitchen_cfg cfg;
  earl_kitchen_state st;
  earl_kitchen_out out;
  const float* action;
  int n_att;
};
// before the stepper: the float32 action promoted to float64 (np.clip keeps float32; the reference's scaling then promotes), the state saved
__global__ void kitchen_pre_kernel(const KitchenArgs a) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.cfg.n * 23) return;
  const int e = i / 23, j = i % 23;
  a.st.qpos_bak[i] = a.st.qpos[i];
  a.st.qvel_bak[i] = a.st.qvel[i];
  if (j < 9) a.st.action64[e * 9 + j] = (double)a.action[e * 9 + j];
  if (j < 3) a.st.mocap_bak[e * 3 + j] = a.st.mocap_pos[e * 3 + j];      // (before earl_kitchen_action moves the target)
  for (int k = j; k < a.n_att * 3; k += 23) a.st.att_bak[(size_t)e * a.n_att * 3 + k] = a.st.att_xpos[(size_t)e * a.n_att * 3 + k];
}
// after the stepper: failure guard (roll a diverged env back), the eight task sites gathered for the reward
__global__ void kitchen_guard_kernel(const KitchenArgs a) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.cfg.n) return;
  bool bad = false;
  for (int j = 0; j < 23; ++j) bad = bad || !(fabs(a.st.qpos[e * 23 + j]) < EARL_BAD_VALUE) || !(fabs(a.st.qvel[e * 23 + j]) < EARL_BAD_VALUE);
  if (bad) {
    // rolled back: state, the mocap target the diverged step was pulled towards, the attachment positions (possibly NaN) the stepper left
    for (int j = 0; j < 23; ++j) { a.st.qpos[e * 23 + j] = a.st.qpos_bak[e * 23 + j]; a.st.qvel[e * 23 + j] = a.st.qvel_bak[e * 23 + j]; }
    for (int j = 0; j < 3; ++j) a.st.mocap_pos[e * 3 + j] = a.st.mocap_bak[e * 3 + j];
    for (int k = 0; k < a.n_att * 3; ++k) a.st.att_xpos[(size_t)e * a.n_att * 3 + k] = a.st.att_bak[(size_t)e * a.n_att * 3 + k];
    if (a.st.fail_count) a.st.fail_count[e] += 1;
  }
  if (a.out.status) a.out.status[e] = bad ? EARL_STEP_DIVERGED : 0;
  a.st.bad[e] = bad ? 1 : 0;
  for (int k = 0; k < 8; ++k)
    for (int c = 0; c < 3; ++c) a.st.sites[(e * 8 + k) * 3 + c] = a.st.att_xpos[(e * a.n_att + a.cfg.site_att[k]) * 3 + c];
}
// last: the observation / reward / flags of the step (a rolled-back env returns its last stable observation, reward 0), wrapper bookkeeping
__global__ void kitchen_finish_kernel(const KitchenArgs a) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= a.cfg.n) return;
  const bool bad = a.st.bad[e] != 0;
  for (int k = 0; k < 46; ++k) {
    const double v = bad ? a.st.last_obs[e * 46 + k] : a.out.obs[e * 46 + k];
    a.out.obs[e * 46 + k] = v;
    a.st.last_obs[e * 46 + k] = v;
    if (k < 9 && !bad) a.st.last_qp_robot[e * 9 + k] = v;          // the newest cached (noisy) robot joint readings
  }
  if (bad) { a.out.reward[e] = 0.0; a.out.success[e] = 0; }
  const int steps = a.st.steps_since_reset[e] + 1;
  a.st.steps_since_reset[e] = steps;
  a.out.done[e] = (a.cfg.horizon > 0 && steps >= a.cfg.horizon) ? 1 : 0;
}
// The whole env step of earl_kitchen_step, T times, in ONE launch: a wave walks its two envs through action glue -> 40 timesteps -> failure guard ->
// observation (Philox noise) -> reward -> bookkeeping without ever meeting the other waves.  A launch of the stepper lasts as long as its slowest
// wave -- the one env with a finger on a fixture -- and between the launches of consecutive env steps every other wave waited for it; here the
// waves drift apart and only the sum over the rollout counts.  Same arithmetic as the per-step kernels above and csrc/glue.hip (expression by
// expression: kitchen_action_kernel, kitchen_obs_kernel, uniform_kernel, kitchen_reward_kernel, kitchen_guard / finish): bit-identical outputs.
struct KitchenRolloutArgs {
  const void* m;
  const earl_collision_model* col;
  earl_kitchen_params p;
  earl_kitchen_cfg cfg;
  earl_kitchen_state st;
  earl_kitchen_out out;          // rows [T, n, ...]
  const float* action;           // [T, n, 9]
  int T;
  int solo;                      // small batches (round 5): 1 = ONE env per wave -- the wave's second 32-lane group shadows the first one's env (same state, same actions, same
                                 // branches; stores nothing), so the env's chain of timesteps is not held up by a wave-mate on a longer path; 2 = also one wave per workgroup
                                 // (waves 1-3 leave after the tables are staged): every env has a CU's LDS and issue slots to itself.  Same numbers as the packed launch.
};
__device__ __forceinline__ double kit_norm_diff(const double* a, const double* b, const int n) {     // glue.hip norm_diff
  double d = 0.0;
  for (int i = 0; i < n; ++i) {
    const double x = a[i] - b[i];
    d = fma(x, x, d);
  }
  return sqrt(d);
}
// DUO (solo == 3, round 5): the FOUR waves of the workgroup, one per SIMD, work on its one env (substep's ROLE 1 - 4).  Per timestep all run the kinematics; then, side by
// side: wave 0 (B, owns the env) the constraint rows, wave 1 (A) the mass matrix into wave 0's LDS block, wave 2 the bias forces, wave 3 the bounding tests and the collision
// phases (contact records into wave 0's block); barrier X; wave 1 builds the equality Hessian in wave 0's block while wave 0 does the contact rows and its right-hand side;
// barrier Y; wave 0 iterates on the active set while wave 1 factorises for the integration (K10's arm block and fixture scalars: barrier Z); wave 0 integrates; barrier 2;
// waves 1 - 3 copy the new state.  Barrier 0, once per env step, keeps them off the state while wave 0
// does the env step's bookkeeping and hands over the step's actuator targets.  Waves 1 - 3 keep no env state of their own and store nothing outside LDS.
// DUO == 2 (solo == 4: batches of at most TWO envs per CU): two envs per workgroup, two waves each -- waves 0 / 2 own an env and run its collision phases too (ROLE 6), waves
// 1 / 3 do its mass matrix, bias forces, equality Hessian and K10's factor (ROLE 5); the same barriers, shared by the workgroup's two envs.
template <int DUO>
__global__ __launch_bounds__(64 * Lim<23>::WPB) void kitchen_rollout_kernel(const KitchenRolloutArgs a) {
#pragma clang fp contract(off)
  constexpr int NV = 23, LPE = 32, EPW = 64 / LPE, WPB = Lim<NV>::WPB;
  __shared__ alignas(16) typename ModelOf<NV>::T m;
  __shared__ alignas(16) BlkTable<Lim<NV>::MB, Lim<NV>::KBT> bt;
  __shared__ alignas(16) Shared<NV> sh[EPW * WPB];
  __shared__ earl_kitchen_params kp;
  stage_blocks(bt, a.col);
  stage_kb<NV>(bt, a.m, a.col);
  if (threadIdx.x == 0) kp = a.p;
  stage_model(m, a.m);                                  // (ends with the workgroup barrier)
  const earl_kitchen_cfg& cfg = a.cfg;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), sub = lane % LPE, grp = lane / LPE, n = cfg.n;
  if (a.solo >= 2 && DUO == 0 && wave > 0) return;          // (after stage_model's barrier; the DUO form keeps all four waves: its later barriers count them)
  const bool role_a = DUO == 1 ? wave >= 1 : (DUO == 2 && (wave & 1));      // the helper waves: compute, store nothing outside LDS
  const int env_raw = a.solo >= 2 ? (DUO == 2 ? (int)(blockIdx.x * 2 + (wave >> 1)) : (int)blockIdx.x) : (a.solo == 1 ? (int)(blockIdx.x * WPB + wave) : (int)((blockIdx.x * WPB + wave) * EPW + grp));
  const bool live = env_raw < n && (a.solo == 0 || grp == 0) && !role_a;
  const int env = env_raw < n ? env_raw : n - 1;
  Shared<NV>& s = sh[wave * EPW + grp];
  Shared<NV>* const peer = DUO == 1 ? &sh[grp] : (DUO == 2 ? &sh[(wave ^ 1) * EPW + grp] : nullptr);      // helpers: the owner's block of the same 32-lane group; owners: their helper's (wave 1 of the four-wave form)
  load_state<NV>(s, m, a.st.qpos + (size_t)env * NV, a.st.qvel + (size_t)env * NV, sub);
  for (int k = sub; k < (int)(sizeof(s.M.v) / sizeof(double)); k += LPE) s.M.v[k] = 0.0;      // (entries between different trees are never written, K5)
  for (int k = sub; k < (int)(sizeof(s.hwst.Hw.v) / sizeof(double)); k += LPE) s.hwst.Hw.v[k] = 0.0;   // (nor the structural zeros of the equality Hessian, K9)
  if (sub < 3) s.mocap[sub] = a.st.mocap_pos[(size_t)env * 3 + sub];
  fence();
  const Q4 mq = ldq(cfg.mocap_quat_dev);
  if constexpr (DUO != 0) {
    __syncthreads();                                    // wave 0's mass matrix is zeroed before wave 1 writes into it
    if (role_a) {
      for (int t = 0; t < a.T; ++t) {
        __syncthreads();                                // barrier 0: wave 0 is through the env step's bookkeeping (a diverged env went back to its stored state); the step's targets are published
        const double ctrl_a[EARL_MAXACT] = {peer->duo_ctrl[0], peer->duo_ctrl[1], 0, 0};
        for (int ts = 0; ts < cfg.frame_skip; ++ts) {
          if (sub < NV) { s.qp[sub] = peer->qp[sub]; s.qv[sub] = peer->qv[sub]; }
          fence();
          if constexpr (DUO == 2) substep<NV, LPE, true, 5>(s, m, bt, a.col, sub, grp, mq, ctrl_a, false, nullptr, nullptr, peer);      // (barriers X, Y, Z inside)
          else if (wave == 1) substep<NV, LPE, true, 1>(s, m, bt, a.col, sub, grp, mq, ctrl_a, false, nullptr, nullptr, peer);
          else if (wave == 2) substep<NV, LPE, true, 3>(s, m, bt, a.col, sub, grp, mq, ctrl_a, false, nullptr, nullptr, peer);
          else substep<NV, LPE, true, 4>(s, m, bt, a.col, sub, grp, mq, ctrl_a, false, nullptr, nullptr, peer);
          __syncthreads();                              // barrier 2: wave 0 has integrated
        }
      }
      return;
    }
  }
  int steps = a.st.steps_since_reset[env];
  const int kk = sub < 9 ? sub : 8;                     // this lane's action component
#ifdef EARL_PHYS_PROF
  const unsigned long long wave_t0 = __builtin_readcyclecounter();
#endif
  for (int t = 0; t < a.T; ++t) {
    const size_t row = (size_t)t * n + env;
    // ---- KitchenV0.step up to do_simulation (kitchen_action_kernel): mocap target, the nine position targets
    const double mocap_prev = s.mocap[sub < 3 ? sub : 0];      // the target before this step's action: a diverged step goes back to it
    {
      const double x = (double)a.action[row * 9 + kk];
      const double c = x < -1.0 ? -1.0 : (x > 1.0 ? 1.0 : x);
      const double ak = kp.act_mid[kk] + c * kp.act_amp[kk];
      if (sub < 3) {
        const double y = s.mocap[sub] + ak * kp.mocap_range[sub];
        s.mocap[sub] = y < kp.mocap_clip_lower[sub] ? kp.mocap_clip_lower[sub] : (y > kp.mocap_clip_upper[sub] ? kp.mocap_clip_upper[sub] : y);
      }
      if (sub < 9) {
        const double v = ak < kp.vel_bound[sub][0] ? kp.vel_bound[sub][0] : (ak > kp.vel_bound[sub][1] ? kp.vel_bound[sub][1] : ak);
        const double y = a.st.last_qp_robot[(size_t)env * 9 + sub] + v * kp.step_duration;
        s.kit.targets[sub] = y < kp.pos_bound[sub][0] ? kp.pos_bound[sub][0] : (y > kp.pos_bound[sub][1] ? kp.pos_bound[sub][1] : y);
      }
    }
    fence();
    const double ctrl[EARL_MAXACT] = {s.kit.targets[0], s.kit.targets[1], 0, 0};      // do_simulation: ctrl[i] = targets[i] for i < nu = 2
    if (sub < 3 && live) a.st.mocap_pos[(size_t)env * 3 + sub] = s.mocap[sub];
    fence();
    if constexpr (DUO != 0) {
      if (sub < 2) s.duo_ctrl[sub] = ctrl[sub];
      __syncthreads();                                  // barrier 0
      for (int ts = 0; ts < cfg.frame_skip; ++ts) {
        if constexpr (DUO == 2) substep<NV, LPE, true, 6>(s, m, bt, a.col, sub, grp, mq, ctrl, ts > 0, nullptr, nullptr, peer);     // (barriers X, Y, Z inside; `peer`: its helper's block, where that leaves K10's factor)
        else substep<NV, LPE, true, 2>(s, m, bt, a.col, sub, grp, mq, ctrl, ts > 0, nullptr, nullptr, &sh[EPW + grp]);
        __syncthreads();                                // barrier 2
      }
    } else
    for (int ts = 0; ts < cfg.frame_skip; ++ts) substep<NV, LPE, true>(s, m, bt, a.col, sub, grp, mq, ctrl, ts > 0, nullptr, nullptr);
    const bool bad_lane = sub < NV && !(fabs(s.qp[sub]) < EARL_BAD_VALUE && fabs(s.qv[sub]) < EARL_BAD_VALUE);
    const bool failed = group_any<LPE>(bad_lane, grp);
    if (failed) {
      // rolled back to the last stable state (the rows in HBM); returns its last stable observation, reward 0 (kitchen_guard / finish kernels)
      load_state<NV>(s, m, a.st.qpos + (size_t)env * NV, a.st.qvel + (size_t)env * NV, sub);
      if (sub < 3) {                                      // ... incl. the mocap target that pulled it there (att_xpos keeps the last stable positions)
        s.mocap[sub] = mocap_prev;
        if (live) a.st.mocap_pos[(size_t)env * 3 + sub] = mocap_prev;
      }
      if (live) {
        for (int k = sub; k < 46; k += LPE) a.out.obs[row * 46 + k] = a.st.last_obs[(size_t)env * 46 + k];
        if (sub == 0) {
          a.out.reward[row] = 0.0; a.out.success[row] = 0;
          if (a.st.fail_count) a.st.fail_count[env] += 1;
        }
      }
    } else {
      if (live) store_state<NV>(s, m, a.st.qpos + (size_t)env * NV, a.st.qvel + (size_t)env * NV, sub);
      // attachments at the kinematics of the last timestep's start (written only for a step that ended finite); the eight task sites for the reward
      if (sub < m.n_att && live) {
        const V3 p = attachment<NV>(s, m, sub);
        double* o = a.st.att_xpos + ((size_t)env * m.n_att + sub) * 3;
        o[0] = p.x; o[1] = p.y; o[2] = p.z;
      }
      if (sub < 8) {
        const V3 p = attachment<NV>(s, m, cfg.site_att[sub]);
        s.kit.sites[sub][0] = p.x; s.kit.sites[sub][1] = p.y; s.kit.sites[sub][2] = p.z;
      }
      // Robot.get_obs + KitchenV0._get_obs: 46 draws of U(-1, 1) per env (uniform_kernel: one Philox block = two draws), then kitchen_obs_kernel
      if (cfg.sensor_noise && sub < 23) {
        const uint64_t ctr = cfg.counter + (uint64_t)t;
        const earl::U4 b = earl::philox4x32_10(earl::U4{0x4B00u + (uint32_t)sub, (uint32_t)(cfg.env_offset + env), (uint32_t)ctr, (uint32_t)(ctr >> 32)},
                                               (uint32_t)cfg.seed, (uint32_t)(cfg.seed >> 32));
        const double lo = -1.0, hi = 1.0;
        s.kit.noise[2 * sub] = lo + (hi - lo) * earl::u01(b.x, b.y);
        s.kit.noise[2 * sub + 1] = lo + (hi - lo) * earl::u01(b.z, b.w);
      }
      fence();
      for (int k = sub; k < 46; k += LPE) {
        double v;
        if (k < 23) {
          v = s.qp[k];
          if (cfg.sensor_noise) v = v + (kp.robot_noise_ratio * kp.pos_noise_amp[k]) * s.kit.noise[k < 9 ? k : k + 9];
        } else {
          v = a.st.goal[(size_t)env * 23 + (k - 23)];
        }
        s.kit.obs[k] = v;
        if (live) {
          a.out.obs[row * 46 + k] = v;
          a.st.last_obs[(size_t)env * 46 + k] = v;
          if (k < 9) a.st.last_qp_robot[(size_t)env * 9 + k] = v;
        }
      }
      fence();
      if (sub == 0 && live) {                           // kitchen.py:141-183 (kitchen_reward_kernel)
        const double* o = s.kit.obs;
        const double dist = kit_norm_diff(o + 9, o + 32, 14);
        double r = -10 * dist;
        const int start[8] = {9, 11, 13, 15, 17, 19, 20, 22}, len[8] = {2, 2, 2, 2, 2, 1, 2, 1};
        bool reaching = false;
        for (int c = 0; c < 8; ++c) {
          if (kit_norm_diff(o + start[c], o + start[c] + 23, len[c]) < len[c] * 0.01) r += 1;
          else if (!reaching) {
            reaching = true;
            r += -0.5 * kit_norm_diff(s.mocap, s.kit.sites[c], 3);
          }
        }
        a.out.reward[row] = r;
        a.out.success[row] = dist <= 0.3;
      }
    }
    ++steps;
    if (sub == 0 && live) {
      if (a.out.status) a.out.status[row] = failed ? EARL_STEP_DIVERGED : 0;
      a.out.done[row] = (cfg.horizon > 0 && steps >= cfg.horizon) ? 1 : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");    // the next step reads last_qp_robot (and, after a failure, the state rows) back through global memory
    fence();
  }
#ifdef EARL_PHYS_PROF
  if (lane == 0 && blockIdx.x * WPB + wave < 4096) g_wave_cycles[blockIdx.x * WPB + wave] = __builtin_readcyclecounter() - wave_t0;
#endif
  if (sub == 0 && live) a.st.steps_since_reset[env] = steps;
}
