// physics_env_minitaur.h -- the minitaur env kernel: reset incl. its settle steps and the fused rollout, on the tree-structured timestep of minitaur_stepper.h or the generic one
// A section of csrc/physics.hip (included there, inside its anonymous namespace, after the stepper): split out in round 5 so that a change to one env's kernels
// recompiles only the translation units that hold them (csrc/Makefile lists the headers per unit).

// ------------------------------------------------------------------------------------------------ minitaur env (include/earl_physics.h; physics_mt.hip)
// One launch = T env steps (or the reset incl. its settle steps) of every env: 32 lanes per env, two envs per wave; lanes 0-7 of a group are also
// the eight MOTORS (Minitaur.ApplyAction per timestep: velocity-limited command, DC-motor model, overheat protection -- csrc/minitaur_device.h),
// whose counters and flags live in those lanes' registers between timesteps.  Reference of every expression: oracle/minitaur_oracle.py.
struct MinitaurArgs {
  const void* m;
  const earl_collision_model* col;
  earl_minitaur_cfg cfg;
  earl_minitaur_state st;
  earl_minitaur_out out;
  const float* action; int T;
  const uint8_t* mask; double* reset_obs;
  int solo;                      // as KitchenRolloutArgs::solo
};
__device__ __forceinline__ double mt_draw(const earl_minitaur_cfg& cfg, const uint32_t stream, const int env, const uint64_t counter) {
  const earl::U4 b = earl::philox4x32_10(earl::U4{stream, (uint32_t)(cfg.env_offset + env), (uint32_t)counter, (uint32_t)(counter >> 32)},
                                         (uint32_t)cfg.seed, (uint32_t)(cfg.seed >> 32));
  return earl::u01(b.x, b.y);
}
// ARROW: the timestep written on the model's tree (minitaur_stepper.h: substep_mt, the product path) or the generic substep<22> above (kept for
// comparison: earl_debug_set_minitaur_stepper(0); same numbers to rounding)
#ifndef EARL_MT_WPB
#define EARL_MT_WPB 4            // wavefronts per workgroup of the tree-structured kernels
#endif
#ifndef EARL_MT_BLOCKS
#define EARL_MT_BLOCKS 1         // ... and workgroups per CU the register budget is set for (2 = two waves per SIMD, 256 registers each: spills 1.4 KB per lane and runs 1.5 x slower, tools/bench_mt_variant.py)
#endif
#ifndef EARL_MT_DUO_DEFAULT
#define EARL_MT_DUO_DEFAULT -1   // g_mt_duo at start-up: -1 = the launcher picks the rollout kernel by batch size (mt_use_duo), 0 = always the one-wave kernel, 1 = the two-waves-per-SIMD
                                 // kernel (minitaur_duo_kernel below) for every packed launch.  The two kernels return the same bits (tests/test_minitaur_gpu.py), so the choice is
                                 // about speed only: 16 resident envs per CU against 8, a round of the two-wave kernel taking 1.6 x a round of the one-wave kernel
#endif
template <bool ARROW> constexpr int mt_wpb() { return ARROW ? EARL_MT_WPB : Lim<22>::WPB; }
template <bool RESET, bool ARROW>
__global__ __launch_bounds__(64 * mt_wpb<ARROW>(), ARROW ? EARL_MT_BLOCKS : 1) void minitaur_kernel(const MinitaurArgs a) {
#pragma clang fp contract(off)
  constexpr int NV = 22, LPE = 32, EPW = 64 / LPE, WPB = mt_wpb<ARROW>();
  using SH = std::conditional_t<ARROW, SharedMT, Shared<NV>>;
  __shared__ alignas(16) typename ModelOf<NV>::T m;
  __shared__ alignas(16) BlkTable<Lim<NV>::MB, Lim<NV>::KBT> bt;
  __shared__ alignas(16) SH sh[EPW * WPB];
  __shared__ alignas(16) std::conditional_t<ARROW, PairTabMT, char> ptab;
  stage_blocks(bt, a.col);
  stage_kb<NV>(bt, a.m, a.col);
  if constexpr (ARROW) stage_pairs_mt(ptab, a.col);
  stage_model(m, a.m);                                  // (ends with the workgroup barrier)
  const earl_minitaur_cfg& cfg = a.cfg;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), sub = lane % LPE, grp = lane / LPE, n = cfg.n;
  if (a.solo == 2 && wave != 0) return;                 // (after stage_model's barrier, the last one of the kernel)
  const int env_raw = a.solo == 2 ? (int)blockIdx.x : (a.solo == 1 ? (int)(blockIdx.x * WPB + wave) : (int)((blockIdx.x * WPB + wave) * EPW + grp));
  const bool in_batch = env_raw < n && (a.solo == 0 || grp == 0);
  const int env = env_raw < n ? env_raw : n - 1;        // idle groups shadow the last env (solo: their wave-mate's) and store nothing
  const bool live = in_batch && (!RESET || !a.mask || a.mask[env] != 0);      // (a reset leaves the envs outside the mask alone: their groups compute and discard)
  SH& s = sh[wave * EPW + grp];
#ifdef EARL_MT_DEBUG
  if constexpr (ARROW) { if (sub == 0) { s.dbg_env = env < 4096 ? env : 4095; s.dbg_ts = 0; } }
#endif
#ifdef EARL_PHYS_PROF
  const unsigned long long wave_t0 = __builtin_readcyclecounter();
#endif
  const double ctrl0[EARL_MAXACT] = {0, 0, 0, 0};
  auto timestep = [&](const bool warm) {
    if constexpr (ARROW) {
      // (the lane index passes through an empty asm: everything derived from it -- the lane's rows of the model tables, its LDS addresses -- is then
      // read / recomputed inside the timestep instead of being hoisted out of the rollout loop into registers that live across the whole kernel and
      // end up in scratch memory; see sawyer_rollout_kernel)
      int sub_ = sub;
      asm volatile("" : "+v"(sub_));
      __builtin_assume(sub_ >= 0 && sub_ < LPE);
      substep_mt<true>(s, m, bt, ptab, sub_, grp, warm, nullptr);
    }
    else substep<NV, LPE, true>(s, m, bt, a.col, sub, grp, Q4{1, 0, 0, 0}, ctrl0, warm, nullptr, nullptr);
  };
  const int mi = sub < 8 ? sub : 7;                     // this lane's motor
  const int mdof = cfg.motor_dof[mi];
  const double mdir = cfg.motor_dir[mi];
  const double lim = m.dt * cfg.motor_velocity_limit;
  double voltage, viscous, goal0, goal1;
  double ms0 = 1.0, ms1 = 1.0, ms2 = 1.0, fmu = -1.0;   // mass factors (root body, upper links, lower links), foot friction: motor_param[2..5]
  int oh; bool en; double obs_t;                        // motor lanes: overheat counter, enabled flag, observed torque of the newest ApplyAction
  if constexpr (RESET) {
    // GoalConditionedMinitaurBulletEnv.reset (minitaur_gym_env.py:476-479, 222-270): goal, [UPSTREAM randomizer] battery voltage and viscous damping, pose
    int gi = (int)(mt_draw(cfg, 0x4D00u, env, cfg.counter) * (double)cfg.n_goals);
    gi = gi >= cfg.n_goals ? cfg.n_goals - 1 : gi;
    goal0 = cfg.goal_table[2 * gi]; goal1 = cfg.goal_table[2 * gi + 1];
    // MinitaurEnvRandomizer.randomize_env [UPSTREAM] through Minitaur.SetBatteryVoltage / SetMotorViscousDamping / SetBaseMass / SetLegMasses / SetFootFriction
    // (minitaur.py:468-508); include/earl_physics.h: earl_minitaur_cfg.randomize
    voltage = (cfg.randomize & 1) ? 14.8 + (16.8 - 14.8) * mt_draw(cfg, 0x4D01u, env, cfg.counter) : 16.0;
    viscous = (cfg.randomize & 1) ? 0.01 * mt_draw(cfg, 0x4D02u, env, cfg.counter) : 0.0;
    if (cfg.randomize & 2) {
      const int root = m.ball_dof + 2;
      const double leg = cfg.leg_mass * (1.0 + cfg.leg_mass_err[0] + (cfg.leg_mass_err[1] - cfg.leg_mass_err[0]) * mt_draw(cfg, 0x4D04u, env, cfg.counter));
      const double motor = cfg.motor_mass * (1.0 + cfg.leg_mass_err[0] + (cfg.leg_mass_err[1] - cfg.leg_mass_err[0]) * mt_draw(cfg, 0x4D05u, env, cfg.counter));
      ms0 = 1.0 + cfg.base_mass_err[0] + (cfg.base_mass_err[1] - cfg.base_mass_err[0]) * mt_draw(cfg, 0x4D03u, env, cfg.counter);
      ms1 = (motor + leg) / m.mass[root + 1];
      ms2 = leg / m.mass[root + 2];
    }
    if (cfg.randomize & 4) fmu = cfg.foot_friction[0] + (cfg.foot_friction[1] - cfg.foot_friction[0]) * mt_draw(cfg, 0x4D06u, env, cfg.counter);
    load_state<NV>(s, m, cfg.reset_qpos, a.st.qvel + (size_t)env * NV, sub);
    if (sub < NV) s.qv[sub] = 0.0;
    oh = 0; en = true; obs_t = 0.0;
  } else {
    goal0 = a.st.goal[(size_t)env * 2]; goal1 = a.st.goal[(size_t)env * 2 + 1];
    const double* mp = a.st.motor_param + (size_t)env * 6;
    voltage = mp[0]; viscous = mp[1]; ms0 = mp[2]; ms1 = mp[3]; ms2 = mp[4]; fmu = mp[5];
    load_state<NV>(s, m, a.st.qpos + (size_t)env * m.nq, a.st.qvel + (size_t)env * NV, sub);
    oh = a.st.overheat[(size_t)env * 8 + mi]; en = a.st.motor_enabled[(size_t)env * 8 + mi] != 0; obs_t = a.st.observed_torque[(size_t)env * 8 + mi];
  }
  if (sub < NV) s.xt.ext[sub] = 0.0;
  if (sub == 0) { s.xt.mscale[0] = ms0; s.xt.mscale[1] = ms1; s.xt.mscale[2] = ms2; s.xt.foot_mu = fmu; s.xt.motor_volt = voltage; s.xt.motor_visc = viscous; }
  fence();
  // Minitaur.ApplyAction (minitaur.py:326-390) of motor `mi`: the command clipped to what the velocity limit allows in one timestep, the DC-motor
  // model, overheat protection, torque x motor direction -> s.xt.ext[dof]
  auto apply_action = [&](const double cmd) {
    if (sub < 8) {
      const double q = s.qp[mdof] * mdir, qd = s.qv[mdof] * mdir;
      const double c = earl::mt_clipd(cmd, q - lim, q + lim);
      double act, obs;
      earl::mt_motor_torque(cfg.motor_kp, cfg.motor_kd, s.xt.motor_volt, s.xt.motor_visc, false, c, q, qd, act, obs);
      oh = fabs(act) > cfg.overheat_torque ? oh + 1 : 0;
      if (oh > cfg.overheat_steps) en = false;
      obs_t = obs;
      s.xt.ext[mdof] = en ? act * mdir : 0.0;
    }
    fence();
  };
  // GetObservation + goal (minitaur.py:300-324, minitaur_gym_env.py:541-546): lane k holds entry k of the 32
  auto observe = [&]() -> double {
    if (sub < 8) s.kit.obs[16 + sub] = obs_t;
    fence();
    double v;
    if (sub < 8) v = s.qp[mdof] * mdir;
    else if (sub < 16) v = s.qv[cfg.motor_dof[sub - 8]] * cfg.motor_dir[sub - 8];
    else if (sub < 24) v = s.kit.obs[sub];
    else if (sub < 28) v = s.bq[sub == 27 ? 0 : sub - 23];       // Bullet's (x, y, z, w)
    else if (sub < 30) v = s.qp[sub - 28];
    else v = sub == 30 ? goal0 : goal1;
    fence();
    s.kit.obs[sub] = v;
    fence();
    return v;
  };
  if constexpr (RESET) {
    const double half_pi = 3.141592653589793 / 2;
    for (int ts = 0; ts < cfg.settle_steps; ++ts) {       // minitaur_gym_env.py:265-269
      apply_action(half_pi);
      timestep(ts > 0);
    }
    const double v = observe();
    if (live) {
      store_state<NV>(s, m, a.st.qpos + (size_t)env * m.nq, a.st.qvel + (size_t)env * NV, sub);
      if (a.reset_obs) a.reset_obs[(size_t)env * 32 + sub] = v;
      if (a.st.last_obs) a.st.last_obs[(size_t)env * 32 + sub] = v;
      if (sub < 8) {
        a.st.overheat[(size_t)env * 8 + sub] = oh; a.st.motor_enabled[(size_t)env * 8 + sub] = en ? 1 : 0; a.st.observed_torque[(size_t)env * 8 + sub] = obs_t;
      }
      if (sub == 0) {
        a.st.goal[(size_t)env * 2] = goal0; a.st.goal[(size_t)env * 2 + 1] = goal1;
        double* mp = a.st.motor_param + (size_t)env * 6;
        mp[0] = s.xt.motor_volt; mp[1] = s.xt.motor_visc; mp[2] = s.xt.mscale[0]; mp[3] = s.xt.mscale[1]; mp[4] = s.xt.mscale[2]; mp[5] = s.xt.foot_mu;
        if (a.st.steps_since_reset) a.st.steps_since_reset[env] = 0;
        if (a.st.steps_since_goal_change) a.st.steps_since_goal_change[env] = 0;
      }
    }
  } else {
    int steps = a.st.steps_since_reset ? a.st.steps_since_reset[env] : 0;
    const int gcf = a.st.steps_since_goal_change ? cfg.goal_change_frequency : 0;
    int sgc = gcf > 0 ? a.st.steps_since_goal_change[env] : 0;
    for (int t = 0; t < a.T; ++t) {
      const size_t row = (size_t)t * n + env;
      double a64[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) a64[k] = earl::mt_clipd((double)a.action[row * 8 + k], -1.01, 1.01);      // (the front end raises beyond the reference's bound)
      const double cmd = earl::mt_leg_to_motor(a64, mi);              // ConvertFromLegModel
      for (int ts = 0; ts < cfg.num_substeps; ++ts) {                  // minitaur_gym_env.py:321-323
        apply_action(cmd);
        timestep(ts > 0);
      }
      const bool bad_lane = (sub < NV && !(fabs(s.qp[sub]) < EARL_BAD_VALUE && fabs(s.qv[sub]) < EARL_BAD_VALUE)) || (sub < 4 && !(fabs(s.bq[sub]) < EARL_BAD_VALUE));
      const bool failed = group_any<LPE>(bad_lane, grp);
      ++steps;
      double v;
      if (failed) {
        // rolled back to the env's last stable state (the rows in HBM); the row carries the last stable observation, reward 0
        load_state<NV>(s, m, a.st.qpos + (size_t)env * m.nq, a.st.qvel + (size_t)env * NV, sub);
        oh = a.st.overheat[(size_t)env * 8 + mi]; en = a.st.motor_enabled[(size_t)env * 8 + mi] != 0; obs_t = a.st.observed_torque[(size_t)env * 8 + mi];
        v = t > 0 ? a.out.obs[(row - n) * 32 + sub] : (a.st.last_obs ? a.st.last_obs[(size_t)env * 32 + sub] : NAN);
        if (live) {
          a.out.obs[row * 32 + sub] = v;
          if (sub == 0) {
            a.out.reward[row] = 0.0; a.out.success[row] = 0;
            if (a.st.fail_count) a.st.fail_count[env] += 1;
          }
        }
        fence();
      } else {
        v = observe();
        {
          // the orientation quaternion as the next launch's load_state would read it back from the row stored below (re-normalised, the same
          // expression): a fused rollout and T single-step launches then walk through the same bits
          const double qn = renormalised_quat_entry<NV>(s, sub);
          fence();
          if (live) store_state<NV>(s, m, a.st.qpos + (size_t)env * m.nq, a.st.qvel + (size_t)env * NV, sub);
          if (sub < 4) s.bq[sub] = qn;
          fence();
        }
        if (live) {
          if (sub < 8) {
            a.st.overheat[(size_t)env * 8 + sub] = oh; a.st.motor_enabled[(size_t)env * 8 + sub] = en ? 1 : 0; a.st.observed_torque[(size_t)env * 8 + sub] = obs_t;
          }
          a.out.obs[row * 32 + sub] = v;
          if (sub == 0) {                                 // _reward (minitaur_gym_env.py:505-521) = compute_reward (:529-535) on this observation; is_successful :495-503
            const double* o = s.kit.obs;
            const double xd = o[28] - goal0, yd = o[29] - goal1;
            double dotp = 0.0;
#pragma unroll
            for (int k = 0; k < 8; ++k) dotp = fma(o[16 + k], o[8 + k], dotp);
            a.out.reward[row] = cfg.distance_weight * (-fabs(xd) - fabs(yd)) - cfg.energy_weight * (fabs(dotp) * m.dt);
            a.out.success[row] = sqrt(xd * xd + yd * yd) < cfg.success_radius;
          }
        }
      }
      if (sub == 0 && live) {
        if (a.out.status) a.out.status[row] = failed ? EARL_STEP_DIVERGED : 0;
        a.out.done[row] = (cfg.horizon > 0 && steps >= cfg.horizon) ? 1 : 0;
      }
      if (gcf > 0 && ++sgc >= gcf) {                      // LifelongWrapper.step (lifelong_wrapper.py:36-42): new goal, the observation re-read with it
        sgc = 0;
        int gi = (int)(mt_draw(cfg, 0xFFFEu, env, cfg.step_counter + (uint64_t)t) * (double)cfg.n_goals);
        gi = gi >= cfg.n_goals ? cfg.n_goals - 1 : gi;
        goal0 = cfg.goal_table[2 * gi]; goal1 = cfg.goal_table[2 * gi + 1];
        if (live && sub >= 30) a.out.obs[row * 32 + sub] = sub == 30 ? goal0 : goal1;
        if (live && sub == 0) { a.st.goal[(size_t)env * 2] = goal0; a.st.goal[(size_t)env * 2 + 1] = goal1; }
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");   // a later step of this launch may read this row / the state rows back (failure guard)
      fence();
    }
    if (live) {
      if (sub == 0) {
        if (a.st.steps_since_reset) a.st.steps_since_reset[env] = steps;
        if (gcf > 0) a.st.steps_since_goal_change[env] = sgc;
      }
      if (a.st.last_obs && a.T > 0) a.st.last_obs[(size_t)env * 32 + sub] = a.out.obs[((size_t)(a.T - 1) * n + env) * 32 + sub];
    }
#ifdef EARL_PHYS_PROF
    if (lane == 0 && blockIdx.x * WPB + wave < 4096) g_wave_cycles[blockIdx.x * WPB + wave] = __builtin_readcyclecounter() - wave_t0;
#endif
  }
}

// ------------------------------------------------------------------------------------------------ two waves per SIMD by ROLE (round 6)
// The one-wave kernel above needs all 512 registers of a SIMD lane (256 + 256 accumulation registers used as spill space): one wave per SIMD, the vector ALU issuing in
// half of its cycles.  Capped at 256 registers it spills 1.2 KB per lane and runs 1.7 x slower (profiles/r06_minitaur_two_waves_per_simd.txt).  What does fit 256 registers is
// HALF a timestep: the dynamics half (frames, bounding and pair tests, mass matrix, bias forces, closure rows: substep_mt<.., 1>) and the solver half (active-set passes: substep_mt<.., 2>; the integration runs at the head of the first-half wave's next visit) are each other's only long-lived register state.  So a workgroup is EIGHT waves, two per SIMD: waves 0 - 3 run first halves, waves
// 4 - 7 second halves, wave p and wave p + 4 work as a pair on TWO env pairs (four envs) alternately -- while A runs the first half of timestep k of env pair X, B runs the
// second half of timestep k of env pair Y (whose first half A finished in the slot before); one barrier of the PAIR per slot (a flag each in LDS).  16 envs per CU: 4096 envs are ONE round of the
// chip instead of two.  An env's per-step state (motor counters, command, goal, wrapper counters) lives in its LDS block (SharedMTData::ev) between the visits of wave A,
// which also runs everything around the timesteps (action fetch and leg model, motor model, observation, reward, state rows).  Same expressions as the one-wave kernel.
constexpr int MT_DUO_PAIRS = 4;
__global__ __launch_bounds__(128 * MT_DUO_PAIRS, 1) void minitaur_duo_kernel(const MinitaurArgs a) {
#pragma clang fp contract(off)
  constexpr int NV = 22, LPE = 32, EPW = 64 / LPE, NP = MT_DUO_PAIRS;
  __shared__ alignas(16) typename ModelOf<NV>::T m;
  __shared__ alignas(16) BlkTable<Lim<NV>::MB, Lim<NV>::KBT> bt;
  __shared__ alignas(16) SharedMT sh[NP * 2 * EPW];
  __shared__ alignas(16) PairTabMT ptab;
  __shared__ int slots_done[NP][2];                     // per pair and role: slots finished (the pair's own barrier; see the slot loop)
  if (threadIdx.x < 2 * NP) (&slots_done[0][0])[threadIdx.x] = 0;
  stage_blocks(bt, a.col);
  stage_kb<NV>(bt, a.m, a.col);
  stage_pairs_mt(ptab, a.col);
  stage_model(m, a.m);                                  // (ends with a workgroup barrier)
  const earl_minitaur_cfg& cfg = a.cfg;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), sub = lane % LPE, grp = lane / LPE, n = cfg.n;
  const int pair = wave & (NP - 1);
  const bool role_b = wave >= NP;                       // (waves w and w + 4 of a workgroup land on the same SIMD: every SIMD hosts one wave of each role)
  // (measured and left out: issue priority for the solver half -- the longer one -- 138.7 -> 147.5 ms per bench launch, for the dynamics half 139.9; s_sleep 1 / 32 in the pair barrier 139.3 / 140.2)
  // (this lane's motor constants are re-read where they are used, from a lane index the compiler cannot follow: hoisted out of the slot loop they sat in registers across both
  // halves of the timestep and were spilled around them)
  const int NS = cfg.num_substeps, TT = a.T * NS;       // timesteps per env of this launch
  const int gcf = a.st.steps_since_goal_change ? cfg.goal_change_frequency : 0;
  auto env_of = [&](const int q, const int grp_) { return (int)((blockIdx.x * NP + pair) * 2 + q) * EPW + grp_; };
  // ---- both slots of this pair: state rows -> LDS (wave A; wave B waits at the first barrier)
  if (!role_b) {
    for (int q = 0; q < 2; ++q) {
      const int env_raw = env_of(q, grp), env = env_raw < n ? env_raw : n - 1;
      SharedMT& s = sh[(pair * 2 + q) * EPW + grp];
      const double* mp = a.st.motor_param + (size_t)env * 6;
      load_state<NV>(s, m, a.st.qpos + (size_t)env * m.nq, a.st.qvel + (size_t)env * NV, sub);
#ifdef EARL_MT_DEBUG
      if (sub == 0) { s.dbg_env = env < 4096 ? env : 4095; s.dbg_ts = 0; }
#endif
      if (sub < NV) s.xt.ext[sub] = 0.0;
      if (sub < 8) {
        s.ev.oh[sub] = a.st.overheat[(size_t)env * 8 + sub]; s.ev.en[sub] = a.st.motor_enabled[(size_t)env * 8 + sub] != 0 ? 1 : 0;
        s.ev.obs_t[sub] = a.st.observed_torque[(size_t)env * 8 + sub]; s.ev.cmd[sub] = 0.0;
      }
      if (sub == 0) {
        s.xt.mscale[0] = mp[2]; s.xt.mscale[1] = mp[3]; s.xt.mscale[2] = mp[4]; s.xt.foot_mu = mp[5]; s.xt.motor_volt = mp[0]; s.xt.motor_visc = mp[1];
        s.ev.goal[0] = a.st.goal[(size_t)env * 2]; s.ev.goal[1] = a.st.goal[(size_t)env * 2 + 1];
        s.ev.steps = a.st.steps_since_reset ? a.st.steps_since_reset[env] : 0;
        s.ev.sgc = gcf > 0 ? a.st.steps_since_goal_change[env] : 0;
      }
    }
    fence();
  }
  // (wave A) what stands between the last timestep of env step t and the first of env step t + 1 of slot q: the tail of minitaur_kernel's step loop
  auto finish_step = [&](SharedMT& s, const int env, const bool live, const int t, const int sub, const int grp) {      // (sub, grp: the caller's laundered lane indices, see the slot loop)
    const size_t row = (size_t)t * n + env;
    const bool bad_lane = (sub < NV && !(fabs(s.qp[sub]) < EARL_BAD_VALUE && fabs(s.qv[sub]) < EARL_BAD_VALUE)) || (sub < 4 && !(fabs(s.bq[sub]) < EARL_BAD_VALUE));
    const bool failed = group_any<LPE>(bad_lane, grp);
    const int steps = s.ev.steps + 1;
    double goal0 = s.ev.goal[0], goal1 = s.ev.goal[1];
    double v;
    if (failed) {
      load_state<NV>(s, m, a.st.qpos + (size_t)env * m.nq, a.st.qvel + (size_t)env * NV, sub);
      if (sub < 8) {
        s.ev.oh[sub] = a.st.overheat[(size_t)env * 8 + sub]; s.ev.en[sub] = a.st.motor_enabled[(size_t)env * 8 + sub] != 0 ? 1 : 0; s.ev.obs_t[sub] = a.st.observed_torque[(size_t)env * 8 + sub];
      }
      v = t > 0 ? a.out.obs[(row - n) * 32 + sub] : (a.st.last_obs ? a.st.last_obs[(size_t)env * 32 + sub] : NAN);
      if (live) {
        a.out.obs[row * 32 + sub] = v;
        if (sub == 0) {
          a.out.reward[row] = 0.0; a.out.success[row] = 0;
          if (a.st.fail_count) a.st.fail_count[env] += 1;
        }
      }
      fence();
    } else {
      // GetObservation + goal (minitaur.py:300-324, minitaur_gym_env.py:541-546): lane k holds entry k of the 32
      if (sub < 8) s.kit.obs[16 + sub] = s.ev.obs_t[sub];
      fence();
      if (sub < 8) v = s.qp[cfg.motor_dof[sub]] * cfg.motor_dir[sub];
      else if (sub < 16) v = s.qv[cfg.motor_dof[sub - 8]] * cfg.motor_dir[sub - 8];
      else if (sub < 24) v = s.kit.obs[sub];
      else if (sub < 28) v = s.bq[sub == 27 ? 0 : sub - 23];       // Bullet's (x, y, z, w)
      else if (sub < 30) v = s.qp[sub - 28];
      else v = sub == 30 ? goal0 : goal1;
      fence();
      s.kit.obs[sub] = v;
      fence();
      {
        const double qn = renormalised_quat_entry<NV>(s, sub);
        fence();
        if (live) store_state<NV>(s, m, a.st.qpos + (size_t)env * m.nq, a.st.qvel + (size_t)env * NV, sub);
        if (sub < 4) s.bq[sub] = qn;
        fence();
      }
      if (live) {
        if (sub < 8) {
          a.st.overheat[(size_t)env * 8 + sub] = s.ev.oh[sub]; a.st.motor_enabled[(size_t)env * 8 + sub] = s.ev.en[sub] ? 1 : 0; a.st.observed_torque[(size_t)env * 8 + sub] = s.ev.obs_t[sub];
        }
        a.out.obs[row * 32 + sub] = v;
        if (sub == 0) {                                 // _reward (minitaur_gym_env.py:505-521) = compute_reward (:529-535) on this observation; is_successful :495-503
          const double* o = s.kit.obs;
          const double xd = o[28] - goal0, yd = o[29] - goal1;
          double dotp = 0.0;
#pragma unroll
          for (int k = 0; k < 8; ++k) dotp = fma(o[16 + k], o[8 + k], dotp);
          a.out.reward[row] = cfg.distance_weight * (-fabs(xd) - fabs(yd)) - cfg.energy_weight * (fabs(dotp) * m.dt);
          a.out.success[row] = sqrt(xd * xd + yd * yd) < cfg.success_radius;
        }
      }
    }
    if (sub == 0 && live) {
      if (a.out.status) a.out.status[row] = failed ? EARL_STEP_DIVERGED : 0;
      a.out.done[row] = (cfg.horizon > 0 && steps >= cfg.horizon) ? 1 : 0;
    }
    int sgc = s.ev.sgc;
    fence();
    if (gcf > 0 && ++sgc >= gcf) {                      // LifelongWrapper.step (lifelong_wrapper.py:36-42): new goal, the observation re-read with it
      sgc = 0;
      int gi = (int)(mt_draw(cfg, 0xFFFEu, env, cfg.step_counter + (uint64_t)t) * (double)cfg.n_goals);
      gi = gi >= cfg.n_goals ? cfg.n_goals - 1 : gi;
      goal0 = cfg.goal_table[2 * gi]; goal1 = cfg.goal_table[2 * gi + 1];
      if (live && sub >= 30) a.out.obs[row * 32 + sub] = sub == 30 ? goal0 : goal1;
      if (live && sub == 0) { a.st.goal[(size_t)env * 2] = goal0; a.st.goal[(size_t)env * 2 + 1] = goal1; }
    }
    if (sub == 0) { s.ev.steps = steps; s.ev.sgc = sgc; s.ev.goal[0] = goal0; s.ev.goal[1] = goal1; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");   // a later step of this launch may read this row / the state rows back (failure guard)
    fence();
  };
  // Slot j: wave A works on (slot q = j & 1, timestep j >> 1), wave B on the slot-timestep A finished in slot j - 1.  One barrier per slot.
#ifdef EARL_PHYS_PROF
  const unsigned long long duo_t0 = __builtin_readcyclecounter();
  unsigned long long duo_wait = 0;
#endif
  for (int j = 0; j <= 2 * TT + 1; ++j) {
#ifdef EARL_PHYS_PROF
    const unsigned long long slot_t0 = __builtin_readcyclecounter();
#endif
    if (!role_b) {
      // (the lane's indices pass through an empty asm once per slot: everything derived from them in the code around the timesteps -- rows of the kernel's tables, addresses of
      // the env's rows in HBM -- is then worked out where it is used instead of being hoisted out of the slot loop, held across both halves of the timestep, spilled, and reloaded
      // from scratch memory with a wait each: 33 reloads per slot)
      int lane_ = lane;
      asm volatile("" : "+v"(lane_));
      const int sub = lane_ % LPE, grp = lane_ / LPE;
      const int q = j & 1, ts = j >> 1, t = ts / NS, k = ts - t * NS;
      const int env_raw = env_of(q, grp), env = env_raw < n ? env_raw : n - 1;  // idle groups shadow the last env and store nothing
      const bool live = env_raw < n;
      SharedMT& s = sh[(pair * 2 + q) * EPW + grp];
      if (ts > 0) {                                     // K10 of this slot's timestep before: wave B left the solution in s.aprev (1.7 k cycles off the longer half)
        const bool isroot = sub < 6, ishinge = sub >= 8 && sub < 24, isl = isroot || ishinge;
        const int l = isroot ? sub : (ishinge ? sub - 2 : NV - 1);
        const double al = s.aprev[l], qd = s.qv[l], ql = s.qp[l];
        const Q4 Qb = ldq(s.bq);
        integrate_mt(s, m, sub, isl, l, m.dt, al, qd, ql, Qb);
      }
      if (k == 0) {
        if (t > 0) finish_step(s, env, live, t - 1, sub, grp);
        if (t < a.T) {                                  // ConvertFromLegModel of env step t's action -> this motor's command, kept for the step's timesteps
          const size_t row = (size_t)t * n + env;
          double a64[8];
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) a64[kk] = earl::mt_clipd((double)a.action[row * 8 + kk], -1.01, 1.01);      // (the front end raises beyond the reference's bound)
          if (sub < 8) s.ev.cmd[sub] = earl::mt_leg_to_motor(a64, sub);
        }
      }
      if (ts < TT) {
        // Minitaur.ApplyAction (minitaur.py:326-390) of motor `mi`: as minitaur_kernel's apply_action, counters in LDS
        if (sub < 8) {
          int ml = sub;
          asm volatile("" : "+v"(ml));
          const int mdof = cfg.motor_dof[ml & 7];
          const double mdir = cfg.motor_dir[ml & 7], lim = m.dt * cfg.motor_velocity_limit;
          const double qm = s.qp[mdof] * mdir, qdm = s.qv[mdof] * mdir;
          const double c = earl::mt_clipd(s.ev.cmd[sub], qm - lim, qm + lim);
          double act, obs;
          earl::mt_motor_torque(cfg.motor_kp, cfg.motor_kd, s.xt.motor_volt, s.xt.motor_visc, false, c, qm, qdm, act, obs);
          const int oh = fabs(act) > cfg.overheat_torque ? s.ev.oh[sub] + 1 : 0;
          int en = s.ev.en[sub];
          if (oh > cfg.overheat_steps) en = 0;
          s.ev.oh[sub] = oh; s.ev.en[sub] = en; s.ev.obs_t[sub] = obs;
          s.xt.ext[mdof] = en ? act * mdir : 0.0;
        }
        fence();
        int sub_ = sub;
        asm volatile("" : "+v"(sub_));
        __builtin_assume(sub_ >= 0 && sub_ < LPE);
        substep_mt<true, 1>(s, m, bt, ptab, sub_, grp, k > 0, nullptr);
      }
    } else if (j >= 1) {
      const int jj = j - 1, q = jj & 1, ts = jj >> 1;
      if (ts < TT) {
        SharedMT& s = sh[(pair * 2 + q) * EPW + grp];
        int sub_ = sub;
        asm volatile("" : "+v"(sub_));
        __builtin_assume(sub_ >= 0 && sub_ < LPE);
        substep_mt<true, 2>(s, m, bt, ptab, sub_, grp, (ts % NS) > 0, nullptr);
      }
    }
#ifdef EARL_PHYS_PROF
    const unsigned long long slot_t1 = __builtin_readcyclecounter();
#endif
    // The PAIR's barrier (not the workgroup's: the four pairs have nothing to wait for in each other, and a slot lasts as long as its active-set passes): each wave
    // publishes the number of slots it has finished and waits for its partner's to reach the same.  (release / acquire at workgroup scope around the flag: the halves
    // hand their results over through LDS.)
    {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      volatile int* mine = &slots_done[pair][role_b ? 1 : 0];
      volatile int* other = &slots_done[pair][role_b ? 0 : 1];
      if (lane == 0) *mine = j + 1;
      while (*other < j + 1) __builtin_amdgcn_s_sleep(4);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
#ifdef EARL_PHYS_PROF
    PCOUNT(26, slot_t1 - slot_t0); PCOUNT(27, __builtin_readcyclecounter() - slot_t1); PCOUNT(28, 1);      // this wave's work and wait per slot
    duo_wait += __builtin_readcyclecounter() - slot_t1;
#endif
  }
#ifdef EARL_PHYS_PROF
  if (lane == 0 && blockIdx.x * 2 * NP + wave < 2048) {      // every wave's duration and the part of it spent at the pair's barrier (load balance: tools/prof_minitaur.py)
    g_wave_cycles[blockIdx.x * 2 * NP + wave] = __builtin_readcyclecounter() - duo_t0;
    g_wave_cycles[2048 + blockIdx.x * 2 * NP + wave] = duo_wait;
  }
#endif
  if (!role_b) {
    for (int q = 0; q < 2; ++q) {
      const int env_raw = env_of(q, grp), env = env_raw < n ? env_raw : n - 1;
      SharedMT& s = sh[(pair * 2 + q) * EPW + grp];
      if (env_raw < n) {
        if (sub == 0) {
          if (a.st.steps_since_reset) a.st.steps_since_reset[env] = s.ev.steps;
          if (gcf > 0) a.st.steps_since_goal_change[env] = s.ev.sgc;
        }
        if (a.st.last_obs && a.T > 0) a.st.last_obs[(size_t)env * 32 + sub] = a.out.obs[((size_t)(a.T - 1) * n + env) * 32 + sub];
      }
    }
  }
}
