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
template <bool ARROW> constexpr int mt_wpb() { return ARROW ? EARL_MT_WPB : Lim<22>::WPB; }
template <bool RESET, bool ARROW>
__global__ __launch_bounds__(64 * mt_wpb<ARROW>(), ARROW ? EARL_MT_BLOCKS : 1) void minitaur_kernel(const MinitaurArgs a) {
#pragma clang fp contract(off)
  constexpr int NV = 22, LPE = 32, EPW = 64 / LPE, WPB = mt_wpb<ARROW>();
  using SH = std::conditional_t<ARROW, SharedMT, Shared<NV>>;
  __shared__ typename ModelOf<NV>::T m;
  __shared__ BlkTable<Lim<NV>::MB, Lim<NV>::KBT> bt;
  __shared__ SH sh[EPW * WPB];
  __shared__ std::conditional_t<ARROW, PairTabMT, char> ptab;
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
