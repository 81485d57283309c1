/* glue_oracle.c -- CPU restatement of the pure-numpy glue of the physics-backed envs (include/earl_glue.h).
 * TEST INFRASTRUCTURE, NOT PRODUCT (same rules as tabletop_oracle.c).  Pinned by tests/test_glue.py against goldens
 * recorded from the reference's own functions and against the Sawyer demonstrations.
 * Arithmetic probed on numpy 2.2.6 / OpenBLAS 0.3.29: float64 dot of 3 / 8 elements = FMA chain
 * (100000/100000 agree, plain sums fail on 10.7 %); float32 dot = double accumulation of float products;
 * np.interp = slope * (x - xp[j]) + fp[j] with separate roundings (the FMA form fails on 3.7 %). */
#include <math.h>
#include <stddef.h>
#include <stdint.h>

#include "../include/earl_glue.h"

/* sawyer_door.py:173-177 / sawyer_peg.py:301-305: np.linalg.norm(obs[4:7] - obs[11:14]) <= radius */
int oracle_sawyer_sparse_f64(int32_t n, const double* obs, double radius, float* reward, uint8_t* success) {
  for (int32_t i = 0; i < n; ++i) {
    const double* o = obs + 14 * (size_t)i;
    double d0 = o[4] - o[11], d1 = o[5] - o[12], d2 = o[6] - o[13];
    int s = sqrt(fma(d2, d2, fma(d1, d1, d0 * d0))) <= radius;
    if (success) success[i] = (uint8_t)s;
    if (reward) reward[i] = (float)s;
  }
  return EARL_OK;
}
int oracle_sawyer_sparse_f32(int32_t n, const float* obs, double radius, float* reward, uint8_t* success) {
  for (int32_t i = 0; i < n; ++i) {
    const float* o = obs + 14 * (size_t)i;
    float d[3] = {o[4] - o[11], o[5] - o[12], o[6] - o[13]};
    double dot = 0.0;
    for (int k = 0; k < 3; ++k) dot += (double)(d[k] * d[k]);
    int s = (double)sqrtf((float)dot) <= radius; /* numpy 1.22: float32 scalar <= python float compares in double */
    if (success) success[i] = (uint8_t)s;
    if (reward) reward[i] = (float)s;
  }
  return EARL_OK;
}

/* minitaur.py:434-457 ConvertFromLegModel */
int oracle_minitaur_leg_to_motor(int32_t n, const double* action, double* motor_angle) {
  const double pi = 3.141592653589793, quater_pi = pi / 4;
  for (int32_t r = 0; r < n; ++r) {
    const double* a = action + 8 * (size_t)r;
    for (int i = 0; i < 8; ++i) {
      int idx = i / 2;
      double fb = (-1 * quater_pi) * (a[idx + 4] + 1.5);
      double ext = ((i & 1) ? -1.0 : 1.0) * quater_pi * a[idx];
      if (i >= 4) ext = -ext;
      motor_angle[8 * (size_t)r + i] = (pi + fb) + ext;
    }
  }
  return EARL_OK;
}

static double clip(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
static double interp7(double x) { /* np.interp(x, [0,10,..,60], [0,1,1.9,2.45,3.0,3.25,3.5]) for x >= 0 */
  static const double xp[7] = {0, 10, 20, 30, 40, 50, 60}, fp[7] = {0, 1, 1.9, 2.45, 3.0, 3.25, 3.5};
  if (x >= xp[6]) return fp[6];
  int j = (x >= xp[1]) + (x >= xp[2]) + (x >= xp[3]) + (x >= xp[4]) + (x >= xp[5]);
  double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
  return slope * (x - xp[j]) + fp[j];
}
/* motor.py:49-94 */
int oracle_minitaur_motor_torque(int32_t m, const earl_motor_params* p, const double* command, const double* angle,
                                 const double* velocity, double* actual_torque, double* observed_torque) {
  const double R = 0.186, Kt = 0.0954;
  for (int32_t i = 0; i < m; ++i) {
    double pwm = p->torque_control ? command[i] : (-p->kp * (angle[i] - command[i]) - p->kd * velocity[i]);
    pwm = clip(pwm, -1.0, 1.0);
    double observed = clip(Kt * (pwm * p->voltage / R), -5.7, 5.7);
    double vnet = clip(pwm * p->voltage - (Kt + p->viscous_damping) * velocity[i], -50, 50);
    double current = vnet / R;
    double sign = current > 0 ? 1.0 : (current < 0 ? -1.0 : (current == 0 ? 0.0 : current));
    double actual = sign * interp7(fabs(current));
    if (actual_torque) actual_torque[i] = actual;
    if (observed_torque) observed_torque[i] = observed;
  }
  return EARL_OK;
}

/* minitaur_gym_env.py:529-535 compute_reward, :495-503 is_successful */
int oracle_minitaur_reward(int32_t n, const double* obs, double distance_weight, double energy_weight, double time_step,
                           double* reward, uint8_t* success) {
  for (int32_t i = 0; i < n; ++i) {
    const double* o = obs + 32 * (size_t)i;
    double x_dist = o[28] - o[30], y_dist = o[29] - o[31];
    double distance_reward = -fabs(x_dist) - fabs(y_dist);
    double dot = 0.0;
    for (int k = 0; k < 8; ++k) dot = fma(o[16 + k], o[8 + k], dot);
    double energy_reward = fabs(dot) * time_step;
    if (reward) reward[i] = distance_weight * distance_reward - energy_weight * energy_reward;
    if (success) success[i] = (uint8_t)(sqrt(x_dist * x_dist + y_dist * y_dist) < 0.1);
  }
  return EARL_OK;
}


/* Kitchen._get_reward_n_score / is_successful (/root/reference/earl_benchmark/envs/kitchen.py:141-183), the numpy part.
 * np.linalg.norm on float64 = sqrt(ddot(x, x)); OpenBLAS's ddot runs these short vectors through its scalar tail loop,
 * which is compiled with FMA contraction: dot = fma(x_i, x_i, dot) in index order (pinned by tests/golden/kitchen_glue.npz). */
static double norm_diff(const double* a, const double* b, int n) {
  double d = 0.0;
  for (int i = 0; i < n; ++i) {
    const double x = a[i] - b[i];
    d = fma(x, x, d);
  }
  return sqrt(d);
}
void oracle_kitchen_reward(int32_t n, const double* obs, const double* mocap, const double* sites, double* reward, uint8_t* success) {
  static const int start[8] = {9, 11, 13, 15, 17, 19, 20, 22}, len[8] = {2, 2, 2, 2, 2, 1, 2, 1};   /* component_to_state_idx :15-25 */
  for (int32_t i = 0; i < n; ++i) {
    const double* o = obs + (size_t)i * 46;
    const double dist = norm_diff(o + 9, o + 32, 14);
    double r = -10 * dist;
    int reaching = 0;
    for (int c = 0; c < 8; ++c) {
      if (norm_diff(o + start[c], o + start[c] + 23, len[c]) < len[c] * 0.01) r += 1;
      else if (!reaching) {
        reaching = 1;
        r += -0.5 * norm_diff(mocap + (size_t)i * 3, sites + ((size_t)i * 8 + c) * 3, 3);
      }
    }
    if (reward) reward[i] = r;
    if (success) success[i] = dist <= 0.3;
  }
}


/* KitchenV0.step up to do_simulation (/root/reference/earl_benchmark/envs/kitchen_assets/adept_envs/adept_envs/franka/kitchen_multitask_v0.py:91-105)
 * and Robot.step's limits (.../franka/robot/franka_robot.py:172-174 ctrl_position_limits, :259-264 Robot_VelAct.ctrl_velocity_limits);
 * pinned by tests/golden/kitchen_step.npz (recorded from those methods).  The parameter struct is the public one of earl_glue.h. */
static double clipd_(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }
void oracle_kitchen_action(int32_t n, const earl_kitchen_params* p, const double* action, double* mocap, const double* last_qp, double* ctrl) {
  for (int32_t i = 0; i < n; ++i) {
    double a[9];
    for (int k = 0; k < 9; ++k) a[k] = p->act_mid[k] + clipd_(action[(size_t)i * 9 + k], -1.0, 1.0) * p->act_amp[k];   /* :92-95 */
    for (int k = 0; k < 3; ++k)                                                                                         /* :99-102 */
      mocap[(size_t)i * 3 + k] = clipd_(mocap[(size_t)i * 3 + k] + a[k] * p->mocap_range[k], p->mocap_clip_lower[k], p->mocap_clip_upper[k]);
    for (int k = 0; k < 9; ++k) {
      const double v = clipd_(a[k], p->vel_bound[k][0], p->vel_bound[k][1]);                                            /* franka_robot.py:262 */
      ctrl[(size_t)i * 9 + k] = clipd_(last_qp[(size_t)i * 9 + k] + v * p->step_duration, p->pos_bound[k][0], p->pos_bound[k][1]);   /* :263, :173 */
    }
  }
}
/* Robot.get_obs + KitchenV0._get_obs (franka_robot.py:137-168, kitchen_multitask_v0.py:127-139) */
void oracle_kitchen_obs(int32_t n, const earl_kitchen_params* p, const double* qpos, const double* goal, const double* noise, double* obs) {
  for (int32_t i = 0; i < n; ++i)
    for (int k = 0; k < 46; ++k) {
      double v;
      if (k < 23) {
        v = qpos[(size_t)i * 23 + k];
        if (noise) v = v + (p->robot_noise_ratio * p->pos_noise_amp[k]) * noise[(size_t)i * 46 + (k < 9 ? k : k + 9)];
      } else v = goal[(size_t)i * 23 + (k - 23)];
      obs[(size_t)i * 46 + k] = v;
    }
}
