/* earl_glue.h -- C ABI of the pure-numpy glue of EARL's physics-backed envs.
 *
 * The DYNAMICS of sawyer_door / sawyer_peg / minitaur live in MuJoCo / Bullet and are out of scope;
 * what the reference itself computes in numpy around them is restated here for batches in HBM, with the same
 * conventions as earl_tabletop.h (device pointers, caller's stream, negative error codes, no allocation):
 *
 *   earl_sawyer_sparse_*      SawyerDoorV2.is_successful (envs/sawyer_door.py:173-177, radius 0.02) and
 *                             SawyerPegV2.is_successful (envs/sawyer_peg.py:301-305, radius 0.05), and the sparse branch
 *                             of compute_reward (:168-169 / :296-297): ||obs[4:7] - obs[11:14]|| <= radius
 *   earl_minitaur_leg_to_motor   Minitaur.ConvertFromLegModel (envs/minitaur.py:434-457)
 *   earl_minitaur_motor_torque   MotorModel.convert_to_torque (envs/motor.py:49-94)
 *   earl_minitaur_reward         GoalConditionedMinitaurBulletEnv.compute_reward / is_successful
 *                                (envs/minitaur_gym_env.py:529-535, :495-503)
 *   earl_kitchen_reward          Kitchen._get_reward_n_score / compute_reward / is_successful (envs/kitchen.py:141-183):
 *                                the numpy part; what the reference reads from the simulator there (mocap position, the
 *                                eight task-site positions) is an input
 *   earl_kitchen_action          KitchenV0.step up to do_simulation (adept_envs/franka/kitchen_multitask_v0.py:91-105): action clip and
 *                                scale, mocap update; Robot.step -> Robot_VelAct.ctrl_velocity_limits + Robot.ctrl_position_limits
 *                                (adept_envs/franka/robot/franka_robot.py:172-207, :259-264): the position targets handed to
 *                                do_simulation
 *   earl_kitchen_obs             Robot.get_obs + KitchenV0._get_obs (franka_robot.py:137-168, kitchen_multitask_v0.py:127-139): the
 *                                observation with sensor noise, the U(-1, 1) draws being an input (row a18)
 * All fp64 like the reference's numpy code; tested bit-exact against goldens recorded from the reference's own
 * functions (tests/golden/make_golden.py: gen_glue) and against the 2,910 Sawyer demonstration rows.
 */
#ifndef EARL_GLUE_H
#define EARL_GLUE_H
#include <stdint.h>

#include "earl_tabletop.h"

#ifdef __cplusplus
extern "C" {
#endif

/* obs [n,14] float64 (what the env holds): success [n] 0/1, reward [n] = (float)success; either may be NULL */
int earl_sawyer_sparse_f64(int32_t n, const double* obs, double radius, float* reward, uint8_t* success, earl_stream_t stream);
/* obs [n,14] float32 (the demonstration layout): numpy float32 semantics of the same expression */
int earl_sawyer_sparse_f32(int32_t n, const float* obs, double radius, float* reward, uint8_t* success, earl_stream_t stream);

/* action [n,8] (4 extension + 4 swing components of the leg model) -> desired motor angles [n,8] */
int earl_minitaur_leg_to_motor(int32_t n, const double* action, double* motor_angle, earl_stream_t stream);

typedef struct earl_motor_params {
  double kp, kd;             /* MotorModel(kp=1.2, kd=0) */
  double voltage;            /* MOTOR_VOLTAGE 16.0 (set_voltage) */
  double viscous_damping;    /* MOTOR_VISCOUS_DAMPING 0 (set_viscous_damping) */
  int32_t torque_control;    /* torque_control_enabled: command is the pwm itself */
} earl_motor_params;
/* m motors (any shape flattened): command, angle, velocity -> actual_torque, observed_torque */
int earl_minitaur_motor_torque(int32_t m, const earl_motor_params* p, const double* command, const double* angle,
                               const double* velocity, double* actual_torque, double* observed_torque, earl_stream_t stream);

/* obs [n,32] -> reward [n] (distance_weight * distance_reward - energy_weight * energy_reward), success [n] 0/1 */
int earl_minitaur_reward(int32_t n, const double* obs, double distance_weight, double energy_weight, double time_step,
                         double* reward, uint8_t* success, earl_stream_t stream);

/* obs [n,46] = qpos[23] + goal[23]; mocap_pos [n,3]; site_xpos [n,8,3] in the order the reward walks the components:
 * knob1..knob4 (burner0..3), light_site, slide_site, hinge_site2, microhandle_site (kitchen.py:148-155, :15-25).
 * reward [n] float64: -10 |obj - goal| + 1 per component within len * 0.01, - 0.5 |mocap - site| of the FIRST unsolved
 * component; success [n]: |obj - goal| <= 0.3.  Either output may be NULL. */
int earl_kitchen_reward(int32_t n, const double* obs, const double* mocap_pos, const double* site_xpos, double* reward,
                        uint8_t* success, earl_stream_t stream);

/* Joint table of the kitchen robot config (adept_envs/franka/robot/franka_config.xml:17-57, as read by Robot._read_specs_from_config) and
 * the step constants of KitchenV0 (kitchen_multitask_v0.py:40-53, :78-79): filled by earl_kitchen_default_params. */
typedef struct earl_kitchen_params {
  double pos_bound[23][2], vel_bound[23][2], pos_noise_amp[23];
  double act_mid[9], act_amp[9];                 /* 0, 2.0 */
  double mocap_range[3];                         /* 0.01 */
  double mocap_clip_lower[3], mocap_clip_upper[3];
  double step_duration;                          /* skip * timestep = 40 * 0.002 */
  double robot_noise_ratio;                      /* 0.1 */
} earl_kitchen_params;
int earl_kitchen_default_params(earl_kitchen_params* p);   /* host call */

/* action [n,9] float64 (a float32 action is promoted exactly as numpy does: the clip keeps float32, the scaling promotes),
 * mocap_pos [n,3] updated in place, last_qpos_robot [n,9] = qpos_robot of the newest cached (noisy) observation,
 * ctrl [n,9] = the targets passed to do_simulation (MuJoCo itself uses the first nu = 2 and clamps them to the actuators' ctrlrange). */
int earl_kitchen_action(int32_t n, const earl_kitchen_params* p, const double* action, double* mocap_pos, const double* last_qpos_robot,
                        double* ctrl, earl_stream_t stream);
/* qpos [n,23], goal [n,23], noise [n,46] = the four env.np_random.uniform(-1, 1) calls of one get_obs in order (robot qpos 9, robot
 * qvel 9, object qpos 14, object qvel 14), or NULL for no noise (env.initializing): obs [n,46] = noisy robot qpos, noisy object qpos, goal */
int earl_kitchen_obs(int32_t n, const earl_kitchen_params* p, const double* qpos, const double* goal, const double* noise, double* obs,
                     earl_stream_t stream);

/* k draws of U(lo, hi) per env, out [n, k] float64 (numpy's low + (high - low) * u): Philox4x32-10 keyed by `seed`, counter block
 * (stream_id + j / 2, env_offset + env, counter).  Stands in for the reference's global / env.np_random streams (sensor noise of
 * Robot.get_obs, franka_robot.py:137-168; np.random.randint of Kitchen.reset_model, kitchen.py:122-124): parity is defined given the draws,
 * and the draws of an env do not depend on how the batch is sharded. */
int earl_philox_uniform(int32_t n, int32_t k, uint64_t seed, uint64_t counter, int32_t env_offset, uint32_t stream_id, double lo, double hi,
                        double* out, earl_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
