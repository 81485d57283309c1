// glue.hip -- HIP kernels + C ABI of include/earl_glue.h: the pure-numpy glue of the physics-backed envs
// (Sawyer sparse success rule, minitaur leg model / DC-motor model / reward).  One lane per row, fp64 like the
// reference, -ffp-contract=off (explicit fma() only where numpy's BLAS dot fuses).  Elementwise, launch-bound at the
// sizes of the BASELINE configs; they exist so that the glue around a future rigid-body stepper is already at parity.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "../../include/earl_glue.h"
#include "philox.h"
#include "minitaur_device.h"

namespace {
constexpr int kB = 256;

__device__ __forceinline__ double clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

template <typename T>
__global__ __launch_bounds__(kB) void sawyer_sparse_kernel(int n, const T* __restrict__ obs, double radius,
                                                           float* __restrict__ reward, uint8_t* __restrict__ success) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  const T* o = obs + (size_t)i * 14;
  bool s;
  if constexpr (sizeof(T) == 8) {   // sawyer_door.py:173-177: float64 obs, numpy dot = FMA chain
    const double d0 = o[4] - o[11], d1 = o[5] - o[12], d2 = o[6] - o[13];
    s = sqrt(fma(d2, d2, fma(d1, d1, d0 * d0))) <= radius;
  } else {                          // float32 rows (demonstrations): products rounded to float, summed in double
    const float d0 = o[4] - o[11], d1 = o[5] - o[12], d2 = o[6] - o[13];
    const double dot = ((double)(d0 * d0) + (double)(d1 * d1)) + (double)(d2 * d2);
    s = (double)sqrtf((float)dot) <= radius;
  }
  if (success) success[i] = s;
  if (reward) reward[i] = s ? 1.0f : 0.0f;
}

// minitaur.py:434-457 (csrc/minitaur_device.h)
__global__ __launch_bounds__(kB) void leg_to_motor_kernel(int n, const double* __restrict__ action, double* __restrict__ out) {
  const int t = blockIdx.x * kB + threadIdx.x;   // one lane per motor
  if (t >= n * 8) return;
  out[t] = earl::mt_leg_to_motor(action + (size_t)(t >> 3) * 8, t & 7);
}

// motor.py:49-94 (csrc/minitaur_device.h)
__global__ __launch_bounds__(kB) void motor_kernel(int m, earl_motor_params p, const double* __restrict__ command,
                                                   const double* __restrict__ angle, const double* __restrict__ velocity,
                                                   double* __restrict__ actual, double* __restrict__ observed) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= m) return;
  double act, obs;
  earl::mt_motor_torque(p.kp, p.kd, p.voltage, p.viscous_damping, p.torque_control != 0, command[i], angle[i], velocity[i], act, obs);
  if (actual) actual[i] = act;
  if (observed) observed[i] = obs;
}

// minitaur_gym_env.py:529-535, :495-503
__global__ __launch_bounds__(kB) void minitaur_reward_kernel(int n, const double* __restrict__ obs, double dw, double ew,
                                                             double dt, double* __restrict__ reward,
                                                             uint8_t* __restrict__ success) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  const double* o = obs + (size_t)i * 32;
  const double xd = o[28] - o[30], yd = o[29] - o[31];
  const double distance_reward = -fabs(xd) - fabs(yd);
  double dot = 0.0;
#pragma unroll
  for (int k = 0; k < 8; ++k) dot = fma(o[16 + k], o[8 + k], dot);
  if (reward) reward[i] = dw * distance_reward - ew * (fabs(dot) * dt);
  if (success) success[i] = sqrt(xd * xd + yd * yd) < 0.1;
}

// numpy's float64 norm = sqrt(ddot(x, x)); OpenBLAS runs vectors this short through its scalar tail loop, which the
// compiler contracts: dot = fma(x_i, x_i, dot) in index order (probed in the build container, see oracle/glue_oracle.c)
__device__ __forceinline__ double norm_diff(const double* a, const double* b, int n) {
  double d = 0.0;
  for (int i = 0; i < n; ++i) {
    const double x = a[i] - b[i];
    d = fma(x, x, d);
  }
  return sqrt(d);
}

// kitchen.py:141-183
__global__ __launch_bounds__(kB) void kitchen_reward_kernel(int n, const double* __restrict__ obs, const double* __restrict__ mocap,
                                                            const double* __restrict__ sites, double* __restrict__ reward,
                                                            uint8_t* __restrict__ success) {
  const int i = blockIdx.x * kB + threadIdx.x;
  if (i >= n) return;
  const double* o = obs + (size_t)i * 46;
  const double dist = norm_diff(o + 9, o + 32, 14);
  double r = -10 * dist;
  const int start[8] = {9, 11, 13, 15, 17, 19, 20, 22}, len[8] = {2, 2, 2, 2, 2, 1, 2, 1};   // component_to_state_idx :15-25, minus 'arm'
  bool reaching = false;
  for (int c = 0; c < 8; ++c) {
    if (norm_diff(o + start[c], o + start[c] + 23, len[c]) < len[c] * 0.01) r += 1;
    else if (!reaching) {
      reaching = true;
      r += -0.5 * norm_diff(mocap + (size_t)i * 3, sites + ((size_t)i * 8 + c) * 3, 3);
    }
  }
  if (reward) reward[i] = r;
  if (success) success[i] = dist <= 0.3;
}

// KitchenV0.step (kitchen_multitask_v0.py:91-105) + Robot.step's limits (franka_robot.py:172-174, :259-264); one lane per env
__global__ void kitchen_action_kernel(const int n, const earl_kitchen_params p, const double* __restrict__ action, double* __restrict__ mocap,
                                      const double* __restrict__ last_qp, double* __restrict__ ctrl) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double a[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const double x = action[(size_t)i * 9 + k];
    const double c = x < -1.0 ? -1.0 : (x > 1.0 ? 1.0 : x);         // np.clip (NaN propagates)
    a[k] = p.act_mid[k] + c * p.act_amp[k];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const double x = mocap[(size_t)i * 3 + k] + a[k] * p.mocap_range[k];
    mocap[(size_t)i * 3 + k] = x < p.mocap_clip_lower[k] ? p.mocap_clip_lower[k] : (x > p.mocap_clip_upper[k] ? p.mocap_clip_upper[k] : x);
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    const double v = a[k] < p.vel_bound[k][0] ? p.vel_bound[k][0] : (a[k] > p.vel_bound[k][1] ? p.vel_bound[k][1] : a[k]);
    const double x = last_qp[(size_t)i * 9 + k] + v * p.step_duration;
    ctrl[(size_t)i * 9 + k] = x < p.pos_bound[k][0] ? p.pos_bound[k][0] : (x > p.pos_bound[k][1] ? p.pos_bound[k][1] : x);
  }
}

// Robot.get_obs + KitchenV0._get_obs (franka_robot.py:137-168, kitchen_multitask_v0.py:127-139); one lane per obs entry
__global__ void kitchen_obs_kernel(const int n, const earl_kitchen_params p, const double* __restrict__ qpos, const double* __restrict__ goal,
                                   const double* __restrict__ noise, double* __restrict__ obs) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)n * 46) return;
  const int i = (int)(t / 46), k = (int)(t % 46);
  double v;
  if (k < 23) {
    v = qpos[(size_t)i * 23 + k];
    // qp += ratio * amp[:9] * u[0:9];  qp_obj += ratio * amp[-14:] * u[18:32]   (left to right)
    if (noise) v = v + (p.robot_noise_ratio * p.pos_noise_amp[k]) * noise[(size_t)i * 46 + (k < 9 ? k : k + 9)];
  } else {
    v = goal[(size_t)i * 23 + (k - 23)];
  }
  obs[t] = v;
}

int done(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "earl_glue: %s: %s\n", what, hipGetErrorString(e));
    return EARL_ERR_LAUNCH;
  }
  return EARL_OK;
}
inline unsigned blocks(long long n) { return (unsigned)((n + kB - 1) / kB); }
// U(lo, hi) draws keyed by (seed; stream, global env id, counter): k draws per env, numpy's low + (high - low) * u
__global__ void uniform_kernel(int n, int k, uint64_t seed, uint64_t counter, int env_offset, uint32_t stream, double lo, double hi, double* __restrict__ out) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // one Philox block = two draws
  const int pairs = (k + 1) / 2;
  if (idx >= (long long)n * pairs) return;
  const int env = (int)(idx / pairs), j = (int)(idx % pairs);
  const earl::U4 b = earl::philox4x32_10(earl::U4{stream + (uint32_t)j, (uint32_t)(env_offset + env), (uint32_t)counter, (uint32_t)(counter >> 32)},
                                         (uint32_t)seed, (uint32_t)(seed >> 32));
  double* o = out + (size_t)env * k + 2 * j;
  o[0] = lo + (hi - lo) * earl::u01(b.x, b.y);
  if (2 * j + 1 < k) o[1] = lo + (hi - lo) * earl::u01(b.z, b.w);
}

}  // namespace

extern "C" {
int earl_sawyer_sparse_f64(int32_t n, const double* obs, double radius, float* reward, uint8_t* success, earl_stream_t s) {
  if (n < 0 || !obs) return EARL_ERR_ARG;
  if (n == 0) return EARL_OK;
  sawyer_sparse_kernel<double><<<blocks(n), kB, 0, (hipStream_t)s>>>(n, obs, radius, reward, success);
  return done("sawyer_sparse_f64");
}
int earl_sawyer_sparse_f32(int32_t n, const float* obs, double radius, float* reward, uint8_t* success, earl_stream_t s) {
  if (n < 0 || !obs) return EARL_ERR_ARG;
  if (n == 0) return EARL_OK;
  sawyer_sparse_kernel<float><<<blocks(n), kB, 0, (hipStream_t)s>>>(n, obs, radius, reward, success);
  return done("sawyer_sparse_f32");
}
int earl_minitaur_leg_to_motor(int32_t n, const double* action, double* motor_angle, earl_stream_t s) {
  if (n < 0 || !action || !motor_angle) return EARL_ERR_ARG;
  if (n == 0) return EARL_OK;
  leg_to_motor_kernel<<<blocks((long long)n * 8), kB, 0, (hipStream_t)s>>>(n, action, motor_angle);
  return done("leg_to_motor");
}
int earl_minitaur_motor_torque(int32_t m, const earl_motor_params* p, const double* command, const double* angle,
                               const double* velocity, double* actual_torque, double* observed_torque, earl_stream_t s) {
  if (m < 0 || !p || !command || !angle || !velocity) return EARL_ERR_ARG;
  if (m == 0) return EARL_OK;
  motor_kernel<<<blocks(m), kB, 0, (hipStream_t)s>>>(m, *p, command, angle, velocity, actual_torque, observed_torque);
  return done("motor_torque");
}
int earl_minitaur_reward(int32_t n, const double* obs, double distance_weight, double energy_weight, double time_step,
                         double* reward, uint8_t* success, earl_stream_t s) {
  if (n < 0 || !obs) return EARL_ERR_ARG;
  if (n == 0) return EARL_OK;
  minitaur_reward_kernel<<<blocks(n), kB, 0, (hipStream_t)s>>>(n, obs, distance_weight, energy_weight, time_step, reward, success);
  return done("minitaur_reward");
}
int earl_kitchen_default_params(earl_kitchen_params* p) {
  if (!p) return EARL_ERR_ARG;
  // franka_config.xml:17-57 (qpos0 .. qpos22)
  static const double pb[23][2] = {{-2.9, 2.9}, {-1.8, 1.8}, {-2.9, 2.9}, {-3.1, 0.0}, {-2.9, 2.9}, {0.0, 3.8}, {-2.9, 2.9}, {0.0, 0.04}, {0.0, 0.04},
                                   {-.5, 0.0}, {-.5, 0.0}, {-.005, 0.0}, {-.005, 0.0}, {-.005, 0.0}, {-.005, 0.0}, {-.005, 0.0}, {-.005, 0.0},
                                   {-1.5, 1.5}, {-1.5, 1.5}, {-1.5, 1.5}, {-10.57, 10.57}, {-10.57, 10.57}, {-10.57, 10.57}};
  static const double amp[23] = {0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.1, 0.005, 0.005, 0.0005, 0.0005, 0.0005, 0.0005, 0.0005, 0.0005,
                                 0.005, 0.005, 0.005, 0.1, 0.1, 0.1};
  for (int k = 0; k < 23; ++k) {
    p->pos_bound[k][0] = pb[k][0]; p->pos_bound[k][1] = pb[k][1];
    const double vb = k < 9 ? 10.0 : (k < 20 ? 5.0 : 0.5);
    p->vel_bound[k][0] = -vb; p->vel_bound[k][1] = vb;
    p->pos_noise_amp[k] = amp[k];
  }
  for (int k = 0; k < 9; ++k) { p->act_mid[k] = 0.0; p->act_amp[k] = 2.0; }            // kitchen_multitask_v0.py:78-79
  const double lo[3] = {-0.7, -0.1, 1.8}, hi[3] = {0.4, 0.5, 2.6};                         // :49-50
  for (int k = 0; k < 3; ++k) { p->mocap_range[k] = 0.01; p->mocap_clip_lower[k] = lo[k]; p->mocap_clip_upper[k] = hi[k]; }   // :47
  p->step_duration = 40 * 0.002;                                                            // skip * model.opt.timestep (:104-105)
  p->robot_noise_ratio = 0.1;                                                               // :42
  return EARL_OK;
}
int earl_kitchen_action(int32_t n, const earl_kitchen_params* p, const double* action, double* mocap_pos, const double* last_qpos_robot,
                        double* ctrl, earl_stream_t s) {
  if (n < 0 || !p || !action || !mocap_pos || !last_qpos_robot || !ctrl) return EARL_ERR_ARG;
  if (n == 0) return EARL_OK;
  kitchen_action_kernel<<<blocks(n), kB, 0, (hipStream_t)s>>>(n, *p, action, mocap_pos, last_qpos_robot, ctrl);
  return done("kitchen_action");
}
int earl_kitchen_obs(int32_t n, const earl_kitchen_params* p, const double* qpos, const double* goal, const double* noise, double* obs,
                     earl_stream_t s) {
  if (n < 0 || !p || !qpos || !goal || !obs) return EARL_ERR_ARG;
  if (n == 0) return EARL_OK;
  kitchen_obs_kernel<<<blocks((long long)n * 46), kB, 0, (hipStream_t)s>>>(n, *p, qpos, goal, noise, obs);
  return done("kitchen_obs");
}
int earl_philox_uniform(int32_t n, int32_t k, uint64_t seed, uint64_t counter, int32_t env_offset, uint32_t stream_id, double lo, double hi,
                        double* out, earl_stream_t s) {
  if (n < 0 || k < 0 || !out) return EARL_ERR_ARG;
  if (n == 0 || k == 0) return EARL_OK;
  uniform_kernel<<<blocks((long long)n * ((k + 1) / 2)), kB, 0, (hipStream_t)s>>>(n, k, seed, counter, env_offset, stream_id, lo, hi, out);
  return done("philox_uniform");
}
int earl_kitchen_reward(int32_t n, const double* obs, const double* mocap_pos, const double* site_xpos, double* reward,
                        uint8_t* success, earl_stream_t s) {
  if (n < 0 || !obs || !mocap_pos || !site_xpos) return EARL_ERR_ARG;
  if (n == 0) return EARL_OK;
  kitchen_reward_kernel<<<blocks(n), kB, 0, (hipStream_t)s>>>(n, obs, mocap_pos, site_xpos, reward, success);
  return done("kitchen_reward");
}
}
