// minitaur_device.h -- the reference's numpy around Bullet for the minitaur, as device functions shared by csrc/glue.hip (the batched glue entry
// points of include/earl_glue.h, pinned bit-exact against goldens recorded from the reference's own functions) and the env kernel in csrc/physics.hip.
//   Minitaur.ConvertFromLegModel   earl_benchmark/envs/minitaur.py:434-457
//   MotorModel.convert_to_torque   earl_benchmark/envs/motor.py:49-94
// fp64 like the reference; callers compile these with FMA contraction off (the expressions are separately rounded).
#pragma once
#include <hip/hip_runtime.h>

namespace earl {

__device__ __forceinline__ double mt_clipd(double x, double lo, double hi) { return x < lo ? lo : (x > hi ? hi : x); }

// motor angle i (0..7) of the leg-model action a[8]  (minitaur.py:434-457)
__device__ __forceinline__ double mt_leg_to_motor(const double* a, const int i) {
#pragma clang fp contract(off)
  const int idx = i >> 1;
  const double pi = 3.141592653589793, quater_pi = pi / 4;
  const double fb = (-1 * quater_pi) * (a[idx + 4] + 1.5);
  double ext = ((i & 1) ? -1.0 : 1.0) * quater_pi * a[idx];
  if (i >= 4) ext = -ext;
  return (pi + fb) + ext;
}

// motor.py's current / torque table, in constant memory: read with compile-time indices these are scalar loads into SGPRs.  As literals in the code they were
// 64-bit constants in VGPR pairs, hoisted out of the minitaur kernel's timestep loop and -- that kernel has no register to spare -- spilled: the one scratch reload
// inside its timestep loop (tests/test_no_scratch_in_timestep_loops.py)
static __device__ __constant__ const double MT_CURRENT_TABLE[7] = {0, 10, 20, 30, 40, 50, 60};
static __device__ __constant__ const double MT_TORQUE_TABLE[7] = {0, 1, 1.9, 2.45, 3.0, 3.25, 3.5};

__device__ __forceinline__ double mt_interp7(double x) {   // np.interp on motor.py's current / torque table
#pragma clang fp contract(off)
  const double* const xp = MT_CURRENT_TABLE;
  const double* const fp = MT_TORQUE_TABLE;
  double x0 = xp[0], f0 = fp[0], x1 = xp[1], f1 = fp[1];
#pragma unroll
  for (int j = 1; j < 6; ++j) {
    const bool ge = x >= xp[j];
    x0 = ge ? xp[j] : x0; f0 = ge ? fp[j] : f0; x1 = ge ? xp[j + 1] : x1; f1 = ge ? fp[j + 1] : f1;
  }
  const double slope = (f1 - f0) / (x1 - x0);
  const double y = slope * (x - x0) + f0;
  return x >= xp[6] ? fp[6] : y;
}

// one motor: command (desired angle, or the pwm itself in torque control), angle, velocity -> actual torque, observed torque  (motor.py:49-94)
__device__ __forceinline__ void mt_motor_torque(const double kp, const double kd, const double voltage, const double viscous_damping, const bool torque_control,
                                                const double command, const double angle, const double velocity, double& actual, double& observed) {
#pragma clang fp contract(off)
  const double R = 0.186, Kt = 0.0954;
  double pwm = torque_control ? command : (-kp * (angle - command) - kd * velocity);
  pwm = mt_clipd(pwm, -1.0, 1.0);
  observed = mt_clipd(Kt * (pwm * voltage / R), -5.7, 5.7);
  const double vnet = mt_clipd(pwm * voltage - (Kt + viscous_damping) * velocity, -50.0, 50.0);
  const double current = vnet / R;
  const double sign = current > 0 ? 1.0 : (current < 0 ? -1.0 : (current == 0 ? 0.0 : current));
  actual = sign * mt_interp7(fabs(current));
}

}  // namespace earl
