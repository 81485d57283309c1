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

__device__ __forceinline__ double mt_interp7(double x) {   // np.interp on motor.py's current / torque table
#pragma clang fp contract(off)
  const double xp[7] = {0, 10, 20, 30, 40, 50, 60}, fp[7] = {0, 1, 1.9, 2.45, 3.0, 3.25, 3.5};
  if (x >= 60.0) return 3.5;
  double x0 = 0, f0 = 0, x1 = 10, f1 = 1;
#pragma unroll
  for (int j = 1; j < 6; ++j)
    if (x >= xp[j]) { x0 = xp[j]; f0 = fp[j]; x1 = xp[j + 1]; f1 = fp[j + 1]; }
  const double slope = (f1 - f0) / (x1 - x0);
  return slope * (x - x0) + f0;
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
