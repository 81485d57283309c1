// physics_mt.hip -- third build of the articulated-body stepper: the MINITAUR (BASELINE configs[4]).
//
// physics.hip instantiated for nv = 22 (Lim<22>: a floating root body + 16 hinges in ONE tree, 32 lanes per env, dense lane-cooperative in-LDS
// factorisations, connect constraints for the four knee closures, the motor model's torques handed in per timestep, no mocap weld, no joint damping)
// plus the env kernel (minitaur_kernel: reset incl. its settle steps, fused T-step rollout) and the entry points earl_minitaur_rollout /
// earl_minitaur_reset.  A translation unit of its own so that the three builds compile side by side.
#define EARL_PHYS_VARIANT_MT 1
#include "minitaur_device.h"
#include "physics.hip"
