// physics_kitchen.hip -- fifth build of the articulated-body stepper: the KITCHEN (BASELINE configs[3]).
//
// physics.hip instantiated for nv = 23 (32 lanes per env, arm block + fixtures solver, joint couplings, dry friction, springs, force-limited actuators) plus the kitchen env
// kernels (physics_env_kitchen.h: the per-step kernels, the fused rollout) and the entry points earl_kitchen_step / earl_kitchen_rollout; earl_physics_step / _forward of the
// main unit forward nv = 23 here.  A translation unit of its own (round 5): the main unit's compile time was 50 s with it.
#define EARL_PHYS_UNIT_KITCHEN 1
#include "physics.hip"
