// physics_l64.hip -- fourth build of the articulated-body stepper: the ONE-WAVEFRONT-PER-ENV (64 lanes) instantiations of the nv = 10 / 15 kernels
// (physics_kernel<10 | 15, 64, *>, sawyer_rollout_kernel<10 | 15, 64>): the measurement / test switch earl_debug_set_physics_lanes(64), which the parity tests use to show
// that results do not depend on the lane layout.  A translation unit of its own (round 5) so that the main unit does not carry them: a third of its compile time.
#define EARL_PHYS_UNIT_L64 1
#include "physics.hip"
