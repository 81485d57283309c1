// physics_w8.hip -- second build of the Sawyer-door rollout kernel: EIGHT wavefronts per CU.
//
// The door model's env block was cut to 4.4 KB (packed mass matrix and Hessian, the weld Jacobian aliased with the Hessian, 8-word contact
// records), so an eight-wave workgroup (32 envs + ONE copy of the model / block tables = 155 KB) fits a CU's 160 KB of LDS: two waves per SIMD
// hide each other's LDS round trips and dependent-issue stalls (one wave per SIMD issues on 55 % of its cycles).  The
// price is the 256-register cap of two waves per SIMD (204 B of scratch per lane, none of it touched inside the timestep loop).  N = 8192 is then ONE round of
// 256 workgroups instead of two rounds of 1024 single-wave ones.  Same source, same arithmetic, bit-identical outputs
// (tests/test_sawyer_full_gpu.py); earl_sawyer_rollout picks it for batches of more than 4096 envs.
#define EARL_DOOR_WPB 8
#define EARL_DOOR_PACKED 1
#define EARL_DOOR_COOP 0        // register-resident factorisation, as in the single-wave build.  (The lane-cooperative in-LDS form was the faster one
                                // while this build still reloaded spilled values inside the timestep -- 73.1 against 77.4 ms; with those gone it is the
                                // slower one: 60.3 against 57.5 ms.)
#define EARL_NO_PREFETCH 1      // no prefetch of the first near block's pair record: the second wave hides that latency, the 22 registers are worth more
#define EARL_PHYS_VARIANT_W8 1
#include "physics.hip"
