/* earl_physics.h -- C ABI of the batched articulated-body stepper (BASELINE config 3:
 * sawyer_door and sawyer_peg).
 *
 * STATUS: smooth dynamics, weld / joint-limit constraints, frictional contacts (spheres / points vs boxes, plate edges vs
 * capsules; friction cone per model: elliptic for the Sawyer scenes, as their MJCF asks, pyramidal for the kitchen and the minitaur -- earl_collision_model.cone,
 * DESIGN.md 16.10).  The weld's regularisers carry two factors identified on the contact-free prefixes of the reference's MuJoCo recordings (translation x 3.35,
 * rotation x 0.07: DESIGN.md 16.9; rounds 1 - 3: 4.0 and the "mocap quaternion as given" rule, section 9).  Parity with MuJoCo is UNPINNED (the
 * simulator is not available to this build): the kernel is tested against this build's own CPU reference
 * (oracle/physics_oracle.py: LinkModel) which follows MuJoCo's documented pipeline and is checked by first principles;
 * the model tables and forward kinematics ARE pinned by numbers recorded in the reference (tests/test_physics.py).
 *
 * One wavefront per env instance, per-link state staged in LDS (north_star).  Conventions as in earl_tabletop.h:
 * device pointers, caller's stream, negative error codes, no allocation, no synchronisation.
 *
 * The model is the reduced "link" form produced by tools/mjcf_compile.py (earl_benchmark_amd/models/<env>_links.npz):
 * one link per dof (the jointed body merged with its fixed descendants), parents before children.
 */
#ifndef EARL_PHYSICS_H
#define EARL_PHYSICS_H
#include <stdint.h>

#include "earl_tabletop.h"

#ifdef __cplusplus
extern "C" {
#endif

#define EARL_MAXV 16   /* links (= dofs) */
#define EARL_MAXATT 8  /* named frames attached to links (bodies / sites / geoms the env observes) */
#define EARL_MAXACT 4

typedef struct earl_link_model {
  int32_t nv, n_att, n_act, weld_att;      /* weld_att: attachment welded to the mocap body */
  int32_t n_jump;                          /* rounds of ancestor doubling the kinematics needs: ceil(log2(max depth)) */
  int32_t ball_dof;                        /* first of the three rotation dofs of the free body (always the last three dofs), -1 = none */
  int32_t nq, pad_;                        /* length of a qpos row: nv, or nv + 1 with a free body (its orientation is a unit quaternion
                                              stored at qpos[ball_dof .. ball_dof + 3], MuJoCo's layout) */
  int32_t jump[4][EARL_MAXV];              /* jump[r][l]: ancestor of link l at distance 2^r, -1 if none */
  int32_t parent[EARL_MAXV];               /* -1 = world */
  int32_t jtype[EARL_MAXV];                /* 0 hinge, 1 slide; a free joint is six links: three slides along the world axes, then
                                              2 = applies the orientation quaternion, axis = body x; 3, 3 = rigid, axes = body y, z
                                              (MuJoCo: angular velocity of a free body in body axes) */
  int32_t limited[EARL_MAXV];
  uint32_t anc_mask[EARL_MAXV];            /* bit i: link i is an ancestor of (or is) this link */
  uint32_t desc_mask[EARL_MAXV];           /* bit i: link i is in the subtree of this link (incl. itself) */
  int32_t att_link[EARL_MAXATT];           /* -1 = fixed to the world */
  int32_t act_joint[EARL_MAXACT];
  double tpos[EARL_MAXV][3], tquat[EARL_MAXV][4];   /* link frame in its parent link's frame (joint at zero) */
  double jaxis[EARL_MAXV][3], jpos[EARL_MAXV][3];
  double mass[EARL_MAXV], com[EARL_MAXV][3], inertia[EARL_MAXV][6];   /* xx yy zz xy xz yz about the COM, link axes */
  double range[EARL_MAXV][2], damping[EARL_MAXV], armature[EARL_MAXV];
  double jsolref[EARL_MAXV][2], jsolimp[EARL_MAXV][5], dof_invweight[EARL_MAXV];
  double att_pos[EARL_MAXATT][3], att_quat[EARL_MAXATT][4];
  double act_kp[EARL_MAXACT], act_ctrlrange[EARL_MAXACT][2];
  double weld_solref[2], weld_solimp[5], weld_invweight[2];
  double gravity[3], dt;
  double drag_G[EARL_MAXV], drag_b[EARL_MAXV];   /* soft velocity row per dof, cost 1/2 G (a + b v)^2: a permanent deep contact reduced at
                                                    model-compile time (the door panel standing in the table top), 0 = none */
  uint32_t cd_mask[EARL_MAXV];             /* links whose velocity enters d/dt of this link's axis: anc_mask, except that the three rotation
                                              axes of a free body all use the velocity before any of them (mj_comVel) */
} earl_link_model;

/* The same tables for models of up to 24 dofs (the kitchen: nv = 23), plus what only that model uses: dry joint
 * friction, joint springs, force-limited actuators and linear joint couplings.  A separate struct so that the 16-dof form above -- staged in LDS
 * by the Sawyer kernels, whose workgroups fill a CU's LDS to the last 200 bytes -- keeps its size.  Kernels take it when nv > EARL_MAXV. */
#define EARL_MAXV24 24
#define EARL_MAXATT24 16
#define EARL_MAXJEQ 8
#define EARL_MAXCONNECT 4
typedef struct earl_link_model24 {
  int32_t nv, n_att, n_act, weld_att;
  int32_t n_jump, ball_dof, nq, n_jeq;
  int32_t jump[5][EARL_MAXV24];
  int32_t parent[EARL_MAXV24], jtype[EARL_MAXV24], limited[EARL_MAXV24];
  uint32_t anc_mask[EARL_MAXV24], desc_mask[EARL_MAXV24];
  int32_t att_link[EARL_MAXATT24];
  int32_t act_joint[EARL_MAXACT];
  int32_t jeq_joint1[EARL_MAXJEQ], jeq_joint2[EARL_MAXJEQ];   /* coupling e: q[joint1] - c0 - c1 q[joint2] = 0 (MuJoCo <equality><joint polycoef>, linear term, qpos0 = 0) */
  double tpos[EARL_MAXV24][3], tquat[EARL_MAXV24][4];
  double jaxis[EARL_MAXV24][3], jpos[EARL_MAXV24][3];
  double mass[EARL_MAXV24], com[EARL_MAXV24][3], inertia[EARL_MAXV24][6];
  double range[EARL_MAXV24][2], damping[EARL_MAXV24], armature[EARL_MAXV24];
  double jsolref[EARL_MAXV24][2], jsolimp[EARL_MAXV24][5], dof_invweight[EARL_MAXV24];   /* every solimp of this struct (and of the collision classes used with it): power 1 or 2, or d0 == dwidth -- the
                                                                                           * nv > 16 kernels evaluate the impedance without pow() (csrc/physics_math.h imp_p2; the Python host side refuses other tables) */
  double att_pos[EARL_MAXATT24][3], att_quat[EARL_MAXATT24][4];
  double act_kp[EARL_MAXACT], act_ctrlrange[EARL_MAXACT][2];
  double weld_solref[2], weld_solimp[5], weld_invweight[2];
  double gravity[3], dt;
  double drag_G[EARL_MAXV24], drag_b[EARL_MAXV24];
  uint32_t cd_mask[EARL_MAXV24];
  double frictionloss[EARL_MAXV24];            /* dry friction: a constraint row per dof whose force is bounded by +- frictionloss (mjCNSTR_FRICTION_DOF) */
  double stiffness[EARL_MAXV24], springref[EARL_MAXV24];   /* joint spring: passive force -stiffness (q - springref) */
  double act_forcerange[EARL_MAXACT][2];       /* force-limited actuators: kp (ctrl - q) clamped to this range (+-inf: not limited) */
  double jeq_coef[EARL_MAXJEQ][2], jeq_solref[EARL_MAXJEQ][2], jeq_solimp[EARL_MAXJEQ][5], jeq_invweight[EARL_MAXJEQ];
  int32_t pair[EARL_MAXV24];                   /* the dof a coupling ties this dof to, -1 = none: dofs beyond the first tree (the arm: the first 9) are
                                                  their own trees, so without contacts the constraint Hessian is the arm's block plus 2 x 2 / 1 x 1 blocks */
  /* connect constraints (MuJoCo mjEQ_CONNECT; Bullet JOINT_POINT2POINT): the world positions of attachments con_att1[e] and con_att2[e] coincide --
   * three soft equality rows each (the minitaur's four knee closures, earl_benchmark/envs/minitaur.py:212-217).  Models with weld_att < 0 have no
   * mocap weld.  A free ROOT body (ball_dof = 3: the minitaur's base) keeps MuJoCo's qpos layout [xyz, quaternion, joints]: the qpos slot of dof
   * l > ball_dof + 2 is l + 1. */
  int32_t n_con, con_att1[EARL_MAXCONNECT], con_att2[EARL_MAXCONNECT], pad3_[3];
  double con_solref[EARL_MAXCONNECT][2], con_solimp[EARL_MAXCONNECT][5], con_invweight[EARL_MAXCONNECT];
} earl_link_model24;

/* Collision geometry of a link model: SPHERES (cylinders are chains of spheres; box corners are spheres of radius 0)
 * tested against BOXES, and EDGES (segments) tested against CAPSULES (blk_cap bit 8), over a fixed pair list; per-pair solver parameters by class (MuJoCo's geom mixing rules applied
 * at model-compile time).  At most max_con (<= EARL_MAXCON) contacts per env and timestep: the first active pairs in list order.
 * Models with nv <= 10 are limited to 8 contact slots and 16 blocks (their workgroup then fits four times into a CU's LDS). */
#define EARL_MAXSPH 96
#define EARL_MAXBOX 16
#define EARL_MAXPAIR 512
#define EARL_MAXCLS 16
#define EARL_MAXCON 12
#define EARL_MAXBLK 64     /* kernels: 16 for nv <= 10, 64 for the kitchen (nv = 23), 32 otherwise (csrc/physics.hip Lim<NV>::MB) */
typedef struct earl_collision_model {
  int32_t n_sph, n_box, n_pair, n_cls;
  /* pairs are stored box-major in blocks (one box x one set of spheres); a block is skipped when the bounding sphere of its
   * set (centre given in the frame of blk_link, -1 = world) is farther than blk_reach from the box centre */
  int32_t n_blk;
  int32_t max_con;                           /* contacts kept per env and timestep (<= EARL_MAXCON; <= 8 for models with nv <= 10): the first active pairs */
  int32_t cone;                              /* friction cone of the model's MJCF: 0 = pyramidal (four edge rows per contact), 1 = elliptic (rows normal, t1, t2 with one regulariser,
                                                MuJoCo's three-zone cost; round 4).  The kernels compile the cone per model size (csrc/physics.hip Lim<NV>::ELLIPTIC: nv <= 16, the
                                                Sawyer door and peg, metaworld_assets/scene/basic_scene.xml:2); an entry point given the other kind returns EARL_ERR_ARG
                                                (the word is copied from the device table once per device address and remembered -- not while the stream is being captured
                                                into a graph --, so a table is not to be rewritten in place with the other cone) */
  int32_t pad_;
  int32_t blk_begin[EARL_MAXBLK], blk_end[EARL_MAXBLK], blk_box[EARL_MAXBLK], blk_link[EARL_MAXBLK];
  int32_t blk_cap[EARL_MAXBLK];              /* bits 0-7: contacts a block may contribute (its first ones in pair order); the block order is the
                                                priority order of the max_con slots.  bit 8: KIND of the block, 0 = spheres / points vs a box,
                                                1 = EDGES vs a CAPSULE: the block's "box" is a capsule (axis = box z, radius = box_half[0],
                                                segment half length = box_half[2] - box_half[0]; box_half bounds it for the block's bounding
                                                test) and its "spheres" are segments (pair_rec.pos = midpoint, .dir = unit direction in the link
                                                frame, .hl = half length); test = closest points of the two segments, normal from the capsule's
                                                axis to the edge.  (A chain sphere on the flat of a plate is pushed along the plate normal; MuJoCo's
                                                box-cylinder contact pushes along the cylinder's radial direction where the plate's edge digs in.) */
  double blk_center[EARL_MAXBLK][3], blk_reach[EARL_MAXBLK];
  /* second bounding test of a block: the box of its set in the frame of blk_link (axis-aligned there; radii, edge lengths, the slack of sliding
   * members and the contact margin included) against the block's box, by the six face axes of the two boxes: a separating axis means no pair
   * of the block is within its margin, so the block is skipped.  Both tests only ever skip blocks without contacts: results do not depend on them */
  double blk_obb_center[EARL_MAXBLK][3], blk_obb_half[EARL_MAXBLK][3];
  int32_t sph_link[EARL_MAXSPH];             /* -1 = fixed to the world */
  int32_t box_link[EARL_MAXBOX];
  double sph_pos[EARL_MAXSPH][3], sph_r[EARL_MAXSPH];
  double box_pos[EARL_MAXBOX][3], box_quat[EARL_MAXBOX][4], box_half[EARL_MAXBOX][3];
  uint8_t pair_sph[EARL_MAXPAIR], pair_box[EARL_MAXPAIR], pair_cls[EARL_MAXPAIR];
  uint8_t pair_kind[EARL_MAXPAIR];           /* 0 = sphere / point vs box, 1 = edge vs capsule (= bit 8 of the block's blk_cap), 2 = CYLINDER vs box (round 5): pair_rec.pos = the cylinder's
                                                centre, .dir = its axis, .hl = half length, .r = radius, flat ends; ONE contact per pair from the box-cylinder narrow phase -- portal
                                                refinement on the two shapes inflated by half the margin each, the routine MuJoCo sends this geom pair to (its general convex
                                                collider); metaworld_assets/objects/assets/doorlockB.xml:17-20 */
  /* the same pairs, self-contained (one load per test): sphere link, class, local centre, radius, class margin */
  struct { int32_t sph_link, cls; double pos[3], r, margin, dir[3], hl; } pair_rec[EARL_MAXPAIR];
  double cls_mu[EARL_MAXCLS], cls_solref[EARL_MAXCLS][2], cls_solimp[EARL_MAXCLS][5], cls_margin[EARL_MAXCLS], cls_invw[EARL_MAXCLS];
  double cls_mu_tor[EARL_MAXCLS];            /* round 5: torsional friction coefficient [length] of a class whose contacts have condim 4 (MuJoCo: the larger condim and the elementwise larger
                                                friction of the two geoms; the Sawyer claws and pads: xyz_base.xml:163-185, friction 2 0.1 0.002); 0 = condim 3, no torsional row.
                                                Elliptic-cone models only: a fourth row per contact, the relative angular velocity about the normal scaled by mu_tor / mu */
} earl_collision_model;

/* nsub timesteps of every env.  model: DEVICE copy of an earl_link_model (nv <= 16) or of an earl_link_model24 (nv = 23); col: DEVICE copy of its earl_collision_model or
 * NULL (no contacts).  State (updated in place):
 * qpos [n, nq], qvel [n, nv]; inputs mocap_pos [n,3], mocap_quat [n,4] (used AS GIVEN in the weld's orientation rows: an unnormalised quaternion scales their residual and
 * Jacobian by its norm -- metaworld's [1, 0, 1, 0] is meant to be passed unchanged), ctrl [n, n_act];
 * att_xpos (may be NULL) [n, n_att, 3]: world positions of the attachments after the last timestep.
 * Solver start: the active-set iteration of the call's FIRST timestep starts from "every instantiated row active"; every later timestep of the same
 * call starts from the set its rows take at the previous timestep's solution (dry-friction rows always from their quadratic zone) -- MuJoCo warm-starts
 * from the previous qacc likewise.  The fixed point does not depend on the start, so one call of nsub timesteps and nsub calls of one agree whenever
 * the iteration converges within its 8 passes (always, in the door and peg soaks; 33 of 1 M kitchen timesteps did not).  The env entry points
 * (earl_sawyer_rollout, earl_kitchen_step) start every ENV step cold, so T calls of one step and one fused rollout of T run the same iteration. */
int earl_physics_step(const void* model, const earl_collision_model* col, int32_t nv, int32_t n, int32_t nsub, double* qpos, double* qvel,
                      const double* mocap_pos, const double* mocap_quat, const double* ctrl, double* att_xpos,
                      earl_stream_t stream);

/* Forward quantities of the CURRENT state without integrating (tests): qacc [n,nv], efc_force [n, 6+2nv] (may be NULL) */
int earl_physics_forward(const void* model, const earl_collision_model* col, int32_t nv, int32_t n, const double* qpos, const double* qvel,
                         const double* mocap_pos, const double* mocap_quat, const double* ctrl, double* qacc,
                         double* efc_force, double* att_xpos, earl_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * Sawyer envs on the stepper.  Replaces, per env instance:
 *   SawyerXYZEnv.step [UPSTREAM metaworld, not in the reference tree; behaviour per SURVEY.md Appendix D]:
 *     mocap += clip(a[:3], -1, 1) * action_scale (float32 product), clipped to [mocap_low, mocap_high], mocap quat fixed;
 *     ctrl = [a[3], -a[3]]; frame_skip timesteps;
 *   SawyerDoorV2._get_obs          earl_benchmark/envs/sawyer_door.py:86-94   obs[14] = hand xyz, gripper opening, object xyz, goal[7]
 *   SawyerDoorV2.compute_reward    earl_benchmark/envs/sawyer_door.py:141-171 (sparse: is_successful :173-177, radius 0.02)
 *   PersistentStateWrapper.step    earl_benchmark/wrappers/persistent_state_wrapper.py:17-31  (done = steps_since_reset >= horizon)
 *   SawyerDoorV2.reset_model       earl_benchmark/envs/sawyer_door.py:111-125 (settled hand pose, object angle init + U(lo, hi))
 *   SawyerPegV2._get_obs           earl_benchmark/envs/sawyer_peg.py:134-142, _get_pos_objects :186-187 (object xyz = site pegHead)
 *   SawyerPegV2.compute_reward     earl_benchmark/envs/sawyer_peg.py:231-299 -- sparse: is_successful :301-305, radius 0.05; dense:
 *                                  metaworld's reward_utils / _gripper_caging_reward restated [UPSTREAM, unpinned]
 *   SawyerPegV2.reset_model        earl_benchmark/envs/sawyer_peg.py:192-229, get_next_goal / reset_goal :144-163
 * obs is float64 like the reference's (the demonstrations store it as float32). */
typedef struct earl_sawyer_cfg {
  int32_t n, env_offset;
  int32_t reward_type;                     /* 0 sparse, 1 dense (dense uses metaworld's tolerance(): unpinned) */
  int32_t horizon;                         /* <= 0: never done */
  int32_t frame_skip;
  int32_t att_hand, att_right, att_left, att_obj;   /* attachment indices the observation reads */
  int32_t obj_dof;                         /* dof re-initialised by reset: the door hinge, or the first of the peg's three translations */
  int32_t obj_kind;                        /* 0: hinge angle <- obj_init_angle + U(angle_noise)   (SawyerDoorV2.reset_model, sawyer_door.py:111-125)
                                              1: free body, xyz <- U(obj_low, obj_high) redrawn while its xy is within obj_reject_radius of
                                                 obj_reject_xy, orientation kept, zero velocity  (SawyerPegV2.reset_model, sawyer_peg.py:192-229)
                                              2: as 1 with probability 1/2, otherwise xyz <- wide_table[randint(n_wide)] + wide_shift +
                                                 U(-wide_noise, wide_noise)^3  (wide_init, sawyer_peg.py:200-209) */
  int32_t n_goal_rows;                     /* > 0: reset draws the goal uniformly from goal_table [n_goal_rows, 7] (SawyerPegV2.get_next_goal
                                              with reset_at_goal, sawyer_peg.py:144-152); 0: goals are left as they are */
  int32_t goal_change_frequency;           /* > 0: LifelongWrapper.step (lifelong_wrapper.py:30-44): every that many steps since the last reset /
                                              switch the goal is redrawn (from goal_table if n_goal_rows > 0, else kept) and the goal block of the
                                              observation returned by that step is the NEW goal; the reward of that step used the old one */
  int32_t n_wide;                          /* rows of wide_table (obj_kind 2) */
  int32_t att_grasp, att_lpad, att_rpad;   /* peg dense reward: site pegGrasp, bodies leftpad / rightpad (-1: sparse only) */
  int32_t pad2_;
  double action_scale;
  double mocap_low[3], mocap_high[3], mocap_quat[4];
  double success_radius;
  double hand_init_pos[3], obj_init_pos[3];
  double obj_init_angle, angle_noise[2];
  double obj_low[3], obj_high[3], obj_reject_xy[2], obj_reject_radius;
  const double* goal_table;                /* device, [n_goal_rows, 7] or NULL */
  const double* wide_table;                /* device, [n_wide, 3] or NULL */
  double wide_shift[3], wide_noise;
  double init_tcp[3];                      /* peg dense reward: midpoint of the finger sites after _reset_hand (SawyerXYZEnv.init_tcp [UPSTREAM]) */
  double box_corners[4][3];                /* ... and the two keep-out prisms in front of the hole block (sites *_corner_collision_box_{1,2}, sawyer_peg.py:252-256) */
  uint64_t seed, counter;                  /* reset draws: Philox(seed; draw, global env id, counter) */
  uint64_t step_counter;                   /* env steps taken before this launch (goal-switch draws: Philox(seed; 0xFFFE, global env id, step)) */
} earl_sawyer_cfg;

typedef struct earl_sawyer_state {
  double* qpos;                 /* [n, nq] */
  double* qvel;                 /* [n, nv] */
  double* mocap_pos;            /* [n, 3] */
  double* goal;                 /* [n, 7] */
  int32_t* steps_since_reset;   /* [n] */
  int32_t* steps_since_goal_change;   /* [n]; may be NULL when cfg.goal_change_frequency == 0 */
  double* obj_init;             /* [n, 6] obj_init_pos, peg_head_pos_init as reset_model leaves them (sawyer_peg.py:213-215); may be NULL
                                   for sparse rewards; written by earl_sawyer_reset, read by the peg's dense reward */
  double* last_obs;             /* [n, 14] may be NULL: the observation last returned for each env (SawyerXYZEnv._last_stable_obs [UPSTREAM]);
                                   written by earl_sawyer_reset and at the end of earl_sawyer_rollout, read when the FIRST step of a launch
                                   diverges (later steps copy the previous row of the launch's own output) */
  int32_t* fail_count;          /* [n] may be NULL: env steps that diverged and were rolled back (see earl_sawyer_out.status) */
  int32_t* sched;               /* may be NULL.  Scratch of the TIME-SLICED schedule of earl_sawyer_rollout: 2 * ceil(n / 4) int32, ZERO on entry (the caller clears it
                                   before every call).  When given, and the batch is larger than what the GPU holds at once (peg model: more than 16 envs per CU),
                                   the launch is a queue of (group of 4 envs, slice of 10 env steps) items taken by persistent waves, least-advanced group first,
                                   instead of one whole rollout per wave: the slow groups (envs in contact) run without a break while the fast ones share the other
                                   wave slots.  Same results (an env's arithmetic does not depend on who runs it, tests/test_sawyer_full_gpu.py); NULL = one group per wave */
} earl_sawyer_state;

typedef struct earl_sawyer_out {
  double* obs;        /* [T, n, 14] */
  float* reward;      /* [T, n] */
  uint8_t* done;      /* [T, n] */
  uint8_t* success;   /* [T, n] is_successful(obs) */
  uint8_t* status;    /* [T, n] may be NULL: 0 = ok, EARL_STEP_DIVERGED = after this env step some qpos / qvel entry was NaN or beyond
                         EARL_BAD_VALUE in magnitude (MuJoCo's mj_checkPos / mj_checkVel test with mjMAXVAL = 1e10).  Such a step is rolled
                         back: state, mocap target <- the env's last stable ones, the row carries the last stable observation, reward 0,
                         success 0 (SawyerXYZEnv.step on MujocoException [UPSTREAM]: `return self._last_stable_obs, 0.0, False, info`);
                         the step still counts for the horizon.  Other envs of the batch are unaffected. */
  double* info;       /* [T, n, EARL_SAWYER_INFO] may be NULL (no cost then): the numbers of the `info` dict the reference's step() returns through evaluate_state --
                         door earl_benchmark/envs/sawyer_door.py:127-139, peg sawyer_peg.py:165-184 -- at the EARL_INFO_* slots below.  NB the reference's
                         info['success'] is NOT is_successful(): the door uses obj_to_target <= 0.08 (is_successful: 0.02), the peg the axis-scaled distance
                         <= 0.05; and the door's 'in_place_reward' is the HAND's tolerance term (its compute_reward returns hand_in_place in that slot).
                         A rolled-back step carries zeros. */
} earl_sawyer_out;
#define EARL_SAWYER_INFO 8
#define EARL_INFO_SUCCESS 0
#define EARL_INFO_NEAR_OBJECT 1
#define EARL_INFO_GRASP_SUCCESS 2
#define EARL_INFO_GRASP_REWARD 3
#define EARL_INFO_IN_PLACE_REWARD 4
#define EARL_INFO_OBJ_TO_TARGET 5
#define EARL_INFO_UNSCALED_REWARD 6
#define EARL_BAD_VALUE 1e10
#define EARL_STEP_DIVERGED 1

/* T env steps of every env in ONE launch (state stays in LDS between steps; qpos / qvel / mocap_pos are written back after every
 * env step that ended finite -- they are the "last stable state" a diverged step rolls back to).  action: float32 [T, n, 4]. */
int earl_sawyer_rollout(const earl_link_model* model, const earl_collision_model* col, int32_t nv, const earl_sawyer_cfg* cfg, const earl_sawyer_state* st,
                        const float* action, int32_t T, const earl_sawyer_out* out, earl_stream_t stream);

/* reset the envs with mask[i] != 0 (mask NULL = all): state <- the settled post-_reset_hand state (reset_qpos [nq], reset_qvel
 * [nv], device), object re-initialised as cfg.obj_kind says, mocap <- hand_init_pos, counters cleared; obs [n,14]
 * (may be NULL: the state, st.obj_init and st.last_obs are still written) is written for the reset envs only. */
int earl_sawyer_reset(const earl_link_model* model, int32_t nv, const earl_sawyer_cfg* cfg, const earl_sawyer_state* st,
                      const double* reset_qpos, const double* reset_qvel, const uint8_t* mask, double* obs,
                      earl_stream_t stream);

/* obs [n,14] of the CURRENT state (kinematics recomputed, nothing integrated): SawyerDoorV2._get_obs sawyer_door.py:86-94 */
int earl_sawyer_observe(const earl_link_model* model, int32_t nv, const earl_sawyer_cfg* cfg, const earl_sawyer_state* st,
                        double* obs, earl_stream_t stream);

/* compute_reward / is_successful on caller-supplied observations [n,14] (sawyer_door.py:141-177); reward / success may be NULL.
 * With reward_type sparse this is also SawyerPegV2's rule (sawyer_peg.py:295-305: radius 0.05 in cfg.success_radius). */
int earl_sawyer_door_reward(const earl_sawyer_cfg* cfg, int32_t n, const double* obs, float* reward, uint8_t* success,
                            earl_stream_t stream);
/* the info dict of SawyerDoorV2.step (evaluate_state, earl_benchmark/envs/sawyer_door.py:127-139) for n observation rows [n, 14] -> info [n, EARL_SAWYER_INFO]
 * (slots EARL_INFO_*); every entry is a function of the observation and the reward type.  status (may be NULL): rows of rolled-back steps get zeros.
 * With cfg->goal_change_frequency > 0 -- and only then -- info[.][7] is also an INPUT: 1.0 marks a row whose slots 0-2 hold the target to evaluate against instead of
 * the row's own goal block.  earl_sawyer_rollout (nv = 10, goal switching on, out->info given) writes that slot for EVERY row it emits: 1.0 and the old target on
 * goal-switch rows -- the reward of such a row was computed with the goal in force BEFORE the switch (wrappers/lifelong_wrapper.py:30-44: step(), i.e.
 * evaluate_state, then reset_goal), while its observation already carries the new goal --, 0.0 on all others; the caller prepares nothing.  Without goal switching
 * the slot is output only (whatever the buffer held is ignored).
 * (The peg's dict needs simulator state -- the pegGrasp site, the pads: earl_sawyer_out.info of earl_sawyer_rollout carries it.) */
int earl_sawyer_door_info(const earl_sawyer_cfg* cfg, int32_t n, const double* obs, const uint8_t* status, double* info, earl_stream_t stream);

/* measurement switch: lanes of a wavefront that work on one env instance -- 16 (default: four envs per wavefront) or 64 (one
 * wavefront per env).  Results are identical; DESIGN.md quotes both timings. */
/* The entry points that take a collision table check its friction cone against the kernels' (earl_collision_model.cone) by reading the cone word of the DEVICE table
 * once per address and remembering it.  The owner of a table must announce that its block is freed or rewritten -- a caching allocator hands the same address to the
 * next table, which may be of the other cone: col = the table's device address, or NULL for every table.  Returns the number of entries dropped.  (The Python front end:
 * earl_benchmark_amd.physics.DeviceModel.__del__; nothing in the reference corresponds -- MuJoCo reads `opt.cone` from the one model it holds.) */
int earl_physics_forget_table(const void* col);
int earl_debug_set_physics_lanes(int lanes_per_env);

/* ---------------------------------------------------------------------------------------------------------------------
 * Kitchen env step (BASELINE configs[3]).  Replaces, per env instance, PersistentStateWrapper.step
 * (wrappers/persistent_state_wrapper.py:17-31) o Kitchen.step (envs/kitchen.py:185-187) o KitchenV0.step
 * (envs/kitchen_assets/adept_envs/adept_envs/franka/kitchen_multitask_v0.py:91-125): action clip / scale and mocap update (:92-102), Robot.step
 * (franka/robot/franka_robot.py:178-207: velocity-limited position targets, do_simulation = frame_skip x mj_step with ctrl = the first nu = 2
 * targets), _get_obs with sensor noise (:127-139, franka_robot.py:137-168), Kitchen._get_reward_n_score / is_successful (kitchen.py:141-183).
 * The launches go to `stream` in order: promote + save, earl_kitchen_action, the nv = 23 stepper, failure guard, noise, earl_kitchen_obs,
 * earl_kitchen_reward, bookkeeping.  Everything is caller-owned device memory, scratch included. */
struct earl_kitchen_params;      /* include/earl_glue.h */
typedef struct earl_kitchen_cfg {
  int32_t n, env_offset;
  int32_t horizon;               /* <= 0: never done */
  int32_t frame_skip;            /* 40 (kitchen_multitask_v0.py:40) */
  int32_t sensor_noise;          /* 0: observations without noise (the reference's `initializing` mode) */
  int32_t n_att;                 /* attachments of the model (rows of st.att_xpos per env) */
  int32_t site_att[8];           /* attachment indices of knob1..4_site, light_site, slide_site, hinge_site2, microhandle_site (kitchen.py:148-155) */
  uint64_t seed, counter;        /* noise draws: Philox(seed; 0x4B00 + j / 2, global env id, counter) */
  const double* mocap_quat_dev;  /* device, [4]: the mocap body's (constant) orientation, used as given */
} earl_kitchen_cfg;
typedef struct earl_kitchen_state {
  double* qpos; double* qvel;    /* [n, 23] */
  double* mocap_pos;             /* [n, 3] */
  const double* goal;            /* [n, 23] */
  double* last_qp_robot;         /* [n, 9] robot joints of the newest (noisy) observation: Robot_VelAct.ctrl_velocity_limits starts from them */
  double* att_xpos;              /* [n, n_att, 3] attachment positions the stepper leaves (kinematics of the last timestep's start) */
  int32_t* steps_since_reset;    /* [n] */
  int32_t* fail_count;           /* [n] may be NULL */
  double* last_obs;              /* [n, 46] */
  /* scratch, caller-owned like everything else: */
  double* action64;              /* [n, 9] */
  double* ctrl9;                 /* [n, 9] */
  double* noise;                 /* [n, 46] (may be NULL when sensor_noise == 0) */
  double* qpos_bak; double* qvel_bak;   /* [n, 23] the state a diverged step is rolled back to */
  double* sites;                 /* [n, 8, 3] */
  uint8_t* bad;                  /* [n] */
  double* mocap_bak;             /* [n, 3] the mocap target before this step's action: a diverged step restores it (like earl_sawyer_out.status) */
  double* att_bak;               /* [n, n_att, 3] attachment positions before this step: a diverged step leaves att_xpos at them */
} earl_kitchen_state;
typedef struct earl_kitchen_out {
  double* obs;                   /* [n, 46] */
  double* reward;                /* [n] float64, like the reference's */
  uint8_t* done; uint8_t* success;
  uint8_t* status;               /* [n] may be NULL: EARL_STEP_DIVERGED as in earl_sawyer_out: state, mocap target and attachment positions are
                                    rolled back to the env's last stable ones, the row carries the last stable observation, reward 0 */
} earl_kitchen_out;
int earl_kitchen_step(const void* model24, const earl_collision_model* col, const struct earl_kitchen_params* params, const earl_kitchen_cfg* cfg,
                      const earl_kitchen_state* st, const float* action /* [n, 9] */, const earl_kitchen_out* out, earl_stream_t stream);

/* T env steps of every env in ONE launch: action [T, n, 9] float32, the rows of `out` are [T, n, ...].  Equal, bit for bit, to T calls of
 * earl_kitchen_step with cfg.counter, cfg.counter + 1, ... (state, scratch-free: action64 / ctrl9 / noise / qpos_bak / qvel_bak / sites / bad of
 * `st` are not used and may be NULL); every wave walks its envs through the whole rollout on its own, so the launch costs the slowest wave's SUM
 * over the T steps instead of T times the slowest wave of a step.  The lifelong wrapper's goal switch is not part of it (callers step those). */
int earl_kitchen_rollout(const void* model24, const earl_collision_model* col, const struct earl_kitchen_params* params, const earl_kitchen_cfg* cfg,
                         const earl_kitchen_state* st, const float* action /* [T, n, 9] */, int32_t T, const earl_kitchen_out* out, earl_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------------
 * Minitaur env (BASELINE configs[4]) on the same stepper: floating base + 16 hinges (nv = 22, nq = 23), four connect
 * constraints (the knee closures), spheres against the ground and the wall tiles; the rigid-body MODEL is this build's own authoring
 * (tools/minitaur_model.py: the reference's URDF is not in its tree) -- parity with the reference's PyBullet simulation is UNPINNED AND MODEL-LESS.
 * Replaces, per env instance, PersistentStateWrapper.step (wrappers/persistent_state_wrapper.py:17-31) o GoalConditionedMinitaurBulletEnv.step
 * (envs/minitaur_gym_env.py:505-546 over MinitaurBulletEnv.step :285-329): Minitaur.ConvertFromLegModel (envs/minitaur.py:434-457), then
 * num_substeps x { Minitaur.ApplyAction (:326-390: velocity-limit clip of the command, MotorModel.convert_to_torque envs/motor.py:49-94, overheat
 * protection, torque x motor direction); one timestep of the stepper }, _reward, is_successful, GetObservation (:300-324) + goal; and
 * reset (:222-270, :476-479): goal drawn from goal_table, [UPSTREAM MinitaurEnvRandomizer, ranges from memory] per cfg.randomize: battery voltage, motor viscous
 * damping, base / leg-link / motor masses, foot friction (the setters are the reference's own, envs/minitaur.py:468-508), pose <- reset_qpos, settle_steps x (ApplyAction(pi / 2), timestep).
 * obs [32] float64: motor angles 8, motor velocities 8, observed motor torques 8 (all in motor space = joint x direction), base orientation (x, y, z, w),
 * base x y, goal x y. */
typedef struct earl_minitaur_cfg {
  int32_t n, env_offset;
  int32_t horizon;                 /* <= 0: never done */
  int32_t num_substeps;            /* 5  (minitaur_gym_env.py:25, 161-164) */
  int32_t settle_steps;            /* 100 (:265-269) */
  int32_t randomize;               /* MinitaurEnvRandomizer [UPSTREAM pybullet_envs.bullet.minitaur_env_randomizer] per reset, bit mask: 1 battery voltage U(14.8, 16.8) and
                                      motor viscous damping U(0, 0.01) (else 16 V, 0); 2 base / leg-link / motor masses (minitaur.py:468-488); 4 foot friction (:490-498) */
  int32_t n_goals;                 /* rows of goal_table: 12 (:467-469) */
  int32_t goal_change_frequency;   /* > 0: LifelongWrapper.step (lifelong_wrapper.py:30-44), as in earl_sawyer_cfg */
  int32_t overheat_steps;          /* OVERHEAT_SHUTDOWN_TIME / time_step = 500 (minitaur.py:14-15, 356) */
  int32_t motor_dof[8];            /* dof of motor i (MOTOR_NAMES order, minitaur.py:18-22) */
  int32_t pad_;
  double motor_dir[8];             /* minitaur.py:80 */
  double motor_kp, motor_kd;       /* 1.0, 0.02 (minitaur_gym_env.py:85-86) */
  double motor_velocity_limit;     /* 150 (:472) */
  double overheat_torque;          /* 2.45 */
  double distance_weight, energy_weight;   /* 2.0, 0.005 (:473, :71) */
  double success_radius;           /* 0.1 (:500) */
  const double* goal_table;        /* device, [n_goals, 2] */
  const double* reset_qpos;        /* device, [23]: base (0, 0, 0.2), identity, motor joints dir pi / 2, knee joints dir -2.1834 (minitaur.py:10-11, 187-211) */
  double base_mass_err[2];         /* (-0.2, 0.2): SetBaseMass(U(m (1 + lo), m (1 + hi))) -> the root body's mass and inertia x that factor */
  double leg_mass_err[2];          /* (-0.2, 0.2): SetLegMasses([U of the leg-link mass, U of the motor mass]): minitaur.py:472-488 writes the FIRST to all 16 leg links --
                                      upper AND lower -- and the second to the 8 motors.  This build's links: upper = motor + upper leg link -> mass (and inertia) x
                                      (motor' + leg') / model mass; lower -> x leg' / model mass (the reference's quirk kept: a lower leg then weighs what an upper one does) */
  double leg_mass, motor_mass;     /* the two masses the randomizer reads back from the URDF (minitaur.py:101-107): leg link LEG_LINK_ID[0], motor MOTOR_LINK_ID[0] */
  double foot_friction[2];         /* (0.8, 1.5): SetFootFriction -> friction of every contact of a lower-leg link's spheres (FOOT_LINK_ID = the lower links) */
  uint64_t seed, counter;          /* reset draws: Philox(seed; 0x4D00 + k, global env id, counter), k = 0 goal, 1 voltage, 2 damping, 3 base mass, 4 leg-link mass,
                                      5 motor mass, 6 foot friction */
  uint64_t step_counter;           /* env steps taken before this launch (goal-switch draws: Philox(seed; 0xFFFE, global env id, step)) */
} earl_minitaur_cfg;
typedef struct earl_minitaur_state {
  double* qpos;                    /* [n, 23] */
  double* qvel;                    /* [n, 22] */
  double* goal;                    /* [n, 2] */
  double* motor_param;             /* [n, 6] what the randomizer set at the last reset: battery voltage, motor viscous damping, mass factor of the base, of the upper
                                      links, of the lower links, foot friction (<= 0: the contact classes' own) */
  double* observed_torque;         /* [n, 8] Minitaur._observed_motor_torques of the newest ApplyAction */
  int32_t* overheat;               /* [n, 8] Minitaur._overheat_counter */
  uint8_t* motor_enabled;          /* [n, 8] Minitaur._motor_enabled_list */
  int32_t* steps_since_reset;      /* [n] */
  int32_t* steps_since_goal_change;   /* [n]; may be NULL when cfg.goal_change_frequency == 0 */
  int32_t* fail_count;             /* [n] may be NULL */
  double* last_obs;                /* [n, 32] may be NULL (as in earl_sawyer_state) */
} earl_minitaur_state;
typedef struct earl_minitaur_out {
  double* obs;                     /* [T, n, 32] */
  double* reward;                  /* [T, n] float64, like the reference's */
  uint8_t* done; uint8_t* success;
  uint8_t* status;                 /* [T, n] may be NULL: EARL_STEP_DIVERGED as in earl_sawyer_out (state rolled back, last stable observation, reward 0) */
} earl_minitaur_out;
/* T env steps of every env in ONE launch; action float32 [T, n, 8] in [-1, 1] (the reference raises ValueError beyond +-1.01: the Python front end
 * checks; the kernel clips to +-1.01).  Solver start: as earl_physics_step states it, every env step starting cold; within an env step a contact slot that holds the
 * same collision pair as at the timestep before starts from the edge set its passes ENDED with (same fixed point, fewer passes: 2.05 -> 1.66 per timestep on random actions). */
int earl_minitaur_rollout(const void* model24, const earl_collision_model* col, const earl_minitaur_cfg* cfg, const earl_minitaur_state* st,
                          const float* action, int32_t T, const earl_minitaur_out* out, earl_stream_t stream);
/* reset the envs with mask[i] != 0 (NULL = all); obs [n, 32] (may be NULL) is written for the reset envs only */
int earl_minitaur_reset(const void* model24, const earl_collision_model* col, const earl_minitaur_cfg* cfg, const earl_minitaur_state* st,
                        const uint8_t* mask, double* obs, earl_stream_t stream);
int earl_minitaur_cfg_size(void);
/* measurement / test switch for the minitaur kernels: 1 (default) = the timestep written on the model's tree (csrc/minitaur_stepper.h: arrow-shaped
 * constraint Hessian eliminated legs -> root body, parent / child exchanges by DPP), 0 = the generic nv = 22 instantiation of the stepper (dense
 * factorisation).  Same algorithm; results agree to rounding (tests/test_minitaur_gpu.py). */
int earl_debug_set_minitaur_stepper(int tree);
/* The minitaur's fused rollout in its two-waves-per-SIMD form (csrc/physics_env_minitaur.h minitaur_duo_kernel: a timestep in two halves run by two waves; same results):
 * 1 = every packed launch, 0 = never, -1 = by batch size (the default).  Returns the previous setting.  Measurement / test switch. */
int earl_debug_set_minitaur_duo(int mode);
/* Small batches of the kitchen / minitaur launches (round 5; BASELINE configs[3] / [4] shard 2048 / 4096 envs over 8 GPUs: 256 / 512 per GPU).  An env is a serial chain of
 * T x frame_skip timesteps walked by one 32-lane group; the launch lasts as long as its slowest wave, and a wave's two envs wait for each other's longer branch in every
 * timestep.  -1 (default) = by batch size: n <= CUs: one env per WORKGROUP -- the kitchen with all FOUR waves on the env (mode 3: a timestep's constraint rows, mass matrix,
 * bias forces and collision phases side by side on the CU's four SIMDs, then one wave's active set and integration; csrc/physics_env_kitchen.h), the minitaur with one; n <= 4 x CUs:
 * one env per WAVE (the second group shadows the first one's env and stores nothing) -- the kitchen, for n <= 2 x CUs, with TWO waves per env, two envs per workgroup (mode 4: the
 * owner wave runs collision, rows, active set and integration, its helper the mass matrix, bias forces, equality Hessian and the integration's factor); otherwise two envs per wave.
 * 0 / 1 / 2 force a mode (2 = one env per workgroup, one wave), 3 / 4 (kitchen only) force the four-wave / two-wave form.  Results are bit-identical in every mode
 * (tests/test_kitchen_gpu.py, tests/test_minitaur_gpu.py).  earl_debug_set_solo: kitchen launches; earl_debug_set_solo_mt: minitaur launches.  Returns the previous setting. */
int earl_debug_set_solo(int mode);
int earl_debug_set_solo_mt(int mode);

/* measurement / test switch for the door model's rollout: 0 (default) = by batch size (n > 4096: one eight-wave workgroup per CU, see
 * csrc/physics_w8.hip; otherwise four single-wave workgroups per CU), 1 / 2 force the one or the other, 3 = single-wave workgroups under the time-sliced work
 * queue of the peg (needs earl_sawyer_state.sched; slower than 2 at N = 8192, tools/bench_door_schedule.py).  Results are bit-identical. */
int earl_debug_set_door_variant(int variant);
/* measurement / test switch for the peg model's rollout: 1 (default) = time-sliced schedule when earl_sawyer_state.sched is given and the batch exceeds one round,
 * 0 = always one group per wave, k >= 2 = time-sliced with k env steps per work item.  Results are bit-identical. */
int earl_debug_set_peg_schedule(int sliced);

/* sizeof(earl_link_model) as compiled into the library (bindings check their struct layout against it) */
int earl_physics_model_size(void);
int earl_physics_model24_size(void);
int earl_collision_model_size(void);
int earl_sawyer_cfg_size(void);

#ifdef __cplusplus
}
#endif
#endif
