/* earl_physics.h -- C ABI of the batched articulated-body stepper (BASELINE config 3).
 *
 * STATUS: smooth dynamics + weld / joint-limit constraints; NO contacts yet.  Parity with MuJoCo is UNPINNED (the
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
  int32_t parent[EARL_MAXV];               /* -1 = world */
  int32_t jtype[EARL_MAXV];                /* 0 hinge, 1 slide */
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
} earl_link_model;

/* nsub timesteps of every env.  model: DEVICE copy of an earl_link_model.  State (updated in place):
 * qpos, qvel [n, nv]; inputs mocap_pos [n,3], mocap_quat [n,4] (normalised internally), ctrl [n, n_act];
 * att_xpos (may be NULL) [n, n_att, 3]: world positions of the attachments after the last timestep. */
int earl_physics_step(const earl_link_model* model, int32_t nv, int32_t n, int32_t nsub, double* qpos, double* qvel,
                      const double* mocap_pos, const double* mocap_quat, const double* ctrl, double* att_xpos,
                      earl_stream_t stream);

/* Forward quantities of the CURRENT state without integrating (tests): qacc [n,nv], efc_force [n, 6+2nv] (may be NULL) */
int earl_physics_forward(const earl_link_model* model, int32_t nv, int32_t n, const double* qpos, const double* qvel,
                         const double* mocap_pos, const double* mocap_quat, const double* ctrl, double* qacc,
                         double* efc_force, double* att_xpos, earl_stream_t stream);

/* sizeof(earl_link_model) as compiled into the library (bindings check their struct layout against it) */
int earl_physics_model_size(void);

#ifdef __cplusplus
}
#endif
#endif
