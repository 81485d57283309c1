/* physics_oracle.c -- plain C restatement of the articulated-body stepper, one env at a time (scalar loops).
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  It restates oracle/physics_oracle.py (LinkModel.forward / step / collide /
 * contact_rows / solve_primal / solve_primal_elliptic -- the friction cone is the model's: earl_collision_model.cone), which remains the readable statement; this file exists (a) as a third implementation to
 * cross-check the numpy one and the HIP kernels, (b) as the CPU baseline of `bench.py --workload sawyer_door` (OpenMP over
 * envs).  PARITY WITH MUJOCO IS UNPINNED, exactly as for the numpy statement (see its header and DESIGN.md section 9);
 * the env glue follows oracle/sawyer_oracle.py (reference: earl_benchmark/envs/sawyer_door.py:86-177, metaworld upstream).
 * The structs are the public ones of include/earl_physics.h (host copies).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/earl_physics.h"

#define NVMAX EARL_MAXV24
#define NROWMAX (6 + 2 * EARL_MAXV24 + EARL_MAXV24 + EARL_MAXJEQ + 3 * EARL_MAXCONNECT + 4 * EARL_MAXCON)

/* The stepper below works on the 24-dof table form (earl_link_model24: the kitchen's, with dry friction / springs / force limits / joint
 * couplings); a 16-dof model (the Sawyer envs') is widened into it first -- the extras empty. */
typedef earl_link_model24 LM;
static void widen(const earl_link_model* s, LM* d) {
  memset(d, 0, sizeof(*d));
  d->nv = s->nv; d->n_att = s->n_att; d->n_act = s->n_act; d->weld_att = s->weld_att; d->n_jump = s->n_jump; d->ball_dof = s->ball_dof; d->nq = s->nq; d->n_jeq = 0;
  for (int l = 0; l < EARL_MAXV; ++l) {
    d->parent[l] = s->parent[l]; d->jtype[l] = s->jtype[l]; d->limited[l] = s->limited[l]; d->anc_mask[l] = s->anc_mask[l]; d->desc_mask[l] = s->desc_mask[l];
    d->cd_mask[l] = s->cd_mask[l]; d->mass[l] = s->mass[l]; d->damping[l] = s->damping[l]; d->armature[l] = s->armature[l]; d->dof_invweight[l] = s->dof_invweight[l];
    d->drag_G[l] = s->drag_G[l]; d->drag_b[l] = s->drag_b[l]; d->pair[l] = -1;
    memcpy(d->tpos[l], s->tpos[l], sizeof(s->tpos[l])); memcpy(d->tquat[l], s->tquat[l], sizeof(s->tquat[l])); memcpy(d->jaxis[l], s->jaxis[l], sizeof(s->jaxis[l]));
    memcpy(d->jpos[l], s->jpos[l], sizeof(s->jpos[l])); memcpy(d->com[l], s->com[l], sizeof(s->com[l])); memcpy(d->inertia[l], s->inertia[l], sizeof(s->inertia[l]));
    memcpy(d->range[l], s->range[l], sizeof(s->range[l])); memcpy(d->jsolref[l], s->jsolref[l], sizeof(s->jsolref[l])); memcpy(d->jsolimp[l], s->jsolimp[l], sizeof(s->jsolimp[l]));
  }
  for (int k = 0; k < EARL_MAXATT; ++k) { d->att_link[k] = s->att_link[k]; memcpy(d->att_pos[k], s->att_pos[k], sizeof(s->att_pos[k])); memcpy(d->att_quat[k], s->att_quat[k], sizeof(s->att_quat[k])); }
  for (int a = 0; a < EARL_MAXACT; ++a) {
    d->act_joint[a] = s->act_joint[a]; d->act_kp[a] = s->act_kp[a]; d->act_ctrlrange[a][0] = s->act_ctrlrange[a][0]; d->act_ctrlrange[a][1] = s->act_ctrlrange[a][1];
    d->act_forcerange[a][0] = -1e300; d->act_forcerange[a][1] = 1e300;
  }
  memcpy(d->weld_solref, s->weld_solref, sizeof(s->weld_solref)); memcpy(d->weld_solimp, s->weld_solimp, sizeof(s->weld_solimp));
  memcpy(d->weld_invweight, s->weld_invweight, sizeof(s->weld_invweight)); memcpy(d->gravity, s->gravity, sizeof(s->gravity)); d->dt = s->dt;
}

typedef struct { double w, x, y, z; } Q4;
typedef struct { double x, y, z; } V3;

static Q4 qmul(Q4 a, Q4 b) {
  Q4 r = {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
  return r;
}
static void qmat(Q4 q, double R[3][3]) {
  R[0][0] = 1 - 2 * (q.y * q.y + q.z * q.z); R[0][1] = 2 * (q.x * q.y - q.w * q.z); R[0][2] = 2 * (q.x * q.z + q.w * q.y);
  R[1][0] = 2 * (q.x * q.y + q.w * q.z); R[1][1] = 1 - 2 * (q.x * q.x + q.z * q.z); R[1][2] = 2 * (q.y * q.z - q.w * q.x);
  R[2][0] = 2 * (q.x * q.z - q.w * q.y); R[2][1] = 2 * (q.y * q.z + q.w * q.x); R[2][2] = 1 - 2 * (q.x * q.x + q.y * q.y);
}
static V3 v3(double x, double y, double z) { V3 r = {x, y, z}; return r; }
static V3 ld3(const double* p) { return v3(p[0], p[1], p[2]); }
static Q4 ldq(const double* p) { Q4 r = {p[0], p[1], p[2], p[3]}; return r; }
static V3 mulv(double R[3][3], V3 v) { return v3(R[0][0] * v.x + R[0][1] * v.y + R[0][2] * v.z, R[1][0] * v.x + R[1][1] * v.y + R[1][2] * v.z, R[2][0] * v.x + R[2][1] * v.y + R[2][2] * v.z); }
static V3 mulvT(double R[3][3], V3 v) { return v3(R[0][0] * v.x + R[1][0] * v.y + R[2][0] * v.z, R[0][1] * v.x + R[1][1] * v.y + R[2][1] * v.z, R[0][2] * v.x + R[1][2] * v.y + R[2][2] * v.z); }
static V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static V3 scl(V3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
static double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static double pick(V3 v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : v.z); }

/* compact spatial inertia about the world origin: {m, m c (3), Io xx yy zz xy xz yz}; I [w; v] = [n; f] */
static void iapply(const double* I, V3 w, V3 v, V3* n, V3* f) {
  V3 h = ld3(I + 1);
  *n = add(v3(I[4] * w.x + I[7] * w.y + I[8] * w.z, I[7] * w.x + I[5] * w.y + I[9] * w.z, I[8] * w.x + I[9] * w.y + I[6] * w.z), cross(h, v));
  *f = sub(scl(v, I[0]), cross(h, w));
}

static void kbimp(const double* solref, const double* solimp, double r, double dt, double* k, double* b, double* d) {
  const double tc = fmax(solref[0], 2 * dt), dr = solref[1];
  const double d0 = solimp[0], dw = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  const double x = width > 0 ? fmin(fabs(r) / width, 1.0) : 1.0;
  double y;
  if (power == 1 || d0 == dw) y = x;
  else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
  else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
  *d = d0 + y * (dw - d0);
  *k = 1.0 / (dw * dw * tc * tc * dr * dr);
  *b = 2.0 / (dw * tc);
}

/* dense Cholesky solve of an SPD system, n <= NVMAX; A is destroyed */
static void chol_solve(int n, double A[NVMAX][NVMAX], double* x) {
  for (int j = 0; j < n; ++j) {
    double d = A[j][j];
    for (int p = 0; p < j; ++p) d -= A[j][p] * A[j][p];
    d = sqrt(d);
    A[j][j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[i][j];
      for (int p = 0; p < j; ++p) s -= A[i][p] * A[j][p];
      A[i][j] = s / d;
    }
  }
  for (int i = 0; i < n; ++i) {
    double s = x[i];
    for (int p = 0; p < i; ++p) s -= A[i][p] * x[p];
    x[i] = s / A[i][i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = x[i];
    for (int p = i + 1; p < n; ++p) s -= A[p][i] * x[p];
    x[i] = s / A[i][i];
  }
}

typedef struct {
  double Xq[NVMAX][4], Xp[NVMAX][3];      /* world frames of the links at the start of the last timestep */
  double qacc[NVMAX];
  double efc[6 + 2 * NVMAX];
  int ncon;
  int warm;                               /* qacc holds the previous timestep's solution of the SAME call (env step): the active-set iteration starts from it */
  /* minitaur (round 6): the edge sets the previous timestep of the same env step ENDED with, per contact slot, and the collision pair each slot held.  A slot that holds the
   * same pair again starts from that set instead of the one a_prev predicts (csrc/minitaur_stepper.h C3): 2.05 -> 1.66 passes per timestep on random actions, same fixed point */
  int carry, pncon, ppair[EARL_MAXCON], pact[EARL_MAXCON][4];
} StepOut;

/* Warm start of the active-set iteration (the rule the kernels implement): the first timestep of a call / of an env step starts from "every
 * instantiated row active" (friction rows in their quadratic zone); every later timestep starts from the set its NEW rows take at the previous
 * timestep's solution a_prev (row active iff J a_prev - aref < 0; dry-friction rows always start in their quadratic zone).  The fixed point -- and so the result, up to
 * the order of the sums -- is the same; the iteration count is what changes.  0 = always the cold start (round 1 / early round 2). */
static int g_warm_start = 1;
int oracle_set_warm_start(int w) { const int prev = g_warm_start; g_warm_start = w; return prev; }
/* experiment switch (tools/minitaur_passes.py): 1 = the minitaur rollout keeps the warm start ACROSS env steps of one call (the product rule is: every env step starts cold) */
/* test switch (tests/test_minitaur.py): 0 = the minitaur's contact slots never start from the set their passes ended with at the timestep before (the start rule of rounds 2 - 5) */
static int g_carry_sets = 1;
int oracle_set_carry_sets(int w) { const int prev = g_carry_sets; g_carry_sets = w; return prev; }
static int g_warm_across_steps = 0;
int oracle_set_warm_across_steps(int w) { const int prev = g_warm_across_steps; g_warm_across_steps = w; return prev; }
static long long g_newton_stats[5];       /* timesteps, Newton iterations, timesteps with contacts, iterations in those, timesteps that used all 8 iterations without reaching a fixed point */
void oracle_newton_stats(long long* out, int reset) {
  for (int i = 0; i < 5; ++i) { out[i] = g_newton_stats[i]; if (reset) g_newton_stats[i] = 0; }
}

/* 1 (the rule the kernels implement): the mocap quaternion enters the weld rows AS GIVEN (metaworld sets [1, 0, 1, 0], norm sqrt 2: residual
 * and Jacobian of the orientation rows scale by sqrt 2); 0 = normalised first (round 1, kept as an experiment switch for tools/heldout_eval.py) */
static int g_raw_mocap_quat = 1;
int oracle_set_raw_mocap_quat(int raw) { const int prev = g_raw_mocap_quat; g_raw_mocap_quat = raw; return prev; }
/* experiment hook (tools/weld_free_motion_fit.py): per-row factors on the six weld regularisers, 1 = the tables' value */
static double g_weld_row_scale[6] = {1, 1, 1, 1, 1, 1};
void oracle_set_weld_row_scale(const double* s6) { for (int r = 0; r < 6; ++r) g_weld_row_scale[r] = s6 ? s6[r] : 1.0; }
/* experiment switch (tools/weld_free_motion_fit.py --forms, tools/heldout_eval.py): 1 = the impedance of the weld's six rows is ONE number, read at the NORM of the six residuals
 * (MuJoCo's engine_core_constraint.c getposdim: a weld is one 6-dimensional constraint); 0 = every row its own (what the kernels implement) */
static int g_weld_norm_imp = 0;
int oracle_set_weld_norm_imp(int on) { const int prev = g_weld_norm_imp; g_weld_norm_imp = on; return prev; }
static Q4 qnormalize(Q4 q) {
  const double s = 1.0 / sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  Q4 r = {q.w * s, q.x * s, q.y * s, q.z * s};
  return r;
}

/* one timestep (integrate != 0) or the forward quantities only; reference: LinkModel.forward / step.
 * qp is a qpos row [nq] (the free body's quaternion at [ball_dof, ball_dof + 4)), qv a qvel row [nv]. */
/* the minitaur's randomised foot friction (Minitaur.SetFootFriction, minitaur.py:490-498): friction of every contact of a LOWER-leg link's spheres
 * (links behind the root body whose parent is not the root); <= 0: the contact classes' own.  Set per env by the minitaur front end below. */

/* ---- box vs finite cylinder: ONE contact per pair (round 5) ---------------------------------------------------------------
 * MuJoCo 2.1 has no analytic box-cylinder routine: the pair goes to its general convex collider (Minkowski portal refinement, one contact: deepest
 * penetration along the portal's direction).  Restated from the published algorithm (G. Snethen, "XenoCollide: Complex Collision Made Simple", Game
 * Programming Gems 7; the form used by libccd's ccdMPRPenetration): support mappings of the two shapes, each inflated by half the contact margin (so that
 * the shapes overlap exactly when their distance is below the margin); interior point = difference of the centres; portal discovery; refinement until the
 * support point in the portal's direction is within `tol` of the portal plane; depth / direction = distance and direction from the origin to the portal
 * triangle; position = the origin's barycentric combination of the support points, midway between the two shapes.  dist = margin - depth.
 * obj1 = box (centre pb, frame Rb, half sizes h), obj2 = cylinder (centre c, axis a, half length hl, radius r); the direction points from the box to the cylinder. */
typedef struct { V3 v, p1, p2; } MprV;
static MprV mpr_support(V3 pb, double Rb[3][3], V3 h, V3 c, V3 a, double hl, double r, double infl, V3 dir) {
  MprV s;
  const V3 dl = mulvT(Rb, dir);
  s.p1 = add(add(pb, mulv(Rb, v3(dl.x > 0 ? h.x : -h.x, dl.y > 0 ? h.y : -h.y, dl.z > 0 ? h.z : -h.z))), scl(dir, infl));
  const V3 nd = scl(dir, -1.0);
  const double da = dot(nd, a);
  const V3 rad = sub(nd, scl(a, da));
  const double nr = sqrt(dot(rad, rad));
  V3 q = add(c, scl(a, da > 0 ? hl : (da < 0 ? -hl : 0.0)));
  if (nr > 1e-15) q = add(q, scl(rad, r / nr));
  s.p2 = add(q, scl(nd, infl));
  s.v = sub(s.p1, s.p2);
  return s;
}
static V3 vnormalize(V3 v) { const double n = sqrt(dot(v, v)); return n > 0 ? scl(v, 1.0 / n) : v; }
/* closest point of triangle (a, b, c) to the origin (Ericson, Real-Time Collision Detection 5.1.5) */
static V3 tri_closest_to_origin(V3 a, V3 b, V3 c) {
  const V3 ab = sub(b, a), ac = sub(c, a), ap = scl(a, -1.0);
  const double d1 = dot(ab, ap), d2 = dot(ac, ap);
  if (d1 <= 0 && d2 <= 0) return a;
  const V3 bp = scl(b, -1.0);
  const double d3 = dot(ab, bp), d4 = dot(ac, bp);
  if (d3 >= 0 && d4 <= d3) return b;
  const double vc = d1 * d4 - d3 * d2;
  if (vc <= 0 && d1 >= 0 && d3 <= 0) return add(a, scl(ab, d1 / (d1 - d3)));
  const V3 cp = scl(c, -1.0);
  const double d5 = dot(ab, cp), d6 = dot(ac, cp);
  if (d6 >= 0 && d5 <= d6) return c;
  const double vb = d5 * d2 - d1 * d6;
  if (vb <= 0 && d2 >= 0 && d6 <= 0) return add(a, scl(ac, d2 / (d2 - d6)));
  const double va = d3 * d6 - d5 * d4;
  if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) return add(b, scl(sub(c, b), (d4 - d3) / ((d4 - d3) + (d5 - d6))));
  const double den = 1.0 / (va + vb + vc);
  return add(a, add(scl(ab, vb * den), scl(ac, vc * den)));
}
#define MPR_TOL 1e-10
#define MPR_MAXIT 64
static int mpr_box_cylinder(V3 pb, double Rb[3][3], V3 h, V3 c, V3 a, double hl, double r, double margin, double* dist, V3* n, V3* pos) {
  const double infl = 0.5 * margin;
#define SUP(d) mpr_support(pb, Rb, h, c, a, hl, r, infl, (d))
  MprV v0, v1, v2, v3_, v4;
  v0.p1 = pb; v0.p2 = c; v0.v = sub(pb, c);
  if (dot(v0.v, v0.v) < 1e-30) v0.v = v3(1e-5, 0, 0);
  V3 dir = vnormalize(scl(v0.v, -1.0));
  v1 = SUP(dir);
  if (dot(v1.v, dir) < 0) return 0;
  dir = cross(v0.v, v1.v);
  if (dot(dir, dir) < 1e-30) {                      /* the origin lies on the ray v0 -> v1: depth and direction along it */
    const double d1 = sqrt(dot(v1.v, v1.v));
    *dist = margin - d1;
    *n = d1 > 0 ? scl(v1.v, 1.0 / d1) : vnormalize(scl(v0.v, -1.0));
    *pos = scl(add(v1.p1, v1.p2), 0.5);
    return 1;
  }
  dir = vnormalize(dir);
  v2 = SUP(dir);
  if (dot(v2.v, dir) < 0) return 0;
  dir = vnormalize(cross(sub(v1.v, v0.v), sub(v2.v, v0.v)));
  if (dot(dir, v0.v) > 0) { const MprV t = v1; v1 = v2; v2 = t; dir = scl(dir, -1.0); }
  for (int it = 0;; ++it) {                          /* portal discovery */
    if (it > MPR_MAXIT) return 0;
    v3_ = SUP(dir);
    if (dot(v3_.v, dir) < 0) return 0;
    if (dot(cross(v1.v, v3_.v), v0.v) < -1e-300) { v2 = v3_; dir = vnormalize(cross(sub(v1.v, v0.v), sub(v2.v, v0.v))); continue; }
    if (dot(cross(v3_.v, v2.v), v0.v) < -1e-300) { v1 = v3_; dir = vnormalize(cross(sub(v1.v, v0.v), sub(v2.v, v0.v))); continue; }
    break;
  }
  int hit = 0;
  for (int it = 0;; ++it) {                          /* portal refinement, then the penetration once the portal has passed the origin */
    dir = vnormalize(cross(sub(v2.v, v1.v), sub(v3_.v, v1.v)));
    if (!hit && dot(dir, v1.v) >= 0) hit = 1;        /* the portal encloses the origin: the shapes (inflated) overlap */
    v4 = SUP(dir);
    const double d4 = dot(v4.v, dir);
    const double reach = fmin(fmin(d4 - dot(v1.v, dir), d4 - dot(v2.v, dir)), d4 - dot(v3_.v, dir));
    if (!hit && d4 < 0) return 0;                    /* the support plane separates the origin */
    if (reach <= MPR_TOL || it >= MPR_MAXIT) {
      if (!hit) return 0;
      const V3 w = tri_closest_to_origin(v1.v, v2.v, v3_.v);
      const double depth = sqrt(dot(w, w));
      *n = depth > 1e-14 ? scl(w, 1.0 / depth) : dir;
      *dist = margin - depth;
      /* barycentric coordinates of the origin ray in the portal (libccd findPos) */
      double b0 = dot(cross(v1.v, v2.v), v3_.v), b1 = dot(cross(v3_.v, v2.v), v0.v), b2 = dot(cross(v0.v, v1.v), v3_.v), b3 = dot(cross(v2.v, v1.v), v0.v);
      double sum = b0 + b1 + b2 + b3;
      if (sum <= 0) { b0 = 0; b1 = dot(cross(v2.v, v3_.v), dir); b2 = dot(cross(v3_.v, v1.v), dir); b3 = dot(cross(v1.v, v2.v), dir); sum = b1 + b2 + b3; }
      const double inv = 1.0 / sum;
      const V3 p1 = add(add(scl(v0.p1, b0), scl(v1.p1, b1)), add(scl(v2.p1, b2), scl(v3_.p1, b3)));
      const V3 p2 = add(add(scl(v0.p2, b0), scl(v1.p2, b1)), add(scl(v2.p2, b2), scl(v3_.p2, b3)));
      *pos = scl(add(p1, p2), 0.5 * inv);
      return 1;
    }
    /* expand the portal with v4 (libccd ccdMPR expandPortal) */
    const V3 v4v0 = cross(v4.v, v0.v);
    if (dot(v1.v, v4v0) > 0) { if (dot(v2.v, v4v0) > 0) v1 = v4; else v3_ = v4; }
    else { if (dot(v3_.v, v4v0) > 0) v2 = v4; else v1 = v4; }
  }
#undef SUP
}
/* diagnostics (single-threaded use): the contacts of the most recent timestep -- pair index, distance, normal, position (tools/door_contact_ablation.py --trace) */
static double g_dbg_contacts[EARL_MAXCON][8];
static int g_dbg_ncon = 0, g_dbg_on = 0;
void oracle_debug_trace(int on) { g_dbg_on = on; }      /* off by default: the buffer is shared, a multi-threaded batch must not write it */
int oracle_debug_contacts(double* out) { for (int c = 0; c < g_dbg_ncon; ++c) for (int k = 0; k < 8; ++k) out[c * 8 + k] = g_dbg_contacts[c][k]; return g_dbg_ncon; }
/* test hook: the narrow phase alone.  box: centre pb[3], rotation Rb[9] (row major, columns = box axes in the world), half h[3]; cylinder: centre c[3], unit axis a[3] */
int oracle_mpr_box_cylinder(const double* pb, const double* Rb9, const double* h, const double* c, const double* a, double hl, double r, double margin, double* out7) {
  double Rb[3][3];
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) Rb[i][j] = Rb9[3 * i + j];
  double dist = 0; V3 n = {0, 0, 0}, pos = {0, 0, 0};
  const int hit = mpr_box_cylinder(ld3(pb), Rb, ld3(h), ld3(c), ld3(a), hl, r, margin, &dist, &n, &pos);
  out7[0] = dist; out7[1] = n.x; out7[2] = n.y; out7[3] = n.z; out7[4] = pos.x; out7[5] = pos.y; out7[6] = pos.z;
  return hit;
}
static double g_torsion_override = -1.0;      /* experiments: >= 0 replaces the classes' torsional coefficient on cylinder contacts (oracle_set_torsion) */
void oracle_set_torsion(double mu_tor) { g_torsion_override = mu_tor; }

static _Thread_local double g_foot_mu = -1.0;
static void substep(const LM* m, const earl_collision_model* col, double* qp, double* qv, V3 mpos, Q4 mq, const double* ctrl,
                    int integrate, StepOut* o, const double* qfrc) {
  /* qfrc: generalized forces applied from outside (the minitaur's motor torques), or NULL */
  const int nv = m->nv;
  const double dt = m->dt;
  /* qpos slot of dof l (LinkModel.qadr): a free ROOT body (the minitaur's base) keeps MuJoCo's layout [xyz, quaternion, joints] */
#define QA(l) ((l) + ((m->ball_dof >= 0 && (l) > m->ball_dof + 2) ? 1 : 0))
  double S[NVMAX][6], I10[NVMAX][10], Ic[NVMAX][10], M[NVMAX][NVMAX], tau[NVMAX];
  /* kinematics (parents precede children) */
  for (int l = 0; l < nv; ++l) {
    const int p = m->parent[l];
    Q4 pq = {1, 0, 0, 0};
    V3 pp = {0, 0, 0};
    if (p >= 0) { pq = ldq(o->Xq[p]); pp = ld3(o->Xp[p]); }
    double Rp[3][3], R[3][3];
    qmat(pq, Rp);
    V3 x = add(pp, mulv(Rp, ld3(m->tpos[l])));
    Q4 q = qmul(pq, ldq(m->tquat[l]));
    qmat(q, R);
    const V3 ax = ld3(m->jaxis[l]);
    const V3 anchor = add(x, mulv(R, ld3(m->jpos[l]))), axw = mulv(R, ax);
    if (m->jtype[l] == 0) {
      const double h = 0.5 * qp[QA(l)], sn = sin(h), cs = cos(h);
      Q4 qj = {cs, sn * ax.x, sn * ax.y, sn * ax.z};
      q = qmul(q, qj);
      qmat(q, R);
      x = sub(anchor, mulv(R, ld3(m->jpos[l])));
      const V3 sv = cross(anchor, axw);
      S[l][0] = axw.x; S[l][1] = axw.y; S[l][2] = axw.z; S[l][3] = sv.x; S[l][4] = sv.y; S[l][5] = sv.z;
    } else if (m->jtype[l] >= 2) {
      /* rotation of a free body (LinkModel.kinematics): type 2 applies the orientation quaternion (normalised, as mj_kinematics
       * does) and its axis is the rotated body x axis; the type-3 links behind it are rigid and carry the body y / z axes */
      V3 a2 = axw;
      if (m->jtype[l] == 2) {
        q = qmul(q, qnormalize(ldq(qp + l)));
        qmat(q, R);
        x = sub(anchor, mulv(R, ld3(m->jpos[l])));
        a2 = mulv(R, ax);
      }
      const V3 sv = cross(anchor, a2);
      S[l][0] = a2.x; S[l][1] = a2.y; S[l][2] = a2.z; S[l][3] = sv.x; S[l][4] = sv.y; S[l][5] = sv.z;
    } else {
      x = add(x, scl(axw, qp[QA(l)]));
      S[l][0] = S[l][1] = S[l][2] = 0; S[l][3] = axw.x; S[l][4] = axw.y; S[l][5] = axw.z;
    }
    o->Xq[l][0] = q.w; o->Xq[l][1] = q.x; o->Xq[l][2] = q.y; o->Xq[l][3] = q.z;
    o->Xp[l][0] = x.x; o->Xp[l][1] = x.y; o->Xp[l][2] = x.z;
    /* spatial inertia */
    const double mass = m->mass[l];
    const V3 c = add(x, mulv(R, ld3(m->com[l])));
    const double* in = m->inertia[l];
    const double I[3][3] = {{in[0], in[3], in[4]}, {in[3], in[1], in[5]}, {in[4], in[5], in[2]}};
    double T[3][3], W[3][3];
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) T[r][cc] = R[r][0] * I[0][cc] + R[r][1] * I[1][cc] + R[r][2] * I[2][cc];
    for (int r = 0; r < 3; ++r) for (int cc = 0; cc < 3; ++cc) W[r][cc] = T[r][0] * R[cc][0] + T[r][1] * R[cc][1] + T[r][2] * R[cc][2];
    const double c2 = dot(c, c);
    double* i10 = I10[l];
    i10[0] = mass; i10[1] = mass * c.x; i10[2] = mass * c.y; i10[3] = mass * c.z;
    i10[4] = W[0][0] + mass * (c2 - c.x * c.x); i10[5] = W[1][1] + mass * (c2 - c.y * c.y); i10[6] = W[2][2] + mass * (c2 - c.z * c.z);
    i10[7] = W[0][1] - mass * c.x * c.y; i10[8] = W[0][2] - mass * c.x * c.z; i10[9] = W[1][2] - mass * c.y * c.z;
  }
  /* composite inertias, mass matrix */
  memcpy(Ic, I10, sizeof(Ic));
  for (int l = nv - 1; l >= 0; --l)
    if (m->parent[l] >= 0) for (int e = 0; e < 10; ++e) Ic[m->parent[l]][e] += Ic[l][e];
  memset(M, 0, sizeof(M));
  for (int i = 0; i < nv; ++i) {
    V3 n, f;
    iapply(Ic[i], ld3(S[i]), ld3(S[i] + 3), &n, &f);
    for (int j = i; j >= 0; j = m->parent[j]) {
      const double v = dot(ld3(S[j]), n) + dot(ld3(S[j] + 3), f);
      M[i][j] = M[j][i] = v;
    }
    M[i][i] += m->armature[i];
  }
  /* bias forces (RNE) */
  {
    double Vw[NVMAX][6], Aw[NVMAX][6], F[NVMAX][6];
    for (int l = 0; l < nv; ++l) {
      const int p = m->parent[l];
      V3 wp = {0, 0, 0}, vp = {0, 0, 0}, awp = {0, 0, 0}, avp = {-m->gravity[0], -m->gravity[1], -m->gravity[2]};
      if (p >= 0) { wp = ld3(Vw[p]); vp = ld3(Vw[p] + 3); awp = ld3(Aw[p]); avp = ld3(Aw[p] + 3); }
      const V3 sw = ld3(S[l]), sv = ld3(S[l] + 3);
      const double qd = qv[l];
      /* d/dt of the axis: the three rotation axes of a free body all use the velocity before any of them (mj_comVel) */
      V3 wc = wp, vc = vp;
      if (m->jtype[l] == 3) {
        const int pc = m->parent[m->ball_dof];
        wc = pc >= 0 ? ld3(Vw[pc]) : v3(0, 0, 0); vc = pc >= 0 ? ld3(Vw[pc] + 3) : v3(0, 0, 0);
      }
      const V3 aw = add(awp, scl(cross(wc, sw), qd)), av = add(avp, scl(add(cross(vc, sw), cross(wc, sv)), qd));
      const V3 w = add(wp, scl(sw, qd)), v = add(vp, scl(sv, qd));
      V3 n1, f1, n2, f2;
      iapply(I10[l], aw, av, &n1, &f1);
      iapply(I10[l], w, v, &n2, &f2);
      const V3 n = add(n1, add(cross(w, n2), cross(v, f2))), f = add(f1, cross(w, f2));
      Vw[l][0] = w.x; Vw[l][1] = w.y; Vw[l][2] = w.z; Vw[l][3] = v.x; Vw[l][4] = v.y; Vw[l][5] = v.z;
      Aw[l][0] = aw.x; Aw[l][1] = aw.y; Aw[l][2] = aw.z; Aw[l][3] = av.x; Aw[l][4] = av.y; Aw[l][5] = av.z;
      F[l][0] = n.x; F[l][1] = n.y; F[l][2] = n.z; F[l][3] = f.x; F[l][4] = f.y; F[l][5] = f.z;
    }
    for (int l = nv - 1; l >= 0; --l)
      if (m->parent[l] >= 0) for (int e = 0; e < 6; ++e) F[m->parent[l]][e] += F[l][e];
    for (int l = 0; l < nv; ++l) {
      double bias = 0;
      for (int e = 0; e < 6; ++e) bias += S[l][e] * F[l][e];
      tau[l] = -m->damping[l] * qv[l] - bias - m->stiffness[l] * (qp[QA(l)] - m->springref[l]);     /* (+ joint spring, mj_passive) */
      if (qfrc) tau[l] += qfrc[l];
    }
    for (int ac = 0; ac < m->n_act; ++ac) {
      const int j = m->act_joint[ac];
      const double c = fmin(fmax(ctrl[ac], m->act_ctrlrange[ac][0]), m->act_ctrlrange[ac][1]);
      tau[j] += fmin(fmax(m->act_kp[ac] * (c - qp[QA(j)]), m->act_forcerange[ac][0]), m->act_forcerange[ac][1]);   /* forcelimited actuator */
    }
  }
  /* constraint rows */
  double J[NROWMAX][NVMAX], aref[NROWMAX], D[NROWMAX];
  int cpair[EARL_MAXCON];                          /* collision pair of contact slot c */
  int iseq[NROWMAX], rowid[NROWMAX], nr = 0;       /* rowid: position in the efc output (weld 0..5, limits 6 + 2 j + side), -1 otherwise */
  memset(J, 0, sizeof(J));
  if (m->weld_att >= 0) {                            /* (the minitaur model has no mocap weld) */
    const int k = m->weld_att, la = m->att_link[k];
    double R[3][3], Rh[3][3];
    const Q4 ql = ldq(o->Xq[la]);
    qmat(ql, R);
    const V3 hp = add(ld3(o->Xp[la]), mulv(R, ld3(m->att_pos[k])));
    const Q4 hq = qmul(ql, ldq(m->att_quat[k]));
    const Q4 hc = {hq.w, -hq.x, -hq.y, -hq.z};
    const Q4 e = qmul(hc, mq);
    const V3 ev = {e.x, e.y, e.z};
    qmat(hq, Rh);
    const V3 rpos = sub(mpos, hp);
    for (int j = la; j >= 0; j = m->parent[j]) {
      const V3 sw = ld3(S[j]), sv = ld3(S[j] + 3);
      const V3 pv = add(sv, cross(sw, hp));
      const V3 a = mulvT(Rh, sw);
      const V3 jq = add(scl(a, e.w), cross(a, ev));
      J[0][j] = -pv.x; J[1][j] = -pv.y; J[2][j] = -pv.z;
      J[3][j] = -0.5 * jq.x; J[4][j] = -0.5 * jq.y; J[5][j] = -0.5 * jq.z;
    }
    const double res_norm = sqrt(dot(rpos, rpos) + dot(ev, ev));
    for (int r = 0; r < 6; ++r) {
      const double res = r < 3 ? pick(rpos, r) : pick(ev, r - 3);
      double Jv = 0, kk, bb, dd;
      for (int j = 0; j < nv; ++j) Jv += J[r][j] * qv[j];
      kbimp(m->weld_solref, m->weld_solimp, g_weld_norm_imp ? res_norm : res, dt, &kk, &bb, &dd);
      aref[r] = -bb * Jv - kk * dd * res;
      D[r] = 1.0 / fmax((1 - dd) / dd * m->weld_invweight[r < 3 ? 0 : 1] * g_weld_row_scale[r], 1e-15);
      iseq[r] = 1; rowid[r] = r;
    }
    nr = 6;
  }
  for (int j = 0; j < nv; ++j) {
    if (!m->limited[j]) continue;
    for (int side = 0; side < 2; ++side) {
      const double res = side == 0 ? qp[QA(j)] - m->range[j][0] : m->range[j][1] - qp[QA(j)];
      if (!(res < 0)) continue;
      const double sg = side == 0 ? 1.0 : -1.0;
      double kk, bb, dd;
      kbimp(m->jsolref[j], m->jsolimp[j], res, dt, &kk, &bb, &dd);
      J[nr][j] = sg;
      aref[nr] = -bb * (sg * qv[j]) - kk * dd * res;
      D[nr] = 1.0 / fmax((1 - dd) / dd * m->dof_invweight[j], 1e-15);
      iseq[nr] = 0; rowid[nr] = 6 + 2 * j + side;
      ++nr;
    }
  }
  for (int j = 0; j < nv; ++j)
    if (m->drag_G[j] != 0) {                        /* soft velocity row of a permanent dragging contact */
      J[nr][j] = 1.0; aref[nr] = -m->drag_b[j] * qv[j]; D[nr] = m->drag_G[j]; iseq[nr] = 1; rowid[nr] = -1;
      ++nr;
    }
  for (int e = 0; e < m->n_jeq; ++e) {              /* joint couplings q1 - c0 - c1 q2 = 0 (LinkModel.forward) */
    const int j1 = m->jeq_joint1[e], j2 = m->jeq_joint2[e];
    const double c0 = m->jeq_coef[e][0], c1 = m->jeq_coef[e][1], res = qp[QA(j1)] - c0 - c1 * qp[QA(j2)];
    double kk, bb, dd;
    kbimp(m->jeq_solref[e], m->jeq_solimp[e], res, dt, &kk, &bb, &dd);
    J[nr][j1] = 1.0; J[nr][j2] = -c1;
    aref[nr] = -bb * (qv[j1] - c1 * qv[j2]) - kk * dd * res;
    D[nr] = 1.0 / fmax((1 - dd) / dd * m->jeq_invweight[e], 1e-15);
    iseq[nr] = 1; rowid[nr] = -1;
    ++nr;
  }
  for (int e = 0; e < m->n_con; ++e) {              /* connect constraints: attachments con_att1 / con_att2 coincide (LinkModel.forward) */
    V3 p[2];
    for (int t = 0; t < 2; ++t) {
      const int k = t == 0 ? m->con_att1[e] : m->con_att2[e], la = m->att_link[k];
      double R[3][3];
      qmat(ldq(o->Xq[la]), R);
      p[t] = add(ld3(o->Xp[la]), mulv(R, ld3(m->att_pos[k])));
      const double sg = t == 0 ? 1.0 : -1.0;
      for (int j = la; j >= 0; j = m->parent[j]) {
        const V3 pv = scl(add(ld3(S[j] + 3), cross(ld3(S[j]), p[t])), sg);
        J[nr][j] += pv.x; J[nr + 1][j] += pv.y; J[nr + 2][j] += pv.z;
      }
    }
    const V3 r3 = sub(p[0], p[1]);
    for (int c = 0; c < 3; ++c) {
      const double res = pick(r3, c);
      double Jv = 0, kk, bb, dd;
      for (int j = 0; j < nv; ++j) Jv += J[nr][j] * qv[j];
      kbimp(m->con_solref[e], m->con_solimp[e], res, dt, &kk, &bb, &dd);
      aref[nr] = -bb * Jv - kk * dd * res;
      D[nr] = 1.0 / fmax((1 - dd) / dd * m->con_invweight[e], 1e-15);
      iseq[nr] = 1; rowid[nr] = -1;
      ++nr;
    }
  }
  /* dry joint friction: rows with a bounded force (state 0 quadratic, +-1 saturated), kept apart from the unilateral rows */
  int fj[NVMAX], fs[NVMAX], nf = 0;
  double far_[NVMAX], fD[NVMAX], floss[NVMAX];
  for (int j = 0; j < nv; ++j)
    if (m->frictionloss[j] > 0) {
      double kk, bb, dd;
      kbimp(m->jsolref[j], m->jsolimp[j], 0.0, dt, &kk, &bb, &dd);
      fj[nf] = j; fs[nf] = 0; far_[nf] = -bb * qv[j]; fD[nf] = 1.0 / fmax((1 - dd) / dd * m->dof_invweight[j], 1e-15); floss[nf] = m->frictionloss[j];
      ++nf;
    }
  o->ncon = 0;
  double cone_mu[EARL_MAXCON];
  int cdim[EARL_MAXCON], crow[EARL_MAXCON];        /* rows of contact c: crow[c] .. crow[c] + cdim[c] (elliptic: 3, or 4 with the torsional row) */
  int nr_contacts0 = -1;           /* index of the first contact row */
  if (col) {
    int ncon = 0;
    nr_contacts0 = nr;
    for (int b = 0; b < col->n_blk && ncon < col->max_con; ++b) {
      const int bi = col->blk_box[b], bl = col->blk_link[b], xl = col->box_link[bi];
      V3 cs = ld3(col->blk_center[b]), pb = ld3(col->box_pos[bi]);
      Q4 qb = ldq(col->box_quat[bi]);
      double R[3][3], Rb[3][3];
      if (bl >= 0) { qmat(ldq(o->Xq[bl]), R); cs = add(ld3(o->Xp[bl]), mulv(R, cs)); }
      if (xl >= 0) { const Q4 ql = ldq(o->Xq[xl]); qmat(ql, R); pb = add(ld3(o->Xp[xl]), mulv(R, pb)); qb = qmul(ql, qb); }
      qmat(qb, Rb);
      const V3 h = ld3(col->box_half[bi]);
      {
        const V3 x = mulvT(Rb, sub(cs, pb));
        const V3 d = v3(x.x - fmin(fmax(x.x, -h.x), h.x), x.y - fmin(fmax(x.y, -h.y), h.y), x.z - fmin(fmax(x.z, -h.z), h.z));
        if (!(dot(d, d) < col->blk_reach[b] * col->blk_reach[b])) continue;
      }
      {
        /* second bounding test (include/earl_physics.h blk_obb_*): a face axis of the set's box or of the block's box separates them */
        double RA[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        V3 ca = ld3(col->blk_obb_center[b]);
        const V3 ha = ld3(col->blk_obb_half[b]);
        if (bl >= 0) { qmat(ldq(o->Xq[bl]), RA); ca = add(ld3(o->Xp[bl]), mulv(RA, ca)); }
        const V3 t = mulvT(RA, sub(pb, ca));
        const double tt[3] = {t.x, t.y, t.z}, hA[3] = {ha.x, ha.y, ha.z}, hB[3] = {h.x, h.y, h.z};
        double Rm[3][3], aR[3][3];
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 3; ++j) {
            Rm[i][j] = RA[0][i] * Rb[0][j] + RA[1][i] * Rb[1][j] + RA[2][i] * Rb[2][j];
            aR[i][j] = fabs(Rm[i][j]);
          }
        int separated = 0;
        for (int i = 0; i < 3; ++i) separated |= fabs(tt[i]) > hA[i] + (aR[i][0] * hB[0] + aR[i][1] * hB[1] + aR[i][2] * hB[2]);
        for (int j = 0; j < 3; ++j)
          separated |= fabs(tt[0] * Rm[0][j] + tt[1] * Rm[1][j] + tt[2] * Rm[2][j]) > hB[j] + (hA[0] * aR[0][j] + hA[1] * aR[1][j] + hA[2] * aR[2][j]);
        if (separated) continue;
      }
      int btaken = 0;                                /* contacts of this block: at most blk_cap[b] (bits 0-7; bit 8: edges vs capsule) */
      const int bcap = col->blk_cap[b] & 255, capsule = (col->blk_cap[b] >> 8) & 1;
      for (int pi = col->blk_begin[b]; pi < col->blk_end[b] && ncon < col->max_con && btaken < bcap; ++pi) {
        const int lk = col->pair_rec[pi].sph_link, cls = col->pair_rec[pi].cls;
        const double r = col->pair_rec[pi].r, margin = col->pair_rec[pi].margin;
        V3 c = ld3(col->pair_rec[pi].pos);
        if (lk >= 0) { qmat(ldq(o->Xq[lk]), R); c = add(ld3(o->Xp[lk]), mulv(R, c)); }
        double dist;
        V3 n, p;
        const int cylinder = col->pair_kind[pi] == 2;
        if (cylinder) {                              /* finite cylinder vs box: one contact from the portal refinement above */
          V3 ax = ld3(col->pair_rec[pi].dir);
          if (lk >= 0) ax = mulv(R, ax);
          if (!mpr_box_cylinder(pb, Rb, h, c, ax, col->pair_rec[pi].hl, r, margin, &dist, &n, &p)) continue;
          if (!(dist < margin)) continue;
        } else
        if (capsule) {                               /* edge (segment) vs capsule: closest points of two segments */
          V3 ed = ld3(col->pair_rec[pi].dir);
          if (lk >= 0) ed = mulv(R, ed);
          const V3 cd = v3(Rb[0][2], Rb[1][2], Rb[2][2]), rr = sub(c, pb);
          const double he = col->pair_rec[pi].hl, hc = h.z - h.x, rad = h.x;
          const double b_ = dot(ed, cd), c_ = dot(ed, rr), f_ = dot(cd, rr), den = 1.0 - b_ * b_;
          double s_ = den > 1e-12 ? fmin(fmax((b_ * f_ - c_) / den, -he), he) : 0.0;
          const double t_ = fmin(fmax(b_ * s_ + f_, -hc), hc);
          s_ = fmin(fmax(b_ * t_ - c_, -he), he);
          const V3 pc = add(pb, scl(cd, t_)), d = sub(add(c, scl(ed, s_)), pc);
          const double nd = sqrt(dot(d, d));
          dist = nd - rad;
          if (!(dist < margin && nd > 1e-9)) continue;
          n = scl(d, 1.0 / nd);
          p = add(pc, scl(n, rad + 0.5 * dist));
        } else {
        const V3 x = mulvT(Rb, sub(c, pb));
        V3 q = v3(fmin(fmax(x.x, -h.x), h.x), fmin(fmax(x.y, -h.y), h.y), fmin(fmax(x.z, -h.z), h.z));
        V3 nl;
        if (fabs(x.x) > h.x || fabs(x.y) > h.y || fabs(x.z) > h.z) {
          const V3 d = sub(x, q);
          const double nd = sqrt(dot(d, d));
          dist = nd - r; nl = scl(d, 1.0 / nd);
        } else {
          const double gx = h.x - fabs(x.x), gy = h.y - fabs(x.y), gz = h.z - fabs(x.z);
          const int ax = (gx <= gy && gx <= gz) ? 0 : (gy <= gz ? 1 : 2);
          const double xa = pick(x, ax), ha = pick(h, ax), sg = xa >= 0 ? 1.0 : -1.0;
          nl = v3(ax == 0 ? sg : 0, ax == 1 ? sg : 0, ax == 2 ? sg : 0);
          q = v3(ax == 0 ? sg * ha : x.x, ax == 1 ? sg * ha : x.y, ax == 2 ? sg * ha : x.z);
          dist = -(ha - fabs(xa)) - r;
        }
        if (!(dist < margin)) continue;
        n = mulv(Rb, nl);
        p = add(add(pb, mulv(Rb, q)), scl(n, 0.5 * dist));
        }
        if (g_dbg_on && ncon < EARL_MAXCON) { double* dc_ = g_dbg_contacts[ncon]; dc_[0] = pi; dc_[1] = dist; dc_[2] = n.x; dc_[3] = n.y; dc_[4] = n.z; dc_[5] = p.x; dc_[6] = p.y; dc_[7] = p.z; }
        /* tangents: n x (the coordinate axis least aligned with n) */
        const double ax_ = fabs(n.x), ay_ = fabs(n.y), az_ = fabs(n.z);
        const int ia = (ax_ <= ay_ && ax_ <= az_) ? 0 : (ay_ <= az_ ? 1 : 2);
        V3 t1 = cross(n, v3(ia == 0, ia == 1, ia == 2));
        t1 = scl(t1, 1.0 / sqrt(dot(t1, t1)));
        const V3 t2 = cross(n, t1);
        double Jn[NVMAX] = {0}, Jt1[NVMAX] = {0}, Jt2[NVMAX] = {0};
        for (int pass = 0; pass < 2; ++pass) {
          const double sgn = pass == 0 ? 1.0 : -1.0;
          for (int j = pass == 0 ? lk : xl; j >= 0; j = m->parent[j]) {
            const V3 Jp = scl(add(ld3(S[j] + 3), cross(ld3(S[j]), p)), sgn);
            Jn[j] += dot(n, Jp); Jt1[j] += dot(t1, Jp); Jt2[j] += dot(t2, Jp);
          }
        }
        const int root_ = m->ball_dof + 2;
        const double mu = (g_foot_mu > 0 && m->ball_dof == 3 && lk > root_ && m->parent[lk] != root_) ? g_foot_mu : col->cls_mu[cls];
        double kk, bb, dd;
        kbimp(col->cls_solref[cls], col->cls_solimp[cls], dist - margin, dt, &kk, &bb, &dd);
        const double R0 = fmax((1 - dd) / dd * col->cls_invw[cls], 1e-15);
        if (col->cone == 1) {   /* elliptic cone: rows (normal, t1, t2), one regulariser (impratio 1), only the normal row has a position term */
          /* condim 4 (round 5): a fourth row, the relative angular velocity about the normal, with torsional coefficient mu_t [length].  MuJoCo's elliptic
           * cone scales friction row i by its coefficient (regulariser R0 mu^2 / mu_i^2, cone in the scaled coordinates): with the torsional row multiplied by
           * mu_t / mu the contact is the same isotropic cone over THREE tangential coordinates */
          const double mu_t = g_torsion_override >= 0 ? (cylinder ? g_torsion_override : 0.0) : col->cls_mu_tor[cls];
          double Jtor[NVMAX] = {0};
          if (mu_t > 0)
            for (int pass = 0; pass < 2; ++pass)
              for (int j = pass == 0 ? lk : xl; j >= 0; j = m->parent[j]) Jtor[j] += (pass == 0 ? 1.0 : -1.0) * (mu_t / mu) * dot(n, ld3(S[j]));
          cdim[ncon] = mu_t > 0 ? 4 : 3; crow[ncon] = nr;
          for (int e = 0; e < cdim[ncon]; ++e) {
            const double* Je = e == 0 ? Jn : (e == 1 ? Jt1 : (e == 2 ? Jt2 : Jtor));
            double vel = 0;
            for (int j = 0; j < nv; ++j) { J[nr][j] = Je[j]; vel += Je[j] * qv[j]; }
            aref[nr] = -bb * vel - (e == 0 ? kk * dd * (dist - margin) : 0.0);
            D[nr] = 1.0 / R0;
            iseq[nr] = 0; rowid[nr] = -1;
            ++nr;
          }
          cone_mu[ncon] = mu;
        } else
        for (int e = 0; e < 4; ++e) {
          const double s1 = e == 0 ? mu : (e == 1 ? -mu : 0.0), s2 = e == 2 ? mu : (e == 3 ? -mu : 0.0);
          double vel = 0;
          for (int j = 0; j < nv; ++j) { J[nr][j] = Jn[j] + s1 * Jt1[j] + s2 * Jt2[j]; vel += J[nr][j] * qv[j]; }
          aref[nr] = -bb * vel - kk * dd * (dist - margin);
          D[nr] = 1.0 / (2 * mu * mu * R0);
          iseq[nr] = 0; rowid[nr] = -1;
          ++nr;
        }
        cpair[ncon] = pi;
        ++ncon; ++btaken;
      }
    }
    o->ncon = ncon;
    if (g_dbg_on) g_dbg_ncon = ncon;
  }
  /* primal active-set Newton (LinkModel.solve_primal; elliptic models: LinkModel.solve_primal_elliptic) */
  int act[NROWMAX];
  double a[NVMAX];
  const int ell = col && col->cone == 1 && o->ncon > 0;
  const int nc = ell ? o->ncon : 0, nru = ell ? nr_contacts0 : nr;       /* rows [nru, nr) are the contacts' (n, t1, t2) triples */
  int zone[EARL_MAXCON];                                                  /* 0 top (no force), 1 bottom (sticking: quadratic), 2 middle (sliding: on the cone) */
  double Jak[EARL_MAXCON][4];                                             /* J a of the iterate the zone was read from */
  for (int r = 0; r < nr; ++r) act[r] = 1;
  for (int c = 0; c < nc; ++c) { zone[c] = 1; Jak[c][0] = Jak[c][1] = Jak[c][2] = Jak[c][3] = 0; }
#define CONE_ZONE(r0, rho, mu) ((r0) >= (mu) * (rho) ? 0 : ((rho) <= -(mu) * (r0) ? 1 : 2))
#define CONE_RHO(q, dimc) sqrt((q)[1] * (q)[1] + (q)[2] * (q)[2] + ((dimc) > 3 ? (q)[3] * (q)[3] : 0.0))
  if (g_warm_start && o->warm) {
    /* (the dry-friction rows keep their cold start, the quadratic zone: started from a_prev's zones the three-state iteration cycled 18 times
     * as often in the kitchen model -- 3,295 against 181 of 1 M timesteps used all 8 iterations; with this rule 33) */
    for (int r = 0; r < nru; ++r) {
      double x = -aref[r];
      for (int j = 0; j < nv; ++j) x += J[r][j] * o->qacc[j];
      act[r] = iseq[r] || x < 0;
    }
    for (int c = 0; c < nc; ++c) {
      double q[4] = {0, 0, 0, 0};
      for (int k = 0; k < cdim[c]; ++k) {
        double x = 0;
        for (int j = 0; j < nv; ++j) x += J[crow[c] + k][j] * o->qacc[j];
        Jak[c][k] = x; q[k] = x - aref[crow[c] + k];
      }
      zone[c] = CONE_ZONE(q[0], CONE_RHO(q, cdim[c]), cone_mu[c]);
    }
  }
  const int carry = g_carry_sets && o->carry && col && col->cone != 1, ncs = carry ? o->ncon : 0, cbase = nr_contacts0;      /* (pyramid contacts: four rows each, slot c at cbase + 4 c) */
  if (carry && g_warm_start && o->warm)
    for (int c = 0; c < ncs && c < o->pncon; ++c)
      if (o->ppair[c] == cpair[c])
        for (int e = 0; e < 4; ++e) act[cbase + 4 * c + e] = o->pact[c][e];
  int iters = 0, converged = 0;
  for (int it = 0; it < 8; ++it) {
    ++iters;
    double H[NVMAX][NVMAX];
    for (int i = 0; i < nv; ++i) { a[i] = tau[i]; for (int j = 0; j < nv; ++j) H[i][j] = M[i][j]; }
    for (int r = 0; r < nru; ++r) {
      if (!act[r]) continue;
      for (int i = 0; i < nv; ++i) {
        if (J[r][i] == 0) continue;
        const double w = D[r] * J[r][i];
        a[i] += w * aref[r];
        for (int j = 0; j < nv; ++j) H[i][j] += w * J[r][j];
      }
    }
    for (int c = 0; c < nc; ++c) {
      if (zone[c] == 0) continue;
      const int r0 = crow[c], dc = cdim[c];
      const double Dn = D[r0], mu = cone_mu[c];
      double Hc[4][4] = {{0}}, h[4] = {0, 0, 0, 0};
      if (zone[c] == 1) {
        for (int k = 0; k < dc; ++k) { Hc[k][k] = Dn; h[k] = Dn * aref[r0 + k]; }
      } else {
        double q[4] = {0, 0, 0, 0}, u[4] = {0, 0, 0, 0};
        for (int k = 0; k < dc; ++k) q[k] = Jak[c][k] - aref[r0 + k];
        const double rho = CONE_RHO(q, dc), K = Dn / (1 + mu * mu), sl = q[0] - mu * rho;
        for (int k = 1; k < dc; ++k) u[k] = q[k] / rho;
        double v[4] = {1.0, 0, 0, 0};
        for (int k = 1; k < dc; ++k) v[k] = -mu * u[k];
        const double qq = -K * mu * sl / rho;
        for (int k = 0; k < dc; ++k) for (int l = 0; l < dc; ++l) Hc[k][l] = K * v[k] * v[l];
        for (int k = 1; k < dc; ++k) for (int l = 1; l < dc; ++l) Hc[k][l] += qq * ((k == l ? 1.0 : 0.0) - u[k] * u[l]);
        for (int k = 0; k < dc; ++k) {
          double x = 0;
          for (int l = 0; l < dc; ++l) x += Hc[k][l] * Jak[c][l];
          h[k] = x - K * sl * v[k];
        }
      }
      for (int i = 0; i < nv; ++i) {
        double w[4] = {0, 0, 0, 0}, any = 0;
        for (int k = 0; k < dc; ++k) {
          any += fabs(J[r0 + k][i]);
          for (int l = 0; l < dc; ++l) w[k] += Hc[k][l] * J[r0 + l][i];
        }
        if (any == 0) continue;
        for (int k = 0; k < dc; ++k) a[i] += J[r0 + k][i] * h[k];
        for (int j = 0; j < nv; ++j) {
          double x = 0;
          for (int k = 0; k < dc; ++k) x += w[k] * J[r0 + k][j];
          H[i][j] += x;
        }
      }
    }
    for (int k = 0; k < nf; ++k) {
      if (fs[k] == 0) { H[fj[k]][fj[k]] += fD[k]; a[fj[k]] += fD[k] * far_[k]; }
      else a[fj[k]] -= fs[k] * floss[k];
    }
    chol_solve(nv, H, a);
    int changed = 0;
    for (int k = 0; k < nf; ++k) {
      const double x = a[fj[k]] - far_[k];
      const int ns = fabs(x) * fD[k] <= floss[k] ? 0 : (x > 0 ? 1 : -1);
      changed |= ns != fs[k];
      fs[k] = ns;
    }
    for (int r = 0; r < nru; ++r) {
      double x = -aref[r];
      for (int j = 0; j < nv; ++j) x += J[r][j] * a[j];
      const int want = iseq[r] || x < 0;
      changed |= want != act[r];
      act[r] = want;
    }
    for (int c = 0; c < nc; ++c) {
      double Jn_[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0}, big = 0, dif = 0;
      for (int k = 0; k < cdim[c]; ++k) {
        double x = 0;
        for (int j = 0; j < nv; ++j) x += J[crow[c] + k][j] * a[j];
        Jn_[k] = x; q[k] = x - aref[crow[c] + k];
        big = fmax(big, fabs(Jak[c][k])); dif = fmax(dif, fabs(x - Jak[c][k]));
      }
      const int z = CONE_ZONE(q[0], CONE_RHO(q, cdim[c]), cone_mu[c]);
      if (z != zone[c] || (z == 2 && dif > 1e-8 * (1.0 + big))) changed = 1;      /* LinkModel.ELL_TOL */
      zone[c] = z;
      for (int k = 0; k < cdim[c]; ++k) { Jak[c][k] = Jn_[k]; act[crow[c] + k] = z != 0; }
    }
    if (!changed) { converged = 1; break; }
  }
  {
    const int hasc = col && o->ncon > 0;
#pragma omp atomic
    g_newton_stats[4] += !converged;
#pragma omp atomic
    g_newton_stats[0] += 1;
#pragma omp atomic
    g_newton_stats[1] += iters;
#pragma omp atomic
    g_newton_stats[2] += hasc;
#pragma omp atomic
    g_newton_stats[3] += hasc ? iters : 0;
  }
  if (carry) {
    o->pncon = ncs;
    for (int c = 0; c < ncs; ++c) { o->ppair[c] = cpair[c]; for (int e = 0; e < 4; ++e) o->pact[c][e] = act[cbase + 4 * c + e]; }
  }
  o->warm = integrate ? 1 : 0;
  for (int i = 0; i < nv; ++i) o->qacc[i] = a[i];
  memset(o->efc, 0, sizeof(o->efc));
  for (int r = 0; r < nr; ++r) {
    if (rowid[r] < 0 || !act[r]) continue;
    double x = -aref[r];
    for (int j = 0; j < nv; ++j) x += J[r][j] * a[j];
    o->efc[rowid[r]] = -D[r] * x;
  }
  if (integrate) {   /* semi-implicit Euler, joint damping implicit: (M + dt B) a' = M a */
    double H[NVMAX][NVMAX], rhs[NVMAX];
    for (int i = 0; i < nv; ++i) {
      rhs[i] = 0;
      for (int j = 0; j < nv; ++j) { rhs[i] += M[i][j] * a[j]; H[i][j] = M[i][j]; }
      H[i][i] += dt * m->damping[i];
    }
    chol_solve(nv, H, rhs);
    const int bd = m->ball_dof;
    for (int i = 0; i < nv; ++i) { qv[i] += dt * rhs[i]; if (bd < 0 || i < bd || i > bd + 2) qp[QA(i)] += dt * qv[i]; }
    if (bd >= 0) {   /* mju_quatIntegrate (LinkModel.integrate_pos) */
      const V3 w = ld3(qv + bd);
      const double nw = sqrt(dot(w, w));
      Q4 q = qnormalize(ldq(qp + bd));
      if (nw > 0) {
        const double ang = dt * nw, sn = sin(0.5 * ang) / nw;
        const Q4 r = {cos(0.5 * ang), sn * w.x, sn * w.y, sn * w.z};
        q = qnormalize(qmul(q, r));
      }
      qp[bd] = q.w; qp[bd + 1] = q.x; qp[bd + 2] = q.y; qp[bd + 3] = q.z;
    }
  }
}

static V3 attachment(const LM* m, const StepOut* o, int k) {
  const int la = m->att_link[k];
  V3 p = ld3(m->att_pos[k]);
  if (la >= 0) { double R[3][3]; qmat(ldq(o->Xq[la]), R); p = add(ld3(o->Xp[la]), mulv(R, p)); }
  return p;
}

int oracle_set_physics_threads(int n) {
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
  return omp_get_max_threads();
#else
  (void)n;
  return 1;
#endif
}

/* earl_physics_step / earl_physics_forward on host arrays (integrate: 1 = step nsub times, 0 = forward quantities); ctrl_stride: doubles per env
 * in ctrl (0: n_act); mq_stride: 4 = a quaternion per env, 0 = one for all */
static int physics24(const LM* m, const earl_collision_model* col, int32_t n, int32_t nsub, int32_t integrate, double* qpos, double* qvel,
                     const double* mocap_pos, const double* mocap_quat, const double* ctrl, double* qacc, double* efc, double* att_xpos, int32_t* ncon) {
  const int nv = m->nv;
#pragma omp parallel for schedule(static)
  for (int e = 0; e < n; ++e) {
    StepOut o;
    o.warm = 0; o.carry = 0;
    const V3 mpos = ld3(mocap_pos + 3 * (size_t)e);
    const Q4 mq = g_raw_mocap_quat ? ldq(mocap_quat + 4 * (size_t)e) : qnormalize(ldq(mocap_quat + 4 * (size_t)e));
    for (int ts = 0; ts < (integrate ? nsub : 1); ++ts)
      substep(m, col, qpos + (size_t)e * m->nq, qvel + (size_t)e * nv, mpos, mq, ctrl + (size_t)e * m->n_act, integrate, &o, NULL);
    if (qacc) memcpy(qacc + (size_t)e * nv, o.qacc, sizeof(double) * nv);
    if (efc) memcpy(efc + (size_t)e * (6 + 2 * nv), o.efc, sizeof(double) * (6 + 2 * nv));
    if (ncon) ncon[e] = o.ncon;
    if (att_xpos)
      for (int k = 0; k < m->n_att; ++k) {
        const V3 p = attachment(m, &o, k);
        double* d = att_xpos + ((size_t)e * m->n_att + k) * 3;
        d[0] = p.x; d[1] = p.y; d[2] = p.z;
      }
  }
  return 0;
}
int oracle_physics(const earl_link_model* m16, const earl_collision_model* col, int32_t n, int32_t nsub, int32_t integrate, double* qpos,
                   double* qvel, const double* mocap_pos, const double* mocap_quat, const double* ctrl, double* qacc, double* efc,
                   double* att_xpos, int32_t* ncon) {
  LM m;
  widen(m16, &m);
  return physics24(&m, col, n, nsub, integrate, qpos, qvel, mocap_pos, mocap_quat, ctrl, qacc, efc, att_xpos, ncon);
}
int oracle_physics24(const earl_link_model24* m, const earl_collision_model* col, int32_t n, int32_t nsub, int32_t integrate, double* qpos,
                     double* qvel, const double* mocap_pos, const double* mocap_quat, const double* ctrl, double* qacc, double* efc,
                     double* att_xpos, int32_t* ncon) {
  return physics24(m, col, n, nsub, integrate, qpos, qvel, mocap_pos, mocap_quat, ctrl, qacc, efc, att_xpos, ncon);
}

static double tolerance_gaussian(double x, double hi, double margin) {
  if (0.0 <= x && x <= hi) return 1.0;
  if (margin == 0) return 0.0;
  const double d = (x < 0.0 ? -x : x - hi) / margin, scale = sqrt(-2.0 * log(0.1));
  return exp(-0.5 * (d * scale) * (d * scale));
}

void oracle_philox4x32_10(const uint32_t* ctr, const uint32_t* key, uint32_t* out);   /* tabletop_oracle.c */
static double u01_(uint32_t lo, uint32_t hi) { return (double)((((uint64_t)hi << 32) | lo) >> 11) * (1.0 / 9007199254740992.0); }

/* earl_sawyer_rollout on host arrays (same cfg / state / out structs with host pointers); reference: oracle/sawyer_oracle.py */
int oracle_sawyer_rollout(const earl_link_model* m16, const earl_collision_model* col, const earl_sawyer_cfg* cfg, const earl_sawyer_state* st,
                          const float* action, int32_t T, const earl_sawyer_out* out) {
  LM mw;
  widen(m16, &mw);
  const LM* m = &mw;
  const int nv = m->nv, n = cfg->n;
  const float scale = (float)cfg->action_scale;
#pragma omp parallel for schedule(static)
  for (int e = 0; e < n; ++e) {
    StepOut o;
    o.carry = 0;
    double* qp = st->qpos + (size_t)e * m->nq;
    double* qv = st->qvel + (size_t)e * nv;
    double* mp = st->mocap_pos + (size_t)e * 3;
    double* goal = st->goal + (size_t)e * 7;
    const Q4 mq = g_raw_mocap_quat ? ldq(cfg->mocap_quat) : qnormalize(ldq(cfg->mocap_quat));
    int steps = st->steps_since_reset ? st->steps_since_reset[e] : 0;
    const int gcf = st->steps_since_goal_change ? cfg->goal_change_frequency : 0;
    int sgc = gcf > 0 ? st->steps_since_goal_change[e] : 0;
    for (int t = 0; t < T; ++t) {
      const float* a = action + ((size_t)t * n + e) * 4;
      const double ctrl[EARL_MAXACT] = {(double)a[3], -(double)a[3], 0, 0};
      /* failure guard (include/earl_physics.h, earl_sawyer_out.status): work on a copy, commit it only when the step ended finite */
      double q2[NVMAX + 1], v2[NVMAX], mp2[3];
      memcpy(q2, qp, sizeof(double) * m->nq); memcpy(v2, qv, sizeof(double) * nv);
      for (int k = 0; k < 3; ++k) {
        const float c = fminf(fmaxf(a[k], -1.f), 1.f) * scale;
        mp2[k] = fmin(fmax(mp[k] + (double)c, cfg->mocap_low[k]), cfg->mocap_high[k]);
      }
      o.warm = 0;                                     /* every env step starts cold: step-by-step and fused rollouts agree */
      for (int ts = 0; ts < cfg->frame_skip; ++ts) substep(m, col, q2, v2, ld3(mp2), mq, ctrl, 1, &o, NULL);
      const size_t row = (size_t)t * n + e;
      double* ob = out->obs + row * 14;
      int failed = 0;
      for (int k = 0; k < m->nq; ++k) failed |= !(fabs(q2[k]) < EARL_BAD_VALUE);
      for (int k = 0; k < nv; ++k) failed |= !(fabs(v2[k]) < EARL_BAD_VALUE);
      ++steps;
      if (out->status) out->status[row] = failed ? EARL_STEP_DIVERGED : 0;
      if (out->done) out->done[row] = (cfg->horizon > 0 && steps >= cfg->horizon) ? 1 : 0;
      if (failed) {
        const double* prev = t > 0 ? out->obs + ((size_t)(t - 1) * n + e) * 14 : (st->last_obs ? st->last_obs + (size_t)e * 14 : NULL);
        for (int k = 0; k < 14; ++k) ob[k] = prev ? prev[k] : NAN;
        if (out->reward) out->reward[row] = 0.f;
        if (out->success) out->success[row] = 0;
        if (st->fail_count) st->fail_count[e] += 1;
      } else {
      memcpy(qp, q2, sizeof(double) * m->nq); memcpy(qv, v2, sizeof(double) * nv); memcpy(mp, mp2, sizeof(mp2));
      const V3 hand = attachment(m, &o, cfg->att_hand), rr = attachment(m, &o, cfg->att_right), ll = attachment(m, &o, cfg->att_left),
               obj = attachment(m, &o, cfg->att_obj);
      const V3 dg = sub(rr, ll);
      ob[0] = hand.x; ob[1] = hand.y; ob[2] = hand.z;
      ob[3] = fmin(fmax(sqrt(dg.x * dg.x + dg.y * dg.y + dg.z * dg.z) / 0.1, 0.0), 1.0);
      ob[4] = obj.x; ob[5] = obj.y; ob[6] = obj.z;
      for (int k = 0; k < 7; ++k) ob[7 + k] = goal[k];
      const V3 target = ld3(goal + 4), d = sub(obj, target);
      const double obj_to_target = sqrt(d.x * d.x + d.y * d.y + d.z * d.z);
      const int ok = obj_to_target <= cfg->success_radius;
      double r = ok ? 1.0 : 0.0;
      if (cfg->reward_type != 0) {
        const V3 e1 = sub(hand, obj), oi = sub(ld3(cfg->obj_init_pos), target), hi = sub(ld3(cfg->hand_init_pos), obj);
        const double in_place = tolerance_gaussian(obj_to_target, 0.05, sqrt(oi.x * oi.x + oi.y * oi.y + oi.z * oi.z));
        const double hand_in_place = tolerance_gaussian(sqrt(e1.x * e1.x + e1.y * e1.y + e1.z * e1.z), 0.25 * 0.05, sqrt(hi.x * hi.x + hi.y * hi.y + hi.z * hi.z) + 0.1);
        r = 3 * hand_in_place + 6 * in_place;
        if (obj_to_target < 0.05) r = 10;
      }
      if (out->reward) out->reward[row] = (float)r;
      if (out->success) out->success[row] = (uint8_t)ok;
      }
      if (gcf > 0 && ++sgc >= gcf) {   /* LifelongWrapper.step, lifelong_wrapper.py:36-42: new goal, observation re-read with it */
        sgc = 0;
        if (cfg->n_goal_rows > 0 && cfg->goal_table) {
          const uint64_t ev = cfg->step_counter + (uint64_t)t;
          const uint32_t ctr[4] = {0xFFFEu, (uint32_t)(cfg->env_offset + e), (uint32_t)ev, (uint32_t)(ev >> 32)};
          const uint32_t key[2] = {(uint32_t)cfg->seed, (uint32_t)(cfg->seed >> 32)};
          uint32_t b[4];
          oracle_philox4x32_10(ctr, key, b);
          int grow = (int)(u01_(b[0], b[1]) * (double)cfg->n_goal_rows);
          if (grow >= cfg->n_goal_rows) grow = cfg->n_goal_rows - 1;
          for (int k = 0; k < 7; ++k) { goal[k] = cfg->goal_table[(size_t)grow * 7 + k]; ob[7 + k] = goal[k]; }
        }
      }
    }
    if (st->steps_since_reset) st->steps_since_reset[e] = steps;
    if (gcf > 0) st->steps_since_goal_change[e] = sgc;
    if (st->last_obs && T > 0) memcpy(st->last_obs + (size_t)e * 14, out->obs + ((size_t)(T - 1) * n + e) * 14, sizeof(double) * 14);
  }
  return 0;
}

/* ---------------------------------------------------------------------------------------------------------------------------------------------
 * Minitaur env on the same stepper (restates oracle/minitaur_oracle.py, which follows earl_benchmark/envs/minitaur_gym_env.py:222-329, 466-546 and
 * envs/minitaur.py:300-457): host-array form of earl_minitaur_rollout / earl_minitaur_reset (include/earl_physics.h).  PARITY WITH THE REFERENCE'S
 * PYBULLET SIMULATION IS UNPINNED AND MODEL-LESS: the robot model is this build's own (tools/minitaur_model.py). */
int oracle_minitaur_leg_to_motor(int32_t n, const double* action, double* motor_angle);                       /* glue_oracle.c: minitaur.py:434-457 */
struct earl_motor_params_ { double kp, kd, voltage, viscous_damping; int32_t torque_control; };              /* = earl_motor_params (include/earl_glue.h) */
int oracle_minitaur_motor_torque(int32_t m, const struct earl_motor_params_* p, const double* command, const double* angle, const double* velocity,
                                 double* actual_torque, double* observed_torque);                             /* glue_oracle.c: motor.py:49-94 */

typedef struct { double voltage, viscous; double* observed; int32_t* overheat; uint8_t* enabled; } MtMotors;

/* Minitaur.ApplyAction (minitaur.py:326-390) -> generalized forces qfrc [nv] */
static void mt_apply_action(const LM* m, const earl_minitaur_cfg* cfg, const double* qp, const double* qv, const double* cmd, MtMotors* mt, double* qfrc) {
  double q[8], qd[8], c[8], actual[8], observed[8];
  const double lim = m->dt * cfg->motor_velocity_limit;
  for (int i = 0; i < 8; ++i) {
    const int d = cfg->motor_dof[i];
    q[i] = qp[QA(d)] * cfg->motor_dir[i];                                  /* GetMotorAngles :392-404 */
    qd[i] = qv[d] * cfg->motor_dir[i];
    c[i] = fmin(fmax(cmd[i], q[i] - lim), q[i] + lim);                     /* :339-343 (np.clip) */
  }
  const struct earl_motor_params_ p = {cfg->motor_kp, cfg->motor_kd, mt->voltage, mt->viscous, 0};
  oracle_minitaur_motor_torque(8, &p, c, q, qd, actual, observed);
  for (int l = 0; l < m->nv; ++l) qfrc[l] = 0;
  for (int i = 0; i < 8; ++i) {
    if (fabs(actual[i]) > cfg->overheat_torque) mt->overheat[i] += 1; else mt->overheat[i] = 0;     /* :351-358 */
    if (mt->overheat[i] > cfg->overheat_steps) mt->enabled[i] = 0;
    mt->observed[i] = observed[i];
    qfrc[cfg->motor_dof[i]] = mt->enabled[i] ? actual[i] * cfg->motor_dir[i] : 0.0;
  }
}
static void mt_observe(const LM* m, const earl_minitaur_cfg* cfg, const double* qp, const double* qv, const double* observed, const double* goal, double* ob) {
  for (int i = 0; i < 8; ++i) {
    const int d = cfg->motor_dof[i];
    ob[i] = qp[QA(d)] * cfg->motor_dir[i]; ob[8 + i] = qv[d] * cfg->motor_dir[i]; ob[16 + i] = observed[i];
  }
  const int bd = m->ball_dof;
  ob[24] = qp[bd + 1]; ob[25] = qp[bd + 2]; ob[26] = qp[bd + 3]; ob[27] = qp[bd];      /* Bullet's (x, y, z, w) */
  ob[28] = qp[0]; ob[29] = qp[1]; ob[30] = goal[0]; ob[31] = goal[1];
}
static double mt_draw(const earl_minitaur_cfg* cfg, int k, int e, uint64_t counter, uint32_t stream) {
  const uint32_t ctr[4] = {stream + (uint32_t)k, (uint32_t)(cfg->env_offset + e), (uint32_t)counter, (uint32_t)(counter >> 32)};
  const uint32_t key[2] = {(uint32_t)cfg->seed, (uint32_t)(cfg->seed >> 32)};
  uint32_t b[4];
  oracle_philox4x32_10(ctr, key, b);
  return u01_(b[0], b[1]);
}

/* the env's own copy of the model tables with what the randomizer set at its last reset (earl_minitaur_state.motor_param row: voltage, damping, mass
 * factor of the root body / the upper links / the lower links, foot friction): mass and inertia x factor, centre of mass kept */
static void mt_env_model(const earl_link_model24* m, const double* mp, earl_link_model24* d) {
  *d = *m;
  const int root = m->ball_dof + 2;
  for (int l = root; l < m->nv; ++l) {
    const double f = l == root ? mp[2] : (m->parent[l] == root ? mp[3] : mp[4]);
    d->mass[l] = m->mass[l] * f;
    for (int k = 0; k < 6; ++k) d->inertia[l][k] = m->inertia[l][k] * f;
  }
  g_foot_mu = mp[5];
}
int oracle_minitaur_reset(const earl_link_model24* m0, const earl_collision_model* col, const earl_minitaur_cfg* cfg, const earl_minitaur_state* st,
                          const uint8_t* mask, double* obs) {
  const int nv = m0->nv, n = cfg->n;
#pragma omp parallel for schedule(static)
  for (int e = 0; e < n; ++e) {
    if (mask && !mask[e]) continue;
    double* qp = st->qpos + (size_t)e * m0->nq;
    double* qv = st->qvel + (size_t)e * nv;
    double* goal = st->goal + (size_t)e * 2;
    int gi = (int)(mt_draw(cfg, 0, e, cfg->counter, 0x4D00u) * (double)cfg->n_goals);       /* get_next_goal :490-493 */
    if (gi >= cfg->n_goals) gi = cfg->n_goals - 1;
    goal[0] = cfg->goal_table[2 * gi]; goal[1] = cfg->goal_table[2 * gi + 1];
    double* mp = st->motor_param + (size_t)e * 6;
    /* MinitaurEnvRandomizer.randomize_env [UPSTREAM] through Minitaur.SetBatteryVoltage / SetMotorViscousDamping / SetBaseMass / SetLegMasses / SetFootFriction */
    mp[0] = 16.0; mp[1] = 0.0; mp[2] = mp[3] = mp[4] = 1.0; mp[5] = -1.0;
    if (cfg->randomize & 1) { mp[0] = 14.8 + (16.8 - 14.8) * mt_draw(cfg, 1, e, cfg->counter, 0x4D00u); mp[1] = 0.01 * mt_draw(cfg, 2, e, cfg->counter, 0x4D00u); }
    if (cfg->randomize & 2) {
      const int root = m0->ball_dof + 2;
      const double leg = cfg->leg_mass * (1.0 + cfg->leg_mass_err[0] + (cfg->leg_mass_err[1] - cfg->leg_mass_err[0]) * mt_draw(cfg, 4, e, cfg->counter, 0x4D00u));
      const double motor = cfg->motor_mass * (1.0 + cfg->leg_mass_err[0] + (cfg->leg_mass_err[1] - cfg->leg_mass_err[0]) * mt_draw(cfg, 5, e, cfg->counter, 0x4D00u));
      mp[2] = 1.0 + cfg->base_mass_err[0] + (cfg->base_mass_err[1] - cfg->base_mass_err[0]) * mt_draw(cfg, 3, e, cfg->counter, 0x4D00u);
      mp[3] = (motor + leg) / m0->mass[root + 1];          /* the first upper link (parent = root) and its lower link: every leg has the same two masses */
      mp[4] = leg / m0->mass[root + 2];
    }
    if (cfg->randomize & 4) mp[5] = cfg->foot_friction[0] + (cfg->foot_friction[1] - cfg->foot_friction[0]) * mt_draw(cfg, 6, e, cfg->counter, 0x4D00u);
    earl_link_model24 me;
    mt_env_model(m0, mp, &me);
    const earl_link_model24* m = &me;
    memcpy(qp, cfg->reset_qpos, sizeof(double) * m->nq);
    for (int k = 0; k < nv; ++k) qv[k] = 0;
    MtMotors mt = {mp[0], mp[1], st->observed_torque + (size_t)e * 8, st->overheat + (size_t)e * 8, st->motor_enabled + (size_t)e * 8};
    for (int i = 0; i < 8; ++i) { mt.observed[i] = 0; mt.overheat[i] = 0; mt.enabled[i] = 1; }
    StepOut o;
    o.warm = 0; o.carry = 1; o.pncon = 0;
    double cmd[8], qfrc[NVMAX];
    for (int i = 0; i < 8; ++i) cmd[i] = 3.141592653589793 / 2;
    const Q4 mq = {1, 0, 0, 0};
    for (int s = 0; s < cfg->settle_steps; ++s) {                                             /* minitaur_gym_env.py:265-269 */
      mt_apply_action(m, cfg, qp, qv, cmd, &mt, qfrc);
      substep(m, col, qp, qv, v3(0, 0, 0), mq, NULL, 1, &o, qfrc);
    }
    if (st->steps_since_reset) st->steps_since_reset[e] = 0;
    if (st->steps_since_goal_change) st->steps_since_goal_change[e] = 0;
    double ob[32];
    mt_observe(m, cfg, qp, qv, mt.observed, goal, ob);
    if (obs) memcpy(obs + (size_t)e * 32, ob, sizeof(ob));
    if (st->last_obs) memcpy(st->last_obs + (size_t)e * 32, ob, sizeof(ob));
    g_foot_mu = -1.0;                                    /* (the override is this env's only) */
  }
  return 0;
}

int oracle_minitaur_rollout(const earl_link_model24* m0, const earl_collision_model* col, const earl_minitaur_cfg* cfg, const earl_minitaur_state* st,
                            const float* action, int32_t T, const earl_minitaur_out* out) {
  const int nv = m0->nv, n = cfg->n;
#pragma omp parallel for schedule(static)
  for (int e = 0; e < n; ++e) {
    StepOut o;
    o.carry = 1; o.pncon = 0;
    double* qp = st->qpos + (size_t)e * m0->nq;
    double* qv = st->qvel + (size_t)e * nv;
    double* goal = st->goal + (size_t)e * 2;
    const double* mp = st->motor_param + (size_t)e * 6;
    earl_link_model24 me;
    mt_env_model(m0, mp, &me);
    const earl_link_model24* m = &me;
    int steps = st->steps_since_reset ? st->steps_since_reset[e] : 0;
    const int gcf = st->steps_since_goal_change ? cfg->goal_change_frequency : 0;
    int sgc = gcf > 0 ? st->steps_since_goal_change[e] : 0;
    const Q4 mq = {1, 0, 0, 0};
    for (int t = 0; t < T; ++t) {
      const float* a = action + ((size_t)t * n + e) * 8;
      double a64[8], cmd[8], qfrc[NVMAX];
      for (int k = 0; k < 8; ++k) a64[k] = fmin(fmax((double)a[k], -1.01), 1.01);             /* (the front end raises beyond the reference's bound) */
      oracle_minitaur_leg_to_motor(1, a64, cmd);
      /* failure guard: work on copies, commit only when the env step ended finite */
      double q2[NVMAX + 1], v2[NVMAX], obs2[8];
      int32_t oh2[8];
      uint8_t en2[8];
      memcpy(q2, qp, sizeof(double) * m->nq); memcpy(v2, qv, sizeof(double) * nv);
      memcpy(obs2, st->observed_torque + (size_t)e * 8, sizeof(obs2)); memcpy(oh2, st->overheat + (size_t)e * 8, sizeof(oh2)); memcpy(en2, st->motor_enabled + (size_t)e * 8, sizeof(en2));
      MtMotors mt = {mp[0], mp[1], obs2, oh2, en2};
      if (!(g_warm_across_steps && t > 0)) o.warm = 0;
      for (int s = 0; s < cfg->num_substeps; ++s) {                                           /* minitaur_gym_env.py:321-323 */
        mt_apply_action(m, cfg, q2, v2, cmd, &mt, qfrc);
        substep(m, col, q2, v2, v3(0, 0, 0), mq, NULL, 1, &o, qfrc);
      }
      const size_t row = (size_t)t * n + e;
      double* ob = out->obs + row * 32;
      int failed = 0;
      for (int k = 0; k < m->nq; ++k) failed |= !(fabs(q2[k]) < EARL_BAD_VALUE);
      for (int k = 0; k < nv; ++k) failed |= !(fabs(v2[k]) < EARL_BAD_VALUE);
      ++steps;
      if (out->status) out->status[row] = failed ? EARL_STEP_DIVERGED : 0;
      if (out->done) out->done[row] = (cfg->horizon > 0 && steps >= cfg->horizon) ? 1 : 0;
      if (failed) {
        const double* prev = t > 0 ? out->obs + ((size_t)(t - 1) * n + e) * 32 : (st->last_obs ? st->last_obs + (size_t)e * 32 : NULL);
        for (int k = 0; k < 32; ++k) ob[k] = prev ? prev[k] : NAN;
        if (out->reward) out->reward[row] = 0.0;
        if (out->success) out->success[row] = 0;
        if (st->fail_count) st->fail_count[e] += 1;
      } else {
        memcpy(qp, q2, sizeof(double) * m->nq); memcpy(qv, v2, sizeof(double) * nv);
        memcpy(st->observed_torque + (size_t)e * 8, obs2, sizeof(obs2)); memcpy(st->overheat + (size_t)e * 8, oh2, sizeof(oh2)); memcpy(st->motor_enabled + (size_t)e * 8, en2, sizeof(en2));
        mt_observe(m, cfg, qp, qv, obs2, goal, ob);
        /* _reward :505-521 (= compute_reward :529-535 on this observation), is_successful :495-503 */
        const double xd = ob[28] - goal[0], yd = ob[29] - goal[1];
        double dotp = 0.0;
        for (int k = 0; k < 8; ++k) dotp = fma(ob[16 + k], ob[8 + k], dotp);
        if (out->reward) out->reward[row] = cfg->distance_weight * (-fabs(xd) - fabs(yd)) - cfg->energy_weight * (fabs(dotp) * m->dt);
        if (out->success) out->success[row] = (uint8_t)(sqrt(xd * xd + yd * yd) < cfg->success_radius);
      }
      if (gcf > 0 && ++sgc >= gcf) {   /* LifelongWrapper.step, lifelong_wrapper.py:36-42: new goal, observation re-read with it */
        sgc = 0;
        int gi = (int)(mt_draw(cfg, 0, e, cfg->step_counter + (uint64_t)t, 0xFFFEu) * (double)cfg->n_goals);
        if (gi >= cfg->n_goals) gi = cfg->n_goals - 1;
        goal[0] = cfg->goal_table[2 * gi]; goal[1] = cfg->goal_table[2 * gi + 1];
        ob[30] = goal[0]; ob[31] = goal[1];
      }
    }
    if (st->steps_since_reset) st->steps_since_reset[e] = steps;
    if (gcf > 0) st->steps_since_goal_change[e] = sgc;
    if (st->last_obs && T > 0) memcpy(st->last_obs + (size_t)e * 32, out->obs + ((size_t)(T - 1) * n + e) * 32, sizeof(double) * 32);
    g_foot_mu = -1.0;
  }
  return 0;
}
