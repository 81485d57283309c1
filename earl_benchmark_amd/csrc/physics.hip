// physics.hip -- batched articulated-body stepper (include/earl_physics.h) for gfx950.
//
// Work decomposition: a wavefront owns EPW = 64 / LPE env instances, LPE lanes each (LPE = 16 by default: the models
// here have nv <= 16 dofs, so a 64-lane group per env would idle 3/4 of the lanes in every per-link / per-dof phase;
// LPE = 64 -- one wavefront per env -- is kept as an instantiation for comparison, DESIGN.md has both measurements).
// Per-link state lives in LDS (one Shared block per env, the model tables once per workgroup); a wave owns its LDS
// blocks, so phases are separated by wavefront-scope fences only -- no s_barrier anywhere.  fp64 like MuJoCo.
//
// Pipeline per timestep (reference: oracle/physics_oracle.py LinkModel.forward / step):
//   K1 local joint transforms (lane = link)           K2 world transforms by ancestor doubling (log depth rounds via LDS)
//   K3 motion subspace S, spatial inertia (compact additive form m, m c, Io about the world origin)
//   K4 composite inertias = masked subtree sums       K5 mass matrix M[i][j] = S_j . (Ic_i S_i) + armature
//   K6 bias forces (RNE as masked ancestor / subtree sums: V_l = sum S_a qd_a, A_l = g + sum (V_a x S_a) qd_a, ...)
//   K7 tau = actuators + passive damping - bias
//   K8 constraint rows: 6 weld rows to the mocap body (exact quaternion-error Jacobian), one limit row per dof, with
//      MuJoCo's solref / solimp impedance -> reference acceleration aref and regulariser R per row
//   K9 primal solve, as MuJoCo's Newton solver poses it: minimise 1/2 (a-a0)' M (a-a0) + sum_rows 1/(2R) (J a - aref)^2
//      over active rows; the Hessian M + J' D J is nv x nv; the unilateral rows enter by an active-set iteration
//      (Cholesky in registers, redundantly per lane: NV is a compile-time constant)
//   K10 semi-implicit Euler with implicit joint damping: (M + dt B) a' = M a.
//   C0-C3 contacts: block bounding tests (lane = block) -> sphere / point vs box tests of the near blocks (lane = pair,
//      ballot compaction into <= EARL_MAXCON contact records) -> 4 pyramid edges per contact as unilateral rows of K9.
// Parity vs MuJoCo is unpinned; parity vs the reference above is tested to 1e-8 (1e-6 through contacts).
//
// Floating point: this file allows FMA contraction in the dynamics (nothing here is a bit-exact contract); the
// observation / reward epilogue switches it off again so the success flag is the rule applied to the emitted numbers.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <mutex>
#include <type_traits>
#include <unordered_map>

#include "../../include/earl_physics.h"
#include "../../include/earl_glue.h"
#include "philox.h"

#ifndef EARL_PHYS_NO_CONTRACT      // (-DEARL_PHYS_NO_CONTRACT: the units under the command line's -ffp-contract=off, as the tabletop path is built -- the measurement of DESIGN.md "contraction")
#pragma clang fp contract(fast)
#endif

// translation units: physics.hip (the main one: nv 10 / 15 / 23 models, every entry point but the two below), physics_w8.hip (the door model's
// eight-waves-per-CU rollout) and physics_mt.hip (the minitaur: nv = 22) include this file under a variant macro and are compiled side by side
// and physics_l64.hip (round 5: the one-wavefront-per-env instantiations of the Sawyer kernels -- a measurement / test switch, earl_debug_set_physics_lanes(64) -- which were a
// third of the main unit's 50 s of compile time)
// ... and physics_kitchen.hip (round 5: the nv = 23 instantiation, the kitchen env kernels and their entry points)
#if defined(EARL_PHYS_VARIANT_W8) || defined(EARL_PHYS_VARIANT_MT) || defined(EARL_PHYS_UNIT_L64) || defined(EARL_PHYS_UNIT_KITCHEN)
#define EARL_PHYS_NOT_MAIN 1
#endif

// cone word of a device collision table, cached per address (defined in the main unit; -1 = unknown); see check_cone
extern "C" __attribute__((visibility("hidden"))) int earl_unit_table_cone(const void* col, void* stream);

namespace {

// Phase timing (tools/prof_physics.py builds this file with -DEARL_PHYS_PROF into a separate library); not in the product build
#ifdef EARL_PHYS_PROF
__device__ unsigned long long g_phys_prof[32];
__device__ int g_prof_sel[2];                            // the wave whose phases are clocked: (workgroup, first thread of the wave); earl_debug_set_prof_wave*
#define PROF_ME (blockIdx.x == g_prof_sel[0] && threadIdx.x == g_prof_sel[1])
#define PSTAMP(i)                                                                          \
  do {                                                                                     \
    const unsigned long long t_ = __builtin_readcyclecounter();                            \
    if (PROF_ME) g_phys_prof[i] += t_ - p_last;                                            \
    p_last = t_;                                                                           \
  } while (0)
#define PSTART() unsigned long long p_last = __builtin_readcyclecounter()
__device__ unsigned long long g_wave_cycles[4096];      // duration of every wave of the last rollout launch (load balance)
#define PCOUNT(i, v) do { if (PROF_ME) g_phys_prof[i] += (v); } while (0)
#define PCOUNT_ALL(i, v) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_phys_prof[i], (unsigned long long)(v)); } while (0)   // every wave
#define RSTAMP(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); PCOUNT(i, t_ - r_last); r_last = t_; } while (0)
#define RSTART() unsigned long long r_last = __builtin_readcyclecounter()
#define KSTART() unsigned long long k_last = __builtin_readcyclecounter()
#define KSTAMP(i) do { const unsigned long long t_ = __builtin_readcyclecounter(); PCOUNT(i, t_ - k_last); k_last = t_; } while (0)
#elif defined(EARL_PHYS_MARK)
// ISA markers (compile with -DEARL_PHYS_MARK -S and count the instructions between them); not in the product build
#define PCOUNT(i, v) do {} while (0)
#define PSTAMP(i) asm volatile("; EARL_PHASE_END " #i ::: "memory")
#define PSTART() asm volatile("; EARL_PHASE_START" ::: "memory")
#define PCOUNT_ALL(i, v) do {} while (0)
#define RSTAMP(i) do {} while (0)
#define RSTART() do {} while (0)
#define KSTART() do {} while (0)
#define KSTAMP(i) asm volatile("; EARL_PHASE_END K" #i ::: "memory")
#else
#define PCOUNT(i, v) do {} while (0)
#define PSTAMP(i) do {} while (0)
#define PSTART() do {} while (0)
#define PCOUNT_ALL(i, v) do {} while (0)
#define RSTAMP(i) do {} while (0)
#define RSTART() do {} while (0)
#define KSTART() do {} while (0)
#define KSTAMP(i) do {} while (0)
#endif

#include "physics_math.h"
#include "physics_lds.h"
#include "physics_scan.h"
#include "physics_solve.h"

template <int LPE> __device__ __forceinline__ bool group_any(const bool pred, const int grp);

// One timestep of one env by its LPE-lane group (`sub` = lane within the group; every lane of the wave runs this, the
// groups working on their own Shared block).  INTEGRATE=false stops after qacc (mj_forward); outputs may be NULL.
// ROLE (round 5, the kitchen's one-env-per-workgroup launches): 0 = the whole timestep in one wave (every other launch).  1 - 4 = the timestep split over the FOUR waves of the
// workgroup, one per SIMD, that work on the same env, each in its own LDS block; all run the kinematics (K1 - K3).  Wave B (ROLE 2) owns the env: constraint rows (K8) before
// barrier X, then the contact rows (C3), the active-set iteration and the integration (K9, K10).  Before barrier X: wave A (ROLE 1) builds the mass matrix into B's block (K4, K5),
// ROLE 3 works out the bias forces (K6, K7) and hands B its lanes' generalized forces, ROLE 4 runs the bounding tests and the collision phases (C0 - C2) and leaves the contact
// records and their count in B's block.  Between X and Y wave A builds the equality Hessian in B's block; after Y, while B iterates on the active set, A factorises the arm's
// block of M + dt B and inverts the fixtures' scalars for K10 (barrier Z: B picks them up from A's block).  Same expressions, same inputs, same order: same bits.
template <int NV, int LPE, bool INTEGRATE, int ROLE = 0>
__device__ __forceinline__ void substep(Shared<NV>& s, const typename ModelOf<NV>::T& m, const BlkTable<Lim<NV>::MB, Lim<NV>::KBT>& bt, const earl_collision_model* __restrict__ col, const int sub,
                                        const int grp, const Q4 mq, const double (&ctrl)[EARL_MAXACT], const bool warm, double* qacc_out,
                                        double* efc_out, Shared<NV>* peer = nullptr) {
  // warm (uniform): s.aprev holds the solution of the previous timestep of the same env step / call, and the active-set iteration of K9 starts
  // from the set the new rows take AT it (MuJoCo warm-starts its solver from the previous qacc likewise) instead of from "every row active".
  // The fixed point is the same and so are the bits of the result (the last iteration builds the same Hessian from the same set); what changes is
  // the number of iterations: 1.81 -> 1.38 per timestep in contact for the door under random actions (oracle/physics_oracle.c g_newton_stats).
  static_assert(NV <= LPE, "one lane per link");
  // which parts of the timestep this instantiation runs (ROLE 5 / 6: the TWO-wave split of batches with two envs per CU -- 5 = mass matrix + bias forces + equality
  // Hessian + K10's factor, 6 = the owner incl. the collision phases)
  constexpr bool R_OWNER = ROLE == 2 || ROLE == 6;                     // constraint rows, contact rows, active set, integration
  constexpr bool R_COL = ROLE == 0 || ROLE == 4 || ROLE == 6;           // C0 - C2
  constexpr bool R_MASS = ROLE == 0 || ROLE == 1 || ROLE == 5;          // K4, K5
  constexpr bool R_BIAS = ROLE == 0 || ROLE == 3 || ROLE == 5;          // K6, K7
  constexpr bool R_HELPS_HW = ROLE == 1 || ROLE == 5;                   // builds the equality Hessian and K10's factor for the owner
  constexpr int MC = Lim<NV>::MC, NA = Lim<NV>::NA, NT = Lim<NV>::NT;
  static_assert(MC <= LPE, "one lane per contact");
  const int maxcon = bt.max_con < MC ? bt.max_con : MC;
  const double dt = m.dt;
  const bool isl = sub < NV;
  const int l = isl ? sub : NV - 1;
  const int ltri = l * (l + 1) / 2;                    // row offset of this lane in the packed symmetric matrices
  PSTART();
  // ------------------------------------------------------------------ K1: joint transform in the parent's frame
  Q4 Q; V3 P;
  {
    const Q4 tq = ldq(m.tquat[l]);
    const V3 ax = ld3(m.jaxis[l]), jp = ld3(m.jpos[l]);
    const int jt = m.jtype[l];
    const bool hinge = jt == 0;
    const double q = s.qp[l];
    double sn, cs;
    sincos_mod(hinge ? 0.5 * q : 0.0, sn, cs);
    double Rt[3][3], Rl[3][3];
    qmat(tq, Rt);
    // free body: link type 2 applies the orientation quaternion, the type-3 links behind it are rigid (sn = 0, cs = 1)
    const Q4 jq = selq(jt == 2, ldq(s.bq), Q4{cs, sn * ax.x, sn * ax.y, sn * ax.z});
    Q = qmul(tq, jq);
    qmat(Q, Rl);
    // hinge: rotate about the anchor; slide: translate along the axis (Rl == Rt then)
    P = add(add(ld3(m.tpos[l]), vsub(mulv(Rt, jp), mulv(Rl, jp))), scl(mulv(Rt, ax), jt == 1 ? q : 0.0));
  }
  // ------------------------------------------------------------------ K2: world frames by ancestor doubling
  if constexpr (Lim<NV>::ARMSCAN) {
    // inclusive prefix PRODUCT of the local transforms along the chains (X_l <- X_{l-k} o X_l, k = 1, 2, 4), in registers; then the fingers on the hand
    auto compose = [](const Q4& qa, const V3& pa, Q4& q, V3& p) {
      double Ra[3][3];
      qmat(qa, Ra);
      p = add(pa, mulv(Ra, p));
      q = qmul(qa, q);
    };
#define EARL_SCAN_ROUND(K) { const Q4 qs_ = dpp_row<DPP_SHR(K)>(Q); const V3 ps_ = dpp_row<DPP_SHR(K)>(P); Q4 qn_ = Q; V3 pn_ = P; compose(qs_, ps_, qn_, pn_); \
                             const bool on = scan_from_below<NV>(sub, K); Q = selq(on, qn_, Q); P = selv(on, pn_, P); }
    EARL_SCAN_ROUND(1) EARL_SCAN_ROUND(2) EARL_SCAN_ROUND(4)
#undef EARL_SCAN_ROUND
    {
      const Q4 q1 = dpp_row<DPP_SHR(1)>(Q), q2 = dpp_row<DPP_SHR(2)>(Q);
      const V3 p1 = dpp_row<DPP_SHR(1)>(P), p2 = dpp_row<DPP_SHR(2)>(P);
      Q4 qn_ = Q; V3 pn_ = P;
      compose(selq(sub == 7, q1, q2), selv(sub == 7, p1, p2), qn_, pn_);
      const bool on = sub == 7 || sub == 8;
      Q = selq(on, qn_, Q); P = selv(on, pn_, P);
    }
    if constexpr (Lim<NV>::EXTRAS) {
      // the phase's results held in registers HERE, whatever consumes them: a product that ends a phase is otherwise contracted into its consumer's add (fp contract fast)
      // or not depending on what else the instantiation does with it -- the waves of a split timestep (ROLE 1 - 4) must compute the bits of the one-wave form
      asm volatile("" : "+v"(Q.w), "+v"(Q.x), "+v"(Q.y), "+v"(Q.z), "+v"(P.x), "+v"(P.y), "+v"(P.z));
    }
    if (isl) {
      double* oq = s.Xq[l];
      double* op = s.Xp[l];
      oq[0] = Q.w; oq[1] = Q.x; oq[2] = Q.y; oq[3] = Q.z; op[0] = P.x; op[1] = P.y; op[2] = P.z;
    }
    fence();
  } else {
    const int rounds = m.n_jump;
    int buf = rounds & 1;                           // so that the last round lands in Xq / Xp
    if (isl) {
      double* oq = buf ? s.k2.Xq1[l] : s.Xq[l];
      double* op = buf ? s.k2.Xp1[l] : s.Xp[l];
      oq[0] = Q.w; oq[1] = Q.x; oq[2] = Q.y; oq[3] = Q.z; op[0] = P.x; op[1] = P.y; op[2] = P.z;
    }
    fence();
    for (int r = 0; r < rounds; ++r) {
      const int a = m.jump[r][l];
      const int ac = a < 0 ? 0 : a;
      const Q4 qa = ldq(buf ? s.k2.Xq1[ac] : s.Xq[ac]);
      const V3 xa = ld3(buf ? s.k2.Xp1[ac] : s.Xp[ac]);
      double Ra[3][3];
      qmat(qa, Ra);
      const V3 xn = add(xa, mulv(Ra, P));
      const Q4 qn = qmul(qa, Q);
      P = selv(a >= 0, xn, P); Q = selq(a >= 0, qn, Q);
      buf ^= 1;
      if (isl) {
        double* oq = buf ? s.k2.Xq1[l] : s.Xq[l];
        double* op = buf ? s.k2.Xp1[l] : s.Xp[l];
        oq[0] = Q.w; oq[1] = Q.x; oq[2] = Q.y; oq[3] = Q.z; op[0] = P.x; op[1] = P.y; op[2] = P.z;
      }
      fence();
    }
  }
  // ------------------------------------------------------------------ C0: collision bounding tests (world frames are final)
  using BlkMask = std::conditional_t<(Lim<NV>::MB > 32), unsigned long long, unsigned int>;
  BlkMask nearw = 0;                                   // blocks with a near bounding test in ANY env of the wave
  BlkMask nearg = 0;                                   // ... in this env
  for (int cb = 0; R_COL && cb < bt.n_blk; cb += LPE) {
    // C0: bounding test per block, lane = block (LPE blocks per pass)
    const int b = cb + sub < bt.n_blk ? cb + sub : 0;
    // (two batches of loads -- the block's table entries, then the frames of the two links they name -- each ONE LDS round trip: physics_math.h pin_batch.  Left to
    // the scheduler they were a dozen round trips one after the other)
    int bl = bt.link[b], xl = bt.box_link[b];
    constexpr bool SAT = BlkTable<Lim<NV>::MB, Lim<NV>::KBT>::SAT;
    double tb[14 + (SAT ? 6 : 0)];
#pragma unroll
    for (int k = 0; k < 3; ++k) { tb[k] = bt.center[b][k]; tb[3 + k] = bt.box_pos[b][k]; tb[10 + k] = bt.box_half[b][k]; }
#pragma unroll
    for (int k = 0; k < 4; ++k) tb[6 + k] = bt.box_quat[b][k];
    tb[13] = bt.reach[b];
    if constexpr (SAT) {
#pragma unroll
      for (int k = 0; k < 3; ++k) { tb[14 + k] = bt.obb_center[b][k]; tb[17 + k] = bt.obb_half[b][k]; }
    }
    asm volatile("" : "+v"(bl), "+v"(xl));
    pin_batch(tb);
    double fr[14];
    {
      const int blc = bl < 0 ? 0 : bl, xlc = xl < 0 ? 0 : xl;
#pragma unroll
      for (int k = 0; k < 4; ++k) { fr[k] = s.Xq[blc][k]; fr[7 + k] = s.Xq[xlc][k]; }
#pragma unroll
      for (int k = 0; k < 3; ++k) { fr[4 + k] = s.Xp[blc][k]; fr[11 + k] = s.Xp[xlc][k]; }
    }
    pin_batch(fr);
    V3 cs{tb[0], tb[1], tb[2]}, cb_{tb[3], tb[4], tb[5]}, ca{tb[SAT ? 14 : 0], tb[SAT ? 15 : 1], tb[SAT ? 16 : 2]};
    Q4 qb{tb[6], tb[7], tb[8], tb[9]};
    double RA[3][3];                                   // frame of the set's link (identity: world)
    {
      double R[3][3];
      qmat(Q4{fr[0], fr[1], fr[2], fr[3]}, R);
      const V3 xa{fr[4], fr[5], fr[6]};
      const V3 w = add(xa, mulv(R, cs));
      cs = selv(bl < 0, cs, w);
      ca = selv(bl < 0, ca, add(xa, mulv(R, ca)));
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) RA[i][j] = bl < 0 ? (i == j ? 1.0 : 0.0) : R[i][j];
      const Q4 ql{fr[7], fr[8], fr[9], fr[10]};
      qmat(ql, R);
      const V3 w2 = add(V3{fr[11], fr[12], fr[13]}, mulv(R, cb_));
      const Q4 q2 = qmul(ql, qb);
      cb_ = selv(xl < 0, cb_, w2);
      qb = selq(xl < 0, qb, q2);
    }
    // distance from the set's bounding-sphere centre to the box (in the box frame) against the set radius + margin
    double Rb[3][3];
    qmat(qb, Rb);
    const V3 x = mulvT(Rb, vsub(cs, cb_)), h{tb[10], tb[11], tb[12]};
    const V3 d{x.x - fmin(fmax(x.x, -h.x), h.x), x.y - fmin(fmax(x.y, -h.y), h.y), x.z - fmin(fmax(x.z, -h.z), h.z)};
    // second test: a face axis of the set's box (frame RA, centre ca, half extents incl. radii and margin) or of the block's box separates them
    bool separated = false;
    if constexpr (SAT) {
      const V3 t = mulvT(RA, vsub(cb_, ca)), ha{tb[SAT ? 17 : 0], tb[SAT ? 18 : 1], tb[SAT ? 19 : 2]};
      const double tt[3] = {t.x, t.y, t.z}, hA[3] = {ha.x, ha.y, ha.z}, hB[3] = {h.x, h.y, h.z};
      double Rm[3][3], aR[3][3];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          Rm[i][j] = RA[0][i] * Rb[0][j] + RA[1][i] * Rb[1][j] + RA[2][i] * Rb[2][j];
          aR[i][j] = fabs(Rm[i][j]);
        }
#pragma unroll
      for (int i = 0; i < 3; ++i) separated |= fabs(tt[i]) > hA[i] + (aR[i][0] * hB[0] + aR[i][1] * hB[1] + aR[i][2] * hB[2]);
#pragma unroll
      for (int j = 0; j < 3; ++j)
        separated |= fabs(tt[0] * Rm[0][j] + tt[1] * Rm[1][j] + tt[2] * Rm[2][j]) > hB[j] + (hA[0] * aR[0][j] + hA[1] * aR[1][j] + hA[2] * aR[2][j]);
    }
    const bool nearb = cb + sub < bt.n_blk && dot(d, d) < tb[13] * tb[13] && !separated;
    const unsigned long long bal = __ballot(nearb);
    if constexpr (LPE == 64) {
      nearg |= (BlkMask)bal; nearw |= (BlkMask)bal;
    } else {
      nearg |= (BlkMask)((bal >> (grp * (LPE & 63))) & ((1ull << (LPE & 63)) - 1ull)) << cb;
      if constexpr (LPE == 32) nearw |= (BlkMask)((bal | (bal >> 32)) & 0xFFFFFFFFull) << cb;
      else nearw |= (BlkMask)((bal | (bal >> 16) | (bal >> 32) | (bal >> 48)) & 0xFFFFull) << cb;
    }
  }
  // prefetch this lane's pair record of the first near block: its latency hides behind K3-K7
  int pf_blk = -1, pf_link = -1, pf_cls = 0;
  double pf_r = 0, pf_margin = 0, pf_hl = 0;
  V3 pf_pos{0, 0, 0}, pf_dir{0, 0, 0};
#ifndef EARL_NO_PREFETCH
#define EARL_NO_PREFETCH 0
#endif
  // Edge-vs-capsule blocks (the door's handle rods: 4 pairs each on 16 lanes per env) share a pass: a run of consecutive near capsule blocks is tested side by side, lane ->
  // (block, pair) as in the kitchen's packed C2 below.  cp_b / cp_off: this lane's block of the FIRST such pass and where its lanes begin (the run that starts at the first
  // near block, if that is a capsule block) -- worked out here so that the prefetch below fetches the record this lane will test.  (Round 6: the wave the door's launch waits
  // for has 5.3 near blocks per timestep, 3.3 of them capsule blocks: one pass instead of three.)
  int cp_b = -1, cp_off = 0;
  BlkMask cp_taken = 0;
  auto capsule_run = [&](const BlkMask from, int& myb, int& myoff) -> BlkMask {     // the leading run of capsule blocks of `from` that fits the group's lanes
    BlkMask taken = 0;
    int used = 0;
    myb = -1; myoff = 0;
    for (BlkMask r2 = from; r2; r2 &= r2 - 1u) {
      const int b = sizeof(BlkMask) == 8 ? __builtin_ctzll((unsigned long long)r2) : __builtin_ctz((unsigned int)r2);
      const int sz = bt.end[b] - bt.begin[b];
      if (!((bt.cap[b] >> 8) & 1) || used + sz > LPE) break;      // (a capsule block of more than LPE pairs is left to the block-per-pass loop)
      if (sub >= used && sub < used + sz) { myb = b; myoff = used; }
      used += sz;
      taken |= (BlkMask)1 << b;
    }
    return taken;
  };
  if constexpr (Lim<NV>::CAPS && !Lim<NV>::PACK) {
    if (nearw) cp_taken = capsule_run(nearw, cp_b, cp_off);
  }
  if (nearw && !(EARL_NO_PREFETCH && NV <= 10) && !Lim<NV>::PACK) {      // (two waves per SIMD hide that latency themselves; the registers are worth more there)
    pf_blk = sizeof(BlkMask) == 8 ? __builtin_ctzll((unsigned long long)nearw) : __builtin_ctz((unsigned int)nearw);
    const int pb_ = cp_taken ? (cp_b >= 0 ? cp_b : pf_blk) : pf_blk;
    const int pend = bt.end[pb_], pi0 = bt.begin[pb_] + sub - (cp_taken && cp_b >= 0 ? cp_off : 0);
    const int pi = pi0 < pend ? pi0 : pend - 1;
    pf_link = col->pair_rec[pi].sph_link; pf_cls = col->pair_rec[pi].cls;
    pf_r = col->pair_rec[pi].r; pf_margin = col->pair_rec[pi].margin;
    pf_pos = ld3(col->pair_rec[pi].pos);
    if constexpr (Lim<NV>::CAPS) { pf_dir = ld3(col->pair_rec[pi].dir); pf_hl = col->pair_rec[pi].hl; }
  }
  PSTAMP(0);
  // ------------------------------------------------------------------ K3: motion subspace + compact spatial inertia
  V3 Sw, Sv;                                         // this lane's column of S
  double I10r[10];                                   // this lane's link: compact spatial inertia about the world origin
  {
    double R[3][3];
    qmat(Q, R);
    const V3 aw = mulv(R, ld3(m.jaxis[l]));
    const V3 anchor = add(P, mulv(R, ld3(m.jpos[l])));
    const bool hinge = m.jtype[l] != 1;               // rotation axes of a free body: body axes after the rotation, like a hinge's
    Sw = selv(hinge, aw, V3{0, 0, 0});
    Sv = selv(hinge, cross(anchor, aw), aw);
    double ms = 1.0;                                 // the env's own mass / inertia factor of this link (minitaur: what the randomizer set at the last reset)
    if constexpr (Lim<NV>::CONNECT) {
      const int root = m.ball_dof + 2;
      ms = l < root ? 1.0 : (l == root ? s.xt.mscale[0] : (m.parent[l] == root ? s.xt.mscale[1] : s.xt.mscale[2]));
    }
    const double mass = m.mass[l] * ms;
    const V3 c = add(P, mulv(R, ld3(m.com[l])));
    const double* in = m.inertia[l];
    const double I[3][3] = {{in[0] * ms, in[3] * ms, in[4] * ms}, {in[3] * ms, in[1] * ms, in[5] * ms}, {in[4] * ms, in[5] * ms, in[2] * ms}};
    double T[3][3], W[3][3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cc = 0; cc < 3; ++cc) T[r][cc] = R[r][0] * I[0][cc] + R[r][1] * I[1][cc] + R[r][2] * I[2][cc];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
      for (int cc = r; cc < 3; ++cc) W[r][cc] = T[r][0] * R[cc][0] + T[r][1] * R[cc][1] + T[r][2] * R[cc][2];
    const double c2 = dot(c, c);
    I10r[0] = mass;
    I10r[1] = mass * c.x; I10r[2] = mass * c.y; I10r[3] = mass * c.z;
    I10r[4] = W[0][0] + mass * (c2 - c.x * c.x);
    I10r[5] = W[1][1] + mass * (c2 - c.y * c.y);
    I10r[6] = W[2][2] + mass * (c2 - c.z * c.z);
    I10r[7] = W[0][1] - mass * c.x * c.y;
    I10r[8] = W[0][2] - mass * c.x * c.z;
    I10r[9] = W[1][2] - mass * c.y * c.z;
    if constexpr (!Lim<NV>::ARMSCAN) {                   // (the scans below keep both in registers)
      if (isl) {
        double* o = s.dyn.S[l];
        o[0] = Sw.x; o[1] = Sw.y; o[2] = Sw.z; o[3] = Sv.x; o[4] = Sv.y; o[5] = Sv.z;
        double* i10 = s.dyn.I10[l];
#pragma unroll
        for (int e = 0; e < 10; ++e) i10[e] = I10r[e];
      }
    }
  }
  if constexpr (!Lim<NV>::ARMSCAN) fence();
  PSTAMP(1);
  const uint32_t amask = m.anc_mask[l], dmask = m.desc_mask[l];
  // the links this lane's masked sums visit: [tbase, tend), KT of them at most (all of [0, NT) unless the model has two multi-link trees)
  constexpr int TS = Lim<NV>::TS, KT = TS < NT ? (TS > NT - TS ? TS : NT - TS) : NT;
  const int tbase = (TS < NT && l >= TS) ? TS : 0, tend = (TS < NT && l < TS) ? TS : NT;
  double tau_l = 0.0;                                  // this lane's applied + passive - bias force (K7; ROLE 2: handed over by wave A)
  SymLds<NV>& Mw = R_HELPS_HW ? peer->M : s.M;         // where K5 puts the mass matrix
  if constexpr (R_MASS) {
  // ------------------------------------------------------------------ K4: composite inertia = masked subtree sum; FS = Ic S
  if constexpr (Lim<NV>::ARMSCAN) {
    double acc[10];
#pragma unroll
    for (int e = 0; e < 10; ++e) acc[e] = I10r[e];
    scan_desc<NV, 10>(acc, sub);                         // suffix sums along the chains, in registers
    V3 n, f;
    iapply(acc, Sw, Sv, n, f);
    if (isl) {
      double* o = s.dyn.crb.FS[l];
      o[0] = n.x; o[1] = n.y; o[2] = n.z; o[3] = f.x; o[4] = f.y; o[5] = f.z;
    }
  } else {
    double acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const int d = TS < NT ? (tbase + k < tend ? tbase + k : tend - 1) : k;
      const double w = ((TS == NT || tbase + k < tend) && ((dmask >> d) & 1u)) ? 1.0 : 0.0;
#pragma unroll
      for (int e = 0; e < 10; ++e) acc[e] = fma(w, s.dyn.I10[d][e], acc[e]);
    }
    if constexpr (NT < NV) {                             // a single-link tree: its composite inertia is its own
      const double w = l >= NT ? 1.0 : 0.0;
#pragma unroll
      for (int e = 0; e < 10; ++e) acc[e] = fma(w, s.dyn.I10[l][e], acc[e]);
    }
    V3 n, f;
    iapply(acc, Sw, Sv, n, f);
    if (isl) {
      double* o = s.dyn.crb.FS[l];
      o[0] = n.x; o[1] = n.y; o[2] = n.z; o[3] = f.x; o[4] = f.y; o[5] = f.z;
    }
  }
  fence();
  PSTAMP(3);
  // ------------------------------------------------------------------ K5: mass matrix, lane = column j
#ifndef EARL_PEG_K5_BATCH
#define EARL_PEG_K5_BATCH 3
#endif
  // Batched form (kitchen: all nine rows at once; peg: EARL_PEG_K5_BATCH rows at a time -- its kernel has no registers for more): the rows' FS first, the products
  // after, a select instead of a branch around the armature's load, and no branch around the store -- a lane without an entry in a row stores into the block's
  // padding.  With a conditional store per row the loop was one LDS round trip per row, one after the other (kitchen: 3.0 k -> 1.8 k cycles per timestep).
#ifndef EARL_DOOR_K5_BATCH
#define EARL_DOOR_K5_BATCH 5
#endif
  constexpr int K5B = Lim<NV>::EXTRAS ? KT : (NV == 15 ? EARL_PEG_K5_BATCH : (NV <= 10 ? EARL_DOOR_K5_BATCH : 0));
  if constexpr (K5B > 0) {
    const double arm_l = m.armature[l];
    double* const dump = reinterpret_cast<double*>(s.bank_pad);
#pragma unroll
    for (int k0 = 0; k0 < KT; k0 += (K5B > 0 ? K5B : 1)) {
      double fsr[K5B > 0 ? K5B : 1][6];
#pragma unroll
      for (int u = 0; u < K5B; ++u) {
        const int k = k0 + u < KT ? k0 + u : KT - 1;
        const bool in = TS == NT || tbase + k < tend;
        const int i = TS < NT ? (in ? tbase + k : tend - 1) : k;
#pragma unroll
        for (int e = 0; e < 6; ++e) fsr[u][e] = s.dyn.crb.FS[i][e];
      }
#pragma unroll
      for (int u = 0; u < K5B; ++u) {
        if (k0 + u < KT) {
          const int k = k0 + u;
          const bool in = TS == NT || tbase + k < tend;
          const int i = TS < NT ? (in ? tbase + k : tend - 1) : k;
          const double* fs = fsr[u];
          double v = Sw.x * fs[0] + Sw.y * fs[1] + Sw.z * fs[2] + Sv.x * fs[3] + Sv.y * fs[4] + Sv.z * fs[5];
          v = ((dmask >> i) & 1u) ? v : 0.0;            // j = l is an ancestor of (or is) i  <=>  i is in l's subtree
          v = i == l ? v + arm_l : v;
          if constexpr (SymLds<NV>::PACKED) {
            *((isl && l <= i && in) ? &Mw.v[i * (i + 1) / 2 + l] : dump) = v;
          } else {                                        // (square form: the entry and its mirror image)
            *((isl && l <= i && in) ? &Mw.v[i * NV + l] : dump) = v;
            *((isl && l <= i && in) ? &Mw.v[l * NV + i] : dump) = v;
          }
        }
      }
    }
  } else {
#pragma unroll
  for (int k = 0; k < KT; ++k) {
    const bool in = TS == NT || tbase + k < tend;
    const int i = TS < NT ? (in ? tbase + k : tend - 1) : k;
    const double* fs = s.dyn.crb.FS[i];
    double v = Sw.x * fs[0] + Sw.y * fs[1] + Sw.z * fs[2] + Sv.x * fs[3] + Sv.y * fs[4] + Sv.z * fs[5];
    v = ((dmask >> i) & 1u) ? v : 0.0;                  // j = l is an ancestor of (or is) i  <=>  i is in l's subtree
    if (i == l) v += m.armature[l];
    if (isl && l <= i && in) Mw.put(i, l, v, true);    // (the entries between the two trees were zeroed when the kernel started and are never written)
  }
  }
  if constexpr (NT < NV) {                               // single-link trees: a diagonal entry each; the entries that join them to anything else were
    if (isl && l >= NT) {                                // zeroed when the kernel started and are never written
      const double* fs = s.dyn.crb.FS[l];
      Mw.put(l, l, Sw.x * fs[0] + Sw.y * fs[1] + Sw.z * fs[2] + Sv.x * fs[3] + Sv.y * fs[4] + Sv.z * fs[5] + m.armature[l], true);
    }
  }
  fence();
  PSTAMP(4);
  }                                                    // (ROLE 0 / 1)
  if constexpr (R_BIAS) {
  // ------------------------------------------------------------------ K6: bias forces (RNE by masked sums)
  if constexpr (Lim<NV>::ARMSCAN) {
    // velocities V_l = sum over the ancestors of S_a qd_a, bias accelerations A_l = -g + sum of crossm(V) S_a qd_a, and the subtree sums of the bias forces:
    // prefix / suffix scans along the chains in registers (no LDS, no fence)
    const double qdl = s.qv[l];
    V3 w = scl(Sw, isl ? qdl : 0.0), v = scl(Sv, isl ? qdl : 0.0);
    scan_anc<NV>(w, v, sub);
    // d/dt of the axis uses the link's own velocity (the own term cancels); the three rotation axes of a free body use the velocity before any of them
    // (mj_comVel): that of its third slide = the translation velocity (world axes, checked by the host side)
    V3 wc = w, vc = v;
    if (m.ball_dof >= 0) {
      const int bd = m.ball_dof;
      const V3 vt{s.qv[bd - 3], s.qv[bd - 2], s.qv[bd - 1]};
      const bool rot = l >= bd && l < bd + 3;
      wc = selv(rot, V3{0, 0, 0}, wc);
      vc = selv(rot, vt, vc);
    }
    V3 cw = scl(cross(wc, Sw), isl ? qdl : 0.0), cv = scl(add(cross(vc, Sw), cross(wc, Sv)), isl ? qdl : 0.0);
    scan_anc<NV>(cw, cv, sub);
    const V3 aw = cw, av = add(cv, V3{-m.gravity[0], -m.gravity[1], -m.gravity[2]});
    V3 n1, f1, n2, f2;
    iapply(I10r, aw, av, n1, f1);
    iapply(I10r, w, v, n2, f2);
    const V3 n = add(n1, add(cross(w, n2), cross(v, f2)));
    const V3 f = add(f1, cross(w, f2));
    double nf[6] = {n.x, n.y, n.z, f.x, f.y, f.z};
    scan_desc<NV, 6>(nf, sub);
    double t = -m.damping[l] * qdl - (Sw.x * nf[0] + Sw.y * nf[1] + Sw.z * nf[2] + Sv.x * nf[3] + Sv.y * nf[4] + Sv.z * nf[5]);
    if constexpr (Lim<NV>::EXTRAS) t -= m.stiffness[l] * (s.qp[l] - m.springref[l]);
    {
      // the actuators' tables as one batch of loads (physics_math.h pin_batch), their forces added under a select: the loop with a branch per actuator was a chain of
      // LDS round trips (its joint, then its ranges and gain)
      int aj[EARL_MAXACT];
      double at[(Lim<NV>::EXTRAS ? 5 : 3) * EARL_MAXACT];
      constexpr int AS = Lim<NV>::EXTRAS ? 5 : 3;
#pragma unroll
      for (int ac = 0; ac < EARL_MAXACT; ++ac) {
        aj[ac] = m.act_joint[ac];
        at[AS * ac] = m.act_ctrlrange[ac][0]; at[AS * ac + 1] = m.act_ctrlrange[ac][1]; at[AS * ac + 2] = m.act_kp[ac];
        if constexpr (Lim<NV>::EXTRAS) { at[AS * ac + 3] = m.act_forcerange[ac][0]; at[AS * ac + 4] = m.act_forcerange[ac][1]; }      // (forcelimited actuators: the 24-dof model form)
      }
      static_assert(EARL_MAXACT == 4, "four actuator slots");
      asm volatile("" : "+v"(aj[0]), "+v"(aj[1]), "+v"(aj[2]), "+v"(aj[3]));
      pin_batch(at);
      const double qpl = s.qp[l];
#pragma unroll
      for (int ac = 0; ac < EARL_MAXACT; ++ac) {
        const double c = fmin(fmax(ctrl[ac], at[AS * ac]), at[AS * ac + 1]);
        double frc = at[AS * ac + 2] * (c - qpl);
        if constexpr (Lim<NV>::EXTRAS) frc = fmin(fmax(frc, at[AS * ac + 3]), at[AS * ac + 4]);
        t = (ac < m.n_act && aj[ac] == l) ? t + frc : t;
      }
    }
    tau_l = t;
  } else {
    V3 w{0, 0, 0}, v{0, 0, 0};
#ifndef EARL_K6_BATCH
#define EARL_K6_BATCH 5
#endif
    if constexpr (NV <= 10 && EARL_K6_BATCH > 0) {
      // (the door build: the ancestors' subspaces and velocities in batches of loads -- physics_math.h pin_batch; the velocity under a select was a branch around its
      // load per row, each with a wait of its own)
      constexpr int KB = EARL_K6_BATCH > 0 ? EARL_K6_BATCH : 1;
#pragma unroll
      for (int k0 = 0; k0 < KT; k0 += KB) {
        double sv[6 * KB], qv_[KB];
#pragma unroll
        for (int u = 0; u < KB; ++u) {
          const int a = k0 + u < KT ? k0 + u : KT - 1;
          qv_[u] = s.qv[a];
#pragma unroll
          for (int e = 0; e < 6; ++e) sv[6 * u + e] = s.dyn.S[a][e];
        }
        pin_batch(sv); pin_batch(qv_);
#pragma unroll
        for (int u = 0; u < KB; ++u) {
          if (k0 + u < KT) {
            const int a = k0 + u;
            const double qd = ((amask >> a) & 1u) ? qv_[u] : 0.0;
            w = add(w, scl(V3{sv[6 * u], sv[6 * u + 1], sv[6 * u + 2]}, qd));
            v = add(v, scl(V3{sv[6 * u + 3], sv[6 * u + 4], sv[6 * u + 5]}, qd));
          }
        }
      }
    } else {
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const bool in = TS == NT || tbase + k < tend;
      const int a = TS < NT ? (in ? tbase + k : tend - 1) : k;
      const double qd = (in && ((amask >> a) & 1u)) ? s.qv[a] : 0.0;
      const double* sa = s.dyn.S[a];
      w = add(w, scl(ld3(sa), qd));
      v = add(v, scl(ld3(sa + 3), qd));
    }
    }
    if constexpr (NT < NV) {                             // single-link tree: only its own joint moves it
      const double qd = l >= NT ? s.qv[l] : 0.0;
      w = add(w, scl(Sw, qd));
      v = add(v, scl(Sv, qd));
    }
    // crossm(V_l) S_l qd_l = [w x sw ; v x sw + w x sv] qd   (V of the parent and V_l differ by S_l qd_l, whose cross with S_l is 0)
    const double qdl = s.qv[l];
    // d/dt of this link's axis uses the velocity of the links in cd_mask: all ancestors, except that the three rotation
    // axes of a free body use the velocity before any of them (mj_comVel computes the three dofdots before updating cvel)
    V3 wc = w, vc = v;
    if (m.ball_dof >= 0) {
      const uint32_t drop = amask & ~m.cd_mask[l];
      for (int a = m.ball_dof; a < m.ball_dof + 3; ++a) {      // (the three rotation links of the free body)
        const double qd = ((drop >> a) & 1u) ? s.qv[a] : 0.0;
        const double* sa = s.dyn.S[a];
        wc = vsub(wc, scl(ld3(sa), qd));
        vc = vsub(vc, scl(ld3(sa + 3), qd));
      }
    }
    const V3 cw = scl(cross(wc, Sw), qdl), cv = scl(add(cross(vc, Sw), cross(wc, Sv)), qdl);
    if (isl) {
      double* o = s.dyn.rne.Cc[l];
      o[0] = cw.x; o[1] = cw.y; o[2] = cw.z; o[3] = cv.x; o[4] = cv.y; o[5] = cv.z;
    }
    fence();
    V3 aw{0, 0, 0}, av{-m.gravity[0], -m.gravity[1], -m.gravity[2]};
    if constexpr (NV <= 10 && EARL_K6_BATCH > 0) {
      constexpr int KB = EARL_K6_BATCH > 0 ? EARL_K6_BATCH : 1;
#pragma unroll
      for (int k0 = 0; k0 < KT; k0 += KB) {
        double cv_[6 * KB];
#pragma unroll
        for (int u = 0; u < KB; ++u)
#pragma unroll
          for (int e = 0; e < 6; ++e) cv_[6 * u + e] = s.dyn.rne.Cc[k0 + u < KT ? k0 + u : KT - 1][e];
        pin_batch(cv_);
#pragma unroll
        for (int u = 0; u < KB; ++u) {
          if (k0 + u < KT) {
            const double wgt = ((amask >> (k0 + u)) & 1u) ? 1.0 : 0.0;
            aw = add(aw, scl(V3{cv_[6 * u], cv_[6 * u + 1], cv_[6 * u + 2]}, wgt));
            av = add(av, scl(V3{cv_[6 * u + 3], cv_[6 * u + 4], cv_[6 * u + 5]}, wgt));
          }
        }
      }
    } else {
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const bool in = TS == NT || tbase + k < tend;
      const int a = TS < NT ? (in ? tbase + k : tend - 1) : k;
      const double wgt = (in && ((amask >> a) & 1u)) ? 1.0 : 0.0;
      const double* ca = s.dyn.rne.Cc[a];
      aw = add(aw, scl(ld3(ca), wgt));
      av = add(av, scl(ld3(ca + 3), wgt));
    }
    }
    if constexpr (NT < NV) {
      const double wgt = l >= NT ? 1.0 : 0.0;
      aw = add(aw, scl(cw, wgt));
      av = add(av, scl(cv, wgt));
    }
    V3 n1, f1, n2, f2;
    iapply(s.dyn.I10[l], aw, av, n1, f1);
    iapply(s.dyn.I10[l], w, v, n2, f2);
    const V3 n = add(n1, add(cross(w, n2), cross(v, f2)));                 // crossf(V) [n; f] = [w x n + v x f ; w x f]
    const V3 f = add(f1, cross(w, f2));
    if (isl) {
      double* o = s.dyn.rne.F[l];
      o[0] = n.x; o[1] = n.y; o[2] = n.z; o[3] = f.x; o[4] = f.y; o[5] = f.z;
    }
    fence();
    V3 ns{0, 0, 0}, fs{0, 0, 0};
    if constexpr (NV <= 10 && EARL_K6_BATCH > 0) {
      constexpr int KB = EARL_K6_BATCH > 0 ? EARL_K6_BATCH : 1;
#pragma unroll
      for (int k0 = 0; k0 < KT; k0 += KB) {
        double fv_[6 * KB];
#pragma unroll
        for (int u = 0; u < KB; ++u)
#pragma unroll
          for (int e = 0; e < 6; ++e) fv_[6 * u + e] = s.dyn.rne.F[k0 + u < KT ? k0 + u : KT - 1][e];
        pin_batch(fv_);
#pragma unroll
        for (int u = 0; u < KB; ++u) {
          if (k0 + u < KT) {
            const double wgt = ((dmask >> (k0 + u)) & 1u) ? 1.0 : 0.0;
            ns = add(ns, scl(V3{fv_[6 * u], fv_[6 * u + 1], fv_[6 * u + 2]}, wgt));
            fs = add(fs, scl(V3{fv_[6 * u + 3], fv_[6 * u + 4], fv_[6 * u + 5]}, wgt));
          }
        }
      }
    } else {
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      const bool in = TS == NT || tbase + k < tend;
      const int d = TS < NT ? (in ? tbase + k : tend - 1) : k;
      const double wgt = (in && ((dmask >> d) & 1u)) ? 1.0 : 0.0;
      const double* fd = s.dyn.rne.F[d];
      ns = add(ns, scl(ld3(fd), wgt));
      fs = add(fs, scl(ld3(fd + 3), wgt));
    }
    }
    if constexpr (NT < NV) {
      const double wgt = l >= NT ? 1.0 : 0.0;
      ns = add(ns, scl(n, wgt));
      fs = add(fs, scl(f, wgt));
    }
    // ---------------------------------------------------------------- K7: applied + passive - bias
    double t = -m.damping[l] * qdl - (dot(Sw, ns) + dot(Sv, fs));
    if constexpr (Lim<NV>::EXTRAS) t -= m.stiffness[l] * (s.qp[l] - m.springref[l]);     // joint spring (mj_passive)
    for (int ac = 0; ac < m.n_act; ++ac)
      if (m.act_joint[ac] == l) {
        const double c = fmin(fmax(ctrl[ac], m.act_ctrlrange[ac][0]), m.act_ctrlrange[ac][1]);
        double frc = m.act_kp[ac] * (c - s.qp[l]);
        if constexpr (Lim<NV>::EXTRAS) frc = fmin(fmax(frc, m.act_forcerange[ac][0]), m.act_forcerange[ac][1]);   // forcelimited actuator
        t += frc;
      }
    if constexpr (Lim<NV>::CONNECT) t += s.xt.ext[l];      // generalized force handed in for this timestep (the minitaur's motor torques)
    tau_l = t;
  }
  }                                                    // (ROLE 0 / 3)
  fence();                                             // dyn.* is dead from here on; col.* then con.* take its place
  PSTAMP(5);
  // The structured models' equality Hessian (K9) as a function: wave A of a split timestep builds it in wave B's block.
  // hw_extras: the equality part has the model's structure (checked by the host side): the arm's NT x NT block (mass matrix + weld rows), one diagonal entry per
  // fixture, one off-diagonal entry per coupled pair of fixtures.  Only those entries of Hw are ever written; the others were zeroed when the
  // kernel started.  (The earlier form built all 23 rows of every column in registers and ran every coupling over all of them with selects: 14 k of
  // the timestep's 62 k cycles.)  Same values, same order of additions per entry.  `o`: the block that holds the mass matrix, the weld rows and the couplings' records and takes Hw.
  // Loads first, stores after, no branch in between: with a conditional store per row the loop was nine LDS round trips one after the other.
  auto hw_extras = [&](Shared<NV>& o, const double (&DJ)[6]) {
    if constexpr (Lim<NV>::EXTRAS) {
      double h[NT];
#pragma unroll
      for (int i = 0; i < NT; ++i) {
        h[i] = o.M.sym(i, l, ltri);
#pragma unroll
        for (int r = 0; r < 6; ++r) h[i] = fma(o.con.J6[r][i], DJ[r], h[i]);
        if (i == l) h[i] += m.drag_G[l];
      }
      // fixture lanes: the diagonal entry, the entry shared with the coupled partner (a coupling q[j1] - c0 - c1 q[j2] = 0 is a soft equality row with two non-zeros,
      // 1 at j1 and -c1 at j2: rows j1 and j2 of column l get D J_l and -c1 D J_l -- for lane j1 that is its diagonal and its partner's row, for lane j2 the other way round)
      const int pl = m.pair[l];
      const double* const rec = o.jeq.rec[l >= NT ? l - NT : 0];
      const double hd = o.M.sym(l, l, ltri) + m.drag_G[l] + (pl >= 0 ? rec[2] : 0.0), ho = pl >= 0 ? rec[3] : 0.0;
      double* const dump = reinterpret_cast<double*>(s.bank_pad);      // (a lane without an entry stores into its own block's padding: no branch per row)
#pragma unroll
      for (int i = 0; i < NT; ++i) *((isl && l < NT && i >= l) ? &o.hwst.Hw.lo(i, l) : dump) = h[i];
      *((isl && l >= NT) ? &o.hwst.Hw.lo(l, l) : dump) = hd;
      *((isl && l >= NT && pl > l) ? &o.hwst.Hw.lo(pl > l ? pl : l, l) : dump) = ho;      // (the lower triangle: the lane with the smaller index of a pair stores the shared entry)
    }
  };
  if constexpr (R_HELPS_HW) {
    // wave A: the mass matrix went straight to the peer's block; when wave B's weld rows are there (barrier X) build the equality Hessian from both, in the peer's block; leave
    static_assert(!R_HELPS_HW || (Lim<NV>::EXTRAS && Lim<NV>::ARMSCAN), "the split timestep is the kitchen model's");
    if constexpr (ROLE == 5) { if (isl) peer->tau[l] = tau_l; }      // (two-wave split: the bias forces are this wave's too)
    __syncthreads();                                   // barrier X
    PSTAMP(10);
    double DJ[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) DJ[r] = peer->con.wD[r] * peer->con.J6[r][l];      // (J6[r][l] is lane l's Jc[r])
    hw_extras(*peer, DJ);
    fence();
    PSTAMP(9);
    __syncthreads();                                   // barrier Y
    PSTAMP(12);
    // ... and, while wave B iterates on the active set, K10's factorisation -- the arm's block of M + dt B and the fixtures' reciprocals depend on nothing B still has to work
    // out -- left in THIS wave's block (its equality-Hessian store is otherwise unused; wave B holds a pointer to it).  The same chol_regs on the same entries: the same factor.
    {
      constexpr int NL = NA * (NA + 1) / 2;
      static_assert(NL + (NV - NA) <= (int)(sizeof(s.hwst.Hw.v) / sizeof(double)), "factor and reciprocals fit the store");
      double Lk[NL];
#pragma unroll
      for (int i = 0; i < NA; ++i) {
#pragma unroll
        for (int j = 0; j < i; ++j) Lk[i * (i + 1) / 2 + j] = peer->M.lo(i, j);
        Lk[i * (i + 1) / 2 + i] = peer->M.lo(i, i) + pinned(dt * m.damping[i]);      // (the rounded product K10 sends through LDS)
      }
      pin_batch(Lk);
      chol_regs<NA, NA, true>(Lk);
      double* const dump = reinterpret_cast<double*>(s.bank_pad);
#pragma unroll
      for (int e = 0; e < NL; ++e) *(sub == (e % LPE) ? &s.hwst.Hw.v[e] : dump) = Lk[e];
      const int lf = l >= NA ? l : NA;
      *((isl && l >= NA) ? &s.hwst.Hw.v[NL + lf - NA] : dump) = rcp_nr(peer->M.lo(lf, lf) + pinned(dt * m.damping[lf]));
      fence();
    }
    __syncthreads();                                   // barrier Z
    return;
  }
  auto jeq_records = [&](Shared<NV>& o) {      // (`o`: the block that takes the records -- wave B's when the collision wave of a split timestep works them out)
  if constexpr (Lim<NV>::EXTRAS) {
    // joint couplings q[j1] - c0 - c1 q[j2] = 0: soft equality rows with two non-zeros (1 at j1, -c1 at j2).  Lane e works out coupling e's regulariser and reference
    // acceleration (a chain of dependent operations incl. three reciprocals) and leaves, for each of its two dofs, what that dof's lane adds in K9: D J_l, aref, the
    // term of its diagonal entry, the term of the entry it shares with its partner.  (Until round 5 every lane walked all couplings in K9, twice five LDS round trips.)
    static_assert(NT < NV || !Lim<NV>::EXTRAS, "coupled dofs lie behind the first tree (checked by the host side)");
    {
      // (no branch: every lane runs the chain -- for coupling 0 beyond the count -- and only lanes e < n_jeq store: the scheduler runs it beside the weld rows' chain above)
      const bool mine = sub < m.n_jeq;
      const int e = mine ? sub : 0, j1 = m.jeq_joint1[e], j2 = m.jeq_joint2[e];
      const double c0 = m.jeq_coef[e][0], c1 = m.jeq_coef[e][1];
      const double res = s.qp[j1] - c0 - c1 * s.qp[j2], Jv = s.qv[j1] - c1 * s.qv[j2];
      const double kk = bt.kb_jeq[e][0], bb = bt.kb_jeq[e][1], dd = imp_p2(m.jeq_solimp[e], res);
      const double D = rcp_nr(fmax((1 - dd) * m.jeq_invweight[e] * rcp_nr(dd), 1e-15));
      const double ar = -bb * Jv - kk * dd * res;
      const double DJ1 = D * 1.0, DJ2 = D * -c1;
      double* const dump = reinterpret_cast<double*>(s.bank_pad);
      double* const r1 = o.jeq.rec[j1 >= NT ? j1 - NT : 0];
      double* const r2 = o.jeq.rec[j2 >= NT ? j2 - NT : 0];
      *(mine ? &r1[0] : dump) = DJ1; *(mine ? &r1[1] : dump) = ar; *(mine ? &r1[2] : dump) = DJ1; *(mine ? &r1[3] : dump) = -c1 * DJ1;
      *(mine ? &r2[0] : dump) = DJ2; *(mine ? &r2[1] : dump) = ar; *(mine ? &r2[2] : dump) = -c1 * DJ2; *(mine ? &r2[3] : dump) = DJ2;
    }
  }
  };
  if constexpr (ROLE == 3) {                           // the bias-force wave: hand the generalized forces over and leave
    if (isl) peer->tau[l] = tau_l;
    __syncthreads();                                   // barrier X
    PSTAMP(10);
    __syncthreads();                                   // barrier Y
    PSTAMP(12);
    __syncthreads();                                   // barrier Z
    return;
  }
  // ------------------------------------------------------------------ C1-C2: collision (reference: LinkModel.collide)
  int nct = 0;                                         // contacts of this env (same value in every lane of the group)
  double (*const ctw)[8] = ROLE == 4 ? peer->con.ct : s.con.ct;      // where the contact records go (the collision wave of a split timestep: into wave B's block)
  if constexpr (R_COL) {
  if constexpr (Lim<NV>::PACK) {
    // C2, PACKED (the kitchen: blocks of 2 - 10 pairs on 32 lanes per env): consecutive near blocks of the wave share a pass as long as their pairs fit the
    // group's LPE lanes -- lane -> (block, pair) by a walk over the pass's blocks, the block's box frame per lane.  Contacts keep the sequential order (blocks
    // ascending on the lanes, pairs ascending within a block) and the per-block caps, so the contact list is the one the block-per-pass loop below builds; the
    // wave whose hand is among the fixtures -- the one the launch waits for -- walked six blocks per timestep one after the other: 8.6 k cycles.
    BlkMask rest = nearw;
    while (rest) {
      // this pass: blocks from `rest` while their sizes fit
      int myb = -1, myoff = 0, used = 0;
      BlkMask taken = 0;
      for (BlkMask r2 = rest; r2; r2 &= r2 - 1u) {
        const int b = sizeof(BlkMask) == 8 ? __builtin_ctzll((unsigned long long)r2) : __builtin_ctz((unsigned int)r2);
        const int sz = bt.end[b] - bt.begin[b];
        if (used + sz > LPE) { if (used == 0) { taken = (BlkMask)1 << b; used = sz; } break; }      // (a block larger than LPE would go alone; the host side refuses such tables)
        if (sub >= used && sub < used + sz) { myb = b; myoff = used; }
        used += sz;
        taken |= (BlkMask)1 << b;
      }
      rest &= ~taken;
      const bool has = myb >= 0;
      const int b = has ? myb : 0;
      const bool mine = has && ((nearg >> b) & 1u);
      const int xl = bt.box_link[b];
      V3 pb = ld3(bt.box_pos[b]);
      Q4 qb = ldq(bt.box_quat[b]);
      {
        const Q4 ql = ldq(s.Xq[xl < 0 ? 0 : xl]);
        double R[3][3];
        qmat(ql, R);
        pb = selv(xl < 0, pb, add(ld3(s.Xp[xl < 0 ? 0 : xl]), mulv(R, pb)));
        qb = selq(xl < 0, qb, qmul(ql, qb));
      }
      double Rb[3][3];
      qmat(qb, Rb);
      const V3 h = ld3(bt.box_half[b]);
      const int room = bt.cap[b] & 255;
      const int pi = has ? bt.begin[b] + (sub - myoff) : 0;
      const int lk = col->pair_rec[pi].sph_link, cls = col->pair_rec[pi].cls;
      const double r = col->pair_rec[pi].r, margin = col->pair_rec[pi].margin;
      V3 c = ld3(col->pair_rec[pi].pos);
      {
        double R[3][3];
        qmat(ldq(s.Xq[lk < 0 ? 0 : lk]), R);
        c = selv(lk < 0, c, add(ld3(s.Xp[lk < 0 ? 0 : lk]), mulv(R, c)));
      }
      const V3 x = mulvT(Rb, vsub(c, pb));
      V3 q{fmin(fmax(x.x, -h.x), h.x), fmin(fmax(x.y, -h.y), h.y), fmin(fmax(x.z, -h.z), h.z)};
      const bool outside = fabs(x.x) > h.x || fabs(x.y) > h.y || fabs(x.z) > h.z;
      const V3 d = vsub(x, q);
      const double d2 = dot(d, d);
      const double inv = rsq_nr(outside ? d2 : 1.0);
      const double gx = h.x - fabs(x.x), gy = h.y - fabs(x.y), gz = h.z - fabs(x.z);
      const int ax = (gx <= gy && gx <= gz) ? 0 : (gy <= gz ? 1 : 2);
      const double xa = pick3(x, ax), ha = pick3(h, ax), sg = xa >= 0 ? 1.0 : -1.0;
      const V3 ni{ax == 0 ? sg : 0.0, ax == 1 ? sg : 0.0, ax == 2 ? sg : 0.0};
      const V3 qi{ax == 0 ? sg * ha : x.x, ax == 1 ? sg * ha : x.y, ax == 2 ? sg * ha : x.z};
      const double dist = outside ? d2 * inv - r : -(ha - fabs(xa)) - r;
      const V3 nl = selv(outside, scl(d, inv), ni);
      q = selv(outside, q, qi);
      const bool hit = mine && dist < margin;
      const unsigned long long bal = __ballot(hit);
      const unsigned int gb = (unsigned int)((bal >> (grp * (LPE & 63))) & ((1ull << (LPE & 63)) - 1ull));
      const unsigned int seg = has ? (unsigned int)((((1ull << (bt.end[b] - bt.begin[b])) - 1ull)) << myoff) : 0u;      // the lanes of this lane's block
      const int before_blk = __popc(gb & seg & ((1u << sub) - 1u));
      const bool accept = hit && before_blk < room;
      const unsigned long long bal2 = __ballot(accept);
      const unsigned int ga = (unsigned int)((bal2 >> (grp * (LPE & 63))) & ((1ull << (LPE & 63)) - 1ull));
      const int slot = nct + __popc(ga & ((1u << sub) - 1u));
      if (accept && slot < maxcon) {
        const V3 n = mulv(Rb, nl);
        const V3 p = add(add(pb, mulv(Rb, q)), scl(n, 0.5 * dist));
        double* o = ctw[slot];
        o[0] = dist; o[1] = n.x; o[2] = n.y; o[3] = n.z; o[4] = p.x; o[5] = p.y; o[6] = p.z;
        o[7] = (double)(cls + 64 * (lk + 1) + 4096 * (xl + 1));
      }
      const int took = __popc(ga);
      nct = nct + took < maxcon ? nct + took : maxcon;
    }
    if (nearw) fence();
  } else
  if (nearw) {
    // C2: pair tests of the near blocks, in pair order; the box frame once per block, the sphere centre per test
    BlkMask rest = nearw;
    while (rest) {
      if constexpr (Lim<NV>::CAPS) {
        // a run of capsule blocks at the head of `rest`: ONE pass, lane -> (block, pair); contacts keep the sequential order and the per-block caps
        int myb, myoff;
        const bool first = rest == nearw;
        const BlkMask taken = first ? cp_taken : capsule_run(rest, myb, myoff);
        if (first) { myb = cp_b; myoff = cp_off; }
        if (taken) {
          const bool has = myb >= 0;
          const int b = has ? myb : (sizeof(BlkMask) == 8 ? __builtin_ctzll((unsigned long long)taken) : __builtin_ctz((unsigned int)taken));
          const bool mine = has && ((nearg >> b) & 1u);
          const int bsz = bt.end[b] - bt.begin[b], xl = bt.box_link[b];
          V3 pb = ld3(bt.box_pos[b]);
          Q4 qb = ldq(bt.box_quat[b]);
          {
            const Q4 ql = ldq(s.Xq[xl < 0 ? 0 : xl]);
            double R[3][3];
            qmat(ql, R);
            pb = selv(xl < 0, pb, add(ld3(s.Xp[xl < 0 ? 0 : xl]), mulv(R, pb)));
            qb = selq(xl < 0, qb, qmul(ql, qb));
          }
          double Rb[3][3];
          qmat(qb, Rb);
          const V3 h = ld3(bt.box_half[b]);
          const int room = bt.cap[b] & 255;
          int lk, cls;
          double r, margin, hl;
          V3 c, ed;
          if (first && pf_blk >= 0) {                     // uniform: the records prefetched after C0 (not in the eight-wave build: EARL_NO_PREFETCH)
            lk = pf_link; cls = pf_cls; r = pf_r; margin = pf_margin; c = pf_pos; ed = pf_dir; hl = pf_hl;
          } else {
            const int pi = has ? bt.begin[b] + (sub - myoff) : bt.begin[b];
            lk = col->pair_rec[pi].sph_link; cls = col->pair_rec[pi].cls;
            r = col->pair_rec[pi].r; margin = col->pair_rec[pi].margin;
            c = ld3(col->pair_rec[pi].pos); ed = ld3(col->pair_rec[pi].dir); hl = col->pair_rec[pi].hl;
          }
          (void)r;
          {
            double R[3][3];
            qmat(ldq(s.Xq[lk < 0 ? 0 : lk]), R);
            const V3 w = add(ld3(s.Xp[lk < 0 ? 0 : lk]), mulv(R, c));
            c = selv(lk < 0, c, w);
            ed = selv(lk < 0, ed, mulv(R, ed));
          }
          // (the pass's inputs held in registers HERE and its results below, whatever else the instantiation does around them: under fp contract(fast) the door's two builds
          // -- four and eight waves per workgroup, the latter for batches beyond 4096 envs -- otherwise fused these sums differently, and a shard of 4096 envs no longer
          // returned the bits of the same envs in a batch of 8192: tests/test_sawyer_full_gpu.py)
          pin6(c.x, c.y, c.z, ed.x, ed.y, ed.z); pin6(pb.x, pb.y, pb.z, Rb[0][0], Rb[0][1], Rb[0][2]); pin6(Rb[1][0], Rb[1][1], Rb[1][2], Rb[2][0], Rb[2][1], Rb[2][2]);
          // closest points of the edge (c +- hl ed) and the capsule's axis segment (pb +- hc cd); normal from the axis to the edge
          const V3 cd{Rb[0][2], Rb[1][2], Rb[2][2]}, rr = vsub(c, pb);
          const double hc = h.z - h.x, rad = h.x;
          const double b_ = dot(ed, cd), c_ = dot(ed, rr), f_ = dot(cd, rr), den = 1.0 - b_ * b_;
          double s_ = den > 1e-12 ? fmin(fmax((b_ * f_ - c_) / den, -hl), hl) : 0.0;
          const double t_ = fmin(fmax(fma(b_, s_, f_), -hc), hc);
          s_ = fmin(fmax(fma(b_, t_, -c_), -hl), hl);
          const V3 d = vsub(add(rr, scl(ed, s_)), scl(cd, t_));
          const double d2 = dot(d, d);
          const bool sane = d2 > 1e-18;
          const double inv = rsq_nr(sane ? d2 : 1.0);
          double dist = d2 * inv - rad;
          const V3 nw = scl(d, inv);
          V3 nl = mulvT(Rb, nw);
          V3 q{nl.x * rad, nl.y * rad, t_ + nl.z * rad};   // surface point of the capsule in its own frame (axis = z)
          dist = pinned(dist); pin6(nl.x, nl.y, nl.z, q.x, q.y, q.z);
          const bool hit = mine && sub - myoff < bsz && sane && dist < margin;
          auto of_group = [&](const unsigned long long bits) { return LPE == 64 ? bits : ((bits >> (grp * (LPE & 63))) & ((1ull << (LPE & 63)) - 1ull)); };
          const unsigned long long below = (1ull << sub) - 1ull;
          const unsigned long long gb = of_group(__ballot(hit));
          const unsigned long long seg = has ? (bsz >= 64 ? ~0ull : ((1ull << bsz) - 1ull) << myoff) : 0ull;      // the lanes of this lane's block
          const int before_blk = __popcll(gb & seg & below);
          const bool accept = hit && before_blk < room;
          const unsigned long long ga = of_group(__ballot(accept));
          const int slot = nct + __popcll(ga & below);
          {
            V3 n = mulv(Rb, nl);
            V3 p = add(add(pb, mulv(Rb, q)), scl(n, 0.5 * dist));
            pin6(n.x, n.y, n.z, p.x, p.y, p.z);
            if (accept && slot < maxcon) {
              double* o = ctw[slot];
              o[0] = dist; o[1] = n.x; o[2] = n.y; o[3] = n.z; o[4] = p.x; o[5] = p.y; o[6] = p.z;
              o[7] = (double)(cls + 64 * (lk + 1) + 4096 * (xl + 1));
            }
          }
          const int took = __popcll(ga);
          nct = nct + took < maxcon ? nct + took : maxcon;
          rest &= ~taken;
          continue;
        }
      }
      const int b = sizeof(BlkMask) == 8 ? __builtin_ctzll((unsigned long long)rest) : __builtin_ctz((unsigned int)rest);
      rest &= rest - 1u;
      const bool mine = (nearg >> b) & 1u;
      const int pend = bt.end[b], xl = bt.box_link[b];
      V3 pb = ld3(bt.box_pos[b]);
      Q4 qb = ldq(bt.box_quat[b]);
      if (xl >= 0) {                                    // uniform over the wave
        const Q4 ql = ldq(s.Xq[xl]);
        double R[3][3];
        qmat(ql, R);
        pb = add(ld3(s.Xp[xl]), mulv(R, pb));
        qb = qmul(ql, qb);
      }
      double Rb[3][3];
      qmat(qb, Rb);
      const V3 h = ld3(bt.box_half[b]);
      int room = bt.cap[b] & 255;                       // contacts this block may still contribute (its first ones in pair order)
      const bool capsule = Lim<NV>::CAPS && ((bt.cap[b] >> 8) & 1);   // uniform: edges vs a capsule instead of spheres / points vs a box
      for (int base = bt.begin[b]; base < pend; base += LPE) {
        const int pi = base + sub < pend ? base + sub : pend - 1;
        const bool valid = mine && base + sub < pend;
        int lk, cls;
        double r, margin, hl;
        V3 c, ed;
        if (b == pf_blk && base == bt.begin[b]) {       // uniform: the record prefetched after C0
          lk = pf_link; cls = pf_cls; r = pf_r; margin = pf_margin; c = pf_pos; ed = pf_dir; hl = pf_hl;
        } else {
          lk = col->pair_rec[pi].sph_link; cls = col->pair_rec[pi].cls;
          r = col->pair_rec[pi].r; margin = col->pair_rec[pi].margin;
          c = ld3(col->pair_rec[pi].pos);
          if constexpr (Lim<NV>::CAPS) { ed = ld3(col->pair_rec[pi].dir); hl = col->pair_rec[pi].hl; } else { ed = V3{0, 0, 0}; hl = 0; }
        }
        {
          double R[3][3];
          qmat(ldq(s.Xq[lk < 0 ? 0 : lk]), R);
          const V3 w = add(ld3(s.Xp[lk < 0 ? 0 : lk]), mulv(R, c));
          c = selv(lk < 0, c, w);
          if constexpr (Lim<NV>::CAPS) ed = selv(lk < 0, ed, mulv(R, ed));
        }
        double dist;
        V3 nl, q;                                        // normal and surface point in the box frame
        bool sane = true;
        if (Lim<NV>::CAPS && capsule) {
          // closest points of the edge (c +- hl ed) and the capsule's axis segment (pb +- hc cd); normal from the axis to the edge
          const V3 cd{Rb[0][2], Rb[1][2], Rb[2][2]}, rr = vsub(c, pb);
          const double hc = h.z - h.x, rad = h.x;
          const double b_ = dot(ed, cd), c_ = dot(ed, rr), f_ = dot(cd, rr), den = 1.0 - b_ * b_;
          double s_ = den > 1e-12 ? fmin(fmax((b_ * f_ - c_) / den, -hl), hl) : 0.0;
          const double t_ = fmin(fmax(fma(b_, s_, f_), -hc), hc);
          s_ = fmin(fmax(fma(b_, t_, -c_), -hl), hl);
          const V3 d = vsub(add(rr, scl(ed, s_)), scl(cd, t_));
          const double d2 = dot(d, d);
          sane = d2 > 1e-18;
          const double inv = rsq_nr(sane ? d2 : 1.0);
          dist = d2 * inv - rad;
          const V3 nw = scl(d, inv);
          nl = mulvT(Rb, nw);
          q = V3{nl.x * rad, nl.y * rad, t_ + nl.z * rad};   // surface point of the capsule in its own frame (axis = z)
        } else {
        const V3 x = mulvT(Rb, vsub(c, pb));
        q = V3{fmin(fmax(x.x, -h.x), h.x), fmin(fmax(x.y, -h.y), h.y), fmin(fmax(x.z, -h.z), h.z)};
        const bool outside = fabs(x.x) > h.x || fabs(x.y) > h.y || fabs(x.z) > h.z;
        {
          const V3 d = vsub(x, q);
          const double d2 = dot(d, d);
          const double inv = rsq_nr(outside ? d2 : 1.0);
          // inside: leave through the nearest face (first minimum of h - |x|)
          const double gx = h.x - fabs(x.x), gy = h.y - fabs(x.y), gz = h.z - fabs(x.z);
          const int ax = (gx <= gy && gx <= gz) ? 0 : (gy <= gz ? 1 : 2);
          const double xa = pick3(x, ax), ha = pick3(h, ax), sg = xa >= 0 ? 1.0 : -1.0;
          const V3 ni{ax == 0 ? sg : 0.0, ax == 1 ? sg : 0.0, ax == 2 ? sg : 0.0};
          const V3 qi{ax == 0 ? sg * ha : x.x, ax == 1 ? sg * ha : x.y, ax == 2 ? sg * ha : x.z};
          dist = outside ? d2 * inv - r : -(ha - fabs(xa)) - r;
          nl = selv(outside, scl(d, inv), ni);
          q = selv(outside, q, qi);
        }
        }
        const bool hit = valid && sane && dist < margin;
        const unsigned long long bal = __ballot(hit);
        const unsigned int gb = LPE == 64 ? 0u : (unsigned int)((bal >> (grp * (LPE & 63))) & ((1ull << (LPE & 63)) - 1ull));
        const int before = LPE == 64 ? __popcll(bal & ((1ull << sub) - 1ull)) : __popc(gb & ((1u << sub) - 1u));
        const int total = LPE == 64 ? __popcll(bal) : __popc(gb);
        const int slot = nct + before;
        if (hit && slot < maxcon && before < room) {
          const V3 n = mulv(Rb, nl);
          const V3 p = add(add(pb, mulv(Rb, q)), scl(n, 0.5 * dist));
          double* o = ctw[slot];
          o[0] = dist; o[1] = n.x; o[2] = n.y; o[3] = n.z; o[4] = p.x; o[5] = p.y; o[6] = p.z;
          o[7] = (double)(cls + 64 * (lk + 1) + 4096 * (xl + 1));
        }
        const int took = total < room ? total : room;
        room -= took;
        nct = nct + took < maxcon ? nct + took : maxcon;
      }
    }
    fence();
  }
  }                                                    // (ROLE 0 / 4)
  PSTAMP(2);
  if constexpr (ROLE == 4) {                           // the collision wave: the records are in wave B's block; leave their count there and go
    fence();
    if (sub == 0) peer->duo_nct = nct;
    __syncthreads();                                   // barrier X
    PSTAMP(10);
    __syncthreads();                                   // barrier Y
    PSTAMP(12);
    __syncthreads();                                   // barrier Z
    return;
  }
  // most over the wave (uniform loop bound for the contact phases)
  PCOUNT(20, 1); PCOUNT(21, nearw ? 1 : 0); PCOUNT(22, __popcll((unsigned long long)nearw));
  PCOUNT(31, __popcll((unsigned long long)nearw & 0x3Full));      // (the door's six 4-pair capsule blocks among them)
  int ncmax = 0;
  if (ROLE != 2 && nearw && __any(nct > 0)) {          // (wave B of a split timestep: after barrier X, from the count the collision wave left)
#pragma unroll
    for (int k = 0; k < MC; ++k) ncmax = __any(nct > k) ? k + 1 : ncmax;
  }
#ifdef EARL_PHYS_VARIANT_W8
  // Two waves share a SIMD in this build, and the launch lasts as long as its slowest wave -- the one whose envs are in contact.  A wave with
  // contacts in this timestep takes the issue slot first (s_setprio) for the rest of it; the wave it delays has slack.
  if (ncmax > 0) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0);
#endif
  // ------------------------------------------------------------------ K8: constraint rows
  double Jc[6];                                        // this lane's column of the weld Jacobian
  V3 rpos, rrot;
  if constexpr (!Lim<NV>::WELD) {                      // no mocap weld: six empty rows (weight 0)
#pragma unroll
    for (int r = 0; r < 6; ++r) Jc[r] = 0.0;
    rpos = V3{0, 0, 0}; rrot = V3{0, 0, 0};
    if (isl) {
#pragma unroll
      for (int r = 0; r < 6; ++r) s.con.J6[r][l] = 0.0;
    }
    if (sub < 6) { s.con.wD[sub] = 0.0; s.con.war[sub] = 0.0; }
  } else {
    const int k = m.weld_att, la = m.att_link[k];
    const Q4 ql = ldq(s.Xq[la]);
    double R[3][3];
    qmat(ql, R);
    const V3 hp = add(ld3(s.Xp[la]), mulv(R, ld3(m.att_pos[k])));
    const Q4 hq = qmul(ql, ldq(m.att_quat[k]));
    // rows as mj_instantiateEqual builds them (body1 = mocap, body2 = hand, relpose = identity): position error
    // mocap - hand; orientation error = vector part of e = conj(q_hand) * q_mocap, with the exact Jacobian of that
    // vector part: -0.5 * (e_w a + a x e_v), a = R_hand^T w_j  (no sign flip for e_w < 0).  q_mocap is used AS GIVEN: metaworld's
    // [1, 0, 1, 0] scales residual and Jacobian by sqrt 2 (the rule that replaced round 1's fitted rotational factor)
    const Q4 qe = qmul(Q4{hq.w, -hq.x, -hq.y, -hq.z}, mq);
    const V3 ev{qe.x, qe.y, qe.z};
    double Rh[3][3];
    qmat(hq, Rh);
    rrot = ev;
    rpos = vsub(ld3(s.mocap), hp);
    const bool inchain = isl && ((m.anc_mask[la] >> l) & 1u);
    const V3 pv = add(Sv, cross(Sw, hp));
    const V3 aa = mulvT(Rh, Sw);
    const V3 jq = add(scl(aa, qe.w), cross(aa, ev));
    Jc[0] = inchain ? -pv.x : 0.0; Jc[1] = inchain ? -pv.y : 0.0; Jc[2] = inchain ? -pv.z : 0.0;
    Jc[3] = inchain ? -0.5 * jq.x : 0.0; Jc[4] = inchain ? -0.5 * jq.y : 0.0; Jc[5] = inchain ? -0.5 * jq.z : 0.0;
    if (isl) {
#pragma unroll
      for (int r = 0; r < 6; ++r) s.con.J6[r][l] = Jc[r];
    }
  }
  fence();
  if constexpr (Lim<NV>::WELD) {
    // weld rows: lane = row (< 6)
    const int r = sub < 6 ? sub : 5;
    double Jv = 0;
#pragma unroll
    for (int j = 0; j < (TS < NT ? TS : NT); ++j) Jv = fma(s.con.J6[r][j], s.qv[j], Jv);          // (the weld's chain lies within the first tree)
    const double res = r < 3 ? pick3(rpos, r) : pick3(rrot, r - 3);
    double kk = bt.kb_weld[0], bb = bt.kb_weld[1], dd;
    if constexpr (Lim<NV>::EXTRAS) dd = imp_p2(m.weld_solimp, res);
    else if constexpr (Lim<NV>::KBT) dd = imp_of(m.weld_solimp, res);
    else kbimp(m.weld_solref, m.weld_solimp, res, dt, kk, bb, dd);
    const double Rg = fmax((1 - dd) * m.weld_invweight[r < 3 ? 0 : 1] * rcp_nr(dd), 1e-15);
    if (sub < 6) { s.con.wD[r] = rcp_nr(Rg); s.con.war[r] = -bb * Jv - kk * dd * res; }
  }
  if constexpr (ROLE == 0 || R_OWNER) jeq_records(s);      // (measured on the collision wave of a split timestep instead: 190 -> 193 ms per 256-env launch -- in the envs the launch waits for,
                                                             // the ones in contact, that wave is the longest before barrier X)
  // connect constraints (reference: LinkModel.forward): attachments con_att1[e] / con_att2[e] coincide; rows 3 e + c, residual (p1 - p2)[c],
  // Jacobian Jp(link1, p1) - Jp(link2, p2); this lane's column first, then lane = row for the reference accelerations
  if constexpr (Lim<NV>::CONNECT) {
    for (int e = 0; e < m.n_con; ++e) {
      const int k1 = m.con_att1[e], k2 = m.con_att2[e], l1 = m.att_link[k1], l2 = m.att_link[k2];
      double R[3][3];
      qmat(ldq(s.Xq[l1]), R);
      const V3 p1 = add(ld3(s.Xp[l1]), mulv(R, ld3(m.att_pos[k1])));
      qmat(ldq(s.Xq[l2]), R);
      const V3 p2 = add(ld3(s.Xp[l2]), mulv(R, ld3(m.att_pos[k2])));
      const double w1 = (isl && ((m.anc_mask[l1] >> l) & 1u)) ? 1.0 : 0.0, w2 = (isl && ((m.anc_mask[l2] >> l) & 1u)) ? 1.0 : 0.0;
      const V3 j1 = scl(add(Sv, cross(Sw, p1)), w1), j2 = scl(add(Sv, cross(Sw, p2)), w2);
      if (isl) {
        s.xt.JE[3 * e][l] = j1.x - j2.x; s.xt.JE[3 * e + 1][l] = j1.y - j2.y; s.xt.JE[3 * e + 2][l] = j1.z - j2.z;
      }
      if (sub == 0) { s.xt.eres[3 * e] = p1.x - p2.x; s.xt.eres[3 * e + 1] = p1.y - p2.y; s.xt.eres[3 * e + 2] = p1.z - p2.z; }
    }
    fence();
    if (sub < 3 * m.n_con) {
      const int e = sub / 3;
      double Jv = 0;
#pragma unroll
      for (int j = 0; j < NV; ++j) Jv = fma(s.xt.JE[sub][j], s.qv[j], Jv);
      const double res = s.xt.eres[sub];
      double kk, bb, dd;
      kbimp(m.con_solref[e], m.con_solimp[e], res, dt, kk, bb, dd);
      s.xt.eD[sub] = rcp_nr(fmax((1 - dd) * m.con_invweight[e] * rcp_nr(dd), 1e-15));
      s.xt.ear[sub] = -bb * Jv - kk * dd * res;
    }
    fence();
  }
  // limit row of this lane's dof: at most one side can be violated
  double lim_D, lim_aref;
  bool lim_lo;                                         // which side: the row's sign +1 (lower limit) / -1 is applied as a select (x or -x: the same bits as
                                                       // the product with +-1.0, and one fp64 value less to keep -- the peg build reloaded it from scratch
                                                       // memory in every pass of the active-set iteration)
  bool lim_inst, lim_start;
  {
    const double q = s.qp[l], lo = m.range[l][0], hi = m.range[l][1];
    const bool islo = q - lo < 0;
    const double res = islo ? q - lo : hi - q;
    lim_lo = islo;
    lim_inst = isl && m.limited[l] && res < 0;
    double kk = bt.kb_lim[Lim<NV>::KBT ? l : 0][0], bb = bt.kb_lim[Lim<NV>::KBT ? l : 0][1];
    double dd;
    if constexpr (Lim<NV>::EXTRAS) dd = imp_p2(m.jsolimp[l], res);
    else if constexpr (Lim<NV>::KBT) dd = imp_of(m.jsolimp[l], res);
    else kbimp(m.jsolref[l], m.jsolimp[l], res, dt, kk, bb, dd);
    lim_D = rcp_nr(fmax((1 - dd) * m.dof_invweight[l] * rcp_nr(dd), 1e-15));
    lim_aref = -bb * (lim_lo ? s.qv[l] : -s.qv[l]) - kk * dd * res;
    // start of the active-set iteration: the row if it is violated; warm: if it also pulls at a_prev.  (Where this line stands matters to the register
    // allocator: here the eight-wave door build spills 100 B less than with the test at the head of K9, there the peg build is 3 % faster.)
    if constexpr (NV <= 10) lim_start = lim_inst && (!warm || (lim_lo ? s.aprev[l] : -s.aprev[l]) - lim_aref < 0);
    else lim_start = lim_inst;
  }
  // dry friction of this lane's dof (mjCNSTR_FRICTION_DOF): residual 0, cost 1/2 D x^2 for |x| <= loss / D, linear beyond (x = a_l - aref);
  // state 0 = quadratic zone (adds D to the diagonal), +-1 = saturated (constant force -+loss)
  double fr_D = 0, fr_aref = 0, fr_loss = 0;
  int fr_state = 0;
  if constexpr (Lim<NV>::EXTRAS) {
    fr_loss = isl ? m.frictionloss[l] : 0.0;
    fr_D = bt.fr_D[l];
    fr_aref = -bt.kb_lim[l][1] * s.qv[l];
  }
  PSTAMP(7);
  if constexpr (R_OWNER) {                             // wave B: its weld rows and coupling records are in place for wave A; mass matrix, generalized forces and contact records are in its block
    static_assert(!R_OWNER || (Lim<NV>::EXTRAS && Lim<NV>::ARMSCAN), "the split timestep is the kitchen model's");
    fence();
    __syncthreads();                                   // barrier X
    PSTAMP(10);
    tau_l = s.tau[l];
    if constexpr (ROLE == 2) {                         // (four-wave split: the collision wave's count; the two-wave split's owner ran the collision itself)
      nct = s.duo_nct;
      if (__any(nct > 0)) {
#pragma unroll
        for (int k = 0; k < MC; ++k) ncmax = __any(nct > k) ? k + 1 : ncmax;
      }
    }
  }
  // ------------------------------------------------------------------ C3: contact rows (reference: LinkModel.contact_rows)
  double cD = 0, cmu = 0, car[4] = {0, 0, 0, 0};       // lane c (< nct) owns contact c: edge weights and reference accelerations
  unsigned int cact = 0;                               // active pyramid edges of that contact (bits 0..3); elliptic models: its zone
  double cja[3] = {0, 0, 0};                           // elliptic models: J a (normal, t1, t2) of the iterate the zone was read from
  bool coupled = false;                                // some contact of some env of the wave joins the two trees (arm / object)
  bool ctA = true, ctP = true;                         // ... and of the contact this lane owns
  unsigned int armmask = 0, pegmask = 0;               // two-tree model: contact slots whose Jacobian has entries in the first / second tree in SOME env of the wave
  if (ncmax > 0) {
    auto contact_jac = [&](const int c) {
      // (the record as one batch of loads, then the two links' ancestor masks as another -- physics_math.h pin_batch; selected per load, the packed entry and the masks
      // each sat under a branch with a wait of its own: four LDS round trips per contact, one after the other)
      double rec[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) rec[k] = s.con.ct[c][k];
      pin_batch(rec);
      const bool cv = c < nct;
      const V3 n = selv(cv, V3{rec[1], rec[2], rec[3]}, V3{0, 0, 1}), p = selv(cv, V3{rec[4], rec[5], rec[6]}, V3{0, 0, 0});
      const int pk = cv ? (int)rec[7] : 0;
      const int ls = ((pk >> 6) & 63) - 1, lb = (pk >> 12) - 1;
      unsigned int am_s = m.anc_mask[ls < 0 ? 0 : ls], am_b = m.anc_mask[lb < 0 ? 0 : lb];
      asm volatile("" : "+v"(am_s), "+v"(am_b));
      coupled = coupled || (ls >= 0 && lb >= 0 && ((ls < NA) != (lb < NA)));
      if constexpr (TS < NT) {
        armmask |= __any(cv && ((ls >= 0 && ls < TS) || (lb >= 0 && lb < TS))) ? (1u << c) : 0u;
        pegmask |= __any(cv && (ls >= TS || lb >= TS)) ? (1u << c) : 0u;
      }
      // tangents: n x (the coordinate axis least aligned with n), normalised, then n x t1
      const double ax_ = fabs(n.x), ay_ = fabs(n.y), az_ = fabs(n.z);
      const int ia = (ax_ <= ay_ && ax_ <= az_) ? 0 : (ay_ <= az_ ? 1 : 2);
      const V3 e{ia == 0 ? 1.0 : 0.0, ia == 1 ? 1.0 : 0.0, ia == 2 ? 1.0 : 0.0};
      V3 t1 = cross(n, e);
      t1 = scl(t1, rsq_nr(dot(t1, t1)));
      const V3 t2 = cross(n, t1);
      const double w = ((ls >= 0 && ((am_s >> l) & 1u)) ? 1.0 : 0.0) - ((lb >= 0 && ((am_b >> l) & 1u)) ? 1.0 : 0.0);
      const V3 Jp = scl(add(Sv, cross(Sw, p)), w);
      if constexpr (Lim<NV>::EXTRAS) {                    // (no branch: two contacts' chains of LDS round trips run side by side below)
        double* const dump = reinterpret_cast<double*>(s.bank_pad);
        *((isl && cv) ? &s.con.CJ[c][0][l] : dump) = dot(n, Jp);
        *((isl && cv) ? &s.con.CJ[c][1][l] : dump) = dot(t1, Jp);
        *((isl && cv) ? &s.con.CJ[c][2][l] : dump) = dot(t2, Jp);
      } else {
      if (isl && cv) {
        s.con.CJ[c][0][l] = dot(n, Jp);
        s.con.CJ[c][1][l] = dot(t1, Jp);
        s.con.CJ[c][2][l] = dot(t2, Jp);
      }
      }
    };
    if constexpr (Lim<NV>::EXTRAS) {
      static_assert(MC % 2 == 0, "contact slots in pairs");
      for (int c2 = 0; c2 < ncmax; c2 += 2) { contact_jac(c2); contact_jac(c2 + 1); }      // (a slot beyond the env's count is selected away inside)
    } else {
      for (int c = 0; c < ncmax; ++c) contact_jac(c);
    }
    fence();
    {
      const int c = sub < MC ? sub : MC - 1;
      const bool cv = sub < nct;
      const double* rec = s.con.ct[c];
      if constexpr (TS < NT) {                          // two-tree model: the trees in which this lane's contact has Jacobian entries (the other rows are exact zeros)
        const int pk = cv ? (int)rec[7] : 0;
        const int ls = ((pk >> 6) & 63) - 1, lb = (pk >> 12) - 1;
        ctA = cv && ((ls >= 0 && ls < TS) || (lb >= 0 && lb < TS));
        ctP = cv && (ls >= TS || lb >= TS);
      }
      double vn = 0, vt1 = 0, vt2 = 0, pn = 0, pt1 = 0, pt2 = 0;      // J qvel; J a_prev (warm start)
      // dofs [J0, J0 + N): their entries of the three rows, velocities and previous accelerations as ONE batch of loads (physics_math.h pin_batch), then the six sums in
      // their order of additions -- left to the scheduler the loads came a row at a time, each with its own wait
      auto rows = [&](auto j0c, auto nc) {
        constexpr int J0 = decltype(j0c)::value, N = decltype(nc)::value;
        double cj[3 * N], qa[2 * N];
#pragma unroll
        for (int j = 0; j < N; ++j) {
          cj[3 * j] = s.con.CJ[c][0][J0 + j]; cj[3 * j + 1] = s.con.CJ[c][1][J0 + j]; cj[3 * j + 2] = s.con.CJ[c][2][J0 + j];
          qa[2 * j] = s.qv[J0 + j]; qa[2 * j + 1] = s.aprev[J0 + j];
        }
        pin_batch(cj); pin_batch(qa);
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const double qd = qa[2 * j], ap = qa[2 * j + 1], jn = cj[3 * j], j1 = cj[3 * j + 1], j2 = cj[3 * j + 2];
          vn = fma(jn, qd, vn); vt1 = fma(j1, qd, vt1); vt2 = fma(j2, qd, vt2);
          pn = fma(jn, ap, pn); pt1 = fma(j1, ap, pt1); pt2 = fma(j2, ap, pt2);
        }
      };
      using std::integral_constant;
      if constexpr (TS < NT) {
        if (ctA) rows(integral_constant<int, 0>{}, integral_constant<int, TS>{});
        if (ctP) rows(integral_constant<int, TS>{}, integral_constant<int, NV - TS>{});
      } else if constexpr (NV <= 12) {
        rows(integral_constant<int, 0>{}, integral_constant<int, NV>{});
      } else {
        rows(integral_constant<int, 0>{}, integral_constant<int, 8>{});
        rows(integral_constant<int, 8>{}, integral_constant<int, 8>{});
        rows(integral_constant<int, 16>{}, integral_constant<int, NV - 16>{});
      }
      const int cls = cv ? ((int)rec[7] & 63) : 0;
      if constexpr (Lim<NV>::CONNECT) {
        // one-tree model with a free root body (dofs 0-5) and chains of at most two hinges, colliding with world-fixed boxes only (checked by the host
        // side): a contact Jacobian has entries in the root's six dofs and in the sphere's own chain -- nothing else.  K9 updates only those rows.
        if (sub < MC) {
          const int ls = cv ? (((int)rec[7] >> 6) & 63) - 1 : -1;
          const int d2 = ls >= 6 ? ls : -1, d1 = (d2 >= 0 && m.parent[d2] >= 6) ? m.parent[d2] : -1;
          s.xt.crow[sub][0] = (signed char)d1; s.xt.crow[sub][1] = (signed char)d2;
        }
      }
      const double margin = bt.cls_margin[cls];
      cmu = bt.cls_mu[cls];
      if constexpr (Lim<NV>::CONNECT) {                 // Minitaur.SetFootFriction: every contact of a lower-leg link (a link behind the root body whose parent is not the root)
        const int lsf = cv ? (((int)rec[7] >> 6) & 63) - 1 : -1, root = m.ball_dof + 2;
        if (s.xt.foot_mu > 0 && lsf > root && m.parent[lsf] != root) cmu = s.xt.foot_mu;
      }
      double kk = bt.kb_cls[Lim<NV>::KBT ? cls : 0][0], bb = bt.kb_cls[Lim<NV>::KBT ? cls : 0][1];
      double dd;
      if constexpr (Lim<NV>::EXTRAS) dd = imp_p2(bt.cls_solimp[cls], rec[0] - margin);
      else if constexpr (Lim<NV>::KBT) dd = imp_of(bt.cls_solimp[cls], rec[0] - margin);
      else kbimp(bt.cls_solref[cls], bt.cls_solimp[cls], rec[0] - margin, dt, kk, bb, dd);
      const double R0 = fmax((1 - dd) * bt.cls_invw[cls] * rcp_nr(dd), 1e-15);
      const double basea = -kk * dd * (rec[0] - margin);
      if constexpr (Lim<NV>::ELLIPTIC) {
        // rows (normal, t1, t2), one regulariser (impratio 1), the position term on the normal row only; cact = the contact's ZONE (0 top, 1 bottom, 2 middle),
        // cja = J a of the iterate the zone was read from (a_prev at a warm start; a cold start puts every contact in the bottom zone)
        cD = cv ? rcp_nr(R0) : 0.0;
        car[0] = -bb * vn + basea; car[1] = -bb * vt1; car[2] = -bb * vt2;
        cja[0] = pn; cja[1] = pt1; cja[2] = pt2;
        cact = cv ? (warm ? (unsigned int)cone_zone(pn - car[0], pt1 - car[1], pt2 - car[2], cmu) : 1u) : 0u;
      } else {
      cD = cv ? rcp_nr(2 * cmu * cmu * R0) : 0.0;
      car[0] = -bb * (vn + cmu * vt1) + basea; car[1] = -bb * (vn - cmu * vt1) + basea;
      car[2] = -bb * (vn + cmu * vt2) + basea; car[3] = -bb * (vn - cmu * vt2) + basea;
      unsigned int wb = 0;                              // the edges that pull at a_prev
      wb |= (pn + cmu * pt1 - car[0] < 0) ? 1u : 0u;
      wb |= (pn - cmu * pt1 - car[1] < 0) ? 2u : 0u;
      wb |= (pn + cmu * pt2 - car[2] < 0) ? 4u : 0u;
      wb |= (pn - cmu * pt2 - car[3] < 0) ? 8u : 0u;
      cact = cv ? (warm ? wb : 0xFu) : 0u;
      }
    }
  }
  fence();
  PSTAMP(6);
  // ------------------------------------------------------------------ K9: Hessian of the equality part, then the active-set Newton
  double hw[(Lim<NV>::EXTRAS || Lim<NV>::CONNECT) ? 1 : NV], rw;             // this lane's column of M + J6' D J6 (+ drag) and its right-hand side: registers, all iterations
                                                       // (big model: the column goes straight to LDS, s.hwst.Hw)
  {
    double DJ[6], g = tau_l;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      DJ[r] = s.con.wD[r] * Jc[r];
      g = fma(DJ[r], s.con.war[r], g);
    }
    if constexpr (Lim<NV>::CONNECT) {
      // nv = 22: column l (its lower part) of M + JE' D JE -- the connect rows are equalities, always active -- goes straight to LDS (s.hwst.Hw): a
      // 22-entry register column kept across the whole active-set iteration is what made this kernel spill 1.3 KB per lane
      double dj[3 * EARL_MAXCONNECT];
#pragma unroll
      for (int r = 0; r < 3 * EARL_MAXCONNECT; ++r) {
        dj[r] = r < 3 * m.n_con ? s.xt.eD[r] * s.xt.JE[r][l] : 0.0;
        g = fma(dj[r], r < 3 * m.n_con ? s.xt.ear[r] : 0.0, g);
      }
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        double h = s.M.sym(i, l, ltri);
#pragma unroll
        for (int r = 0; r < 3 * EARL_MAXCONNECT; ++r) h = fma(s.xt.JE[r][i], dj[r], h);
        if (isl && i >= l) s.hwst.Hw.lo(i, l) = h;
      }
    } else if constexpr (!Lim<NV>::EXTRAS) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        double h = s.M.sym(i, l, ltri);
        if (i < (TS < NT ? TS : NT)) {                     // (the weld's chain lies within the first tree: its Jacobian is zero in the other rows)
#pragma unroll
          for (int r = 0; r < 6; ++r) h = fma(s.con.J6[r][i], DJ[r], h);
        }
        if (i == l) h += m.drag_G[l];                     // soft velocity row of a permanent dragging contact
        hw[i] = h;
      }
    }
    g -= m.drag_G[l] * m.drag_b[l] * s.qv[l];
    if constexpr (Lim<NV>::EXTRAS) {
      if (m.pair[l] >= 0) {                             // (every other coupling's term in this lane's sum was D . 0 . aref)
        const double* const rec = s.jeq.rec[l >= NT ? l - NT : 0];
        g = fma(rec[0], rec[1], g);
      }
      if constexpr (!R_OWNER) hw_extras(s, DJ);         // (split timestep: wave A's, in place at barrier Y)
    }
    rw = g;
  }
  PSTAMP(9);
  if constexpr (R_OWNER) { __syncthreads(); PSTAMP(12); }       // barrier Y: mass matrix and equality Hessian are wave A's
  coupled = __any(coupled);
  if constexpr (NV > 10) { if (warm) lim_start = lim_inst && ((lim_lo ? s.aprev[l] : -s.aprev[l]) - lim_aref < 0); }
  bool act = lim_start;                                // (dry-friction rows keep their cold start, the quadratic zone: from a_prev's zones the
                                                       // three-state iteration cycled 18 times as often in the kitchen model)
  double a[NV];
  double L[NV * (NV + 1) / 2];
  PCOUNT(23, ncmax > 0 ? 1 : 0); PCOUNT(24, ncmax); PCOUNT(26, coupled ? 1 : 0);
#ifdef EARL_PHYS_PROF
  const unsigned long long k9_t0 = __builtin_readcyclecounter();
#endif
  bool frozen = false;                                 // elliptic models: this env reached its fixed point in an earlier pass (its solution is parked in s.aprev; the wave goes on
                                                       // for the envs that have not) -- an env's result must not depend on which envs share its wave
  for (int it = 0; it < 8; ++it) {
    PCOUNT(25, 1);
    KSTART();
    if (isl) {
      const double lda = lim_D * lim_aref;
      double dlv = act ? lim_D : 0.0, rlv = act ? (lim_lo ? lda : -lda) : 0.0;
      if constexpr (Lim<NV>::EXTRAS) {
        if (fr_loss > 0) {
          dlv += fr_state == 0 ? fr_D : 0.0;
          rlv += fr_state == 0 ? fr_D * fr_aref : -(double)fr_state * fr_loss;
        }
      }
      s.con.dl[l] = dlv;
      s.con.rl[l] = rlv;
    }
    if (ncmax > 0 && sub < MC) {
      // edges (n + mu t1, n - mu t1, n + mu t2, n - mu t2): sum_e D a_e u_e u_e' on (Jn, Jt1, Jt2) and sum_e D a_e aref_e u_e
      double* w = s.con.cw[sub];
      if constexpr (Lim<NV>::ELLIPTIC) {
        // the record cone_apply reads: (K, m1, m2, q, 1 / mu^2) and the right-hand side h = W (J a_k) - grad.  Bottom zone: W = D I, h = D aref.  Middle zone, with
        // r = J a_k - aref, rho = |r_t|, sl = r_n - mu rho < 0: K = D / (1 + mu^2), m = -mu r_t / rho, q = -K mu sl / rho, grad = K sl (1, m1, m2)
        double K = 0, m1 = 0, m2 = 0, q = 0, h0 = 0, h1 = 0, h2 = 0;
        const double i2 = cmu > 0 ? rcp_nr(cmu * cmu) : 0.0;      // (a frictionless class: m = 0 and q = 0, the record is the normal row alone)
        if (cact == 1u) {
          K = cD; q = cD; h0 = cD * car[0]; h1 = cD * car[1]; h2 = cD * car[2];
        } else if (cact == 2u) {
          const double r0 = cja[0] - car[0], r1 = cja[1] - car[1], r2 = cja[2] - car[2];
          const double rho = sqrt(r1 * r1 + r2 * r2), ir = 1.0 / rho, sl = r0 - cmu * rho;
          K = cD / (1.0 + cmu * cmu); m1 = -cmu * r1 * ir; m2 = -cmu * r2 * ir; q = -K * cmu * sl * ir;
          const double rec_[5] = {K, m1, m2, q, i2};
          cone_apply<true>(rec_, cja[0], cja[1], cja[2], h0, h1, h2);
          h0 -= K * sl; h1 -= K * sl * m1; h2 -= K * sl * m2;
        }
        w[0] = K; w[1] = m1; w[2] = m2; w[3] = q; w[4] = i2; w[5] = h0; w[6] = h1; w[7] = h2;
      } else {
      const double a1 = (cact & 1u) ? cD : 0.0, a2 = (cact & 2u) ? cD : 0.0, a3 = (cact & 4u) ? cD : 0.0, a4 = (cact & 8u) ? cD : 0.0;
      w[0] = a1 + a2 + a3 + a4; w[1] = cmu * (a1 - a2); w[2] = cmu * (a3 - a4); w[3] = cmu * cmu * (a1 + a2); w[4] = cmu * cmu * (a3 + a4);
      w[5] = a1 * car[0] + a2 * car[1] + a3 * car[2] + a4 * car[3];
      w[6] = cmu * (a1 * car[0] - a2 * car[1]);
      w[7] = cmu * (a3 * car[2] - a4 * car[3]);
      }
    }
    fence();
    KSTAMP(16);
    if constexpr (Lim<NV>::EXTRAS) {
      // The iteration's Hessian = the stored equality part + the active contact edges, by the model's structure (checked by the host side): a contact joins arm
      // links (dofs < NA) and at most ONE fixture (a single-dof tree), so its J' W J has entries in the arm's block, in that fixture's row against the arm
      // and on that fixture's diagonal -- nothing between two fixtures.  Every lane accumulates, in registers, v = W J_l for its own dof and the NA arm rows
      // J_i . v: an arm lane keeps rows i >= l of its column, a fixture lane f gets its row (f, i) against the arm (its own J entry is zero unless the
      // contact touches it) and its diagonal.  (The earlier form walked all NV rows of the column per contact with a read-modify-write in LDS each:
      // 20 k cycles per timestep in the wave whose fingers are on a fixture -- the wave the launch waits for.)
      // No contact in any env of the wave (most timesteps of most waves): the iteration's Hessian IS the equality part -- K9's solver reads it where it lies
      // (Hs below) instead of from a copy made with nine LDS round trips in a row (x + 0.0 = x: the same bits)
      if (ncmax == 0) {
        if (isl) s.con.rc[l] = rw;
      } else {
      double rr = rw, acc[NA], accd = 0.0;
#pragma unroll
      for (int i = 0; i < NA; ++i) acc[i] = 0.0;
      static_assert(MC % 2 == 0, "contact slots in pairs");
      // this lane's entries of the equality part, loaded before the contact loop (independent of it): an arm lane's column, a fixture lane's diagonal and shared entry
      double hwv[NA];
      const int plh = m.pair[l];
#pragma unroll
      for (int i = 0; i < NA; ++i) hwv[i] = s.hwst.Hw.lo(i >= l ? i : NA - 1, l < NA ? l : 0);
      const double hwd = s.hwst.Hw.lo(l, l), hwo = s.hwst.Hw.lo(plh > l ? plh : l, l);
      for (int c2 = 0; c2 < ncmax; c2 += 2) {               // two contacts per iteration, their loads side by side (a slot beyond the count is selected away, not multiplied)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int c = c2 + u;
          // (the contact's weights, this lane's entries and the arm's rows as ONE batch of loads -- physics_math.h pin_batch -- and the selects on the sums: with the
          // loads inside the selects every row was a branch around three loads with a wait of their own)
          double w[8], jl[3], cj[3 * NA];
#pragma unroll
          for (int k = 0; k < 8; ++k) w[k] = s.con.cw[c][k];
#pragma unroll
          for (int k = 0; k < 3; ++k) jl[k] = s.con.CJ[c][k][l];
#pragma unroll
          for (int i = 0; i < NA; ++i) { cj[3 * i] = s.con.CJ[c][0][i]; cj[3 * i + 1] = s.con.CJ[c][1][i]; cj[3 * i + 2] = s.con.CJ[c][2][i]; }
          pin_batch(w); pin_batch(jl); pin_batch(cj);
          const double j0 = jl[0], j1 = jl[1], j2 = jl[2];
          const bool cv = c < nct;
          double v0, v1, v2;
        cone_apply<Lim<NV>::ELLIPTIC>(w, j0, j1, j2, v0, v1, v2);
        v0 = cv ? v0 : 0.0; v1 = cv ? v1 : 0.0; v2 = cv ? v2 : 0.0;
          rr += cv ? w[5] * j0 + w[6] * j1 + w[7] * j2 : 0.0;
#pragma unroll
          for (int i = 0; i < NA; ++i) acc[i] += cv ? cj[3 * i] * v0 + cj[3 * i + 1] * v1 + cj[3 * i + 2] * v2 : 0.0;
          accd += cv ? j0 * v0 + j1 * v1 + j2 * v2 : 0.0;
        }
      }
      {
        // an arm lane stores rows i >= l of its column, a fixture lane its row against the arm, its diagonal and the entry it shares with its partner: the same
        // nine + two stores in every lane, the ones a lane does not have go to the block's padding (no branch per row)
        double* const dump = reinterpret_cast<double*>(s.bank_pad);
        const bool arm = l < NA;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
          const double v = arm ? hwv[i] + acc[i] : acc[i];                       // (the equality part has no entries between the arm and a fixture)
          *(!isl ? dump : (arm ? (i >= l ? &s.con.Hc.lo(i >= l ? i : l, l) : dump) : &s.con.Hc.lo(l, i))) = v;
        }
        *((isl && !arm) ? &s.con.Hc.lo(l, l) : dump) = hwd + accd;
        *((isl && !arm && plh > l) ? &s.con.Hc.lo(plh > l ? plh : l, l) : dump) = hwo;
        if (isl) s.con.rc[l] = rr;
      }
      }
    } else if constexpr (Lim<NV>::CONNECT) {
      // column l of the iteration's Hessian: the stored equality part + the active contact edges, summed in registers, stored once (lower part)
      // Rows touched by a contact: the root body's six (accumulated in registers) and the at most two dofs of the sphere's own chain (s.xt.crow, updated in
      // place): 8 of the 22 rows per contact instead of all 22 (the others' Jacobian entries are exact zeros).
      double acc[6] = {0, 0, 0, 0, 0, 0}, rr = rw;
      if (isl) {
#pragma unroll
        for (int i = 0; i < NV; ++i) if (i >= l) s.con.Hc.lo(i, l) = s.hwst.Hw.lo(i, l);
      }
      for (int c = 0; c < ncmax; ++c) {
        const double* w = s.con.cw[c];
        const double j0 = s.con.CJ[c][0][l], j1 = s.con.CJ[c][1][l], j2 = s.con.CJ[c][2][l];
        const bool cv = c < nct;
        double v0, v1, v2;
        cone_apply<Lim<NV>::ELLIPTIC>(w, j0, j1, j2, v0, v1, v2);
        v0 = cv ? v0 : 0.0; v1 = cv ? v1 : 0.0; v2 = cv ? v2 : 0.0;
        rr += cv ? w[5] * j0 + w[6] * j1 + w[7] * j2 : 0.0;
#pragma unroll
        for (int i = 0; i < 6; ++i) acc[i] += cv ? s.con.CJ[c][0][i] * v0 + s.con.CJ[c][1][i] * v1 + s.con.CJ[c][2][i] * v2 : 0.0;   // (a slot beyond this env's count holds whatever
                                                                                   // LDS held: 0 x NaN would poison the column -- tests/test_lds_hygiene_gpu.py)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int d = (int)s.xt.crow[c][k];
          if (cv && isl && d >= l) s.con.Hc.lo(d, l) += s.con.CJ[c][0][d] * v0 + s.con.CJ[c][1][d] * v1 + s.con.CJ[c][2][d] * v2;
        }
      }
      if (isl) {
#pragma unroll
        for (int i = 0; i < 6; ++i) if (i >= l) s.con.Hc.lo(i, l) += acc[i];
        s.con.rc[l] = rr;
      }
    } else if constexpr (Lim<NV>::COOP) {
      // (in-LDS factorisation, i.e. the eight-waves-per-CU door build: the same column built in place, from the register copy of the equality part --
      // one 10-entry register vector less under the 256-register cap)
      double rr = rw;
      if (isl) {
#pragma unroll
        for (int i = 0; i < NV; ++i) s.con.Hc.put(i, l, hw[i], false);
      }
      for (int c = 0; c < ncmax; ++c) {
        const double* w = s.con.cw[c];
        const double j0 = s.con.CJ[c][0][l], j1 = s.con.CJ[c][1][l], j2 = s.con.CJ[c][2][l];
        const bool cv = c < nct;
        double v0, v1, v2;
        cone_apply<Lim<NV>::ELLIPTIC>(w, j0, j1, j2, v0, v1, v2);
        v0 = cv ? v0 : 0.0; v1 = cv ? v1 : 0.0; v2 = cv ? v2 : 0.0;
        rr += cv ? w[5] * j0 + w[6] * j1 + w[7] * j2 : 0.0;
        if (isl) {
#pragma unroll
          for (int i = 0; i < NV; ++i) {
            const double t = cv ? s.con.CJ[c][0][i] * v0 + s.con.CJ[c][1][i] * v1 + s.con.CJ[c][2][i] * v2 : 0.0;
            if (!SymLds<NV>::PACKED || i >= l) s.con.Hc.lo(i, l) += t;      // (lo(i, l) addresses entry (i, l): any i in the square form, i >= l in the packed one)
          }
        }
      }
      if (isl) s.con.rc[l] = rr;
    } else {
      double hcol[NV], rr = rw;
#pragma unroll
      for (int i = 0; i < NV; ++i) hcol[i] = hw[i];
#pragma unroll 2
      for (int c = 0; c < ncmax; ++c) {
        double w[8], jl[3];                               // (the contact's weights and this lane's entries as one batch of loads: physics_math.h pin_batch)
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = s.con.cw[c][k];
#pragma unroll
        for (int k = 0; k < 3; ++k) jl[k] = s.con.CJ[c][k][l];
        pin_batch(w); pin_batch(jl);
        const double j0 = jl[0], j1 = jl[1], j2 = jl[2];
        const bool cv = c < nct;
        double v0, v1, v2;
        cone_apply<Lim<NV>::ELLIPTIC>(w, j0, j1, j2, v0, v1, v2);
        v0 = cv ? v0 : 0.0; v1 = cv ? v1 : 0.0; v2 = cv ? v2 : 0.0;
        rr += cv ? w[5] * j0 + w[6] * j1 + w[7] * j2 : 0.0;
        if constexpr (TS < NT) {
          // the rows of a tree that slot c touches in no env of the wave hold exact zeros (C3 wrote them): skipped.  A peg lying on the table gives
          // four contacts with entries in the peg's six rows only -- 18 of the 45 reads and multiply-adds per contact
          // (a tree's rows as ONE batch of loads -- physics_math.h pin_batch -- and the select on the sums: with the loads inside the select every row was a branch
          // around three loads with a wait of their own, 77 LDS round trips one after the other per pass)
#ifndef EARL_PEG_ROW_BATCH
#define EARL_PEG_ROW_BATCH 1
#endif
          if ((armmask >> c) & 1u) {
            if constexpr (EARL_PEG_ROW_BATCH) {
              double cj[3 * TS];
#pragma unroll
              for (int i = 0; i < TS; ++i) { cj[3 * i] = s.con.CJ[c][0][i]; cj[3 * i + 1] = s.con.CJ[c][1][i]; cj[3 * i + 2] = s.con.CJ[c][2][i]; }
              pin_batch(cj);
#pragma unroll
              for (int i = 0; i < TS; ++i) hcol[i] += cv ? cj[3 * i] * v0 + cj[3 * i + 1] * v1 + cj[3 * i + 2] * v2 : 0.0;
            } else {
#pragma unroll
            for (int i = 0; i < TS; ++i) hcol[i] += cv ? s.con.CJ[c][0][i] * v0 + s.con.CJ[c][1][i] * v1 + s.con.CJ[c][2][i] * v2 : 0.0;
            }
          }
          if ((pegmask >> c) & 1u) {
            if constexpr (EARL_PEG_ROW_BATCH) {
              double cj[3 * (NV - TS)];
#pragma unroll
              for (int i = TS; i < NV; ++i) { cj[3 * (i - TS)] = s.con.CJ[c][0][i]; cj[3 * (i - TS) + 1] = s.con.CJ[c][1][i]; cj[3 * (i - TS) + 2] = s.con.CJ[c][2][i]; }
              pin_batch(cj);
#pragma unroll
              for (int i = TS; i < NV; ++i) hcol[i] += cv ? cj[3 * (i - TS)] * v0 + cj[3 * (i - TS) + 1] * v1 + cj[3 * (i - TS) + 2] * v2 : 0.0;
            } else {
#pragma unroll
            for (int i = TS; i < NV; ++i) hcol[i] += cv ? s.con.CJ[c][0][i] * v0 + s.con.CJ[c][1][i] * v1 + s.con.CJ[c][2][i] * v2 : 0.0;
            }
          }
        } else {
#ifndef EARL_DOOR_ROW_BATCH
#define EARL_DOOR_ROW_BATCH 1
#endif
          if constexpr (EARL_DOOR_ROW_BATCH) {
            double cj[3 * NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) { cj[3 * i] = s.con.CJ[c][0][i]; cj[3 * i + 1] = s.con.CJ[c][1][i]; cj[3 * i + 2] = s.con.CJ[c][2][i]; }
            pin_batch(cj);
#pragma unroll
            for (int i = 0; i < NV; ++i) hcol[i] += cv ? cj[3 * i] * v0 + cj[3 * i + 1] * v1 + cj[3 * i + 2] * v2 : 0.0;
          } else {
#pragma unroll
          for (int i = 0; i < NV; ++i) hcol[i] += cv ? s.con.CJ[c][0][i] * v0 + s.con.CJ[c][1][i] * v1 + s.con.CJ[c][2][i] * v2 : 0.0;
          }
        }
      }
      if (isl) {
#pragma unroll
        for (int i = 0; i < NV; ++i) s.con.Hc.put(i, l, hcol[i], false);    // column l (packed form: its lower part)
        s.con.rc[l] = rr;
      }
    }
    fence();
    KSTAMP(17);
    {                                                    // (both vectors as one batch of loads: physics_math.h pin_batch -- they came a pair at a time, each with its own wait)
      double rlv[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) { a[i] = s.con.rc[i]; rlv[i] = s.con.rl[i]; }
      pin_batch(a); pin_batch(rlv);
#pragma unroll
      for (int i = 0; i < NV; ++i) a[i] = a[i] + rlv[i];
    }
    if constexpr (Lim<NV>::EXTRAS) {
      if (coupled && EARL_KITCHEN_DENSE) {             // (measurement switch: the generic dense factorisation in its looping form)
        if (isl) s.con.rc[l] += s.con.rl[l];
        chol_coop_loop<NV>(s.con.Hc, s.con.dl, l, isl);
        solve_lds_loop<NV>(s.con.Hc, s.con.rc);
        fence();
#pragma unroll
        for (int i = 0; i < NV; ++i) a[i] = s.con.rc[i];
      } else if (coupled) {
        // A finger touches a fixture (uniform over the wave).  H = [A B'; B F]: A the arm's NA x NA block, F the fixtures' block (1 x 1 / 2 x 2
        // blocks, no fill), B the rows that contacts put between them.  Eliminate the fixtures: S = A - B' F^-1 B, solve the arm, back-substitute.
        double* const fi0 = &s.con.cw[0][0];             // F^-1 (f, f), F^-1 (f, pair f), F^-1 g_f, B_f . x_arm: the edge weights are dead until the next iteration
        double* const fi1 = fi0 + NV;
        double* const fy = fi1 + NV;
        double* const ft = fy + NV;
        static_assert(4 * NV <= MC * 8, "scratch vectors fit the edge-weight block");
        if (isl && l >= NA) {
          const int p = m.pair[l];
          const double d = s.con.Hc.lo(l, l) + s.con.dl[l], gl = s.con.rc[l] + s.con.rl[l];
          double i0, i1 = 0.0, y;
          // (y = F^-1 g in the SAME expressions as the uncoupled path below: an env of this wave whose own fingers touch nothing -- its B is zero -- then gets the bits
          // it would get in a wave without any coupling contact; its result must not depend on which env shares its wave.  Round 5: the small-batch launches pair envs differently)
          if (p >= 0) {
            const int hi = l > p ? l : p, lo_ = l > p ? p : l;
            const double o = s.con.Hc.lo(hi, lo_), dp = s.con.Hc.lo(p, p) + s.con.dl[p], idet = rcp_nr(d * dp - o * o);
            i0 = dp * idet; i1 = -o * idet;
            y = (dp * gl - o * (s.con.rc[p] + s.con.rl[p])) * idet;
          } else {
            i0 = rcp_nr(d);
            y = gl * i0;
          }
          fi0[l] = i0; fi1[l] = i1; fy[l] = y;
        }
        fence();
        KSTAMP(13);
        if (isl && l < NA) {                              // row l of the Schur complement and of its right-hand side, in place
          double row[NA], g = s.con.rc[l] + s.con.rl[l];
          bool touched = false;                           // (this env has a contact between the arm and a fixture)
#pragma unroll
          for (int c = 0; c < NA; ++c) { const double hv = s.con.Hc.lo(l, c <= l ? c : l); row[c] = c <= l ? hv : 0.0; }      // (loads without a branch per entry)
          double blv[NV - NA];                            // (all of column l of B first: one LDS latency, not one per fixture)
#pragma unroll
          for (int f = NA; f < NV; ++f) blv[f - NA] = s.con.Hc.lo(f, l);
          // only the fixtures a finger touches have a row in B: the set of them over the wave's arm lanes (uniform), one pass of the loop per fixture of the set,
          // ascending -- a lane whose own entry is zero subtracts exact zeros.  (Until round 5: fourteen unrolled tests, each body under its own branch with its loads
          // and waits inside; 68 LDS round trips one after the other in the listing.)
          unsigned int tset = 0;
#pragma unroll
          for (int f = NA; f < NV; ++f) tset |= __ballot(blv[f - NA] != 0.0) ? (1u << (f - NA)) : 0u;
          for (unsigned int r = tset; r; r &= r - 1u) {
            const int f = NA + __builtin_ctz(r);
            int p = m.pair[f];
            double fv[4] = {s.con.Hc.lo(f, l), fi0[f], fi1[f], fy[f]}, hf[NA], hp[NA];
#pragma unroll
            for (int c = 0; c < NA; ++c) hf[c] = s.con.Hc.lo(f, c);
            asm volatile("" : "+v"(p));
            pin_batch(fv); pin_batch(hf);
            const int pc = p >= 0 ? p : f;
#pragma unroll
            for (int c = 0; c < NA; ++c) hp[c] = s.con.Hc.lo(pc, c);
            pin_batch(hp);
            const double bl = fv[0];
            touched = touched || bl != 0.0;
            const double w0 = bl * fv[1], w1 = bl * fv[2];
            g = fma(-bl, fv[3], g);
#pragma unroll
            for (int c = 0; c < NA; ++c) {
              const double t0 = w0 * hf[c], t1 = fma(w1, hp[c], t0);
              const double t = p >= 0 ? t1 : t0;
              row[c] -= c <= l ? t : 0.0;
            }
          }
#pragma unroll
          for (int c = 0; c < NA; ++c) *(c <= l ? &s.con.Hc.lo(l, c <= l ? c : l) : reinterpret_cast<double*>(s.bank_pad)) = row[c];      // (... and stores)
          if (touched) s.con.rc[l] = g - s.con.rl[l];     // (a[] below is formed as rc + rl again; an untouched row keeps its rc: (rc + rl) - rl + rl is not rc + rl in floating point)
        }
        fence();
        KSTAMP(14);
#pragma unroll
        for (int i = 0; i < NA; ++i) a[i] = s.con.rc[i] + s.con.rl[i];
        // (the Schur complement is factorised and solved in registers, redundantly per lane, like the arm's block without contacts: the lane-cooperative
        // in-LDS form -- chol_coop_lead + solve_lds_lead, nine plus eighteen dependent LDS round trips -- was a third of this path)
        solve_lead_regs<NV, NA>(s.con.Hc, [&](int i) { return s.con.dl[i]; }, a);
        KSTAMP(15);
        if (isl && l >= NA) {                             // t_f = B_f . x_arm
          double t = 0;
#pragma unroll
          for (int c = 0; c < NA; ++c) t = fma(s.con.Hc.lo(l, c), a[c], t);
          ft[l] = t;
        }
        fence();
        if (isl && l >= NA) {
          const int p = m.pair[l];
          s.con.rc[l] = fy[l] - fi0[l] * ft[l] - (p >= 0 ? fi1[l] * ft[p] : 0.0);
        }
        fence();
#pragma unroll
        for (int i = NA; i < NV; ++i) a[i] = s.con.rc[i];
      } else {
        // no contact joins the arm and the fixtures: the Hessian is the arm's NA x NA block plus, per fixture, a scalar or -- for the
        // knob / burner and switch / light couplings -- a 2 x 2 block with its partner (earl_link_model24.pair)
        const SymLds<NV>& Hs = ncmax == 0 ? s.hwst.Hw : s.con.Hc;
        {
          // the fixtures' scalars / 2 x 2 blocks: no branch (an arm lane works on fixture NA and stores into the block's padding), so that this chain of LDS round
          // trips and a reciprocal overlaps the arm block's factorisation below instead of preceding it
          const int lf = l >= NA ? l : NA;
          const int p = m.pair[lf], pc = p >= 0 ? p : lf;
          const double d = Hs.lo(lf, lf) + s.con.dl[lf];
          const int hi = lf > pc ? lf : pc, lo_ = lf > pc ? pc : lf;
          const double o = Hs.lo(hi, lo_), dp = Hs.lo(pc, pc) + s.con.dl[pc];
          double gl = 0, gp = 0;
#pragma unroll
          for (int i = NA; i < NV; ++i) { gl = i == lf ? a[i] : gl; gp = i == p ? a[i] : gp; }
          const double r = rcp_nr(p >= 0 ? d * dp - o * o : d);
          const double x = p >= 0 ? (dp * gl - o * gp) * r : gl * r;
          *((isl && l >= NA) ? &s.con.rc[l] : reinterpret_cast<double*>(s.bank_pad)) = x;      // (every lane already holds the right-hand side in a[])
        }
        solve_lead_regs<NV, NA>(Hs, [&](int i) { return s.con.dl[i]; }, a);
        fence();
#pragma unroll
        for (int i = NA; i < NV; ++i) a[i] = s.con.rc[i];
      }
    } else if constexpr (Lim<NV>::CONNECT) {
      // nv = 22, dense: the LOOPING forms of the lane-cooperative factorisation and substitution (row of L and right-hand side stay in LDS): fully
      // unrolled, chol_coop + solve_lds keep two 22-entry vectors in registers over 231 column steps and spilled 1.3 KB per lane into scratch
#if EARL_MT_LOOP_SOLVER
      if (isl) s.con.rc[l] += s.con.rl[l];
      chol_coop_loop<NV>(s.con.Hc, s.con.dl, l, isl);
      solve_lds_loop<NV>(s.con.Hc, s.con.rc);
      fence();
#pragma unroll
      for (int i = 0; i < NV; ++i) a[i] = s.con.rc[i];
#else
      {
        const double xl = chol_solve_rows<NV>(s.con.Hc, s.con.dl, s.con.rc[l] + s.con.rl[l], l, isl, grp);
        fence();
        if (isl) s.con.rc[l] = xl;
        fence();
#pragma unroll
        for (int i = 0; i < NV; ++i) a[i] = s.con.rc[i];
      }
#endif
    } else if constexpr (Lim<NV>::COOP) {
      chol_coop<NV>(s.con.Hc, s.con.dl, l, isl);
      solve_lds<NV>(s.con.Hc, a);
    } else if constexpr (NA == NV) {                   // small model: dense, in registers
      load_tri<NV, NV>(L, s.con.Hc, [&](int i) { return s.con.dl[i]; });
      chol_regs<NV, NV, (NV > 10)>(L);
      solve_regs<NV, NV>(L, a);
    } else if (coupled) {                              // a contact joins the arm and the object (uniform over the wave): the object's block eliminated first
#ifdef EARL_PEG_COUPLED_LDS                            // (measurement switch: round 3's shared dense factorisation in LDS)
      chol_coop<NV>(s.con.Hc, s.con.dl, l, isl);
      solve_lds<NV>(s.con.Hc, a);
#else
      solve_schur_regs<NV, NA>(s.con.Hc, [&](int i) { return s.con.dl[i]; }, a);
#endif
    } else {
      load_tri<NV, NA>(L, s.con.Hc, [&](int i) { return s.con.dl[i]; });
      chol_regs<NV, NA, (NV > 10)>(L);
      solve_regs<NV, NA>(L, a);
    }
    KSTAMP(18);
    double al = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) al = l == i ? a[i] : al;
    const bool want = lim_inst && ((lim_lo ? al : -al) - lim_aref < 0);
    bool changed = want != act;
    act = frozen ? act : want;
    if constexpr (Lim<NV>::EXTRAS) {
      if (fr_loss > 0) {
        const double x = al - fr_aref;
        const int ns = fabs(x) * fr_D <= fr_loss ? 0 : (x > 0 ? 1 : -1);
        changed = changed || ns != fr_state;
        fr_state = ns;
      }
    }
    if (ncmax > 0) {
      const int c = sub < MC ? sub : MC - 1;
      double an = 0, at1 = 0, at2 = 0;
      if constexpr (TS < NT) {
        if (ctA) {                                        // (a tree's entries of the three rows as one batch of loads: physics_math.h pin_batch)
          double cj[3 * TS];
#pragma unroll
          for (int j = 0; j < TS; ++j) { cj[3 * j] = s.con.CJ[c][0][j]; cj[3 * j + 1] = s.con.CJ[c][1][j]; cj[3 * j + 2] = s.con.CJ[c][2][j]; }
          pin_batch(cj);
#pragma unroll
          for (int j = 0; j < TS; ++j) { an = fma(cj[3 * j], a[j], an); at1 = fma(cj[3 * j + 1], a[j], at1); at2 = fma(cj[3 * j + 2], a[j], at2); }
        }
        if (ctP) {
          double cj[3 * (NV - TS)];
#pragma unroll
          for (int j = TS; j < NV; ++j) { cj[3 * (j - TS)] = s.con.CJ[c][0][j]; cj[3 * (j - TS) + 1] = s.con.CJ[c][1][j]; cj[3 * (j - TS) + 2] = s.con.CJ[c][2][j]; }
          pin_batch(cj);
#pragma unroll
          for (int j = TS; j < NV; ++j) { an = fma(cj[3 * (j - TS)], a[j], an); at1 = fma(cj[3 * (j - TS) + 1], a[j], at1); at2 = fma(cj[3 * (j - TS) + 2], a[j], at2); }
        }
      } else if constexpr (Lim<NV>::EXTRAS) {
        // (a row of the contact's Jacobian as one batch of loads -- physics_math.h pin_batch: left to the scheduler the 69 loads came one or two at a time, 40 LDS
        // round trips one after the other; the three sums keep their order of additions)
        double row[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) row[j] = s.con.CJ[c][0][j];
        pin_batch(row);
#pragma unroll
        for (int j = 0; j < NV; ++j) an = fma(row[j], a[j], an);
#pragma unroll
        for (int j = 0; j < NV; ++j) row[j] = s.con.CJ[c][1][j];
        pin_batch(row);
#pragma unroll
        for (int j = 0; j < NV; ++j) at1 = fma(row[j], a[j], at1);
#pragma unroll
        for (int j = 0; j < NV; ++j) row[j] = s.con.CJ[c][2][j];
        pin_batch(row);
#pragma unroll
        for (int j = 0; j < NV; ++j) at2 = fma(row[j], a[j], at2);
      } else {
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          an = fma(s.con.CJ[c][0][j], a[j], an); at1 = fma(s.con.CJ[c][1][j], a[j], at1); at2 = fma(s.con.CJ[c][2][j], a[j], at2);
        }
      }
      unsigned int nb = 0;
      if constexpr (Lim<NV>::ELLIPTIC) {
        // the contact's zone at the new iterate; a sliding contact is linearised again until its row values stand still (LinkModel.ELL_TOL)
        nb = sub < nct ? (unsigned int)cone_zone(an - car[0], at1 - car[1], at2 - car[2], cmu) : 0u;
        const double big = fmax(fmax(fabs(cja[0]), fabs(cja[1])), fabs(cja[2])), dif = fmax(fmax(fabs(an - cja[0]), fabs(at1 - cja[1])), fabs(at2 - cja[2]));
        changed = changed || nb != cact || (nb == 2u && dif > 1e-8 * (1.0 + big));
        if (!frozen) { cja[0] = an; cja[1] = at1; cja[2] = at2; }
        nb = frozen ? cact : nb;
      } else {
      nb |= (an + cmu * at1 - car[0] < 0) ? 1u : 0u;
      nb |= (an - cmu * at1 - car[1] < 0) ? 2u : 0u;
      nb |= (an + cmu * at2 - car[2] < 0) ? 4u : 0u;
      nb |= (an - cmu * at2 - car[3] < 0) ? 8u : 0u;
      nb = sub < nct ? nb : 0u;
      changed = changed || nb != cact;
      }
      cact = nb;
    }
    if constexpr (Lim<NV>::ELLIPTIC) {
      // A sliding contact is a Newton iteration stopped at a tolerance: one more pass would move the solution in its last digits.  So an env stops at ITS OWN fixed point
      // (pyramid models reach theirs exactly: more passes for a wave-mate's sake rebuild the same Hessian from the same set and change no bit)
      const bool env_changed = group_any<LPE>(changed, grp);
      if (!frozen && !env_changed) {
        if (isl) s.aprev[l] = al;
        frozen = true;
      }
      changed = !frozen;
    }
    fence();
    KSTAMP(19);
    if (!__any(changed)) break;
  }
  if constexpr (Lim<NV>::ELLIPTIC) {
    if (!frozen && isl) {                               // (the cap of eight passes: the last iterate stands)
      double al = 0;
#pragma unroll
      for (int i = 0; i < NV; ++i) al = l == i ? a[i] : al;
      s.aprev[l] = al;
    }
    fence();
#pragma unroll
    for (int i = 0; i < NV; ++i) a[i] = s.aprev[i];
  }
  if constexpr (INTEGRATE) {
    double al = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) al = l == i ? a[i] : al;
    if (isl) s.aprev[l] = al;                           // (read after the fences of the next timestep)
  }
#ifdef EARL_PHYS_PROF
  {
    const unsigned long long dk = __builtin_readcyclecounter() - k9_t0;
    PCOUNT(coupled ? 27 : 28, dk);
#ifdef EARL_PHYS_PROF_ALL                                // (atomics of every wave: perturbs the clocks; for the counts only)
    PCOUNT_ALL(29, 1); PCOUNT_ALL(30, coupled ? 1 : 0); PCOUNT_ALL(31, coupled ? dk : 0);
#endif
  }
#endif
  PSTAMP(8);
  if constexpr (!INTEGRATE) {
    double al = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) al = l == i ? a[i] : al;
    if (qacc_out && isl) qacc_out[l] = al;
    if (efc_out) {
      // the weld Jacobian shares its storage with the Hessian: put this lane's column back before the rows are read
      fence();
      if (isl) {
#pragma unroll
        for (int r = 0; r < 6; ++r) s.con.J6[r][l] = Jc[r];
      }
      fence();
      if (sub < 6) {
        double Ja = 0;
#pragma unroll
        for (int j = 0; j < NV; ++j) Ja = fma(s.con.J6[sub][j], a[j], Ja);
        efc_out[sub] = -s.con.wD[sub] * (Ja - s.con.war[sub]);
      }
      if (isl) {
        const double f = act ? -lim_D * ((lim_lo ? al : -al) - lim_aref) : 0.0;
        efc_out[6 + 2 * l] = lim_lo ? f : 0.0;
        efc_out[7 + 2 * l] = lim_lo ? 0.0 : f;
      }
    }
  } else {
    // ---------------------------------------------------------------- K10: Euler, joint damping implicit
    if constexpr (Lim<NV>::DAMPED) {
    {
      double acc = 0;
      if constexpr (Lim<NV>::EXTRAS) {                   // (this lane's column of M as one batch of loads: physics_math.h pin_batch; the peg build measured 2 % slower with it)
        double mc[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) mc[j] = s.M.sym(j, l, ltri);
        pin_batch(mc);
#pragma unroll
        for (int j = 0; j < NV; ++j) acc = fma(mc[j], a[j], acc);
      } else {
#pragma unroll
      for (int j = 0; j < NV; ++j) acc = fma(s.M.sym(j, l, ltri), a[j], acc);
      }
      if (isl) s.con.rc[l] = acc;
    }
    fence();
#pragma unroll
    for (int i = 0; i < NV; ++i) a[i] = s.con.rc[i];
    if constexpr (Lim<NV>::EXTRAS) pin_batch(a);
    if constexpr (R_OWNER) {
      // split timestep: wave A factorised the arm's block and inverted the fixtures' scalars while this wave iterated (barrier Z: they are in its block, `peer`)
      constexpr int NL = NA * (NA + 1) / 2;
      __syncthreads();                                 // barrier Z
      const double* const kf = peer->hwst.Hw.v;
      double Lk[NL], y[NA];
#pragma unroll
      for (int e = 0; e < NL; ++e) Lk[e] = kf[e];
      const int lf = l >= NA ? l : NA;
      const double ki = kf[NL + lf - NA];
      pin_batch(Lk);
      double gl = 0;
#pragma unroll
      for (int i = NA; i < NV; ++i) gl = i == lf ? a[i] : gl;
      *((isl && l >= NA) ? &s.con.rc[l] : reinterpret_cast<double*>(s.bank_pad)) = gl * ki;
#pragma unroll
      for (int i = 0; i < NA; ++i) y[i] = a[i];
      solve_regs<NA, NA>(Lk, y);
#pragma unroll
      for (int i = 0; i < NA; ++i) a[i] = y[i];
      fence();
#pragma unroll
      for (int i = NA; i < NV; ++i) a[i] = s.con.rc[i];
    } else {
    // the implicit-damping diagonal dt * B goes through LDS in both forms: as an operand of the add in load_tri the product would be contracted
    // into an fma, in chol_coop it is a rounded product -- the two door builds must agree to the bit
    if (isl) s.con.dl[l] = dt * m.damping[l];
    fence();
    if constexpr (Lim<NV>::EXTRAS) {                   // the mass matrix is ALWAYS the arm's block + one scalar per fixture
      {
        const int lf = l >= NA ? l : NA;                 // (no branch: see K9's fixtures)
        double gl = 0;
#pragma unroll
        for (int i = NA; i < NV; ++i) gl = i == lf ? a[i] : gl;
        *((isl && l >= NA) ? &s.con.rc[l] : reinterpret_cast<double*>(s.bank_pad)) = gl * rcp_nr(s.M.lo(lf, lf) + s.con.dl[lf]);
      }
      solve_lead_regs<NV, NA>(s.M, [&](int i) { return s.con.dl[i]; }, a);
      fence();
#pragma unroll
      for (int i = NA; i < NV; ++i) a[i] = s.con.rc[i];
    } else if constexpr (Lim<NV>::COOP) {              // M is rebuilt next timestep: factorise it in place
      chol_coop<NV>(s.M, s.con.dl, l, isl);
      solve_lds<NV>(s.M, a);
    } else {
      load_tri<NV, NA>(L, s.M, [&](int i) { return s.con.dl[i]; });           // the mass matrix is block diagonal: two trees
      chol_regs<NV, NA, (NV > 10)>(L);
      solve_regs<NV, NA>(L, a);
    }
    }                                                  // (ROLE != 2)
    }
    double al = 0;
#pragma unroll
    for (int i = 0; i < NV; ++i) al = l == i ? a[i] : al;
    fence();
    if (isl) {
      const double nv_ = s.qv[l] + dt * al;
      s.qv[l] = nv_;
      s.qp[l] = s.qp[l] + dt * nv_;                    // (unused for the rotation dofs of a free body)
    }
    fence();
    if (m.ball_dof >= 0) {
      // mju_quatIntegrate: q <- normalize(q) * quat(axis = w / |w|, angle = dt |w|), w = angular velocity in body axes; every lane
      // computes it, lane 0 stores
      const int bd = m.ball_dof;
      const V3 wb{s.qv[bd], s.qv[bd + 1], s.qv[bd + 2]};
      Q4 q0 = ldq(s.bq);
      const double n0 = rsq_nr(q0.w * q0.w + q0.x * q0.x + q0.y * q0.y + q0.z * q0.z);
      q0 = Q4{q0.w * n0, q0.x * n0, q0.y * n0, q0.z * n0};
      const double w2 = dot(wb, wb);
      const double iw = w2 > 0 ? rsq_nr(w2 > 0 ? w2 : 1.0) : 0.0;
      double sn, cs;
      sincos_mod(0.5 * dt * (w2 * iw), sn, cs);
      Q4 q1 = qmul(q0, Q4{cs, sn * wb.x * iw, sn * wb.y * iw, sn * wb.z * iw});
      const double n1 = rsq_nr(q1.w * q1.w + q1.x * q1.x + q1.y * q1.y + q1.z * q1.z);
      fence();
      if (sub == 0) { s.bq[0] = q1.w * n1; s.bq[1] = q1.x * n1; s.bq[2] = q1.y * n1; s.bq[3] = q1.z * n1; }
      fence();
    }
    PSTAMP(11);
  }
}

// world position of attachment k from the kinematics currently in LDS
template <int NV>
__device__ __forceinline__ V3 attachment(const Shared<NV>& s, const typename ModelOf<NV>::T& m, const int k) {
  const int la = m.att_link[k];
  V3 p = ld3(m.att_pos[k]);
  if (la >= 0) {
    double R[3][3];
    qmat(ldq(s.Xq[la]), R);
    p = add(ld3(s.Xp[la]), mulv(R, p));
  }
  return p;
}

// the model tables, once per workgroup, into LDS (all 64 lanes copy)
template <typename MT>
__device__ __forceinline__ void stage_model(MT& dst, const void* __restrict__ src) {
  static_assert(sizeof(MT) % 8 == 0, "copied as 8-byte words");
  const unsigned long long* g = reinterpret_cast<const unsigned long long*>(src);
  unsigned long long* d = reinterpret_cast<unsigned long long*>(&dst);
  for (int i = threadIdx.x; i < (int)(sizeof(MT) / 8); i += blockDim.x) d[i] = g[i];
  __syncthreads();                                     // the only workgroup barrier: afterwards every wave works on its own LDS blocks
}

// state rows <-> LDS.  qpos rows are [nq]: one entry per dof, except that the free body's orientation quaternion sits at
// [ball_dof, ball_dof + 4) (normalised on load, as mj_kinematics does)
template <int NV, typename SH>
__device__ __forceinline__ void load_state(SH& s, const typename ModelOf<NV>::T& m, const double* __restrict__ qrow, const double* __restrict__ vrow, const int sub) {
  const int bd = m.ball_dof;
  if (sub < NV) {
    // (a free ROOT body -- the minitaur's base, ball_dof = 3 -- keeps MuJoCo's layout [xyz, quaternion, joints]: dof l > bd + 2 sits at qrow[l + 1])
    s.qp[sub] = (bd < 0 || sub < bd) ? qrow[sub] : (sub > bd + 2 ? qrow[sub + 1] : 0.0);
    s.qv[sub] = vrow[sub];
    s.aprev[sub] = 0.0;              // (read, and discarded, by the cold first timestep)
  }
  if (sub < 4) {
    double v = sub == 0 ? 1.0 : 0.0;
    if (bd >= 0) {
      const Q4 q = ldq(qrow + bd);
      v = qrow[bd + sub] * rsq_nr(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
    }
    s.bq[sub] = v;
  }
}
// the orientation quaternion in LDS as load_state would read it back from a stored row (the same expression, compiled under the same contraction mode):
// lets a fused rollout walk through the same bits as one launch per env step
template <int NV, typename SH>
__device__ __forceinline__ double renormalised_quat_entry(const SH& s, const int sub) {
  const Q4 q = ldq(s.bq);
  return s.bq[sub < 4 ? sub : 0] * rsq_nr(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
}
template <int NV, typename SH>
__device__ __forceinline__ void store_state(const SH& s, const typename ModelOf<NV>::T& m, double* __restrict__ qrow, double* __restrict__ vrow, const int sub) {
  const int bd = m.ball_dof;
  if (sub < NV) {
    if (bd < 0 || sub < bd) qrow[sub] = s.qp[sub];
    else if (sub > bd + 2) qrow[sub + 1] = s.qp[sub];
    vrow[sub] = s.qv[sub];
  }
  if (bd >= 0 && sub < 4) qrow[bd + sub] = s.bq[sub];
}

struct PArgs {
  const void* m;                 // earl_link_model (nv <= 16) or earl_link_model24
  const earl_collision_model* col;
  int n, nsub;
  double* qpos; double* qvel;
  const double* mocap_pos; const double* mocap_quat; const double* ctrl;
  double* att_xpos; double* qacc_out; double* efc_out;
  int ctrl_stride;               // doubles per env in `ctrl` (0: n_act; the kitchen hands over its nine position targets, of which the first n_act = 2 count)
  int mq_stride;                 // doubles per env in `mocap_quat` (0: ONE quaternion for the whole batch; else 4)
};

template <int NV, int LPE, bool INTEGRATE>
__global__ __launch_bounds__(64 * Lim<NV>::WPB) void physics_kernel(const PArgs a) {
  constexpr int EPW = 64 / LPE, WPB = Lim<NV>::WPB;
  __shared__ alignas(16) typename ModelOf<NV>::T m;
  __shared__ alignas(16) BlkTable<Lim<NV>::MB, Lim<NV>::KBT> bt;
  __shared__ alignas(16) Shared<NV> sh[EPW * WPB];
  stage_blocks(bt, a.col);
  stage_kb<NV>(bt, a.m, a.col);
  stage_model(m, a.m);
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), sub = lane % LPE, grp = lane / LPE;
  const int env_raw = (blockIdx.x * WPB + wave) * EPW + grp;
  const bool live = env_raw < a.n;
  const int env = live ? env_raw : a.n - 1;            // idle groups shadow the last env and store nothing
  Shared<NV>& s = sh[wave * EPW + grp];
  load_state<NV>(s, m, a.qpos + (size_t)env * m.nq, a.qvel + (size_t)env * NV, sub);
  if constexpr (Lim<NV>::NT < NV || Lim<NV>::TS < Lim<NV>::NT) {   // the mass-matrix entries between different trees are never written (K5): zero, once
    for (int k = sub; k < (int)(sizeof(s.M.v) / sizeof(double)); k += LPE) s.M.v[k] = 0.0;
  }
  if constexpr (Lim<NV>::EXTRAS) {                       // ... and so are the structural zeros of the equality Hessian (K9)
    for (int k = sub; k < (int)(sizeof(s.hwst.Hw.v) / sizeof(double)); k += LPE) s.hwst.Hw.v[k] = 0.0;
  }
  if (sub < 3) s.mocap[sub] = a.mocap_pos[(size_t)env * 3 + sub];
  fence();
  const Q4 mq = ldq(a.mocap_quat + (size_t)env * a.mq_stride);      // as given, NOT normalised (include/earl_physics.h)
  double ctrl[EARL_MAXACT] = {0, 0, 0, 0};
  for (int ac = 0; ac < m.n_act; ++ac) ctrl[ac] = a.ctrl[(size_t)env * (a.ctrl_stride ? a.ctrl_stride : m.n_act) + ac];
  constexpr int NC = 6 + 2 * NV;
  for (int ts = 0; ts < a.nsub; ++ts)
    substep<NV, LPE, INTEGRATE>(s, m, bt, a.col, sub, grp, mq, ctrl, INTEGRATE && ts > 0, (a.qacc_out && live) ? a.qacc_out + (size_t)env * NV : nullptr,
                                (a.efc_out && live) ? a.efc_out + (size_t)env * NC : nullptr);
  if constexpr (INTEGRATE) {
    if (live) store_state<NV>(s, m, a.qpos + (size_t)env * m.nq, a.qvel + (size_t)env * NV, sub);
  }
  // attachments at the kinematics of the LAST timestep's start (what mj_step leaves in data.xpos / site_xpos)
  if (a.att_xpos && sub < m.n_att && live) {
    const V3 p = attachment<NV>(s, m, sub);
    double* o = a.att_xpos + ((size_t)env * m.n_att + sub) * 3;
    o[0] = p.x; o[1] = p.y; o[2] = p.z;
  }
}

template <int LPE>
__device__ __forceinline__ bool group_any(const bool pred, const int grp) {
  const unsigned long long bal = __ballot(pred);
  if constexpr (LPE == 64) return bal != 0ull;
  else return ((bal >> (grp * (LPE & 63))) & ((1ull << (LPE & 63)) - 1ull)) != 0ull;
}

#if !defined(EARL_PHYS_VARIANT_MT) && !defined(EARL_PHYS_UNIT_KITCHEN)      // (the minitaur and kitchen units hold no Sawyer kernel)
#include "physics_env_sawyer.h"
#endif

#ifdef EARL_PHYS_UNIT_KITCHEN
#include "physics_env_kitchen.h"
#endif

#ifdef EARL_PHYS_VARIANT_MT
#include "minitaur_stepper.h"
#include "physics_env_minitaur.h"
#endif   // EARL_PHYS_VARIANT_MT


int launched(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "earl_physics: %s: %s\n", what, hipGetErrorString(e));
    return EARL_ERR_LAUNCH;
  }
  return EARL_OK;
}

// The kernels compile the friction cone per model size (Lim<NV>::ELLIPTIC) and the two cones lay the contact arrays out differently, so a collision table
// compiled for the other cone must be refused (include/earl_physics.h: earl_collision_model.cone).  `col` is a DEVICE table: its cone word is copied to the
// host once per device address (4 bytes, a blocking copy -- never while the stream is being captured into a graph: the check is then skipped for an address
// not yet seen) and remembered in ONE cache for all translation units (earl_unit_table_cone, main unit) until its owner announces that the block is gone:
// earl_physics_forget_table (the Python front end calls it when a DeviceModel is freed -- a caching allocator hands the block to the next table, which may be of
// the other cone: ADVICE r05).
int check_cone(const earl_collision_model* col, const bool want_elliptic, hipStream_t st, const char* what) {
  if (!col) return EARL_OK;
  const int cone = earl_unit_table_cone(col, st);
  if (cone < 0) return EARL_OK;            // (unknown: a capture in progress, or not a device table -- the launch reports that)
  if (cone != (want_elliptic ? 1 : 0)) {
    fprintf(stderr, "earl_physics: %s: the collision table is compiled for the %s friction cone, this model size runs the %s one (earl_collision_model.cone)\n", what,
            cone == 1 ? "elliptic" : "pyramidal", want_elliptic ? "elliptic" : "pyramidal");
    return EARL_ERR_ARG;
  }
  return EARL_OK;
}

#ifdef EARL_PHYS_VARIANT_MT
int g_mt_stepper = 1;     // earl_debug_set_minitaur_stepper: 1 = the tree-structured timestep (minitaur_stepper.h), 0 = the generic substep<22>
int g_mt_duo = EARL_MT_DUO_DEFAULT;   // earl_debug_set_minitaur_duo: 1 = the two-waves-per-SIMD rollout (minitaur_duo_kernel), 0 = the one-wave kernel, -1 = by batch size
#endif
#ifndef EARL_PEG_SLICE
#define EARL_PEG_SLICE 10   // env steps per work item (tools/bench_peg_schedule.py: 5 and 10: 43.1 ms, 20: 44.3, 40: 47.0, 100: 50.9, one group per wave: 54.8) of the peg's time-sliced schedule (a slice is ~1 ms; claiming one costs a scan of the queue: microseconds)
#endif
int g_peg_sliced = 1;     // earl_debug_set_peg_schedule
int g_door_variant = 0;   // earl_debug_set_door_variant: 0 = by batch size, 1 = four single-wave workgroups per CU, 2 = one eight-wave workgroup per CU
int g_lpe = 16;   // lanes per env (earl_debug_set_physics_lanes): 16 = four envs per wavefront, 64 = one wavefront per env

// Small batches of the 32-lanes-per-env kernels (kitchen, minitaur): an env is a serial chain of T x frame_skip timesteps walked by one wave, so a batch that leaves
// wave slots empty gains nothing from them -- except by giving every env a wave (solo 1: up to 4 x CUs envs) or a whole CU (solo 2: up to CUs envs) to itself.
// earl_debug_set_solo: -1 = by batch size (default), 0 / 1 / 2 = forced (measurement, tests); the kitchen's fused rollout runs its one-env-per-workgroup launches with
// all FOUR waves on the env (solo 3: by default, or forced; 2 forces the one-wave form)
int g_solo = -1;
int solo_mode(int n) {
  if (g_solo >= 0) return g_solo;
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  return n <= cus ? 2 : (n <= 4 * cus ? 1 : 0);
}
// minitaur rollout, packed launches: the one-wave kernel holds 8 envs per CU, the two-wave kernel 16 at 1.6 x the time per round (measured: 4096 x 1000 in 174 ms = two rounds of
// 87 against one round of 139): whichever needs less time for the batch's rounds
bool mt_use_duo(int n) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  const long rounds_one = (n + 8L * cus - 1) / (8L * cus), rounds_two = (n + 16L * cus - 1) / (16L * cus);
  return 16 * rounds_two < 10 * rounds_one;
}
int solo_grid(int n, int solo, int wpb) { return solo == 2 ? n : (solo == 1 ? (n + wpb - 1) / wpb : (n + 2 * wpb - 1) / (2 * wpb)); }

// launch geometry: Lim<NV>::WPB wavefronts per workgroup, 64 / LPE envs per wavefront
template <int NV, int LPE> int grid_for(int n) { constexpr int epb = (64 / LPE) * Lim<NV>::WPB; return (n + epb - 1) / epb; }
template <int NV> constexpr int block_for() { return 64 * Lim<NV>::WPB; }

}  // namespace
// the 64-lanes-per-env instantiations live in physics_l64.hip (argument structs by address: same source, same layout in both units); not exported from the library
extern "C" __attribute__((visibility("hidden"))) void earl_unit_l64_physics(const void* pargs, int nv, int integrate, void* stream);
extern "C" __attribute__((visibility("hidden"))) void earl_unit_l64_sawyer_rollout(const void* sawyer_args, int nv, void* stream);
extern "C" __attribute__((visibility("hidden"))) void earl_unit_kitchen_physics(const void* pargs, int integrate, void* stream);      // physics_kitchen.hip: nv = 23
namespace {

#ifndef EARL_PHYS_UNIT_L64
template <int NV, bool INTEGRATE>
void launch_physics(const PArgs& a, hipStream_t st) {
  if constexpr (NV > 16) {
#ifdef EARL_PHYS_UNIT_KITCHEN
    physics_kernel<NV, 32, INTEGRATE><<<grid_for<NV, 32>(a.n), block_for<NV>(), 0, st>>>(a);   // 32 lanes per env: two envs per wave
#else
    earl_unit_kitchen_physics(&a, INTEGRATE ? 1 : 0, st);                                       // (physics_kitchen.hip)
#endif
  }
  else if (g_lpe == 64) earl_unit_l64_physics(&a, NV, INTEGRATE ? 1 : 0, st);
#ifndef EARL_PHYS_UNIT_KITCHEN
  else physics_kernel<NV, 16, INTEGRATE><<<grid_for<NV, 16>(a.n), block_for<NV>(), 0, st>>>(a);
#endif
}
#endif

}  // namespace

#ifndef EARL_PHYS_NOT_MAIN
namespace {
std::mutex g_cone_mu;
std::unordered_map<const void*, int> g_cone_seen;
}  // namespace
extern "C" __attribute__((visibility("hidden"))) int earl_unit_table_cone(const void* col, void* stream) {
  std::lock_guard<std::mutex> lock(g_cone_mu);
  const auto it = g_cone_seen.find(col);
  if (it != g_cone_seen.end()) return it->second;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing((hipStream_t)stream, &cs) != hipSuccess) { (void)hipGetLastError(); return -1; }
  if (cs != hipStreamCaptureStatusNone) return -1;
  int cone = -1;
  if (hipMemcpy(&cone, &static_cast<const earl_collision_model*>(col)->cone, sizeof cone, hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); return -1; }
  g_cone_seen[col] = cone;
  return cone;
}
#endif

extern "C" {

#ifndef EARL_PHYS_NOT_MAIN
// include/earl_physics.h: the owner of a device collision table announces that the block is freed / rewritten (NULL: every table)
int earl_physics_forget_table(const void* col) {
  std::lock_guard<std::mutex> lock(g_cone_mu);
  if (!col) { const int k = (int)g_cone_seen.size(); g_cone_seen.clear(); return k; }
  return (int)g_cone_seen.erase(col);
}
int earl_physics_step(const void* model, const earl_collision_model* col, int32_t nv, int32_t n, int32_t nsub, double* qpos, double* qvel,
                      const double* mocap_pos, const double* mocap_quat, const double* ctrl, double* att_xpos,
                      earl_stream_t stream) {
  if (!model || n < 0 || nsub < 0 || !qpos || !qvel || !mocap_pos || !mocap_quat || !ctrl) return EARL_ERR_ARG;
  if (n == 0 || nsub == 0) return EARL_OK;
  if (nv != 10 && nv != 15 && nv != 23) return EARL_ERR_ARG;
  if (int rc = check_cone(col, nv <= 16, (hipStream_t)stream, "physics_step")) return rc;
  PArgs a{model, col, n, nsub, qpos, qvel, mocap_pos, mocap_quat, ctrl, att_xpos, nullptr, nullptr, 0, 4};
  if (nv == 10) launch_physics<10, true>(a, (hipStream_t)stream);
  else if (nv == 15) launch_physics<15, true>(a, (hipStream_t)stream);
  else if (nv == 23) launch_physics<23, true>(a, (hipStream_t)stream);
  else return EARL_ERR_ARG;
  return launched("physics_step");
}

int earl_physics_forward(const void* model, const earl_collision_model* col, int32_t nv, int32_t n, const double* qpos, const double* qvel,
                         const double* mocap_pos, const double* mocap_quat, const double* ctrl, double* qacc,
                         double* efc_force, double* att_xpos, earl_stream_t stream) {
  if (!model || n < 0 || !qpos || !qvel || !mocap_pos || !mocap_quat || !ctrl || !qacc) return EARL_ERR_ARG;
  if (n == 0) return EARL_OK;
  if (nv != 10 && nv != 15 && nv != 23) return EARL_ERR_ARG;
  if (int rc = check_cone(col, nv <= 16, (hipStream_t)stream, "physics_forward")) return rc;
  PArgs a{model, col, n, 1, const_cast<double*>(qpos), const_cast<double*>(qvel), mocap_pos, mocap_quat, ctrl, att_xpos, qacc, efc_force, 0, 4};
  if (nv == 10) launch_physics<10, false>(a, (hipStream_t)stream);
  else if (nv == 15) launch_physics<15, false>(a, (hipStream_t)stream);
  else if (nv == 23) launch_physics<23, false>(a, (hipStream_t)stream);
  else return EARL_ERR_ARG;
  return launched("physics_forward");
}

#endif
#if defined(EARL_PHYS_VARIANT_MT)
// This translation unit is physics_mt.hip: the minitaur's kernels (the nv = 22 instantiation of the stepper) and entry points.
int earl_minitaur_rollout(const void* model24, const earl_collision_model* col, const earl_minitaur_cfg* cfg, const earl_minitaur_state* st,
                          const float* action, int32_t T, const earl_minitaur_out* out, earl_stream_t stream) {
  if (!model24 || !cfg || !st || !out || !action || T < 0 || cfg->n < 0) return EARL_ERR_ARG;
  if (!st->qpos || !st->qvel || !st->goal || !st->motor_param || !st->observed_torque || !st->overheat || !st->motor_enabled) return EARL_ERR_ARG;
  if (!out->obs || !out->reward || !out->done || !out->success || !cfg->goal_table || cfg->n_goals < 1 || cfg->num_substeps < 0) return EARL_ERR_ARG;
  if (cfg->n == 0 || T == 0) return EARL_OK;
  if (int rc = check_cone(col, false, (hipStream_t)stream, "minitaur_rollout")) return rc;
  MinitaurArgs a{model24, col, *cfg, *st, *out, action, T, nullptr, nullptr, solo_mode(cfg->n)};
  // two waves per SIMD by role (minitaur_duo_kernel: 16 envs per workgroup of eight waves) for batches that fill the chip's wave slots in the packed form anyway
  if (g_mt_stepper && a.solo == 0 && cfg->num_substeps > 0 && (g_mt_duo > 0 || (g_mt_duo < 0 && mt_use_duo(cfg->n)))) {      // (num_substeps = 0: nothing to split)
    minitaur_duo_kernel<<<(unsigned)((cfg->n + 16 * MT_DUO_PAIRS / 4 - 1) / (4 * MT_DUO_PAIRS)), 128 * MT_DUO_PAIRS, 0, (hipStream_t)stream>>>(a);
    return launched("minitaur_rollout (two waves per SIMD)");
  }
  if (g_mt_stepper) minitaur_kernel<false, true><<<solo_grid(cfg->n, a.solo, EARL_MT_WPB), 64 * EARL_MT_WPB, 0, (hipStream_t)stream>>>(a);
  else minitaur_kernel<false, false><<<solo_grid(cfg->n, a.solo, Lim<22>::WPB), block_for<22>(), 0, (hipStream_t)stream>>>(a);
  return launched("minitaur_rollout");
}
int earl_minitaur_reset(const void* model24, const earl_collision_model* col, const earl_minitaur_cfg* cfg, const earl_minitaur_state* st,
                        const uint8_t* mask, double* obs, earl_stream_t stream) {
  if (!model24 || !cfg || !st || cfg->n < 0) return EARL_ERR_ARG;
  if (!st->qpos || !st->qvel || !st->goal || !st->motor_param || !st->observed_torque || !st->overheat || !st->motor_enabled) return EARL_ERR_ARG;
  if (!cfg->goal_table || !cfg->reset_qpos || cfg->n_goals < 1 || cfg->settle_steps < 0) return EARL_ERR_ARG;
  if (cfg->n == 0) return EARL_OK;
  if (int rc = check_cone(col, false, (hipStream_t)stream, "minitaur_reset")) return rc;
  MinitaurArgs a{model24, col, *cfg, *st, earl_minitaur_out{nullptr, nullptr, nullptr, nullptr, nullptr}, nullptr, 0, mask, obs, solo_mode(cfg->n)};
  if (g_mt_stepper) minitaur_kernel<true, true><<<solo_grid(cfg->n, a.solo, EARL_MT_WPB), 64 * EARL_MT_WPB, 0, (hipStream_t)stream>>>(a);
  else minitaur_kernel<true, false><<<solo_grid(cfg->n, a.solo, Lim<22>::WPB), block_for<22>(), 0, (hipStream_t)stream>>>(a);
  return launched("minitaur_reset");
}
int earl_minitaur_cfg_size(void) { return (int)sizeof(earl_minitaur_cfg); }
int earl_debug_set_solo_mt(int mode) {       // this unit's copy of the small-batch switch (earl_debug_set_solo): the minitaur launches
  const int prev = g_solo;
  if (mode >= -1 && mode <= 2) g_solo = mode;
  return prev;
}
int earl_debug_set_minitaur_stepper(int tree) {          // 1 (default): minitaur_stepper.h, 0: the generic substep<22> (comparison / measurement)
  if (tree != 0 && tree != 1) return EARL_ERR_ARG;
  g_mt_stepper = tree;
  return EARL_OK;
}
int earl_debug_set_minitaur_duo(int mode) {              // 1: the two-waves-per-SIMD rollout kernel for every packed launch, 0: never, -1: by batch size (default).  Returns the previous setting
  const int prev = g_mt_duo;
  if (mode >= -1 && mode <= 1) g_mt_duo = mode;
  return prev;
}
#ifdef EARL_MT_DEBUG
int earl_debug_read_mt_dbg(int* out_i, double* out_d) {
  if (hipMemcpyFromSymbol(out_i, HIP_SYMBOL(g_mt_dbg), sizeof(int) * 4096 * 8 * 32) != hipSuccess) return EARL_ERR_LAUNCH;
  if (hipMemcpyFromSymbol(out_d, HIP_SYMBOL(g_mt_dbg_al), sizeof(double) * 4096 * 8 * 32) != hipSuccess) return EARL_ERR_LAUNCH;
  if (hipMemcpyFromSymbol(out_d + 4096 * 8 * 32, HIP_SYMBOL(g_mt_dbg_x), sizeof(double) * 5 * 4096 * 8 * 32) != hipSuccess) return EARL_ERR_LAUNCH;
  return hipMemcpyFromSymbol(out_d + 6 * 4096 * 8 * 32, HIP_SYMBOL(g_mt_dbg_ph), sizeof(double) * 8 * 4096 * 8 * 32) == hipSuccess ? EARL_OK : EARL_ERR_LAUNCH;
}
#endif
#ifdef EARL_PHYS_PROF
int earl_debug_set_prof_wave_mt(int block, int thread) {     // this unit's copy of earl_debug_set_prof_wave (minitaur_duo_kernel: thread 0 = a first-half wave, thread 256 = its partner)
  const int v[2] = {block, thread};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_prof_sel), v, sizeof(v)) == hipSuccess ? EARL_OK : EARL_ERR_LAUNCH;
}
int earl_debug_read_wave_cycles_mt(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_cycles), sizeof(unsigned long long) * 4096) == hipSuccess ? EARL_OK : EARL_ERR_LAUNCH;
}
int earl_debug_read_phys_profile_mt(unsigned long long* out, int reset) {          // this unit's own copy of the phase counters (tools/prof_minitaur.py)
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phys_prof), sizeof(unsigned long long) * 32) != hipSuccess) return EARL_ERR_LAUNCH;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phys_prof), z, sizeof(z)) != hipSuccess) return EARL_ERR_LAUNCH;
  }
  return EARL_OK;
}
#endif
#elif defined(EARL_PHYS_UNIT_KITCHEN)
// This translation unit is physics_kitchen.hip: the nv = 23 instantiation of the stepper (32 lanes per env), the kitchen env kernels and the kitchen's entry points
void earl_unit_kitchen_physics(const void* pargs, int integrate, void* stream) {
  const PArgs& a = *static_cast<const PArgs*>(pargs);
  if (integrate) launch_physics<23, true>(a, (hipStream_t)stream);
  else launch_physics<23, false>(a, (hipStream_t)stream);
}
int earl_kitchen_step(const void* model, const earl_collision_model* col, const earl_kitchen_params* params, const earl_kitchen_cfg* cfg,
                      const earl_kitchen_state* st, const float* action, const earl_kitchen_out* out, earl_stream_t stream) {
  if (!model || !params || !cfg || !st || !action || !out || cfg->n < 0 || cfg->n_att < 10 || cfg->frame_skip < 0) return EARL_ERR_ARG;
  if (!st->qpos || !st->qvel || !st->mocap_pos || !st->goal || !st->last_qp_robot || !st->att_xpos || !st->steps_since_reset || !st->last_obs) return EARL_ERR_ARG;
  if (!st->action64 || !st->ctrl9 || !st->qpos_bak || !st->qvel_bak || !st->sites || !st->bad || !st->mocap_bak || !st->att_bak || (cfg->sensor_noise && !st->noise)) return EARL_ERR_ARG;
  if (!out->obs || !out->reward || !out->done || !out->success) return EARL_ERR_ARG;
  for (int k = 0; k < 8; ++k) if (cfg->site_att[k] < 0 || cfg->site_att[k] >= cfg->n_att) return EARL_ERR_ARG;
  const int n = cfg->n;
  if (n == 0) return EARL_OK;
  const hipStream_t hs = (hipStream_t)stream;
  if (int rc = check_cone(col, false, hs, "kitchen_step")) return rc;
  KitchenArgs k{*cfg, *st, *out, action, cfg->n_att};
  kitchen_pre_kernel<<<(n * 23 + 255) / 256, 256, 0, hs>>>(k);
  // KitchenV0.step up to do_simulation: mocap target, the nine position targets (csrc/glue.hip)
  if (int rc = earl_kitchen_action(n, params, st->action64, st->mocap_pos, st->last_qp_robot, st->ctrl9, stream)) return rc;
  // do_simulation: ctrl[i] = targets[i] for i < nu = 2, frame_skip timesteps (adept_envs/mujoco_env.py:148-157)
  PArgs a{model, col, n, cfg->frame_skip, st->qpos, st->qvel, st->mocap_pos, cfg->mocap_quat_dev, st->ctrl9, st->att_xpos, nullptr, nullptr, 9, 0};
  launch_physics<23, true>(a, hs);
  kitchen_guard_kernel<<<(n + 255) / 256, 256, 0, hs>>>(k);
  // Robot.get_obs + KitchenV0._get_obs: sensor noise from Philox draws keyed by the global env id
  if (cfg->sensor_noise)
    if (int rc = earl_philox_uniform(n, 46, cfg->seed, cfg->counter, cfg->env_offset, 0x4B00u, -1.0, 1.0, st->noise, stream)) return rc;
  if (int rc = earl_kitchen_obs(n, params, st->qpos, st->goal, cfg->sensor_noise ? st->noise : nullptr, out->obs, stream)) return rc;
  if (int rc = earl_kitchen_reward(n, out->obs, st->mocap_pos, st->sites, out->reward, out->success, stream)) return rc;
  kitchen_finish_kernel<<<(n + 255) / 256, 256, 0, hs>>>(k);
  return launched("kitchen_step");
}

int earl_kitchen_rollout(const void* model, const earl_collision_model* col, const earl_kitchen_params* params, const earl_kitchen_cfg* cfg,
                         const earl_kitchen_state* st, const float* action, int32_t T, const earl_kitchen_out* out, earl_stream_t stream) {
  if (!model || !params || !cfg || !st || !action || !out || cfg->n < 0 || T < 0 || cfg->n_att < 10 || cfg->n_att > 32 || cfg->frame_skip < 0) return EARL_ERR_ARG;
  if (!st->qpos || !st->qvel || !st->mocap_pos || !st->goal || !st->last_qp_robot || !st->att_xpos || !st->steps_since_reset || !st->last_obs) return EARL_ERR_ARG;
  if (!out->obs || !out->reward || !out->done || !out->success || !cfg->mocap_quat_dev) return EARL_ERR_ARG;
  for (int k = 0; k < 8; ++k) if (cfg->site_att[k] < 0 || cfg->site_att[k] >= cfg->n_att) return EARL_ERR_ARG;
  if (cfg->n == 0 || T == 0) return EARL_OK;
  if (int rc = check_cone(col, false, (hipStream_t)stream, "kitchen_rollout")) return rc;
  KitchenRolloutArgs k{model, col, *params, *cfg, *st, *out, action, T, solo_mode(cfg->n)};
  if (k.solo == 2 && g_solo < 0) k.solo = 3;   // one env per workgroup: four waves per env (rows | mass matrix | bias forces | collision, then one wave's active set)
  if (g_solo < 0 && k.solo == 1) {               // at most two envs per CU: two envs per workgroup, two waves per env
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (cfg->n <= 2 * cus) k.solo = 4;
  }
  if (k.solo == 3) kitchen_rollout_kernel<1><<<cfg->n, block_for<23>(), 0, (hipStream_t)stream>>>(k);
  else if (k.solo == 4) kitchen_rollout_kernel<2><<<(cfg->n + 1) / 2, block_for<23>(), 0, (hipStream_t)stream>>>(k);
  else kitchen_rollout_kernel<0><<<solo_grid(cfg->n, k.solo, Lim<23>::WPB), block_for<23>(), 0, (hipStream_t)stream>>>(k);
  return launched("kitchen_rollout");
}

int earl_debug_set_solo(int mode) {          // -1 = by batch size, 0 = two envs per wave, 1 = one env per wave, 2 = one env per workgroup (one wave), 3 = one env per workgroup, four waves, 4 = two envs per workgroup, two waves each (kitchen launches)
  const int prev = g_solo;
  if (mode >= -1 && mode <= 4) g_solo = mode;
  return prev;
}
#ifdef EARL_PHYS_PROF
int earl_debug_set_prof_wave_kitchen(int block, int thread) {     // this unit's copies of the profiling hooks (tools/prof_kitchen_phases.py)
  const int v[2] = {block, thread};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_prof_sel), v, sizeof(v)) == hipSuccess ? EARL_OK : EARL_ERR_LAUNCH;
}
int earl_debug_read_wave_cycles_kitchen(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_cycles), sizeof(unsigned long long) * 4096) == hipSuccess ? EARL_OK : EARL_ERR_LAUNCH;
}
int earl_debug_read_phys_profile_kitchen(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phys_prof), sizeof(unsigned long long) * 32) != hipSuccess) return EARL_ERR_LAUNCH;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phys_prof), z, sizeof(z)) != hipSuccess) return EARL_ERR_LAUNCH;
  }
  return EARL_OK;
}
#endif
#elif defined(EARL_PHYS_UNIT_L64)
// This translation unit is physics_l64.hip: one wavefront per env (64 lanes) for the nv = 10 / 15 models -- launched from physics.hip when earl_debug_set_physics_lanes(64) is on
void earl_unit_l64_physics(const void* pargs, int nv, int integrate, void* stream) {
  const PArgs& a = *static_cast<const PArgs*>(pargs);
  const hipStream_t st = (hipStream_t)stream;
  if (nv == 10 && integrate) physics_kernel<10, 64, true><<<grid_for<10, 64>(a.n), block_for<10>(), 0, st>>>(a);
  else if (nv == 10) physics_kernel<10, 64, false><<<grid_for<10, 64>(a.n), block_for<10>(), 0, st>>>(a);
  else if (nv == 15 && integrate) physics_kernel<15, 64, true><<<grid_for<15, 64>(a.n), block_for<15>(), 0, st>>>(a);
  else if (nv == 15) physics_kernel<15, 64, false><<<grid_for<15, 64>(a.n), block_for<15>(), 0, st>>>(a);
}
void earl_unit_l64_sawyer_rollout(const void* sawyer_args, int nv, void* stream) {
  const SawyerArgs& a = *static_cast<const SawyerArgs*>(sawyer_args);
  if (nv == 10) sawyer_rollout_kernel<10, 64><<<grid_for<10, 64>(a.cfg.n), block_for<10>(), 0, (hipStream_t)stream>>>(a);
  else if (nv == 15) sawyer_rollout_kernel<15, 64><<<grid_for<15, 64>(a.cfg.n), block_for<15>(), 0, (hipStream_t)stream>>>(a);
}
#elif defined(EARL_PHYS_VARIANT_W8)
// This translation unit is physics_w8.hip: the door model's rollout kernel built with eight-wave workgroups (EARL_DOOR_WPB 8: 32 envs share one
// copy of the tables, packed matrices, in-LDS factorisations, 256 registers per wave) = eight waves per CU.  Same arithmetic, bit-identical
// outputs; chosen by earl_sawyer_rollout for batches of more than 4096 envs (one round of 8192 envs instead of two of 4096).
int earl_sawyer_rollout_door_w8(const earl_link_model* model, const earl_collision_model* col, const earl_sawyer_cfg* cfg, const earl_sawyer_state* st,
                                const float* action, int32_t T, const earl_sawyer_out* out, earl_stream_t stream) {
  SawyerArgs a{model, col, *cfg, *st, action, T, *out, nullptr, nullptr, nullptr, nullptr, 0};
  sawyer_rollout_kernel<10, 16><<<grid_for<10, 16>(cfg->n), block_for<10>(), 0, (hipStream_t)stream>>>(a);
  return launched("sawyer_rollout (door, 8 waves per CU)");
}
#ifdef EARL_PHYS_PROF
int earl_debug_set_prof_wave_w8(int block, int thread) {     // this unit's copy of earl_debug_set_prof_wave (eight-wave workgroups: thread = 64 x the wave)
  const int v[2] = {block, thread};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_prof_sel), v, sizeof(v)) == hipSuccess ? EARL_OK : EARL_ERR_LAUNCH;
}
int earl_debug_read_wave_cycles_w8(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_cycles), sizeof(unsigned long long) * 4096) == hipSuccess ? EARL_OK : EARL_ERR_LAUNCH;
}
int earl_debug_read_phys_profile_w8(unsigned long long* out, int reset) {          // this unit's own copy of the phase counters
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phys_prof), sizeof(unsigned long long) * 32) != hipSuccess) return EARL_ERR_LAUNCH;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phys_prof), z, sizeof(z)) != hipSuccess) return EARL_ERR_LAUNCH;
  }
  return EARL_OK;
}
#endif
#else
int earl_sawyer_rollout_door_w8(const earl_link_model* model, const earl_collision_model* col, const earl_sawyer_cfg* cfg, const earl_sawyer_state* st,
                                const float* action, int32_t T, const earl_sawyer_out* out, earl_stream_t stream);      // physics_w8.hip
int earl_sawyer_rollout(const earl_link_model* model, const earl_collision_model* col, int32_t nv, const earl_sawyer_cfg* cfg, const earl_sawyer_state* st,
                        const float* action, int32_t T, const earl_sawyer_out* out, earl_stream_t stream) {
  if (!model || !cfg || !st || !out || !action || T < 0 || cfg->n < 0) return EARL_ERR_ARG;
  if (!st->qpos || !st->qvel || !st->mocap_pos || !st->goal || !out->obs) return EARL_ERR_ARG;
  if (cfg->frame_skip < 0 || cfg->att_hand < 0 || cfg->att_right < 0 || cfg->att_left < 0 || cfg->att_obj < 0) return EARL_ERR_ARG;
  if (cfg->n == 0 || T == 0) return EARL_OK;
  if (nv != 10 && nv != 15) return EARL_ERR_ARG;
  if (int rc = check_cone(col, true, (hipStream_t)stream, "sawyer_rollout")) return rc;
  SawyerArgs a{model, col, *cfg, *st, action, T, *out, nullptr, nullptr, nullptr, nullptr, 0};
  if (cfg->obj_kind >= 1 && cfg->reward_type != 0 && (!st->obj_init || cfg->att_grasp < 0 || cfg->att_lpad < 0 || cfg->att_rpad < 0))
    return EARL_ERR_ARG;                                  // the peg's dense reward needs the reset-time state and the pad / grasp attachments
  if (nv == 10 && g_lpe != 64 && g_door_variant == 3 && st->sched && T > 1) {
    // (measurement switch: the single-wave build, four workgroups per CU, under the time-sliced work queue of the peg -- tools/bench_variant.py)
    int dev = 0, cus = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    a.slice = g_peg_sliced >= 2 ? g_peg_sliced : EARL_PEG_SLICE;
    sawyer_rollout_kernel<10, 16, true><<<4 * cus, block_for<10>(), 0, (hipStream_t)stream>>>(a);
    return launched("sawyer_rollout (door, time-sliced)");
  }
  if (nv == 10 && g_lpe != 64 && (g_door_variant == 2 || (g_door_variant == 0 && cfg->n > 4096)))
    return earl_sawyer_rollout_door_w8(model, col, cfg, st, action, T, out, stream);      // eight waves per CU: wins from two rounds of 4096 envs on
  if (nv == 10) {
    if (g_lpe == 64) earl_unit_l64_sawyer_rollout(&a, 10, stream);
    else sawyer_rollout_kernel<10, 16><<<grid_for<10, 16>(cfg->n), block_for<10>(), 0, (hipStream_t)stream>>>(a);
  } else if (nv == 15) {
    if (g_lpe == 64) earl_unit_l64_sawyer_rollout(&a, 15, stream);
    else {
      // more workgroups than the GPU holds at once (one four-wave workgroup = 16 envs per CU): time-sliced schedule, one persistent workgroup per CU
      int dev = 0, cus = 256;
      if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      if (st->sched && g_peg_sliced && grid_for<15, 16>(cfg->n) > cus && T > 1) {
        a.slice = g_peg_sliced >= 2 ? g_peg_sliced : EARL_PEG_SLICE;
        sawyer_rollout_kernel<15, 16, true><<<cus, block_for<15>(), 0, (hipStream_t)stream>>>(a);
      } else sawyer_rollout_kernel<15, 16><<<grid_for<15, 16>(cfg->n), block_for<15>(), 0, (hipStream_t)stream>>>(a);
    }
  } else return EARL_ERR_ARG;
  return launched("sawyer_rollout");
}

int earl_sawyer_reset(const earl_link_model* model, int32_t nv, const earl_sawyer_cfg* cfg, const earl_sawyer_state* st,
                      const double* reset_qpos, const double* reset_qvel, const uint8_t* mask, double* obs,
                      earl_stream_t stream) {
  if (!model || !cfg || !st || !reset_qpos || !reset_qvel || cfg->n < 0) return EARL_ERR_ARG;
  if (!st->qpos || !st->qvel || !st->mocap_pos || !st->goal) return EARL_ERR_ARG;
  if (cfg->obj_dof < 0 || cfg->obj_dof >= nv || cfg->obj_kind < 0 || cfg->obj_kind > 2) return EARL_ERR_ARG;
  if (cfg->obj_kind >= 1 && cfg->obj_dof + 6 > nv) return EARL_ERR_ARG;
  if (cfg->obj_kind == 2 && (cfg->n_wide <= 0 || !cfg->wide_table)) return EARL_ERR_ARG;
  if (cfg->n == 0) return EARL_OK;
  SawyerArgs a{model, nullptr, *cfg, *st, nullptr, 0, earl_sawyer_out{nullptr, nullptr, nullptr, nullptr, nullptr}, reset_qpos, reset_qvel, mask, obs, 0};
  if (nv == 10) sawyer_reset_kernel<10, 16><<<grid_for<10, 16>(cfg->n), block_for<10>(), 0, (hipStream_t)stream>>>(a);
  else if (nv == 15) sawyer_reset_kernel<15, 16><<<grid_for<15, 16>(cfg->n), block_for<15>(), 0, (hipStream_t)stream>>>(a);
  else return EARL_ERR_ARG;
  return launched("sawyer_reset");
}

int earl_sawyer_observe(const earl_link_model* model, int32_t nv, const earl_sawyer_cfg* cfg, const earl_sawyer_state* st, double* obs,
                        earl_stream_t stream) {
  if (!model || !cfg || !st || !obs || cfg->n < 0) return EARL_ERR_ARG;
  if (!st->qpos || !st->qvel || !st->mocap_pos || !st->goal) return EARL_ERR_ARG;
  if (cfg->n == 0) return EARL_OK;
  SawyerArgs a{model, nullptr, *cfg, *st, nullptr, 0, earl_sawyer_out{nullptr, nullptr, nullptr, nullptr, nullptr}, nullptr, nullptr, nullptr, obs, 1};
  if (nv == 10) sawyer_reset_kernel<10, 16><<<grid_for<10, 16>(cfg->n), block_for<10>(), 0, (hipStream_t)stream>>>(a);
  else if (nv == 15) sawyer_reset_kernel<15, 16><<<grid_for<15, 16>(cfg->n), block_for<15>(), 0, (hipStream_t)stream>>>(a);
  else return EARL_ERR_ARG;
  return launched("sawyer_observe");
}

int earl_sawyer_door_reward(const earl_sawyer_cfg* cfg, int32_t n, const double* obs, float* reward, uint8_t* success,
                            earl_stream_t stream) {
  if (!cfg || n < 0 || !obs) return EARL_ERR_ARG;
  if (n == 0) return EARL_OK;
  sawyer_door_reward_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(n, obs, *cfg, reward, success);
  return launched("sawyer_door_reward");
}

int earl_sawyer_door_info(const earl_sawyer_cfg* cfg, int32_t n, const double* obs, const uint8_t* status, double* info, earl_stream_t stream) {
  if (!cfg || n < 0 || !obs || !info) return EARL_ERR_ARG;
  if (n == 0) return EARL_OK;
  sawyer_door_info_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(n, obs, *cfg, status, info);
  return launched("sawyer_door_info");
}



int earl_debug_set_physics_lanes(int lanes_per_env) {
  if (lanes_per_env != 16 && lanes_per_env != 64) return EARL_ERR_ARG;
  g_lpe = lanes_per_env;
  return EARL_OK;
}

#ifdef EARL_PHYS_PROF
int earl_debug_set_prof_wave(int block, int thread) {     // the wave whose phases the profiling build clocks (default: workgroup 0, thread 0)
  const int v[2] = {block, thread};
  return hipMemcpyToSymbol(HIP_SYMBOL(g_prof_sel), v, sizeof(v)) == hipSuccess ? EARL_OK : EARL_ERR_LAUNCH;
}
int earl_debug_read_wave_cycles(unsigned long long* out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_cycles), sizeof(unsigned long long) * 4096) == hipSuccess ? EARL_OK : EARL_ERR_LAUNCH;
}
int earl_debug_read_phys_profile(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phys_prof), sizeof(unsigned long long) * 32) != hipSuccess) return EARL_ERR_LAUNCH;
  if (reset) {
    unsigned long long z[32] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_phys_prof), z, sizeof(z)) != hipSuccess) return EARL_ERR_LAUNCH;
  }
  return EARL_OK;
}
#endif

int earl_physics_model_size(void) { return (int)sizeof(earl_link_model); }
int earl_physics_model24_size(void) { return (int)sizeof(earl_link_model24); }
int earl_collision_model_size(void) { return (int)sizeof(earl_collision_model); }
int earl_sawyer_cfg_size(void) { return (int)sizeof(earl_sawyer_cfg); }

int earl_debug_set_peg_schedule(int sliced) {          // 0: one group per wave; 1: time-sliced, EARL_PEG_SLICE env steps per item; k >= 2: time-sliced, k env steps per item
  if (sliced < 0) return EARL_ERR_ARG;
  g_peg_sliced = sliced;
  return EARL_OK;
}
int earl_debug_set_door_variant(int v) {
  if (v < 0 || v > 3) return EARL_ERR_ARG;
  g_door_variant = v;
  return EARL_OK;
}
#endif   // EARL_PHYS_VARIANT_W8

}  // extern "C"
