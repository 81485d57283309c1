// physics_lds.h -- per-model limits (Lim<NV>), the per-env LDS block (Shared<NV>), packed symmetric matrices, friction-cone weights, the collision block table and its staging
// A section of csrc/physics.hip (included there, inside its anonymous namespace): split out in round 5.

// Symmetric NV x NV matrix in LDS.  Big models (nv > 10) keep the lower triangle packed row-major (row i, column j <= i at
// i (i + 1) / 2 + j): half the LDS of a square array, and the env block is what limits the number of waves on a CU.  Small models
// keep the plain square (both triangles written): their workgroup fits four times into a CU either way, and the packed form's index
// arithmetic cost the door kernel 500 VALU instructions per timestep.
#ifndef EARL_DOOR_PACKED
#define EARL_DOOR_PACKED 0
#endif
template <int NV>
struct SymLds {
  static constexpr bool PACKED = NV > 10 || EARL_DOOR_PACKED;   // (small model: packed only in the eight-waves-per-CU build, where the block size decides)
  double v[PACKED ? NV * (NV + 1) / 2 : NV * NV];
  __device__ __forceinline__ double& lo(const int i, const int j) { return v[PACKED ? i * (i + 1) / 2 + j : i * NV + j]; }               // i >= j
  __device__ __forceinline__ const double& lo(const int i, const int j) const { return v[PACKED ? i * (i + 1) / 2 + j : i * NV + j]; }
  // entry (i, l) in either order; ltri = l (l + 1) / 2 is kept per lane (i is a compile-time index at every call site)
  __device__ __forceinline__ double sym(const int i, const int l, const int ltri) const {
    if constexpr (PACKED) return v[i >= l ? i * (i + 1) / 2 + l : ltri + i];
    else return v[i * NV + l];
  }
  // lane l stores entry (i, l) of its column: the packed form keeps the lower part only, the square form also the mirror image
  __device__ __forceinline__ void put(const int i, const int l, const double x, const bool mirror) {
    if constexpr (PACKED) { if (i >= l) v[i * (i + 1) / 2 + l] = x; }
    else { v[i * NV + l] = x; if (mirror) v[l * NV + i] = x; }
  }
  __device__ __forceinline__ double& rowl(const int l, const int ltri, const int j) { return v[PACKED ? ltri + j : l * NV + j]; }        // (l, j), j <= l
};

// model table type by size: nv <= 16 the compact form, nv = 23 (kitchen) the 24-dof form with the extra joint tables
template <int NV> struct ModelOf { using T = earl_link_model; };
template <> struct ModelOf<23> { using T = earl_link_model24; };
template <> struct ModelOf<22> { using T = earl_link_model24; };   // the minitaur: floating root + 16 hinges

// Static bounds by model size: the door model (nv 10) keeps 8 contact slots and 16 collision blocks, which keeps its workgroup
// under 40 KB of LDS (four workgroups per CU, one wave per SIMD); the peg model (nv 15) needs 12 / 32.
template <int NV> struct Lim {
  static constexpr int MC = NV <= 10 ? 8 : EARL_MAXCON;     // contact slots (runtime cap: earl_collision_model.max_con <= MC)
  static constexpr int MB = NV <= 10 ? 16 : (NV == 23 ? 64 : (NV == 22 ? 8 : 32));   // collision blocks (<= EARL_MAXBLK; the peg model has 29, the kitchen 56 since round 3: 64-bit near masks, the minitaur 7)
#ifndef EARL_DOOR_WPB
#define EARL_DOOR_WPB 1
#endif
  static constexpr int WPB = NV <= 10 ? EARL_DOOR_WPB : 4;  // wavefronts per workgroup.  nv 10: 33 KB per single-wave workgroup, four per CU.  nv 15: an
                                                            // env block is 9.5 KB; a four-wave workgroup (16 envs + the tables once = 163,672 B of the CU's
                                                            // 163,840) puts one wave on every SIMD where single-wave workgroups would fit two or three.
                                                            // nv 23: 32 lanes per env, two envs per wave, four waves = 8 envs per workgroup (one per CU)
#ifndef EARL_DOOR_COOP
#define EARL_DOOR_COOP 0
#endif
  static constexpr bool COOP = (NV <= 10 && EARL_DOOR_COOP) || NV > 16;  // every factorisation shared in LDS instead of per lane in registers: an experiment for
                                                            // the small model, the only possibility for nv = 23 (a register-resident factor would need 552 VGPRs)
  static constexpr bool ELLIPTIC = NV <= 16;               // friction cone of the model's MJCF: the Sawyer door and peg (metaworld's basic_scene.xml: cone="elliptic") carry the contact rows
                                                            // (normal, t1, t2) with MuJoCo's three-zone cost (round 4); the kitchen and the minitaur keep the four pyramid edges.  The host
                                                            // side refuses tables of the other kind (earl_collision_model.cone, physics/__init__.py)
  static constexpr bool CAPS = NV <= 10;                    // edge-vs-capsule blocks compiled in (the door model's handle rods; the peg model has none, and
                                                            // its kernel has no registers to spare: the host side refuses such tables for it)
  static constexpr int NA = (NV == 15 || NV == 23) ? 9 : NV;              // block split of the factorisations: the peg model's arm (7 hinges + 2 claw slides)
                                                            // and free peg are separate trees (checked by the host side); the door model
                                                            // (9 + 1) is factorised densely -- the split did not pay there; the kitchen's arm (7 + 2)
                                                            // is one tree and each of its 14 fixtures its own (coupled at most in pairs)
  static constexpr int NT = NV == 23 ? 9 : NV;               // links that can have ancestors / descendants other than themselves: all, except in the kitchen
                                                            // model, where only the arm's nine do (every fixture is a tree of one link; checked by the host
                                                            // side).  The masked ancestor / subtree sums run over [0, NT) plus the lane's own link.
  static constexpr int TS = NV == 15 ? 9 : NT;              // peg model: two trees, links [0, 9) = arm and [9, 15) = the free peg (checked by the host side, like NA): a
                                                            // lane's ancestor / subtree sums then run over its OWN tree only, 9 terms instead of 15 (the others had weight 0)
  static constexpr bool KBT = NV != 15;                     // take the rows' (k, b) from the per-launch table (stage_kb) instead of recomputing them in every timestep: door +0.9 %,
                                                            // kitchen +1.5 %; the peg build (512 registers, one wave per SIMD: the recomputation hides behind LDS latency) -4 %
  static constexpr bool EXTRAS = NV == 23;                  // dry joint friction, joint springs, force-limited actuators, joint couplings (earl_link_model24), and
                                                            // the kitchen's structured solver (arm block + fixtures)
  // the minitaur (nv = 22): ONE tree (a free root body + 16 hinges), no mocap weld, connect constraints (the knee closures), generalized forces
  // handed in per timestep (the motor model's torques), no joint damping; dense in-LDS factorisations (COOP)
  static constexpr bool WELD = NV != 22;                    // six weld rows to the mocap body
  static constexpr bool CONNECT = NV == 22;                 // connect constraints (earl_link_model24.n_con) and the external-force vector s.xt.ext
  static constexpr bool DAMPED = NV != 22;                  // joint damping (K10's implicit step (M + dt B) a' = M a; without damping a' = a)
  static constexpr int LPE = NV > 16 ? 32 : 16;             // lanes per env instance (64 = one wavefront per env: measurement switch for nv <= 16)
  // Models whose first tree is the ARM of these robots -- a serial chain of seven hinges (links 0-6) with the two finger slides (7, 8) on the hand -- and whose
  // other links are a free body's chain of six (the peg: links 9-14) or single-link trees (the kitchen's fixtures); checked by the host side.  For them the
  // world frames, velocities, bias accelerations and the two subtree sums (composite inertia, bias force) are SCANS along the chain, done in registers with
  // DPP row shifts (the arm sits in lanes 0-8 of one 16-lane row) instead of masked sums over every link through LDS: K1-K7 were 20 k cycles per timestep.
#ifndef EARL_NO_ARMSCAN
#define EARL_NO_ARMSCAN 0
#endif
  static constexpr bool ARMSCAN = (NV == 15 || NV == 23) && !EARL_NO_ARMSCAN;
#ifndef EARL_NO_PACK
#define EARL_NO_PACK 0
#endif
  static constexpr bool PACK = NV == 23 && !EARL_NO_PACK;   // pair tests: several near blocks per pass (blocks of <= 10 pairs on 32 lanes); results unchanged
  static constexpr int BODY0 = NV == 15 ? 9 : -100;         // first link of the free body's chain (its six links: three slides, the quaternion link, two rigid ones)
};

// v = W (j0, j1, j2) for one contact's weight record w (K9).  Pyramid: W = [[w0, w1, w2], [w1, w3, 0], [w2, 0, w4]] (sums over the active edges).  Elliptic cone: the record is
// (K, m1, m2, q, 1 / mu^2): W = K (1, m1, m2)(1, m1, m2)' + q (I2 - m m' / mu^2) on the tangential block -- the bottom zone is (D, 0, 0, D, .), the top zone all zeros
// (reference: LinkModel.solve_primal_elliptic)
template <bool ELL>
__device__ __forceinline__ void cone_apply(const double* w, const double j0, const double j1, const double j2, double& v0, double& v1, double& v2) {
  if constexpr (ELL) {
    const double K = w[0], m1 = w[1], m2 = w[2], q = w[3], i2 = w[4];
    const double h01 = K * m1, h02 = K * m2, h11 = K * m1 * m1 + q * (1.0 - m1 * m1 * i2), h22 = K * m2 * m2 + q * (1.0 - m2 * m2 * i2), h12 = m1 * m2 * (K - q * i2);
    v0 = K * j0 + h01 * j1 + h02 * j2; v1 = h01 * j0 + h11 * j1 + h12 * j2; v2 = h02 * j0 + h12 * j1 + h22 * j2;
  } else {
    v0 = w[0] * j0 + w[1] * j1 + w[2] * j2; v1 = w[1] * j0 + w[3] * j1; v2 = w[2] * j0 + w[4] * j2;
  }
}
__device__ __forceinline__ int cone_zone(const double r0, const double r1, const double r2, const double mu) {      // 0 top (separating), 1 bottom (sticking), 2 middle (sliding)
  const double rho = sqrt(r1 * r1 + r2 * r2);
  return r0 >= mu * rho ? 0 : (rho <= -mu * r0 ? 1 : 2);
}


// the equality part of the Hessian (M + weld / coupling / drag rows), kept in LDS for the big model: its 23-entry columns would otherwise sit in
// registers across the whole active-set iteration (the nv = 23 kernel spilled 1.6 KB per lane into scratch)
template <int NV, bool ON> struct HwStore {};
template <int NV> struct HwStore<NV, true> { SymLds<NV> Hw; };

// connect constraints (3 rows each) and the generalized forces applied from outside: only in the models that have them (the Sawyer workgroups fill a
// CU's LDS to the last 200 bytes)
template <int NV, bool ON> struct ConStore {};
template <int NV> struct ConStore<NV, true> {
  double JE[3 * EARL_MAXCONNECT][NV];   // Jacobian rows: Jp(att1) - Jp(att2)
  double eD[3 * EARL_MAXCONNECT], ear[3 * EARL_MAXCONNECT], eres[3 * EARL_MAXCONNECT];
  double ext[NV];
  signed char crow[EARL_MAXCON][2];      // per contact: the (at most two) dofs beyond the root body's six that its Jacobian touches, -1 = none (K9's column update)
  double motor_volt, motor_visc;         // the env's battery voltage and motor viscous damping (earl_minitaur_state.motor_param[0..1]): read by ApplyAction in every timestep -- kept here, not in two
                                         // registers that live across the whole rollout (they were the one spill reloaded inside the timestep loop; tools/scratch_in_loops.py)
  double mscale[3], foot_mu;             // the minitaur's per-env randomisation (earl_minitaur_state.motor_param[2..5]): mass / inertia factor of the root body, the upper links,
                                         // the lower links; friction of the lower links' contacts (<= 0: the classes' own).  Unused (1, 1, 1, -1) elsewhere.
};

// Per-env LDS block.  The three phase groups of the union are live at disjoint times.
template <int NV>
struct SharedData {
#ifndef EARL_PAD_ROWS
#define EARL_PAD_ROWS 0             // (measurement switch, round 6: rows of NV doubles padded to an even length for the peg's odd NV and M / the phase union / rc / mocap aligned
                                    // to 16 bytes: ds_read_b128 221 -> 307, ds_read2_b64 199 -> 113, LDS 152 -> 158 kB, outputs identical -- and 42.4 -> 42.8 ms per launch: the peg is not bound by
                                    // LDS-array cycles, unlike the minitaur; profiles/r06_stepper_build_experiments.txt)
#endif
  static constexpr int MC = Lim<NV>::MC;
  static constexpr int NVP = (EARL_PAD_ROWS && NV <= 16) ? ((NV + 1) & ~1) : NV;
  double qp[NV], qv[NV];
  double bq[4];                      // orientation of the free body (unit quaternion), identity if the model has none
  double Xq[NV][4], Xp[NV][3];       // world frame of every link (final buffer of the ancestor doubling)
#if EARL_PAD_ROWS
#define EARL_A16 alignas(16)
#else
#define EARL_A16
#endif
  EARL_A16 SymLds<NV> M;             // mass matrix
  HwStore<NV, (Lim<NV>::EXTRAS || Lim<NV>::CONNECT)> hwst;
  ConStore<NV, Lim<NV>::CONNECT> xt;
  EARL_A16 union {
    struct { double Xq1[NV][4], Xp1[NV][3]; } k2;                    // second buffer of the doubling
    struct { double att[8][3]; } emit;                               // observation epilogue (after the last timestep of an env step)
    struct { double obs[46], noise[46], sites[8][3], targets[9]; } kit;   // kitchen env step inside the fused rollout (before / after the timesteps)
    struct {
      double S[NV][6];                 // motion subspace, world coordinates about the origin: [angular; linear] (every lane keeps its own column in registers)
      double I10[NV][10];
      union {
        struct { double Ic[NV][10], FS[NV][6]; } crb;
        struct { double V[NV][6], Cc[NV][6], F[NV][6]; } rne;
      };
    } dyn;
    struct {
      double wD[6], war[6], dl[NV], rl[NV];
      double CJ[MC][3][NVP];           // contact Jacobians: normal, tangent 1, tangent 2
      union {
        double ct[MC][8];              // contact records (C2 -> C3): dist, normal (3), point (3), (class, sphere link + 1, box link + 1) packed as class + 64 (ls + 1) + 4096 (lb + 1)
        double cw[MC][8];              // per-iteration weights of the active pyramid edges (K9)
      };
      union {
        double J6[6][NVP];             // weld Jacobian (K8 -> the equality part of K9; dead once every lane holds its Hessian column hw)
        SymLds<NV> Hc;                 // Hessian of the iteration (written after that); the shared factorisation overwrites it with L
      };
      EARL_A16 double rc[NV];          // its right-hand side; then the right-hand side of K10
    } con;
  };
  EARL_A16 double mocap[4];          // mocap position of this env (input of the weld rows; in LDS rather than in six registers that live across the whole rollout)
  double aprev[(NV + 1) & ~1];       // solution of the previous timestep of this env step: warm start of the active-set iteration (last, even
                                     // length: the 16-byte alignment of the arrays above decides between ds_read_b128 and two b64)
};
// The four env blocks of a wave must not start on the same LDS banks (every broadcast access would conflict 4 ways): the block
// size is padded to 64 or 192 mod 256 bytes, whichever is nearer
#ifndef EARL_STRIDE_MOD_10
#define EARL_STRIDE_MOD_10 -1
#endif
#ifndef EARL_STRIDE_MOD_15
#define EARL_STRIDE_MOD_15 -1
#endif
#ifndef EARL_STRIDE_MOD_23
#define EARL_STRIDE_MOD_23 -1
#endif
// What only the kitchen model's block holds (an empty base class elsewhere: the other models' blocks keep their size and the 16-byte alignment of their arrays).
// jeq.rec: joint couplings -- per coupled dof l >= NT the four numbers lane l's part of the equality Hessian takes: D J_l, the coupling's reference acceleration, the
// term of its diagonal entry, the term of the entry it shares with its partner (K8 writes, K9 reads).  duo_ctrl, tau, duo_nct: four waves per env (substep's ROLE 1 - 4) --
// the env step's two actuator targets, published by wave B for the bias-force wave (whose K7 applies them); the generalized forces that wave computed and the contact
// count the collision wave found, both read by wave B after barrier X.
template <int NV, bool ON> struct KitchenStore { static constexpr int SIZE = 0; };
template <int NV> struct KitchenStore<NV, true> {
  struct { double rec[NV - Lim<NV>::NT][4]; } jeq;
  double duo_ctrl[2];
  double tau[NV];
  int duo_nct, duo_pad_;               // the env's contact count, left by the collision wave for wave B
  static constexpr int SIZE = (int)sizeof(double) * ((NV - Lim<NV>::NT) * 4 + 2 + NV + 1);
};
template <int NV>
struct Shared : SharedData<NV>, KitchenStore<NV, Lim<NV>::EXTRAS> {
  static constexpr int R = (int)((sizeof(SharedData<NV>) + KitchenStore<NV, Lim<NV>::EXTRAS>::SIZE) % 256);
  static constexpr int TARGET = NV <= 10 ? (EARL_STRIDE_MOD_10) : (NV <= 15 ? (EARL_STRIDE_MOD_15) : (EARL_STRIDE_MOD_23));   // block size mod 256 (-1: the rule above)
  static constexpr int PAD = TARGET >= 0 ? (TARGET - R + 256) % 256 : (R <= 64 ? 64 - R : (R <= 192 ? 192 - R : 320 - R));
  char bank_pad[PAD == 0 ? 16 : PAD];        // (never 8: see SharedMT)
};
static_assert(sizeof(Shared<10>) % 16 == 0 && sizeof(Shared<15>) % 16 == 0 && sizeof(Shared<22>) % 16 == 0 && sizeof(Shared<23>) % 16 == 0, "env blocks are whole numbers of 16-byte words");


// block table of the collision model (bounding tests), staged once per workgroup
template <int MB, bool KB>
struct BlkTable {
  int n_blk, max_con;
  int begin[MB], end[MB], box[MB], link[MB], box_link[MB], cap[MB];
  double center[MB][3], reach[MB], box_pos[MB][3], box_quat[MB][4], box_half[MB][3];
  // second bounding test (include/earl_physics.h blk_obb_*).  Compiled in for the small model only: the door's random-action workload has 2.4 near
  // blocks per env by the sphere test and 0.5 by both (35.1 -> 36.1 M env-steps/s); the peg lies on the table (that block is always near) and the
  // kitchen's hands are far from the fixtures, so there the extra test only costs (peg -3.5 %, kitchen 0).  Results do not depend on it.
  static constexpr bool SAT = MB <= 16;
  double obb_center[SAT ? MB : 1][3], obb_half[SAT ? MB : 1][3];
  double cls_mu[EARL_MAXCLS], cls_margin[EARL_MAXCLS], cls_invw[EARL_MAXCLS], cls_solref[EARL_MAXCLS][2], cls_solimp[EARL_MAXCLS][5];   // contact classes
  // (k, b) of every row kind, once per launch (stage_kb): weld, joint limit / dry friction of dof l, contact class, joint coupling; the dry-friction
  // rows' regulariser (their residual is always 0, so the whole of it is a constant)
  // (not in the peg build, Lim<15>::KBT: even unused, the 976 B in front of the env blocks cost it 4 %)
  double kb_weld[2], kb_lim[KB ? 24 : 1][2], kb_cls[KB ? EARL_MAXCLS : 1][2], kb_jeq[KB ? 8 : 1][2], fr_D[KB ? 24 : 1];
};
template <int MB, bool KB>
__device__ __forceinline__ void stage_blocks(BlkTable<MB, KB>& t, const earl_collision_model* __restrict__ col) {
  const int i = threadIdx.x;                           // n_blk <= 32 < one wavefront
  // (bounds are clamped here; the Python / C front ends refuse models that exceed them)
  const int nb = col ? (col->n_blk < MB ? col->n_blk : MB) : 0;
  if (i == 0) { t.n_blk = nb; t.max_con = col ? col->max_con : 0; }
  if (col && i < EARL_MAXCLS) {
    t.cls_mu[i] = col->cls_mu[i]; t.cls_margin[i] = col->cls_margin[i]; t.cls_invw[i] = col->cls_invw[i];
    t.cls_solref[i][0] = col->cls_solref[i][0]; t.cls_solref[i][1] = col->cls_solref[i][1];
#pragma unroll
    for (int k = 0; k < 5; ++k) t.cls_solimp[i][k] = col->cls_solimp[i][k];
  }
  if (i < nb) {
    const int b = col->blk_box[i];
    t.begin[i] = col->blk_begin[i]; t.end[i] = col->blk_end[i]; t.box[i] = b; t.link[i] = col->blk_link[i]; t.cap[i] = col->blk_cap[i];
    t.box_link[i] = col->box_link[b]; t.reach[i] = col->blk_reach[i];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      t.center[i][k] = col->blk_center[i][k]; t.box_pos[i][k] = col->box_pos[b][k]; t.box_half[i][k] = col->box_half[b][k];
      if constexpr (BlkTable<MB, KB>::SAT) { t.obb_center[i][k] = col->blk_obb_center[i][k]; t.obb_half[i][k] = col->blk_obb_half[i][k]; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) t.box_quat[i][k] = col->box_quat[b][k];
  }
}

template <int NV, int MB, bool KB>
__device__ __forceinline__ void stage_kb(BlkTable<MB, KB>& t, const void* __restrict__ model, const earl_collision_model* __restrict__ col) {
  if constexpr (!Lim<NV>::KBT) return;
  const typename ModelOf<NV>::T* mg = reinterpret_cast<const typename ModelOf<NV>::T*>(model);     // (global memory: the LDS copy is not complete yet)
  const int i = threadIdx.x;
  const double dt = mg->dt;
  if (i < NV) {
    kb_of(mg->jsolref[i], mg->jsolimp[i], dt, t.kb_lim[i][0], t.kb_lim[i][1]);
    if constexpr (Lim<NV>::EXTRAS) {
      const double dd = imp_of(mg->jsolimp[i], 0.0);
      t.fr_D[i] = rcp_nr(fmax((1 - dd) * mg->dof_invweight[i] * rcp_nr(dd), 1e-15));
    }
  }
  if (i == 32) kb_of(mg->weld_solref, mg->weld_solimp, dt, t.kb_weld[0], t.kb_weld[1]);
  if (col && i >= 33 && i < 33 + EARL_MAXCLS) kb_of(col->cls_solref[i - 33], col->cls_solimp[i - 33], dt, t.kb_cls[i - 33][0], t.kb_cls[i - 33][1]);
  if constexpr (Lim<NV>::EXTRAS) {
    if (i >= 56 && i < 64 && i - 56 < mg->n_jeq) kb_of(mg->jeq_solref[i - 56], mg->jeq_solimp[i - 56], dt, t.kb_jeq[i - 56][0], t.kb_jeq[i - 56][1]);
  }
}

