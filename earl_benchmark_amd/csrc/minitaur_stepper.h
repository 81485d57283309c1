// minitaur_stepper.h -- the minitaur's timestep written ON THE MODEL'S TREE: a floating root body (dofs 0-5) carrying four legs
// of four hinges each (leg k = dofs 6 + 4 k ... 9 + 4 k: upper / lower link of the L chain, upper / lower link of the R chain, the two lower links tied
// by the leg's connect constraint).  Included by physics.hip inside its anonymous namespace when built as physics_mt.hip (EARL_PHYS_VARIANT_MT).
//
// Same algorithm and the same numbers (to rounding) as the generic substep<22> of physics.hip, which it replaces on the product path (the generic
// one stays selectable: earl_debug_set_minitaur_stepper, tests/test_minitaur_gpu.py compares the two); reference: oracle/physics_oracle.py
// LinkModel.forward / step, oracle/physics_oracle.c.  What the structure buys (cycles per timestep of a wave = two envs, round-3 generic kernel
// in brackets; tools/prof_minitaur.py):
//   * frames, velocities, bias forces: a link's ancestors are the root body and at most one hinge -- everything about the root body is computed
//     in registers from qpos / qvel by every lane, parent <-> child values move by DPP quad permutes (legs sit on aligned lane quads:
//     lane 8 + 4 k + j), and the only LDS exchanges are the two subtree sums of the root body (composite inertia, bias force) [K1-K7 37 k]
//   * the constraint Hessian M + J' D J is an ARROW matrix: root block R (6 x 6), leg blocks A_k (4 x 4), couplings B_k (4 x 6), nothing between
//     different legs (closures and contacts touch the root body and one leg).  Only those 157 of the 253 entries exist here, the equality part
//     stays in registers (11 numbers per lane), and contact terms are accumulated in registers per pass [K9a 8 k, Hessian columns 11 k per pass]
//   * elimination order legs -> root has no fill-in: four 4 x 4 factorisations side by side, a Schur complement onto the root block, one 6 x 6
//     factorisation redundantly per lane in registers -- three wave-level LDS exchanges per solve instead of one per column (22) plus one per
//     substitution step [factor + solve 15.6 k per pass]
//   * contact rows: lane = contact, all contacts at once; a contact's Jacobian has the root body's six entries and the two of its own chain
//     (12 x 3 x 8 numbers instead of 12 x 3 x 22) [C3 6.5 k]
// The structure is CHECKED by the host side (earl_benchmark_amd/physics/__init__.py load_link_model / load_collision_model, nv == 22); the kernels
// assume it.
#pragma once

// ---- DPP moves within an aligned quad of lanes (a leg): CTRL = quad_perm selector (lane i of the quad reads lane (CTRL >> 2 i) & 3)
template <int CTRL>
__device__ __forceinline__ double dpp_quad(const double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ V3 dpp_quad(const V3& v) { return {dpp_quad<CTRL>(v.x), dpp_quad<CTRL>(v.y), dpp_quad<CTRL>(v.z)}; }
template <int CTRL> __device__ __forceinline__ Q4 dpp_quad(const Q4& q) { return {dpp_quad<CTRL>(q.w), dpp_quad<CTRL>(q.x), dpp_quad<CTRL>(q.y), dpp_quad<CTRL>(q.z)}; }
constexpr int QP_PARENT = 0xA0;     // [0, 0, 2, 2]: a lower link's lane reads its upper link's lane
constexpr int QP_CHILD = 0xF5;      // [1, 1, 3, 3]: an upper link's lane reads its lower link's lane
constexpr int QP_SWAP1 = 0xB1;      // [1, 0, 3, 2]
constexpr int QP_SWAP2 = 0x4E;      // [2, 3, 0, 1]
template <int P> constexpr int qp_bcast() { return P * 0x55; }   // every lane of the quad reads lane P

// A constant the compiler may not hoist out of the rollout loop: it is materialised into a scalar register pair where it is used (two s_mov).  Hoisted,
// the coefficient sets below sat in registers across the whole kernel, were spilled, and every timestep reloaded them from scratch memory one dependent
// round trip after the other (seen in the ISA: five serial scratch loads inside one sincos).
__device__ __forceinline__ double kc(double x) {
  asm volatile("" : "+s"(x));
  return x;
}
// sincos_mod of physics.hip with such constants (same operations in the same order: identical results)
__device__ __forceinline__ void sincos_kc(double x, double& sn, double& cs) {
  const double k = rint(x * kc(6.36619772367581382433e-01));
  double r = fma(-k, kc(1.57079632673412561417e+00), x);
  r = fma(-k, kc(6.07710050650619224932e-11), r);
  const double z = r * r;
  double ps = fma(z, kc(1.58969099521155010221e-10), kc(-2.50507602534068634195e-08));
  ps = fma(z, ps, kc(2.75573137070700676789e-06));
  ps = fma(z, ps, kc(-1.98412698298579493134e-04));
  ps = fma(z, ps, kc(8.33333333332248946124e-03));
  ps = fma(z, ps, kc(-1.66666666666666324348e-01));
  const double sr = fma(r * z, ps, r);
  double pc = fma(z, kc(-1.13596475577881948265e-11), kc(2.08757232129817482790e-09));
  pc = fma(z, pc, kc(-2.75573143513906633035e-07));
  pc = fma(z, pc, kc(2.48015872894767294178e-05));
  pc = fma(z, pc, kc(-1.38888888888741095749e-03));
  pc = fma(z, pc, kc(4.16666666666666019037e-02));
  const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
  const int q = (int)k & 3;
  sn = (q == 0) ? sr : ((q == 1) ? cr : ((q == 2) ? -sr : -cr));
  cs = (q == 0) ? cr : ((q == 1) ? -sr : ((q == 2) ? -cr : sr));
}

#ifdef EARL_MT_DEBUG
__device__ int g_mt_dbg[4096 * 8 * 32];          // per env, timestep (mod 8): [0] contacts, [1] warm, [2] passes, [4..15] the contacts' edge sets before the passes, [16..27] after
__device__ double g_mt_dbg_al[4096 * 8 * 32];    // ... and the solution
__device__ double g_mt_dbg_x[5][4096 * 8 * 32];  // per dof: 0 rw, 1 a weighted sum of Bw, 2 of Aw, 3 qv after the integration, 4 the external force of the timestep
__device__ double g_mt_dbg_ph[8][4096 * 8 * 32]; // per dof, a weighted sum of what each phase of the dynamics half leaves: 0 frames, 1 subspace + inertia, 2 composite inertia, 3 mass-matrix entries, 4 tau, 5 closure rows
#define DBG_PH(k, v) do { if (isl) g_mt_dbg_ph[k][((size_t)s.dbg_env * 8 + (s.dbg_ts & 7)) * 32 + l] = (v); } while (0)
#else
#define DBG_PH(k, v) do {} while (0)
#endif
struct MTDims {
  static constexpr int NV = 22, NR = 6, NLEG = 4, LS = 4, NH = 16, LPE = 32, MC = EARL_MAXCON, MB = 8;
};

// The collision model's pair records, once per workgroup in LDS (the generic kernels read them from global memory inside the timestep: one L2 round trip per
// near block -- 6.2 k of this model's 56 k cycles per timestep)
struct PairTabMT {
  static constexpr int MP = 64;          // (the model has 52 pairs; checked by the host side)
  double pos[MP][3], r[MP], margin[MP];
  int link[MP], cls[MP];
};
__device__ __forceinline__ void stage_pairs_mt(PairTabMT& t, const earl_collision_model* __restrict__ col) {
  const int i = threadIdx.x;
  if (col && i < PairTabMT::MP && i < col->n_pair) {
    t.pos[i][0] = col->pair_rec[i].pos[0]; t.pos[i][1] = col->pair_rec[i].pos[1]; t.pos[i][2] = col->pair_rec[i].pos[2];
    t.r[i] = col->pair_rec[i].r; t.margin[i] = col->pair_rec[i].margin;
    t.link[i] = col->pair_rec[i].sph_link; t.cls[i] = col->pair_rec[i].cls;
  }
}
// chol_regs of physics.hip (dense, diagonal left inverted) with the two-step reciprocal root
template <int N>
__device__ __forceinline__ void chol_small(double (&L)[N * (N + 1) / 2]) {
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double d = L[j * (j + 1) / 2 + j];
#pragma unroll
    for (int p = 0; p < j; ++p) d = fma(-L[j * (j + 1) / 2 + p], L[j * (j + 1) / 2 + p], d);
    const double inv = rsq2(d);
    L[j * (j + 1) / 2 + j] = inv;
#pragma unroll
    for (int i = j + 1; i < N; ++i) {
      double t = L[i * (i + 1) / 2 + j];
#pragma unroll
      for (int p = 0; p < j; ++p) t = fma(-L[i * (i + 1) / 2 + p], L[j * (j + 1) / 2 + p], t);
      L[i * (i + 1) / 2 + j] = t * inv;
    }
  }
}

// Per-env LDS block of the minitaur stepper.  (xt / kit keep the member names the env kernel uses with the generic block.)  The buffers of the dynamics
// phases (frames ... contact rows), of the active-set passes and of the env step's observation are live at disjoint times.
struct SharedMTData {
  static constexpr int NV = MTDims::NV, MC = MTDims::MC;
  double qp[NV], qv[NV];
  double bq[4];
  double aprev[NV];                  // solution of the newest solve (warm start of the next timestep; exchange buffer of the row tests)
  struct { double ext[NV]; double motor_volt, motor_visc; double mscale[3], foot_mu; } xt;      // (same fields as Shared<NV>::xt)
  double eres[12], eD[12], ear[12];  // closure rows: residual, weight, reference acceleration
  // minitaur_duo_kernel: what an env's step keeps between the visits of its wave (the one-wave kernel holds these in registers): motors 0 - 7 -- observed torque, the env step's
  // command, overheat counter, enabled flag --, the goal, the wrappers' counters
  struct { double obs_t[8], cmd[8]; int oh[8], en[8]; double goal[2]; int steps, sgc, pad_[2]; } ev;      // (a multiple of 16 bytes: what follows keeps its 16-byte alignment --
                                                                                                           // without it 220 of the kernel's 255 ds_read_b128 became ds_read2_b64, twice the LDS cycles: 178 -> 212 ms per bench launch)
  // ... and what the contact rows (C3, wave A's half) leave in the contact's LANE for the passes (wave B's half): weight, friction, the four edges' reference accelerations,
  // the warm-started active edges, the chain's hinges
  struct { double cD, cmu, car[4]; unsigned int cact; int cd1, cd2, pid; } c3[MC];
  unsigned int tb[NV][4];             // ... and per dof: which contact slots it takes part in (tbits, sbits, cbits of K9), worked out once per timestep by the first half
#ifdef EARL_MT_DEBUG
  int dbg_env, dbg_ts, dbg_pad[2];   // (tools/mt_duo_bisect.py, debug builds only: which env this block holds, timesteps since the launch began)
#endif
  union {
    double ct[MC][8];                // contact records (pair tests -> contact rows)
    double cw[MC][8];                // per-pass edge weights
  };
  double CJr[MC][3][6];              // contact Jacobians (normal, tangent 1, tangent 2): the root body's six entries ...
  double CJh[MC][3][2];              // ... and the (at most two) hinges of the sphere's own chain
  int crow[MC][2];                   // those hinges' dofs (-1: none)
  union {
    struct { double obs[46]; } kit;    // env step: observation (between timesteps)
    struct {
      double Xq[NV][4], Xp[NV][3];       // world frames (collision)
      double Sh[MTDims::NH][6];          // motion subspace of the hinges (contact Jacobians)
      double redI[9][10];                // composite inertias of the eight upper links + the root body's own: summed by every lane
      double redF[9][6];                 // the same for the bias forces
    } dyn;
    struct {
      // ROLE-split timestep (the two-waves-per-SIMD launch, physics_env_minitaur.h minitaur_duo_kernel): what wave A's half of the timestep (frames ... closure rows) hands
      // to wave B's half (contact rows, active-set passes, integration) besides what already lives in LDS -- this lane's entries of the equality Hessian and right-hand side.
      // It lies where the frames (dead after the pair tests) and the subtree sums (dead after K7) lay and leaves the motion subspaces alone, which the contact rows read.
      double Bw[NV][6];
      double rw[NV];
      double sh_keep[MTDims::NH][6];
      double Aw[MTDims::NH][4];
      int nct;
    } hand;
    struct {
      // the pass's Hessian, by blocks, and right-hand side
      double HA[MTDims::NLEG][10];       // leg blocks, packed lower triangle (p, q) at p (p + 1) / 2 + q
      double HB[MTDims::NH][6];          // couplings: row = hinge, column = root dof
      double HR[6][6];                   // root block (both triangles)
      double rc[NV];
      // exchange buffers of the solve
      double Wl[6][16];                  // W = B' L^-T, row = root dof
      double part[6][4];                 // W_k y_k per root dof and leg
      double Y[16];                      // y = L^-1 b of the legs
      double LL[MTDims::NLEG][10];       // the legs' Cholesky factors (diagonal inverted)
      double SS[22];                     // Schur complement of the root block, packed lower triangle (21)
      double xr0[6];                     // the root block's right-hand side: rc - sum of the legs' parts (six lanes of the Schur stage work it out, every lane reads it)
    } pas;
  };
};
static_assert(sizeof(((SharedMTData*)nullptr)->ev) % 16 == 0 && sizeof(((SharedMTData*)nullptr)->c3) % 16 == 0 && sizeof(((SharedMTData*)nullptr)->tb) % 16 == 0 && offsetof(SharedMTData, dyn) % 16 == 0, "16-byte alignment of the blocks the kernels read as b128");
static_assert(offsetof(SharedMTData, hand.sh_keep) == offsetof(SharedMTData, dyn.Sh) && offsetof(SharedMTData, hand.Aw) == offsetof(SharedMTData, dyn.redI) &&
              sizeof(((SharedMTData*)nullptr)->hand) <= sizeof(((SharedMTData*)nullptr)->dyn), "the hand-over block keeps clear of the motion subspaces and fits the dynamics block");
struct SharedMT : SharedMTData {
  static constexpr int R = (int)(sizeof(SharedMTData) % 256);
  static constexpr int PAD = R <= 64 ? 64 - R : (R <= 192 ? 192 - R : 320 - R);      // the two env blocks of a wave on different banks (physics.hip Shared<NV>)
  char bank_pad[PAD == 0 ? 16 : PAD];        // (never 8: a block size that is no multiple of 16 bytes turns every ds_read_b128 of the odd blocks into ds_read2_b64 -- twice the LDS cycles, 20 % of the kernel)
};
static_assert(sizeof(SharedMT) % 16 == 0, "an env block is a whole number of 16-byte words: the blocks of one wave all keep the alignment ds_read_b128 needs");

// K10 of the timestep: semi-implicit Euler from the solution `al` (no joint damping in this model: checked by the host side).  A function of its own since round 6: the one-wave
// form calls it at the end of the timestep, the two-wave form's FIRST-half wave calls it for the timestep before, at the head of its next visit of the env (the solution waits in
// s.aprev): 1.7 k cycles off the longer half.  Same expressions on the same values either way.
__device__ __forceinline__ void integrate_mt(SharedMT& s, const earl_link_model24& m, const int sub, const bool isl, const int l, const double dt, const double al,
                                             const double qd, const double ql_, const Q4 Qb) {
  // ---------------------------------------------------------------- K10: semi-implicit Euler (no joint damping in this model: checked by the host side)
  fence();
  if (isl) {
    // (explicit fused multiply-adds: left to fp contract(fast) the one-wave and the two-wave instantiations of this function chose differently here -- a position that
    // differs in its last bit about once in 10^4 env steps was the only thing that told them apart: tools/mt_duo_bisect.py)
    const double nv_ = fma(dt, al, qd);
    s.qv[l] = nv_;
    s.qp[l] = fma(dt, nv_, ql_);                        // (unused for the rotation dofs of the root body)
  }
  fence();
  {
    const int bd = m.ball_dof;
    const V3 wbd{s.qv[bd], s.qv[bd + 1], s.qv[bd + 2]};
    // The orientation update with every sum of products written as an EXPLICIT chain of fused multiply-adds (norm2, mul4 below): left to fp contract(fast), the one-wave
    // and the two-wave instantiations of this function fused these sums differently, and the base quaternion's last bit -- about once in 10^4 env steps -- was all that told
    // their results apart (tools/mt_duo_bisect.py: located phase by phase, then component by component).
    auto norm2 = [](const Q4& q) { return fma(q.z, q.z, fma(q.y, q.y, fma(q.x, q.x, q.w * q.w))); };
    auto mul4 = [](const Q4& a, const Q4& b) {
      return Q4{fma(-a.z, b.z, fma(-a.y, b.y, fma(-a.x, b.x, a.w * b.w))), fma(-a.z, b.y, fma(a.y, b.z, fma(a.x, b.w, a.w * b.x))),
                fma(a.z, b.x, fma(a.y, b.w, fma(-a.x, b.z, a.w * b.y))), fma(a.z, b.w, fma(-a.y, b.x, fma(a.x, b.y, a.w * b.z)))};
    };
    Q4 q0 = Qb;
    const double n0 = rsq2(norm2(q0));
    q0 = Q4{q0.w * n0, q0.x * n0, q0.y * n0, q0.z * n0};
    const double w2 = fma(wbd.z, wbd.z, fma(wbd.y, wbd.y, wbd.x * wbd.x));
    const double iw = w2 > 0 ? rsq2(w2 > 0 ? w2 : 1.0) : 0.0;
    double sn, cs;
    sincos_kc(0.5 * dt * (w2 * iw), sn, cs);
    const Q4 q1 = mul4(q0, Q4{cs, sn * wbd.x * iw, sn * wbd.y * iw, sn * wbd.z * iw});
    const double n1 = rsq2(norm2(q1));
    if (sub == 0) { s.bq[0] = q1.w * n1; s.bq[1] = q1.x * n1; s.bq[2] = q1.y * n1; s.bq[3] = q1.z * n1; }
    fence();
  }
}

// One timestep of one env by its 32-lane group.  Lane roles: sub 0-5 = the root body's dofs, sub 8 + 4 k + j = hinge j of leg k (dof 6 + 4 k + j),
// the other lanes idle (they shadow a hinge and store nothing).  INTEGRATE = false stops after qacc.
// ROLE 0: the whole timestep in this wave.  ROLE 1 / 2 (minitaur_duo_kernel): the timestep in two halves run by two waves, one after the other on the same LDS block --
// 1 = frames, bounding and pair tests, mass matrix, bias forces, closure rows, contact rows (K1 - K8, C0 - C3), handing over this lane's equality entries (s.hand) and
// its contact's row data (s.c3); 2 = active-set passes and integration (K9, K10), beginning with that hand-over.  Same expressions on the same values in the same order: same results.
template <bool INTEGRATE, int ROLE = 0>
__device__ __forceinline__ void substep_mt(SharedMT& s, const earl_link_model24& m, const BlkTable<MTDims::MB, true>& bt, const PairTabMT& pt,
                                           const int sub, const int grp, const bool warm, double* qacc_out) {
  constexpr int NV = MTDims::NV, MC = MTDims::MC, LPE = MTDims::LPE;
  const int maxcon = bt.max_con < MC ? bt.max_con : MC;
  const double dt = m.dt;
  const bool isroot = sub < 6, ishinge = sub >= 8 && sub < 24, isl = isroot || ishinge;
  const int l = isroot ? sub : (ishinge ? sub - 2 : NV - 1);      // this lane's dof
  const int hq = (sub - 8) & 3;                                    // hinge lanes: position in the leg
  const int leg = ishinge ? (sub - 8) >> 2 : 0;
  const bool lower = ishinge && (hq & 1), upper = ishinge && !(hq & 1);
  PSTART();
  // ------------------------------------------------------------------ K1-K2: world frames, in registers
  const V3 Pb = ld3(s.qp);
  const Q4 Qb = ldq(s.bq);
  double Rb[3][3];
  qmat(Qb, Rb);
  Q4 Q; V3 P;
  const V3 ax = ld3(m.jaxis[l]);
  const int jt = m.jtype[l];
  const double ql_ = s.qp[l], qd = s.qv[l];
  double Bw[6], Aw[4], rw;                                // this lane's entries of the equality Hessian and right-hand side (K8 -> K9)
  int nct = 0, ncmax = 0;                                 // contacts of this lane's env, of the wave's two envs at most (C1-2 -> C3, K9)
  if constexpr (ROLE != 2) {
  {
    const Q4 tq = ldq(m.tquat[l]);
    const V3 tp = ld3(m.tpos[l]);
    double sn, cs;
    sincos_kc(jt == 0 ? 0.5 * ql_ : 0.0, sn, cs);
    const Q4 qloc = qmul(tq, Q4{cs, sn * ax.x, sn * ax.y, sn * ax.z});
    const Q4 Qu = qmul(Qb, qloc);                                   // as a link hanging off the root body
    const V3 Pu = add(Pb, mulv(Rb, tp));
    const Q4 Qp = dpp_quad<QP_PARENT>(Qu);                          // a lower link: its upper link's frame from the lane before
    const V3 Pp = dpp_quad<QP_PARENT>(Pu);
    double Rp[3][3];
    qmat(Qp, Rp);
    const Q4 Qw = qmul(Qp, qloc);
    const V3 Pw = add(Pp, mulv(Rp, tp));
    // root body: three slides along the world axes (identity orientation, the translation so far), then the orientation quaternion
    const Q4 Qr = selq(sub < 3, Q4{1, 0, 0, 0}, Qb);
    const V3 Pr{Pb.x, sub >= 1 ? Pb.y : 0.0, sub >= 2 ? Pb.z : 0.0};
    Q = selq(isroot, Qr, selq(lower, Qw, Qu));
    P = selv(isroot, Pr, selv(lower, Pw, Pu));
  }
#ifdef EARL_MT_DEBUG_FRAMES
  DBG_PH(0, Q.w); DBG_PH(1, Q.x); DBG_PH(2, Q.y); DBG_PH(3, Q.z); DBG_PH(4, P.x); DBG_PH(5, P.y); DBG_PH(6, P.z); DBG_PH(7, ql_);
#else
  DBG_PH(0, Q.w + 2 * Q.x + 3 * Q.y + 5 * Q.z + 7 * P.x + 11 * P.y + 13 * P.z);
  DBG_PH(5, ql_); DBG_PH(6, Qb.w + 2 * Qb.x + 3 * Qb.y + 5 * Qb.z + 7 * Pb.x + 11 * Pb.y + 13 * Pb.z); DBG_PH(7, qd);
#endif
  if (isl) {
    double* oq = s.dyn.Xq[l];
    double* op = s.dyn.Xp[l];
    oq[0] = Q.w; oq[1] = Q.x; oq[2] = Q.y; oq[3] = Q.z; op[0] = P.x; op[1] = P.y; op[2] = P.z;
  }
  fence();
  // ------------------------------------------------------------------ C0: collision bounding tests (lane = block; world-fixed boxes)
  unsigned int nearw = 0, nearg = 0;
  {
    const int b = sub < bt.n_blk ? sub : 0;
    const int bl = bt.link[b];
    V3 cs = ld3(bt.center[b]);
    {
      double Rl[3][3];
      qmat(ldq(s.dyn.Xq[bl < 0 ? 0 : bl]), Rl);
      cs = selv(bl < 0, cs, add(ld3(s.dyn.Xp[bl < 0 ? 0 : bl]), mulv(Rl, cs)));
    }
    double Rbx[3][3];
    qmat(ldq(bt.box_quat[b]), Rbx);
    const V3 x = mulvT(Rbx, vsub(cs, ld3(bt.box_pos[b]))), h = ld3(bt.box_half[b]);
    const V3 d{x.x - fmin(fmax(x.x, -h.x), h.x), x.y - fmin(fmax(x.y, -h.y), h.y), x.z - fmin(fmax(x.z, -h.z), h.z)};
    const bool nearb = sub < bt.n_blk && dot(d, d) < bt.reach[b] * bt.reach[b];
    const unsigned long long bal = __ballot(nearb);
    nearg = (unsigned int)((bal >> (grp * 32)) & 0xFFFFFFFFull);
    nearw = (unsigned int)((bal | (bal >> 32)) & 0xFFFFFFFFull);
  }
  PSTAMP(0);
  // ------------------------------------------------------------------ K3: motion subspace + spatial inertia of this lane's link
  V3 Sw, Sv;
  double I10[10];
  double R[3][3];
  qmat(Q, R);
  {
    const V3 aw = mulv(R, ax);
    const V3 anchor = add(P, mulv(R, ld3(m.jpos[l])));
    const bool rot = jt != 1;
    Sw = selv(rot, aw, V3{0, 0, 0});
    Sv = selv(rot, cross(anchor, aw), aw);
    const int root = m.ball_dof + 2;
    const double ms = l < root ? 1.0 : (l == root ? s.xt.mscale[0] : (m.parent[l] == root ? s.xt.mscale[1] : s.xt.mscale[2]));
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
    I10[0] = mass;
    I10[1] = mass * c.x; I10[2] = mass * c.y; I10[3] = mass * c.z;
    I10[4] = W[0][0] + mass * (c2 - c.x * c.x);
    I10[5] = W[1][1] + mass * (c2 - c.y * c.y);
    I10[6] = W[2][2] + mass * (c2 - c.z * c.z);
    I10[7] = W[0][1] - mass * c.x * c.y;
    I10[8] = W[0][2] - mass * c.x * c.z;
    I10[9] = W[1][2] - mass * c.y * c.z;
    if (ishinge) {
      double* o = s.dyn.Sh[l - 6];
      o[0] = Sw.x; o[1] = Sw.y; o[2] = Sw.z; o[3] = Sv.x; o[4] = Sv.y; o[5] = Sv.z;
    }
  }
  
#ifndef EARL_MT_DEBUG_FRAMES
  DBG_PH(1, Sw.x + 2 * Sw.y + 3 * Sw.z + 5 * Sv.x + 7 * Sv.y + 11 * Sv.z + 13 * I10[0] + 17 * I10[1] + 19 * I10[2] + 23 * I10[3] + 29 * I10[4] + 31 * I10[5] + 37 * I10[6] + 41 * I10[7] + 43 * I10[8] + 47 * I10[9]);
#endif
  PSTAMP(1);
  // ------------------------------------------------------------------ K4: composite inertia (own + child by DPP; the root body: everything, through LDS)
  double Ic[10];
#pragma unroll
  for (int e = 0; e < 10; ++e) {
    const double ch = dpp_quad<QP_CHILD>(I10[e]);
    Ic[e] = upper ? I10[e] + ch : I10[e];
  }
  if (upper || sub == 5) {
    double* o = s.dyn.redI[upper ? (l - 6) >> 1 : 8];
#pragma unroll
    for (int e = 0; e < 10; ++e) o[e] = Ic[e];
  }
  fence();
  {
    double tot[10];
#pragma unroll
    for (int e = 0; e < 10; ++e) tot[e] = s.dyn.redI[8][e];
#ifndef EARL_MT_NO_CHUNK
#define EARL_MT_NO_CHUNK 0       // (measurement: 1 = the role builds sum the subtrees like the one-wave build and spill)
#endif
    if constexpr (ROLE == 0 || EARL_MT_NO_CHUNK) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
#pragma unroll
        for (int e = 0; e < 10; ++e) tot[e] += s.dyn.redI[u][e];
    } else {
      // (256 registers: all ninety loads at once -- what the scheduler does with the loop above -- end in scratch memory; two subtrees per LDS round trip, same sums in the same order)
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        double part[20];
#pragma unroll
        for (int e = 0; e < 10; ++e) { part[e] = s.dyn.redI[u][e]; part[10 + e] = s.dyn.redI[u + 1][e]; }
        pin_batch(part);
#pragma unroll
        for (int e = 0; e < 10; ++e) tot[e] += part[e];
#pragma unroll
        for (int e = 0; e < 10; ++e) tot[e] += part[10 + e];
        pin_batch(tot);
      }
    }
#pragma unroll
    for (int e = 0; e < 10; ++e) Ic[e] = isroot ? tot[e] : Ic[e];
  }
  
#ifndef EARL_MT_DEBUG_FRAMES
  DBG_PH(2, Ic[0] + 2 * Ic[1] + 3 * Ic[2] + 5 * Ic[3] + 7 * Ic[4] + 11 * Ic[5] + 13 * Ic[6] + 17 * Ic[7] + 19 * Ic[8] + 23 * Ic[9]);
#endif
  PSTAMP(3);
  // ------------------------------------------------------------------ K5: this lane's entries of the mass matrix
  // g6 = row l of the six root columns: M[l][r] = S_r . (Ic_l S_l) = [f ; Rb' (n + f x Pb)] with [n; f] = Ic_l S_l  (hinge: its row of B; root dof: its row of R);
  // hinge lanes: the diagonal entry and, an upper link, the entry it shares with its lower link
  double g6[6], Add, Ach;
  {
    V3 n, f;
    iapply(Ic, Sw, Sv, n, f);
    const V3 mo = mulvT(Rb, add(n, cross(f, Pb)));
    g6[0] = f.x; g6[1] = f.y; g6[2] = f.z; g6[3] = mo.x; g6[4] = mo.y; g6[5] = mo.z;
    Add = dot(Sw, n) + dot(Sv, f) + m.armature[l];
    const V3 nc = dpp_quad<QP_CHILD>(n), fc = dpp_quad<QP_CHILD>(f);
    Ach = dot(Sw, nc) + dot(Sv, fc);
  }
  
#ifndef EARL_MT_DEBUG_FRAMES
  DBG_PH(3, g6[0] + 2 * g6[1] + 3 * g6[2] + 5 * g6[3] + 7 * g6[4] + 11 * g6[5] + 13 * Add + 17 * Ach);
#endif
  PSTAMP(4);
  // ------------------------------------------------------------------ K6-K7: bias forces
  const V3 vlin = ld3(s.qv), om = ld3(s.qv + 3);
  const V3 wb = mulv(Rb, om);                            // the root body's velocity [wb; vb] = sum of its six dofs' S qd
  const V3 vb = add(vlin, cross(Pb, wb));
  double tau_l;
  {
    const V3 wu = add(wb, scl(Sw, qd)), vu = add(vb, scl(Sv, qd));
    const V3 wpar = dpp_quad<QP_PARENT>(wu), vpar = dpp_quad<QP_PARENT>(vu);
    const V3 w = selv(isroot, wb, selv(lower, add(wpar, scl(Sw, qd)), wu));
    const V3 v = selv(isroot, vb, selv(lower, add(vpar, scl(Sv, qd)), vu));
    // d/dt of the axis: hinges use their link's velocity (the own term cancels), the root body's three rotation axes the velocity before any of them
    // (mj_comVel), the slides have none
    const V3 cwh = scl(cross(w, Sw), qd), cvh = scl(add(cross(v, Sw), cross(w, Sv)), qd);
    const V3 cw = selv(ishinge, cwh, V3{0, 0, 0});
    const V3 cv = selv(ishinge, cvh, scl(cross(vlin, Sw), qd));           // (slides: Sw = 0)
    const V3 ab{-m.gravity[0], -m.gravity[1], -m.gravity[2]};
    const V3 avb = add(ab, cross(vlin, wb));                                // root body: gravity + the three rotation dofs' terms
    const V3 awu = cw, avu = add(avb, cv);
    const V3 awp = dpp_quad<QP_PARENT>(awu), avp = dpp_quad<QP_PARENT>(avu);
    const V3 aw = selv(isroot, V3{0, 0, 0}, selv(lower, add(awp, cw), awu));
    const V3 av = selv(isroot, avb, selv(lower, add(avp, cv), avu));
    V3 n1, f1, n2, f2;
    iapply(I10, aw, av, n1, f1);
    iapply(I10, w, v, n2, f2);
    const V3 n = add(n1, add(cross(w, n2), cross(v, f2)));
    const V3 f = add(f1, cross(w, f2));
    const V3 nch = dpp_quad<QP_CHILD>(n), fch = dpp_quad<QP_CHILD>(f);
    V3 ns = selv(upper, add(n, nch), n), fs = selv(upper, add(f, fch), f);
    if (upper || sub == 5) {
      double* o = s.dyn.redF[upper ? (l - 6) >> 1 : 8];
      o[0] = ns.x; o[1] = ns.y; o[2] = ns.z; o[3] = fs.x; o[4] = fs.y; o[5] = fs.z;
    }
    fence();
    V3 nt = ld3(s.dyn.redF[8]), ft = ld3(s.dyn.redF[8] + 3);
    if constexpr (ROLE == 0 || EARL_MT_NO_CHUNK) {
#pragma unroll
      for (int u = 0; u < 8; ++u) { nt = add(nt, ld3(s.dyn.redF[u])); ft = add(ft, ld3(s.dyn.redF[u] + 3)); }
    } else {
      // (256 registers: four subtrees per LDS round trip instead of all eight loaded at once and spilled; same sums in the same order)
#pragma unroll
      for (int u = 0; u < 8; u += 4) {
        double part[24];
#pragma unroll
        for (int k = 0; k < 24; ++k) part[k] = s.dyn.redF[u + k / 6][k % 6];
        pin_batch(part);
#pragma unroll
        for (int w = 0; w < 4; ++w) { nt = add(nt, V3{part[6 * w], part[6 * w + 1], part[6 * w + 2]}); ft = add(ft, V3{part[6 * w + 3], part[6 * w + 4], part[6 * w + 5]}); }
        nt = V3{pinned(nt.x), pinned(nt.y), pinned(nt.z)}; ft = V3{pinned(ft.x), pinned(ft.y), pinned(ft.z)};
      }
    }
    ns = selv(isroot, nt, ns); fs = selv(isroot, ft, fs);
    tau_l = -m.damping[l] * qd - (dot(Sw, ns) + dot(Sv, fs)) + s.xt.ext[l];
  }
  
#ifndef EARL_MT_DEBUG_FRAMES
  DBG_PH(4, tau_l);
#endif
  PSTAMP(5);
  // ------------------------------------------------------------------ C1-C2: pair tests of the near blocks (lane = pair; spheres vs world-fixed boxes)
  if (nearw) {
    // PACKED (round 5; the form of the kitchen's C2 in physics.hip): consecutive near blocks of the wave share a pass as long as their pairs fit the group's 32 lanes --
    // lane -> (block, pair) by a walk over the pass's blocks, the block's box frame per lane.  Contacts keep the sequential order (blocks ascending on the lanes, pairs
    // ascending within a block) and the per-block caps, so the contact list is the one a block-per-pass loop builds.  The model's seven blocks hold 4 + 6 x 8 pairs: the
    // three blocks under the feet, near in almost every timestep, were three passes of two LDS round trips each (5.9 k cycles per timestep) and are one now.
    unsigned int rest = nearw;
    while (rest) {
      int myb = -1, myoff = 0, used = 0;
      unsigned int taken = 0;
      for (unsigned int r2 = rest; r2; r2 &= r2 - 1u) {
        const int b = __builtin_ctz(r2);
        const int sz = bt.end[b] - bt.begin[b];
        if (used + sz > LPE) { if (used == 0) { taken = 1u << b; used = sz; if (sub < sz) { myb = b; myoff = 0; } } break; }      // (a block of more than 32 pairs goes alone; its tail is not tested: the host side refuses such tables)
        if (sub >= used && sub < used + sz) { myb = b; myoff = used; }
        used += sz;
        taken |= 1u << b;
      }
      rest &= ~taken;
      const bool has = myb >= 0;
      const int b = has ? myb : 0;
      const bool mine = has && ((nearg >> b) & 1u);
      const int bsz = bt.end[b] - bt.begin[b];
      const V3 pb = ld3(bt.box_pos[b]);
      double Rbx[3][3];
      qmat(ldq(bt.box_quat[b]), Rbx);
      const V3 h = ld3(bt.box_half[b]);
      const int room = bt.cap[b] & 255;
      const int pi = has ? bt.begin[b] + (sub - myoff < bsz ? sub - myoff : bsz - 1) : 0;
      const bool valid = mine && sub - myoff < bsz;
      const int lk = pt.link[pi], cls = pt.cls[pi];
      const double r = pt.r[pi], margin = pt.margin[pi];
      V3 c = ld3(pt.pos[pi]);
      {
        double Rl[3][3];
        qmat(ldq(s.dyn.Xq[lk < 0 ? 0 : lk]), Rl);
        c = selv(lk < 0, c, add(ld3(s.dyn.Xp[lk < 0 ? 0 : lk]), mulv(Rl, c)));
      }
      const V3 x = mulvT(Rbx, vsub(c, pb));
      V3 q{fmin(fmax(x.x, -h.x), h.x), fmin(fmax(x.y, -h.y), h.y), fmin(fmax(x.z, -h.z), h.z)};
      const bool outside = fabs(x.x) > h.x || fabs(x.y) > h.y || fabs(x.z) > h.z;
      const V3 dd = vsub(x, q);
      const double d2 = dot(dd, dd);
      const double inv = rsq2(outside ? d2 : 1.0);
      const double gx = h.x - fabs(x.x), gy = h.y - fabs(x.y), gz = h.z - fabs(x.z);
      const int axn = (gx <= gy && gx <= gz) ? 0 : (gy <= gz ? 1 : 2);
      const double xa = pick3(x, axn), ha = pick3(h, axn), sg = xa >= 0 ? 1.0 : -1.0;
      const V3 ni{axn == 0 ? sg : 0.0, axn == 1 ? sg : 0.0, axn == 2 ? sg : 0.0};
      const V3 qi{axn == 0 ? sg * ha : x.x, axn == 1 ? sg * ha : x.y, axn == 2 ? sg * ha : x.z};
      const double dist = outside ? d2 * inv - r : -(ha - fabs(xa)) - r;
      const V3 nl = selv(outside, scl(dd, inv), ni);
      q = selv(outside, q, qi);
      const bool hit = valid && dist < margin;
      const unsigned long long bal = __ballot(hit);
      const unsigned int gb = (unsigned int)((bal >> (grp * 32)) & 0xFFFFFFFFull);
      const unsigned int seg = has ? (unsigned int)(((1ull << bsz) - 1ull) << myoff) : 0u;      // the lanes of this lane's block
      const int before_blk = __popc(gb & seg & ((1u << sub) - 1u));
      const bool accept = hit && before_blk < room;
      const unsigned long long bal2 = __ballot(accept);
      const unsigned int ga = (unsigned int)((bal2 >> (grp * 32)) & 0xFFFFFFFFull);
      const int slot = nct + __popc(ga & ((1u << sub) - 1u));
      if (accept && slot < maxcon) {
        const V3 n = mulv(Rbx, nl);
        const V3 p = add(add(pb, mulv(Rbx, q)), scl(n, 0.5 * dist));
        double* o = s.ct[slot];
        o[0] = dist; o[1] = n.x; o[2] = n.y; o[3] = n.z; o[4] = p.x; o[5] = p.y; o[6] = p.z;
        o[7] = (double)(cls + 64 * (lk + 1) + 4096 * pi);      // (+ the collision pair: what C3 recognises a slot's contact of the previous timestep by)
      }
      const int took = __popc(ga);
      nct = nct + took < maxcon ? nct + took : maxcon;
    }
    fence();
  }
  PSTAMP(2);
  PCOUNT(20, 1); PCOUNT(21, nearw ? 1 : 0); PCOUNT(22, __popc(nearw));
  if (nearw && __any(nct > 0)) {
#pragma unroll
    for (int k = 0; k < MC; ++k) ncmax = __any(nct > k) ? k + 1 : ncmax;
  }
  // ------------------------------------------------------------------ K8: closure rows (connect constraints; leg k carries constraint k: attachment 1 on its
  // hinge 3, attachment 2 on its hinge 1) and the equality part of the Hessian -- this lane's entries, kept in registers over the passes
  {
    const int k1 = m.con_att1[leg], k2 = m.con_att2[leg];
    const V3 pa = add(P, mulv(R, ld3(m.att_pos[hq == 3 ? k1 : k2])));
    const V3 p1 = dpp_quad<qp_bcast<3>()>(pa), p2 = dpp_quad<qp_bcast<1>()>(pa);
    const V3 d = vsub(p1, p2);
    const double w1 = (ishinge && hq >= 2) ? 1.0 : 0.0, w2 = (ishinge && hq < 2) ? 1.0 : 0.0;
    const V3 JE = vsub(scl(add(Sv, cross(Sw, p1)), w1), scl(add(Sv, cross(Sw, p2)), w2));      // this hinge's column of the closure's three rows
    // J qvel of the three rows: the leg's four hinges (summed over the quad) + the root body (its columns are Sw_r x d, so their sum is wb x d)
    V3 jv = scl(JE, qd);
    jv = add(jv, dpp_quad<QP_SWAP1>(jv));
    jv = add(jv, dpp_quad<QP_SWAP2>(jv));
    jv = add(jv, cross(wb, d));
    const int c = hq < 3 ? hq : 2;                          // hinge j < 3 works out row j
    const double res = pick3(d, c);
    double kk, bb;
    kb_of(m.con_solref[leg], m.con_solimp[leg], dt, kk, bb);
    const double dd = imp_p2(m.con_solimp[leg], res);
    const double eDm = rcp_nr(fmax((1 - dd) * m.con_invweight[leg] * rcp_nr(dd), 1e-15));
    const double earm = -bb * pick3(jv, c) - kk * dd * res;
    if (ishinge && hq < 3) { s.eres[3 * leg + hq] = res; s.eD[3 * leg + hq] = eDm; s.ear[3 * leg + hq] = earm; }
    const V3 eDv{dpp_quad<qp_bcast<0>()>(eDm), dpp_quad<qp_bcast<1>()>(eDm), dpp_quad<qp_bcast<2>()>(eDm)};
    const V3 earv{dpp_quad<qp_bcast<0>()>(earm), dpp_quad<qp_bcast<1>()>(earm), dpp_quad<qp_bcast<2>()>(earm)};
    const V3 dj{eDv.x * JE.x, eDv.y * JE.y, eDv.z * JE.z};
    // hinge lanes: row of B (mass matrix + closure: sum_c (Sw_r x d)_c dj_c = Sw_r . (d x dj)), column of the leg block
    const V3 bx = mulvT(Rb, cross(d, dj));
    double Bh[6] = {g6[0], g6[1], g6[2], g6[3] + bx.x, g6[4] + bx.y, g6[5] + bx.z};
    double Ah[4];
    {
      const V3 j0 = dpp_quad<qp_bcast<0>()>(JE), j1 = dpp_quad<qp_bcast<1>()>(JE), j2 = dpp_quad<qp_bcast<2>()>(JE), j3 = dpp_quad<qp_bcast<3>()>(JE);
      Ah[0] = dot(j0, dj); Ah[1] = dot(j1, dj); Ah[2] = dot(j2, dj); Ah[3] = dot(j3, dj);
#pragma unroll
      for (int p = 0; p < 4; ++p) Ah[p] += p == hq ? Add : ((upper && p == hq + 1) ? Ach : 0.0);
    }
    const double gh = tau_l + dot(dj, earv);
    fence();                                                // eres / eD / ear of all four legs
    // root lanes: row of R, right-hand side
    V3 T{0, 0, 0}, G{0, 0, 0};
#pragma unroll
    for (int e = 0; e < MTDims::NLEG; ++e) {
      const V3 de = ld3(s.eres + 3 * e), De = ld3(s.eD + 3 * e), are = ld3(s.ear + 3 * e);
      const V3 sd = cross(Sw, de);
      const V3 u{De.x * sd.x, De.y * sd.y, De.z * sd.z};
      T = add(T, cross(de, u));
      G = add(G, cross(de, V3{De.x * are.x, De.y * are.y, De.z * are.z}));
    }
    const V3 rx = mulvT(Rb, T);
    double Br[6] = {g6[0], g6[1], g6[2], g6[3] + rx.x, g6[4] + rx.y, g6[5] + rx.z};
#pragma unroll
    for (int i = 0; i < 6; ++i) Br[i] += sub == i ? m.armature[l] : 0.0;
    const double gr = tau_l + dot(Sw, G);
#pragma unroll
    for (int i = 0; i < 6; ++i) Bw[i] = isroot ? Br[i] : Bh[i];
#pragma unroll
    for (int p = 0; p < 4; ++p) Aw[p] = Ah[p];
    rw = isroot ? gr : gh;
  }
  PSTAMP(7);
  }   // ROLE != 2
  if constexpr (ROLE == 1) {
    // hand-over to the wave that runs the second half (it waits at a workgroup barrier): this lane's dof's entries, the env's contact count
    if (isl) {
#pragma unroll
      for (int i = 0; i < 6; ++i) s.hand.Bw[l][i] = Bw[i];
      s.hand.rw[l] = rw;
    }
    if (ishinge) {
#pragma unroll
      for (int p = 0; p < 4; ++p) s.hand.Aw[l - 6][p] = Aw[p];
    }
    if (sub == 0) s.hand.nct = nct;
  }
  if constexpr (ROLE == 0) {
    // the one-wave form holds the same values in registers HERE, opaque to the optimiser as a hand-over through LDS is: what the compiler may fuse or share across this point
    // (fp contract(fast)) is then the same in the one-wave and the two-wave forms -- they walk through the same bits
    pin_batch(Bw); pin_batch(Aw); rw = pinned(rw);
  }
  if constexpr (ROLE == 2) {
    nct = s.hand.nct;
    if (__any(nct > 0)) {
#pragma unroll
      for (int k = 0; k < MC; ++k) ncmax = __any(nct > k) ? k + 1 : ncmax;
    }
  }
  // ------------------------------------------------------------------ C3: contact rows, lane = contact (all contacts of the env at once)
  double cD = 0, cmu = 0, car[4] = {0, 0, 0, 0};
  unsigned int cact = 0;
  int cd1 = -1, cd2 = -1;                               // lane c: the hinges of its contact's chain
  int cpid = -1;                                        // ... and its collision pair (-1: the slot is empty)
  if constexpr (ROLE == 2) {
    const int c = sub < MC ? sub : MC - 1;
    cD = s.c3[c].cD; cmu = s.c3[c].cmu;
#pragma unroll
    for (int k = 0; k < 4; ++k) car[k] = s.c3[c].car[k];
    cact = s.c3[c].cact; cd1 = s.c3[c].cd1; cd2 = s.c3[c].cd2;
  }
  if (ROLE != 2 && ncmax > 0) {
    const int c = sub < MC ? sub : MC - 1;
    const bool cv = sub < nct;
    // the slot's record of the timestep before (its pair, the edge set its passes ENDED with): s.c3 keeps them between the timesteps of an env step
    unsigned int pact = s.c3[c].cact;
    int ppid = s.c3[c].pid;
    double rec[8];                                        // (a lane reads its own record only, and later writes its own edge weights over it: no exchange.  Slots
#pragma unroll                                            // beyond the env's count hold whatever LDS held: taken as zeros -- tests/test_lds_hygiene_gpu.py)
    for (int k = 0; k < 8; ++k) rec[k] = s.ct[c][k];
    // (loads in three batches, by what their addresses depend on -- the record; the chain's parent and the class's tables; the chain's motion subspaces and velocities --
    // each batch one LDS round trip: physics_math.h pin_batch.  Selected per load, they were eleven round trips one after the other)
    pin_batch(rec);
    asm volatile("" : "+v"(pact), "+v"(ppid));
#pragma unroll
    for (int k = 0; k < 8; ++k) rec[k] = cv ? rec[k] : 0.0;
    const V3 n = selv(cv, V3{rec[1], rec[2], rec[3]}, V3{0, 0, 1}), p = selv(cv, V3{rec[4], rec[5], rec[6]}, V3{0, 0, 0});
    const int pk = cv ? (int)rec[7] : 0;
    const int cls = pk & 63, ls = ((pk >> 6) & 63) - 1;
    cpid = cv ? (pk >> 12) : -1;
    const double ax_ = fabs(n.x), ay_ = fabs(n.y), az_ = fabs(n.z);
    const int ia = (ax_ <= ay_ && ax_ <= az_) ? 0 : (ay_ <= az_ ? 1 : 2);
    const V3 e{ia == 0 ? 1.0 : 0.0, ia == 1 ? 1.0 : 0.0, ia == 2 ? 1.0 : 0.0};
    V3 t1 = cross(n, e);
    t1 = scl(t1, rsq2(dot(t1, t1)));
    const V3 t2 = cross(n, t1);
    cd2 = ls >= 6 ? ls : -1;
    double ctab[6] = {bt.cls_margin[cls], bt.cls_mu[cls], bt.kb_cls[cls][0], bt.kb_cls[cls][1], bt.cls_invw[cls], s.xt.foot_mu};
    double simp[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) simp[k] = bt.cls_solimp[cls][k];
    int par2 = m.parent[cd2 >= 0 ? cd2 : 0], parls = m.parent[ls > 0 ? ls : 0];
    asm volatile("" : "+v"(par2), "+v"(parls));
    pin_batch(ctab); pin_batch(simp);
    cd1 = (cd2 >= 0 && par2 >= 6) ? par2 : -1;
    // Jacobian of the contact point, direction t: root dofs [t ; Rb' ((p - Pb) x t)], hinge d: t . (Sv_d + Sw_d x p)
    const V3 rp = vsub(p, Pb);
    const V3 dirs[3] = {n, t1, t2};
    double Jr[3][6], Jh[3][2];
    double sh1[6], sh2[6], qa[4] = {s.qv[cd1 >= 0 ? cd1 : 0], s.qv[cd2 >= 0 ? cd2 : 0], s.aprev[cd1 >= 0 ? cd1 : 0], s.aprev[cd2 >= 0 ? cd2 : 0]};
#pragma unroll
    for (int k = 0; k < 6; ++k) { sh1[k] = s.dyn.Sh[cd1 >= 0 ? cd1 - 6 : 0][k]; sh2[k] = s.dyn.Sh[cd2 >= 0 ? cd2 - 6 : 0][k]; }
    double qr[6], ar[6];
#pragma unroll
    for (int r = 0; r < 6; ++r) { qr[r] = s.qv[r]; ar[r] = s.aprev[r]; }
    pin_batch(sh1); pin_batch(sh2); pin_batch(qa); pin_batch(qr); pin_batch(ar);
    const double* s1 = sh1;
    const double* s2 = sh2;
    const V3 jp1 = add(ld3(s1 + 3), cross(ld3(s1), p)), jp2 = add(ld3(s2 + 3), cross(ld3(s2), p));
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const V3 t = dirs[k];
      const V3 mo = mulvT(Rb, cross(rp, t));
      Jr[k][0] = cv ? t.x : 0.0; Jr[k][1] = cv ? t.y : 0.0; Jr[k][2] = cv ? t.z : 0.0;
      Jr[k][3] = cv ? mo.x : 0.0; Jr[k][4] = cv ? mo.y : 0.0; Jr[k][5] = cv ? mo.z : 0.0;
      Jh[k][0] = cd1 >= 0 ? dot(t, jp1) : 0.0;
      Jh[k][1] = cd2 >= 0 ? dot(t, jp2) : 0.0;
    }
    if (sub < MC) {                                       // (slots beyond the env's count are written too, as zeros: the passes read every slot up to ncmax)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int r = 0; r < 6; ++r) s.CJr[c][k][r] = Jr[k][r];
        s.CJh[c][k][0] = Jh[k][0]; s.CJh[c][k][1] = Jh[k][1];
      }
      s.crow[c][0] = cd1; s.crow[c][1] = cd2;
    }
    // J qvel and J a_prev (warm start)
    double jv[3], jp[3];
    {
      const double q1 = cd1 >= 0 ? qa[0] : 0.0, q2 = cd2 >= 0 ? qa[1] : 0.0;
      const double a1 = cd1 >= 0 ? qa[2] : 0.0, a2 = cd2 >= 0 ? qa[3] : 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        double a_ = 0, b_ = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) { a_ = fma(Jr[k][r], qr[r], a_); b_ = fma(Jr[k][r], ar[r], b_); }
        a_ = fma(Jh[k][0], q1, a_); a_ = fma(Jh[k][1], q2, a_);
        b_ = fma(Jh[k][0], a1, b_); b_ = fma(Jh[k][1], a2, b_);
        jv[k] = a_; jp[k] = b_;
      }
    }
    const double margin = ctab[0];
    cmu = ctab[1];
    {
      const int root = m.ball_dof + 2;                    // Minitaur.SetFootFriction: every contact of a lower-leg link
      cmu = (ctab[5] > 0 && ls > root && parls != root) ? ctab[5] : cmu;
    }
    const double kk = ctab[2], bb = ctab[3];
    const double dd = imp_p2(simp, rec[0] - margin);
    const double R0 = fmax((1 - dd) * ctab[4] * rcp_nr(dd), 1e-15);
    cD = cv ? rcp_nr(2 * cmu * cmu * R0) : 0.0;
    const double basea = -kk * dd * (rec[0] - margin);
    // (the edges' combinations as EXPLICIT fused multiply-adds, here, in the warm-start test below and in the passes' row test: left to fp contract(fast) the one-wave and
    // the two-wave instantiations fused them differently, the warm-started edge set of a row within rounding of zero differed, and with it -- about once in 10^4 env steps --
    // the pass sequence and the last bits of the solution)
    car[0] = fma(-bb, fma(cmu, jv[1], jv[0]), basea); car[1] = fma(-bb, fma(-cmu, jv[1], jv[0]), basea);
    car[2] = fma(-bb, fma(cmu, jv[2], jv[0]), basea); car[3] = fma(-bb, fma(-cmu, jv[2], jv[0]), basea);
    unsigned int wbits = 0;
    wbits |= (fma(cmu, jp[1], jp[0]) - car[0] < 0) ? 1u : 0u;
    wbits |= (fma(-cmu, jp[1], jp[0]) - car[1] < 0) ? 2u : 0u;
    wbits |= (fma(cmu, jp[2], jp[0]) - car[2] < 0) ? 4u : 0u;
    wbits |= (fma(-cmu, jp[2], jp[0]) - car[3] < 0) ? 8u : 0u;
    // Start of the active-set passes: the first timestep of an env step from "every edge active"; later ones from the set this slot's passes ended with at the timestep before if
    // the slot holds the same collision pair again, else from the set the previous solution predicts for the new rows.  (The fixed point does not depend on the start; the
    // number of passes does: 2.05 -> 1.66 per timestep on random actions with the carried sets -- oracle/physics_oracle.c StepOut.pact.)
#ifndef EARL_MT_NO_CARRY                                  // (measurement switch: the start rule of rounds 2 - 5, for same-build comparisons -- tools/bench_mt_variant.py)
    cact = cv ? (warm ? ((ppid == cpid) ? pact : wbits) : 0xFu) : 0u;
#else
    cact = cv ? (warm ? wbits : 0xFu) : 0u;
#endif
  }
#ifdef EARL_MT_DEBUG
  if constexpr (ROLE != 2) {
    int* row = g_mt_dbg + ((size_t)s.dbg_env * 8 + (s.dbg_ts & 7)) * 32;
    if (sub < MC) row[4 + sub] = (int)cact;
    if (sub == 0) { row[0] = nct; row[1] = warm ? 1 : 0; }
  }
#endif
  if constexpr (ROLE == 0) {                             // (as above: the contact rows' results as the two-wave form hands them over)
    cD = pinned(cD); cmu = pinned(cmu); pin_batch(car);
    asm volatile("" : "+v"(cact), "+v"(cd1), "+v"(cd2));
  }
  if constexpr (ROLE == 1) {
    if (sub < MC) {
      s.c3[sub].cD = cD; s.c3[sub].cmu = cmu;
#pragma unroll
      for (int k = 0; k < 4; ++k) s.c3[sub].car[k] = car[k];
      s.c3[sub].cact = cact; s.c3[sub].cd1 = cd1; s.c3[sub].cd2 = cd2; s.c3[sub].pid = cpid;
    }
  }
  // which contact slots this lane's dof takes part in (bit c: it is the root's, or one of the two hinges of contact c's chain), which of the chain's two entries is
  // its own, whether it is the upper hinge: from the slots' chain records, ONCE per timestep (every pass read them again, a round trip before each contact's rows)
  unsigned int tbits = 0, sbits = 0, cbits = 0;
  if constexpr (ROLE == 2) {
    tbits = s.tb[l][0]; sbits = s.tb[l][1]; cbits = s.tb[l][2];
  }
  if (ROLE != 2 && ncmax > 0) {
    fence();
#pragma unroll
    for (int c = 0; c < MC; ++c) {
      const int d1 = s.crow[c][0], d2 = s.crow[c][1];
      tbits |= (isroot || l == d1 || l == d2) ? (1u << c) : 0u;
      sbits |= l == d2 ? (1u << c) : 0u;
      cbits |= l == d1 ? (1u << c) : 0u;
    }
  }
  if constexpr (ROLE == 1) {
    if (isl) { s.tb[l][0] = tbits; s.tb[l][1] = sbits; s.tb[l][2] = cbits; }
    return;
  }
  PSTAMP(6);
  PSTAMP(9);
  if constexpr (ROLE == 2) {
    // (after the contact rows, which need none of it; before the first pass writes s.pas, which lies over it)
    const int hl = ishinge ? l - 6 : 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) Bw[i] = s.hand.Bw[l][i];
#pragma unroll
    for (int p = 0; p < 4; ++p) Aw[p] = s.hand.Aw[hl][p];
    rw = s.hand.rw[l];
    pin_batch(Bw); pin_batch(Aw); rw = pinned(rw);
    fence();
  }
#ifdef EARL_MT_DEBUG
  if (isl) {
    const size_t o_ = ((size_t)s.dbg_env * 8 + (s.dbg_ts & 7)) * 32 + l;
    g_mt_dbg_x[0][o_] = rw;
    g_mt_dbg_x[1][o_] = Bw[0] + 2 * Bw[1] + 3 * Bw[2] + 5 * Bw[3] + 7 * Bw[4] + 11 * Bw[5];
    g_mt_dbg_x[2][o_] = Aw[0] + 2 * Aw[1] + 3 * Aw[2] + 5 * Aw[3];
    g_mt_dbg_x[4][o_] = s.xt.ext[l];
  }
#endif
  // ------------------------------------------------------------------ K9: active-set Newton on the arrow-shaped Hessian
  double al = 0.0;                                       // this lane's entry of the solution
  PCOUNT(23, ncmax > 0 ? 1 : 0); PCOUNT(24, ncmax);
  for (int it = 0; it < 8; ++it) {
    PCOUNT(25, 1);
#ifdef EARL_MT_DEBUG
    if (sub == 0) g_mt_dbg[((size_t)s.dbg_env * 8 + (s.dbg_ts & 7)) * 32 + 2] = it + 1;
#endif
    KSTART();
    if (ncmax > 0) {
      if (sub < MC) {
        const double a1 = (cact & 1u) ? cD : 0.0, a2 = (cact & 2u) ? cD : 0.0, a3 = (cact & 4u) ? cD : 0.0, a4 = (cact & 8u) ? cD : 0.0;
        double* w = s.cw[sub];
        w[0] = a1 + a2 + a3 + a4; w[1] = cmu * (a1 - a2); w[2] = cmu * (a3 - a4); w[3] = cmu * cmu * (a1 + a2); w[4] = cmu * cmu * (a3 + a4);
        w[5] = a1 * car[0] + a2 * car[1] + a3 * car[2] + a4 * car[3];
        w[6] = cmu * (a1 * car[0] - a2 * car[1]);
        w[7] = cmu * (a3 * car[2] - a4 * car[3]);
      }
      fence();
    }
    KSTAMP(16);
    {
      // this lane's entries of the pass's Hessian: equality part + the active contact edges
      double acc6[6] = {0, 0, 0, 0, 0, 0}, accd = 0, accc = 0, rr = rw;
      // (four contacts per iteration, their loads side by side: slots beyond the wave's count hold zero rows and zero weights -- C3 and the weights
      // above write all MC slots -- so they are simply summed; one contact per iteration paid an LDS round trip each, 1.3 k cycles per contact)
      static_assert(MC % 4 == 0, "contact slots in groups of four");
      for (int c4 = 0; c4 < ncmax; c4 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int c = c4 + u;
          const double* w = s.cw[c];
          const int slot = (sbits >> c) & 1u;
          const double tmask = ((tbits >> c) & 1u) ? 1.0 : 0.0, cmask = ((cbits >> c) & 1u) ? 1.0 : 0.0;
          // The contact's rows and weights: ALL loads first (pin6 / pin_batch: one s_waitcnt), then the arithmetic.  Left to the scheduler they came two at a time, each
          // pair with its own wait; and this lane's own entries -- ONE load each, used unconditionally (times 1 or 0, exact: slots beyond the count hold zero rows) --
          // selected by `touch` ended up under a branch per load: twenty LDS round trips one after the other per group of four contacts
          double j[3], jc[3], wv[8], jr[18];
  #pragma unroll
          for (int k = 0; k < 3; ++k) {
            j[k] = *(isroot ? &s.CJr[c][k][l] : &s.CJh[c][k][slot]);
            jc[k] = s.CJh[c][k][1];
          }
  #pragma unroll
          for (int k = 0; k < 8; ++k) wv[k] = w[k];
  #pragma unroll
          for (int k = 0; k < 3; ++k)
  #pragma unroll
            for (int i = 0; i < 6; ++i) jr[6 * k + i] = s.CJr[c][k][i];
          pin6(j[0], j[1], j[2], jc[0], jc[1], jc[2]); pin_batch(wv); pin_batch(jr);
  #pragma unroll
          for (int k = 0; k < 3; ++k) j[k] *= tmask;
          const double v0 = wv[0] * j[0] + wv[1] * j[1] + wv[2] * j[2], v1 = wv[1] * j[0] + wv[3] * j[1], v2 = wv[2] * j[0] + wv[4] * j[2];
          rr += wv[5] * j[0] + wv[6] * j[1] + wv[7] * j[2];
  #pragma unroll
          for (int i = 0; i < 6; ++i) acc6[i] += jr[i] * v0 + jr[6 + i] * v1 + jr[12 + i] * v2;
          accd += j[0] * v0 + j[1] * v1 + j[2] * v2;
          accc += (jc[0] * v0 + jc[1] * v1 + jc[2] * v2) * cmask;
        }
      }
      if (isroot) {
#pragma unroll
        for (int i = 0; i < 6; ++i) s.pas.HR[l][i] = Bw[i] + acc6[i];
      }
      if (ishinge) {
#pragma unroll
        for (int i = 0; i < 6; ++i) s.pas.HB[l - 6][i] = Bw[i] + acc6[i];
#pragma unroll
        for (int p = 0; p < 4; ++p)
          if (p >= hq) s.pas.HA[leg][p * (p + 1) / 2 + hq] = Aw[p] + (p == hq ? accd : ((upper && p == hq + 1) ? accc : 0.0));
      }
      if (isl) s.pas.rc[l] = rr;
    }
    fence();
    KSTAMP(17);
    // ---- solve H x = rc.  Lanes 0-23 = (root dof i, leg k): factor the leg block (redundantly, six lanes per leg), row i of W_k = B_k' L_k^-T, y_k, W_k y_k
    double xr[6];
    {
      const int t = sub < 24 ? sub : 23;
      const int i = t % 6, k = t / 6;
      double L[10], Bc[4], bk[4];
#pragma unroll
      for (int e = 0; e < 10; ++e) L[e] = s.pas.HA[k][e];
#pragma unroll
      for (int j = 0; j < 4; ++j) { Bc[j] = s.pas.HB[4 * k + j][i]; bk[j] = s.pas.rc[6 + 4 * k + j]; }
      pin_batch(L); pin_batch(Bc); pin_batch(bk);         // (all of the stage's loads before its first use: physics_math.h pin_batch)
      chol_small<4>(L);
      double W[4], y[4], pt = 0.0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        double sw = Bc[j], sy = bk[j];
#pragma unroll
        for (int p = 0; p < j; ++p) { sw = fma(-W[p], L[j * (j + 1) / 2 + p], sw); sy = fma(-y[p], L[j * (j + 1) / 2 + p], sy); }
        W[j] = sw * L[j * (j + 1) / 2 + j];
        y[j] = sy * L[j * (j + 1) / 2 + j];
        pt = fma(W[j], y[j], pt);
      }
      if (sub < 24) {
#pragma unroll
        for (int j = 0; j < 4; ++j) s.pas.Wl[i][4 * k + j] = W[j];
        s.pas.part[i][k] = pt;
        if (i == 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) s.pas.Y[4 * k + j] = y[j];
#pragma unroll
          for (int e = 0; e < 10; ++e) s.pas.LL[k][e] = L[e];
        }
      }
    }
    fence();
    {
      // Schur complement of the root block, lane = entry (i, c), c <= i
      const int e = sub < 21 ? sub : 20;
      const int i = e >= 15 ? 5 : (e >= 10 ? 4 : (e >= 6 ? 3 : (e >= 3 ? 2 : (e >= 1 ? 1 : 0))));
      const int c = e - i * (i + 1) / 2;
      double s0 = s.pas.HR[i][c], s1 = 0.0, s2 = 0.0, s3 = 0.0;
      double wi[16], wc[16];
#pragma unroll
      for (int p = 0; p < 16; ++p) { wi[p] = s.pas.Wl[i][p]; wc[p] = s.pas.Wl[c][p]; }
      s0 = pinned(s0); pin_batch(wi); pin_batch(wc);
#pragma unroll
      for (int p = 0; p < 16; p += 4) {
        s0 = fma(-wi[p], wc[p], s0); s1 = fma(-wi[p + 1], wc[p + 1], s1);
        s2 = fma(-wi[p + 2], wc[p + 2], s2); s3 = fma(-wi[p + 3], wc[p + 3], s3);
      }
      if (sub < 21) s.pas.SS[e] = (s0 + s1) + (s2 + s3);
      // ... and the root block's right-hand side, lane 21 + i = root dof i (until round 5 every lane formed all six in the next stage: 30 loads into the same
      // registers, twelve LDS round trips one after the other)
      const int ri = sub >= 21 && sub < 27 ? sub - 21 : 0;
      const double x0 = s.pas.rc[ri] - ((s.pas.part[ri][0] + s.pas.part[ri][1]) + (s.pas.part[ri][2] + s.pas.part[ri][3]));
      if (sub >= 21 && sub < 27) s.pas.xr0[ri] = x0;
    }
    fence();
    {
      // every lane: the root block's 6 x 6 system in registers
      double Lr[21];
#pragma unroll
      for (int e = 0; e < 21; ++e) Lr[e] = s.pas.SS[e];
#pragma unroll
      for (int i = 0; i < 6; ++i) xr[i] = s.pas.xr0[i];
      // (the back-substitution's operands too: everything this stage reads is on its way before the factorisation starts -- physics_math.h pin_batch)
      double Lk[10], z[4], wl[24];
#pragma unroll
      for (int e = 0; e < 10; ++e) Lk[e] = s.pas.LL[leg][e];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        z[j] = s.pas.Y[4 * leg + j];
#pragma unroll
        for (int i = 0; i < 6; ++i) wl[6 * j + i] = s.pas.Wl[i][4 * leg + j];
      }
      pin_batch(Lr); pin_batch(xr); pin_batch(Lk); pin_batch(z);
      chol_small<6>(Lr);
      solve_regs<6, 6>(Lr, xr);
      // hinge lanes: back-substitution of their leg, x_k = L_k^-T (y_k - W_k' x_root)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        double zz = z[j];
#pragma unroll
        for (int i = 0; i < 6; ++i) zz = fma(-wl[6 * j + i], xr[i], zz);
        z[j] = zz;
      }
#pragma unroll
      for (int j = 3; j >= 0; --j) {
        double zz = z[j];
#pragma unroll
        for (int p = j + 1; p < 4; ++p) zz = fma(-Lk[p * (p + 1) / 2 + j], z[p], zz);
        z[j] = zz * Lk[j * (j + 1) / 2 + j];
      }
      const double xh = hq == 0 ? z[0] : (hq == 1 ? z[1] : (hq == 2 ? z[2] : z[3]));
      double xo = 0.0;
#pragma unroll
      for (int i = 0; i < 6; ++i) xo = sub == i ? xr[i] : xo;
      al = isroot ? xo : xh;
      if (isl) s.aprev[l] = al;
    }
    KSTAMP(18);
    bool changed = false;
    if (ncmax > 0) {
      fence();                                            // the hinges' entries of the solution
      const int c = sub < MC ? sub : MC - 1;
      double cj[18], ch[6], ah[2] = {s.aprev[cd1 >= 0 ? cd1 : 0], s.aprev[cd2 >= 0 ? cd2 : 0]};      // (one batch of loads: physics_math.h pin_batch)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int r = 0; r < 6; ++r) cj[6 * k + r] = s.CJr[c][k][r];
        ch[2 * k] = s.CJh[c][k][0]; ch[2 * k + 1] = s.CJh[c][k][1];
      }
      pin_batch(ah); pin_batch(ch); pin_batch(cj);
      const double a1 = cd1 >= 0 ? ah[0] : 0.0, a2 = cd2 >= 0 ? ah[1] : 0.0;
      double an[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        double a_ = 0;
#pragma unroll
        for (int r = 0; r < 6; ++r) a_ = fma(cj[6 * k + r], xr[r], a_);
        a_ = fma(ch[2 * k], a1, a_); a_ = fma(ch[2 * k + 1], a2, a_);
        an[k] = a_;
      }
      unsigned int nb = 0;
      nb |= (fma(cmu, an[1], an[0]) - car[0] < 0) ? 1u : 0u;
      nb |= (fma(-cmu, an[1], an[0]) - car[1] < 0) ? 2u : 0u;
      nb |= (fma(cmu, an[2], an[0]) - car[2] < 0) ? 4u : 0u;
      nb |= (fma(-cmu, an[2], an[0]) - car[3] < 0) ? 8u : 0u;
      nb = sub < nct ? nb : 0u;
      changed = nb != cact;
      cact = nb;
    }
    KSTAMP(19);
    if (!__any(changed)) break;
    PCOUNT(10, it == 2 ? 1 : 0);                          // (profiling build: timesteps that go beyond three passes ...
    PCOUNT(12, it == 6 ? 1 : 0);                          //  ... and those that use all eight)
  }
#ifdef EARL_MT_DEBUG
  {
    int* row = g_mt_dbg + ((size_t)s.dbg_env * 8 + (s.dbg_ts & 7)) * 32;
    if (sub < MC) row[16 + sub] = (int)cact;
    if (isl) reinterpret_cast<double*>(g_mt_dbg_al)[((size_t)s.dbg_env * 8 + (s.dbg_ts & 7)) * 32 + l] = al;
    fence();
    if (sub == 0) s.dbg_ts = s.dbg_ts + 1;
    fence();
  }
#endif
  // what the next timestep's C3 starts from (above): the slot's final edge set, and -- in the one-wave form, which hands nothing over through s.c3 -- its pair
  if constexpr (ROLE == 2) {
    if (ncmax > 0 && sub < MC) s.c3[sub].cact = cact;
  } else {
    if (sub < MC) { s.c3[sub].cact = cact; s.c3[sub].pid = cpid; }
  }
  PSTAMP(8);
  if constexpr (ROLE == 2) return;                       // (the two-wave form integrates at the head of the first-half wave's next visit: integrate_mt from s.aprev)
  if constexpr (!INTEGRATE) {
    if (qacc_out && isl) qacc_out[l] = al;
  } else {
    integrate_mt(s, m, sub, isl, l, dt, al, qd, ql_, Qb);
#ifdef EARL_MT_DEBUG
    if (isl) g_mt_dbg_x[3][((size_t)s.dbg_env * 8 + ((s.dbg_ts - 1) & 7)) * 32 + l] = s.qv[l];
#endif
    PSTAMP(11);
  }
}
