// physics_solve.h -- Cholesky factorisations and solves: register-resident (per lane, redundant), lane-cooperative in LDS, row-distributed, leading-block and Schur-complement forms
// A section of csrc/physics.hip (included there, inside its anonymous namespace): split out in round 5.

// Cholesky of an SPD matrix held in registers (lower triangle, row-major packed); the diagonal is left INVERTED.
// NA < NV: the matrix is block diagonal, rows / columns [0, NA) and [NA, NV) -- the arm and the free object are separate
// trees, so the mass matrix always is, and the Hessian is unless a contact joins the two.  The entries of the off-diagonal block
// are then never read or written (their registers are dead on that path).
// (every multiply-subtract is an explicit fma in the same order as chol_coop / solve_lds below: the register-resident and the in-LDS
// factorisation then produce the same bits, which is what lets earl_sawyer_rollout switch between its two door builds by batch size)
template <int NV, int NA, bool FAST = false>
__device__ __forceinline__ void chol_regs(double (&L)[NV * (NV + 1) / 2]) {
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int p0 = j >= NA ? NA : 0;                  // first column of row j's block
    double d = L[j * (j + 1) / 2 + j];
#pragma unroll
    for (int p = p0; p < j; ++p) d = fma(-L[j * (j + 1) / 2 + p], L[j * (j + 1) / 2 + p], d);
    const double inv = FAST ? rsq2(d) : rsq_nr(d);       // (FAST: the peg and kitchen models; the door's two builds stay pinned bit for bit)
    L[j * (j + 1) / 2 + j] = inv;
#pragma unroll
    for (int i = j + 1; i < (j < NA ? NA : NV); ++i) {
      double s = L[i * (i + 1) / 2 + j];
#pragma unroll
      for (int p = p0; p < j; ++p) s = fma(-L[i * (i + 1) / 2 + p], L[j * (j + 1) / 2 + p], s);
      L[i * (i + 1) / 2 + j] = s * inv;
    }
  }
}
template <int NV, int NA>
__device__ __forceinline__ void solve_regs(const double (&L)[NV * (NV + 1) / 2], double (&x)[NV]) {   // (L L') x' = x
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double s = x[i];
#pragma unroll
    for (int p = (i >= NA ? NA : 0); p < i; ++p) s = fma(-L[i * (i + 1) / 2 + p], x[p], s);
    x[i] = s * L[i * (i + 1) / 2 + i];
  }
#pragma unroll
  for (int i = NV - 1; i >= 0; --i) {
    double s = x[i];
#pragma unroll
    for (int p = i + 1; p < (i < NA ? NA : NV); ++p) s = fma(-L[p * (p + 1) / 2 + i], x[p], s);
    x[i] = s * L[i * (i + 1) / 2 + i];
  }
}
// the lower triangle of an LDS matrix (+ a diagonal term) into the packed register form, skipping the off-diagonal block when NA < NV
template <int NV, int NA, typename D>
__device__ __forceinline__ void load_tri(double (&L)[NV * (NV + 1) / 2], const SymLds<NV>& H, D diag) {
#pragma unroll
  for (int i = 0; i < NV; ++i) {
#pragma unroll
    for (int j = (i >= NA ? NA : 0); j < i; ++j) L[i * (i + 1) / 2 + j] = H.lo(i, j);
    L[i * (i + 1) / 2 + i] = H.lo(i, i) + diag(i);
  }
}

// Dense factorisation for the rare timesteps in which a contact joins the two trees of a big model (gripper plates on the
// peg): a register-resident 15 x 15 factor would need 240 VGPRs and spills the whole kernel into scratch memory.  Instead the
// lanes share the work, lane = row, the factor overwrites the lower triangle of H in LDS (diagonal INVERTED): left-looking
// by columns, every lane recomputes the pivot redundantly from the pivot row it has just read, so a column costs one LDS
// round trip.  The two triangular solves then read L back from LDS, redundantly per lane (no exchange).
template <int NV>
__device__ __forceinline__ void chol_coop(SymLds<NV>& H, const double (&dl)[NV], const int l, const bool isl) {
  const int ltri = l * (l + 1) / 2;
  double r[NV];                                        // row l of H, then of L (entries j <= l; the others are never used)
#pragma unroll
  for (int j = 0; j < NV; ++j) r[j] = H.sym(j, l, ltri);
  if (isl) H.rowl(l, ltri, l) = r[l] + dl[l];
  fence();
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    double sj = H.lo(j, j), si = r[j] + (l == j ? dl[l] : 0.0);
#pragma unroll
    for (int p = 0; p < j; ++p) {
      const double pj = H.lo(j, p);
      sj = fma(-pj, pj, sj);
      si = fma(-r[p], pj, si);
    }
    const double inv = rsq_nr(sj);
    r[j] = si * inv;
    if (isl && l >= j) H.rowl(l, ltri, j) = l == j ? inv : r[j];
    fence();
  }
}
#ifndef EARL_KITCHEN_DENSE
#define EARL_KITCHEN_DENSE 0
#endif
#if EARL_KITCHEN_DENSE
#error "EARL_KITCHEN_DENSE: since round 4 the nv = 23 Hessian is stored by its structure only (arm block, fixture rows against the arm, fixture diagonal / pairs); the dense path would read entries nobody writes"
#endif
#ifndef EARL_MT_LOOP_SOLVER
#define EARL_MT_LOOP_SOLVER 0       // nv = 22: the looping in-LDS factorisation / substitution instead of the unrolled ones (measurement switch: 70 k against ~10 k cycles per solve)
#endif
// Looping form of the two for the big model (nv = 23): fully unrolled, chol_coop + solve_lds keep two 23-entry vectors in registers and made the
// kernel spill 1.4 KB per lane into scratch.  Here the lane's row of L stays where it is (in H), the right-hand side / solution stays in LDS
// (every lane of the env runs the same substitution on the same numbers, so the redundant stores agree), and nothing is indexed dynamically in
// registers.  Only taken when a contact joins the arm to a fixture.
template <int NV>
__device__ __forceinline__ void chol_coop_loop(SymLds<NV>& H, const double (&dl)[NV], const int l, const bool isl) {
  static_assert(SymLds<NV>::PACKED, "packed storage");
  const int ltri = l * (l + 1) / 2;
  if (isl) H.v[ltri + l] += dl[l];
  fence();
  for (int j = 0; j < NV; ++j) {
    const int jtri = j * (j + 1) / 2;
    double sj = H.v[jtri + j], si = H.v[(l >= j ? ltri : jtri) + j];
#pragma unroll 4
    for (int p = 0; p < j; ++p) {
      const double pj = H.v[jtri + p], rp = H.v[(l >= j ? ltri : jtri) + p];
      sj = fma(-pj, pj, sj);
      si = fma(-rp, pj, si);
    }
    const double inv = rsq_nr(sj);
    fence();                                            // every lane has read column j's inputs before the pivot row is overwritten
    if (isl && l >= j) H.v[ltri + j] = l == j ? inv : si * inv;
    fence();
  }
}
template <int NV>
__device__ __forceinline__ void solve_lds_loop(const SymLds<NV>& H, double (&x)[NV]) {   // x in LDS, in place; L as chol_coop_loop leaves it
  for (int i = 0; i < NV; ++i) {
    const int itri = i * (i + 1) / 2;
    double s = x[i];
#pragma unroll 4
    for (int p = 0; p < i; ++p) s = fma(-H.v[itri + p], x[p], s);
    x[i] = s * H.v[itri + i];
  }
  for (int i = NV - 1; i >= 0; --i) {
    double s = x[i];
#pragma unroll 4
    for (int p = i + 1; p < NV; ++p) s = fma(-H.v[p * (p + 1) / 2 + i], x[p], s);
    x[i] = s * H.v[i * (i + 1) / 2 + i];
  }
}

// Dense SPD solve for the one-tree model (nv = 22, 32 lanes per env, two envs per wave): H (packed lower triangle in LDS, column l written by lane l)
// plus dl on the diagonal, right-hand side b_l in this lane -> this lane's entry of the solution.  Lane = row.  The lane keeps ITS ROW of L in
// registers (r[]); a column of the left-looking factorisation costs one broadcast read of the pivot row from LDS and one cross-lane broadcast of the
// pivot's inverse root (v_readlane with a constant lane per env group, no LDS round trip); the two substitutions exchange one solution entry per step the
// same way and read nothing but the lane's own row (forward) / own column (backward, fetched from LDS in one batch).  The unrolled chol_coop +
// solve_lds pair reads the whole factor back per lane (462 loads the compiler hoists: 1.3 KB of scratch per lane here); their looping forms walk through
// LDS one dependent round trip at a time (70 k cycles per solve, measured: 57 % of the minitaur's timestep).
template <int J>
__device__ __forceinline__ double group_bcast(const double v, const int grp) {      // v of lane J of this lane's 32-lane group
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const int lo0 = __builtin_amdgcn_readlane(lo, J), hi0 = __builtin_amdgcn_readlane(hi, J);
  const int lo1 = __builtin_amdgcn_readlane(lo, 32 + J), hi1 = __builtin_amdgcn_readlane(hi, 32 + J);
  return __hiloint2double(grp ? hi1 : hi0, grp ? lo1 : lo0);
}
template <int NV, int J>
struct CholRows {
  // column J of the factorisation, then the later ones
  static __device__ __forceinline__ void factor(SymLds<NV>& H, double (&r)[NV], double& inv_l, const int l, const int ltri, const bool isl, const int grp) {
    // s_l = H[l][J] - sum_{p < J} L[l][p] L[J][p]  (meaningful for l >= J); two partial sums halve the dependent chain
    double s0 = r[J], s1 = 0.0;
#pragma unroll
    for (int p = 0; p < J; ++p) {
      const double pj = H.v[J * (J + 1) / 2 + p];              // pivot row: the same address in every lane of the env (LDS broadcast)
      if (p & 1) s1 = fma(-r[p], pj, s1); else s0 = fma(-r[p], pj, s0);
    }
    const double sj = s0 + s1;
    const double inv = group_bcast<J>(rsq_nr(sj), grp);        // 1 / L[J][J], from the pivot's own lane
    r[J] = sj * inv;                                           // L[l][J] (lane J: L[J][J] itself)
    inv_l = l == J ? inv : inv_l;
    if (isl && l > J) H.v[ltri + J] = r[J];                    // my row's entry: lane l's row is the pivot row of column l
    fence();
    if constexpr (J + 1 < NV) CholRows<NV, J + 1>::factor(H, r, inv_l, l, ltri, isl, grp);
  }
  // forward substitution L y = b: step J hands y_J to the rows below
  static __device__ __forceinline__ void forward(const double (&r)[NV], const double inv_l, double& t, const int l, const int grp) {
    const double yj = group_bcast<J>(t * inv_l, grp);
    t = l > J ? fma(-r[J], yj, t) : (l == J ? yj : t);
    if constexpr (J + 1 < NV) CholRows<NV, J + 1>::forward(r, inv_l, t, l, grp);
  }
  // backward substitution L' x = y: step J (from the last row up) hands x_J to the rows above; c[] = this lane's COLUMN of L
  static __device__ __forceinline__ void backward(const double (&c)[NV], const double inv_l, double& t, const int l, const int grp) {
    const double xj = group_bcast<J>(t * inv_l, grp);
    t = l < J ? fma(-c[J], xj, t) : (l == J ? xj : t);
    if constexpr (J > 0) CholRows<NV, J - 1>::backward(c, inv_l, t, l, grp);
  }
};
template <int NV>
__device__ __forceinline__ double chol_solve_rows(SymLds<NV>& H, const double (&dl)[NV], const double b_l, const int l, const bool isl, const int grp) {
  static_assert(SymLds<NV>::PACKED, "packed storage");
  const int ltri = l * (l + 1) / 2;
  double r[NV], inv_l = 1.0;
#pragma unroll
  for (int j = 0; j < NV; ++j) r[j] = H.v[(j <= l ? ltri + j : j * (j + 1) / 2 + l)] + (j == l ? dl[l] : 0.0);     // row l of H (symmetric: entry (l, j))
  fence();
  CholRows<NV, 0>::factor(H, r, inv_l, l, ltri, isl, grp);
  double t = b_l;
  CholRows<NV, 0>::forward(r, inv_l, t, l, grp);
#pragma unroll
  for (int k = 0; k < NV; ++k) r[k] = H.v[k * (k + 1) / 2 + (k > l ? l : 0)];     // column l of L: entries (k, l), k > l (the others are not used)
  CholRows<NV, NV - 1>::backward(r, inv_l, t, l, grp);
  return t;
}

// the same on the leading N x N block only (a model whose first N dofs are one tree and whose other dofs are decoupled from it: the kitchen's arm)
template <int NV, int N>
__device__ __forceinline__ void chol_coop_lead(SymLds<NV>& H, const double (&dl)[NV], const int l, const bool isl) {
  const int ltri = l * (l + 1) / 2;
  const bool mine = isl && l < N;
  double r[N];
#pragma unroll
  for (int j = 0; j < N; ++j) r[j] = H.sym(j, l < N ? l : 0, l < N ? ltri : 0);
  if (mine) H.rowl(l, ltri, l) = r[l] + dl[l];
  fence();
#pragma unroll
  for (int j = 0; j < N; ++j) {
    double sj = H.lo(j, j), si = r[j] + (l == j ? dl[l] : 0.0);
#pragma unroll
    for (int p = 0; p < j; ++p) {
      const double pj = H.lo(j, p);
      sj = fma(-pj, pj, sj);
      si = fma(-r[p], pj, si);
    }
    const double inv = rsq_nr(sj);
    r[j] = si * inv;
    if (mine && l >= j) H.rowl(l, ltri, j) = l == j ? inv : r[j];
    fence();
  }
}
template <int NV, int N>
__device__ __forceinline__ void solve_lds_lead(const SymLds<NV>& H, double (&x)[NV]) {   // leading block of (L L') x' = x
#pragma unroll
  for (int i = 0; i < N; ++i) {
    double s = x[i];
#pragma unroll
    for (int p = 0; p < i; ++p) s = fma(-H.lo(i, p), x[p], s);
    x[i] = s * H.lo(i, i);
  }
#pragma unroll
  for (int i = N - 1; i >= 0; --i) {
    double s = x[i];
#pragma unroll
    for (int p = i + 1; p < N; ++p) s = fma(-H.lo(p, i), x[p], s);
    x[i] = s * H.lo(i, i);
  }
}
// the leading N x N block factorised and solved in REGISTERS, redundantly per lane (chol_regs / solve_regs on a copy: same operations in the same order
// as chol_coop_lead / solve_lds_lead, which cost nine plus eighteen LDS round trips in a row)
template <int NV, int N, typename D>
__device__ __forceinline__ void solve_lead_regs(const SymLds<NV>& H, D diag, double (&x)[NV]) {
  double L[N * (N + 1) / 2], y[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
#pragma unroll
    for (int j = 0; j < i; ++j) L[i * (i + 1) / 2 + j] = H.lo(i, j);
    L[i * (i + 1) / 2 + i] = H.lo(i, i) + diag(i);
    y[i] = x[i];
  }
  pin_batch(L);                                          // (the block as one batch of loads, ahead of the factorisation's first use)
  chol_regs<N, N, true>(L);
  solve_regs<N, N>(L, y);
#pragma unroll
  for (int i = 0; i < N; ++i) x[i] = y[i];
}
template <int NV>
__device__ __forceinline__ void solve_lds(const SymLds<NV>& H, double (&x)[NV]) {   // (L L') x' = x, L in LDS as chol_coop leaves it
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double s = x[i];
#pragma unroll
    for (int p = 0; p < i; ++p) s = fma(-H.lo(i, p), x[p], s);
    x[i] = s * H.lo(i, i);
  }
#pragma unroll
  for (int i = NV - 1; i >= 0; --i) {
    double s = x[i];
#pragma unroll
    for (int p = i + 1; p < NV; ++p) s = fma(-H.lo(p, i), x[p], s);
    x[i] = s * H.lo(i, i);
  }
}

// Two-tree model (arm [0, NA) + free object [NA, NV)) in the timesteps in which a contact JOINS the trees: H = [A B'; B P] is dense.  Instead of the shared
// in-LDS factorisation of all NV columns (chol_coop + solve_lds: one LDS round trip per column and per substitution step -- 52 k cycles per timestep in the
// waves whose gripper holds the peg, the waves the launch waits for), eliminate the object's block first, everything in registers and redundantly per lane
// like the contact-free path: P = Lp Lp', W = B' Lp^-T, S = A - W W' = La La', x_A = S^-1 (b_A - W Lp^-1 b_P), x_P = Lp^-T (Lp^-1 b_P - W' x_A).
// No exchange between lanes at all: every lane reads the same Hessian from LDS and ends up with the whole solution.
template <int NV, int NA, typename D>
__device__ __forceinline__ void solve_schur_regs(const SymLds<NV>& H, D diag, double (&x)[NV]) {
  constexpr int NP = NV - NA;
  double Lp[NP * (NP + 1) / 2], yp[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) {
#pragma unroll
    for (int j = 0; j < i; ++j) Lp[i * (i + 1) / 2 + j] = H.lo(NA + i, NA + j);
    Lp[i * (i + 1) / 2 + i] = H.lo(NA + i, NA + i) + diag(NA + i);
    yp[i] = x[NA + i];
  }
  chol_regs<NP, NP, true>(Lp);
#pragma unroll
  for (int i = 0; i < NP; ++i) {                        // yp = Lp^-1 b_P
    double t = yp[i];
#pragma unroll
    for (int p = 0; p < i; ++p) t = fma(-Lp[i * (i + 1) / 2 + p], yp[p], t);
    yp[i] = t * Lp[i * (i + 1) / 2 + i];
  }
  double W[NA][NP];                                     // W[i][j] = (B[j][i] - sum_{p < j} W[i][p] Lp[j][p]) / Lp[j][j]
#pragma unroll
  for (int i = 0; i < NA; ++i) {
#pragma unroll
    for (int j = 0; j < NP; ++j) {
      double t = H.lo(NA + j, i);
#pragma unroll
      for (int p = 0; p < j; ++p) t = fma(-W[i][p], Lp[j * (j + 1) / 2 + p], t);
      W[i][j] = t * Lp[j * (j + 1) / 2 + j];
    }
  }
  double La[NA * (NA + 1) / 2], xa[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
#pragma unroll
    for (int c = 0; c <= i; ++c) {
      double t = H.lo(i, c) + (c == i ? diag(i) : 0.0);
#pragma unroll
      for (int j = 0; j < NP; ++j) t = fma(-W[i][j], W[c][j], t);
      La[i * (i + 1) / 2 + c] = t;
    }
    double t = x[i];
#pragma unroll
    for (int j = 0; j < NP; ++j) t = fma(-W[i][j], yp[j], t);
    xa[i] = t;
  }
  chol_regs<NA, NA, true>(La);
  solve_regs<NA, NA>(La, xa);
#pragma unroll
  for (int j = 0; j < NP; ++j) {                        // z = yp - W' x_A
    double t = yp[j];
#pragma unroll
    for (int i = 0; i < NA; ++i) t = fma(-W[i][j], xa[i], t);
    yp[j] = t;
  }
#pragma unroll
  for (int j = NP - 1; j >= 0; --j) {                   // x_P = Lp^-T z
    double t = yp[j];
#pragma unroll
    for (int p = j + 1; p < NP; ++p) t = fma(-Lp[p * (p + 1) / 2 + j], yp[p], t);
    yp[j] = t * Lp[j * (j + 1) / 2 + j];
  }
#pragma unroll
  for (int i = 0; i < NA; ++i) x[i] = xa[i];
#pragma unroll
  for (int j = 0; j < NP; ++j) x[NA + j] = yp[j];
}

