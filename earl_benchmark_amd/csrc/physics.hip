// physics.hip -- batched articulated-body stepper (include/earl_physics.h), one WAVEFRONT per env instance.
//
// Per-link state lives in LDS; lanes take the role the phase offers (link, (link, component), matrix entry, constraint
// row).  A single wave owns its LDS block, so phases are separated by wavefront-scope fences only (the LDS queue of
// a wave is in order; the fence pins the compiler) -- no s_barrier anywhere.  fp64 like MuJoCo.
//
// Pipeline per timestep (reference: oracle/physics_oracle.py LinkModel.forward / step, phase by phase):
//   kinematics (chain composition, lane = link) -> per-link spatial inertia in the compact additive form (m, m c, Io)
//   -> composite inertias (subtree sums) -> mass matrix (lane = (i, j) pair) + armature -> bias forces (RNE: velocity /
//   acceleration along the ancestor chain, subtree sums of forces) -> Cholesky of M in registers (NV is a compile-time
//   constant) -> constraint rows (6 weld rows to the mocap body, 2 limit rows per dof) with MuJoCo's solref / solimp
//   impedance -> Y = L^-1 J^T, A + R = Y^T Y + R -> exact active-set solve (compacted dense Cholesky in LDS) ->
//   qacc -> semi-implicit Euler with implicit joint damping.
// NO contacts yet.  Parity vs MuJoCo is unpinned; parity vs the reference above is tested to 1e-9.
#include <hip/hip_runtime.h>

#include <cstdio>

#include "../../include/earl_physics.h"
#include "philox.h"

namespace {

__device__ __forceinline__ void fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct Q4 { double w, x, y, z; };
struct V3 { double x, y, z; };
__device__ __forceinline__ Q4 qmul(const Q4& a, const Q4& b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x, a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w};
}
__device__ __forceinline__ void qmat(const Q4& q, double (&R)[3][3]) {
  R[0][0] = 1 - 2 * (q.y * q.y + q.z * q.z); R[0][1] = 2 * (q.x * q.y - q.w * q.z); R[0][2] = 2 * (q.x * q.z + q.w * q.y);
  R[1][0] = 2 * (q.x * q.y + q.w * q.z); R[1][1] = 1 - 2 * (q.x * q.x + q.z * q.z); R[1][2] = 2 * (q.y * q.z - q.w * q.x);
  R[2][0] = 2 * (q.x * q.z - q.w * q.y); R[2][1] = 2 * (q.y * q.z + q.w * q.x); R[2][2] = 1 - 2 * (q.x * q.x + q.y * q.y);
}
__device__ __forceinline__ V3 mulv(const double (&R)[3][3], const V3& v) {
  return {R[0][0] * v.x + R[0][1] * v.y + R[0][2] * v.z, R[1][0] * v.x + R[1][1] * v.y + R[1][2] * v.z,
          R[2][0] * v.x + R[2][1] * v.y + R[2][2] * v.z};
}
__device__ __forceinline__ V3 cross(const V3& a, const V3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3 add(const V3& a, const V3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 sub(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 scl(const V3& a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 ld3(const double* p) { return {p[0], p[1], p[2]}; }

// compact spatial inertia about the world origin: I10 = {m, h = m c (3), Io xx yy zz xy xz yz}; I [w; v] = [n; f]
__device__ __forceinline__ void iapply(const double* I, const V3& w, const V3& v, V3& n, V3& f) {
  const V3 h = ld3(I + 1);
  n = add(V3{I[4] * w.x + I[7] * w.y + I[8] * w.z, I[7] * w.x + I[5] * w.y + I[9] * w.z, I[8] * w.x + I[9] * w.y + I[6] * w.z}, cross(h, v));
  f = sub(scl(v, I[0]), cross(h, w));
}

// MuJoCo impedance d(r) and (k, b) of a soft constraint row (reference: physics_oracle.kbimp)
__device__ __forceinline__ void kbimp(const double* solref, const double* solimp, double r, double dt, double& k, double& b, double& d) {
  const double tc = fmax(solref[0], 2 * dt), dr = solref[1];
  const double d0 = solimp[0], dw = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  const double x = width > 0 ? fmin(fabs(r) / width, 1.0) : 1.0;
  double y;
  if (power == 1 || d0 == dw) y = x;
  else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
  else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
  d = d0 + y * (dw - d0);
  k = 1.0 / (dw * dw * tc * tc * dr * dr);
  b = 2.0 / (dw * tc);
}

template <int NV>
struct Shared {
  static constexpr int NC = 6 + 2 * NV;
  double qp[NV], qv[NV], ch[NV], sh[NV];
  double pos[NV][3], quat[NV][4], S[NV][6];
  double I10[NV][10], Ic[NV][10];
  double F[NV][6], Fs[NV][6];
  double M[NV][NV];
  double tau[NV], yf[NV], Mq[NV];
  double J[NC][NV], Y[NV][NC], AR[NC][NC], C[NC][NC];
  double rhs[NC], f[NC], bz[NC];
  int idx[NC];
  double att[4][3];
};

struct PArgs {
  const earl_link_model* m;
  int n, nsub;
  double* qpos; double* qvel;
  const double* mocap_pos; const double* mocap_quat; const double* ctrl;
  double* att_xpos; double* qacc_out; double* efc_out;
};

// Cholesky of a dense SPD matrix held in registers (lower triangle, row-major packed), all lanes redundantly
template <int NV>
__device__ __forceinline__ void chol_regs(double (&L)[NV * (NV + 1) / 2]) {
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    double d = L[j * (j + 1) / 2 + j];
#pragma unroll
    for (int p = 0; p < j; ++p) d -= L[j * (j + 1) / 2 + p] * L[j * (j + 1) / 2 + p];
    d = sqrt(d);
    L[j * (j + 1) / 2 + j] = d;
    const double inv = 1.0 / d;
#pragma unroll
    for (int i = j + 1; i < NV; ++i) {
      double s = L[i * (i + 1) / 2 + j];
#pragma unroll
      for (int p = 0; p < j; ++p) s -= L[i * (i + 1) / 2 + p] * L[j * (j + 1) / 2 + p];
      L[i * (i + 1) / 2 + j] = s * inv;
    }
  }
}
template <int NV>
__device__ __forceinline__ void fwd_regs(const double (&L)[NV * (NV + 1) / 2], double (&x)[NV]) {   // L x' = x
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    double s = x[i];
#pragma unroll
    for (int p = 0; p < i; ++p) s -= L[i * (i + 1) / 2 + p] * x[p];
    x[i] = s / L[i * (i + 1) / 2 + i];
  }
}
template <int NV>
__device__ __forceinline__ void bwd_regs(const double (&L)[NV * (NV + 1) / 2], double (&x)[NV]) {   // L^T x' = x
#pragma unroll
  for (int i = NV - 1; i >= 0; --i) {
    double s = x[i];
#pragma unroll
    for (int p = i + 1; p < NV; ++p) s -= L[p * (p + 1) / 2 + i] * x[p];
    x[i] = s / L[i * (i + 1) / 2 + i];
  }
}

// One timestep of one env (the whole wave).  INTEGRATE=false stops after qacc (mj_forward); qacc_out / efc_out may be NULL.
template <int NV, bool INTEGRATE>
__device__ __forceinline__ void substep(Shared<NV>& s, const earl_link_model* __restrict__ m, const int lane, const V3 mpos, const Q4 mq,
                                        const double (&ctrl)[EARL_MAXACT], double* qacc_out, double* efc_out) {
  constexpr int NC = Shared<NV>::NC;
  const double dt = m->dt;
  {
    // ---------------------------------------------------------------- P0: half-angle sin / cos of the hinges
    if (lane < NV) {
      double sn = 0, cs = 1;
      if (m->jtype[lane] == 0) sincos(0.5 * s.qp[lane], &sn, &cs);
      s.ch[lane] = cs; s.sh[lane] = sn;
    }
    fence();
    // ---------------------------------------------------------------- P1: kinematics, lane = link (compose the chain)
    const int l = lane < NV ? lane : NV - 1;
    {
      const uint32_t amask = m->anc_mask[l];
      V3 P{0, 0, 0}; Q4 Q{1, 0, 0, 0};
      V3 axis_w{0, 0, 0}, anchor{0, 0, 0};
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if ((amask >> i) & 1u) {
          double R[3][3];
          qmat(Q, R);
          V3 x = add(P, mulv(R, ld3(m->tpos[i])));
          Q4 q = qmul(Q, Q4{m->tquat[i][0], m->tquat[i][1], m->tquat[i][2], m->tquat[i][3]});
          double Rq[3][3];
          qmat(q, Rq);
          const V3 ax = ld3(m->jaxis[i]), jp = ld3(m->jpos[i]);
          const V3 aw = mulv(Rq, ax);
          const V3 anc = add(x, mulv(Rq, jp));
          if (m->jtype[i] == 0) {
            const double cs = s.ch[i], sn = s.sh[i];
            q = qmul(q, Q4{cs, sn * ax.x, sn * ax.y, sn * ax.z});
            double Rn[3][3];
            qmat(q, Rn);
            x = sub(anc, mulv(Rn, jp));
          } else {
            x = add(x, scl(aw, s.qp[i]));
          }
          if (i == l) { axis_w = aw; anchor = anc; }
          P = x; Q = q;
        }
      }
      // ------------------------------------------------------------ P2: motion subspace + compact spatial inertia
      if (lane < NV) {
        s.pos[l][0] = P.x; s.pos[l][1] = P.y; s.pos[l][2] = P.z;
        s.quat[l][0] = Q.w; s.quat[l][1] = Q.x; s.quat[l][2] = Q.y; s.quat[l][3] = Q.z;
        if (m->jtype[l] == 0) {
          const V3 v = cross(anchor, axis_w);
          s.S[l][0] = axis_w.x; s.S[l][1] = axis_w.y; s.S[l][2] = axis_w.z; s.S[l][3] = v.x; s.S[l][4] = v.y; s.S[l][5] = v.z;
        } else {
          s.S[l][0] = 0; s.S[l][1] = 0; s.S[l][2] = 0; s.S[l][3] = axis_w.x; s.S[l][4] = axis_w.y; s.S[l][5] = axis_w.z;
        }
        double R[3][3];
        qmat(Q, R);
        const double mass = m->mass[l];
        const V3 c = add(P, mulv(R, ld3(m->com[l])));
        const double* in = m->inertia[l];
        const double I[3][3] = {{in[0], in[3], in[4]}, {in[3], in[1], in[5]}, {in[4], in[5], in[2]}};
        double T[3][3], W[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) T[r][cc] = R[r][0] * I[0][cc] + R[r][1] * I[1][cc] + R[r][2] * I[2][cc];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
          for (int cc = 0; cc < 3; ++cc) W[r][cc] = T[r][0] * R[cc][0] + T[r][1] * R[cc][1] + T[r][2] * R[cc][2];
        const double c2 = dot(c, c);
        s.I10[l][0] = mass;
        s.I10[l][1] = mass * c.x; s.I10[l][2] = mass * c.y; s.I10[l][3] = mass * c.z;
        s.I10[l][4] = W[0][0] + mass * (c2 - c.x * c.x);
        s.I10[l][5] = W[1][1] + mass * (c2 - c.y * c.y);
        s.I10[l][6] = W[2][2] + mass * (c2 - c.z * c.z);
        s.I10[l][7] = W[0][1] - mass * c.x * c.y;
        s.I10[l][8] = W[0][2] - mass * c.x * c.z;
        s.I10[l][9] = W[1][2] - mass * c.y * c.z;
      }
    }
    fence();
    // ---------------------------------------------------------------- P3: composite inertias (subtree sums), clear M
    for (int idx = lane; idx < NV * 10; idx += 64) {
      const int li = idx / 10, e = idx % 10;
      const uint32_t dm = m->desc_mask[li];
      double acc = 0;
#pragma unroll
      for (int d = 0; d < NV; ++d) acc += ((dm >> d) & 1u) ? s.I10[d][e] : 0.0;
      s.Ic[li][e] = acc;
    }
    for (int idx = lane; idx < NV * NV; idx += 64) (&s.M[0][0])[idx] = 0.0;
    fence();
    // ---------------------------------------------------------------- P4: mass matrix, lane = (i, j <= i)
    for (int idx = lane; idx < NV * (NV + 1) / 2; idx += 64) {
      int i = 0;
      while ((i + 1) * (i + 2) / 2 <= idx) ++i;
      const int j = idx - i * (i + 1) / 2;
      if ((m->anc_mask[i] >> j) & 1u) {
        V3 n, f;
        iapply(s.Ic[i], ld3(&s.S[i][0]), ld3(&s.S[i][3]), n, f);
        double v = dot(ld3(&s.S[j][0]), n) + dot(ld3(&s.S[j][3]), f);
        if (i == j) v += m->armature[i];
        s.M[i][j] = v; s.M[j][i] = v;
      }
    }
    // ---------------------------------------------------------------- P5: bias forces (RNE), lane = link
    if (lane < NV) {
      const uint32_t amask = m->anc_mask[l];
      V3 w{0, 0, 0}, v{0, 0, 0}, aw{0, 0, 0}, av{-m->gravity[0], -m->gravity[1], -m->gravity[2]};
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        if ((amask >> i) & 1u) {
          const V3 sw = ld3(&s.S[i][0]), sv = ld3(&s.S[i][3]);
          const double qd = s.qv[i];
          aw = add(aw, scl(cross(w, sw), qd));                               // crossm(V) S = [w x sw ; v x sw + w x sv]
          av = add(av, scl(add(cross(v, sw), cross(w, sv)), qd));
          w = add(w, scl(sw, qd));
          v = add(v, scl(sv, qd));
        }
      }
      V3 n1, f1, n2, f2;
      iapply(s.I10[l], aw, av, n1, f1);
      iapply(s.I10[l], w, v, n2, f2);
      const V3 n = add(n1, add(cross(w, n2), cross(v, f2)));                 // crossf(V) [n; f] = [w x n + v x f ; w x f]
      const V3 f = add(f1, cross(w, f2));
      s.F[l][0] = n.x; s.F[l][1] = n.y; s.F[l][2] = n.z; s.F[l][3] = f.x; s.F[l][4] = f.y; s.F[l][5] = f.z;
    }
    fence();
    for (int idx = lane; idx < NV * 6; idx += 64) {
      const int li = idx / 6, e = idx % 6;
      const uint32_t dm = m->desc_mask[li];
      double acc = 0;
#pragma unroll
      for (int d = 0; d < NV; ++d) acc += ((dm >> d) & 1u) ? s.F[d][e] : 0.0;
      s.Fs[li][e] = acc;
    }
    fence();
    if (lane < NV) {
      double bias = 0;
#pragma unroll
      for (int e = 0; e < 6; ++e) bias += s.S[l][e] * s.Fs[l][e];
      double t = -m->damping[l] * s.qv[l] - bias;
      for (int ac = 0; ac < m->n_act; ++ac)
        if (m->act_joint[ac] == l) {
          const double c = fmin(fmax(ctrl[ac], m->act_ctrlrange[ac][0]), m->act_ctrlrange[ac][1]);
          t += m->act_kp[ac] * (c - s.qp[l]);
        }
      s.tau[l] = t;
    }
    fence();
    // ---------------------------------------------------------------- P6: Cholesky of M and a0 = M^-1 tau, in registers
    double L[NV * (NV + 1) / 2];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
      for (int j = 0; j <= i; ++j) L[i * (i + 1) / 2 + j] = s.M[i][j];
    chol_regs<NV>(L);
    double a0[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) a0[i] = s.tau[i];
    fwd_regs<NV>(L, a0);
    bwd_regs<NV>(L, a0);
    // ---------------------------------------------------------------- P7: constraint rows, lane = row
    bool act = false;
    {
      const int k = m->weld_att, la = m->att_link[k];
      double R[3][3];
      const Q4 ql{s.quat[la][0], s.quat[la][1], s.quat[la][2], s.quat[la][3]};
      qmat(ql, R);
      const V3 hp = add(ld3(s.pos[la]), mulv(R, ld3(m->att_pos[k])));
      const Q4 hq = qmul(ql, Q4{m->att_quat[k][0], m->att_quat[k][1], m->att_quat[k][2], m->att_quat[k][3]});
      // weld rows as mj_instantiateEqual builds them (body1 = mocap, body2 = hand, relpose = identity): position error
      // mocap - hand; orientation error = vector part of e = conj(q_hand) * q_mocap with the exact Jacobian of that vector
      // part, -0.5 * (e_w a + a x e_v), a = R_hand^T w_j (no sign flip for e_w < 0)
      const Q4 qe = qmul(Q4{hq.w, -hq.x, -hq.y, -hq.z}, mq);
      const V3 ev{qe.x, qe.y, qe.z};
      double Rh[3][3];
      qmat(hq, Rh);
      const V3 rrot = ev;
      const V3 rpos = sub(mpos, hp);
      if (lane < NC) {
        const int r = lane;
        double Jr[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) Jr[j] = 0.0;
        double res, invw;
        const double *solref, *solimp;
        if (r < 6) {
          const int c = r % 3;
          const uint32_t amask = m->anc_mask[la];
#pragma unroll
          for (int j = 0; j < NV; ++j)
            if ((amask >> j) & 1u) {
              const V3 sw = ld3(&s.S[j][0]), sv = ld3(&s.S[j][3]);
              if (r < 3) {
                const V3 pv = add(sv, cross(sw, hp));
                Jr[j] = -(c == 0 ? pv.x : (c == 1 ? pv.y : pv.z));
              } else {
                const V3 aa{Rh[0][0] * sw.x + Rh[1][0] * sw.y + Rh[2][0] * sw.z, Rh[0][1] * sw.x + Rh[1][1] * sw.y + Rh[2][1] * sw.z,
                            Rh[0][2] * sw.x + Rh[1][2] * sw.y + Rh[2][2] * sw.z};
                const V3 jq = add(scl(aa, qe.w), cross(aa, ev));
                Jr[j] = -0.5 * (c == 0 ? jq.x : (c == 1 ? jq.y : jq.z));
              }
            }
          res = r < 3 ? (c == 0 ? rpos.x : (c == 1 ? rpos.y : rpos.z)) : (c == 0 ? rrot.x : (c == 1 ? rrot.y : rrot.z));
          solref = m->weld_solref; solimp = m->weld_solimp;
          invw = m->weld_invweight[r < 3 ? 0 : 1];
          act = true;
        } else {
          const int j = (r - 6) >> 1, up = (r - 6) & 1;
#pragma unroll
          for (int jj = 0; jj < NV; ++jj) Jr[jj] = jj == j ? (up ? -1.0 : 1.0) : 0.0;
          res = up ? m->range[j][1] - s.qp[j] : s.qp[j] - m->range[j][0];
          solref = m->jsolref[j]; solimp = m->jsolimp[j];
          invw = m->dof_invweight[j];
          act = m->limited[j] && res < 0;
        }
        double Jv = 0, Ja0 = 0;
#pragma unroll
        for (int j = 0; j < NV; ++j) { Jv += Jr[j] * s.qv[j]; Ja0 += Jr[j] * a0[j]; s.J[r][j] = Jr[j]; }
        double kk, bb, dd;
        kbimp(solref, solimp, res, dt, kk, bb, dd);
        const double aref = -bb * Jv - kk * dd * res;
        s.rhs[r] = aref - Ja0;
        s.bz[r] = fmax((1 - dd) / dd * invw, 1e-15);       // regulariser R, parked in bz until AR is built
        fwd_regs<NV>(L, Jr);                                // column r of Y = L^-1 J^T
#pragma unroll
        for (int kx = 0; kx < NV; ++kx) s.Y[kx][r] = Jr[kx];
      }
    }
    fence();
    // ---------------------------------------------------------------- P8: A + R = Y^T Y + diag(R)
    for (int idx = lane; idx < NC * NC; idx += 64) {
      const int r = idx / NC, c = idx % NC;
      double acc = r == c ? s.bz[r] : 0.0;
#pragma unroll
      for (int kx = 0; kx < NV; ++kx) acc += s.Y[kx][r] * s.Y[kx][c];
      s.AR[r][c] = acc;
    }
    fence();
    // ---------------------------------------------------------------- P9: active-set solve (compacted dense Cholesky in LDS)
    for (int it = 0; it < 4; ++it) {
      const unsigned long long am = __ballot(act);
      const int nact = __popcll(am);
      if (lane < NC) {
        s.f[lane] = 0.0;
        if (act) s.idx[__popcll(am & ((1ull << lane) - 1ull))] = lane;
      }
      fence();
      for (int idx = lane; idx < nact * nact; idx += 64) {
        const int p = idx / nact, q = idx % nact;
        s.C[p][q] = s.AR[s.idx[p]][s.idx[q]];
      }
      if (lane < nact) s.bz[lane] = s.rhs[s.idx[lane]];
      fence();
      for (int k = 0; k < nact; ++k) {                      // right-looking Cholesky, then forward substitution fused in
        const double d = sqrt(s.C[k][k]);
        fence();
        if (lane == 0) { s.C[k][k] = d; s.bz[k] = s.bz[k] / d; }
        if (lane > k && lane < nact) s.C[lane][k] = s.C[lane][k] / d;
        fence();
        const double zk = s.bz[k];
        for (int idx = lane; idx < (nact - k - 1) * (nact - k - 1); idx += 64) {
          const int p = k + 1 + idx / (nact - k - 1), q = k + 1 + idx % (nact - k - 1);
          if (q <= p) s.C[p][q] -= s.C[p][k] * s.C[q][k];
        }
        if (lane > k && lane < nact) s.bz[lane] -= s.C[lane][k] * zk;
        fence();
      }
      for (int k = nact - 1; k >= 0; --k) {                 // back substitution L^T x = z
        const double xk = s.bz[k] / s.C[k][k];
        fence();
        if (lane == 0) s.bz[k] = xk;
        if (lane < k) s.bz[lane] -= s.C[k][lane] * xk;
        fence();
      }
      if (lane < nact) s.f[s.idx[lane]] = s.bz[lane];
      fence();
      const bool bad = act && lane >= 6 && lane < NC && s.f[lane] < 0;
      if (!__any(bad)) break;
      if (bad) act = false;
      fence();
    }
    if (lane < NC && !act) s.f[lane] = 0.0;
    fence();
    // ---------------------------------------------------------------- P10: qacc = a0 + L^-T (Y f)
    if (lane < NV) {
      double acc = 0;
#pragma unroll
      for (int c = 0; c < NC; ++c) acc += s.Y[lane][c] * s.f[c];
      s.yf[lane] = acc;
    }
    fence();
    double qacc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) qacc[i] = s.yf[i];
    bwd_regs<NV>(L, qacc);
#pragma unroll
    for (int i = 0; i < NV; ++i) qacc[i] += a0[i];
    if constexpr (!INTEGRATE) {
      if (lane < NV) {
        double v = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) v = lane == i ? qacc[i] : v;
        if (qacc_out) qacc_out[lane] = v;
      }
      if (efc_out && lane < NC) efc_out[lane] = s.f[lane];
    } else {
      // -------------------------------------------------------------- P11: Euler, joint damping implicit
      if (lane < NV) {
        double acc = 0;
#pragma unroll
        for (int j = 0; j < NV; ++j) acc += s.M[lane][j] * qacc[j];
        s.Mq[lane] = acc;
      }
      fence();
#pragma unroll
      for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) L[i * (i + 1) / 2 + j] = s.M[i][j] + (i == j ? dt * m->damping[i] : 0.0);
      chol_regs<NV>(L);
      double qe[NV];
#pragma unroll
      for (int i = 0; i < NV; ++i) qe[i] = s.Mq[i];
      fwd_regs<NV>(L, qe);
      bwd_regs<NV>(L, qe);
      fence();
      if (lane < NV) {
        double v = 0;
#pragma unroll
        for (int i = 0; i < NV; ++i) v = lane == i ? qe[i] : v;
        const double nv_ = s.qv[lane] + dt * v;
        s.qv[lane] = nv_;
        s.qp[lane] = s.qp[lane] + dt * nv_;
      }
      fence();
    }
  }
}

// world position of attachment k from the kinematics currently in LDS
template <int NV>
__device__ __forceinline__ V3 attachment(const Shared<NV>& s, const earl_link_model* __restrict__ m, const int k) {
  const int la = m->att_link[k];
  V3 p = ld3(m->att_pos[k]);
  if (la >= 0) {
    double R[3][3];
    qmat(Q4{s.quat[la][0], s.quat[la][1], s.quat[la][2], s.quat[la][3]}, R);
    p = add(ld3(s.pos[la]), mulv(R, p));
  }
  return p;
}

__device__ __forceinline__ Q4 qnormalize(const Q4& q) {
  const double nrm = 1.0 / sqrt(q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z);
  return {q.w * nrm, q.x * nrm, q.y * nrm, q.z * nrm};
}

template <int NV, bool INTEGRATE>
__global__ __launch_bounds__(64) void physics_kernel(const PArgs a) {
  __shared__ Shared<NV> s;
  const earl_link_model* __restrict__ m = a.m;
  const int env = blockIdx.x, lane = threadIdx.x;
  if (lane < NV) {
    s.qp[lane] = a.qpos[(size_t)env * NV + lane];
    s.qv[lane] = a.qvel[(size_t)env * NV + lane];
  }
  fence();
  const V3 mpos = ld3(a.mocap_pos + (size_t)env * 3);
  const Q4 mq = qnormalize(Q4{a.mocap_quat[(size_t)env * 4], a.mocap_quat[(size_t)env * 4 + 1], a.mocap_quat[(size_t)env * 4 + 2], a.mocap_quat[(size_t)env * 4 + 3]});
  double ctrl[EARL_MAXACT] = {0, 0, 0, 0};
  for (int ac = 0; ac < m->n_act; ++ac) ctrl[ac] = a.ctrl[(size_t)env * m->n_act + ac];
  for (int ts = 0; ts < a.nsub; ++ts)
    substep<NV, INTEGRATE>(s, m, lane, mpos, mq, ctrl, a.qacc_out ? a.qacc_out + (size_t)env * NV : nullptr,
                           a.efc_out ? a.efc_out + (size_t)env * Shared<NV>::NC : nullptr);
  if constexpr (INTEGRATE) {
    if (lane < NV) {
      a.qpos[(size_t)env * NV + lane] = s.qp[lane];
      a.qvel[(size_t)env * NV + lane] = s.qv[lane];
    }
  }
  // attachments at the kinematics of the LAST timestep's start (what mj_step leaves in data.xpos / site_xpos)
  if (a.att_xpos && lane < m->n_att) {
    const V3 p = attachment<NV>(s, m, lane);
    double* o = a.att_xpos + ((size_t)env * m->n_att + lane) * 3;
    o[0] = p.x; o[1] = p.y; o[2] = p.z;
  }
}


// ------------------------------------------------------------------------------------------------ Sawyer env kernels
struct SawyerArgs {
  const earl_link_model* m;
  earl_sawyer_cfg cfg;
  earl_sawyer_state st;
  const float* action; int T;
  earl_sawyer_out out;
  const double* reset_qpos; const double* reset_qvel; const uint8_t* mask; double* reset_obs;
  int observe_only;
};

// metaworld reward_utils.tolerance(x, bounds=(0, hi), margin, sigmoid='gaussian') [UPSTREAM, dm_control semantics; unpinned]
__device__ __forceinline__ double tolerance_gaussian(double x, double hi, double margin) {
  if (0.0 <= x && x <= hi) return 1.0;
  if (margin == 0) return 0.0;
  const double d = (x < 0.0 ? -x : x - hi) / margin;
  const double scale = sqrt(-2.0 * log(0.1));
  return exp(-0.5 * (d * scale) * (d * scale));
}

// obs[14] + reward + flags of one env from the kinematics in LDS (sawyer_door.py:86-94, :141-177); all lanes call it
template <int NV>
__device__ __forceinline__ void sawyer_emit(Shared<NV>& s, const earl_link_model* __restrict__ m, const earl_sawyer_cfg& cfg,
                                            const int lane, const double* __restrict__ goal, double* __restrict__ obs,
                                            float* reward, uint8_t* success) {
  if (lane < 4) {
    const int k = lane == 0 ? cfg.att_hand : (lane == 1 ? cfg.att_right : (lane == 2 ? cfg.att_left : cfg.att_obj));
    const V3 p = attachment<NV>(s, m, k);
    s.att[lane][0] = p.x; s.att[lane][1] = p.y; s.att[lane][2] = p.z;
  }
  fence();
  if (lane < 14) {
    double v;
    if (lane < 3) v = s.att[0][lane];
    else if (lane == 3) {
      const V3 d = sub(ld3(s.att[1]), ld3(s.att[2]));
      v = fmin(fmax(sqrt(dot(d, d)) / 0.1, 0.0), 1.0);
    } else if (lane < 7) v = s.att[3][lane - 4];
    else v = goal[lane - 7];
    obs[lane] = v;
  }
  if (lane == 0) {
    const V3 tcp = ld3(s.att[0]), obj = ld3(s.att[3]), target = ld3(goal + 4);
    const V3 d = sub(obj, target);
    const double obj_to_target = sqrt(dot(d, d));                 // np.linalg.norm in f64
    const bool ok = obj_to_target <= cfg.success_radius;
    double r = ok ? 1.0 : 0.0;
    if (cfg.reward_type != 0) {
      const V3 e = sub(tcp, obj);
      const double tcp_to_obj = sqrt(dot(e, e));
      const V3 oi = sub(ld3(cfg.obj_init_pos), target), hi = sub(ld3(cfg.hand_init_pos), obj);
      const double in_place = tolerance_gaussian(obj_to_target, 0.05, sqrt(dot(oi, oi)));
      const double hand_in_place = tolerance_gaussian(tcp_to_obj, 0.25 * 0.05, sqrt(dot(hi, hi)) + 0.1);
      r = 3 * hand_in_place + 6 * in_place;
      if (obj_to_target < 0.05) r = 10;
    }
    if (reward) *reward = (float)r;
    if (success) *success = ok ? 1 : 0;
  }
  fence();
}

template <int NV>
__global__ __launch_bounds__(64) void sawyer_rollout_kernel(const SawyerArgs a) {
  __shared__ Shared<NV> s;
  const earl_link_model* __restrict__ m = a.m;
  const earl_sawyer_cfg& cfg = a.cfg;
  const int env = blockIdx.x, lane = threadIdx.x, n = cfg.n;
  if (lane < NV) {
    s.qp[lane] = a.st.qpos[(size_t)env * NV + lane];
    s.qv[lane] = a.st.qvel[(size_t)env * NV + lane];
  }
  fence();
  V3 mpos = ld3(a.st.mocap_pos + (size_t)env * 3);
  const Q4 mq = qnormalize(Q4{cfg.mocap_quat[0], cfg.mocap_quat[1], cfg.mocap_quat[2], cfg.mocap_quat[3]});
  int steps = a.st.steps_since_reset ? a.st.steps_since_reset[env] : 0;
  const float scale = (float)cfg.action_scale;
  for (int t = 0; t < a.T; ++t) {
    const float4 act = *reinterpret_cast<const float4*>(a.action + ((size_t)t * n + env) * 4);
    // set_xyz_action [UPSTREAM]: clip, float32 product with the scale, float64 add, box clip
    const float cx = fminf(fmaxf(act.x, -1.f), 1.f) * scale, cy = fminf(fmaxf(act.y, -1.f), 1.f) * scale, cz = fminf(fmaxf(act.z, -1.f), 1.f) * scale;
    mpos.x = fmin(fmax(mpos.x + (double)cx, cfg.mocap_low[0]), cfg.mocap_high[0]);
    mpos.y = fmin(fmax(mpos.y + (double)cy, cfg.mocap_low[1]), cfg.mocap_high[1]);
    mpos.z = fmin(fmax(mpos.z + (double)cz, cfg.mocap_low[2]), cfg.mocap_high[2]);
    const double ctrl[EARL_MAXACT] = {(double)act.w, -(double)act.w, 0, 0};
    for (int ts = 0; ts < cfg.frame_skip; ++ts) substep<NV, true>(s, m, lane, mpos, mq, ctrl, nullptr, nullptr);
    const size_t row = (size_t)t * n + env;
    sawyer_emit<NV>(s, m, cfg, lane, a.st.goal + (size_t)env * 7, a.out.obs + row * 14, a.out.reward ? a.out.reward + row : nullptr,
                    a.out.success ? a.out.success + row : nullptr);
    ++steps;
    if (lane == 0 && a.out.done) a.out.done[row] = (cfg.horizon > 0 && steps >= cfg.horizon) ? 1 : 0;
  }
  if (lane < NV) {
    a.st.qpos[(size_t)env * NV + lane] = s.qp[lane];
    a.st.qvel[(size_t)env * NV + lane] = s.qv[lane];
  }
  if (lane < 3) a.st.mocap_pos[(size_t)env * 3 + lane] = lane == 0 ? mpos.x : (lane == 1 ? mpos.y : mpos.z);
  if (lane == 0 && a.st.steps_since_reset) a.st.steps_since_reset[env] = steps;
}

template <int NV>
__global__ __launch_bounds__(64) void sawyer_reset_kernel(const SawyerArgs a) {
  __shared__ Shared<NV> s;
  const earl_link_model* __restrict__ m = a.m;
  const earl_sawyer_cfg& cfg = a.cfg;
  const int env = blockIdx.x, lane = threadIdx.x;
  if (a.observe_only) {
    if (lane < NV) {
      s.qp[lane] = a.st.qpos[(size_t)env * NV + lane];
      s.qv[lane] = a.st.qvel[(size_t)env * NV + lane];
    }
    fence();
    const Q4 mq = qnormalize(Q4{cfg.mocap_quat[0], cfg.mocap_quat[1], cfg.mocap_quat[2], cfg.mocap_quat[3]});
    const double ctrl[EARL_MAXACT] = {0, 0, 0, 0};
    substep<NV, false>(s, m, lane, ld3(a.st.mocap_pos + (size_t)env * 3), mq, ctrl, nullptr, nullptr);
    sawyer_emit<NV>(s, m, cfg, lane, a.st.goal + (size_t)env * 7, a.reset_obs + (size_t)env * 14, nullptr, nullptr);
    return;
  }
  if (a.mask && !a.mask[env]) return;
  const earl::U4 b = earl::philox4x32_10(earl::U4{0u, (uint32_t)(cfg.env_offset + env), (uint32_t)cfg.counter, (uint32_t)(cfg.counter >> 32)},
                                         (uint32_t)cfg.seed, (uint32_t)(cfg.seed >> 32));
  // np.random.uniform(lo, hi) = lo + (hi - lo) * u   (sawyer_door.py:116-118)
  const double angle = cfg.obj_init_angle + (cfg.angle_noise[0] + (cfg.angle_noise[1] - cfg.angle_noise[0]) * earl::u01(b.x, b.y));
  if (lane < NV) {
    s.qp[lane] = lane == cfg.obj_dof ? angle : a.reset_qpos[lane];
    s.qv[lane] = lane == cfg.obj_dof ? 0.0 : a.reset_qvel[lane];
    a.st.qpos[(size_t)env * NV + lane] = s.qp[lane];
    a.st.qvel[(size_t)env * NV + lane] = s.qv[lane];
  }
  if (lane < 3) a.st.mocap_pos[(size_t)env * 3 + lane] = cfg.hand_init_pos[lane];
  if (lane == 0 && a.st.steps_since_reset) a.st.steps_since_reset[env] = 0;
  fence();
  if (a.reset_obs) {
    // set_state -> sim.forward(): kinematics of the state just written
    const Q4 mq = qnormalize(Q4{cfg.mocap_quat[0], cfg.mocap_quat[1], cfg.mocap_quat[2], cfg.mocap_quat[3]});
    const double ctrl[EARL_MAXACT] = {0, 0, 0, 0};
    substep<NV, false>(s, m, lane, ld3(cfg.hand_init_pos), mq, ctrl, nullptr, nullptr);
    sawyer_emit<NV>(s, m, cfg, lane, a.st.goal + (size_t)env * 7, a.reset_obs + (size_t)env * 14, nullptr, nullptr);
  }
}

// compute_reward / is_successful on given observations (sawyer_door.py:141-177), one lane per row
__global__ void sawyer_door_reward_kernel(const int n, const double* __restrict__ obs, const earl_sawyer_cfg cfg, float* __restrict__ reward,
                                          uint8_t* __restrict__ success) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* o = obs + (size_t)i * 14;
  const V3 tcp = ld3(o), obj = ld3(o + 4), target = ld3(o + 11);
  const V3 d = sub(obj, target);
  const double obj_to_target = sqrt(dot(d, d));
  const bool ok = obj_to_target <= cfg.success_radius;
  double r = ok ? 1.0 : 0.0;
  if (cfg.reward_type != 0) {
    const V3 e = sub(tcp, obj);
    const V3 oi = sub(ld3(cfg.obj_init_pos), target), hi = sub(ld3(cfg.hand_init_pos), obj);
    const double in_place = tolerance_gaussian(obj_to_target, 0.05, sqrt(dot(oi, oi)));
    const double hand_in_place = tolerance_gaussian(sqrt(dot(e, e)), 0.25 * 0.05, sqrt(dot(hi, hi)) + 0.1);
    r = 3 * hand_in_place + 6 * in_place;
    if (obj_to_target < 0.05) r = 10;
  }
  if (reward) reward[i] = (float)r;
  if (success) success[i] = ok ? 1 : 0;
}

int launched(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    fprintf(stderr, "earl_physics: %s: %s\n", what, hipGetErrorString(e));
    return EARL_ERR_LAUNCH;
  }
  return EARL_OK;
}

}  // namespace

extern "C" {

int earl_physics_step(const earl_link_model* model, int32_t nv, int32_t n, int32_t nsub, double* qpos, double* qvel,
                      const double* mocap_pos, const double* mocap_quat, const double* ctrl, double* att_xpos,
                      earl_stream_t stream) {
  if (!model || n < 0 || nsub < 0 || !qpos || !qvel || !mocap_pos || !mocap_quat || !ctrl) return EARL_ERR_ARG;
  if (n == 0 || nsub == 0) return EARL_OK;
  PArgs a{model, n, nsub, qpos, qvel, mocap_pos, mocap_quat, ctrl, att_xpos, nullptr, nullptr};
  if (nv == 10) physics_kernel<10, true><<<n, 64, 0, (hipStream_t)stream>>>(a);
  else return EARL_ERR_ARG;
  return launched("physics_step");
}

int earl_physics_forward(const earl_link_model* model, int32_t nv, int32_t n, const double* qpos, const double* qvel,
                         const double* mocap_pos, const double* mocap_quat, const double* ctrl, double* qacc,
                         double* efc_force, double* att_xpos, earl_stream_t stream) {
  if (!model || n < 0 || !qpos || !qvel || !mocap_pos || !mocap_quat || !ctrl || !qacc) return EARL_ERR_ARG;
  if (n == 0) return EARL_OK;
  PArgs a{model, n, 1, const_cast<double*>(qpos), const_cast<double*>(qvel), mocap_pos, mocap_quat, ctrl, att_xpos, qacc, efc_force};
  if (nv == 10) physics_kernel<10, false><<<n, 64, 0, (hipStream_t)stream>>>(a);
  else return EARL_ERR_ARG;
  return launched("physics_forward");
}

int earl_sawyer_rollout(const earl_link_model* model, int32_t nv, const earl_sawyer_cfg* cfg, const earl_sawyer_state* st,
                        const float* action, int32_t T, const earl_sawyer_out* out, earl_stream_t stream) {
  if (!model || !cfg || !st || !out || !action || T < 0 || cfg->n < 0) return EARL_ERR_ARG;
  if (!st->qpos || !st->qvel || !st->mocap_pos || !st->goal || !out->obs) return EARL_ERR_ARG;
  if (cfg->frame_skip < 0 || cfg->att_hand < 0 || cfg->att_right < 0 || cfg->att_left < 0 || cfg->att_obj < 0) return EARL_ERR_ARG;
  if (cfg->n == 0 || T == 0) return EARL_OK;
  SawyerArgs a{model, *cfg, *st, action, T, *out, nullptr, nullptr, nullptr, nullptr, 0};
  if (nv == 10) sawyer_rollout_kernel<10><<<cfg->n, 64, 0, (hipStream_t)stream>>>(a);
  else return EARL_ERR_ARG;
  return launched("sawyer_rollout");
}

int earl_sawyer_reset(const earl_link_model* model, int32_t nv, const earl_sawyer_cfg* cfg, const earl_sawyer_state* st,
                      const double* reset_qpos, const double* reset_qvel, const uint8_t* mask, double* obs,
                      earl_stream_t stream) {
  if (!model || !cfg || !st || !reset_qpos || !reset_qvel || cfg->n < 0) return EARL_ERR_ARG;
  if (!st->qpos || !st->qvel || !st->mocap_pos || !st->goal) return EARL_ERR_ARG;
  if (cfg->obj_dof < 0 || cfg->obj_dof >= nv) return EARL_ERR_ARG;
  if (cfg->n == 0) return EARL_OK;
  SawyerArgs a{model, *cfg, *st, nullptr, 0, earl_sawyer_out{nullptr, nullptr, nullptr, nullptr}, reset_qpos, reset_qvel, mask, obs, 0};
  if (nv == 10) sawyer_reset_kernel<10><<<cfg->n, 64, 0, (hipStream_t)stream>>>(a);
  else return EARL_ERR_ARG;
  return launched("sawyer_reset");
}

int earl_sawyer_observe(const earl_link_model* model, int32_t nv, const earl_sawyer_cfg* cfg, const earl_sawyer_state* st, double* obs,
                        earl_stream_t stream) {
  if (!model || !cfg || !st || !obs || cfg->n < 0) return EARL_ERR_ARG;
  if (!st->qpos || !st->qvel || !st->mocap_pos || !st->goal) return EARL_ERR_ARG;
  if (cfg->n == 0) return EARL_OK;
  SawyerArgs a{model, *cfg, *st, nullptr, 0, earl_sawyer_out{nullptr, nullptr, nullptr, nullptr}, nullptr, nullptr, nullptr, obs, 1};
  if (nv == 10) sawyer_reset_kernel<10><<<cfg->n, 64, 0, (hipStream_t)stream>>>(a);
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

int earl_physics_model_size(void) { return (int)sizeof(earl_link_model); }

}  // extern "C"
