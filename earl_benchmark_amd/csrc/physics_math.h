// physics_math.h -- small fp64 vector / quaternion algebra, reciprocal / root iterations, impedance formulas
// A section of csrc/physics.hip (included there, inside its anonymous namespace): split out in round 5.

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
__device__ __forceinline__ V3 mulvT(const double (&R)[3][3], const V3& v) {
  return {R[0][0] * v.x + R[1][0] * v.y + R[2][0] * v.z, R[0][1] * v.x + R[1][1] * v.y + R[2][1] * v.z,
          R[0][2] * v.x + R[1][2] * v.y + R[2][2] * v.z};
}
__device__ __forceinline__ V3 cross(const V3& a, const V3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ V3 add(const V3& a, const V3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 vsub(const V3& a, const V3& b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 scl(const V3& a, double s) { return {a.x * s, a.y * s, a.z * s}; }
__device__ __forceinline__ double dot(const V3& a, const V3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ V3 ld3(const double* p) { return {p[0], p[1], p[2]}; }
__device__ __forceinline__ Q4 ldq(const double* p) { return {p[0], p[1], p[2], p[3]}; }
__device__ __forceinline__ double pick3(const V3& v, int c) { return c == 0 ? v.x : (c == 1 ? v.y : v.z); }
// selects of whole vectors, component by component: a `cond ? V3 : V3` on the structs is lowered by hipcc to a select of two
// stack ADDRESSES and a load through scratch memory (store both, wait, load one) -- seen in the ISA of every phase that had one
__device__ __forceinline__ V3 selv(const bool c, const V3& a, const V3& b) { return {c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z}; }
__device__ __forceinline__ Q4 selq(const bool c, const Q4& a, const Q4& b) { return {c ? a.w : b.w, c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z}; }

// reciprocal / reciprocal square root: hardware seed + Newton steps (about 1 ulp; not correctly rounded -- fine here)
// a value the compiler must hold in a register HERE: keeps an LDS load out of a branch the optimiser would otherwise sink it into (a conditional load has its own
// s_waitcnt; in a one-wave-per-SIMD kernel every such wait is a full LDS round trip)
__device__ __forceinline__ double pinned(double v) {
  asm volatile("" : "+v"(v));
  return v;
}
// ... and a whole batch of loaded values at once: every load of the batch is issued before the first use, ONE s_waitcnt instead of one per pair of loads (the
// scheduler, minimising register pressure, otherwise interleaves "read two, wait, use": in a one-wave-per-SIMD kernel a chain of full LDS round trips)
__device__ __forceinline__ void pin6(double& a, double& b, double& c, double& d, double& e, double& f) {
  asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f));
}
template <int N>
__device__ __forceinline__ void pin_batch(double (&x)[N]) {
#pragma unroll
  for (int i = 0; i + 6 <= N; i += 6) pin6(x[i], x[i + 1], x[i + 2], x[i + 3], x[i + 4], x[i + 5]);
#pragma unroll
  for (int i = N - N % 6; i < N; ++i) asm volatile("" : "+v"(x[i]));
}
__device__ __forceinline__ double rcp_nr(double x) {
  double r = __builtin_amdgcn_rcp(x);
  r = fma(fma(-x, r, 1.0), r, r);
  r = fma(fma(-x, r, 1.0), r, r);
  return r;
}
__device__ __forceinline__ double rsq_nr(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}

// the same with TWO Newton steps: the hardware seed is good to 2^-26 or better, so two steps already reach double precision (~1 ulp); used by the
// factorisations of the bigger models (nv > 10), where fifteen to thirty of these chains stand in a row on the timestep's critical path.  (The
// door model keeps rsq_nr: its two builds are pinned bit for bit against round 2's outputs.)
__device__ __forceinline__ double rsq2(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * fma(-0.5 * x * y, y, 1.5);
  y = y * fma(-0.5 * x * y, y, 1.5);
  return y;
}

// sin / cos for moderate arguments (|x| < ~1e3; joint half-angles are < 3): Cody-Waite reduction by pi/2 and the usual
// minimax kernels on [-pi/4, pi/4] (the coefficient sets are the classic fdlibm ones), quadrant fix-up by selects
__device__ __forceinline__ void sincos_mod(double x, double& sn, double& cs) {
  const double k = rint(x * 6.36619772367581382433e-01);
  double r = fma(-k, 1.57079632673412561417e+00, x);
  r = fma(-k, 6.07710050650619224932e-11, r);      // pi/2 = 1.57079632673412561417 + 6.07710050650619224932e-11 (to 1e-27)
  const double z = r * r;
  double ps = fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
  ps = fma(z, ps, 2.75573137070700676789e-06);
  ps = fma(z, ps, -1.98412698298579493134e-04);
  ps = fma(z, ps, 8.33333333332248946124e-03);
  ps = fma(z, ps, -1.66666666666666324348e-01);
  const double sr = fma(r * z, ps, r);
  double pc = fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
  pc = fma(z, pc, -2.75573143513906633035e-07);
  pc = fma(z, pc, 2.48015872894767294178e-05);
  pc = fma(z, pc, -1.38888888888741095749e-03);
  pc = fma(z, pc, 4.16666666666666019037e-02);
  const double cr = fma(z * z, pc, fma(-0.5, z, 1.0));
  const int q = (int)k & 3;
  sn = (q == 0) ? sr : ((q == 1) ? cr : ((q == 2) ? -sr : -cr));
  cs = (q == 0) ? cr : ((q == 1) ? -sr : ((q == 2) ? -cr : sr));
}

// compact spatial inertia about the world origin: I10 = {m, h = m c (3), Io xx yy zz xy xz yz}; I [w; v] = [n; f]
__device__ __forceinline__ void iapply(const double* I, const V3& w, const V3& v, V3& n, V3& f) {
  const V3 h = ld3(I + 1);
  n = add(V3{I[4] * w.x + I[7] * w.y + I[8] * w.z, I[7] * w.x + I[5] * w.y + I[9] * w.z, I[8] * w.x + I[9] * w.y + I[6] * w.z}, cross(h, v));
  f = vsub(scl(v, I[0]), cross(h, w));
}

// MuJoCo impedance d(r) and (k, b) of a soft constraint row (reference: physics_oracle.kbimp).  (k, b) depend on the row's solref / solimp and the
// timestep only: the kernels compute them ONCE per launch into the block table (stage_kb) instead of in every timestep -- two reciprocals with their
// Newton steps, a chain of ~25 dependent operations per row kind
__device__ __forceinline__ void kb_of(const double* solref, const double* solimp, double dt, double& k, double& b) {
  const double tc = fmax(solref[0], 2 * dt), dr = solref[1], dw = solimp[1];
  k = rcp_nr(dw * dw * tc * tc * dr * dr);
  b = 2.0 * rcp_nr(dw * tc);
}
__device__ __forceinline__ void kbimp(const double* solref, const double* solimp, double r, double dt, double& k, double& b, double& d) {   // (all three, per call: the peg build)
  const double tc = fmax(solref[0], 2 * dt), dr = solref[1];
  const double d0 = solimp[0], dw = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  const double x = width > 0 ? fmin(fabs(r) * rcp_nr(width), 1.0) : 1.0;
  double y;
  if (power == 1 || d0 == dw) y = x;
  else if (power == 2) y = x <= mid ? x * x * rcp_nr(mid) : 1 - (1 - x) * (1 - x) * rcp_nr(1 - mid);
  else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
  else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
  d = d0 + y * (dw - d0);
  k = rcp_nr(dw * dw * tc * tc * dr * dr);
  b = 2.0 * rcp_nr(dw * tc);
}
__device__ __forceinline__ double imp_of(const double* solimp, double r) {
  const double d0 = solimp[0], dw = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  const double x = width > 0 ? fmin(fabs(r) * rcp_nr(width), 1.0) : 1.0;
  double y;
  if (power == 1 || d0 == dw) y = x;
  else if (power == 2) y = x <= mid ? x * x * rcp_nr(mid) : 1 - (1 - x) * (1 - x) * rcp_nr(1 - mid);
  else if (x <= mid) y = pow(x, power) / pow(mid, power - 1);
  else y = 1 - pow(1 - x, power) / pow(1 - mid, power - 1);
  return d0 + y * (dw - d0);
}
// MuJoCo's impedance for solimp powers 1 and 2 only -- the kitchen's and the minitaur's tables (the host side refuses others for them: physics/__init__.py
// check_impedance_powers): imp_of without its pow() branches, which are never taken there, are a quarter of the timestep's code and stand between chains of
// dependent operations the scheduler could otherwise run side by side
__device__ __forceinline__ double imp_p2(const double* solimp, double r) {
  const double d0 = solimp[0], dw = solimp[1], width = solimp[2], mid = solimp[3], power = solimp[4];
  const double x = width > 0 ? fmin(fabs(r) * rcp_nr(width), 1.0) : 1.0;
  const double y2 = x <= mid ? x * x * rcp_nr(mid) : 1 - (1 - x) * (1 - x) * rcp_nr(1 - mid);
  const double y = (power == 1 || d0 == dw) ? x : y2;
  return d0 + y * (dw - d0);
}

