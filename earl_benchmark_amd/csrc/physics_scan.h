// physics_scan.h -- DPP row shifts and the prefix / suffix scans along the arm chain (Lim<NV>::ARMSCAN)
// A section of csrc/physics.hip (included there, inside its anonymous namespace): split out in round 5.

// DPP moves within a 16-lane row: lane l reads lane l - K (shr) / l + K (shl) of its row, 0 beyond the row
template <int CTRL>
__device__ __forceinline__ double dpp_row(const double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
template <int CTRL> __device__ __forceinline__ V3 dpp_row(const V3& v) { return {dpp_row<CTRL>(v.x), dpp_row<CTRL>(v.y), dpp_row<CTRL>(v.z)}; }
template <int CTRL> __device__ __forceinline__ Q4 dpp_row(const Q4& q) { return {dpp_row<CTRL>(q.w), dpp_row<CTRL>(q.x), dpp_row<CTRL>(q.y), dpp_row<CTRL>(q.z)}; }
constexpr int DPP_SHR(int k) { return 0x110 + k; }
constexpr int DPP_SHL(int k) { return 0x100 + k; }
// chain membership of a lane: the arm's serial part [0, 6], the free body's chain [BODY0, BODY0 + 5]
template <int NV> __device__ __forceinline__ bool scan_from_below(const int sub, const int k) {     // lane sub - k is sub's ancestor at distance k
  return (sub <= 6 && sub >= k) || (Lim<NV>::BODY0 >= 0 && sub >= Lim<NV>::BODY0 + k && sub <= Lim<NV>::BODY0 + 5);
}
template <int NV> __device__ __forceinline__ bool scan_from_above(const int sub, const int k) {     // lane sub + k is sub's descendant at distance k
  return (sub + k <= 6) || (Lim<NV>::BODY0 >= 0 && sub >= Lim<NV>::BODY0 && sub + k <= Lim<NV>::BODY0 + 5);
}
// inclusive prefix sums along the chains (ancestors incl. the link itself), then the fingers take the hand's
template <int NV>
__device__ __forceinline__ void scan_anc(V3& a, V3& b, const int sub) {
#define EARL_SCAN_ROUND(K) { const V3 as_ = dpp_row<DPP_SHR(K)>(a), bs_ = dpp_row<DPP_SHR(K)>(b); const bool on = scan_from_below<NV>(sub, K); a = selv(on, add(a, as_), a); b = selv(on, add(b, bs_), b); }
  EARL_SCAN_ROUND(1) EARL_SCAN_ROUND(2) EARL_SCAN_ROUND(4)
#undef EARL_SCAN_ROUND
  const V3 a1 = dpp_row<DPP_SHR(1)>(a), a2 = dpp_row<DPP_SHR(2)>(a), b1 = dpp_row<DPP_SHR(1)>(b), b2 = dpp_row<DPP_SHR(2)>(b);
  a = selv(sub == 7, add(a1, a), selv(sub == 8, add(a2, a), a));
  b = selv(sub == 7, add(b1, b), selv(sub == 8, add(b2, b), b));
}
// inclusive suffix sums (the link's subtree): the fingers fold into the hand first
template <int NV, int N>
__device__ __forceinline__ void scan_desc(double (&x)[N], const int sub) {
#pragma unroll
  for (int e = 0; e < N; ++e) {
    const double f1 = dpp_row<DPP_SHL(1)>(x[e]), f2 = dpp_row<DPP_SHL(2)>(x[e]);
    x[e] = sub == 6 ? x[e] + f1 + f2 : x[e];
  }
#define EARL_SCAN_ROUND(K) { _Pragma("unroll") for (int e = 0; e < N; ++e) { const double xs_ = dpp_row<DPP_SHL(K)>(x[e]); x[e] = scan_from_above<NV>(sub, K) ? x[e] + xs_ : x[e]; } }
  EARL_SCAN_ROUND(1) EARL_SCAN_ROUND(2) EARL_SCAN_ROUND(4)
#undef EARL_SCAN_ROUND
}

