// spx_group_common.hpp -- device code shared by the group kernels (spx_group.hip: register tiles, wavefront / workgroup per
// group; spx_group_team.hip: a TEAM OF WORKGROUPS per group): team reductions, element providers, the Binf root find
// (src/shiftedGroupNormL2Binf.jl:85-108) and the per-group body (src/shiftedGroupNormL2.jl:67-76, shiftedGroupNormL2Binf.jl:84-117).
#pragma once
#include <cmath>

#include "spx_common.hpp"

// ---------------------------------------------------------------------------------------------
// team reductions: TEAM lanes (8, 16, 32, 64: aligned lane ranges of one wave; 256: the workgroup)
// ---------------------------------------------------------------------------------------------
// (dpp_f64, wave_sum_dpp, wave_max_dpp, fold16_sum: spx_common.hpp)
template <int TEAM>
__device__ __forceinline__ double lanes_sum(double v) {
  static_assert(TEAM == 1 || TEAM == 2 || TEAM == 4 || TEAM == 8 || TEAM == 16 || TEAM == 32 || TEAM == 64, "TEAM");
  if constexpr (TEAM >= 2) v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  if constexpr (TEAM >= 4) v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  if constexpr (TEAM >= 8) v += dpp_f64<0x141>(v);  // row_half_mirror: the other quad of each 8
  if constexpr (TEAM >= 16) v += dpp_f64<0x140>(v);  // row_mirror: the other half of each 16-lane row
  if constexpr (TEAM >= 32) v += __shfl_xor(v, 16, 64);
  if constexpr (TEAM >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}
// ---------------------------------------------------------------------------------------------
// TEAM == kTeamGrid: the team is a range of 1024-lane WORKGROUPS of one resident grid (spx_group_team.hip: a group too
// large for one workgroup -- first of all the reference's default GroupNormL2, one group over the whole vector).  The `lds`
// argument of the team reductions then points at a GridTeam in LDS.  A reduction = wavefront butterflies, the 16 wave slots
// in a fixed order, then -- for a team of more than one workgroup -- an exchange of the workgroup totals through words in
// spx_ctx::sync that carry their own ready flag (bits(v) + 1, never 0: the idea of spx_b2.hip), read back by the lanes
// t < W and combined in the same fixed shape: every workgroup of the team forms bit-identical totals, so the scalar
// decisions that follow are taken alike everywhere.  Rows are reused cyclically: a workgroup clears its own words of row
// k - 1 once it has read all of row k (every member stored into row k after it was done with row k - 1); the words of the
// last two reductions are left behind and the next launch, which works on the other set, clears them.
// ---------------------------------------------------------------------------------------------
constexpr int kTeamGrid = 1024;
constexpr int kGtRows = 8;      // exchange rows, reused cyclically
constexpr int kGtCols = 256;    // workgroups of a launch at most
constexpr int kGtWords = 16;    // words per workgroup and row (one 128-byte line)
constexpr int kGtVals = 10;     // values per reduction at most
constexpr size_t kGtSetWords = (size_t)kGtRows * kGtCols * kGtWords;
static_assert(2 * kGtSetWords * sizeof(unsigned long long) + 2 * kGtCols * sizeof(unsigned int) <= kSpxSyncTeamBytes, "team exchange words");
struct GridTeam {
  double slot[2][kGtVals][16];  // wave totals; two sets, alternating between calls (one workgroup barrier per combine)
  unsigned long long* rows;     // this launch's set of exchange words
  SpxSyncHeader* hdr;
  int first, W;                 // the team = workgroups [first, first + W) of the grid
  // per WAVEFRONT copies (lane 0 of a wavefront writes its own, the wavefront reads its own: program order, no barrier):
  int wnp[16];                  // reductions this team has exchanged so far in this launch (the row of the exchange words)
  int wcalls[16];               // workgroup combines so far (parity = slot set)
};
// x[k] <- sum / max over the 1024 lanes of the workgroup, the same bits in every lane: wavefront reductions, one slot per
// wavefront, ONE barrier, then every 16-lane row reads the 16 slots and folds them by DPP (a fixed shape; every lane reading
// all 16 slots itself cost 16 NV LDS instructions per wavefront: 5.7 us for ten values, tools/exp/team_reduce.hip).
// `set`: the slot set of this call (the callers alternate, so that no barrier is needed before the slots are rewritten).
template <int NV>
__device__ __forceinline__ void grid_team_block_combine(GridTeam* gt, double (&x)[NV], unsigned int maxmask, int set) {
  const int t = threadIdx.x, w = t >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) x[k] = ((maxmask >> k) & 1u) ? wave_max_dpp(x[k]) : wave_sum_dpp(x[k]);
  if ((t & 63) == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) gt->slot[set][k][w] = x[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    double v = gt->slot[set][k][t & 15];
    if ((maxmask >> k) & 1u) {
      v = fmax(v, dpp_f64<0xB1>(v));
      v = fmax(v, dpp_f64<0x4E>(v));
      v = fmax(v, dpp_f64<0x141>(v));
      v = fmax(v, dpp_f64<0x140>(v));
    } else {
      v += dpp_f64<0xB1>(v);
      v += dpp_f64<0x4E>(v);
      v += dpp_f64<0x141>(v);
      v += dpp_f64<0x140>(v);
    }
    x[k] = v;
  }
}
// v[k] <- sum (or max, where bit k of maxmask is set; NaN-free values) of v[k] over all lanes of all workgroups of the team
template <int NV>
__device__ __forceinline__ void grid_team_reduce(GridTeam* gt, double (&v)[NV], unsigned int maxmask) {
  static_assert(NV <= kGtVals && NV <= kGtWords, "NV");
  const int t = threadIdx.x, w = t >> 6;
  const int calls = gt->wcalls[w];
  grid_team_block_combine<NV>(gt, v, maxmask, calls & 1);
  const int W = gt->W;
  if (W == 1) {
    if ((t & 63) == 0) gt->wcalls[w] = calls + 1;
    return;
  }
  const int np = gt->wnp[w];
  unsigned long long* row = gt->rows + (size_t)(np % kGtRows) * kGtCols * kGtWords;
  if (t < NV) {  // lane k publishes word k
    double mine = v[0];
#pragma unroll
    for (int k = 1; k < NV; ++k) mine = (t == k) ? v[k] : mine;
    const unsigned long long b = (mine != mine) ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(mine);
    __hip_atomic_store(row + (size_t)blockIdx.x * kGtWords + t, b + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  double g[NV];
#pragma unroll
  for (int k = 0; k < NV; ++k) g[k] = ((maxmask >> k) & 1u) ? -INFINITY : 0.0;
  if (t < W) {
    const unsigned long long* theirs = row + (size_t)(gt->first + t) * kGtWords;
    unsigned long long word[NV];
    unsigned int spins = 0;
    for (;;) {  // (every workgroup of the team is resident and stores these words once per reduction; bounded all the same)
      bool all = true;
#pragma unroll
      for (int k = 0; k < NV; ++k) {
        word[k] = __hip_atomic_load(theirs + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        all = all && (word[k] != 0ull);
      }
      if (all) break;
      if (spx_wait_expired(spins, gt->hdr)) break;  // (the totals come out as garbage: the caller stores NaN, spx_poisoned)
      __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int k = 0; k < NV; ++k) g[k] = __longlong_as_double((long long)(word[k] - 1ull));
  }
  grid_team_block_combine<NV>(gt, g, maxmask, (calls + 1) & 1);
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = g[k];
  // this workgroup's words of the previous row: nobody reads them any more
  if (np > 0 && t < kGtWords)
    __hip_atomic_store(gt->rows + (size_t)((np - 1) % kGtRows) * kGtCols * kGtWords + (size_t)blockIdx.x * kGtWords + t, 0ull,
                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if ((t & 63) == 0) { gt->wnp[w] = np + 1; gt->wcalls[w] = calls + 2; }
}

template <int TEAM>
__device__ __forceinline__ double team_sum(double v, double* lds /* 8 doubles per block, TEAM == 256 only */) {
  if constexpr (TEAM == kTeamGrid) {
    double a[1] = {v};
    grid_team_reduce<1>(reinterpret_cast<GridTeam*>(lds), a, 0u);
    return a[0];
  } else if constexpr (TEAM == 256) {
    v = lanes_sum<64>(v);
    const int w = threadIdx.x >> 6;
    __syncthreads();  // previous use of lds finished
    if ((threadIdx.x & 63) == 0) lds[w] = v;
    __syncthreads();
    return (lds[0] + lds[1]) + (lds[2] + lds[3]);
  } else {
    return lanes_sum<TEAM>(v);
  }
}
template <int TEAM>
__device__ __forceinline__ void team_sum2(double& a, double& b, double* lds) {
  if constexpr (TEAM == kTeamGrid) {
    double ab[2] = {a, b};
    grid_team_reduce<2>(reinterpret_cast<GridTeam*>(lds), ab, 0u);
    a = ab[0];
    b = ab[1];
  } else if constexpr (TEAM == 256) {
    a = lanes_sum<64>(a);
    b = lanes_sum<64>(b);
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { lds[w] = a; lds[4 + w] = b; }
    __syncthreads();
    a = (lds[0] + lds[1]) + (lds[2] + lds[3]);
    b = (lds[4] + lds[5]) + (lds[6] + lds[7]);
  } else {
    a = lanes_sum<TEAM>(a);
    b = lanes_sum<TEAM>(b);
  }
}

// max over the team of a NaN-free value
template <int TEAM>
__device__ __forceinline__ double team_max(double v, double* lds) {
  if constexpr (TEAM == kTeamGrid) {
    double a[1] = {v};
    grid_team_reduce<1>(reinterpret_cast<GridTeam*>(lds), a, 1u);
    return a[0];
  }
  if constexpr (TEAM >= 2) v = fmax(v, dpp_f64<0xB1>(v));
  if constexpr (TEAM >= 4) v = fmax(v, dpp_f64<0x4E>(v));
  if constexpr (TEAM >= 8) v = fmax(v, dpp_f64<0x141>(v));
  if constexpr (TEAM >= 16) v = fmax(v, dpp_f64<0x140>(v));
  if constexpr (TEAM >= 32) v = fmax(v, __shfl_xor(v, 16, 64));
  if constexpr (TEAM >= 64) v = fmax(v, __shfl_xor(v, 32, 64));
  if constexpr (TEAM == 256) {
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds[w] = v;
    __syncthreads();
    v = fmax(fmax(lds[0], lds[1]), fmax(lds[2], lds[3]));
  }
  return v;
}

// softthres(x, a) = sign(x) * max(0, |x| - a)          src/shiftedGroupNormL2Binf.jl:82
__device__ __forceinline__ double softthres(double x, double a) { return jl_sign(x) * jl_max(0.0, fabs(x) - a); }

// 1/x to a few ulp: hardware seed + 2 Newton steps.  Only used inside the self-correcting Newton iteration.
__device__ __forceinline__ double fast_rcp(double x) {
  const double y0 = __builtin_amdgcn_rcp(x);  // +-inf for +-0, +-0 for +-inf
  double y = __builtin_fma(__builtin_fma(-x, y0, 1.0), y0, y0);
  y = __builtin_fma(__builtin_fma(-x, y, 1.0), y, y);
  return (y == y) ? y : y0;  // the Newton steps turn 1/0 and 1/inf into NaN: keep the seed there
}

// ---------------------------------------------------------------------------------------------
// Per-group element access.  Two providers with the same interface:
//   RegGroup<EPL>  : S = (q + xk) + sj and X = xk of this lane's EPL elements live in registers
//   MemGroup<TEAM> : elements are re-read from global memory (L1/L2 resident for moderate groups)
// for_each(f) calls f(S_i, X_i) for every element this lane owns.
// ---------------------------------------------------------------------------------------------
// PADDED (round 4): the groups do not fill the tile -- the slots from `live` on (wave-uniform: the same for every lane and every
// group of the launch) hold zeros in every lane; the element loops skip them behind a scalar branch instead of adding zeros
// (a group of 100 on the 8 x 16 tile: 7 of 8 pair-slices live; a group of 66: 5 of 8 -- the Binf form is VALU-bound on such
// sizes).  The full-tile instantiations (PADDED = false) are unchanged.
template <int EPL, bool PADDED = false>
struct RegGroup {
  static constexpr bool kReg = true;
  static constexpr int kEpl = EPL;
  double S[EPL], X[EPL], XS[EPL];  // XS = xk + sj (subtracted at the end)
  int live = EPL;                  // (PADDED) slots < live may hold elements; even
  template <class F>
  __device__ __forceinline__ void for_each(F&& f) const {
    if constexpr (PADDED) {
#pragma unroll
      for (int k = 0; k < EPL; k += 2) {
        if (k < live) { f(S[k], X[k]); f(S[k + 1], X[k + 1]); }
      }
    } else {
#pragma unroll
      for (int k = 0; k < EPL; ++k) f(S[k], X[k]);
    }
  }
};

template <int TEAM>
struct MemGroup {
  static constexpr bool kReg = false;
  static constexpr int kEpl = 1;
  const double* q;
  const double* xk;
  const double* sj;
  int64_t lo, hi;
  int lane;  // index inside the team
  template <class F>
  __device__ __forceinline__ void for_each(F&& f) const {
    for (int64_t i = lo + lane; i < hi; i += TEAM) {
      double x = xk[i];
      f((q[i] + x) + sj[i], x);
    }
  }
  // y[i] = f(S, X) - (xk + sj) for every element this lane owns (q[i] is read before y[i] is written: y may alias q)
  template <class F>
  __device__ __forceinline__ void store(double* y, F&& f) const {
    for (int64_t i = lo + lane; i < hi; i += TEAM) {
      const double x = xk[i], s = sj[i];
      const double S = (q[i] + x) + s;
      y[i] = f(S, x) - (x + s);
    }
  }
};

// ---------------------------------------------------------------------------------------------
// Binf root find.  src/shiftedGroupNormL2Binf.jl:85-108
//   froot(n) = n - || sigma * softthres(S/sigma - step X, Delta step) - S ||,  step = n / (sigma (n - sl))
//
// Structure used here.  With u = n - sl > 0 and tau = u / n = 1 / (sigma step) in (0, 1) an element is
// thresholded to zero iff |tau S_i - X_i| <= Delta (its term is then -S_i); otherwise the term equals
// -(n/u) b_i,  b_i = X_i + Delta sgn(tau S_i - X_i).  So
//   froot(n) = (n/u) * psi(u),   psi(u) = u - phi(u),   phi(u) = sqrt(B(u) + tau^2 A(u)),
//   A = sum_{inactive} S_i^2,  B = sum_{active} b_i^2,
// and the prox itself is  w_i = S_i - sigma softthres(...) = (n/u) b_i (active) or S_i (inactive)  (:111).
// Every |term| is non-increasing in n, hence froot is strictly increasing on n > sl: the root inside the
// reference's bracket [lmin, lmax] is unique, and a bracketing iteration of any kind lands on the root the
// reference's bisection (Roots.fzero) converges to.  We solve psi(u) = 0 in u (no cancellation in n - sl, no
// pole) by a bracket-safeguarded Newton iteration and keep the root AS u: at the root ||w|| = n, so the last step
// alpha = 1 - sl/||w|| (:83) is u/n = tau and  alpha w_i = b_i (active) or tau S_i (inactive)  -- a closed form without
// the two cancellations (n - sl, 1 - sl/||w||) the reference's Float64 evaluation goes through.  Where u << n those
// cancellations cost the REFERENCE up to ~1e-9 of its own result (the granularity of the double n next to the pole of
// step(n)); round 1 reproduced that error by "polishing" n to the double Roots' bisection returns and then sat, like the
// reference, 1e-9 away from the exact value -- and further away than the reference in half of such groups.  Adjudicated in
// round 2 with the binary128 arbiter (tests/arbiter.py): the closed form is within ~1e-15 of the exact value of the
// reference's formula everywhere, so it is never the worse side.  All loops are bounded.
// The reference's literal expression is kept for the degenerate bracket only (binf_froot_literal).
// ---------------------------------------------------------------------------------------------
#define SPX_BINF_NEWTON_MAXIT 60
#ifndef SPX_BINF_CONFIRM
#define SPX_BINF_CONFIRM 1  // A/B switch of the cheap active-set confirmation (binf_same_active_set)
#endif
#ifndef SPX_BINF_LAZY
#define SPX_BINF_LAZY 1  // A/B switch (round 4): lmax (the zlmax pass), froot(lmin) inside the trust region and the start -- see binf_root
#endif
#ifndef SPX_BINF_START_MAXTEAM
#define SPX_BINF_START_MAXTEAM 64  // A/B switch (round 4): largest team (lanes per group) whose iteration starts at ||S|| - sl instead of the bound
#endif
#ifndef SPX_BINF_POLY
#define SPX_BINF_POLY 1  // A/B switch (round 4): approach to a piece's root on the quartic in t instead of Newton in v
#endif
// (tried: a first piece solve without its final accurate Newton step -- slower, 0.99 vs 0.85 ms: the next pass then
//  starts from a point that is not a piece root and an extra pass follows)

// literal froot(n)  (:87-93)
template <int TEAM, class G>
__device__ __forceinline__ double binf_froot_literal(const G& grp, double n, double sigma, double sl, double delta,
                                                     double* lds) {
  const double step = n / (sigma * (n - sl));
  const double thr = delta * step;
  double sw = 0.0;
  grp.for_each([&](double S, double X) {
    double w = sigma * softthres(S / sigma - step * X, thr) - S;
    sw += w * w;
  });
  return n - sqrt(team_sum<TEAM>(sw, lds));
}

// Delta with the sign of z
__device__ __forceinline__ double signed_delta(double delta, double z) {
  return __hiloint2double((__double2hiint(delta) & 0x7fffffff) | (__double2hiint(z) & 0x80000000), __double2loint(delta));
}

// v if keep, else (essentially) zero: clears the high dword only -> one v_cndmask instead of two; the leftover
// low dword is a denormal below 2^-1042, invisible in the sums below
__device__ __forceinline__ double keep_if(double v, bool keep) {
  return __hiloint2double(keep ? __double2hiint(v) : 0, __double2loint(v));
}

// sums A (inactive S^2) and B (active b^2) at a given tau
template <int TEAM, class G>
__device__ __forceinline__ void binf_ab(const G& grp, double tau, double delta, double* lds, double& sa, double& sb) {
  sa = 0.0;
  sb = 0.0;
  grp.for_each([&](double S, double X) {
    const double z = __builtin_fma(tau, S, -X);
    const bool act = fabs(z) > delta;
    const double b = X + signed_delta(delta, z);
    // masked operand, then one fma per sum (8 VALU instructions per element instead of 10 with mul + mask + add: the
    // kernel is VALU-bound and this loop is half of it); the leftover low dword of a masked operand squares to zero
    const double Sm = keep_if(S, !act), bm = keep_if(b, act);
    sa = __builtin_fma(Sm, Sm, sa);
    sb = __builtin_fma(bm, bm, sb);
  });
  team_sum2<TEAM>(sa, sb, lds);
  // keep_if leaves sub-2^-1042 residues of the masked-out terms in the sums: an empty set must sum to exactly 0 (the
  // piece confirmation compares sums bit for bit, and sb == 0 identifies the all-inactive piece)
  sa = (sa < 1e-300) ? 0.0 : sa;
  sb = (sb < 1e-300) ? 0.0 : sb;
}

// Register-resident groups: true in the lanes of a team whose active set {i : |tau S_i - X_i| > Delta} is the same at
// tau_a and tau_b (2 fma + 2 compares per element; nothing is stored between passes -- keeping the previous pass's lane
// masks alive costs more registers than the kernel has, measured: 0.83 -> 1.01 ms).
template <int TEAM, class G>
__device__ __forceinline__ bool binf_same_active_set(const G& grp, double tau_a, double tau_b, double delta) {
  unsigned long long diff = 0ull;
#pragma unroll
  for (int k = 0; k < G::kEpl; ++k) {
    if (k < grp.live) {  // (wave-uniform; always true for the full-tile instantiations)
      const double za = __builtin_fma(tau_a, grp.S[k], -grp.X[k]);
      const double zb = __builtin_fma(tau_b, grp.S[k], -grp.X[k]);
      diff |= __ballot((fabs(za) > delta) != (fabs(zb) > delta));
    }
  }
  const int lane = threadIdx.x & 63;
  const unsigned long long mine = (TEAM >= 64) ? ~0ull : (((1ull << (TEAM & 63)) - 1ull) << ((lane / TEAM) * TEAM));
  return (diff & mine) == 0ull;
}

__device__ __forceinline__ double sqrt_pos(double v) {  // sqrt for v >= 0: rsq seed + two corrections
  const double r = __builtin_amdgcn_rsq(v);
  double s = v * r;
  s = __builtin_fma(0.5 * r, __builtin_fma(-s, s, v), s);
  s = __builtin_fma(0.5 * r, __builtin_fma(-s, s, v), s);
  return (v > 0.0) ? s : 0.0;
}

// psi(u) and psi'(u)
template <int TEAM, class G>
__device__ __forceinline__ void binf_psi(const G& grp, double u, double sl, double delta, double* lds, double& psi,
                                         double& dpsi) {
  const double n = sl + u;
  const double rn = fast_rcp(n);
  const double tau = u * rn;
  double sa, sb;
  binf_ab<TEAM>(grp, tau, delta, lds, sa, sb);
  const double phi = sqrt_pos(__builtin_fma(tau * tau, sa, sb));
  psi = u - phi;
  const double dtau = sl * rn * rn;
  dpsi = 1.0 - ((phi > 0.0) ? (sa * tau * dtau * fast_rcp(phi)) : 0.0);
}
// Roots.__middle for positive doubles: the double whose bit pattern is the mean of the two bit patterns
__device__ __forceinline__ double bit_middle(double a, double b) {
  const unsigned long long m = ((unsigned long long)__double_as_longlong(fabs(a)) +
                                (unsigned long long)__double_as_longlong(fabs(b))) >> 1;
  return jl_sign(a + b) * __longlong_as_double((long long)m);
}

// Roots.fzero(froot, a, b) literally: sorted bracket, bit-pattern midpoint, sign bisection to exhaustion,
// the end with the smaller |f| (a NaN value moves the lower end, as `sign(fa) * sign(fc) < 0` is false).
template <int TEAM, class G>
__device__ __forceinline__ double binf_bisect(const G& grp, double a, double fa, double b, double fb, double sigma,
                                              double sl, double delta, double* lds) {
  if (a > b) { double t = a; a = b; b = t; t = fa; fa = fb; fb = t; }
  if (fa == 0.0) return a;
  if (fb == 0.0) return b;
  for (int it = 0; it < 130; ++it) {
    const double m = bit_middle(a, b);
    if (!(a < m && m < b)) break;
    const double fmid = binf_froot_literal<TEAM>(grp, m, sigma, sl, delta, lds);
    if (jl_sign(fa) * jl_sign(fmid) < 0) { b = m; fb = fmid; }
    else { a = m; fa = fmid; }
  }
  return (fabs(fa) < fabs(fb)) ? a : b;
}

// The reference's own evaluation, literally: froot at both ends (:95,:101), `fl * fm > 0` -> zeros (:102), else
// Roots.fzero (:105).  Used for degenerate brackets, exact zeros at an end and NaNs, where the A/B form of the
// fast path (which assumes n > sl) does not describe what the reference computes.
template <int TEAM, class G>
__device__ __forceinline__ bool binf_literal_root(const G& grp, double lam, double sigma, double delta, double* lds,
                                                  double& root) {
  const double eps = 2.220446049250313e-16;
  const double sl = lam * sigma;
  const double lmin = sl * (1 + eps);                                     // :94
  const double fl = binf_froot_literal<TEAM>(grp, lmin, sigma, sl, delta, lds);  // :95
  const double ansatz = lmin + 1.0;                                       // :97
  const double step = ansatz / (sigma * (ansatz - sl));                   // :98
  double sz = 0.0, sS = 0.0, sX = 0.0, mX = 0.0, gap = INFINITY;
  grp.for_each([&](double S, double X) {
    const double z = softthres(S / sigma - step * X, delta * step);       // :99
    sz += z * z;
    sS += S * S;
    sX += X * X;
    mX = fmax(mX, fabs(X));
    gap = fmin(gap, fabs(fabs(X) - delta));
  });
  team_sum2<TEAM>(sz, sS, lds);
  sX = team_sum<TEAM>(sX, lds);
  const double lmax = sqrt(sS) + sigma * (sqrt(sz) + 1.0 * lam * sqrt(sX));  // :100
  // Reversed bracket with entries outside the trust region: the piece iteration of binf_literal_reg (derivation and
  // guards there), for the element providers of the general / LDS / gather kernels.  `false` = the reference writes zeros.
  {
    mX = team_max<TEAM>(mX, lds);
    gap = -team_max<TEAM>(-gap, lds);
    if (lmax < lmin * (1.0 - 1e-9) && lmin > sl && (sS + sX < INFINITY) && sqrt(sS) <= 1e6 * delta && sl <= 1e6 * delta &&
        mX - delta >= 1e-6 * sl && gap > 1e-9 * delta) {
      auto rsq_at = [&](double n) -> double {  // R(n)^2
        const double a = n / (sigma * (sl - n));
        double acc = 0.0;
        grp.for_each([&](double S, double X) {
          const double z = __builtin_fma(a, X, S / sigma);
          const double b = (z == 0.0) ? X : X + signed_delta(delta, z);
          acc += b * b;
        });
        return team_sum<TEAM>(acc, lds);
      };
      double r2 = rsq_at(lmax);
      double R = sqrt(r2);
      const double g0 = (sl - lmax) - R;
      bool decided = g0 < -1e-9 * sl;
      if (!decided && g0 > 1e-9 * sl) {
        for (int k = 0; k < 8 && !decided; ++k) {
          if (!(sl - R > 0.0)) break;
          const double r2n = rsq_at(sl - R);
          decided = (r2n == r2);
          r2 = r2n;
          R = sqrt(r2);
        }
      }
      if (decided) return false;
    }
  }
  const double fm = binf_froot_literal<TEAM>(grp, lmax, sigma, sl, delta, lds);  // :101
  if (fl * fm > 0) return false;                                          // :102
  root = binf_bisect<TEAM>(grp, lmin, fl, lmax, fm, sigma, sl, delta, lds);  // :105
  return true;
}

// The same for a register-resident group (the deferred list of k_group_reg, run by its LIT instantiation): the ~60
// dependent passes of fzero then cost arithmetic only -- with the group re-read from memory for every pass (k_group_mem)
// a list that holds a large share of the groups is bound by L2 misses, 14 ms for 2.7e5 groups of 128.  S/sigma (:89) is
// loop-invariant and formed once.  out[k] = y before the final "- (xk + sj)".
template <int LPG, int EPL, bool PADDED>
__device__ __forceinline__ void binf_literal_reg(const RegGroup<EPL, PADDED>& grp, double lam, double sigma, double delta,
                                                 double* out) {
  const double eps = 2.220446049250313e-16;
  const double sl = lam * sigma;
  double Sd[EPL];
#pragma unroll
  for (int k = 0; k < EPL; ++k) Sd[k] = grp.S[k] / sigma;
  auto froot = [&](double n) -> double {  // :87-93
    const double step = n / (sigma * (n - sl));
    const double thr = delta * step;
    double sw = 0.0;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const double w = sigma * softthres(Sd[k] - step * grp.X[k], thr) - grp.S[k];
      sw += w * w;
    }
    return n - sqrt(lanes_sum<LPG>(sw));
  };
  const double lmin = sl * (1 + eps);  // :94
  double fa = froot(lmin);             // :95
  const double ansatz = lmin + 1.0;    // :97
  const double stepa = ansatz / (sigma * (ansatz - sl));  // :98
  double sz = 0.0, sS = 0.0, sX = 0.0;
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    const double z = softthres(Sd[k] - stepa * grp.X[k], delta * stepa);  // :99
    sz += z * z;
    sS += grp.S[k] * grp.S[k];
    sX += grp.X[k] * grp.X[k];
  }
  sz = lanes_sum<LPG>(sz);
  sS = lanes_sum<LPG>(sS);
  sX = lanes_sum<LPG>(sX);
  const double lmax = sqrt(sS) + sigma * (sqrt(sz) + 1.0 * lam * sqrt(sX));  // :100
  // Reversed bracket (lmax < lmin) with entries OUTSIDE the trust region (|X_i| > Delta: X is the iterate, Delta the radius
  // of the step -- the normal state of a small group late in a run), decided without the ~60 literal passes of fzero.
  // froot(lmin) is hugely negative (an active entry next to the pole).  Below the pole step < 0, softthres only adds, and
  //   froot(n) = n (1 - R(n) / (sl - n)),  R^2 = sum (X_i + sgn(z_i) Delta)^2,  z_i = S_i/sigma + |step| X_i;
  // a z_i changes sign only from sgn(S_i) to sgn(X_i) as n grows, raising its term from (|X_i| - Delta)^2 to
  // (|X_i| + Delta)^2: R is non-decreasing, sl - n - R(n) strictly decreasing, the bracket holds ONE sign change and any
  // bracketing iteration finds the one fzero finds:
  //   * sl - lmax - R(lmax) < 0: froot(lmax) < 0 as well -> :102-103, zeros;
  //   * else iterate n <- sl - R(n) (the root of the current piece; the iterates close in on the sign change from both
  //     sides): when R comes back unchanged the sign change is a genuine root n* = sl - R inside a piece, where
  //     ||w|| = n* < sl and :111 is l2prox(w, sl) = 0 -- zeros again;
  //   * no repeat within a few steps: the sign change sits on a jump of R, fzero returns one side of it and the result is
  //     not zero in general -> the literal evaluation below (0.7 % of such groups).
  // Guards: bracket reversed by 1e-9, no |X_i| within 1e-9 Delta of Delta, ||S|| and sl <= 1e6 Delta (froot(lmin) safely
  // negative), max|X| - Delta >= 1e-6 sl (R is not lost in the rounding of ||w|| against sl).  Checked on CPU against the
  // literal restatement: 5.4e5 such groups, 99.3 % decided, none wrongly (tools/fuzz_binf_reversed.py pins it on the GPU).
  {
    double mX = 0.0, gap = INFINITY;
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      mX = fmax(mX, fabs(grp.X[k]));
      gap = fmin(gap, fabs(fabs(grp.X[k]) - delta));
    }
    mX = team_max<LPG>(mX, nullptr);
    gap = -team_max<LPG>(-gap, nullptr);
    const double nS = sqrt(sS);
    if (lmax < lmin * (1.0 - 1e-9) && lmin > sl && (sS + sX < INFINITY) && nS <= 1e6 * delta && sl <= 1e6 * delta &&
        mX - delta >= 1e-6 * sl && gap > 1e-9 * delta) {
      auto rsq_at = [&](double n) -> double {  // R(n)^2
        const double a = n / (sigma * (sl - n));
        double acc = 0.0;
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
          const double z = __builtin_fma(a, grp.X[k], Sd[k]);
          const double b = (z == 0.0) ? grp.X[k] : grp.X[k] + signed_delta(delta, z);
          acc += b * b;
        }
        return lanes_sum<LPG>(acc);
      };
      double r2 = rsq_at(lmax);
      double R = sqrt(r2);
      const double g0 = (sl - lmax) - R;
      bool decided = g0 < -1e-9 * sl;
      if (!decided && g0 > 1e-9 * sl) {
        for (int k = 0; k < 8 && !decided; ++k) {
          if (!(sl - R > 0.0)) break;  // (R can exceed sl after a jump: leave it to the literal evaluation)
          const double r2n = rsq_at(sl - R);
          decided = (r2n == r2);
          r2 = r2n;
          R = sqrt(r2);
        }
      }
      if (decided) {
#pragma unroll
        for (int k = 0; k < EPL; ++k) out[k] = 0.0;
        return;
      }
    }
  }
  double fb = froot(lmax);                                                   // :101
  bool zeros = fa * fb > 0;                                                  // :102
  double root = lmin;
  if (!zeros) {  // Roots.fzero(froot, lmin, lmax), as binf_bisect
    double a = lmin, b = lmax;
    if (a > b) { double t = a; a = b; b = t; t = fa; fa = fb; fb = t; }
    if (fa == 0.0) root = a;
    else if (fb == 0.0) root = b;
    else {
      for (int it = 0; it < 130; ++it) {
        const double m = bit_middle(a, b);
        if (!(a < m && m < b)) break;
        const double fmid = froot(m);
        if (jl_sign(fa) * jl_sign(fmid) < 0) { b = m; fb = fmid; }
        else { a = m; fa = fmid; }
      }
      root = (fabs(fa) < fabs(fb)) ? a : b;
    }
    zeros = (root - sl) == 0.0;  // :107
  }
  if (zeros) {
#pragma unroll
    for (int k = 0; k < EPL; ++k) out[k] = 0.0;
    return;
  }
  const double step = root / (sigma * (root - sl));  // :106
  double sw = 0.0;
#pragma unroll
  for (int k = 0; k < EPL; ++k) {
    out[k] = grp.S[k] - sigma * softthres(Sd[k] - step * grp.X[k], delta * step);  // :111
    sw += out[k] * out[k];
  }
  const double nw = sqrt(lanes_sum<LPG>(sw));
  const double alpha = jl_max(0.0, 1 - sl / nw);  // :83
#pragma unroll
  for (int k = 0; k < EPL; ++k) out[k] = alpha * out[k];
}

enum { BINF_ZERO = 0, BINF_ROOT = 1, BINF_LITERAL = 2 };
// BINF_ROOT: the root n = sl + u, returned as u (> 0) in `root_u`;  BINF_ZERO: fl * fm > 0, the reference writes zeros
// (:102-103);  BINF_LITERAL: degenerate bracket / exact zero at an end / NaN -> the caller must run binf_literal_root.
template <int TEAM, class G>
__device__ __forceinline__ int binf_root(const G& grp, double lam, double sigma, double delta, double* lds,
                                         double& root_u, bool pole_lit) {
  const double eps = 2.220446049250313e-16;
  // (the team-of-workgroups kernels sit at their 128-VGPR cap and have a sample-predicted fast path of their own: as they were)
  constexpr bool kLazy = SPX_BINF_LAZY && TEAM != kTeamGrid;
  const double sl = lam * sigma;          // :85
  const double lmin = sl * (1 + eps);     // :94
  // lmax, zlmax and froot(lmin) only steer the bracket (their exact rounding never reaches y): the wave-uniform
  // divisions and square roots below use the few-ulp fast forms
  // first pass: ||S||, ||X|| and max |X_i|
  const double ul = lmin - sl;
  const double taul = ul * fast_rcp(lmin);
  double sS = 0.0, sX = 0.0, mX = 0.0;
  grp.for_each([&](double S, double X) {
    sS = __builtin_fma(S, S, sS);
    sX = __builtin_fma(X, X, sX);
    mX = fmax(mX, fabs(X));
  });
  if constexpr (TEAM == kTeamGrid) {  // (a team of workgroups: one exchange for the three values instead of two)
    double v3[3] = {sS, sX, mX};
    grid_team_reduce<3>(reinterpret_cast<GridTeam*>(lds), v3, 4u);
    sS = v3[0];
    sX = v3[1];
    mX = v3[2];
  } else {
    team_sum2<TEAM>(sS, sX, lds);
    mX = team_max<TEAM>(mX, lds);
  }
  const double nS = sqrt_pos(sS), nX = sqrt_pos(sX);
  // X == 0 in the whole group and sigma lambda > ||S||: the group of a sparse iterate that stays zero -- the bulk of the
  // groups in a group-lasso run, so it must not take the literal path its (usually degenerate: lmax = ||S|| + sigma zlmax
  // < lmin) bracket would otherwise send it to.  The reference returns zeros whichever way it goes:
  //  * lmax >= lmin: on n >= lmin every |w_i| <= |S_i|, so froot(n) >= n - ||S|| > 0 at both ends -> :102-103;
  //  * lmax < lmin: froot(lmin) > 0 as before; froot(lmax) = lmax (1 - Delta sqrt(#S_i != 0) / (sigma lambda - lmax)) (step < 0:
  //    softthres only ADDS |thr|); if positive -> :102-103, if negative the only sign change in (lmax, lmin] is the pole
  //    n = sigma lambda (froot -> -inf below it, n - ||w|| > 0 above it), fzero ends on a double just above the pole, and
  //    there :111 is l2prox(v, sigma lambda) with ||v|| <= ||S|| < sigma lambda, i.e. zeros again.
  // (1e-9 away from the tie sigma lambda = ||S||, which stays with the general code below.)
  if (mX == 0.0 && sl < INFINITY && sl * (1.0 - 1e-9) > nS) return BINF_ZERO;
  const double ub = sqrt_pos(sS + sX) * (1.0 + 8 * eps);  // a-priori bound on the root in u (see below)
  // lmax = ||S|| + sigma (zlmax + lambda ||X||) (:100) needs one more pass for zlmax (:99).  zlmax >= 0, so
  // lmax >= lmax_lb := ||S|| + sigma lambda ||X||.  If already lmax_lb clears both lmin (bracket not degenerate) and
  // sl + ub (the iteration starts from the bound ub, not from lmax) the exact lmax is never used: skip the pass.
  const double lmax_lb = nS + sigma * (lam * nX);
  // Round 4: lmax on demand.  With several groups in a wavefront -- 64 of them on the one-lane tiles of groups of <= 8
  // elements -- a branch that 5 % of the groups take is executed by every wavefront: the zlmax pass below and the A/B sums
  // at lmin were ~15 % of the vector instructions of the kernel on groups of 8, for 6 % / 5 % of the groups.  So, when the
  // lower bound lmax_lb alone says that the bracket is regular (lmin < lmax_lb <= lmax), round 0 looks for the root in
  // (lmin, sl + ub) as the `from_bound` iteration does and accepts it if it lies below lmax_lb with a margin: froot is
  // increasing, so froot(lmax) >= froot(lmax_lb) > 0 then, :102 does not fire and the reference's bisection converges to
  // this root; likewise froot(lmin) > 0 means froot(lmax) > 0 and :102 fires whatever lmax is exactly.  A root that is
  // not safely below lmax_lb has the zlmax pass run then, and is accepted against the exact lmax or handed to the literal
  // evaluation (lmax is the reference's upper BOUND on the root: next to never).
  double lmax;
  bool lmax_is_normS = false;  // lmax == ||S|| exactly in the reference: zlmax == 0 and lambda ||X|| == 0
  bool lazy = false;
  if (lmax_lb > lmin * (1.0 + 8 * eps) && (lmax_lb - sl) > ub * (1.0 + 8 * eps)) {
    lmax = sl + ub * (1.0 + 8 * eps);  // any point >= the root serves as the upper end from here on
  } else if (kLazy && lmax_lb > lmin * (1.0 + 8 * eps) && ub > ul * (1.0 + 8 * eps) && (sS + sX < INFINITY)) {
    lazy = true;
    lmax = sl + ub * (1.0 + 8 * eps);
  } else {
    const double ansatz = lmin + 1.0;                                      // :97 (epsilon = 1)
    const double rsig = fast_rcp(sigma);
    const double stepa = ansatz * rsig * fast_rcp(ansatz - sl);            // :98
    const double thra = delta * stepa;
    double sz = 0.0;
    grp.for_each([&](double S, double X) {
      const double za = fabs(__builtin_fma(-stepa, X, S * rsig)) - thra;  // |softthres| = max(0, |.| - thr)  (:99)
      sz += (za > 0.0) ? za * za : 0.0;
    });
    sz = team_sum<TEAM>(sz, lds);
    lmax = nS + sigma * (sqrt_pos(sz) + 1.0 * lam * nX);                   // :100 (|(eps-1)/eps + 1| = 1)
    lmax_is_normS = (sz == 0.0) && (lam * nX == 0.0);
    // Reversed bracket (lmax < lmin: ||S|| + sigma (zlmax + lambda ||X||) below sigma lambda -- a small group about to be
    // zeroed) with every |X_i| inside the trust region: the reference returns zeros, decided here without its literal
    // bisection (whole data sets fell into that path: 16 ms instead of 0.7 ms at 1e6 x 128, tools/sweep_params.py).
    //  * froot(lmin) = lmin - ||S|| > 0: at lmin (1-2 ulp above the pole) step ~ 1/(sigma eps), |S_i/sigma - step X_i| <=
    //    step Delta for every i (|X_i| < Delta(1 - 1e-9), ||S|| <= 1e6 Delta), all entries thresholded; ||S|| <= lmax < lmin.
    //  * below the pole step < 0 and softthres only adds: froot(n) = n (1 - R(n) / (sigma lambda - n)) with
    //    R^2 = sum (X_i + sgn(z_i) Delta)^2, z_i = S_i/sigma + |step| X_i.  A z_i changes sign only from sgn(S_i) to
    //    sgn(X_i) as n grows, which raises its term from (|X_i| - Delta)^2 to (|X_i| + Delta)^2: R is non-decreasing, so
    //    sigma lambda - n - R(n) is decreasing -- if froot(lmax) > 0 the signs agree (:102-103: zeros); if froot(lmax) < 0
    //    froot stays negative up to the pole, the only sign change of the bracket is the pole itself, fzero ends on a
    //    double just above it, everything is thresholded there and :111 is l2prox(S, sigma lambda) = 0 (||S|| < sigma lambda).
    // (tests/test_gpu_parity.py::test_group_binf_small_groups_being_zeroed and tools/fuzz_binf_reversed.py pin this regime.)
    const bool reversed = lmax < lmin * (1.0 - 1e-9) && ul > 0.0 && nS <= 1e6 * delta && (sS + sX < INFINITY);
    if (reversed && mX < delta * (1.0 - 1e-9)) return BINF_ZERO;
    // (Reversed brackets with entries OUTSIDE the trust region go to the deferred list: binf_literal_reg decides most of
    //  them without the literal bisection, see there -- the code would cost this kernel its register budget.)
  }
  // fl = froot(lmin) (:95): only its sign is used.  lmin sits eps above the pole of step(n): tau(lmin) ~ eps, the element
  // with the largest |X_i| has |tau S_i - X_i| >= max|X| - tau ||S||; if that exceeds Delta by a relative 1e-9 the
  // element is active with |b_i| >= 1e-9 Delta, so froot(lmin) <= lmin - (lmin/ul) 1e-9 Delta < 0 as soon as
  // ul = eps sl < 1e-9 Delta -- decided without the A/B sums at lmin.  Otherwise (all |X_i| <= Delta, huge lambda) they
  // are formed as in every other evaluation.
  double fl;
  if ((sS + sX < INFINITY) && (mX - taul * nS > delta * (1.0 + 1e-9)) && (ul < 1e-9 * delta)) {
    fl = -1.0;
  } else if (kLazy && (sS + sX < INFINITY) && (mX + taul * nS < delta * (1.0 - 1e-9))) {
    // (round 4) every entry safely inside the trust region at lmin: |tau S_i - X_i| <= max|X| + tau ||S|| < Delta, nothing is
    // active, A = ||S||^2 and B = 0: froot(lmin) = lmin - (lmin/ul) tau(lmin) ||S|| = lmin - ||S|| without a pass (46 % of
    // random groups of 2, 5 % of groups of 8 -- and so nearly every wavefront of 64 such groups took the branch below)
    fl = lmin - nS;
  } else {
    // No entry is safely active at lmin.  The A/B form gives the exact value of froot(lmin); the reference's own
    // evaluation agrees with it unless some |X_i| equals Delta to the last bits: lmin sits one ulp above the pole of
    // step(n), S/sigma - step X is formed at magnitude ~1e16, and such an entry comes out of softthres as 0, 1 or 2 by
    // rounding alone -- which then decides the sign of froot(lmin), the bracket, and a bisection that ends on a spurious
    // root next to the pole.  Those groups are handed to the literal evaluation, which reproduces exactly that
    // (found by tools/fuzz_binf_scenarios.py: |X_i| = Delta gave the exact-arithmetic answer, not the reference's, in
    // ~1 % of 2-element groups).
    double gap = INFINITY;  // min over the group of | |X_i| - Delta |
    grp.for_each([&](double, double X) { gap = fmin(gap, fabs(fabs(X) - delta)); });
    gap = -team_max<TEAM>(-gap, lds);
    if (gap <= 1e-9 * delta) return BINF_LITERAL;
    double sal, sbl;
    binf_ab<TEAM>(grp, taul, delta, lds, sal, sbl);
    fl = lmin - (lmin * fast_rcp(ul)) * sqrt_pos(__builtin_fma(taul * taul, sal, sbl));
  }
  if (!(lmin < lmax) || !(ul > 0.0) || !(fl == fl)) return BINF_LITERAL;  // degenerate bracket, sl == 0, NaN
  // regular bracket sl < lmin < lmax:  fm = froot(lmax) has the sign of psi(uhi)   (:101).
  // |b_i| <= max(|S_i|, |X_i|) gives phi(u) <= ub := sqrt(||S||^2 + ||X||^2) for every u, so the root is <= ub:
  // if uhi >= ub then psi(uhi) >= 0 is known without evaluating it and the iteration starts from ub instead.
  //
  // Iteration: psi is piecewise smooth -- on a fixed active set it is u - sqrt(B + A tau(u)^2) with constant
  // A, B.  Each pass over the group yields (A, B) of the current u; the root of THAT piece is then found by a
  // scalar Newton iteration (no element work), and the next pass checks it: if the sums come back bit-identical
  // the active set did not change and u is the root; otherwise continue from the new piece.  2-3 passes on
  // the BASELINE data instead of 5-6 for Newton on psi itself; bracket-safeguarded, bounded.
  double ulo = ul, uhi = lmax - sl;
  const bool from_bound = (uhi > ub && ub > ulo);
  if (from_bound) uhi = ub;
  double u = uhi;
  if (kLazy && from_bound) {
    // (round 4) from the bound fm > 0 is known: the decisions of :102 that only need fl are taken before the first pass
    // (a group being zeroed -- froot(lmin) > 0, the bulk of a sparse iterate with xk != 0 -- then costs no pass at all)
    if (fabs(fl) <= 1e-12 * lmin) return BINF_LITERAL;
    if (fl > 0.0) return BINF_ZERO;
    // (round 4) nothing is decided on psi at the bound (fm > 0 is known), so the iteration may start anywhere inside the
    // bracket: ||S|| - sl is the root when nothing is active and close to it when little is -- fewer passes and piece steps
    // for the slowest lane of a wavefront (tools/r4/binf_small_emul.py: 2.68 -> 2.36 passes, 10.9 -> 7.9 steps on groups of 8)
    const double us = nS - sl;
    if (TEAM <= SPX_BINF_START_MAXTEAM && us > ulo && us < uhi) u = us;
  }
  double sa, sb, psi;
  double tau_full;  // tau of the last full pass (the one sa, sb belong to)
  {
    const double tau = u * fast_rcp(sl + u);
    binf_ab<TEAM>(grp, tau, delta, lds, sa, sb);
    psi = u - sqrt_pos(__builtin_fma(tau * tau, sa, sb));
    tau_full = tau;
  }
#ifdef SPX_DEBUG_BINF
  if ((threadIdx.x % TEAM) == 0 && blockIdx.x == 0 && threadIdx.x < TEAM)
    printf("[binf] sl %.17g lmin %.17g lmax %.17g ub %.17g nS %.17g nX %.17g mX %.17g fl %.17g from_bound %d u %.17g sa %.17g sb %.17g psi %.17g\n",
           sl, lmin, lmax, ub, nS, nX, mX, fl, (int)from_bound, u, sa, sb, psi);
#endif
  {
    // fm = froot(lmax) (:101).  When the bracket's upper end IS the root -- X = 0 and every entry thresholded at lmax,
    // the usual state at x0 = 0 with a wide trust region: lmax = ||S|| and froot(lmax) = ||S|| - ||S|| = 0 exactly in
    // the reference -- the A/B form returns a rounding-sized psi of either sign.  Its sign must not decide :102: such
    // groups (and a froot(lmin) that is not safely negative) go to the literal evaluation.
    if (!from_bound && fabs(psi) <= 1e-12 * u) {
      // The common instance, decided here: lmax = ||S|| (zlmax = 0, lambda ||X|| = 0) and no entry active at lmax.  The
      // reference then has froot(lmax) = ||S|| - ||-S|| = 0 exactly, fzero returns lmax, and w = S whatever the last bits
      // of the root are (:111 with everything thresholded): root = lmax, sums of the all-inactive piece.
      if (lmax_is_normS && sb == 0.0) {
        root_u = u;  // = lmax - sl
        return BINF_ROOT;
      }
      return BINF_LITERAL;
    }
    if (fabs(fl) <= 1e-12 * lmin) return BINF_LITERAL;
    const double fm = from_bound ? ((psi > 0.0) ? psi : 1.0) : (lmax * fast_rcp(u)) * psi;
    if (lazy && !from_bound) return BINF_LITERAL;           // (cannot happen: lazy requires ub > ul)
    if (fl * fm > 0) return BINF_ZERO;                      // :102
    if (!(fl < 0.0) || !(fm > 0.0)) return BINF_LITERAL;    // an exact zero at an end (or NaN)
  }
  double pa = -1.0, pb = -1.0;  // sums of the piece solved last (A, B >= 0 always)
  for (int it = 0; it < SPX_BINF_NEWTON_MAXIT; ++it) {
    // converged: psi vanishes to rounding (|psi| <= 4 eps u: u is the root of its own piece to the last bits), or the piece
    // just solved is confirmed by identical sums
    if (fabs(psi) <= 4 * eps * u || (sa == pa && sb == pb)) break;
    if (psi < 0.0) ulo = u; else uhi = u;
    // root of the current piece: g(v) = v - sqrt(sb + sa (v / (sl + v))^2), scalar Newton from u
    // (the slope only steers the step: unrefined v_rcp/v_rsq seeds, ~1e-8 relative, are enough there; the step that
    // follows a relative move below 1e-8 lands within ~1e-16 by quadratic convergence, so it is the last one)
    // (round 2: the result now depends on u itself -- the closed form of the last step -- so the piece root must be
    //  converged, not merely close in n.  g has a second root at v = 0; when the wanted one is small against sl the two
    //  are close on the scale of the start, Newton halves its way down (log2(n/u) steps: 12 were not always enough and the
    //  unconverged v was then "confirmed" by identical sums), and the all-inactive piece (sb == 0) is solved directly.)
    double v = u;
    bool piece_ok = false;
    if (sb == 0.0) {
      v = sqrt_pos(sa) - sl;  // g(v) = v (1 - sqrt(sa) / (sl + v)): the reference's 1 - sl/||S|| in disguise
      piece_ok = true;
    } else {
      // Round 4: the approach to the piece's root runs on the quartic the piece equation becomes in t = v / (sl + v),
      //   P(t) = (1 - t)^2 (sb + sa t^2) - sl^2 t^2  (the unique root in (0, 1) is the wanted one: both sides of
      //   sl t / (1 - t) = sqrt(sb + sa t^2) are positive there),
      // by Newton in its factored form: 6 fma/mul for P, 4 for P'/2, one v_rcp -- no square root, no refined reciprocal
      // (~18 issue slots a step against ~40 for the step in v below; on groups of <= 8 elements, one lane per group, the
      // piece solves were 40 % of the kernel's vector instructions and a wavefront waits for its slowest lane:
      // tools/r4/binf_small_emul.py).  It stops within 1e-8 of the root relative to min(t, 1 - t), i.e. within 2e-8 of v;
      // the fully accurate Newton step in v that follows is the one the iteration in v ended with, so the piece root has
      // the accuracy it had.  A piece on which this does not settle (t leaves (0, 1), sl^2 overflows, a vanishing slope)
      // takes the iteration in v as before.
      bool poly_ok = false;
#if SPX_BINF_POLY
      {
        const double c2 = sl * sl;
        double t = tau_full;  // = u / (sl + u): (sa, sb) belong to u
        for (int k = 0; k < 24; ++k) {
          const double om = 1.0 - t, t2 = t * t;
          const double w = __builtin_fma(sa, t2, sb);
          const double P = __builtin_fma(om * om, w, -(c2 * t2));
          const double dPh = __builtin_fma(-c2, t, om * __builtin_fma(om, sa * t, -w));  // P'(t) / 2
          const double tn = __builtin_fma(-0.5 * P, __builtin_amdgcn_rcp(dPh), t);
          const bool last = fabs(tn - t) <= 1e-8 * fmin(tn, 1.0 - tn);
          t = tn;
          if (!(t > 0.0 && t < 1.0)) break;  // (also NaN)
          if (last) { poly_ok = true; break; }
        }
        if (poly_ok) {
          v = sl * t * fast_rcp(1.0 - t);
          const double rn2 = fast_rcp(sl + v);
          const double t2 = v * rn2;
          const double p2 = __builtin_fma(t2 * t2, sa, sb);
          const double ph_ = sqrt_pos(p2);
          const double gp2 = 1.0 - ((ph_ > 0.0) ? sa * t2 * (sl * rn2 * rn2) * fast_rcp(ph_) : 0.0);
          v = v - (v - ph_) * fast_rcp(gp2);
          piece_ok = (v == v);
          poly_ok = piece_ok;
          if (!poly_ok) v = u;
        }
      }
#endif
      for (int k = 0; k < 64 && !poly_ok; ++k) {
        const double rn = fast_rcp(sl + v);
        const double t = v * rn;
        const double ph2 = __builtin_fma(t * t, sa, sb);
        const double rph = __builtin_amdgcn_rsq(ph2);
        double ph = ph2 * rph;
        ph = (ph2 > 0.0) ? __builtin_fma(0.5 * rph, __builtin_fma(-ph, ph, ph2), ph) : 0.0;
        ph = (ph2 > 0.0) ? __builtin_fma(0.5 * rph, __builtin_fma(-ph, ph, ph2), ph) : 0.0;
        const double g = v - ph;
        const double gp = 1.0 - ((ph2 > 0.0) ? sa * t * (sl * rn * rn) * rph : 0.0);
        const double vn = v - g * __builtin_amdgcn_rcp(gp);
        const bool last = fabs(vn - v) <= 1e-8 * fabs(vn);
        v = vn;
        if (last) {
          // one more, fully accurate step
          const double rn2 = fast_rcp(sl + v);
          const double t2 = v * rn2;
          const double p2 = __builtin_fma(t2 * t2, sa, sb);
          const double ph_ = sqrt_pos(p2);
          const double gp2 = 1.0 - ((ph_ > 0.0) ? sa * t2 * (sl * rn2 * rn2) * fast_rcp(ph_) : 0.0);
          v = v - (v - ph_) * fast_rcp(gp2);
          piece_ok = true;
          break;
        }
      }
    }
    // the piece's root is u itself (to rounding): converged -- must be seen BEFORE the bracket test below, which would
    // otherwise reject v == u (u has just become a bracket end) and bisect away from the root
    if (fabs(v - u) <= 4 * eps * fabs(u)) break;
    const bool exact_step = piece_ok && (v > ulo && v < uhi);  // v is the (converged) root of the piece (sa, sb)
    if (!exact_step) v = sqrt_pos(ulo) * sqrt_pos(uhi);  // safeguard: geometric bisection of the bracket
    if (!(v > ulo && v < uhi)) break;
    const bool small = fabs(v - u) <= 4 * eps * v;
    pa = exact_step ? sa : -1.0;  // only an exact piece root may be "confirmed" by identical sums in the next pass
    pb = exact_step ? sb : -1.0;
    u = v;
    if (small) break;
    const double tau = u * fast_rcp(sl + u);
    if constexpr (G::kReg) {
      // Cheap confirmation instead of a pass whose sums would only come back bit-identical: the active set at the piece's
      // root is the one the piece was built from (and no active z can have changed sign: |d tau| ||S|| <= 2 Delta) -> u
      // is the root.  Not tried after the first pass (it == 0): the start from the bound is far from the root and a
      // wavefront of several groups almost never confirms there.
      if (SPX_BINF_CONFIRM && exact_step && it >= 1 && fabs(tau - tau_full) * nS <= 2.0 * delta &&
          binf_same_active_set<TEAM>(grp, tau, tau_full, delta))
        break;
    }
    binf_ab<TEAM>(grp, tau, delta, lds, sa, sb);
    psi = u - sqrt_pos(__builtin_fma(tau * tau, sa, sb));
    tau_full = tau;
#ifdef SPX_DEBUG_BINF
    if ((threadIdx.x % TEAM) == 0 && blockIdx.x == 0 && threadIdx.x < TEAM)
      printf("[binf] it %d u %.17g sa %.17g sb %.17g psi %.17g ulo %.17g uhi %.17g exact %d\n", it, u, sa, sb, psi, ulo, uhi, (int)exact_step);
#endif
  }
  root_u = fmin(fmax(u, ul), lmax - sl);  // inside the reference's bracket [lmin, lmax]
  if (lazy) {
    // accepted only if psi at the (unknown) upper end lmax - sl >= lmax_lb - sl is positive beyond the 1e-12 u of the tie test
    // above: psi(uhi) >= tau (uhi - root) -- otherwise round 1 with the exact lmax
    const double tau_r = root_u * fast_rcp(sl + root_u);
    if (!(tau_r * ((lmax_lb - sl) - root_u) > 1e-9 * root_u)) {
      const double ansatz = lmin + 1.0;                                      // :97 (epsilon = 1)
      const double rsig = fast_rcp(sigma);
      const double stepa = ansatz * rsig * fast_rcp(ansatz - sl);            // :98
      const double thra = delta * stepa;
      double sz = 0.0;
      grp.for_each([&](double S, double X) {
        const double za = fabs(__builtin_fma(-stepa, X, S * rsig)) - thra;  // :99
        sz += (za > 0.0) ? za * za : 0.0;
      });
      sz = team_sum<TEAM>(sz, lds);
      const double lmax_x = nS + sigma * (sqrt_pos(sz) + 1.0 * lam * nX);    // :100
      if (!(tau_r * ((lmax_x - sl) - root_u) > 1e-9 * root_u)) return BINF_LITERAL;
    }
  }
  // spx_ctx_set_tuning key 9 (pole_lit): a root next to the pole of step(n) (u < n / 1000) is where the reference's own
  // Float64 evaluation is up to 4.5e-9 of the scale off its formula (n - sigma lambda and 1 - sigma lambda / ||w|| cancel);
  // the closed form above is the accurate side, the literal evaluation reproduces the reference's doubles -- for callers
  // who must reproduce a reference run (src/shiftedGroupNormL2Binf.jl:105-113)
  if (pole_lit && root_u * 1000.0 < sl + root_u) return BINF_LITERAL;
  return BINF_ROOT;
}

// the prox of one element at the root (:110-113): alpha w_i with alpha = tau and w_i = b_i / tau (active) or S_i
// (inactive), i.e. b_i = X_i + Delta sgn(tau S_i - X_i) or tau S_i -- continuous across the activity boundary
__device__ __forceinline__ double binf_y(double S, double X, double tau, double delta) {
  const double z = __builtin_fma(tau, S, -X);
  const double b = X + signed_delta(delta, z);
  return (fabs(z) > delta) ? b : tau * S;
}

// ---------------------------------------------------------------------------------------------
// One group on an element provider (for_each / store): ShiftedGroupNormL2 (:67-76) or ShiftedGroupNormL2Binf (:84-117).
// ---------------------------------------------------------------------------------------------
template <int TEAM, bool BINF, class GRP>
__device__ __forceinline__ void group_body(const GRP& grp, double* y, double lam, double sigma, double delta, double* lds,
                                           bool literal_only, bool pole_lit) {
  if constexpr (!BINF) {
    double ss = 0.0;
    grp.for_each([&](double S, double) { ss += S * S; });
    const double snorm = sqrt(team_sum<TEAM>(ss, lds));
    const double alpha = (snorm == 0.0) ? 0.0 : jl_max(1 - sigma * lam / snorm, 0.0);
    grp.store(y, [&](double S, double) { return (snorm == 0.0) ? 0.0 : alpha * S; });
  } else {
    double root, ru = 0.0;
    int status = literal_only ? BINF_LITERAL : binf_root<TEAM>(grp, lam, sigma, delta, lds, ru, pole_lit);
    const double sl = lam * sigma;
    if (status == BINF_LITERAL) {
      // the reference, literally, including its last step (:106-113) -- the root may lie below sl here
      const bool ok = binf_literal_root<TEAM>(grp, lam, sigma, delta, lds, root);
      if (!ok || (root - sl) == 0.0) {
        grp.store(y, [&](double, double) { return 0.0; });
      } else {
        const double step = root / (sigma * (root - sl));
        double sw = 0.0;
        grp.for_each([&](double S, double X) {
          double w = S - sigma * softthres(S / sigma - step * X, delta * step);
          sw += w * w;
        });
        const double nw = sqrt(team_sum<TEAM>(sw, lds));
        const double alpha = jl_max(0.0, 1 - sl / nw);
        grp.store(y, [&](double S, double X) {
          return alpha * (S - sigma * softthres(S / sigma - step * X, delta * step));
        });
      }
    } else if (status == BINF_ZERO || ru == 0.0) {
      grp.store(y, [&](double, double) { return 0.0; });
    } else {
      const double tau = ru / (sl + ru);  // = alpha at the root
      grp.store(y, [&](double S, double X) { return binf_y(S, X, tau, delta); });
    }
  }
}
