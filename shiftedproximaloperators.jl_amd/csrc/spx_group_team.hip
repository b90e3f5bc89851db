// spx_group_team.hip -- ShiftedGroupNormL2.prox! / ShiftedGroupNormL2Binf.prox! on FEW, LARGE groups: a group is owned by a
// TEAM OF WORKGROUPS of one resident grid.  First of all the reference's default GroupNormL2 -- `shifted(NormL2(lambda), xk)`
// wraps as GroupNormL2([lambda]) with idx = [:] (src/shiftedGroupNormL2.jl:34-35, src/shiftedGroupNormL2Binf.jl:48-49,
// src/groupNormL2.jl:30-31): ONE group over the whole vector -- which the one-workgroup-per-group kernels of spx_group.hip
// ran on 1/256 of the chip (n = 1e8: 384 ms plain, 1349 ms Binf, profiles/r04_big_groups_baseline.txt).
//
// HBM layout: as spx_group.hip (q, xk, sj, y contiguous fp64, groups = contiguous index ranges, one lambda per group).
// Algorithmic traffic 32 B/element; this form moves 32 B/element when the group fits on chip (S = (q + xk) + sj and X = xk
// parked in LDS, 9216 elements per workgroup: 2.36 Mi elements on 256 CUs) and 56 B/element beyond (one reducing pass, one
// storing pass).  Roofline: HBM bandwidth.
//
// Structure.  The reductions of a group (||S||, and for Binf the sums of the root find) are `team_sum<kTeamGrid>`: wavefront
// butterflies -> workgroup -> an exchange of the workgroup totals between the team's workgroups through self-flagged words
// (spx_group_common.hpp: grid_team_reduce), every workgroup forming bit-identical totals in a fixed order.  On top of that
// the per-group body IS the one of the other group kernels (group_body / binf_root / binf_literal_root on an element
// provider): ChipGroup reads the LDS-resident elements, StreamGroup streams the team's share of the vectors from HBM for
// every reduction (LDS-DMA staged like the separable skeleton).
// Binf, streaming form: the generic root find would stream the vectors once per reduction (5-6 passes, ~60 in the literal
// evaluation).  As ShiftedNormL1B2 (spx_b2.hip), a SAMPLE (read first, solved on chip) predicts the root, and the one
// reducing pass forms EVERY sum the reference's decisions need at parameters known beforehand (||S||, ||X||, max|X|,
// zlmax at the ansatz, the sums at lmin: src/shiftedGroupNormL2Binf.jl:94-100) and classifies each element against a bracket
// [tau_a, tau_b] around the prediction: z_i(tau) = tau S_i - X_i is monotone in tau, so an element inactive (|z| <= Delta)
// at both ends is inactive in between (a fixed term S_i^2 of A), one active with the same sign at both ends is active in
// between (a fixed term (X_i + Delta sgn z)^2 of B), and the few per cent whose breakpoint lies inside are recorded as
// (S_i, X_i) candidates in the wavefront's own region.  psi(u) = u - sqrt(B + tau^2 A) is then exact anywhere inside the
// bracket for a sweep over the candidates (microseconds), the piece iteration of binf_root runs to its fixed point on those
// sweeps, and the second pass stores y: 56 B/element.  Whatever the fast path cannot decide from this (bracket missed,
// candidate region full, degenerate brackets that need the reference's literal bisection) falls through to the generic
// body -- slower, same results.
#include "spx_group_common.hpp"

namespace {

constexpr int kTmThreads = 1024;
constexpr int kTmDmaKiB = 3;                                           // KiB per wavefront, vector and tile
constexpr int kTmLdsBytes = (kTmThreads / 64) * 3 * kTmDmaKiB * 1024;  // 144 KiB
constexpr int kTmChipElems = kTmLdsBytes / 16;                         // S and X of 9216 elements per workgroup
constexpr int64_t kTmTilePairs = (int64_t)(kTmThreads / 64) * 64 * kTmDmaKiB;  // 3072 pairs = 6144 elements per tile
constexpr int kTmMaxSpl = 4;                                           // samples per lane at most
constexpr int kTmMaxJobs = 65536;                                      // large groups a device-side plan holds at most
enum { kFormChip = 0, kFormStream = 1, kFormFast = 2 };               // the three kernels (k_group_team)

// Device-side plan of a ragged layout (CSR offsets): which groups are large, and which workgroups own them.
struct TeamJob {
  int64_t lo, hi;
  int gid, first, W, pad;
};
struct TeamPlanHdr {
  int active;   // 0: no large group (or more than the plan holds) -- nothing to do here
  int njobs;
  int loop;     // 1: more jobs than workgroups, W = 1, workgroup b takes jobs b, b + grid, ...
  int pad;
  int wg_job[kGtCols];  // !loop: the job of workgroup b, or -1
};

// One workgroup: large groups of a CSR layout -> jobs (in group order), teams sized by the groups' shares of the elements.
__global__ __launch_bounds__(1024) void k_team_plan(const int64_t* __restrict__ offsets, int64_t ngroups, int64_t n, int G,
                                                     int64_t big_min, TeamPlanHdr* hdr, TeamJob* jobs, int max_jobs) {
  __shared__ int cnt[1024];
  __shared__ double msum[16];
  const int t = threadIdx.x;
  const int64_t per = (ngroups + 1023) / 1024;
  const int64_t g0 = (int64_t)t * per, g1 = (g0 + per < ngroups) ? g0 + per : ngroups;
  auto range = [&](int64_t g, int64_t& lo, int64_t& hi) {
    lo = offsets[g];
    hi = offsets[g + 1];
    if (lo < 0) lo = 0;
    if (hi > n) hi = n;
    if (hi < lo) hi = lo;
  };
  int mine = 0;
  double mbig = 0.0;
  for (int64_t g = g0; g < g1; ++g) {
    int64_t lo, hi;
    range(g, lo, hi);
    if (hi - lo >= big_min) { ++mine; mbig += (double)(hi - lo); }
  }
  cnt[t] = mine;
  mbig = wave_sum(mbig);
  if ((t & 63) == 0) msum[t >> 6] = mbig;
  __syncthreads();
  if (t == 0) {  // exclusive scan (1024 entries: microseconds, once per call)
    int acc = 0;
    for (int k = 0; k < 1024; ++k) { const int c = cnt[k]; cnt[k] = acc; acc += c; }
    double M = 0.0;
    for (int k = 0; k < 16; ++k) M += msum[k];
    msum[0] = M;
    hdr->njobs = acc;
    hdr->active = (acc > 0 && acc <= max_jobs) ? 1 : 0;
    hdr->loop = (acc > G) ? 1 : 0;
  }
  __syncthreads();
  if (!hdr->active) return;
  int pos = cnt[t];
  for (int64_t g = g0; g < g1; ++g) {
    int64_t lo, hi;
    range(g, lo, hi);
    if (hi - lo >= big_min) {
      jobs[pos].lo = lo;
      jobs[pos].hi = hi;
      jobs[pos].gid = (int)g;
      jobs[pos].first = 0;
      jobs[pos].W = 1;
      ++pos;
    }
  }
  __syncthreads();
  if (t < kGtCols) hdr->wg_job[t] = -1;
  __syncthreads();
  if (t == 0 && !hdr->loop) {
    const int nj = hdr->njobs;
    const double M = msum[0];
    const int spare = G - nj;
    int first = 0;
    for (int j = 0; j < nj; ++j) {
      const double m = (double)(jobs[j].hi - jobs[j].lo);
      int W = 1 + (int)floor((double)spare * (m / M));
      const int wmax = (int)((jobs[j].hi - jobs[j].lo + 2047) / 2048);  // no fewer than ~2048 elements per workgroup
      if (W > wmax) W = wmax;
      if (W < 1) W = 1;
      if (first + W > G) W = G - first;
      jobs[j].first = first;
      jobs[j].W = W;
      for (int b = first; b < first + W; ++b) hdr->wg_job[b] = j;
      first += W;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// element providers (interface of spx_group_common.hpp: for_each(f(S, X)), store(y, f(S, X)))
// ---------------------------------------------------------------------------------------------
// LDS-resident share of a group: elements [base, base + cnt)
struct ChipGroup {
  static constexpr bool kReg = false;
  static constexpr int kEpl = 1;
  const double* S;   // LDS
  const double* X;   // LDS
  const double* sj;  // global
  int64_t base;      // first element of this workgroup's share
  int cnt;
  SpxSyncHeader* hdr;  // teams of several workgroups: a workgroup that gave up waiting for the others (its sums are garbage) stores NaN
  template <class F>
  __device__ __forceinline__ void for_each(F&& f) const {
    for (int j = threadIdx.x; j < cnt; j += kTmThreads) f(S[j], X[j]);
  }
  template <class F>
  __device__ __forceinline__ void store(double* y, F&& f) const {
    const bool poisoned = hdr != nullptr && spx_poisoned(hdr);
    for (int j = threadIdx.x; j < cnt; j += kTmThreads) {
      const double x = X[j], s = sj[base + j];
      const double v = f(S[j], x) - (x + s);
      y[base + j] = poisoned ? __longlong_as_double(0x7ff8000000000000ll) : v;
    }
  }
};

// The pairs [0, npairs) of the vectors from element `a` on, tiles of kTmTilePairs dealt round-robin to the W workgroups of
// the team (the same element -> lane mapping in every pass).  visit(valid, pair, q pair, xk pair, sj pair), by every lane.
// VEC: `a` sits on a 16-byte boundary of all vectors; LDS-DMA staging as in the separable skeleton (global_load_lds ... nt: no
// VGPR destination, 1 KiB per wave instruction; every wavefront owns 3 KiB per vector and tile: 144 KiB in flight per CU).
// !VEC (vectors of mixed alignment): 8-byte register loads, the next tile's issued before the current one is evaluated.
// ctr != NULL (VEC only; the storing pass, when it is the only pass of the launch that stores y): tiles are handed out by the
// team's atomic counter instead of round-robin, the next index fetched while the current tile is processed.
template <bool VEC, class V>
__device__ __forceinline__ void tm_stream(const double* q, const double* xk, const double* sj, int64_t a, int64_t npairs,
                                          int wl, int W, char* dma, unsigned int* ctr, unsigned int* next, V&& visit) {
  const int t = threadIdx.x;
  const int64_t ntiles = (npairs + kTmTilePairs - 1) / kTmTilePairs;
  if constexpr (VEC) {
    typedef __attribute__((address_space(3))) void lds_void;
    const int wave = t >> 6, lane = t & 63;
    char* wlds = dma + wave * (3 * kTmDmaKiB * 1024);
    const f64x2* q2 = reinterpret_cast<const f64x2*>(q + a);
    const f64x2* x2 = reinterpret_cast<const f64x2*>(xk + a);
    const f64x2* s2 = reinterpret_cast<const f64x2*>(sj + a);
    const bool dynamic = ctr != nullptr;
    int64_t tile = wl;
    if (dynamic) {
      if (t == 0) *next = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __syncthreads();
      tile = (int64_t)*next;
    }
    while (tile < ntiles) {
      if (dynamic) {
        __syncthreads();  // every lane has read *next
        if (t == 0) *next = __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (in flight during this tile)
      }
      const int64_t base = tile * kTmTilePairs + (int64_t)wave * (64 * kTmDmaKiB) + lane;
#pragma unroll
      for (int k = 0; k < kTmDmaKiB; ++k) {
        int64_t i = base + k * 64;
        if (i >= npairs) i = npairs - 1;
        __builtin_amdgcn_global_load_lds((const void*)(q2 + i), (lds_void*)(wlds + (0 * kTmDmaKiB + k) * 1024), 16, 0, 2);
        __builtin_amdgcn_global_load_lds((const void*)(x2 + i), (lds_void*)(wlds + (1 * kTmDmaKiB + k) * 1024), 16, 0, 2);
        __builtin_amdgcn_global_load_lds((const void*)(s2 + i), (lds_void*)(wlds + (2 * kTmDmaKiB + k) * 1024), 16, 0, 2);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int k = 0; k < kTmDmaKiB; ++k) {
        const int64_t i = base + k * 64;
        const f64x2 qa = *reinterpret_cast<const f64x2*>(wlds + (0 * kTmDmaKiB + k) * 1024 + lane * 16);
        const f64x2 xa = *reinterpret_cast<const f64x2*>(wlds + (1 * kTmDmaKiB + k) * 1024 + lane * 16);
        const f64x2 sa = *reinterpret_cast<const f64x2*>(wlds + (2 * kTmDmaKiB + k) * 1024 + lane * 16);
        visit(i < npairs, i, qa, xa, sa);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this tile's LDS reads are done before the next tile's loads are issued
      if (dynamic) {
        __syncthreads();
        tile = (int64_t)*next;
      } else {
        tile += W;
      }
    }
  } else {
    constexpr int KP = 2;
    constexpr int64_t kPairs = (int64_t)kTmThreads * KP;  // (its own tile size: any fixed mapping will do)
    const int64_t nt = (npairs + kPairs - 1) / kPairs;
    auto ld = [&](int64_t tile, f64x2* qa, f64x2* xa, f64x2* sa) {
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        int64_t i = tile * kPairs + t + k * kTmThreads;
        if (i >= npairs) i = npairs - 1;
        const int64_t e = a + 2 * i;
        qa[k] = f64x2{__builtin_nontemporal_load(q + e), __builtin_nontemporal_load(q + e + 1)};
        xa[k] = f64x2{__builtin_nontemporal_load(xk + e), __builtin_nontemporal_load(xk + e + 1)};
        sa[k] = f64x2{__builtin_nontemporal_load(sj + e), __builtin_nontemporal_load(sj + e + 1)};
      }
    };
    auto comp = [&](int64_t tile, const f64x2* qa, const f64x2* xa, const f64x2* sa) {
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        const int64_t i = tile * kPairs + t + k * kTmThreads;
        visit(i < npairs, i, qa[k], xa[k], sa[k]);
      }
    };
    f64x2 q0[KP], x0[KP], s0[KP], q1[KP], x1[KP], s1[KP];
    int64_t tile = wl;
    if (tile < nt) ld(tile, q0, x0, s0);
    while (tile < nt) {
      const int64_t t1 = tile + W;
      if (t1 < nt) ld(t1, q1, x1, s1);
      comp(tile, q0, x0, s0);
      const int64_t t2 = t1 + W;
      if (t1 < nt) {
        if (t2 < nt) ld(t2, q0, x0, s0);
        comp(t1, q1, x1, s1);
      }
      tile = t2;
    }
  }
}

// A group streamed from HBM by its team: pairs from `a` on, plus at most one element before (lo < a: the group starts off the
// 16-byte grid of the vectors) and one after the last pair; those ride with lane 0 of the team's first workgroup.
template <bool VEC>
struct StreamGroup {
  static constexpr bool kReg = false;
  static constexpr int kEpl = 1;
  const double* q;
  const double* xk;
  const double* sj;
  int64_t lo, hi, a, npairs;
  int wl, W;
  char* dma;
  unsigned int* ctr;   // tiles of the storing pass on demand (tm_stream), or NULL
  unsigned int* next;  // LDS word
  SpxSyncHeader* hdr;  // as ChipGroup::hdr
  template <class F>
  __device__ __forceinline__ void edges(F&& f) const {  // f(element index), lane 0 of workgroup 0 of the team
    if (wl == 0 && threadIdx.x == 0) {
      if (lo < a) f(lo);
      if (a + 2 * npairs < hi) f(hi - 1);
    }
  }
  template <class F>
  __device__ __forceinline__ void for_each(F&& f) const {
    if (npairs > 0)
      tm_stream<VEC>(q, xk, sj, a, npairs, wl, W, dma, nullptr, nullptr, [&](bool valid, int64_t, f64x2 qa, f64x2 xa, f64x2 sa) {
        if (valid) {
          f((qa.x + xa.x) + sa.x, xa.x);
          f((qa.y + xa.y) + sa.y, xa.y);
        }
      });
    edges([&](int64_t i) { const double x = xk[i]; f((q[i] + x) + sj[i], x); });
  }
  // y[i] = f(S, X) - (xk + sj); q[i] is read by the lane that writes y[i], before it does: y may alias q
  template <class F>
  __device__ __forceinline__ void store(double* y, F&& f_) const {
    const bool poisoned = hdr != nullptr && spx_poisoned(hdr);
    auto f = [&](double S, double X) -> double { const double v = f_(S, X); return poisoned ? __longlong_as_double(0x7ff8000000000000ll) : v; };
    if (npairs > 0)
      tm_stream<VEC>(q, xk, sj, a, npairs, wl, W, dma, VEC ? ctr : nullptr, next, [&](bool valid, int64_t p, f64x2 qa, f64x2 xa, f64x2 sa) {
        if (valid) {
          const f64x2 o{f((qa.x + xa.x) + sa.x, xa.x) - (xa.x + sa.x), f((qa.y + xa.y) + sa.y, xa.y) - (xa.y + sa.y)};
          if constexpr (VEC) __builtin_nontemporal_store(o, reinterpret_cast<f64x2*>(y + a) + p);
          else { __builtin_nontemporal_store(o.x, y + a + 2 * p); __builtin_nontemporal_store(o.y, y + a + 2 * p + 1); }
        }
      });
    edges([&](int64_t i) { const double x = xk[i], s = sj[i]; y[i] = f((q[i] + x) + s, x) - (x + s); });
  }
};

// root of the piece g(v) = v - sqrt(sb + sa (v / (sl + v))^2) by Newton from u (the scalar solve of binf_root, which see)
__device__ __forceinline__ double tm_piece_root(double sa, double sb, double sl, double u, bool& ok) {
  ok = false;
  if (sb == 0.0) { ok = true; return sqrt_pos(sa) - sl; }
  double v = u;
  for (int k = 0; k < 64; ++k) {
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
    if (last) {  // one more, fully accurate step
      const double rn2 = fast_rcp(sl + v);
      const double t2 = v * rn2;
      const double p2 = __builtin_fma(t2 * t2, sa, sb);
      const double ph_ = sqrt_pos(p2);
      const double gp2 = 1.0 - ((ph_ > 0.0) ? sa * t2 * (sl * rn2 * rn2) * fast_rcp(ph_) : 0.0);
      v = v - (v - ph_) * fast_rcp(gp2);
      ok = true;
      break;
    }
  }
  return v;
}

#ifdef SPX_TEAM_PROFILE  // A/B builds only: time stamps of workgroup 0 (10 ns units), read with spx_debug_team_stamps
__device__ unsigned long long g_tm_stamp[32];
__device__ int g_tm_nstamp;
#define TM_STAMP() do { if (FORM == kFormFast && blockIdx.x == 0 && threadIdx.x == 0 && nst < 32) { g_tm_stamp[nst++] = wall_clock64(); g_tm_nstamp = nst; } } while (0)
#else
#define TM_STAMP() do { } while (0)
#endif

constexpr int BINF_UNDECIDED = 3;  // the fast path cannot tell: the generic body decides (binf_root on the streamed group)

// The decisions of binf_root (spx_group_common.hpp; src/shiftedGroupNormL2Binf.jl:94-108) from the sums of the ONE reducing
// pass, P = {||S||^2, ||X||^2, zlmax^2 at the ansatz, A and B at lmin, ., ., ., max |X|, -min | |X| - Delta |}, and from
// evaluations of (A, B) INSIDE the bracket [ua, ub] (eval3: up to three points per sweep, exact there).  Same tests in the
// same order; the only difference is where the piece iteration starts (at the sample's prediction instead of the a-priori
// bound), which does not matter: froot is strictly increasing, the root is unique.  have_sz / have_l: the pass formed zlmax /
// the sums at lmin and the gap (it skips them when the sample says the decisions will not ask for them: `lean`); a decision
// that does ask after all is left to the generic body.
template <class Eval3>
__device__ __forceinline__ int binf_team_fast(const double* P, double lam, double sigma, double delta, bool have_bracket,
                                              bool have_sz, bool have_l, double ua, double ub, double uc, Eval3&& eval3,
                                              double& root_u, bool pole_lit) {
  const double eps = 2.220446049250313e-16;
  const double sl = lam * sigma;
  const double lmin = sl * (1 + eps);
  const double ul = lmin - sl;
  const double taul = ul * fast_rcp(lmin);
  const double sS = P[0], sX = P[1], sz = P[2], sal = P[3], sbl = P[4], mX = P[8], gap = -P[9];
  const double nS = sqrt_pos(sS), nX = sqrt_pos(sX);
  if (mX == 0.0 && sl < INFINITY && sl * (1.0 - 1e-9) > nS) return BINF_ZERO;
  const double ubound = sqrt_pos(sS + sX) * (1.0 + 8 * eps);
  const double lmax_lb = nS + sigma * (lam * nX);
  double lmax;
  bool lmax_is_normS = false;
  if (lmax_lb > lmin * (1.0 + 8 * eps) && (lmax_lb - sl) > ubound * (1.0 + 8 * eps)) {
    lmax = sl + ubound * (1.0 + 8 * eps);
  } else {
    if (!have_sz) return BINF_UNDECIDED;
    lmax = nS + sigma * (sqrt_pos(sz) + 1.0 * lam * nX);  // :100
    lmax_is_normS = (sz == 0.0) && (lam * nX == 0.0);
    const bool reversed = lmax < lmin * (1.0 - 1e-9) && ul > 0.0 && nS <= 1e6 * delta && (sS + sX < INFINITY);
    if (reversed && mX < delta * (1.0 - 1e-9)) return BINF_ZERO;
  }
  double fl;
  if ((sS + sX < INFINITY) && (mX - taul * nS > delta * (1.0 + 1e-9)) && (ul < 1e-9 * delta)) {
    fl = -1.0;
  } else {
    if (!have_l) return BINF_UNDECIDED;
    if (gap <= 1e-9 * delta) return BINF_LITERAL;
    fl = lmin - (lmin * fast_rcp(ul)) * sqrt_pos(__builtin_fma(taul * taul, sal, sbl));
  }
  if (!(lmin < lmax) || !(ul > 0.0) || !(fl == fl)) return BINF_LITERAL;
  double ulo = ul, uhi = lmax - sl;
  const bool from_bound = (uhi > ubound && ubound > ulo);
  if (from_bound) uhi = ubound;
  if (!have_bracket) return BINF_UNDECIDED;
  auto psi_of = [&](double u, double s_a, double s_b) -> double {
    const double tau = u * fast_rcp(sl + u);
    return u - sqrt_pos(__builtin_fma(tau * tau, s_a, s_b));
  };
  // the sample's bracket, clipped to the reference's; one sweep: both ends and the sample's root
  const double a = ua > ulo ? ua : ulo, b = ub < uhi ? ub : uhi;
  if (!(a < b)) return BINF_UNDECIDED;
  const double c = (uc > a && uc < b) ? uc : sqrt_pos(a) * sqrt_pos(b);
  if (!(c > a && c < b)) return BINF_UNDECIDED;
  double us[3] = {a, b, c}, sa3[3], sb3[3];
  eval3(us, 3, sa3, sb3);
  const double psia = psi_of(a, sa3[0], sb3[0]), psib = psi_of(b, sa3[1], sb3[1]);
  if (!from_bound) {
    // fm = froot(lmax) is an evaluation at the bracket's own end: possible here only if the sample's bracket reaches it
    if (b != uhi) return BINF_UNDECIDED;
    const double psi = psib, sb = sb3[1];
    if (fabs(psi) <= 1e-12 * uhi) {
      if (lmax_is_normS && sb == 0.0) { root_u = uhi; return BINF_ROOT; }
      return BINF_LITERAL;
    }
    if (fabs(fl) <= 1e-12 * lmin) return BINF_LITERAL;
    const double fm = (lmax * fast_rcp(uhi)) * psi;
    if (fl * fm > 0) return BINF_ZERO;
    if (!(fl < 0.0) || !(fm > 0.0)) return BINF_LITERAL;
  } else {
    if (fabs(fl) <= 1e-12 * lmin) return BINF_LITERAL;
    if (fl > 0.0) return BINF_ZERO;  // (fm > 0 from the bound)
    if (!(fl < 0.0)) return BINF_LITERAL;
  }
  double u;
  if (fabs(psia) <= 4 * eps * a) {
    u = a;
  } else if (fabs(psib) <= 4 * eps * b) {
    u = b;
  } else {
    if (!(psia < 0.0) || !(psib > 0.0)) return BINF_UNDECIDED;  // the sample misled: the root is not in its bracket
    ulo = a;
    uhi = b;
    u = c;
    double sa = sa3[2], sb = sb3[2];
    double psi = psi_of(u, sa, sb);
    double pa = -1.0, pb = -1.0;
    for (int it = 0; it < SPX_BINF_NEWTON_MAXIT; ++it) {  // (the loop of binf_root)
      if (fabs(psi) <= 4 * eps * u || (sa == pa && sb == pb)) break;
      if (psi < 0.0) ulo = u; else uhi = u;
      bool piece_ok;
      double v = tm_piece_root(sa, sb, sl, u, piece_ok);
      if (fabs(v - u) <= 4 * eps * fabs(u)) break;
      const bool exact_step = piece_ok && (v > ulo && v < uhi);
      if (!exact_step) v = sqrt_pos(ulo) * sqrt_pos(uhi);
      if (!(v > ulo && v < uhi)) break;
      const bool small = fabs(v - u) <= 4 * eps * v;
      pa = exact_step ? sa : -1.0;
      pb = exact_step ? sb : -1.0;
      u = v;
      if (small) break;
      us[0] = u;
      eval3(us, 1, sa3, sb3);
      sa = sa3[0];
      sb = sb3[0];
      psi = psi_of(u, sa, sb);
    }
  }
  root_u = fmin(fmax(u, ul), lmax - sl);
  if (pole_lit && root_u * 1000.0 < sl + root_u) return BINF_LITERAL;
  return BINF_ROOT;
}

// ---------------------------------------------------------------------------------------------
// The phases of the fast form (Binf, streamed).  Everything they return is team-uniform.
// ---------------------------------------------------------------------------------------------
struct TmSample {
  double u_s;        // the sample's root in u = n - sigma lambda, or -1: no prediction
  double u_a, u_b;   // the bracket around it
  int lean;          // the reducing pass may skip zlmax and the sums at lmin (the sample says nobody will ask for them)
};
// The sample: spl elements per lane, chunks of 32 consecutive elements (256 bytes) spread evenly over the group (on SORTED input
// a longer chunk is a run of nearly equal values: spx_b2.hip); its root by the piece iteration on sums scaled by m / nsample.
// A prediction: nothing is decided on it.
__device__ __noinline__ TmSample tm_sample(const double* q, const double* xk, const double* sj, int64_t lo, int64_t m, int wl,
                                           int W, GridTeam* gt, double sl, double sigma, double lam, double delta) {
  const int t = threadIdx.x;
  TmSample r;
  r.u_s = -1.0; r.u_a = 0.0; r.u_b = 0.0; r.lean = 0;
  int spl = (W >= 64) ? 1 : kTmMaxSpl;
  while (spl > 1 && 4 * (int64_t)W * kTmThreads * spl > m) spl >>= 1;
  const int64_t nsample = (int64_t)W * kTmThreads * spl;
  if (!(m >= 4 * nsample && m > 64)) return r;
  double Ss[kTmMaxSpl], Xs[kTmMaxSpl];
  const int64_t kchunks = nsample / 32;
#pragma unroll
  for (int s = 0; s < kTmMaxSpl; ++s) {
    Ss[s] = 0.0;
    Xs[s] = 0.0;
    if (s < spl) {
      const int64_t chunk = ((int64_t)wl * spl + s) * (kTmThreads / 32) + (t >> 5);
      const int64_t i = lo + (int64_t)((double)chunk * (double)(m - 32) / (double)(kchunks - 1)) + (t & 31);
      const double xv = xk[i];
      Ss[s] = (q[i] + xv) + sj[i];
      Xs[s] = xv;
    }
  }
  const double eps = 2.220446049250313e-16;
  const double lmin = sl * (1 + eps);
  const double ul = lmin - sl;
  const double taul = ul * fast_rcp(lmin);
  const double scale = (double)m / (double)nsample;
  double v3[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int s = 0; s < kTmMaxSpl; ++s) {
    v3[0] = __builtin_fma(Ss[s], Ss[s], v3[0]);
    v3[1] = __builtin_fma(Xs[s], Xs[s], v3[1]);
    v3[2] = fmax(v3[2], fabs(Xs[s]));
  }
  grid_team_reduce<3>(gt, v3, 4u);
  const double nSe = sqrt_pos(v3[0] * scale), nXe = sqrt_pos(v3[1] * scale), mXs = v3[2];
  const double ube = sqrt_pos((v3[0] + v3[1]) * scale);
  {
    // (margins of 10 % on the sample's estimates; max |X| over the sample is a lower bound of the true one)
    const double lmax_lb = 0.9 * (nSe + sigma * (lam * nXe));
    const bool skip_sz = lmax_lb > lmin * 1.01 && (lmax_lb - sl) > 1.1 * ube;
    const bool skip_l = (mXs - taul * nSe * 1.1 > delta * (1.0 + 1e-6)) && (ul < 1e-9 * delta);
    r.lean = (skip_sz && skip_l && (v3[0] + v3[1] < INFINITY)) ? 1 : 0;
  }
  double u = ube;
  for (int it = 0; it < 12 && u > 0.0 && u < INFINITY; ++it) {
    const double tau = u * fast_rcp(sl + u);
    double ab[2] = {0.0, 0.0};
#pragma unroll
    for (int s = 0; s < kTmMaxSpl; ++s) {
      const double z = __builtin_fma(tau, Ss[s], -Xs[s]);
      const bool act = fabs(z) > delta;
      const double bb = Xs[s] + signed_delta(delta, z);
      const double Sm = act ? 0.0 : Ss[s], bm = act ? bb : 0.0;
      ab[0] = __builtin_fma(Sm, Sm, ab[0]);
      ab[1] = __builtin_fma(bm, bm, ab[1]);
    }
    grid_team_reduce<2>(gt, ab, 0u);
    bool ok;
    const double v = tm_piece_root(ab[0] * scale, ab[1] * scale, sl, u, ok);
    if (!ok || !(v > 0.0) || !(v < INFINITY)) break;
    const bool done = fabs(v - u) <= 1e-4 * v;  // (far below the sample's own statistical error)
    u = v;
    if (done) { r.u_s = u; break; }
  }
  if (r.u_s > 0.0) {  // the bracket: half-width by the sample's size (its root is ~1 / sqrt(nsample) off)
    double hw = 4.0 / sqrt((double)nsample);
    if (hw < 0.015) hw = 0.015;
    r.u_a = r.u_s * (1.0 - hw);
    r.u_b = r.u_s * (1.0 + hw);
  }
  return r;
}

// The reducing pass: every sum the decisions of binf_root can ask for, at parameters known beforehand, + the classification
// of every element against the bracket.  Totals -> Pout (LDS; layout in binf_team_fast); returns this wavefront's candidates.
// LEAN: without zlmax, the sums at lmin and the gap (six accumulators instead of ten: the full form spills inside the loop).
template <bool VEC, bool LEAN>
__device__ __noinline__ unsigned int tm_pass1(const double* q, const double* xk, const double* sj, int64_t lo, int64_t hi,
                                              int64_t a, int64_t npairs, int wl, int W, char* dma, GridTeam* gt, double sl,
                                              double sigma, double delta, double u_a, double u_b, f64x2* myreg,
                                              unsigned int cand_cap, double* Pout) {
  const int t = threadIdx.x;
  const double eps = 2.220446049250313e-16;
  const double lmin = sl * (1 + eps);
  const double ul = lmin - sl;
  const double taul = ul * fast_rcp(lmin);
  const double ansatz = lmin + 1.0;  // :97
  const double rsig = fast_rcp(sigma);
  const double stepa = ansatz * rsig * fast_rcp(ansatz - sl);  // :98
  const double thra = delta * stepa;
  const bool have_bracket = u_b > 0.0;
  const double tau_a = have_bracket ? u_a * fast_rcp(sl + u_a) : 0.0, tau_b = have_bracket ? u_b * fast_rcp(sl + u_b) : 0.0;
  const int lane = t & 63;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  unsigned int ncand = 0;  // (wave-uniform)
  unsigned int nlost = 0;  // candidates that found their region full
  double P[10] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, -INFINITY};
  auto one = [&](bool valid, double S, double X) {  // called by every lane of a wavefront together
    const double Sv = valid ? S : 0.0, Xv = valid ? X : 0.0;  // (zeros are neutral in P[0], P[1], P[8])
    P[0] = __builtin_fma(Sv, Sv, P[0]);
    P[1] = __builtin_fma(Xv, Xv, P[1]);
    P[8] = fmax(P[8], fabs(Xv));
    if (!LEAN && valid) {
      P[9] = fmax(P[9], -fabs(fabs(X) - delta));
      const double za = fabs(__builtin_fma(-stepa, X, S * rsig)) - thra;  // |softthres| at the ansatz (:99)
      const double zm = (za > 0.0) ? za : 0.0;
      P[2] = __builtin_fma(zm, zm, P[2]);
      const double zl = __builtin_fma(taul, S, -X);  // the sums at lmin (:95)
      const bool actl = fabs(zl) > delta;
      const double bl = X + signed_delta(delta, zl);
      const double Sl = actl ? 0.0 : S, bm = actl ? bl : 0.0;
      P[3] = __builtin_fma(Sl, Sl, P[3]);
      P[4] = __builtin_fma(bm, bm, P[4]);
    }
    if (have_bracket) {  // (team-uniform; every lane takes part in the ballot)
      const double z1 = __builtin_fma(tau_a, S, -X), z2 = __builtin_fma(tau_b, S, -X);
      const bool a1 = fabs(z1) > delta, a2 = fabs(z2) > delta;
      const bool fixa = valid && !a1 && !a2;
      const bool fixb = valid && a1 && a2 && ((z1 > 0.0) == (z2 > 0.0));
      const double Sf = fixa ? S : 0.0;
      const double bf = fixb ? X + signed_delta(delta, z1) : 0.0;
      P[5] = __builtin_fma(Sf, Sf, P[5]);
      P[6] = __builtin_fma(bf, bf, P[6]);
      const bool is_cand = valid && !fixa && !fixb;
      const unsigned long long mk = __ballot(is_cand);
      if (mk) {
        const unsigned int pos = ncand + (unsigned int)__popcll(mk & lt_mask);
        if (is_cand) {
          if (pos < cand_cap) myreg[pos] = f64x2{S, X};
          else ++nlost;
        }
        ncand += (unsigned int)__popcll(mk);
      }
    }
  };
  if (npairs > 0)
    tm_stream<VEC>(q, xk, sj, a, npairs, wl, W, dma, nullptr, nullptr, [&](bool valid, int64_t, f64x2 qa, f64x2 xa, f64x2 sa) {
      one(valid, (qa.x + xa.x) + sa.x, xa.x);
      one(valid, (qa.y + xa.y) + sa.y, xa.y);
    });
  if (wl == 0 && t < 64) {  // the (at most two) elements off the pair grid ride with wavefront 0 of the team's first workgroup
    const bool e0 = lo < a, e1 = a + 2 * npairs < hi;
    if (e0) { const double xv = xk[lo]; one(t == 0, (q[lo] + xv) + sj[lo], xv); }
    if (e1) { const double xv = xk[hi - 1]; one(t == 0, (q[hi - 1] + xv) + sj[hi - 1], xv); }
  }
  P[7] = (double)nlost;
  grid_team_reduce<10>(gt, P, 0x300u);
  __syncthreads();
  if (t == 0) {  // (the same totals in every lane; static indices: P stays in registers)
#pragma unroll
    for (int k = 0; k < 10; ++k) Pout[k] = P[k];
  }
  __syncthreads();
  return ncand;
}

struct TmRoot {
  int status;
  double ru;
};
// The decisions (binf_team_fast) on the totals of the reducing pass, with (A, B) inside the bracket = the fixed part + ONE sweep
// over this team's candidates per evaluation (up to three points per sweep).
__device__ __noinline__ TmRoot tm_decide(const double* Pin, double lam, double sigma, double delta, double u_a, double u_b,
                                         double u_s, int lean, const f64x2* myreg, unsigned int ncand_mine, GridTeam* gt,
                                         bool pole_lit) {
  double P[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) P[k] = Pin[k];
  const double sl = lam * sigma;
  const int lane = threadIdx.x & 63;
  const double Afix = P[5], Bfix = P[6];
  auto eval3 = [&](const double* us, int cnt, double* sa, double* sb) {
    double tau[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) tau[k] = us[k < cnt ? k : 0] * fast_rcp(sl + us[k < cnt ? k : 0]);
    double ab[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    for (unsigned int e = (unsigned int)lane; e < ncand_mine; e += 64) {
      const f64x2 rec = myreg[e];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (k < cnt) {
          const double z = __builtin_fma(tau[k], rec.x, -rec.y);
          const bool act = fabs(z) > delta;
          const double bb = rec.y + signed_delta(delta, z);
          const double Sm = act ? 0.0 : rec.x, bm = act ? bb : 0.0;
          ab[2 * k] = __builtin_fma(Sm, Sm, ab[2 * k]);
          ab[2 * k + 1] = __builtin_fma(bm, bm, ab[2 * k + 1]);
        }
      }
    }
    if (cnt == 1) {
      double a2[2] = {ab[0], ab[1]};
      grid_team_reduce<2>(gt, a2, 0u);
      ab[0] = a2[0];
      ab[1] = a2[1];
    } else {
      grid_team_reduce<6>(gt, ab, 0u);
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) { sa[k] = Afix + ab[2 * k]; sb[k] = Bfix + ab[2 * k + 1]; }
  };
  TmRoot r;
  r.ru = 0.0;
  const bool bracket_ok = u_s > 0.0 && P[7] == 0.0;  // (P[7] != 0: a candidate region overflowed -- clustered breakpoints)
  r.status = binf_team_fast(P, lam, sigma, delta, bracket_ok, !lean, !lean, u_a, u_b, u_s, eval3, r.ru, pole_lit);
  return r;
}

// The storing pass.  mode 0: NaN (a workgroup gave up waiting), 1: zeros (:102-103 / :107-108), 2: the prox at the root (binf_y)
template <bool VEC>
__device__ __noinline__ void tm_store_pass(double* y, const double* q, const double* xk, const double* sj, int64_t lo,
                                           int64_t hi, int64_t a, int64_t npairs, int wl, int W, char* dma, unsigned int* ctr,
                                           unsigned int* next, int mode, double tau, double delta) {
  const double kNaN = __longlong_as_double(0x7ff8000000000000ll);
  auto out = [&](double S, double X, double s) -> double {
    const double v = (mode == 2) ? binf_y(S, X, tau, delta) : (mode == 1 ? 0.0 : kNaN);
    return v - (X + s);
  };
  if (npairs > 0)
    tm_stream<VEC>(q, xk, sj, a, npairs, wl, W, dma, ctr, next, [&](bool valid, int64_t p, f64x2 qa, f64x2 xa, f64x2 sa) {
      if (valid) {
        const f64x2 o{out((qa.x + xa.x) + sa.x, xa.x, sa.x), out((qa.y + xa.y) + sa.y, xa.y, sa.y)};
        if constexpr (VEC) __builtin_nontemporal_store(o, reinterpret_cast<f64x2*>(y + a) + p);
        else { __builtin_nontemporal_store(o.x, y + a + 2 * p); __builtin_nontemporal_store(o.y, y + a + 2 * p + 1); }
      }
    });
  if (wl == 0 && threadIdx.x == 0) {
    if (lo < a) { const double x = xk[lo], s = sj[lo]; y[lo] = out((q[lo] + x) + s, x, s); }
    if (a + 2 * npairs < hi) { const double x = xk[hi - 1], s = sj[hi - 1]; y[hi - 1] = out((q[hi - 1] + x) + s, x, s); }
  }
}

// =============================================================================================
// The kernels: a resident grid of 1024-lane workgroups; every workgroup belongs to one team, a team owns one group (or, with
// teams of one, a workgroup takes several groups in turn).  Three forms of one body, compiled apart (one kernel holding all
// three spilled 350-470 registers, and every reduction of the root find paid for it):
//   kFormChip    groups that fit into the LDS of their team (<= 9216 elements per workgroup): vectors read once
//   kFormStream  the others, generic: one streaming pass per reduction (ShiftedGroupNormL2: 2 passes; the Binf jobs the fast
//                form left undecided -- job_status -- with the full binf_root / binf_literal_root)
//   kFormFast    Binf, streamed: the sample-predicted two-pass path; what it decides it stores (job_status = 0)
// =============================================================================================
template <bool BINF, bool VEC, int FORM>
__global__ __launch_bounds__(kTmThreads) void k_group_team(double* y, const double* q, const double* xk, const double* sj,
                                                             int64_t n, int64_t gsize, int64_t ngroups, int Wu,
                                                             const TeamPlanHdr* plan, const TeamJob* jobs,
                                                             const double* __restrict__ lambda, double sigma, double delta,
                                                             int pole_lit, int par, int* job_status, int use_status /* bit 1: test hook */,
                                                             unsigned long long* rows, unsigned long long* clear_rows,
                                                             unsigned int* tile_ctr, unsigned int* clear_ctr, SpxSyncHeader* hdr,
                                                             f64x2* cand, unsigned int cand_cap) {
  __shared__ __attribute__((aligned(16))) char dma[kTmLdsBytes];
  __shared__ GridTeam gt;
  __shared__ unsigned int next_tile;
  __shared__ double team_P[10];  // (fast form) the totals of the reducing pass
  const int t = threadIdx.x;
  const int G = (int)gridDim.x;
  int nst = 0;
  TM_STAMP();
  // the other set of exchange words and tile counters, for the launch after this one (what the previous launch on this
  // context left in it)
  for (int64_t idx = (int64_t)blockIdx.x * kTmThreads + t; idx < (int64_t)kGtSetWords; idx += (int64_t)G * kTmThreads)
    clear_rows[idx] = 0ull;
  if (blockIdx.x == 0 && t < kGtCols) clear_ctr[t] = 0u;
  // ---- this workgroup's team and jobs
  int first = (int)blockIdx.x, W = 1;
  int64_t job = blockIdx.x, njobs = 0, job_step = G;
  if (plan) {
    if (!plan->active) return;
    njobs = plan->njobs;
    if (!plan->loop) {
      const int j = plan->wg_job[blockIdx.x];
      if (j < 0) return;
      first = jobs[j].first;
      W = jobs[j].W;
      job = j;
      njobs = j + 1;  // exactly this one
    }
  } else {
    njobs = ngroups;
    if (Wu > 1) {
      job = blockIdx.x / Wu;
      if (job >= ngroups) return;
      first = (int)job * Wu;
      W = Wu;
      njobs = job + 1;
    }
  }
  const int wl = (int)blockIdx.x - first;
  if (t == 0) { gt.rows = rows; gt.hdr = hdr; gt.first = first; gt.W = W; }
  if (t < 16) { gt.wnp[t] = 0; gt.wcalls[t] = 0; }
  __syncthreads();
  double* const lds = reinterpret_cast<double*>(&gt);
  const double kNaN = __longlong_as_double(0x7ff8000000000000ll);
  bool late = false;
#ifdef SPX_TEST_HOOKS
  // planted (tuning key 102, tests/test_gpu_robustness.py): the last workgroup of a team behaves as one that was not resident while
  // the others waited for it -- it takes part in no reduction and only runs once they have given up (the flag), then stores its share
  if ((use_status & 2) && W > 1 && wl == W - 1) {
    late = true;
    for (unsigned int spins = 0; !spx_poisoned(hdr) && spins < (1u << 22); ++spins) __builtin_amdgcn_s_sleep(20);
  }
  use_status &= 1;
#endif
  for (; job < njobs; job += job_step) {
    int64_t lo, hi;
    int gid;
    if (plan) { lo = jobs[job].lo; hi = jobs[job].hi; gid = jobs[job].gid; }
    else { lo = job * gsize; hi = lo + gsize; gid = (int)job; }
    const int64_t m = hi - lo;
    const double lam = lambda[gid];
    const bool on_chip = m <= (int64_t)W * kTmChipElems;
    if (late) {  // (test builds only) its share of the group, poisoned: NaN
      if (on_chip == (FORM == kFormChip)) {
        if (on_chip) {
          const int64_t chunk = (m + W - 1) / W;
          const int64_t base = lo + (int64_t)wl * chunk;
          int64_t c64 = hi - base;
          if (c64 > chunk) c64 = chunk;
          ChipGroup lg{reinterpret_cast<double*>(dma), reinterpret_cast<double*>(dma) + kTmChipElems, sj, base, c64 > 0 ? (int)c64 : 0, hdr};
          lg.store(y, [&](double, double) { return 0.0; });
        } else if (FORM != kFormStream || !BINF || !use_status || job_status[job] != 0) {
          StreamGroup<VEC> lg;
          lg.q = q; lg.xk = xk; lg.sj = sj; lg.lo = lo; lg.hi = hi;
          lg.a = VEC ? lo + ((lo + par) & 1) : lo;
          lg.npairs = (hi - lg.a) >> 1;
          lg.wl = wl; lg.W = W; lg.dma = dma; lg.ctr = nullptr; lg.next = &next_tile; lg.hdr = hdr;
          lg.store(y, [&](double, double) { return 0.0; });
        }
      }
      continue;
    }
    if constexpr (FORM == kFormChip) {
      if (!on_chip) continue;
      // ---- on chip: S = (q + xk) + sj and X = xk of this workgroup's share in LDS, the vectors are read once
      double* S = reinterpret_cast<double*>(dma);
      double* X = S + kTmChipElems;
      const int64_t chunk = (m + W - 1) / W;
      const int64_t base = lo + (int64_t)wl * chunk;
      int64_t c64 = hi - base;
      if (c64 > chunk) c64 = chunk;
      const int cnt = c64 > 0 ? (int)c64 : 0;
      if (cnt > 0) {
#pragma unroll
        for (int k = 0; k < kTmChipElems / kTmThreads; ++k) {  // clamped index, unconditional loads: all 27 in flight
          const int j = t + k * kTmThreads;
          const int64_t i = base + (j < cnt ? j : cnt - 1);
          const double xv = xk[i], qv = q[i], sv = sj[i];
          if (j < cnt) {
            S[j] = (qv + xv) + sv;  // shiftedGroupNormL2.jl:65 / shiftedGroupNormL2Binf.jl:80
            X[j] = xv;
          }
        }
      }
      __syncthreads();
      ChipGroup grp{S, X, sj, base, cnt, W > 1 ? hdr : nullptr};
      group_body<kTeamGrid, BINF>(grp, y, lam, sigma, delta, lds, false, pole_lit != 0);
      __syncthreads();  // all reads of S / X done before the next group overwrites them
    } else {
      if (on_chip) continue;
      // ---- streamed from HBM
      StreamGroup<VEC> grp;
      grp.q = q; grp.xk = xk; grp.sj = sj; grp.lo = lo; grp.hi = hi;
      grp.a = VEC ? lo + ((lo + par) & 1) : lo;
      grp.npairs = (hi - grp.a) >> 1;
      grp.wl = wl; grp.W = W; grp.dma = dma;
      grp.ctr = nullptr; grp.next = &next_tile; grp.hdr = W > 1 ? hdr : nullptr;
      if constexpr (FORM == kFormStream) {
        bool literal_only = false;
        if (BINF && use_status) {
          const int st = job_status[job];
          if (st == 0) continue;  // (stored by the fast form)
          literal_only = st == BINF_LITERAL;
        }
        // every path of group_body stores y exactly once, in its last pass: the tiles of that pass can be handed out on demand
        // (a persistent grid with a static partition waits for its slowest CU: spx_b2.hip, tools/exp/persistent_stream.hip)
        {
          const int64_t ntiles = (grp.npairs + kTmTilePairs - 1) / kTmTilePairs;
          if (VEC && njobs - job <= job_step && W > 1 && ntiles >= 8 * (int64_t)W) grp.ctr = tile_ctr + first;
        }
        group_body<kTeamGrid, BINF>(grp, y, lam, sigma, delta, lds, literal_only, pole_lit != 0);
      } else {
        static_assert(FORM != kFormFast || BINF, "the fast form is a Binf form");
        // (phases as functions of their own, not inlined: in one body the sample arrays, the ten sums of the reducing pass
        //  and three streaming loops spilled 400-550 registers, scratch traffic inside the loops included)
        const double sl = lam * sigma;
        const TmSample smp = tm_sample(q, xk, sj, lo, m, wl, W, &gt, sl, sigma, lam, delta);
        TM_STAMP();  // 1: sample solved
        f64x2* const myreg = cand + ((size_t)blockIdx.x * (kTmThreads / 64) + (size_t)(t >> 6)) * cand_cap;
        const unsigned int ncand =
            smp.lean ? tm_pass1<VEC, true>(q, xk, sj, lo, hi, grp.a, grp.npairs, wl, W, dma, &gt, sl, sigma, delta, smp.u_a, smp.u_b,
                                           myreg, cand_cap, team_P)
                     : tm_pass1<VEC, false>(q, xk, sj, lo, hi, grp.a, grp.npairs, wl, W, dma, &gt, sl, sigma, delta, smp.u_a, smp.u_b,
                                            myreg, cand_cap, team_P);
        TM_STAMP();  // 2: reducing pass streamed and exchanged
        const TmRoot rt = tm_decide(team_P, lam, sigma, delta, smp.u_a, smp.u_b, smp.u_s, smp.lean, myreg,
                                    ncand < cand_cap ? ncand : cand_cap, &gt, pole_lit != 0);
        TM_STAMP();  // 3: root found
        if (rt.status == BINF_UNDECIDED || rt.status == BINF_LITERAL) {
          if (wl == 0 && t == 0) job_status[job] = rt.status;  // (the generic form follows)
        } else {
          // the storing pass: the only pass of this launch that stores y, so its tiles can be handed out on demand (a
          // persistent grid with a static partition waits for its slowest CU: spx_b2.hip, tools/exp/persistent_stream.hip)
          const int64_t ntiles = (grp.npairs + kTmTilePairs - 1) / kTmTilePairs;
          unsigned int* ctr = (VEC && njobs - job <= job_step && W > 1 && ntiles >= 8 * (int64_t)W) ? tile_ctr + first : nullptr;
          int mode = 2;  // binf_y at the root
          double tau = 0.0;
          if (W > 1 && spx_poisoned(hdr)) mode = 0;  // (a workgroup gave up waiting: the sums were garbage) NaN
          else if (rt.status == BINF_ZERO || rt.ru == 0.0) mode = 1;  // zeros
          else tau = rt.ru / (sl + rt.ru);  // = alpha at the root
          tm_store_pass<VEC>(y, q, xk, sj, lo, hi, grp.a, grp.npairs, wl, W, dma, ctr, &next_tile, mode, tau, delta);
          if (wl == 0 && t == 0) job_status[job] = 0;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TM_STAMP();  // 4: y stored
#ifdef SPX_TEAM_PROFILE
        if (blockIdx.x == 0 && t == 0) { g_tm_stamp[30] = (unsigned long long)rt.status; g_tm_stamp[31] = (unsigned long long)gt.wnp[0]; }
#endif
      }
    }
  }
}

}  // namespace

#ifdef SPX_TEAM_PROFILE
extern "C" __attribute__((visibility("default"))) int spx_debug_team_stamps(unsigned long long* out32, int* count) {
  hipError_t e = hipMemcpyFromSymbol(out32, HIP_SYMBOL(g_tm_stamp), sizeof(g_tm_stamp));
  if (e == hipSuccess) e = hipMemcpyFromSymbol(count, HIP_SYMBOL(g_tm_nstamp), sizeof(int));
  return (int)e;
}
#endif

// ---------------------------------------------------------------------------------------------
// Host side: called by run_group (spx_group.hip) for contiguous groups too large for its one-workgroup-per-group kernels.
//   uniform groups (offsets == NULL):  spx_group_team_launch
//   CSR offsets:  spx_group_team_plan (k_team_plan: which groups are large, which workgroups own them; *active_dev points at
//   the plan's `active` word, which the caller's kernel for the OTHER groups reads to skip the large ones), the caller's
//   kernel, then spx_group_team_launch -- nothing in between may touch ctx->ws.
// ---------------------------------------------------------------------------------------------
namespace {
struct TeamLayout {
  bool vec;
  int par, G, Wu;
  int64_t grid;
  size_t plan_bytes, status_bytes, cand_bytes;
  unsigned int cand_cap;
};
int team_layout(spx_ctx* ctx, bool binf, const double* y, const double* q, const double* xk, const double* sj, int64_t n,
                const int64_t* offsets, int64_t gsize, int64_t ngroups, TeamLayout* L) {
  const auto bit3 = [](const void* p) { return (int)((reinterpret_cast<uintptr_t>(p) >> 3) & 1u); };
  L->par = bit3(q);
  L->vec = bit3(xk) == L->par && bit3(sj) == L->par && bit3(y) == L->par;
  // (every form is a 1024-lane workgroup holding 144 KiB of LDS: one per CU whatever the registers; asked of the largest)
  const void* fn = binf ? (L->vec ? reinterpret_cast<const void*>(&k_group_team<true, true, kFormFast>) : reinterpret_cast<const void*>(&k_group_team<true, false, kFormFast>))
                        : (L->vec ? reinterpret_cast<const void*>(&k_group_team<false, true, kFormStream>) : reinterpret_cast<const void*>(&k_group_team<false, false, kFormStream>));
  int64_t cap = spx_resident_cap(ctx, fn, kTmThreads, 0);
  if (cap > kGtCols) cap = kGtCols;
  if (cap < 1) return SPX_ERR_INTERNAL;  // (message set by spx_resident_cap)
  L->G = (int)cap;
  L->Wu = 1;
  L->grid = L->G;
  int64_t per_wg = 2 * ((n + L->G - 1) / L->G);  // elements of one job per workgroup (a device-side plan: at most about that)
  if (!offsets) {
    if (ngroups < L->G) {
      int64_t w = L->G / ngroups;
      const int64_t wmax = (gsize + 2047) / 2048;  // no fewer than ~2048 elements per workgroup
      if (w > wmax) w = wmax;
      if (w < 1) w = 1;
      L->Wu = (int)w;
      L->grid = ngroups * w;
    }
    per_wg = (gsize + L->Wu - 1) / L->Wu;
  }
  L->plan_bytes = offsets ? ((sizeof(TeamPlanHdr) + (size_t)kTmMaxJobs * sizeof(TeamJob) + 255) & ~(size_t)255) : 0;
  {  // one word per job: at most kTmMaxJobs jobs in a device-side plan, ngroups jobs for uniform groups
    size_t jobs_max = (size_t)kTmMaxJobs;
    if (!offsets && (size_t)ngroups > jobs_max) jobs_max = (size_t)ngroups;
    L->status_bytes = (jobs_max * sizeof(int) + 255) & ~(size_t)255;
  }
  L->cand_cap = 0;
  L->cand_bytes = 0;
  if (binf && ctx->tune_team_fast) {
    // one candidate region per wavefront of the grid: a quarter of the wavefront's share of a job (the breakpoints inside a
    // bracket of +-hw around the sample's root are a few times hw of the elements on ordinary data; a full region = the
    // generic body decides)
    int64_t c = per_wg / (kTmThreads / 64) / 4 + 64;
    if (c > 0x7fffffff) c = 0x7fffffff;
    L->cand_cap = (unsigned int)c;
    L->cand_bytes = (size_t)L->grid * (kTmThreads / 64) * (size_t)c * sizeof(f64x2);
  }
  return SPX_OK;
}
}  // namespace

int spx_group_team_max_grid(spx_ctx* ctx, bool binf) {  // workgroups a team launch can have at most (<= kGtCols)
  TeamLayout L;
  if (team_layout(ctx, binf, nullptr, nullptr, nullptr, nullptr, 0, nullptr, 1, 1, &L)) return 0;
  return L.G;
}

int spx_group_team_plan(spx_ctx* ctx, bool binf, const double* y, const double* q, const double* xk, const double* sj,
                        int64_t n, const int64_t* offsets, int64_t ngroups, int64_t big_min, const int** active_dev) {
  TeamLayout L;
  int rc = team_layout(ctx, binf, y, q, xk, sj, n, offsets, 0, ngroups, &L);
  if (rc) return rc;
  rc = spx_ws_reserve(ctx, L.plan_bytes + L.status_bytes + L.cand_bytes + 256);
  if (rc) return rc;
  rc = spx_sync_reserve(ctx, kSpxSyncTeamOffset + kSpxSyncTeamBytes);
  if (rc) return rc;
  TeamPlanHdr* plan = reinterpret_cast<TeamPlanHdr*>(ctx->ws);
  TeamJob* jobs = reinterpret_cast<TeamJob*>(static_cast<char*>(ctx->ws) + sizeof(TeamPlanHdr));
  hipLaunchKernelGGL(k_team_plan, dim3(1), dim3(1024), 0, ctx->stream, offsets, ngroups, n, L.G, big_min, plan, jobs, kTmMaxJobs);
  SPX_LAUNCH_CHECK();
  *active_dev = &plan->active;
  return SPX_OK;
}

int spx_group_team_launch(spx_ctx* ctx, bool binf, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                          const int64_t* offsets, int64_t gsize, int64_t ngroups, const double* lambda, double sigma,
                          double delta) {
  TeamLayout L;
  int rc = team_layout(ctx, binf, y, q, xk, sj, n, offsets, gsize, ngroups, &L);
  if (rc) return rc;
  rc = spx_ws_reserve(ctx, L.plan_bytes + L.status_bytes + L.cand_bytes + 256);  // (after spx_group_team_plan: the same size, nothing moves)
  if (rc) return rc;
  rc = spx_sync_reserve(ctx, kSpxSyncTeamOffset + kSpxSyncTeamBytes);
  if (rc) return rc;
  char* ws = static_cast<char*>(ctx->ws);
  const TeamPlanHdr* plan = offsets ? reinterpret_cast<const TeamPlanHdr*>(ws) : nullptr;
  const TeamJob* jobs = offsets ? reinterpret_cast<const TeamJob*>(ws + sizeof(TeamPlanHdr)) : nullptr;
  int* job_status = reinterpret_cast<int*>(ws + L.plan_bytes);
  f64x2* cand = L.cand_bytes ? reinterpret_cast<f64x2*>(ws + L.plan_bytes + L.status_bytes) : nullptr;
  SpxSyncHeader* hdr = reinterpret_cast<SpxSyncHeader*>(ctx->sync);
  unsigned long long* sets = reinterpret_cast<unsigned long long*>(static_cast<char*>(ctx->sync) + kSpxSyncTeamOffset);
  unsigned int* ctrs = reinterpret_cast<unsigned int*>(sets + 2 * kGtSetWords);  // two sets of kGtCols tile counters
  const bool graph_safe = spx_capture_check(ctx) || ctx->graph_safe;  // (see spx_ctx::graph_safe)
  // which forms can occur: uniform groups are all alike; a device-side plan may hold both kinds
  const bool chip_possible = offsets ? true : gsize <= (int64_t)L.Wu * kTmChipElems;
  const bool stream_possible = offsets ? true : !chip_possible;
  const bool fast = binf && ctx->tune_team_fast && cand != nullptr;
  auto launch = [&](int form) -> int {
    int use = ctx->team_set, other = use ^ 1;
    if (graph_safe) {  // set 0 (words and counters), zeroed by nodes in front of the launch; nothing alternates
      use = 0;
      other = 1;
      int rz = spx_zero_async(ctx, sets, kGtSetWords * sizeof(unsigned long long));
      if (rz) return rz;
      rz = spx_zero_async(ctx, ctrs, kGtCols * sizeof(unsigned int));
      if (rz) return rz;
    }
    unsigned long long* rows = sets + (size_t)use * kGtSetWords;
    unsigned long long* clear_rows = sets + (size_t)other * kGtSetWords;
    unsigned int* tile_ctr = ctrs + (size_t)use * kGtCols;
    unsigned int* clear_ctr = ctrs + (size_t)other * kGtCols;
    int use_status = (form == kFormStream && fast) ? 1 : 0;
#ifdef SPX_TEST_HOOKS
    if (ctx->tune_force_team > 0) use_status |= 2;
#endif
    {
      SpxCoopLaunchGuard guard(ctx);
#define SPX_TEAM_LAUNCH(B, V, F)                                                                                                \
  hipLaunchKernelGGL((k_group_team<B, V, F>), dim3((unsigned)L.grid), dim3(kTmThreads), 0, ctx->stream, y, q, xk, sj, n, gsize,  \
                     ngroups, L.Wu, plan, jobs, lambda, sigma, delta, ctx->tune_binf_literal, L.par, job_status, use_status,     \
                     rows, clear_rows, tile_ctr, clear_ctr, hdr, cand, L.cand_cap)
#define SPX_TEAM_LAUNCH_BV(F)                                                                  \
  do {                                                                                         \
    if (binf) { if (L.vec) SPX_TEAM_LAUNCH(true, true, F); else SPX_TEAM_LAUNCH(true, false, F); } \
    else { if (L.vec) SPX_TEAM_LAUNCH(false, true, F); else SPX_TEAM_LAUNCH(false, false, F); }    \
  } while (0)
      if (form == kFormChip) SPX_TEAM_LAUNCH_BV(kFormChip);
      else if (form == kFormStream) SPX_TEAM_LAUNCH_BV(kFormStream);
      else { if (L.vec) SPX_TEAM_LAUNCH(true, true, kFormFast); else SPX_TEAM_LAUNCH(true, false, kFormFast); }
#undef SPX_TEAM_LAUNCH_BV
#undef SPX_TEAM_LAUNCH
    }
    if (!graph_safe) ctx->team_set = other;
    SPX_LAUNCH_CHECK();
    return SPX_OK;
  };
  if (chip_possible) { rc = launch(kFormChip); if (rc) return rc; }
  if (stream_possible) {
    if (fast) { rc = launch(kFormFast); if (rc) return rc; }
    rc = launch(kFormStream);  // (Binf after the fast form: only what that left undecided -- usually nothing, the kernel returns at once)
    if (rc) return rc;
  }
  return SPX_OK;
}
