// spx_group.hip -- ShiftedGroupNormL2.prox! and ShiftedGroupNormL2Binf.prox! (group-l2 block soft-threshold,
// with an l-infinity trust region in the Binf form).
//
// HBM layout: q, xk, sj, y contiguous fp64; groups are contiguous index ranges (CSR offsets or a uniform
// size); lambda is one fp64 per group.  Algorithmic traffic: 32 B/element + 8 B/group.
// Roofline: HBM bandwidth; the Binf form adds a per-group scalar root find that runs out of registers.
//
// Mapping: one TEAM per group, TEAM = one 64-lane wavefront (small groups) or one 256-lane workgroup
// (large groups).  Reductions: xor-butterfly over the wavefront (wave_sum), plus an LDS hop for 256-lane
// teams.  The fast path (uniform group size 64*EPL, EPL <= 8, 16-byte aligned) keeps the whole group in
// registers: each lane owns EPL/2 (or 1) 16-byte pairs, so q/xk/sj are read once and y written once.
#include <cmath>

#include "spx_common.hpp"

// ---------------------------------------------------------------------------------------------
// team reductions
// ---------------------------------------------------------------------------------------------
template <int TEAM>
__device__ __forceinline__ double team_sum(double v, double* lds /* 8 doubles per block, TEAM==256 only */) {
  v = wave_sum(v);
  if constexpr (TEAM == 256) {
    const int w = threadIdx.x >> 6;
    __syncthreads();  // previous use of lds finished
    if ((threadIdx.x & 63) == 0) lds[w] = v;
    __syncthreads();
    v = (lds[0] + lds[1]) + (lds[2] + lds[3]);
  }
  return v;
}
template <int TEAM>
__device__ __forceinline__ void team_sum2(double& a, double& b, double* lds) {
  a = wave_sum(a);
  b = wave_sum(b);
  if constexpr (TEAM == 256) {
    const int w = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { lds[w] = a; lds[4 + w] = b; }
    __syncthreads();
    a = (lds[0] + lds[1]) + (lds[2] + lds[3]);
    b = (lds[4] + lds[5]) + (lds[6] + lds[7]);
  }
}

// softthres(x, a) = sign(x) * max(0, |x| - a)          src/shiftedGroupNormL2Binf.jl:82
__device__ __forceinline__ double softthres(double x, double a) { return jl_sign(x) * jl_max(0.0, fabs(x) - a); }

// ---------------------------------------------------------------------------------------------
// Per-group element access.  Two providers with the same interface:
//   RegGroup<EPL>  : the group's S = (q + xk) + sj and X = xk live in registers (fast path)
//   MemGroup<TEAM> : elements are re-read from global memory (L1/L2 resident for moderate groups)
// for_each(f) calls f(S_i, X_i, slot) for every element this lane owns.
// ---------------------------------------------------------------------------------------------
template <int EPL>
struct RegGroup {
  double S[EPL], X[EPL], XS[EPL];  // XS = xk + sj (subtracted at the end)
  template <class F>
  __device__ __forceinline__ void for_each(F&& f) const {
#pragma unroll
    for (int k = 0; k < EPL; ++k) f(S[k], X[k], k);
  }
};

template <int TEAM>
struct MemGroup {
  const double* q;
  const double* xk;
  const double* sj;
  int64_t lo, hi;
  int lane;  // index inside the team
  template <class F>
  __device__ __forceinline__ void for_each(F&& f) const {
    for (int64_t i = lo + lane; i < hi; i += TEAM) {
      double x = xk[i];
      f((q[i] + x) + sj[i], x, 0);
    }
  }
};

// ---------------------------------------------------------------------------------------------
// Binf root find.  src/shiftedGroupNormL2Binf.jl:85-108
//   froot(n) = n - || sigma * softthres(S/sigma - step X, Delta step) - S ||,  step = n / (sigma (n - sl))
//
// Structure used here.  With u = n - sl > 0 and tau = u / n = 1 / (sigma step) in (0, 1) an element is
// thresholded to zero iff |tau S_i - X_i| <= Delta (its term is then -S_i); otherwise the term equals
// -(n/u) (X_i + Delta sgn(tau S_i - X_i)).  So
//   froot(n) = (n/u) * psi(u),   psi(u) = u - phi(u),   phi(u) = sqrt(B(u) + tau^2 A(u)),
//   A = sum_{inactive} S_i^2,  B = sum_{active} (X_i + Delta sgn(tau S_i - X_i))^2.
// Every |term| is non-increasing in n, hence froot is strictly increasing on n > sl: the root inside the
// reference's bracket [lmin, lmax] is unique, and a bracketing iteration of any kind lands on the root
// the reference's bisection (Roots.fzero) converges to.  psi is solved in u (no cancellation in n - sl,
// no pole) by a bracket-safeguarded Newton iteration; the result is then POLISHED on the literal froot:
// the adjacent pair of doubles with froot(a) < 0 < froot(b) is located and the end with the smaller
// |froot| returned -- exactly the double Roots' bisection-to-exhaustion returns.  All loops are bounded.
// ---------------------------------------------------------------------------------------------
#define SPX_BINF_NEWTON_MAXIT 60
#define SPX_BINF_WALK_MAXIT 6

// literal froot(n)  (:87-93)
template <int TEAM, class G>
__device__ __forceinline__ double binf_froot(const G& grp, double n, double sigma, double sl, double delta,
                                             double* lds) {
  const double step = n / (sigma * (n - sl));
  const double thr = delta * step;
  double sw = 0.0;
  grp.for_each([&](double S, double X, int) {
    double w = sigma * softthres(S / sigma - step * X, thr) - S;
    sw += w * w;
  });
  return n - sqrt(team_sum<TEAM>(sw, lds));
}

// psi(u) and psi'(u)
template <int TEAM, class G>
__device__ __forceinline__ void binf_psi(const G& grp, double u, double sl, double delta, double* lds, double& psi,
                                         double& dpsi) {
  const double n = sl + u;
  const double tau = u / n;
  double sa = 0.0, sb = 0.0;
  grp.for_each([&](double S, double X, int) {
    const double z = tau * S - X;
    const bool act = fabs(z) > delta;
    const double b = X + ((z > 0.0) ? delta : -delta);
    sa += act ? 0.0 : S * S;
    sb += act ? b * b : 0.0;
  });
  team_sum2<TEAM>(sa, sb, lds);
  const double phi = sqrt(sb + tau * tau * sa);
  psi = u - phi;
  const double dtau = sl / (n * n);
  dpsi = 1.0 - ((phi > 0.0) ? (sa * tau * dtau / phi) : 0.0);
}

__device__ __forceinline__ double next_up(double x) { return __longlong_as_double(__double_as_longlong(x) + 1); }
__device__ __forceinline__ double next_down(double x) { return __longlong_as_double(__double_as_longlong(x) - 1); }
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
    const double fmid = binf_froot<TEAM>(grp, m, sigma, sl, delta, lds);
    if (jl_sign(fa) * jl_sign(fmid) < 0) { b = m; fb = fmid; }
    else { a = m; fa = fmid; }
  }
  return (fabs(fa) < fabs(fb)) ? a : b;
}

// returns true and the root in `root`, or false when fl * fm > 0 (reference writes zeros, :102-103)
template <int TEAM, class G>
__device__ __forceinline__ bool binf_root(const G& grp, double lam, double sigma, double delta, double* lds,
                                          double& root) {
  const double eps = 2.220446049250313e-16;
  const double sl = lam * sigma;          // :85
  const double lmin = sl * (1 + eps);     // :94
  const double fl = binf_froot<TEAM>(grp, lmin, sigma, sl, delta, lds);  // :95
  const double ansatz = lmin + 1.0;                                      // :97 (epsilon = 1)
  const double stepa = ansatz / (sigma * (ansatz - sl));                 // :98
  double sz = 0.0, sS = 0.0, sX = 0.0;
  grp.for_each([&](double S, double X, int) {
    double z = softthres(S / sigma - stepa * X, delta * stepa);  // :99
    sz += z * z;
    sS += S * S;
    sX += X * X;
  });
  team_sum2<TEAM>(sz, sS, lds);
  sX = team_sum<TEAM>(sX, lds);
  const double lmax = sqrt(sS) + sigma * (sqrt(sz) + 1.0 * lam * sqrt(sX));  // :100 (|(eps-1)/eps + 1| = 1)
  const double fm = binf_froot<TEAM>(grp, lmax, sigma, sl, delta, lds);     // :101
  if (fl * fm > 0) return false;                                            // :102
  if (!(lmin < lmax) || !(fl < 0.0) || !(fm > 0.0)) {
    // Degenerate bracket (||S|| + sigma (zlmax + lambda ||X||) <= sigma lambda puts the "upper" end at or below
    // the pole n = sl of step(n)), an exact zero at an end, or a NaN: do literally what Roots.fzero does.
    root = binf_bisect<TEAM>(grp, lmin, fl, lmax, fm, sigma, sl, delta, lds);
    return true;
  }
  // regular bracket: fl < 0 < fm, sl < lmin < lmax.  Newton on psi(u), u = n - sl in [ulo, uhi].
  double ulo = lmin - sl, uhi = lmax - sl;
  double u = uhi, psi, dpsi;
  binf_psi<TEAM>(grp, u, sl, delta, lds, psi, dpsi);
  for (int it = 0; it < SPX_BINF_NEWTON_MAXIT; ++it) {
    double un = u - psi / dpsi;
    if (!(un > ulo && un < uhi)) un = sqrt(ulo) * sqrt(uhi);  // geometric bisection: the bracket spans decades
    if (!(un > ulo && un < uhi)) break;
    double psin, dpsin;
    binf_psi<TEAM>(grp, un, sl, delta, lds, psin, dpsin);
    const bool small = fabs(un - u) <= 4 * eps * un;
    u = un; psi = psin; dpsi = dpsin;
    if (psin == 0.0 || small) break;
    if (psin < 0.0) ulo = un; else uhi = un;
  }
  // polish on the literal froot: find adjacent doubles a < b with froot(a) < 0 < froot(b)
  double n0 = sl + u;
  n0 = fmin(fmax(n0, lmin), lmax);
  double f0 = binf_froot<TEAM>(grp, n0, sigma, sl, delta, lds);
  if (f0 == 0.0) { root = n0; return true; }
  double a = lmin, fa = fl, b = lmax, fb = fm;  // running bracket for the fallback
  bool found = false;
  if (f0 < 0.0) {
    a = n0; fa = f0;
    for (int k = 0; k < SPX_BINF_WALK_MAXIT && a < lmax; ++k) {
      const double n1 = next_up(a);
      const double f1 = binf_froot<TEAM>(grp, n1, sigma, sl, delta, lds);
      if (f1 < 0.0) { a = n1; fa = f1; }
      else { b = n1; fb = f1; found = true; break; }
    }
  } else {
    b = n0; fb = f0;
    for (int k = 0; k < SPX_BINF_WALK_MAXIT && b > lmin; ++k) {
      const double n1 = next_down(b);
      const double f1 = binf_froot<TEAM>(grp, n1, sigma, sl, delta, lds);
      if (f1 > 0.0) { b = n1; fb = f1; }
      else { a = n1; fa = f1; found = true; break; }
    }
  }
  if (found) {
    root = (fb == 0.0) ? b : ((fa == 0.0) ? a : ((fabs(fa) < fabs(fb)) ? a : b));
    return true;
  }
  root = binf_bisect<TEAM>(grp, a, fa, b, fb, sigma, sl, delta, lds);  // rare: Newton ended far from the sign change
  return true;
}

// ---------------------------------------------------------------------------------------------
// fast path kernel: uniform groups of 64*EPL elements, wave per group, group resident in registers
// ---------------------------------------------------------------------------------------------
template <int EPL, bool BINF>
__global__ __launch_bounds__(256) void k_group_reg(double* y_, const double* q_, const double* xk_, const double* sj_,
                                                    int64_t ngroups, const double* __restrict__ lambda, double sigma,
                                                    double delta) {
  static_assert(EPL == 1 || (EPL % 2) == 0, "EPL must be 1 or even");
  constexpr int GS = 64 * EPL;
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t g = wave; g < ngroups; g += nwaves) {
    RegGroup<EPL> grp;
    const int64_t base = g * GS;
    if constexpr (EPL == 1) {
      double qq = q_[base + lane], xx = xk_[base + lane], ss = sj_[base + lane];
      grp.S[0] = (qq + xx) + ss;  // shiftedGroupNormL2.jl:65 / shiftedGroupNormL2Binf.jl:80
      grp.X[0] = xx;
      grp.XS[0] = xx + ss;
    } else {
      const f64x2* q2 = reinterpret_cast<const f64x2*>(q_ + base);
      const f64x2* x2 = reinterpret_cast<const f64x2*>(xk_ + base);
      const f64x2* s2 = reinterpret_cast<const f64x2*>(sj_ + base);
      f64x2 vq[EPL / 2], vx[EPL / 2], vs[EPL / 2];
#pragma unroll
      for (int k = 0; k < EPL / 2; ++k) {
        vq[k] = q2[k * 64 + lane];
        vx[k] = x2[k * 64 + lane];
        vs[k] = s2[k * 64 + lane];
      }
#pragma unroll
      for (int k = 0; k < EPL / 2; ++k) {
        grp.S[2 * k] = (vq[k].x + vx[k].x) + vs[k].x;
        grp.S[2 * k + 1] = (vq[k].y + vx[k].y) + vs[k].y;
        grp.X[2 * k] = vx[k].x;
        grp.X[2 * k + 1] = vx[k].y;
        grp.XS[2 * k] = vx[k].x + vs[k].x;
        grp.XS[2 * k + 1] = vx[k].y + vs[k].y;
      }
    }
    const double lam = lambda[g];
    double out[EPL];
    if constexpr (!BINF) {
      double ss = 0.0;
#pragma unroll
      for (int k = 0; k < EPL; ++k) ss += grp.S[k] * grp.S[k];
      const double snorm = sqrt(wave_sum(ss));                                     // shiftedGroupNormL2.jl:69
      const double alpha = (snorm == 0.0) ? 0.0 : jl_max(1 - sigma * lam / snorm, 0.0);  // :70-73
#pragma unroll
      for (int k = 0; k < EPL; ++k) out[k] = ((snorm == 0.0) ? 0.0 : alpha * grp.S[k]) - grp.XS[k];  // :74,:77
    } else {
      double root;
      const bool ok = binf_root<64>(grp, lam, sigma, delta, nullptr, root);
      const double sl = lam * sigma;
      if (!ok || (root - sl) == 0.0) {  // shiftedGroupNormL2Binf.jl:102-103, :107-108
#pragma unroll
        for (int k = 0; k < EPL; ++k) out[k] = 0.0 - grp.XS[k];
      } else {
        const double step = root / (sigma * (root - sl));  // :106
        double w[EPL], sw = 0.0;
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
          w[k] = grp.S[k] - sigma * softthres(grp.S[k] / sigma - step * grp.X[k], delta * step);  // :111
          sw += w[k] * w[k];
        }
        const double nw = sqrt(wave_sum(sw));
        const double alpha = jl_max(0.0, 1 - sl / nw);  // l2prox, :83
#pragma unroll
        for (int k = 0; k < EPL; ++k) out[k] = alpha * w[k] - grp.XS[k];  // :110-116
      }
    }
    if constexpr (EPL == 1) {
      y_[base + lane] = out[0];
    } else {
      f64x2* y2 = reinterpret_cast<f64x2*>(y_ + base);
#pragma unroll
      for (int k = 0; k < EPL / 2; ++k) y2[k * 64 + lane] = f64x2{out[2 * k], out[2 * k + 1]};
    }
  }
}

// ---------------------------------------------------------------------------------------------
// general kernel: any contiguous groups (CSR offsets or uniform size), TEAM lanes per group,
// elements re-read from memory for every reduction.
// ---------------------------------------------------------------------------------------------
template <int TEAM, bool BINF>
__global__ __launch_bounds__(256) void k_group_mem(double* y, const double* q, const double* xk, const double* sj,
                                                    int64_t n, const int64_t* __restrict__ offsets, int64_t gsize,
                                                    int64_t ngroups, const double* __restrict__ lambda, double sigma,
                                                    double delta) {
  __shared__ double lds[8];
  constexpr int TPB = 256 / TEAM;  // teams per block
  const int lane = threadIdx.x % TEAM;
  const int64_t team = (int64_t)blockIdx.x * TPB + threadIdx.x / TEAM;
  const int64_t nteams = (int64_t)gridDim.x * TPB;
  for (int64_t g = team; g < ngroups; g += nteams) {  // for TEAM == 256 the trip count is block-uniform
    int64_t lo, hi;
    if (offsets) { lo = offsets[g]; hi = offsets[g + 1]; }
    else { lo = g * gsize; hi = lo + gsize; }
    if (lo < 0) lo = 0;
    if (hi > n) hi = n;
    MemGroup<TEAM> grp{q, xk, sj, lo, hi, lane};
    const double lam = lambda[g];
    if constexpr (!BINF) {
      double ss = 0.0;
      grp.for_each([&](double S, double, int) { ss += S * S; });
      const double snorm = sqrt(team_sum<TEAM>(ss, lds));
      const double alpha = (snorm == 0.0) ? 0.0 : jl_max(1 - sigma * lam / snorm, 0.0);
      for (int64_t i = lo + lane; i < hi; i += TEAM) {
        double x = xk[i], s = sj[i];
        double S = (q[i] + x) + s;
        y[i] = ((snorm == 0.0) ? 0.0 : alpha * S) - (x + s);
      }
    } else {
      double root;
      const bool ok = binf_root<TEAM>(grp, lam, sigma, delta, lds, root);
      const double sl = lam * sigma;
      if (!ok || (root - sl) == 0.0) {
        for (int64_t i = lo + lane; i < hi; i += TEAM) y[i] = 0.0 - (xk[i] + sj[i]);
      } else {
        const double step = root / (sigma * (root - sl));
        double sw = 0.0;
        grp.for_each([&](double S, double X, int) {
          double w = S - sigma * softthres(S / sigma - step * X, delta * step);
          sw += w * w;
        });
        const double nw = sqrt(team_sum<TEAM>(sw, lds));
        const double alpha = jl_max(0.0, 1 - sl / nw);
        for (int64_t i = lo + lane; i < hi; i += TEAM) {
          double x = xk[i], s = sj[i];
          double S = (q[i] + x) + s;
          double w = S - sigma * softthres(S / sigma - step * x, delta * step);
          y[i] = alpha * w - (x + s);
        }
      }
    }
    if constexpr (TEAM == 256) __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
template <bool BINF>
static int run_group(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                     const int64_t* offsets, int64_t gsize, int64_t ngroups, const double* lambda, double sigma,
                     double delta) {
  int rc = spx_check_common(ctx, y, q, xk, sj, n);
  if (rc) return rc;
  SPX_REQUIRE(ngroups >= 0, "ngroups < 0");
  if (ngroups == 0 || n == 0) return SPX_OK;
  SPX_REQUIRE(lambda != nullptr, "lambda_vec is NULL");
  if (!offsets) {
    SPX_REQUIRE(gsize > 0, "group_size <= 0 with NULL group_offsets");
    SPX_REQUIRE(ngroups <= n / gsize && ngroups * gsize == n, "ngroups * group_size != n");
  }
  // y may alias q: every kernel finishes all reductions of a group (team barrier / wave lockstep) before
  // the group's first store, and the storing lane re-reads q[i] itself just before writing y[i].
  SPX_HIP(hipSetDevice(ctx->device));
  const int64_t cap_blocks = (int64_t)ctx->num_cu * 8;
  const bool aligned = spx_aligned16(y) && spx_aligned16(q) && spx_aligned16(xk) && spx_aligned16(sj);
  if (!offsets && (gsize % 64) == 0 && gsize / 64 <= 8 && (gsize == 64 || (gsize / 64) % 2 == 0) && aligned) {
    int64_t blocks = (ngroups + 3) / 4;  // 4 waves (groups) per 256-thread block
    if (blocks > cap_blocks) blocks = cap_blocks;
    dim3 grid((unsigned)blocks), block(256);
#define SPX_LAUNCH_REG(EPL)                                                                                       \
  hipLaunchKernelGGL((k_group_reg<EPL, BINF>), grid, block, 0, ctx->stream, y, q, xk, sj, ngroups, lambda, sigma, \
                     delta)
    switch (gsize / 64) {
      case 1: SPX_LAUNCH_REG(1); break;
      case 2: SPX_LAUNCH_REG(2); break;
      case 4: SPX_LAUNCH_REG(4); break;
      case 6: SPX_LAUNCH_REG(6); break;
      case 8: SPX_LAUNCH_REG(8); break;
    }
#undef SPX_LAUNCH_REG
    SPX_LAUNCH_CHECK();
    return SPX_OK;
  }
  // team width: wavefront per group unless groups are large on average
  const double avg = (double)n / (double)ngroups;
  if (avg <= 2048.0) {
    int64_t blocks = (ngroups + 3) / 4;
    if (blocks > cap_blocks) blocks = cap_blocks;
    hipLaunchKernelGGL((k_group_mem<64, BINF>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, y, q, xk, sj, n,
                       offsets, gsize, ngroups, lambda, sigma, delta);
  } else {
    int64_t blocks = ngroups < cap_blocks ? ngroups : cap_blocks;
    hipLaunchKernelGGL((k_group_mem<256, BINF>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, y, q, xk, sj, n,
                       offsets, gsize, ngroups, lambda, sigma, delta);
  }
  SPX_LAUNCH_CHECK();
  return SPX_OK;
}

SPX_EXPORT int spx_prox_group_l2(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                                 int64_t n, const int64_t* group_offsets, int64_t group_size, int64_t ngroups,
                                 const double* lambda_vec, double sigma) {
  return run_group<false>(ctx, y, q, xk, sj, n, group_offsets, group_size, ngroups, lambda_vec, sigma, 0.0);
}

SPX_EXPORT int spx_prox_group_l2_binf(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                                      int64_t n, const int64_t* group_offsets, int64_t group_size, int64_t ngroups,
                                      const double* lambda_vec, double sigma, double delta) {
  return run_group<true>(ctx, y, q, xk, sj, n, group_offsets, group_size, ngroups, lambda_vec, sigma, delta);
}
