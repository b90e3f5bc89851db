/*
 * spx_oracle_q.c -- extended-precision ARBITER for the floating-point operators of the prox!() hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rule as spx_oracle.c: only tests/, smoke() and bench.py's
 * cpu_baseline leg may load anything under oracle/).
 *
 * What it is: the same literal restatement of the reference's Julia bodies as spx_oracle.c, evaluated in IEEE
 * binary128 (__float128, libquadmath: 113-bit significand) instead of Float64.  It answers one question for the parity
 * tests: where the HIP result and the Float64 oracle differ by more than the 1e-12 bar, WHICH SIDE is further from the
 * exact value of the reference's formula?  A test may accept such a difference only if
 *
 *      |y_gpu - y_q|  <=  1e-12 * scale  +  |y_oracle64 - y_q|
 *
 * i.e. the GPU is within the bar of the reference's own double evaluation, or closer to the exact value than that
 * evaluation is (tests/arbiter.py).  Everything that is DATA to the formula stays what the reference holds in
 * Float64: the inputs, the stored sums sol = (q + xk) + sj / xk + sj (one rounded add each, bit-identical on the GPU),
 * the per-call / per-group scalar product sigma*lambda and the Float64 constant lmin = sigma*lambda*(1 + eps).  Everything the formula COMPUTES (norms, step, soft thresholds,
 * froot, the root itself, pow / acos / cos, candidate values) is carried in binary128.
 *
 * Roots.fzero(froot, lmin, lmax) (src/shiftedGroupNormL2Binf.jl:105): the bracket is sorted and bisected exactly as in
 * spx_oracle.c::orc_bisect (bit midpoints of the Float64 ends, so that a reversed bracket that straddles the pole of
 * step(n) is walked along the same path), with froot evaluated in binary128; once the two Float64 ends are adjacent the
 * interval is bisected further in binary128 (arithmetic midpoint) to 2^-110 relative width.
 *
 * Cited lines: see spx_oracle.c; every function here names the oracle function it shadows.
 */
#define _GNU_SOURCE
#include <math.h>
#include <quadmath.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))
typedef __float128 Q;

static inline Q q_sign(Q x) { return (x > 0) ? (Q)1 : (x < 0) ? (Q)-1 : x; }
static inline Q q_max(Q a, Q b) { return (a != a || b != b) ? a + b : (a > b ? a : b); }
static inline Q q_min(Q a, Q b) { return (a != a || b != b) ? a + b : (a < b ? a : b); }
static inline int is_selected(const uint8_t* mask, int64_t i) { return mask == NULL || mask[i] != 0; }

/* ------------------------------------------------------------------------------------------
 * ShiftedRootNormLhalf.prox!  (shadows orc_prox_lhalf; src/shiftedRootNormLhalf.jl:41-63)
 * ------------------------------------------------------------------------------------------ */
ORC_API void orcq_prox_lhalf(double* y, const double* q, const double* xk, const double* sj, int64_t n, double lambda,
                             double sigma) {
  const Q nl = (Q)(sigma * lambda); /* :47 -- one Float64 product per call: data */
  const Q p = powq(54, (Q)1 / 3) * powq(2 * nl, (Q)2 / 3) / 4; /* :49 */
  for (int64_t i = 0; i < n; ++i) {
    double xs = xk[i] + sj[i]; /* data: Float64 */
    double sol = q[i] + xs;    /* data: Float64 (:50) */
    Q a = fabsq((Q)sol), yi;
    if (a <= p) yi = 0;
    else {
      Q phi = acosq(nl / 4 * powq(a / 3, (Q)-3 / 2));                          /* :48 */
      yi = 2 * q_sign((Q)sol) / 3 * a * (1 + cosq(2 * M_PIq / 3 - 2 * phi / 3)); /* :57 */
    }
    y[i] = (double)(yi - (Q)xs);
  }
}

/* ShiftedRootNormLhalfBox.prox!  (shadows orc_prox_lhalf_box; src/shiftedRootNormLhalfBox.jl:86-120).
 * cand_out (optional): index 0..3 of the candidate findmin picks in binary128, -1 for unselected entries. */
static inline Q q_rnorm(Q tt, Q qi, Q sigma, Q lambda, Q xs) {
  Q d = tt - qi;
  return d * d / 2 / sigma + lambda * sqrtq(fabsq(tt + xs)); /* :95 */
}
ORC_API void orcq_prox_lhalf_box(double* y, const double* q, const double* xk, const double* sj, int64_t n,
                                 double lambda, double sigma, const double* lvec, const double* uvec, double lscal,
                                 double uscal, const uint8_t* mask, int8_t* cand_out) {
  const Q twopi3 = 2 * M_PIq / 3;
  const Q S = sigma, L = lambda;
  for (int64_t i = 0; i < n; ++i) {
    double li = lvec ? lvec[i] : lscal, ui = uvec ? uvec[i] : uscal;
    double xi = xk[i], si = sj[i], qi = q[i];
    if (!is_selected(mask, i)) {
      y[i] = (double)q_min(q_max((Q)qi, (Q)li - si), (Q)ui - si); /* :116 */
      if (cand_out) cand_out[i] = -1;
      continue;
    }
    double xs = xi + si;   /* data: ψ.sol (:94) */
    double xsq = xs + qi;  /* Float64 in the reference (:104) */
    Q a = (Q)(sigma * lambda) / 4 * powq(fabsq((Q)xsq) / 3, (Q)-3 / 2);
    __complex128 phi = cacosq(a + 0.0Q * 1.0i); /* :92 */
    __complex128 ang = (twopi3 - crealq(2 * phi / 3)) + (-cimagq(2 * phi / 3)) * 1.0i;
    Q val = (2 * q_sign((Q)xsq) / 3 * fabsq((Q)xsq)) * (1 + crealq(ccosq(ang))); /* :106 */
    Q t[4] = {(Q)li - si, (Q)ui - si, -(Q)xs, val - xs};
    Q c[4];
    c[0] = q_rnorm(t[0], qi, S, L, xs);
    c[1] = q_rnorm(t[1], qi, S, L, xs);
    Q mxi = -(Q)xi, vx = val - xi;
    c[2] = (li <= mxi && mxi <= ui) ? q_rnorm(t[2], qi, S, L, xs) : (Q)INFINITY;
    c[3] = (li <= vx && vx <= ui) ? q_rnorm(t[3], qi, S, L, xs) : (Q)INFINITY;
    int k = 0;
    for (int j = 1; j < 4; ++j)
      if (c[j] < c[k] || (c[j] != c[j] && c[k] == c[k])) k = j; /* findmin: first minimum, NaN wins */
    y[i] = (double)t[k];
    if (cand_out) cand_out[i] = (int8_t)k;
  }
}

/* ------------------------------------------------------------------------------------------
 * Groups (contiguous ranges; CSR offsets or uniform gsize).  `which` (nwhich entries) lists the groups to
 * evaluate; y is written for those groups only.
 * ------------------------------------------------------------------------------------------ */
static inline void group_range(const int64_t* offsets, int64_t gsize, int64_t g, int64_t* lo, int64_t* hi) {
  if (offsets) { *lo = offsets[g]; *hi = offsets[g + 1]; }
  else { *lo = g * gsize; *hi = (g + 1) * gsize; }
}

/* ShiftedGroupNormL2.prox!  (shadows orc_prox_group_l2; src/shiftedGroupNormL2.jl:52-79) */
ORC_API void orcq_prox_group_l2(double* y, const double* q, const double* xk, const double* sj, int64_t n,
                                const int64_t* offsets, int64_t gsize, int64_t ngroups, const double* lambda,
                                double sigma, const int64_t* which, int64_t nwhich) {
  (void)n; (void)ngroups;
  for (int64_t k = 0; k < nwhich; ++k) {
    int64_t g = which[k], lo, hi;
    group_range(offsets, gsize, g, &lo, &hi);
    Q ss = 0;
    for (int64_t i = lo; i < hi; ++i) { double s = (q[i] + xk[i]) + sj[i]; ss += (Q)s * s; }
    Q snorm = sqrtq(ss);
    Q alpha = (snorm == 0) ? (Q)0 : q_max(1 - (Q)sigma * lambda[g] / snorm, 0);
    for (int64_t i = lo; i < hi; ++i) {
      double s = (q[i] + xk[i]) + sj[i];
      y[i] = (double)(alpha * s - ((Q)xk[i] + sj[i]));
    }
  }
}

typedef struct {
  const double* S;
  const double* X;
  int64_t m;
  Q sigma, sl, delta;
  Q* w;
} qctx;

static inline Q q_soft(Q x, Q a) { return q_sign(x) * q_max(0, fabsq(x) - a); }
static Q q_norm2(const Q* v, int64_t m) {
  Q s = 0;
  for (int64_t i = 0; i < m; ++i) s += v[i] * v[i];
  return sqrtq(s);
}
static Q q_froot(const qctx* c, Q nn) { /* :87-93 */
  Q step = nn / (c->sigma * (nn - c->sl));
  for (int64_t i = 0; i < c->m; ++i)
    c->w[i] = c->sigma * q_soft(c->S[i] / c->sigma - step * c->X[i], c->delta * step) - c->S[i];
  return nn - q_norm2(c->w, c->m);
}
static double bit_middle(double x, double y) { /* Roots' __middle, as in spx_oracle.c */
  uint64_t xi, yi;
  double ax = fabs(x), ay = fabs(y);
  memcpy(&xi, &ax, 8);
  memcpy(&yi, &ay, 8);
  uint64_t mid = (xi + yi) >> 1;
  double m;
  memcpy(&m, &mid, 8);
  double sg = (x + y > 0) ? 1.0 : (x + y < 0) ? -1.0 : 0.0;
  return sg * m;
}
/* sorted bracket [a, b] with values fa, fb (binary128); returns the root to ~2^-110 */
static Q q_bisect(const qctx* c, Q a, Q fa, Q b, Q fb) {
  if (a > b) { Q t = a; a = b; b = t; t = fa; fa = fb; fb = t; }
  if (fa == 0) return a;
  if (fb == 0) return b;
  /* phase 1: the Float64 bit-midpoint walk between the Float64 neighbours of the ends */
  double da = (double)a, db = (double)b;
  if ((Q)da < a) da = nextafter(da, INFINITY);
  if ((Q)db > b) db = nextafter(db, -INFINITY);
  if (da < db) {
    for (int it = 0; it < 4096; ++it) {
      double dm = bit_middle(da, db);
      if (!(da < dm && dm < db)) break;
      Q fm = q_froot(c, dm);
      if (fm == 0) return dm;
      if (q_sign(fa) * q_sign(fm) < 0) { db = dm; b = dm; fb = fm; }
      else { da = dm; a = dm; fa = fm; }
    }
  }
  /* phase 2: binary128 arithmetic midpoints */
  for (int it = 0; it < 130; ++it) {
    Q m = a + (b - a) / 2;
    if (!(a < m && m < b)) break;
    Q fm = q_froot(c, m);
    if (fm == 0) return m;
    if (q_sign(fa) * q_sign(fm) < 0) { b = m; fb = fm; }
    else { a = m; fa = fm; }
  }
  return (fabsq(fa) < fabsq(fb)) ? a : b;
}

/* One group of ShiftedGroupNormL2Binf.prox!  (shadows the loop body of orc_prox_group_l2_binf;
 * src/shiftedGroupNormL2Binf.jl:85-116).  S, X, xs = xk + sj: Float64 data of the group.  Returns the branch taken:
 * 0 = zeros by :102 (no sign change), 1 = root found, 2 = zeros by :107; *root = the root (as Float64) for branch 1. */
static int q_binf_group(double* y, const double* S, const double* X, const double* xsd, int64_t m, double lam,
                        double sigma, double delta, Q* w, double* root) {
  const double eps = 2.220446049250313e-16;
  Q sl = (Q)(lam * sigma); /* :85 -- one Float64 product per group: data (the pole of step(n) sits at this double) */
  qctx c = {S, X, m, sigma, sl, delta, w};
  double lmin_d = (lam * sigma) * (1 + eps); /* :94 -- a Float64 constant of the algorithm */
  Q lmin = lmin_d;
  Q fl = q_froot(&c, lmin);
  Q ansatz = lmin + 1; /* :97 */
  Q step = ansatz / (c.sigma * (ansatz - sl));
  Q ssS = 0, ssX = 0;
  for (int64_t i = 0; i < m; ++i) {
    w[i] = q_soft(S[i] / c.sigma - step * X[i], c.delta * step);
    ssS += (Q)S[i] * S[i];
    ssX += (Q)X[i] * X[i];
  }
  Q zlmax = q_norm2(w, m);
  Q lmax = sqrtq(ssS) + c.sigma * (zlmax + (Q)lam * sqrtq(ssX)); /* :100 with epsilon = 1 */
  Q fm = q_froot(&c, lmax);
  int branch;
  if (root) *root = NAN;
  if (fl * fm > 0) { /* :102 */
    for (int64_t i = 0; i < m; ++i) w[i] = 0;
    branch = 0;
  } else {
    Q nn = q_bisect(&c, lmin, fl, lmax, fm); /* :105 */
    if (root) *root = (double)nn;
    if (fabsq(nn - sl) == 0) { /* :107 */
      for (int64_t i = 0; i < m; ++i) w[i] = 0;
      branch = 2;
    } else {
      step = nn / (c.sigma * (nn - sl));
      for (int64_t i = 0; i < m; ++i) w[i] = S[i] - c.sigma * q_soft(S[i] / c.sigma - step * X[i], c.delta * step); /* :111 */
      Q nw = q_norm2(w, m);
      Q alpha = q_max(0, 1 - sl / nw); /* :83 */
      for (int64_t i = 0; i < m; ++i) w[i] = alpha * w[i];
      branch = 1;
    }
  }
  for (int64_t i = 0; i < m; ++i) y[i] = (double)(w[i] - (Q)xsd[i]); /* :116 */
  return branch;
}

/* ShiftedGroupNormL2Binf.prox! on the listed groups (shadows orc_prox_group_l2_binf).
 * branch_out / root_out (optional, nwhich entries each): see q_binf_group. */
ORC_API void orcq_prox_group_l2_binf(double* y, const double* q, const double* xk, const double* sj, int64_t n,
                                     const int64_t* offsets, int64_t gsize, int64_t ngroups, const double* lambda,
                                     double sigma, double delta, const int64_t* which, int64_t nwhich,
                                     int32_t* branch_out, double* root_out) {
  (void)n; (void)ngroups;
  int64_t mmax = 1;
  for (int64_t k = 0; k < nwhich; ++k) {
    int64_t lo, hi;
    group_range(offsets, gsize, which[k], &lo, &hi);
    if (hi - lo > mmax) mmax = hi - lo;
  }
  double* S = (double*)malloc((size_t)mmax * sizeof(double));
  double* xs = (double*)malloc((size_t)mmax * sizeof(double));
  Q* w = (Q*)malloc((size_t)mmax * sizeof(Q));
  for (int64_t k = 0; k < nwhich; ++k) {
    int64_t g = which[k], lo, hi;
    group_range(offsets, gsize, g, &lo, &hi);
    for (int64_t i = lo; i < hi; ++i) {
      S[i - lo] = (q[i] + xk[i]) + sj[i]; /* :80, Float64 data */
      xs[i - lo] = xk[i] + sj[i];
    }
    double root;
    int b = q_binf_group(y + lo, S, xk + lo, xs, hi - lo, lambda[g], sigma, delta, w, &root);
    if (branch_out) branch_out[k] = b;
    if (root_out) root_out[k] = root;
  }
  free(S); free(xs); free(w);
}

/* ------------------------------------------------------------------------------------------
 * ShiftedNormL1B2.prox!  (shadows orc_prox_l1_b2; src/shiftedNormL1B2.jl:50-67).  The root of froot is unique
 * (spx_oracle.c), so plain bisection in binary128 from the same expanding bracket.
 * ------------------------------------------------------------------------------------------ */
typedef struct { const double *q, *xk, *sj; int64_t n; Q ls, delta, chil; } qb2;
static Q qb2_norm(const qb2* c, Q scale) {
  Q ss = 0;
  for (int64_t i = 0; i < c->n; ++i) {
    double sq = c->sj[i] + c->q[i]; /* Float64 in the reference (:56 broadcasts over Float64 vectors) */
    Q p = q_min(q_max(-(Q)c->xk[i] * scale, (Q)sq - c->ls), (Q)sq + c->ls);
    ss += p * p;
  }
  return c->chil * sqrtq(ss);
}
ORC_API void orcq_prox_l1_b2(double* y, const double* q, const double* xk, const double* sj, int64_t n, double lambda,
                             double sigma, double delta, double chi_lambda) {
  qb2 c = {q, xk, sj, n, (Q)(lambda * sigma), delta, chi_lambda};
  Q chiy = qb2_norm(&c, 1);
  Q scale = 1, back = 1;
  if ((Q)delta <= chiy) {
    Q a = delta, fa = a - qb2_norm(&c, a / c.delta), eta = a;
    if (fa != 0) {
      Q b = 2 * a, fb = b - qb2_norm(&c, b / c.delta);
      for (int it = 0; it < 20000 && !(fb >= 0) && finiteq(b); ++it) { a = b; fa = fb; b = 2 * b; fb = b - qb2_norm(&c, b / c.delta); }  /* (fb == 0: the root, as in spx_oracle.c) */
      if (fb == 0) eta = b;
      else {
        for (int it = 0; it < 240; ++it) {
          Q m = a + (b - a) / 2;
          if (!(a < m && m < b)) break;
          Q fm = m - qb2_norm(&c, m / c.delta);
          if (fm == 0) { a = b = m; fa = fb = 0; break; }
          if (q_sign(fa) * q_sign(fm) < 0) { b = m; fb = fm; } else { a = m; fa = fm; }
        }
        eta = (fabsq(fa) < fabsq(fb)) ? a : b;
      }
    }
    scale = eta / c.delta;
    back = c.delta / eta;
  }
  for (int64_t i = 0; i < n; ++i) {
    double sq = sj[i] + q[i];
    Q p = q_min(q_max(-(Q)xk[i] * scale, (Q)sq - c.ls), (Q)sq + c.ls);
    y[i] = (double)(p * back - (Q)sj[i]);
  }
}

ORC_API int orcq_abi_version(void) { return 1; }
