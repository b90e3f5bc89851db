// spx_separable.hip -- the six separable prox! kernels (ShiftedNormL1/L0/RootNormLhalf and their Box
// forms).  One streaming skeleton, one functor per operator.
//
// HBM layout: q, xk, sj, y (and l, u when they are vectors) are plain contiguous fp64 vectors.
// Algorithmic traffic: 3 reads + 1 write = 32 B/element (48 B with vector bounds, +1 B with a mask).
// Roofline: HBM bandwidth (no reuse, no contraction -> MFMA not applicable).
//
// Skeleton: a workgroup of 256 lanes walks tiles of 256*UNROLL 16-byte pairs with a block stride.
// Within a tile every wave instruction touches 1 KiB contiguous (lane i -> base + 16*i), all
// 3*UNROLL loads of a tile are issued before the first use so each lane keeps 3*UNROLL*16 B in
// flight, results are written with 16-byte stores.  q[i] is read before y[i] is written by the same
// lane and no other lane touches index i, so y may alias q.
#include <cmath>

#include <limits>

#include "spx_common.hpp"

// ---------------------------------------------------------------------------------------------
// per-element operators.  Signature: (q, x, s, l, u, selected) -> y.  Unboxed ones ignore l, u, selected.
// Every expression keeps the reference's association (cited); the library is built with
// -ffp-contract=off so a*b+c never becomes an FMA.
// ---------------------------------------------------------------------------------------------
struct OpL1 {  // src/shiftedNormL1.jl:46-51
  double ls;   // lambda * sigma
  static constexpr bool kBox = false;
  static constexpr int kLdsKiB = 6;  // KiB per wave and input vector in the LDS-staged skeleton
  static constexpr int kNIn = 3;
  static constexpr bool kObj = false;     // input vectors besides bounds: q, xk, sj
  __device__ __forceinline__ double operator()(double q, double x, double s, double, double, bool) const {
    double t = (-x) - s;                          // :47  @. y = -xk - sj
    return jl_min(jl_max(t, q - ls), q + ls);     // :50
  }
};
struct OpL1Aliased {  // y === q in the reference: the broadcast at :47 overwrites q before :50 reads it
  static constexpr bool kBox = false;
  static constexpr int kLdsKiB = 6;  // KiB per wave and input vector in the LDS-staged skeleton
  static constexpr int kNIn = 3;
  static constexpr bool kObj = false;     // input vectors besides bounds: q, xk, sj
  __device__ __forceinline__ double operator()(double, double x, double s, double, double, bool) const {
    return (-x) - s;  // min(max(t, t - ls), t + ls) == t bit for bit whenever ls >= 0
  }
};
struct OpL0 {  // src/shiftedNormL0.jl:45-52
  double c;    // sqrt(2 * lambda * sigma)
  static constexpr bool kBox = false;
  static constexpr int kLdsKiB = 6;  // KiB per wave and input vector in the LDS-staged skeleton
  static constexpr int kNIn = 3;
  static constexpr bool kObj = false;     // input vectors besides bounds: q, xk, sj
  __device__ __forceinline__ double operator()(double q, double x, double s, double, double, bool) const {
    double xps = x + s;
    return (fabs(xps + q) <= c) ? -xps : q;
  }
};
struct OpL1Box {  // src/shiftedNormL1Box.jl:96-122
  double sl;      // sigma * lambda
  static constexpr bool kBox = true;
  static constexpr int kLdsKiB = 6;  // KiB per wave and input vector in the LDS-staged skeleton
  static constexpr int kNIn = 3;
  static constexpr bool kObj = false;     // input vectors besides bounds: q, xk, sj
  __device__ __forceinline__ double operator()(double q, double x, double s, double l, double u, bool sel) const {
    double xs = x + s;
    double xsq = xs + q;
    double t = (xsq <= -sl) ? (q + sl) : ((xsq >= sl) ? (q - sl) : -xs);  // :111-117
    t = sel ? t : q;                                                       // :121 prox_zero(qi, ...)
    return jl_min(jl_max(t, l - s), u - s);                                // :118
  }
};
struct OpL0Box {  // src/shiftedNormL0Box.jl:96-128
  double c;       // 2 * lambda * sigma
  static constexpr bool kBox = true;
  static constexpr int kLdsKiB = 6;  // KiB per wave and input vector in the LDS-staged skeleton
  static constexpr int kNIn = 3;
  static constexpr bool kObj = false;     // input vectors besides bounds: q, xk, sj
  __device__ __forceinline__ double operator()(double q, double x, double s, double l, double u, bool sel) const {
    double sq = s + q;
    double xs = x + s;
    double xsq = xs + q;
    double dl = l - sq, du = u - sq;
    double val_left = dl * dl + ((x == -l) ? 0.0 : c);   // :110
    double val_right = du * du + ((x == -u) ? 0.0 : c);  // :111
    double yi = (val_left < val_right) ? (l - s) : (u - s);  // :114
    double val_min = jl_min(val_left, val_right);
    double mx = -x;
    if (l <= mx && mx <= u) {  // :116
      double val_0 = xsq * xsq;
      yi = (val_0 < val_min) ? -xs : yi;
      val_min = jl_min(val_0, val_min);
    }
    if (l <= sq && sq <= u) {  // :121
      double val_xsq = (xsq == 0.0) ? 0.0 : c;
      yi = (val_xsq < val_min) ? q : yi;
    }
    return sel ? yi : prox_zero(q, l - s, u - s);  // :127
  }
};

// ---------------------------------------------------------------------------------------------
// RootNormLhalf closed form.  Reference (src/shiftedRootNormLhalf.jl:48,57; shiftedRootNormLhalfBox.jl:92,106):
//   val = (2/3) sign(z) |z| (1 + cos(2 pi/3 - (2/3) acos(a))),   a = (sigma lambda / 4) (|z|/3)^(-3/2),  a <= 1.
// With phi = acos(-a) = pi - acos(a) and w = cos(phi/3) in [1/2, sqrt(3)/2]:  cos(2 pi/3 - (2/3) acos a) = 2 w^2 - 1,
// so  val = sign(z) * 4 t w^2  with t = |z|/3, and w is the largest root of 4 w^3 - 3 w + a = 0.
// Writing w = 1/2 + d:  4 d^3 + 6 d^2 = m := 1 - a  (d ~ sqrt(m/6) near the threshold a = 1).
// d is obtained by 2 Newton steps on g(d) = 4 d^3 + 6 d^2 - m from a cubic fit d0 = e P(e), e = sqrt(m)
// (relative error of the fit 5.4e-5 on [0, 1]; Newton squares it: 1e-9, 1e-18).  The divisions by
// g'(d) = 12 d (d + 1) use the f32 reciprocal: Newton is self-correcting, a 1e-7 error in the slope costs
// 1e-7 * |last correction| <= 1e-16.  a itself comes from an fp64 rsqrt (hardware seed + 2 Newton steps).
// Accuracy vs an 80-bit evaluation of the reference formula: <= 6e-16 |z| everywhere, the same as the
// reference's own double evaluation (tools/lhalf_proto.py); agreement with the CPU restatement is checked to
// 1e-12 in tests/.  No pow / acos / cos calls: ~55 fp64 VALU ops instead of ~250.
// ---------------------------------------------------------------------------------------------
#ifndef SPX_LHALF_NEWTON
#define SPX_LHALF_NEWTON 2  // fit 5.4e-5 -> 1.5e-9 -> 1e-18 (tools/lhalf_proto.py: 2 and 3 steps give identical errors)
#endif
__device__ __forceinline__ double rsqrt_f64(double t) {
  double y = __builtin_amdgcn_rsq(t);            // v_rsq_f64 seed
  const double h = 0.5 * t;
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    double e = __builtin_fma(-h * y, y, 0.5);    // 1/2 - t y^2 / 2
    y = __builtin_fma(y, e, y);
  }
  return y;
}
// sqrt(v) for v >= 0: coupled Goldschmidt step on (s, h) = (v y, y / 2) from the hardware rsq seed, then one
// residual correction (faithfully rounded; 8 VALU ops)
__device__ __forceinline__ double sqrt_f64(double v) {
  const double y = __builtin_amdgcn_rsq(v);
  double s = v * y;
  double h = 0.5 * y;
  const double r = __builtin_fma(-s, h, 0.5);
  s = __builtin_fma(s, r, s);
  h = __builtin_fma(h, r, h);
  s = __builtin_fma(__builtin_fma(-s, s, v), h, s);
  return (v > 0.0) ? s : 0.0;
}
// a = sl4 * (az/3)^(-3/2) and t = az/3
__device__ __forceinline__ double lhalf_a(double az, double sl4, double& t, double* r_out = nullptr) {
  t = az * 0.33333333333333331;
  const double r = rsqrt_f64(t);
  if (r_out) *r_out = r;
  return sl4 * (r * r * r);
}
// sign(z) * 4 t w^2 for a in [0, 1]
__device__ __forceinline__ double lhalf_val_from_a(double z, double t, double a, double* w_out = nullptr) {
  const double m = fmax(1.0 - a, 0.0);
  const double e = (double)__builtin_amdgcn_sqrtf((float)m);
  double d = e * __builtin_fma(e, __builtin_fma(e, __builtin_fma(e, -0.003676457097091405, 0.016543212998449176),
                                                 -0.05508522799226874), 0.4082262283672896);
#pragma unroll
  for (int k = 0; k < SPX_LHALF_NEWTON; ++k) {
    const double g = __builtin_fma(__builtin_fma(4.0, d, 6.0) * d, d, -m);
    const double gp = 12.0 * d * (d + 1.0);
    const double inv = (gp > 0.0) ? (double)__builtin_amdgcn_rcpf((float)gp) : 0.0;
    d = __builtin_fma(-g, inv, d);
  }
  const double w = 0.5 + d;
  if (w_out) *w_out = w;
  const double v = 4.0 * t * (w * w);
  return (z < 0.0) ? -v : v;
}
struct OpLhalf {  // src/shiftedRootNormLhalf.jl:47-60
  double sl4;     // (sigma * lambda) / 4
  double p;       // 54^(1/3) * (2 sigma lambda)^(2/3) / 4
  static constexpr bool kBox = false;
  static constexpr int kLdsKiB = 6;  // KiB per wave and input vector in the LDS-staged skeleton
  static constexpr int kNIn = 3;
  static constexpr bool kObj = false;     // input vectors besides bounds: q, xk, sj
  __device__ __forceinline__ double operator()(double q, double x, double s, double, double, bool) const {
    double xs = x + s;
    double sol = q + xs;  // :50
    double aq = fabs(sol);
    double t;
    double a = lhalf_a(aq, sl4, t);
    double val = lhalf_val_from_a(sol, t, fmin(a, 1.0));
    double yi = (aq <= p) ? 0.0 : val;  // :53-57
    return yi - xs;                     // :59
  }
};
struct OpLhalfBox {  // src/shiftedRootNormLhalfBox.jl:92-117
  double sl4;        // sigma * lambda / 4
  double lambda;
  double h2;         // 1 / (2 sigma)
  static constexpr bool kBox = true;
  static constexpr int kLdsKiB = 0;  // 0: register-staged skeleton (VALU-heavy: needs the occupancy; 5.97 vs 5.68 TB/s)
  static constexpr int kNIn = 3;
  static constexpr bool kObj = false;
  // RNorm(tt) = (tt - q)^2 / 2 / sigma + lambda sqrt|tt + xs|   (:95); used only to pick the argmin
  __device__ __forceinline__ double rnorm(double tt, double q, double xs) const {
    double d = tt - q;
    return __builtin_fma(lambda, sqrt_f64(fabs(tt + xs)), d * d * h2);
  }
  // the same for a bound candidate, where tt may be +-Inf (one-sided / absent bounds): sqrt_f64 is an rsq iteration and
  // turns Inf into NaN, so its argument is capped -- the (tt - q)^2 term is Inf there anyway and so is the sum
  __device__ __forceinline__ double rnorm_bound(double tt, double q, double xs) const {
    double d = tt - q;
    return __builtin_fma(lambda, sqrt_f64(fmin(fabs(tt + xs), 1.0e300)), d * d * h2);
  }
  __device__ __forceinline__ double operator()(double q, double x, double s, double l, double u, bool sel) const {
    double xs = x + s;  // :94
    double xsq = xs + q;
    double axsq = fabs(xsq);
    double tl = l - s, tu = u - s;
    // candidates 1..3 (:109-111); findmin keeps the FIRST minimum -> a later one replaces only on strict <
    double best = rnorm_bound(tl, q, xs);
    double yi = tl;
    double c2 = rnorm_bound(tu, q, xs);
    if (c2 < best) { best = c2; yi = tu; }
    double mx = -x;
    double c3 = xsq * xsq * h2;  // RNorm(-xs): (-xs - q)^2 = xsq^2 and sqrt|(-xs) + xs| = 0 exactly
    if (l <= mx && mx <= u && c3 < best) { best = c3; yi = -xs; }
    // candidate 4 (:106,:112): the stationary point.  When the acos argument a exceeds 1 the reference
    // takes the real part of a complex expression, which is not a stationary point; the objective is
    // then V-shaped around v = 0 so that candidate can never be strictly smaller than the first three
    //  and is skipped here.  a is Inf/NaN for xsq == 0 -> skipped too, as in the
    // reference (`li <= NaN <= ui` is false).
    double t, r, w;
    double a = lhalf_a(axsq, sl4, t, &r);
    if (a <= 1.0) {
      double val = lhalf_val_from_a(xsq, t, a, &w);
      double vx = val - x;
      double t4 = val - xs;
#ifdef SPX_LHALFBOX_SQRT_T4
      double c4 = rnorm(t4, q, xs);
#else
      // RNorm(t4) = (t4 - q)^2 / 2 sigma + lambda sqrt|t4 + xs| with |t4 + xs| = |val| = 4 t w^2 up to rounding:
      // sqrt|val| = 2 sqrt(t) w = 2 (t r) w from the reciprocal square root already at hand -- one square root less.
      // (The value only ranks the candidates; the other three keep the faithfully rounded square root.)
      const double d4 = t4 - q;
      double c4 = __builtin_fma(lambda, 2.0 * (t * r) * w, d4 * d4 * h2);
#endif
      if (l <= vx && vx <= u && c4 < best) { best = c4; yi = t4; }
    }
    // NaN in q, xk or sj: every candidate value is NaN and findmin returns the first one (:114, `findmin` treats NaN as
    // the smallest value) -> candidate 1
    yi = (xsq != xsq) ? tl : yi;
    return sel ? yi : prox_zero(q, tl, tu);  // :116
  }
};

// ---------------------------------------------------------------------------------------------
// iprox!  (SURVEY.md 8f rank 1): argmin 1/2 y'Dy + g'y + psi(y), D = diag(d).  Four input vectors (g, d, xk, sj):
// 40 B/element.  Only +, -, *, /, sqrt and comparisons: bit-exact against the reference formulas.
// ---------------------------------------------------------------------------------------------
// iprox_zero(d, g, l, u)   src/ShiftedProximalOperators.jl:217-236
__device__ __forceinline__ double iprox_zero(double d, double g, double l, double u) {
  const double eps = 2.220446049250313e-16;
  const double a = jl_min(jl_max(-g / d, l), u);                       // d > eps
  const double d_2 = d / 2;
  const double b = ((d_2 * (l * l) + g * l) < (d_2 * (u * u) + g * u)) ? l : u;  // d < -eps
  const double c = (g > 0.0) ? l : ((g < 0.0) ? u : 0.0);              // |d| <= eps
  return (d > eps) ? a : ((d < -eps) ? b : c);
}
struct OpIproxL1 {  // src/shiftedNormL1.jl:60-75
  double lambda;
  int* flag;  // set when some d[i] <= 0 (the reference's `@assert d[i] > 0`)
  static constexpr bool kBox = false;
  static constexpr int kLdsKiB = 4;
  static constexpr int kNIn = 4;
  static constexpr bool kObj = false;
  __device__ __forceinline__ double call4(double g, double d, double x, double s, double, double, bool) const {
    // (raised once: a d that is wrong everywhere must not queue 1e8 atomics on one address -- 12 ns apiece)
    if (!(d > 0.0) && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(flag, 1);
    const double t = (-x) - s;                                                       // :67
    return jl_min(jl_max(t, -g / d - lambda / d), -g / d + lambda / d);              // :71
  }
};
struct OpIproxL0 {  // src/shiftedNormL0.jl:61-80
  double lambda;
  int* flag;
  static constexpr bool kBox = false;
  static constexpr int kLdsKiB = 4;
  static constexpr int kNIn = 4;
  static constexpr bool kObj = false;
  __device__ __forceinline__ double call4(double g, double d, double x, double s, double, double, bool) const {
    // (raised once: a d that is wrong everywhere must not queue 1e8 atomics on one address -- 12 ns apiece)
    if (!(d > 0.0) && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(flag, 1);
    const double ci = sqrt(2 * lambda * d);                                          // :71
    const double xps = x + s;
    return (fabs(d * xps - g) <= ci) ? -xps : (-g / d);                              // :73-77
  }
};
struct OpIproxL1Box {  // src/shiftedNormL1Box.jl:131-225
  double lambda;
  static constexpr bool kBox = true;
  static constexpr int kLdsKiB = 0;  // register-staged: five fp64 divisions per element want the occupancy (6.10 vs 5.70 TB/s)
  static constexpr int kNIn = 4;
  static constexpr bool kObj = false;
  __device__ __forceinline__ double call4(double g, double d, double x, double s, double l, double u, bool sel) const {
    const double eps = 2.220446049250313e-16;
    const double xs = x + s;
    const double left = l - s, right = u - s;
    double yi;
    if (fabs(d) <= eps) {  // :152
      yi = (fabs(g) <= lambda) ? jl_min(jl_max(left, -xs), right) : ((g > 0) ? left : right);
    } else {
      const double d_2 = d / 2;
      const double lx = l + x, ux = u + x;
      const double g2_d = g / d_2;
      const double f2_d = g2_d - 2 * xs;
      const double l2_d = lambda / d_2;
      const double val_left = lx * lx + f2_d * lx + l2_d * fabs(lx);
      const double val_right = ux * ux + f2_d * ux + l2_d * fabs(ux);
      if (d > eps) {  // :161
        double val_min = jl_min(val_left, val_right);
        yi = (val_left < val_right) ? left : right;
        const double y1 = -(g + lambda) / d;
        const double y2 = (lambda - g) / d;
        if (lx >= 0.0) {
          if (left <= y1 && y1 <= right) yi = y1;
        } else if (0.0 >= ux) {
          if (left <= y2 && y2 <= right) yi = y2;
        } else {
          if (left <= y1 && y1 <= right) {
            const double v1 = xs + y1;
            const double q1 = v1 * v1 + f2_d * v1 + l2_d * fabs(v1);
            if (q1 < val_min) yi = y1;
            val_min = jl_min(q1, val_min);
          }
          if (left <= y2 && y2 <= right) {
            const double v2 = xs + y2;
            const double q2 = v2 * v2 + f2_d * v2 + l2_d * fabs(v2);
            if (q2 < val_min) yi = y2;
            val_min = jl_min(q2, val_min);
          }
          if (0.0 < val_min) yi = -xs;  // val_0 = 0
        }
      } else {  // d <= -eps, :199
        const double val_max = jl_max(val_left, val_right);
        yi = (val_left > val_right) ? left : right;
        const double mx = -x;
        if (l <= mx && mx <= u && 0.0 > val_max) yi = -xs;
      }
    }
    return sel ? yi : iprox_zero(d, g, left, right);  // :221
  }
};
struct OpIproxL0Box {  // src/shiftedNormL0Box.jl:137-231
  double lambda;
  static constexpr bool kBox = true;
  static constexpr int kLdsKiB = 4;
  static constexpr int kNIn = 4;
  static constexpr bool kObj = false;
  __device__ __forceinline__ double call4(double g, double d, double x, double s, double l, double u, bool sel) const {
    const double eps = 2.220446049250313e-16;
    const double xs = x + s;
    const double mx = -x;
    const bool zero_ok = (l <= mx && mx <= u);
    const double left = l - s, right = u - s;
    double yi;
    if (fabs(d) < eps) {  // :154
      if (g == 0.0) {
        yi = zero_ok ? -xs : 0.0;
      } else {
        const bool pos = g > 0.0;
        const double t = pos ? left : right;
        const double val_min = g * t + ((x == (pos ? -l : -u)) ? 0.0 : lambda);
        yi = t;
        if (zero_ok && (-g * xs) < val_min) yi = -xs;
      }
    } else {
      const double d_2 = d / 2;
      const double lx = l + x, ux = u + x;
      const double g2_d = g / d_2;
      const double f2_d = g2_d - 2 * xs;
      const double l2_d = lambda / d_2;
      const double val_left = (lx == 0.0) ? 0.0 : (lx * lx + f2_d * lx + l2_d);
      const double val_right = (ux == 0.0) ? 0.0 : (ux * ux + f2_d * ux + l2_d);
      if (d >= eps) {  // :190
        const double aqy = -g / d;
        const double aqv = aqy + xs;
        double val_min;
        if (lx <= aqv && aqv <= ux) {
          val_min = (aqv == 0.0) ? (-(aqv * aqv)) : (-(aqv * aqv) + l2_d);
          yi = aqy;
        } else {
          yi = (val_left < val_right) ? left : right;
          val_min = jl_min(val_left, val_right);
        }
        if (zero_ok && 0.0 < val_min) yi = -xs;
      } else {  // :213
        yi = (val_left > val_right) ? left : right;
        const double val_max = jl_max(val_left, val_right);
        if (zero_ok && 0.0 > val_max) yi = -xs;
      }
    }
    return sel ? yi : iprox_zero(d, g, left, right);  // :227
  }
};

// ---------------------------------------------------------------------------------------------
// prox! fused with the value of h at the result ("fused with prox where possible"): the kernels
// below add Term((xk + sj) + y) over the selected indices into one partial per wavefront / workgroup
// (src/ShiftedProximalOperators.jl:51-54 for the association), a second small kernel adds the partials in index order.
// ---------------------------------------------------------------------------------------------
struct HTermL1 { __device__ __forceinline__ double operator()(double v) const { return fabs(v); } };               // NormL1 [ext]
struct HTermL0 { __device__ __forceinline__ double operator()(double v) const { return (v != 0.0) ? 1.0 : 0.0; } };  // NormL0 [ext]
struct HTermLhalf { __device__ __forceinline__ double operator()(double v) const { return sqrt(fabs(v)); } };      // src/rootNormLhalf.jl:27-29
template <class Base, class Term>
struct WithValue : Base {
  static constexpr bool kObj = true;
  double* partials;  // one slot per wavefront (LDS skeleton) / workgroup (register skeleton, scalar kernel)
  double qscale;     // the prox is taken at qscale * q (R2: q = -nu * grad f, formed on the fly; 1.0 = q itself)
  // Round 4 (value_publish below): when the call is ONE launch of at most kValueFuseMax workgroups, the workgroup that
  // finishes last adds the partials itself and the k_value_reduce launch is not queued (fin_hdr != NULL)
  SpxSyncHeader* fin_hdr = nullptr;
  double* fin_result = nullptr;  // the library's result slot (read back by the host form)
  double* fin_target = nullptr;  // spx_ctx::value_target (may be NULL)
  double fin_scale = 1.0;
  __device__ __forceinline__ double operator()(double q, double x, double s, double l, double u, bool sel) const {
    return Base::operator()(qscale * q, x, s, l, u, sel);
  }
  __device__ __forceinline__ double hterm(double x, double s, double y, bool sel) const {
    return sel ? Term{}((x + s) + y) : 0.0;
  }
};
__device__ __forceinline__ double block_sum4(double v, double* lds4) {  // 256-lane workgroup
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds4[threadIdx.x >> 6] = v;
  __syncthreads();
  return (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
}
// partials[0..count), count <= kValueFuseMax, added by the first 256 lanes of a workgroup in a fixed order (eight loads in
// flight per lane, a fixed tree, wavefront butterflies, the four wavefronts in order): the order of BOTH the one-launch form
// (ATOMIC: the slots were written by other workgroups of the same launch) and k_value_reduce on short lists, so that the two
// forms of a call give the same bits.  Every lane of the workgroup must call it.
constexpr int kValueFuseMax = 2048;
template <bool ATOMIC>
__device__ __forceinline__ double value_reduce_small(const double* partials, int count) {
  __shared__ double vr_lds4[4];
  const int t = threadIdx.x;
  double v8[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int i = t + k * 256;
    const bool in = t < 256 && i < count;
    if constexpr (ATOMIC) v8[k] = in ? spx_atomic_load_f64(partials + i) : 0.0;
    else v8[k] = in ? partials[i] : 0.0;
  }
  double acc = ((v8[0] + v8[1]) + (v8[2] + v8[3])) + ((v8[4] + v8[5]) + (v8[6] + v8[7]));
  acc = wave_sum(acc);
  __syncthreads();
  if ((t & 63) == 0 && t < 256) vr_lds4[t >> 6] = acc;
  __syncthreads();
  return (vr_lds4[0] + vr_lds4[1]) + (vr_lds4[2] + vr_lds4[3]);
}
// The partial sum `t` of this workgroup (valid in thread 0) goes to its slot; in the one-launch form the workgroup that takes
// the last ticket (spx_fin_ticket) then adds all of them and stores the value.  Every lane of the workgroup must call it.
template <class Op>
__device__ __forceinline__ void value_publish(const Op& op, int64_t slot, double t) {
  if (op.fin_hdr == nullptr) {
    if (threadIdx.x == 0) op.partials[slot] = t;
    return;
  }
  __shared__ int vp_last;
  if (threadIdx.x == 0) {
    spx_atomic_store_f64(op.partials + slot, t);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    vp_last = spx_fin_ticket(op.fin_hdr) ? 1 : 0;
  }
  __syncthreads();
  if (!vp_last) return;
  const double sum = value_reduce_small<true>(op.partials, (int)gridDim.x);
  if (threadIdx.x == 0) {
    *op.fin_result = sum;
    if (op.fin_target) *op.fin_target = op.fin_scale * sum;
  }
}

// partials[0..count) -> *out, fixed order: reproducible run to run
// target != NULL: also *target = scale * sum (the caller's device double, spx_ctx_set_value_target)
__global__ __launch_bounds__(1024) void k_value_reduce(const double* partials, int64_t count, double* out, double scale,
                                                        double* target) {
  __shared__ double lds[16];
  if (count <= kValueFuseMax) {  // (the order of the one-launch form)
    const double t = value_reduce_small<false>(partials, (int)count);
    if (threadIdx.x == 0) {
      *out = t;
      if (target) *target = scale * t;
    }
    return;
  }
  // eight independent loads in flight per lane, added in a fixed order (reproducible run to run)
  double a8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
  for (int64_t i0 = threadIdx.x; i0 < count; i0 += 8 * 1024) {
    double v8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int64_t i = i0 + (int64_t)k * 1024;
      v8[k] = (i < count) ? partials[i < count ? i : 0] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) a8[k] += v8[k];
  }
  double acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) lds[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 16; ++w) t += lds[w];
    *out = t;
    if (target) *target = scale * t;
  }
}

// uniform call: 3-input operators ignore d
template <class Op>
__device__ __forceinline__ double apply_op(const Op& op, double q, double d, double x, double s, double l, double u,
                                           bool sel) {
  if constexpr (Op::kNIn == 4) return op.call4(q, d, x, s, l, u, sel);
  else return op(q, x, s, l, u, sel);
}

// ---------------------------------------------------------------------------------------------
// streaming skeleton
// ---------------------------------------------------------------------------------------------
template <bool NT>
__device__ __forceinline__ f64x2 ld2(const f64x2* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT>
__device__ __forceinline__ void st2(f64x2* p, f64x2 v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// n2 = number of 16-byte pairs.  VECB: l/u are vectors.  MASK: sel mask present.
template <class Op, int UNROLL, bool VECB, bool MASK, bool NT>
__global__ __launch_bounds__(256) void k_sep_vec(double* y_, const double* q_, const double* d_, const double* xk_,
                                                  const double* sj_, const double* l_, const double* u_,
                                                  const uint8_t* mask_, double ls, double us, int64_t n2, Op op) {
  const f64x2* dv = reinterpret_cast<const f64x2*>(d_);
  constexpr int64_t TILE = 256 * UNROLL;
  f64x2* y = reinterpret_cast<f64x2*>(y_);
  const f64x2* q = reinterpret_cast<const f64x2*>(q_);
  const f64x2* xk = reinterpret_cast<const f64x2*>(xk_);
  const f64x2* sj = reinterpret_cast<const f64x2*>(sj_);
  const f64x2* lv = reinterpret_cast<const f64x2*>(l_);
  const f64x2* uv = reinterpret_cast<const f64x2*>(u_);
  const uint16_t* mk = reinterpret_cast<const uint16_t*>(mask_);
  const int64_t ntiles = (n2 + TILE - 1) / TILE;
  double hacc = 0.0;
  for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int64_t base = tile * TILE + threadIdx.x;
    f64x2 vq[UNROLL], vx[UNROLL], vs[UNROLL], vl[UNROLL], vu[UNROLL], vd[UNROLL];
    uint16_t vm[UNROLL];
    if (base - threadIdx.x + TILE <= n2) {
#pragma unroll
      for (int k = 0; k < UNROLL; ++k) {
        const int64_t i = base + k * 256;
        vq[k] = ld2<NT>(q + i);
        if constexpr (Op::kNIn == 4) vd[k] = ld2<NT>(dv + i); else vd[k] = f64x2{0.0, 0.0};
        vx[k] = ld2<NT>(xk + i);
        vs[k] = ld2<NT>(sj + i);
        if constexpr (VECB && Op::kBox) {
          vl[k] = l_ ? ld2<NT>(lv + i) : f64x2{ls, ls};
          vu[k] = u_ ? ld2<NT>(uv + i) : f64x2{us, us};
        }
        if constexpr (MASK && Op::kBox) vm[k] = mk[i];
      }
#pragma unroll
      for (int k = 0; k < UNROLL; ++k) {
        const int64_t i = base + k * 256;
        double l0 = ls, l1 = ls, u0 = us, u1 = us;
        bool s0 = true, s1 = true;
        if constexpr (VECB && Op::kBox) { l0 = vl[k].x; l1 = vl[k].y; u0 = vu[k].x; u1 = vu[k].y; }
        if constexpr (MASK && Op::kBox) { s0 = (vm[k] & 0xff) != 0; s1 = (vm[k] >> 8) != 0; }
        f64x2 r;
        r.x = apply_op(op, vq[k].x, vd[k].x, vx[k].x, vs[k].x, l0, u0, s0);
        r.y = apply_op(op, vq[k].y, vd[k].y, vx[k].y, vs[k].y, l1, u1, s1);
        if constexpr (Op::kObj) hacc += op.hterm(vx[k].x, vs[k].x, r.x, s0) + op.hterm(vx[k].y, vs[k].y, r.y, s1);
        st2<NT>(y + i, r);
      }
    } else {  // last, partial tile
#pragma unroll
      for (int k = 0; k < UNROLL; ++k) {
        const int64_t i = base + k * 256;
        if (i < n2) {
          f64x2 a = q[i], b = xk[i], c = sj[i];
          f64x2 dd = f64x2{0.0, 0.0};
          if constexpr (Op::kNIn == 4) dd = dv[i];
          double l0 = ls, l1 = ls, u0 = us, u1 = us;
          bool s0 = true, s1 = true;
          if constexpr (VECB && Op::kBox) {
            if (l_) { f64x2 t = lv[i]; l0 = t.x; l1 = t.y; }
            if (u_) { f64x2 t = uv[i]; u0 = t.x; u1 = t.y; }
          }
          if constexpr (MASK && Op::kBox) { uint16_t m = mk[i]; s0 = (m & 0xff) != 0; s1 = (m >> 8) != 0; }
          f64x2 r;
          r.x = apply_op(op, a.x, dd.x, b.x, c.x, l0, u0, s0);
          r.y = apply_op(op, a.y, dd.y, b.y, c.y, l1, u1, s1);
          if constexpr (Op::kObj) hacc += op.hterm(b.x, c.x, r.x, s0) + op.hterm(b.y, c.y, r.y, s1);
          y[i] = r;
        }
      }
    }
  }
  if constexpr (Op::kObj) {
    __shared__ double lds4[4];
    const double t = block_sum4(hacc, lds4);
    value_publish(op, blockIdx.x, t);
  }
}

// ---------------------------------------------------------------------------------------------
// LDS-staged skeleton (the default): each wavefront owns a contiguous chunk of UNROLL KiB per input vector and
// stages it through LDS with `global_load_lds_dwordx4 ... nt` -- LDS-DMA: no VGPR destination, one contiguous
// 1 KiB piece per wave instruction -- then waits once (vmcnt(0)), reads its own 16 bytes per piece back with
// ds_read_b128 (lane-private slots: conflict-free, no barrier: nothing is shared between waves), evaluates
// the operator and writes y with non-temporal 16-byte stores.  3*UNROLL KiB are in flight per wave without
// holding registers; UNROLL = 6 -> 72 KiB of LDS per 4-wave workgroup -> two workgroups (144 KiB in flight)
// per CU.  Measured on MI355X (tools/exp/exp_stream.hip, n = 1e8, same process, interleaved rounds):
// 6.45-6.57 TB/s against 6.11-6.30 TB/s for the register-staged form above; 7 KiB per wave and more (one
// workgroup per CU) or a piece-by-piece pipelined wait lose 10-20 %.
// All loads of a wave have landed before its first store, and waves touch disjoint ranges: y may alias q.
// ---------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void lds_void;
__device__ __forceinline__ void dma16_nt(const f64x2* g, char* wave_lds_piece) {
  __builtin_amdgcn_global_load_lds((const void*)g, (lds_void*)wave_lds_piece, 16, 0, 2);  // aux = 2: nt
}

template <class Op, int UNROLL, bool VECB, bool MASK>
__global__ __launch_bounds__(256) void k_sep_lds(double* y_, const double* q_, const double* d_, const double* xk_,
                                                  const double* sj_, const double* l_, const double* u_,
                                                  const uint8_t* mask_, double ls, double us, int64_t n2, Op op,
                                                  int64_t xcd_chunk) {
  constexpr int NARR = Op::kNIn + ((VECB && Op::kBox) ? 2 : 0);
  const f64x2* dv = reinterpret_cast<const f64x2*>(d_);
  __shared__ __attribute__((aligned(16))) char lds[4 * NARR * UNROLL * 1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  char* wl = lds + wave * (NARR * UNROLL * 1024);
  f64x2* y = reinterpret_cast<f64x2*>(y_);
  const f64x2* q = reinterpret_cast<const f64x2*>(q_);
  const f64x2* xk = reinterpret_cast<const f64x2*>(xk_);
  const f64x2* sj = reinterpret_cast<const f64x2*>(sj_);
  const f64x2* lv = reinterpret_cast<const f64x2*>(l_);
  const f64x2* uv = reinterpret_cast<const f64x2*>(u_);
  const uint16_t* mk = reinterpret_cast<const uint16_t*>(mask_);
  // Workgroups are dealt to the 8 XCDs round-robin.  xcd_chunk == 0 (default): tile = workgroup id, so neighbouring tiles
  // land on different XCDs; xcd_chunk > 0 (spx_ctx_set_tuning key 5, experiment): XCD x works through the contiguous range
  // [x * xcd_chunk, (x + 1) * xcd_chunk) of tiles.  No tile is ever re-read, so neither L2 has anything to win -- measured.
  int64_t bid = blockIdx.x;
  if (xcd_chunk > 0) {
    bid = (int64_t)(blockIdx.x & 7) * xcd_chunk + (blockIdx.x >> 3);
    if (bid * (256 * UNROLL) >= n2) {
      if constexpr (Op::kObj) {
        if (threadIdx.x == 0) op.partials[bid] = 0.0;  // (a slot of its own: every slot the host counts is written)
      }
      return;
    }
  }
  const int64_t base = (bid * 4 + wave) * (64 * UNROLL) + lane;  // this lane's first pair
  double hacc = 0.0;
  uint16_t vm[UNROLL];
#pragma unroll
  for (int k = 0; k < UNROLL; ++k) {
    int64_t i = base + k * 64;
    if (i >= n2) i = n2 - 1;  // tail: load something valid, the store is guarded
    dma16_nt(q + i, wl + (0 * UNROLL + k) * 1024);
    dma16_nt(xk + i, wl + (1 * UNROLL + k) * 1024);
    dma16_nt(sj + i, wl + (2 * UNROLL + k) * 1024);
    if constexpr (Op::kNIn == 4) dma16_nt(dv + i, wl + (3 * UNROLL + k) * 1024);
    if constexpr (VECB && Op::kBox) {
      if (l_) dma16_nt(lv + i, wl + ((Op::kNIn + 0) * UNROLL + k) * 1024);
      if (u_) dma16_nt(uv + i, wl + ((Op::kNIn + 1) * UNROLL + k) * 1024);
    }
    if constexpr (MASK && Op::kBox) vm[k] = mk[i];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every piece of this wave has landed in LDS
#pragma unroll
  for (int k = 0; k < UNROLL; ++k) {
    const int64_t i = base + k * 64;
    const f64x2 a = *reinterpret_cast<const f64x2*>(wl + (0 * UNROLL + k) * 1024 + lane * 16);
    const f64x2 b = *reinterpret_cast<const f64x2*>(wl + (1 * UNROLL + k) * 1024 + lane * 16);
    const f64x2 c = *reinterpret_cast<const f64x2*>(wl + (2 * UNROLL + k) * 1024 + lane * 16);
    double l0 = ls, l1 = ls, u0 = us, u1 = us;
    bool s0 = true, s1 = true;
    f64x2 dd = f64x2{0.0, 0.0};
    if constexpr (Op::kNIn == 4) dd = *reinterpret_cast<const f64x2*>(wl + (3 * UNROLL + k) * 1024 + lane * 16);
    if constexpr (VECB && Op::kBox) {
      if (l_) { const f64x2 t = *reinterpret_cast<const f64x2*>(wl + ((Op::kNIn + 0) * UNROLL + k) * 1024 + lane * 16); l0 = t.x; l1 = t.y; }
      if (u_) { const f64x2 t = *reinterpret_cast<const f64x2*>(wl + ((Op::kNIn + 1) * UNROLL + k) * 1024 + lane * 16); u0 = t.x; u1 = t.y; }
    }
    if constexpr (MASK && Op::kBox) { s0 = (vm[k] & 0xff) != 0; s1 = (vm[k] >> 8) != 0; }
    f64x2 r;
    r.x = apply_op(op, a.x, dd.x, b.x, c.x, l0, u0, s0);
    r.y = apply_op(op, a.y, dd.y, b.y, c.y, l1, u1, s1);
    if constexpr (Op::kObj) {
      if (i < n2) hacc += op.hterm(b.x, c.x, r.x, s0) + op.hterm(b.y, c.y, r.y, s1);
    }
    if (i < n2) __builtin_nontemporal_store(r, y + i);
  }
  if constexpr (Op::kObj) {
    // one partial per WORKGROUP (round 3; round 2 wrote one per wavefront: 130 208 of them at n = 1e8, and the ordered
    // reduction behind the pass cost ~25 us of the fused call).  The wave's staging area has been consumed: its first 8 bytes
    // carry the wave's sum to lane 0 of the workgroup, which adds the four in a fixed order.
    const double t = wave_sum(hacc);
    if (lane == 0) *reinterpret_cast<double*>(wl) = t;
    __syncthreads();
    double tb = 0.0;
    if (threadIdx.x == 0) {
      const double* w0 = reinterpret_cast<const double*>(lds);
      constexpr int stride = NARR * UNROLL * 1024 / 8;
      tb = (w0[0] + w0[stride]) + (w0[2 * stride] + w0[3 * stride]);
    }
    value_publish(op, bid, tb);
  }
}

// scalar path: unaligned vectors, and the odd last element of the vector path ([begin, n))
template <class Op>
__global__ __launch_bounds__(256) void k_sep_scalar(double* y, const double* q, const double* d_, const double* xk,
                                                     const double* sj, const double* l_, const double* u_,
                                                     const uint8_t* mask, double ls, double us, int64_t begin,
                                                     int64_t n, Op op) {
  int64_t i = begin + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  double hacc = 0.0;
  for (; i < n; i += stride) {
    double li = l_ ? l_[i] : ls;
    double ui = u_ ? u_[i] : us;
    bool sel = mask ? (mask[i] != 0) : true;
    double di = 0.0;
    if constexpr (Op::kNIn == 4) di = d_[i];
    const double xi = xk[i], si = sj[i];
    const double yi = apply_op(op, q[i], di, xi, si, li, ui, sel);
    if constexpr (Op::kObj) hacc += op.hterm(xi, si, yi, sel);
    y[i] = yi;
  }
  if constexpr (Op::kObj) {
    __shared__ double lds4[4];
    const double t = block_sum4(hacc, lds4);
    if (threadIdx.x == 0) op.partials[blockIdx.x] = t;
  }
}

// Tuning knobs (spx_ctx_set_tuning).  Defaults from tools/sweep_sep.py on MI355X, n = 1e8 (profiles/r01_sweep_sep.txt):
// one tile per workgroup (no cap) + non-temporal loads/stores: 6.15 TB/s vs 5.66 TB/s for 16 WG/CU, plain.
// (the knobs live in the context: spx_ctx::tune_sep_*, set by spx_ctx_set_tuning)

#ifndef SPX_VECB_KIB
#define SPX_VECB_KIB 3  // KiB per wave and vector with vector bounds (5 vectors); measured at n = 1e8 (tools/bench_vecb.py): 2 -> 0.81 ms, 3 -> 0.76 ms, 4 -> 0.775 ms
#endif
template <class Op, bool VECB, bool MASK>
static int launch_vec(spx_ctx* ctx, double* y, const double* q, const double* d, const double* xk, const double* sj,
                      const double* l, const double* u, const uint8_t* mask, double ls, double us, int64_t n2, Op op,
                      int64_t* value_slots /* out: partial slots written (Op::kObj) */,
                      const SpxSyncHeader* fuse_hdr = nullptr /* Op::kObj: this launch is the whole call -- it may finish the value itself */,
                      bool* fused = nullptr) {
  // (Op::kObj) the one-launch form: the grid is the list of slots, short enough for one workgroup to add
  auto try_fuse = [&](int64_t blocks) {
    if constexpr (Op::kObj) {
      if (fuse_hdr != nullptr && blocks <= kValueFuseMax && !ctx->tune_sep_xcd) {
        op.fin_hdr = const_cast<SpxSyncHeader*>(fuse_hdr);
        *fused = true;
      }
    }
  };
  if constexpr (Op::kLdsKiB > 0) if (ctx->tune_sep_lds) {
    // 3 input vectors: 6 KiB per wave and vector -> 72 KiB per workgroup; 5 vectors (vector bounds): 3 KiB -> 60 KiB
    constexpr int U = (VECB && Op::kBox) ? (Op::kLdsKiB > SPX_VECB_KIB ? SPX_VECB_KIB : Op::kLdsKiB) : Op::kLdsKiB;  // <= 72 KiB per workgroup
    int64_t blocks = (n2 + 256 * U - 1) / (256 * U);
    int64_t xcd_chunk = 0;
    if (ctx->tune_sep_xcd) {
      xcd_chunk = (blocks + 7) / 8;
      blocks = xcd_chunk * 8;
    }
    *value_slots = blocks;  // one per workgroup (idle workgroups of the XCD-contiguous experiment write a zero)
    try_fuse(blocks);
    hipLaunchKernelGGL((k_sep_lds<Op, U, VECB, MASK>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, y, q, d, xk,
                       sj, l, u, mask, ls, us, n2, op, xcd_chunk);
    SPX_LAUNCH_CHECK();
    return SPX_OK;
  }
  constexpr int UNROLL = 4;
  const int64_t ntiles = (n2 + 256 * UNROLL - 1) / (256 * UNROLL);
  int64_t blocks = ntiles;
  const int64_t cap = ctx->tune_sep_blocks_per_cu > 0 ? (int64_t)ctx->num_cu * ctx->tune_sep_blocks_per_cu : (int64_t)0x7fffffff;
  if (blocks > cap) blocks = cap;
  *value_slots = blocks;  // one per workgroup
  try_fuse(blocks);
  if (ctx->tune_sep_nt)
    hipLaunchKernelGGL((k_sep_vec<Op, UNROLL, VECB, MASK, true>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, y,
                       q, d, xk, sj, l, u, mask, ls, us, n2, op);
  else
    hipLaunchKernelGGL((k_sep_vec<Op, UNROLL, VECB, MASK, false>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       y, q, d, xk, sj, l, u, mask, ls, us, n2, op);
  SPX_LAUNCH_CHECK();
  return SPX_OK;
}

template <class Op>
static int run_separable(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                         const double* l, const double* u, double ls, double us, const uint8_t* mask, Op op,
                         const double* d = nullptr, double* value = nullptr /* Op::kObj: sum of the h terms */,
                         double value_scale = 1.0 /* device-resident value = value_scale * sum */) {
  if constexpr (Op::kObj) *value = 0.0;
  if (n == 0) return SPX_OK;
  SPX_ON_DEVICE(ctx);
  double* partials = nullptr;  // ws: [result | pad to 256 B | partial slots]
  int64_t used = 0;
  if constexpr (Op::kObj) {
    const int64_t maxslots = ((n / 2) / (256 * 3) + 2) * 4 + 2 * (int64_t)ctx->num_cu * 8 + 16;
    int rcw = spx_ws_reserve(ctx, 256 + (size_t)maxslots * sizeof(double));
    if (rcw) return rcw;
    partials = reinterpret_cast<double*>(static_cast<char*>(ctx->ws) + 256);
    if (ctx->tune_fewer_launches) {  // (the one-launch form keeps its tickets in the synchronisation state)
      rcw = spx_sync_reserve(ctx, sizeof(SpxSyncHeader));
      if (rcw) return rcw;
    }
  }
  bool fused = false;
  // The vector skeletons need 16-byte aligned vectors (and a 2-byte aligned mask).  Views that all start 8 bytes off
  // (e.g. view(x, 2:n) of aligned arrays) are peeled: element 0 through the scalar kernel, the rest aligned again.
  auto vec_ok_at = [&](int64_t h) {
    auto a16 = [&](const double* p) { return !p || spx_aligned16(p + h); };
    return a16(y) && a16(q) && a16(xk) && a16(sj) && a16(d) && a16(l) && a16(u) &&
           (!mask || ((reinterpret_cast<uintptr_t>(mask) + (uintptr_t)h) & 1u) == 0);
  };
  auto launch_scalar = [&](int64_t begin, int64_t end) -> int {
    int64_t blocks = (end - begin + 255) / 256;
    const int64_t cap = (int64_t)ctx->num_cu * 8;
    if (blocks > cap) blocks = cap;
    if constexpr (Op::kObj) op.partials = partials + used;
    hipLaunchKernelGGL((k_sep_scalar<Op>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, y, q, d, xk, sj, l, u,
                       mask, ls, us, begin, end, op);
    SPX_LAUNCH_CHECK();
    used += blocks;
    return SPX_OK;
  };
  int64_t head = 0;
  if (!vec_ok_at(0) && n >= 3 && vec_ok_at(1)) head = 1;
  int64_t done = 0;
  if (head) {
    int rch = launch_scalar(0, head);
    if (rch) return rch;
    done = head;
  }
  if (vec_ok_at(head) && n - head >= 2) {
    const int64_t n2 = (n - head) / 2;
    auto sh = [&](const double* p) { return p ? p + head : p; };
    double* yv = y + head;
    const double *qv = sh(q), *dv = sh(d), *xv = sh(xk), *sv = sh(sj), *lv = sh(l), *uv = sh(u);
    const uint8_t* mv = mask ? mask + head : mask;
    int rc;
    int64_t slots = 0;
    const SpxSyncHeader* fh = nullptr;
    if constexpr (Op::kObj) {
      op.partials = partials + used;
      // one launch covers the whole vector (no peeled element in front, no odd element behind): it may finish the value
      if (ctx->tune_fewer_launches && head == 0 && 2 * n2 == n) {
        fh = reinterpret_cast<const SpxSyncHeader*>(ctx->sync);
        op.fin_result = reinterpret_cast<double*>(ctx->ws);
        op.fin_target = ctx->value_target;
        op.fin_scale = value_scale;
      }
    }
    if constexpr (Op::kBox) {
      const bool vecb = (l || u);
      const bool msk = (mask != nullptr);
      if (vecb && msk) rc = launch_vec<Op, true, true>(ctx, yv, qv, dv, xv, sv, lv, uv, mv, ls, us, n2, op, &slots, fh, &fused);
      else if (vecb) rc = launch_vec<Op, true, false>(ctx, yv, qv, dv, xv, sv, lv, uv, mv, ls, us, n2, op, &slots, fh, &fused);
      else if (msk) rc = launch_vec<Op, false, true>(ctx, yv, qv, dv, xv, sv, lv, uv, mv, ls, us, n2, op, &slots, fh, &fused);
      else rc = launch_vec<Op, false, false>(ctx, yv, qv, dv, xv, sv, lv, uv, mv, ls, us, n2, op, &slots, fh, &fused);
    } else {
      rc = launch_vec<Op, false, false>(ctx, yv, qv, dv, xv, sv, lv, uv, mv, ls, us, n2, op, &slots, fh, &fused);
    }
    if (rc) return rc;
    used += slots;
    done = head + 2 * n2;
  }
  if (done < n) {
    int rct = launch_scalar(done, n);
    if (rct) return rct;
  }
  if constexpr (Op::kObj) {
    double* result = reinterpret_cast<double*>(ctx->ws);
    if (!fused) {
      hipLaunchKernelGGL(k_value_reduce, dim3(1), dim3(1024), 0, ctx->stream, (const double*)partials, used, result,
                         value_scale, ctx->value_target);
      SPX_LAUNCH_CHECK();
    }
    if (ctx->value_target) {  // device-resident value: nothing is read back, the call returns after enqueueing
      *value = std::numeric_limits<double>::quiet_NaN();
      return SPX_OK;
    }
    { const int rcc = spx_require_not_capturing(ctx, "returning the value of prox_value to the host"); if (rcc) return rcc; }
    SPX_HIP(hipMemcpyAsync(value, result, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    SPX_HIP(hipStreamSynchronize(ctx->stream));
  }
  return SPX_OK;
}

// ---------------------------------------------------------------------------------------------
// C entry points
// ---------------------------------------------------------------------------------------------
SPX_EXPORT int spx_prox_l1(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                           double lambda, double sigma) {
  int rc = spx_check_common(ctx, y, q, xk, sj, n);
  if (rc) return rc;
  if (y == q && lambda * sigma >= 0.0)  // the reference's two-pass body with y === q (see OpL1Aliased)
    return run_separable(ctx, y, q, xk, sj, n, nullptr, nullptr, 0.0, 0.0, nullptr, OpL1Aliased{});
  return run_separable(ctx, y, q, xk, sj, n, nullptr, nullptr, 0.0, 0.0, nullptr, OpL1{lambda * sigma});
}

SPX_EXPORT int spx_prox_l0(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                           double lambda, double sigma) {
  int rc = spx_check_common(ctx, y, q, xk, sj, n);
  if (rc) return rc;
  return run_separable(ctx, y, q, xk, sj, n, nullptr, nullptr, 0.0, 0.0, nullptr, OpL0{std::sqrt(2 * lambda * sigma)});
}

SPX_EXPORT int spx_prox_lhalf(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                              double lambda, double sigma) {
  int rc = spx_check_common(ctx, y, q, xk, sj, n);
  if (rc) return rc;
  const double nl = sigma * lambda;
  const double p = std::pow(54.0, 1.0 / 3.0) * std::pow(2 * nl, 2.0 / 3.0) / 4;  // shiftedRootNormLhalf.jl:49
  return run_separable(ctx, y, q, xk, sj, n, nullptr, nullptr, 0.0, 0.0, nullptr, OpLhalf{nl / 4, p});
}

SPX_EXPORT int spx_prox_l1_box(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                               double lambda, double sigma, const double* l_vec, const double* u_vec, double l_scalar,
                               double u_scalar, const uint8_t* sel_mask) {
  int rc = spx_check_common(ctx, y, q, xk, sj, n);
  if (rc) return rc;
  return run_separable(ctx, y, q, xk, sj, n, l_vec, u_vec, l_scalar, u_scalar, sel_mask, OpL1Box{sigma * lambda});
}

SPX_EXPORT int spx_prox_l0_box(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                               double lambda, double sigma, const double* l_vec, const double* u_vec, double l_scalar,
                               double u_scalar, const uint8_t* sel_mask) {
  int rc = spx_check_common(ctx, y, q, xk, sj, n);
  if (rc) return rc;
  return run_separable(ctx, y, q, xk, sj, n, l_vec, u_vec, l_scalar, u_scalar, sel_mask, OpL0Box{2 * lambda * sigma});
}

SPX_EXPORT int spx_prox_lhalf_box(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                                  int64_t n, double lambda, double sigma, const double* l_vec, const double* u_vec,
                                  double l_scalar, double u_scalar, const uint8_t* sel_mask) {
  int rc = spx_check_common(ctx, y, q, xk, sj, n);
  if (rc) return rc;
  return run_separable(ctx, y, q, xk, sj, n, l_vec, u_vec, l_scalar, u_scalar, sel_mask,
                       OpLhalfBox{sigma * lambda / 4, lambda, 0.5 / sigma});
}

// ---------------------------------------------------------------------------------------------
// prox! + value of h at the result, in one pass (synchronous: *value is written on the host)
// ---------------------------------------------------------------------------------------------
template <class Base, class Term>
static int run_proxval(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                       const double* l, const double* u, double ls, double us, const uint8_t* mask, Base base,
                       double lambda, double q_scale, double* value) {
  int rc = spx_check_common(ctx, y, q, xk, sj, n);
  if (rc) return rc;
  SPX_REQUIRE(value != nullptr, "value is NULL");
  WithValue<Base, Term> op{base, nullptr, q_scale};
  double sum = 0.0;
  rc = run_separable(ctx, y, q, xk, sj, n, l, u, ls, us, mask, op, nullptr, &sum, lambda);
  *value = lambda * sum;  // (NaN when the value went to the context's device target)
  return rc;
}

SPX_EXPORT int spx_proxval_l1(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                              double lambda, double sigma, double q_scale, double* value) {
  if (y == q && lambda * sigma >= 0.0) {  // the reference's two-pass body with y === q (see OpL1Aliased)
    SPX_REQUIRE(q_scale == 1.0, "q_scale != 1 with y aliasing q");
    return run_proxval<OpL1Aliased, HTermL1>(ctx, y, q, xk, sj, n, nullptr, nullptr, 0.0, 0.0, nullptr, OpL1Aliased{},
                                             lambda, 1.0, value);
  }
  return run_proxval<OpL1, HTermL1>(ctx, y, q, xk, sj, n, nullptr, nullptr, 0.0, 0.0, nullptr, OpL1{lambda * sigma},
                                    lambda, q_scale, value);
}
SPX_EXPORT int spx_proxval_l0(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                              double lambda, double sigma, double q_scale, double* value) {
  return run_proxval<OpL0, HTermL0>(ctx, y, q, xk, sj, n, nullptr, nullptr, 0.0, 0.0, nullptr,
                                    OpL0{std::sqrt(2 * lambda * sigma)}, lambda, q_scale, value);
}
SPX_EXPORT int spx_proxval_lhalf(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                                 double lambda, double sigma, double q_scale, double* value) {
  const double nl = sigma * lambda;
  const double p = std::pow(54.0, 1.0 / 3.0) * std::pow(2 * nl, 2.0 / 3.0) / 4;
  return run_proxval<OpLhalf, HTermLhalf>(ctx, y, q, xk, sj, n, nullptr, nullptr, 0.0, 0.0, nullptr, OpLhalf{nl / 4, p},
                                          lambda, q_scale, value);
}
SPX_EXPORT int spx_proxval_l1_box(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                                  double lambda, double sigma, const double* l_vec, const double* u_vec,
                                  double l_scalar, double u_scalar, const uint8_t* sel_mask, double q_scale, double* value) {
  return run_proxval<OpL1Box, HTermL1>(ctx, y, q, xk, sj, n, l_vec, u_vec, l_scalar, u_scalar, sel_mask,
                                       OpL1Box{sigma * lambda}, lambda, q_scale, value);
}
SPX_EXPORT int spx_proxval_l0_box(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                                  double lambda, double sigma, const double* l_vec, const double* u_vec,
                                  double l_scalar, double u_scalar, const uint8_t* sel_mask, double q_scale, double* value) {
  return run_proxval<OpL0Box, HTermL0>(ctx, y, q, xk, sj, n, l_vec, u_vec, l_scalar, u_scalar, sel_mask,
                                       OpL0Box{2 * lambda * sigma}, lambda, q_scale, value);
}
SPX_EXPORT int spx_proxval_lhalf_box(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                                     int64_t n, double lambda, double sigma, const double* l_vec, const double* u_vec,
                                     double l_scalar, double u_scalar, const uint8_t* sel_mask, double q_scale, double* value) {
  return run_proxval<OpLhalfBox, HTermLhalf>(ctx, y, q, xk, sj, n, l_vec, u_vec, l_scalar, u_scalar, sel_mask,
                                             OpLhalfBox{sigma * lambda / 4, lambda, 0.5 / sigma}, lambda, q_scale, value);
}

// ---------------------------------------------------------------------------------------------
// iprox! entry points
// ---------------------------------------------------------------------------------------------
template <class Op>
static int run_iprox_unboxed(spx_ctx* ctx, double* y, const double* g, const double* d, const double* xk,
                             const double* sj, int64_t n, double lambda, int check_d) {
  int rc = spx_check_common(ctx, y, g, xk, sj, n);
  if (rc) return rc;
  SPX_REQUIRE(n == 0 || d != nullptr, "d is NULL");
  if (n == 0) return SPX_OK;
  rc = spx_ws_reserve(ctx, 256);
  if (rc) return rc;
  SPX_ON_DEVICE(ctx);
  int* flag = reinterpret_cast<int*>(ctx->ws);
  { const int rz = spx_zero_async(ctx, flag, sizeof(int)); if (rz) return rz; }
  rc = run_separable(ctx, y, g, xk, sj, n, nullptr, nullptr, 0.0, 0.0, nullptr, Op{lambda, flag}, d);
  if (rc || !check_d) return rc;
  int bad = 0;
  { const int rcc = spx_require_not_capturing(ctx, "the d > 0 check of iprox! (pass check = 0)"); if (rcc) return rcc; }
  SPX_HIP(hipMemcpyAsync(&bad, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  SPX_HIP(hipStreamSynchronize(ctx->stream));
  if (bad) {
    spx_set_error("AssertionError: d[i] > 0");
    return SPX_ERR_ASSERT;
  }
  return SPX_OK;
}

SPX_EXPORT int spx_iprox_l1(spx_ctx* ctx, double* y, const double* g, const double* d, const double* xk,
                            const double* sj, int64_t n, double lambda, int check_d) {
  return run_iprox_unboxed<OpIproxL1>(ctx, y, g, d, xk, sj, n, lambda, check_d);
}
SPX_EXPORT int spx_iprox_l0(spx_ctx* ctx, double* y, const double* g, const double* d, const double* xk,
                            const double* sj, int64_t n, double lambda, int check_d) {
  return run_iprox_unboxed<OpIproxL0>(ctx, y, g, d, xk, sj, n, lambda, check_d);
}
SPX_EXPORT int spx_iprox_l1_box(spx_ctx* ctx, double* y, const double* g, const double* d, const double* xk,
                                const double* sj, int64_t n, double lambda, const double* l_vec, const double* u_vec,
                                double l_scalar, double u_scalar, const uint8_t* sel_mask) {
  int rc = spx_check_common(ctx, y, g, xk, sj, n);
  if (rc) return rc;
  SPX_REQUIRE(n == 0 || d != nullptr, "d is NULL");
  return run_separable(ctx, y, g, xk, sj, n, l_vec, u_vec, l_scalar, u_scalar, sel_mask, OpIproxL1Box{lambda}, d);
}
SPX_EXPORT int spx_iprox_l0_box(spx_ctx* ctx, double* y, const double* g, const double* d, const double* xk,
                                const double* sj, int64_t n, double lambda, const double* l_vec, const double* u_vec,
                                double l_scalar, double u_scalar, const uint8_t* sel_mask) {
  int rc = spx_check_common(ctx, y, g, xk, sj, n);
  if (rc) return rc;
  SPX_REQUIRE(n == 0 || d != nullptr, "d is NULL");
  return run_separable(ctx, y, g, xk, sj, n, l_vec, u_vec, l_scalar, u_scalar, sel_mask, OpIproxL0Box{lambda}, d);
}
