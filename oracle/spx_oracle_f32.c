/*
 * spx_oracle_f32.c -- Float32 build of the CPU restatement, for the operators whose reference bodies are pure Float32
 * arithmetic when R = Float32 (the NormL1 / NormL0 families).
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (same rule as spx_oracle.c).
 *
 * The reference is generic in R <: Real (src/shiftedNormL1Box.jl:89-94 etc.); with R = Float32, Int literals promote to
 * Float32 (`2 * psi.lambda * sigma`, `0`) and every +, -, *, comparison, min / max and the one sqrt is a Float32 operation.
 * x86-64 SSE evaluates `float` expressions in IEEE binary32 (no excess precision), gcc -ffp-contract=off: the loops
 * below are the same statements as in spx_oracle.c with `float` for `double`.
 * Pinning: the reference's tests hold no Float32 prox values (test/runtests.jl:196-209 checks types and psi(0) only):
 * PARITY UNPINNED by reference vectors; checked against the Float64 restatement on data where both are exact
 * (tests/test_oracle_golden.py::test_f32_oracle_agrees_with_f64_on_dyadic_data).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#define ORC_API __attribute__((visibility("default")))

static inline float jl_minf(float x, float y) {
  float d = x - y;
  float a = signbit(d) ? x : y;
  return (isnan(x) || isnan(y)) ? d : a;
}
static inline float jl_maxf(float x, float y) {
  float d = x - y;
  float a = signbit(d) ? y : x;
  return (isnan(x) || isnan(y)) ? d : a;
}
static inline int is_selected(const uint8_t* mask, int64_t i) { return mask == NULL || mask[i] != 0; }

/* ShiftedNormL1.prox!  src/shiftedNormL1.jl:40-54 (two passes: y === q clobbers q first, as in the reference) */
ORC_API void orc32_prox_l1(float* y, const float* q, const float* xk, const float* sj, int64_t n, float lambda, float sigma) {
  for (int64_t i = 0; i < n; ++i) y[i] = (-xk[i]) - sj[i];
  for (int64_t i = 0; i < n; ++i) {
    float qi = q[i];
    y[i] = jl_minf(jl_maxf(y[i], qi - lambda * sigma), qi + lambda * sigma);
  }
}

/* ShiftedNormL0.prox!  src/shiftedNormL0.jl:38-55 */
ORC_API void orc32_prox_l0(float* y, const float* q, const float* xk, const float* sj, int64_t n, float lambda, float sigma) {
  const float c = sqrtf(2 * lambda * sigma); /* :45 */
  for (int64_t i = 0; i < n; ++i) {
    float xps = xk[i] + sj[i];
    float qi = q[i];
    y[i] = (fabsf(xps + qi) <= c) ? -xps : qi;
  }
}

/* ShiftedNormL1Box.prox!  src/shiftedNormL1Box.jl:89-125 */
ORC_API void orc32_prox_l1_box(float* y, const float* q, const float* xk, const float* sj, int64_t n, float lambda,
                               float sigma, const float* lvec, const float* uvec, float lscal, float uscal,
                               const uint8_t* mask) {
  const float sl = sigma * lambda; /* :96 */
  for (int64_t i = 0; i < n; ++i) {
    float li = lvec ? lvec[i] : lscal, ui = uvec ? uvec[i] : uscal;
    float qi = q[i], si = sj[i];
    if (is_selected(mask, i)) {
      float xs = xk[i] + si;
      float xsq = xs + qi;
      float t;
      if (xsq <= -sl) t = qi + sl;
      else if (xsq >= sl) t = qi - sl;
      else t = -xs;
      y[i] = jl_minf(jl_maxf(t, li - si), ui - si); /* :118 */
    } else {
      y[i] = jl_minf(jl_maxf(qi, li - si), ui - si); /* :121 */
    }
  }
}

/* ShiftedNormL0Box.prox!  src/shiftedNormL0Box.jl:89-131 */
ORC_API void orc32_prox_l0_box(float* y, const float* q, const float* xk, const float* sj, int64_t n, float lambda,
                               float sigma, const float* lvec, const float* uvec, float lscal, float uscal,
                               const uint8_t* mask) {
  const float c = 2 * lambda * sigma; /* :96 */
  for (int64_t i = 0; i < n; ++i) {
    float li = lvec ? lvec[i] : lscal, ui = uvec ? uvec[i] : uscal;
    float qi = q[i], si = sj[i];
    float sq = si + qi;
    if (is_selected(mask, i)) {
      float xi = xk[i];
      float xs = xi + si;
      float xsq = xs + qi;
      float dl = li - sq, du = ui - sq;
      float val_left = dl * dl + ((xi == -li) ? 0.0f : c);
      float val_right = du * du + ((xi == -ui) ? 0.0f : c);
      float yi = (val_left < val_right) ? (li - si) : (ui - si);
      float val_min = jl_minf(val_left, val_right);
      float mxi = -xi;
      if (li <= mxi && mxi <= ui) {
        float val_0 = xsq * xsq;
        if (val_0 < val_min) yi = -xs;
        val_min = jl_minf(val_0, val_min);
      }
      if (li <= sq && sq <= ui) {
        float val_xsq = (xsq == 0.0f) ? 0.0f : c;
        if (val_xsq < val_min) yi = qi;
      }
      y[i] = yi;
    } else {
      y[i] = jl_minf(jl_maxf(qi, li - si), ui - si); /* :127 */
    }
  }
}

/* ==========================================================================================
 * iprox! with R = Float32 (round 3): the Float64 block of spx_oracle.c, type-substituted by the Makefile
 * (src/shiftedNormL1.jl:60-75, shiftedNormL0.jl:61-80, shiftedNormL1Box.jl:131-225, shiftedNormL0Box.jl:137-231,
 * ShiftedProximalOperators.jl:217-236; thresholds eps(R) = eps(Float32)).
 * ========================================================================================== */
#include "_gen/iprox_f32.inc"
