// spx_separable_f32.hip -- the separable prox! operators on Float32 vectors (round 2 widening).
//
// The reference's structs and prox! methods are generic in R <: Real (src/shiftedNormL1Box.jl:89-94: `y::AbstractVector{R},
// psi::ShiftedNormL1Box{R, ...}, q::AbstractVector{R}, sigma::R`) and its tests build Float32 operators on views
// (test/runtests.jl:196-209).  With R = Float32 every operation of the L1 / L0 bodies below is a Float32 operation
// (Int literals promote to Float32: `2 * psi.lambda * sigma`, `0`), so the results are reproduced BIT FOR BIT in fp32:
//   ShiftedNormL1      src/shiftedNormL1.jl:40-54         y = min(max((-xk) - sj, q - lambda sigma), q + lambda sigma)
//   ShiftedNormL0      src/shiftedNormL0.jl:38-55         c = sqrt(2 lambda sigma);  |xk + sj + q| <= c ? -(xk + sj) : q
//   ShiftedNormL1Box   src/shiftedNormL1Box.jl:89-125
//   ShiftedNormL0Box   src/shiftedNormL0Box.jl:89-131
// (RootNormLhalf is NOT here: its body mixes Float64 literals -- `^(-3 / 2)`, `54^(1 / 3)` -- into the Float32 data, i.e. the
// reference itself computes it in Float64 and rounds; the fp64 entry points cover that after a conversion by the caller.)
//
// HBM: 16 B/element (read q, xk, sj, write y; +8 with vector bounds, +1 with a mask) -- half the bytes of the fp64 path.
// Skeleton: one tile of 256 lanes x 4 x (4 floats = 16 bytes) per workgroup, all 12 non-temporal 16-byte loads of a lane
// issued before first use, 16-byte non-temporal stores.  Views that start at any element (4-byte granularity) are
// peeled to a 16-byte boundary when all vectors share the misalignment; mixed alignments take 4-byte accesses.
// y === q is safe (a lane reads q[i] before it writes y[i]; lanes own disjoint indices).
#include <cmath>

#include "spx_common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Base.min / Base.max on Float32 (IEEE-754-2019 minimum / maximum: -0.0 < +0.0, NaN propagates), as jl_min / jl_max
__device__ __forceinline__ float jl_minf(float x, float y) {
  const float d = x - y;
  const float a = (__float_as_int(d) < 0) ? x : y;
  return (x != x || y != y) ? d : a;
}
__device__ __forceinline__ float jl_maxf(float x, float y) {
  const float d = x - y;
  const float a = (__float_as_int(d) < 0) ? y : x;
  return (x != x || y != y) ? d : a;
}

struct F32L1 {  // src/shiftedNormL1.jl:46-51
  float ls;     // lambda * sigma (Float32 product)
  static constexpr bool kBox = false;
  static constexpr int kNIn = 3;
  __device__ __forceinline__ float operator()(float q, float x, float s, float, float, bool) const {
    const float t = (-x) - s;                      // :47
    return jl_minf(jl_maxf(t, q - ls), q + ls);    // :50
  }
};
struct F32L1Aliased {  // y === q: the broadcast at :47 overwrites q before :50 reads it (as OpL1Aliased)
  static constexpr bool kBox = false;
  static constexpr int kNIn = 3;
  __device__ __forceinline__ float operator()(float, float x, float s, float, float, bool) const { return (-x) - s; }
};
struct F32L0 {  // src/shiftedNormL0.jl:45-52
  float c;      // sqrt(2 * lambda * sigma) in Float32
  static constexpr bool kBox = false;
  static constexpr int kNIn = 3;
  __device__ __forceinline__ float operator()(float q, float x, float s, float, float, bool) const {
    const float xps = x + s;
    return (fabsf(xps + q) <= c) ? -xps : q;
  }
};
struct F32L1Box {  // src/shiftedNormL1Box.jl:96-122
  float sl;        // sigma * lambda
  static constexpr bool kBox = true;
  static constexpr int kNIn = 3;
  __device__ __forceinline__ float operator()(float q, float x, float s, float l, float u, bool sel) const {
    const float xs = x + s;
    const float xsq = xs + q;
    float t = (xsq <= -sl) ? (q + sl) : ((xsq >= sl) ? (q - sl) : -xs);  // :111-117
    t = sel ? t : q;                                                      // :121 prox_zero(qi, ...)
    return jl_minf(jl_maxf(t, l - s), u - s);                             // :118
  }
};
struct F32L0Box {  // src/shiftedNormL0Box.jl:96-128
  float c;         // 2 * lambda * sigma
  static constexpr bool kBox = true;
  static constexpr int kNIn = 3;
  __device__ __forceinline__ float operator()(float q, float x, float s, float l, float u, bool sel) const {
    const float sq = s + q;
    const float xs = x + s;
    const float xsq = xs + q;
    const float dl = l - sq, du = u - sq;
    const float val_left = dl * dl + ((x == -l) ? 0.0f : c);   // :110
    const float val_right = du * du + ((x == -u) ? 0.0f : c);  // :111
    float yi = (val_left < val_right) ? (l - s) : (u - s);     // :114
    float val_min = jl_minf(val_left, val_right);
    const float mx = -x;
    if (l <= mx && mx <= u) {  // :116
      const float val_0 = xsq * xsq;
      yi = (val_0 < val_min) ? -xs : yi;
      val_min = jl_minf(val_0, val_min);
    }
    if (l <= sq && sq <= u) {  // :121
      const float val_xsq = (xsq == 0.0f) ? 0.0f : c;
      yi = (val_xsq < val_min) ? q : yi;
    }
    return sel ? yi : jl_minf(jl_maxf(q, l - s), u - s);  // :127 prox_zero
  }
};

// ---------------------------------------------------------------------------------------------
// iprox! on Float32 vectors (round 3).  The reference's iprox! methods are generic in R too (src/shiftedNormL1.jl:60-75,
// shiftedNormL0.jl:61-80, shiftedNormL1Box.jl:131-225, shiftedNormL0Box.jl:137-231, ShiftedProximalOperators.jl:217-236): with
// R = Float32 every +, -, *, /, sqrt and comparison is a Float32 operation and the thresholds are eps(Float32) = 2^-23.
// The bodies below are the reference's branch structure, statement by statement; the tests compare them bit for bit with a
// Float32 build of the CPU restatement.  Four input vectors (g, d, xk, sj): 20 B/element.
// ---------------------------------------------------------------------------------------------
constexpr float kEps32 = 1.1920928955078125e-07f;  // eps(Float32)
__device__ __forceinline__ float iprox_zero_f32(float d, float g, float l, float u) {  // ShiftedProximalOperators.jl:217-236
  if (d > kEps32) return jl_minf(jl_maxf(-g / d, l), u);
  if (d < -kEps32) {
    const float d_2 = d / 2;
    return ((d_2 * (l * l) + g * l) < (d_2 * (u * u) + g * u)) ? l : u;
  }
  return (g > 0.0f) ? l : ((g < 0.0f) ? u : 0.0f);
}
struct F32IproxL1 {  // src/shiftedNormL1.jl:60-75
  float lambda;
  int* flag;  // set when some d[i] <= 0 (the reference's `@assert d[i] > 0`)
  static constexpr bool kBox = false;
  static constexpr int kNIn = 4;
  __device__ __forceinline__ float call4(float g, float d, float x, float s, float, float, bool) const {
    if (!(d > 0.0f) && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(flag, 1);
    const float t = (-x) - s;                                                // :67
    return jl_minf(jl_maxf(t, -g / d - lambda / d), -g / d + lambda / d);    // :71
  }
};
struct F32IproxL0 {  // src/shiftedNormL0.jl:61-80
  float lambda;
  int* flag;
  static constexpr bool kBox = false;
  static constexpr int kNIn = 4;
  __device__ __forceinline__ float call4(float g, float d, float x, float s, float, float, bool) const {
    if (!(d > 0.0f) && __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(flag, 1);
    const float ci = sqrtf(2 * lambda * d);                                  // :71
    const float xps = x + s;
    return (fabsf(d * xps - g) <= ci) ? -xps : (-g / d);                     // :73-77
  }
};
struct F32IproxL1Box {  // src/shiftedNormL1Box.jl:131-225
  float lambda;
  static constexpr bool kBox = true;
  static constexpr int kNIn = 4;
  __device__ __forceinline__ float call4(float g, float d, float x, float s, float l, float u, bool sel) const {
    const float xs = x + s;
    const float left = l - s, right = u - s;
    if (!sel) return iprox_zero_f32(d, g, left, right);  // :221
    float yi;
    if (fabsf(d) <= kEps32) {  // :152
      yi = (fabsf(g) <= lambda) ? jl_minf(jl_maxf(left, -xs), right) : ((g > 0.0f) ? left : right);
    } else {
      const float d_2 = d / 2;
      const float lx = l + x, ux = u + x;
      const float g2_d = g / d_2;
      const float f2_d = g2_d - 2 * xs;
      const float l2_d = lambda / d_2;
      const float val_left = lx * lx + f2_d * lx + l2_d * fabsf(lx);
      const float val_right = ux * ux + f2_d * ux + l2_d * fabsf(ux);
      if (d > kEps32) {  // :161
        float val_min = jl_minf(val_left, val_right);
        yi = (val_left < val_right) ? left : right;
        const float y1 = -(g + lambda) / d;
        const float y2 = (lambda - g) / d;
        if (lx >= 0.0f) {
          if (left <= y1 && y1 <= right) yi = y1;
        } else if (0.0f >= ux) {
          if (left <= y2 && y2 <= right) yi = y2;
        } else {
          if (left <= y1 && y1 <= right) {
            const float v1 = xs + y1;
            const float q1 = v1 * v1 + f2_d * v1 + l2_d * fabsf(v1);
            if (q1 < val_min) yi = y1;
            val_min = jl_minf(q1, val_min);
          }
          if (left <= y2 && y2 <= right) {
            const float v2 = xs + y2;
            const float q2 = v2 * v2 + f2_d * v2 + l2_d * fabsf(v2);
            if (q2 < val_min) yi = y2;
            val_min = jl_minf(q2, val_min);
          }
          if (0.0f < val_min) yi = -xs;  // val_0 = 0
        }
      } else {  // d <= -eps, :199
        const float val_max = jl_maxf(val_left, val_right);
        yi = (val_left > val_right) ? left : right;
        const float mx = -x;
        if (l <= mx && mx <= u && 0.0f > val_max) yi = -xs;
      }
    }
    return yi;
  }
};
struct F32IproxL0Box {  // src/shiftedNormL0Box.jl:137-231
  float lambda;
  static constexpr bool kBox = true;
  static constexpr int kNIn = 4;
  __device__ __forceinline__ float call4(float g, float d, float x, float s, float l, float u, bool sel) const {
    const float xs = x + s;
    const float mx = -x;
    const bool zero_ok = (l <= mx && mx <= u);
    const float left = l - s, right = u - s;
    if (!sel) return iprox_zero_f32(d, g, left, right);  // :227
    float yi;
    if (fabsf(d) < kEps32) {  // :154
      if (g == 0.0f) {
        yi = zero_ok ? -xs : 0.0f;
      } else {
        const bool pos = g > 0.0f;
        const float t = pos ? left : right;
        const float val_min = g * t + ((x == (pos ? -l : -u)) ? 0.0f : lambda);
        yi = t;
        if (zero_ok && (-g * xs) < val_min) yi = -xs;
      }
    } else {
      const float d_2 = d / 2;
      const float lx = l + x, ux = u + x;
      const float g2_d = g / d_2;
      const float f2_d = g2_d - 2 * xs;
      const float l2_d = lambda / d_2;
      const float val_left = (lx == 0.0f) ? 0.0f : (lx * lx + f2_d * lx + l2_d);
      const float val_right = (ux == 0.0f) ? 0.0f : (ux * ux + f2_d * ux + l2_d);
      if (d >= kEps32) {  // :190
        const float aqy = -g / d;
        const float aqv = aqy + xs;
        float val_min;
        if (lx <= aqv && aqv <= ux) {
          val_min = (aqv == 0.0f) ? (-(aqv * aqv)) : (-(aqv * aqv) + l2_d);
          yi = aqy;
        } else {
          yi = (val_left < val_right) ? left : right;
          val_min = jl_minf(val_left, val_right);
        }
        if (zero_ok && 0.0f < val_min) yi = -xs;
      } else {  // :213
        yi = (val_left > val_right) ? left : right;
        const float val_max = jl_maxf(val_left, val_right);
        if (zero_ok && 0.0f > val_max) yi = -xs;
      }
    }
    return yi;
  }
};
template <class Op>
__device__ __forceinline__ float apply_f32(const Op& op, float q, float d, float x, float s, float l, float u, bool sel) {
  if constexpr (Op::kNIn == 4) return op.call4(q, d, x, s, l, u, sel);
  else return op(q, x, s, l, u, sel);
}

constexpr int kF32U = 4;                       // 16-byte groups per lane and vector
constexpr int kF32Tile = 256 * kF32U;          // 16-byte groups per workgroup

// body: `n4` groups of 4 floats starting at the (16-byte aligned) pointers; head: `head` (< 4) elements in front of them
// (at [-head .. -1]) and tail: `tail` (< 4) elements behind them, taken by single lanes of workgroup 0.
template <class Op, bool VECB, bool MASK>
__global__ __launch_bounds__(256) void k_sep_f32(float* y_, const float* q_, const float* d_, const float* xk_, const float* sj_,
                                                  const float* l_, const float* u_, const uint8_t* mask, float ls,
                                                  float us, int64_t n4, int head, int tail, Op op) {
  f32x4* y = reinterpret_cast<f32x4*>(y_);
  const f32x4* q = reinterpret_cast<const f32x4*>(q_);
  const f32x4* xk = reinterpret_cast<const f32x4*>(xk_);
  const f32x4* sj = reinterpret_cast<const f32x4*>(sj_);
  const f32x4* lv = reinterpret_cast<const f32x4*>(l_);
  const f32x4* uv = reinterpret_cast<const f32x4*>(u_);
  const f32x4* dv = reinterpret_cast<const f32x4*>(d_);
  const int64_t base = (int64_t)blockIdx.x * kF32Tile + threadIdx.x;
  f32x4 a[kF32U], b[kF32U], c[kF32U], lo[kF32U], up[kF32U], dd[kF32U];
#pragma unroll
  for (int k = 0; k < kF32U; ++k) {
    int64_t i = base + k * 256;
    if (i >= n4) i = n4 > 0 ? n4 - 1 : 0;
    if (n4 > 0) {
      a[k] = __builtin_nontemporal_load(q + i);
      b[k] = __builtin_nontemporal_load(xk + i);
      c[k] = __builtin_nontemporal_load(sj + i);
      if constexpr (Op::kNIn == 4) dd[k] = __builtin_nontemporal_load(dv + i);
      if constexpr (VECB && Op::kBox) {
        lo[k] = l_ ? __builtin_nontemporal_load(lv + i) : f32x4{ls, ls, ls, ls};
        up[k] = u_ ? __builtin_nontemporal_load(uv + i) : f32x4{us, us, us, us};
      }
    }
  }
#pragma unroll
  for (int k = 0; k < kF32U; ++k) {
    const int64_t i = base + k * 256;
    if (i < n4) {
      f32x4 o;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float l1 = ls, u1 = us;
        if constexpr (VECB && Op::kBox) { l1 = lo[k][e]; u1 = up[k][e]; }
        bool sel = true;
        if constexpr (MASK && Op::kBox) sel = mask[4 * i + e] != 0;
        float d1 = 0.0f;
        if constexpr (Op::kNIn == 4) d1 = dd[k][e];
        o[e] = apply_f32(op, a[k][e], d1, b[k][e], c[k][e], l1, u1, sel);
      }
      __builtin_nontemporal_store(o, y + i);
    }
  }
  if (blockIdx.x == 0 && (int)threadIdx.x < head + tail) {  // the few elements outside the aligned body
    const int64_t i = ((int)threadIdx.x < head) ? (int64_t)threadIdx.x - head : 4 * n4 + ((int)threadIdx.x - head);
    float l1 = ls, u1 = us;
    if constexpr (Op::kBox) {
      if (l_) l1 = l_[i];
      if (u_) u1 = u_[i];
    }
    bool sel = true;
    if constexpr (MASK && Op::kBox) sel = mask[i] != 0;
    float d1 = 0.0f;
    if constexpr (Op::kNIn == 4) d1 = d_[i];
    y_[i] = apply_f32(op, q_[i], d1, xk_[i], sj_[i], l1, u1, sel);
  }
}

// mixed alignments: 4-byte accesses
template <class Op>
__global__ __launch_bounds__(256) void k_sep_f32_scalar(float* y, const float* q, const float* d, const float* xk, const float* sj,
                                                         const float* l, const float* u, const uint8_t* mask, float ls,
                                                         float us, int64_t n, Op op) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float l1 = ls, u1 = us;
    bool sel = true;
    if constexpr (Op::kBox) {
      if (l) l1 = l[i];
      if (u) u1 = u[i];
      if (mask) sel = mask[i] != 0;
    }
    float d1 = 0.0f;
    if constexpr (Op::kNIn == 4) d1 = d[i];
    y[i] = apply_f32(op, q[i], d1, xk[i], sj[i], l1, u1, sel);
  }
}

template <class Op>
int run_f32(spx_ctx* ctx, float* y, const float* q, const float* xk, const float* sj, int64_t n, const float* l,
            const float* u, float ls, float us, const uint8_t* mask, Op op, const float* d = nullptr) {
  int rc = spx_check_common(ctx, y, q, xk, sj, n);
  if (rc) return rc;
  if (n == 0) return SPX_OK;
  SPX_ON_DEVICE(ctx);
  auto mis = [](const void* p) { return (unsigned)(reinterpret_cast<uintptr_t>(p) & 15u); };
  const unsigned m = mis(y);
  bool same = (m % 4 == 0) && mis(q) == m && mis(xk) == m && mis(sj) == m && (!d || mis(d) == m);
  if (Op::kBox) same = same && (!l || mis(l) == m) && (!u || mis(u) == m);
  if (!same) {
    int64_t blocks = (n + 255) / 256;
    const int64_t cap = (int64_t)ctx->num_cu * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((k_sep_f32_scalar<Op>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, y, q, d, xk, sj, l, u, mask, ls,
                       us, n, op);
    SPX_LAUNCH_CHECK();
    return SPX_OK;
  }
  int head = (int)(((16u - m) / 4u) & 3u);
  if (head > n) head = (int)n;
  const int64_t n4 = (n - head) / 4;
  const int tail = (int)(n - head - 4 * n4);
  int64_t blocks = (n4 + kF32Tile - 1) / kF32Tile;
  if (blocks < 1) blocks = 1;
  const dim3 grid((unsigned)blocks), block(256);
  const bool vecb = Op::kBox && (l || u);
  const bool msk = Op::kBox && mask != nullptr;
  const uint8_t* mk = mask ? mask + head : nullptr;
  const float* lp = l ? l + head : nullptr;
  const float* up = u ? u + head : nullptr;
  const float* dp = d ? d + head : nullptr;
#define SPX_F32_LAUNCH(VB, MK)                                                                                         \
  hipLaunchKernelGGL((k_sep_f32<Op, VB, MK>), grid, block, 0, ctx->stream, y + head, q + head, dp, xk + head, sj + head, lp, \
                     up, mk, ls, us, n4, head, tail, op)
  if (vecb && msk) SPX_F32_LAUNCH(true, true);
  else if (vecb) SPX_F32_LAUNCH(true, false);
  else if (msk) SPX_F32_LAUNCH(false, true);
  else SPX_F32_LAUNCH(false, false);
#undef SPX_F32_LAUNCH
  SPX_LAUNCH_CHECK();
  return SPX_OK;
}

}  // namespace

SPX_EXPORT int spx_prox_l1_f32(spx_ctx* ctx, float* y, const float* q, const float* xk, const float* sj, int64_t n,
                               float lambda, float sigma) {
  if (y == q && n > 0) return run_f32(ctx, y, q, xk, sj, n, nullptr, nullptr, 0.0f, 0.0f, nullptr, F32L1Aliased{});
  return run_f32(ctx, y, q, xk, sj, n, nullptr, nullptr, 0.0f, 0.0f, nullptr, F32L1{lambda * sigma});
}

SPX_EXPORT int spx_prox_l0_f32(spx_ctx* ctx, float* y, const float* q, const float* xk, const float* sj, int64_t n,
                               float lambda, float sigma) {
  return run_f32(ctx, y, q, xk, sj, n, nullptr, nullptr, 0.0f, 0.0f, nullptr, F32L0{sqrtf(2 * lambda * sigma)});
}

SPX_EXPORT int spx_prox_l1_box_f32(spx_ctx* ctx, float* y, const float* q, const float* xk, const float* sj, int64_t n,
                                   float lambda, float sigma, const float* l_vec, const float* u_vec, float l_scalar,
                                   float u_scalar, const uint8_t* sel_mask) {
  return run_f32(ctx, y, q, xk, sj, n, l_vec, u_vec, l_scalar, u_scalar, sel_mask, F32L1Box{sigma * lambda});
}

SPX_EXPORT int spx_prox_l0_box_f32(spx_ctx* ctx, float* y, const float* q, const float* xk, const float* sj, int64_t n,
                                   float lambda, float sigma, const float* l_vec, const float* u_vec, float l_scalar,
                                   float u_scalar, const uint8_t* sel_mask) {
  return run_f32(ctx, y, q, xk, sj, n, l_vec, u_vec, l_scalar, u_scalar, sel_mask, F32L0Box{2 * lambda * sigma});
}

// ---- iprox! on Float32 vectors (round 3) -----------------------------------------------------------------------------
namespace {
template <class Op>
int run_iprox_unboxed_f32(spx_ctx* ctx, float* y, const float* g, const float* d, const float* xk, const float* sj, int64_t n,
                          float lambda, int check_d) {
  int rc = spx_check_common(ctx, y, g, xk, sj, n);
  if (rc) return rc;
  SPX_REQUIRE(n == 0 || d != nullptr, "d is NULL");
  if (n == 0) return SPX_OK;
  if (check_d) { rc = spx_require_not_capturing(ctx, "the d > 0 check of iprox! (pass check = 0)"); if (rc) return rc; }
  rc = spx_ws_reserve(ctx, 256);
  if (rc) return rc;
  SPX_ON_DEVICE(ctx);
  int* flag = reinterpret_cast<int*>(ctx->ws);
  { const int rz = spx_zero_async(ctx, flag, sizeof(int)); if (rz) return rz; }
  rc = run_f32(ctx, y, g, xk, sj, n, nullptr, nullptr, 0.0f, 0.0f, nullptr, Op{lambda, flag}, d);
  if (rc || !check_d) return rc;
  int bad = 0;
  SPX_HIP(hipMemcpyAsync(&bad, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  SPX_HIP(hipStreamSynchronize(ctx->stream));
  if (bad) {
    spx_set_error("AssertionError: d[i] > 0");
    return SPX_ERR_ASSERT;
  }
  return SPX_OK;
}
}  // namespace

SPX_EXPORT int spx_iprox_l1_f32(spx_ctx* ctx, float* y, const float* g, const float* d, const float* xk, const float* sj,
                                int64_t n, float lambda, int check_d) {
  return run_iprox_unboxed_f32<F32IproxL1>(ctx, y, g, d, xk, sj, n, lambda, check_d);
}
SPX_EXPORT int spx_iprox_l0_f32(spx_ctx* ctx, float* y, const float* g, const float* d, const float* xk, const float* sj,
                                int64_t n, float lambda, int check_d) {
  return run_iprox_unboxed_f32<F32IproxL0>(ctx, y, g, d, xk, sj, n, lambda, check_d);
}
SPX_EXPORT int spx_iprox_l1_box_f32(spx_ctx* ctx, float* y, const float* g, const float* d, const float* xk, const float* sj,
                                    int64_t n, float lambda, const float* l_vec, const float* u_vec, float l_scalar,
                                    float u_scalar, const uint8_t* sel_mask) {
  int rc = spx_check_common(ctx, y, g, xk, sj, n);
  if (rc) return rc;
  SPX_REQUIRE(n == 0 || d != nullptr, "d is NULL");
  return run_f32(ctx, y, g, xk, sj, n, l_vec, u_vec, l_scalar, u_scalar, sel_mask, F32IproxL1Box{lambda}, d);
}
SPX_EXPORT int spx_iprox_l0_box_f32(spx_ctx* ctx, float* y, const float* g, const float* d, const float* xk, const float* sj,
                                    int64_t n, float lambda, const float* l_vec, const float* u_vec, float l_scalar,
                                    float u_scalar, const uint8_t* sel_mask) {
  int rc = spx_check_common(ctx, y, g, xk, sj, n);
  if (rc) return rc;
  SPX_REQUIRE(n == 0 || d != nullptr, "d is NULL");
  return run_f32(ctx, y, g, xk, sj, n, l_vec, u_vec, l_scalar, u_scalar, sel_mask, F32IproxL0Box{lambda}, d);
}
