// spx_common.hpp -- shared host/device helpers of libspx (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdarg>
#include <cstdio>

#include "spx.h"

#define SPX_EXPORT extern "C" __attribute__((visibility("default")))

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
struct spx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool owns_stream = false;
  int num_cu = 256;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  // library-owned scratch (top-r selection state, flags); grown on demand, never shrunk
  void* ws = nullptr;
  size_t ws_bytes = 0;
  // device staging area of the host-pointer entry points (spx_host.hip); grown on demand, never shrunk
  void* stage = nullptr;
  size_t stage_bytes = 0;
  // ShiftedNormL1B2: did the last call take the scaled branch (trust region active)?  Only steers whether the first
  // reduction pass also stores y (it is the result when the trust region is inactive); never affects results.
  int b2_last_scaled = 0;
};

void spx_set_error(const char* fmt, ...);
int spx_ws_reserve(spx_ctx* ctx, size_t bytes);

#define SPX_HIP(call)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      spx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return SPX_ERR_HIP;                                                                      \
    }                                                                                          \
  } while (0)

#define SPX_REQUIRE(cond, msg)                 \
  do {                                         \
    if (!(cond)) {                             \
      spx_set_error("invalid argument: %s", msg); \
      return SPX_ERR_INVALID_ARG;              \
    }                                          \
  } while (0)

#define SPX_LAUNCH_CHECK()                                                   \
  do {                                                                       \
    hipError_t e_ = hipGetLastError();                                       \
    if (e_ != hipSuccess) {                                                  \
      spx_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
      return SPX_ERR_HIP;                                                    \
    }                                                                        \
  } while (0)

static inline int spx_check_common(spx_ctx* ctx, const void* y, const void* q, const void* xk,
                                   const void* sj, int64_t n) {
  SPX_REQUIRE(ctx != nullptr, "ctx is NULL");
  SPX_REQUIRE(n >= 0, "n < 0");
  if (n > 0) SPX_REQUIRE(y && q && xk && sj, "NULL vector with n > 0");
  return SPX_OK;
}

static inline bool spx_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---------------------------------------------------------------------------------------------
// device math with the reference's (Julia Base) semantics
// ---------------------------------------------------------------------------------------------
typedef double f64x2 __attribute__((ext_vector_type(2)));

// Base.min / Base.max on Float64: the sign of x - y picks the argument, so -0.0 < +0.0;
// a NaN in either argument propagates (IEEE-754-2019 minimum / maximum).
__device__ __forceinline__ double jl_min(double x, double y) {
  double d = x - y;
  double a = (__double_as_longlong(d) < 0) ? x : y;
  return (x != x || y != y) ? d : a;
}
__device__ __forceinline__ double jl_max(double x, double y) {
  double d = x - y;
  double a = (__double_as_longlong(d) < 0) ? y : x;
  return (x != x || y != y) ? d : a;
}
// Base.sign: +-1.0, or x itself for +-0.0 / NaN
__device__ __forceinline__ double jl_sign(double x) { return (x > 0.0) ? 1.0 : (x < 0.0) ? -1.0 : x; }
// prox_zero(q, l, u) = min(max(q, l), u)   src/ShiftedProximalOperators.jl:203
__device__ __forceinline__ double prox_zero(double q, double l, double u) { return jl_min(jl_max(q, l), u); }

// sum over the 64 lanes of a wavefront, result in every lane (xor butterfly; DPP/permute lowered by hipcc)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
