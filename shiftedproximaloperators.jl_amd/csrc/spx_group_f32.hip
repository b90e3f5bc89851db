// spx_group_f32.hip -- ShiftedGroupNormL2.prox! on Float32 vectors (round 3 widening).
//
// The reference's method is generic in R (src/shiftedGroupNormL2.jl:52-79): with R = Float32
//     sol = (q + xk) + sj;   per group: snorm = norm(sol[idx]);  y[idx] = snorm == 0 ? 0 : max(1 - sigma lambda / snorm, 0) sol[idx];
//     y -= xk + sj
// every elementwise operation is a Float32 operation; `norm` of a Float32 vector is BLAS snrm2 / a scaled generic loop [ext],
// whose accumulation order is not pinned -- here the squares are summed in Float64 and the norm rounded to Float32 once (within
// an ulp of the exact Float32 norm); parity is therefore to a few Float32 ulps of the operands' scale, not bits.
// Contiguous groups (uniform size or CSR offsets), 16 B/element.  1..64 lanes per group, the group re-read from L1 / L2 for
// the second pass: a generality path, not the tuned register-tile kernels of the Float64 form (spx_group.hip).
// ShiftedGroupNormL2Binf has NO Float32 form: its root find would run `fzero` in Float32, whose result next to the pole of
// step(n) is rounding noise in the reference itself.
#include "spx_common.hpp"

namespace {

// TEAM lanes of a wavefront per group (1..64 by the group size, about four elements per lane: a whole wavefront per group left
// most lanes idle on small groups), 64 / TEAM groups per wavefront and trip.
template <int TEAM>
__global__ __launch_bounds__(256) void k_group_l2_f32(float* y, const float* q, const float* xk, const float* sj, int64_t n,
                                                       const int64_t* __restrict__ offsets, int64_t gsize, int64_t ngroups,
                                                       const float* __restrict__ lambda, float sigma) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  constexpr int GPW = 64 / TEAM;
  const int j = lane % TEAM, slot = lane / TEAM;
  for (int64_t g0 = wave * GPW; g0 < ngroups; g0 += nwaves * GPW) {  // wave-uniform
    const int64_t g = g0 + slot;
    const bool live = g < ngroups;
    int64_t lo = 0, hi = 0;
    if (live) {
      if (offsets) { lo = offsets[g]; hi = offsets[g + 1]; }
      else { lo = g * gsize; hi = lo + gsize; }
    }
    if (lo < 0) lo = 0;
    if (hi > n) hi = n;
    double ss = 0.0;
    for (int64_t i = lo + j; i < hi; i += TEAM) {
      const float S = (q[i] + xk[i]) + sj[i];  // :65
      ss += (double)S * (double)S;
    }
#pragma unroll
    for (int off = TEAM / 2; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);  // (inside the team's aligned lane range)
    const float snorm = (float)sqrt(ss);                                 // :69
    const float alpha = (snorm == 0.0f || !live) ? 0.0f : jl_max(1 - sigma * lambda[g] / snorm, 0.0f);  // :70-73
    // every lane's reads of q for the norm are done (the shuffles above) before any store of this group: y may alias q; the
    // storing lane re-reads q[i] itself just before it writes y[i]
    for (int64_t i = lo + j; i < hi; i += TEAM) {
      const float x = xk[i], s = sj[i];
      const float S = (q[i] + x) + s;
      y[i] = ((snorm == 0.0f) ? 0.0f : alpha * S) - (x + s);             // :74, :77
    }
  }
}

// indices before offsets[0] / from offsets[ngroups] on: y - (xk + sj)   (:77 runs over every index)
__global__ __launch_bounds__(256) void k_csr_uncovered_f32(float* y, const float* xk, const float* sj,
                                                            const int64_t* __restrict__ offsets, int64_t ngroups, int64_t n) {
  int64_t head = offsets[0], tail0 = offsets[ngroups];
  if (head < 0) head = 0;
  if (head > n) head = n;
  if (tail0 < head) tail0 = head;
  if (tail0 > n) tail0 = n;
  const int64_t total = head + (n - tail0);
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t i = (t < head) ? t : tail0 + (t - head);
    y[i] = y[i] - (xk[i] + sj[i]);
  }
}

}  // namespace

SPX_EXPORT int spx_prox_group_l2_f32(spx_ctx* ctx, float* y, const float* q, const float* xk, const float* sj, int64_t n,
                                     const int64_t* group_offsets, int64_t group_size, int64_t ngroups,
                                     const float* lambda_vec, float sigma) {
  int rc = spx_check_common(ctx, y, q, xk, sj, n);
  if (rc) return rc;
  SPX_REQUIRE(ngroups >= 0, "ngroups < 0");
  if (n == 0) return SPX_OK;
  SPX_ON_DEVICE(ctx);
  if (group_offsets) {
    hipLaunchKernelGGL(k_csr_uncovered_f32, dim3(256), dim3(256), 0, ctx->stream, y, xk, sj, group_offsets, ngroups, n);
    SPX_LAUNCH_CHECK();
  }
  if (ngroups == 0) return SPX_OK;
  SPX_REQUIRE(lambda_vec != nullptr, "lambda_vec is NULL");
  if (!group_offsets) {
    SPX_REQUIRE(group_size > 0, "group_size <= 0 with NULL group_offsets");
    SPX_REQUIRE(ngroups <= n / group_size && ngroups * group_size == n, "ngroups * group_size != n");
  }
  const int64_t typical = group_size > 0 ? group_size : (n + ngroups - 1) / ngroups;
  int team = 1;
  while (team < 64 && (int64_t)team * 4 < typical) team *= 2;
  const int gpw = 64 / team;
  int64_t blocks = (ngroups + 4 * gpw - 1) / (4 * gpw);
  const int64_t cap = (int64_t)ctx->num_cu * 16;
  if (blocks > cap) blocks = cap;
#define SPX_GROUP_F32(TEAM)                                                                                             \
  hipLaunchKernelGGL(k_group_l2_f32<TEAM>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, y, q, xk, sj, n, group_offsets, \
                     group_size, ngroups, lambda_vec, sigma)
  switch (team) {
    case 1: SPX_GROUP_F32(1); break;
    case 2: SPX_GROUP_F32(2); break;
    case 4: SPX_GROUP_F32(4); break;
    case 8: SPX_GROUP_F32(8); break;
    case 16: SPX_GROUP_F32(16); break;
    case 32: SPX_GROUP_F32(32); break;
    default: SPX_GROUP_F32(64); break;
  }
#undef SPX_GROUP_F32
  SPX_LAUNCH_CHECK();
  return SPX_OK;
}
