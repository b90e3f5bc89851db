// spx_b2.hip -- ShiftedNormL1B2.prox!  (src/shiftedNormL1B2.jl:50-67): l1 norm + l2-ball trust region
// (SURVEY.md 8f rank 4).  Unlike the separable operators this one couples all elements through one scalar:
//
//   ProjB(z)_i = min(max(z_i, (sj_i + q_i) - lambda sigma), (sj_i + q_i) + lambda sigma)
//   y = ProjB(-xk);   if Delta <= chi(y):  eta = find_zero(froot, Delta),  froot(eta) = eta - chi(ProjB((-xk) eta/Delta)),
//                                          y = ProjB((-xk) (eta/Delta)) (Delta/eta);      y -= sj          chi = chi_lambda ||.||_2
//
// |ProjB((-xk) eta/Delta)_i| / eta is non-increasing in eta for every i, so froot(eta)/eta is non-decreasing: the sign
// change the reference's find_zero (Roots.jl, [ext]) converges to is unique.  On a fixed set of clamped components
// ||ProjB||^2 = (eta/Delta)^2 P + C with P = sum_{unclamped} xk_i^2, C = sum_{clamped} bound_i^2, whose root is closed
// form: eta = chi_lambda sqrt(C / (1 - chi_lambda^2 P / Delta^2)).  Iteration: one reduction pass gives (P, C) at the
// current eta, the piece's root is taken, the next pass verifies it (identical sums = same piece = done); a bracket
// [froot < 0, froot > 0] safeguards every step.  Typically 2-3 passes of 24 B/element + the final 32 B/element pass.
// The host drives the loop (one 16-byte read-back per pass).
#include <cmath>

#include "spx_common.hpp"

namespace {

constexpr int kB2Blocks = 2048;

struct B2Ws {
  double partP[kB2Blocks];
  double partC[kB2Blocks];
  double P, C;  // reduced sums of the last pass
};

__device__ __forceinline__ double b2_block_sum(double v, double* lds4) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds4[w] = v;
  __syncthreads();
  return (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
}

// sums at scale r = eta / Delta: P over the components ProjB leaves at z_i = -xk_i r, C over the clamped ones
__global__ __launch_bounds__(256) void k_b2_pass(const double* __restrict__ q, const double* __restrict__ xk,
                                                  const double* __restrict__ sj, int64_t n, double ls, double r,
                                                  B2Ws* ws) {
  __shared__ double lds4[4];
  double p = 0.0, c = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double x = xk[i];
    const double sq = sj[i] + q[i];
    const double lo = sq - ls, hi = sq + ls;
    const double z = (-x) * r;
    const double pz = jl_min(jl_max(z, lo), hi);
    if (pz == z) p += x * x; else c += pz * pz;
  }
  p = b2_block_sum(p, lds4);
  c = b2_block_sum(c, lds4);
  if (threadIdx.x == 0) { ws->partP[blockIdx.x] = p; ws->partC[blockIdx.x] = c; }
}

__global__ __launch_bounds__(256) void k_b2_reduce(B2Ws* ws, int nblocks) {
  __shared__ double lds4[4];
  double p = 0.0, c = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) { p += ws->partP[b]; c += ws->partC[b]; }
  p = b2_block_sum(p, lds4);
  c = b2_block_sum(c, lds4);
  if (threadIdx.x == 0) { ws->P = p; ws->C = c; }
}

// y = ProjB((-xk) r) * rinv - sj     (r = eta/Delta, rinv = Delta/eta; r = rinv = 1 with scaled == 0: y = ProjB(-xk) - sj)
__global__ __launch_bounds__(256) void k_b2_final(double* y, const double* q, const double* xk, const double* sj,
                                                   int64_t n, double ls, double r, double rinv, int scaled) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double x = xk[i], s = sj[i];
    const double sq = s + q[i];
    const double lo = sq - ls, hi = sq + ls;
    double t;
    if (scaled) t = jl_min(jl_max((-x) * r, lo), hi) * rinv;  // :63
    else t = jl_min(jl_max(-x, lo), hi);                      // :59
    y[i] = t - s;                                             // :65
  }
}

int b2_sums(spx_ctx* ctx, const double* q, const double* xk, const double* sj, int64_t n, double ls, double r, B2Ws* ws,
            int blocks, double* P, double* C) {
  hipLaunchKernelGGL(k_b2_pass, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, q, xk, sj, n, ls, r, ws);
  hipLaunchKernelGGL(k_b2_reduce, dim3(1), dim3(256), 0, ctx->stream, ws, blocks);
  SPX_LAUNCH_CHECK();
  double pc[2];
  SPX_HIP(hipMemcpyAsync(pc, &ws->P, sizeof(pc), hipMemcpyDeviceToHost, ctx->stream));
  SPX_HIP(hipStreamSynchronize(ctx->stream));
  *P = pc[0];
  *C = pc[1];
  return SPX_OK;
}

}  // namespace

SPX_EXPORT int spx_prox_l1_b2(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                              double lambda, double sigma, double delta, double chi_lambda) {
  int rc = spx_check_common(ctx, y, q, xk, sj, n);
  if (rc) return rc;
  if (n == 0) return SPX_OK;
  rc = spx_ws_reserve(ctx, sizeof(B2Ws) + 256);
  if (rc) return rc;
  SPX_HIP(hipSetDevice(ctx->device));
  B2Ws* ws = reinterpret_cast<B2Ws*>(ctx->ws);
  const double ls = lambda * sigma;  // `psi.lambda * sigma`, :56
  int64_t blocks = (n + 256 * 8 - 1) / (256 * 8);
  if (blocks > kB2Blocks) blocks = kB2Blocks;
  int64_t fblocks = (n + 255) / 256;
  if (fblocks > (int64_t)ctx->num_cu * 16) fblocks = (int64_t)ctx->num_cu * 16;
  // y = ProjB(-xk); chi(y) = chi_lambda * ||y||: at r = 1,  ||y||^2 = P + C
  double P, C;
  rc = b2_sums(ctx, q, xk, sj, n, ls, 1.0, ws, (int)blocks, &P, &C);
  if (rc) return rc;
  const double chiy = chi_lambda * std::sqrt(P + C);
  int scaled = 0;
  double eta = delta;
  if (delta <= chiy) {  // :61
    scaled = 1;
    // froot(eta) = eta - chi_lambda sqrt((eta/Delta)^2 P + C); froot(Delta) <= 0 here.  Bracket lo: froot <= 0, hi: froot > 0.
    double lo = delta, hi = INFINITY;
    double pP = -1.0, pC = -1.0;
    bool exact_step = false;  // eta was set to the exact root of the piece (pP, pC)
    for (int it = 0; it < 200; ++it) {
      const double r = eta / delta;
      const double f = eta - chi_lambda * std::sqrt(r * r * P + C);
      if (f == 0.0 || (exact_step && P == pP && C == pC)) break;  // exact hit / the piece just solved is confirmed
      if (f < 0.0) lo = eta; else hi = eta;
      // root of the current piece
      const double den = 1.0 - chi_lambda * chi_lambda * P / (delta * delta);
      double next = (den > 0.0) ? chi_lambda * std::sqrt(C / den) : INFINITY;
      exact_step = (next > lo && next < hi);
      if (!exact_step) next = std::isinf(hi) ? 2.0 * lo : 0.5 * (lo + hi);
      if (!(next > lo && next < hi)) break;  // bracket exhausted
      const bool small = std::fabs(next - eta) <= 4e-16 * next;
      pP = P; pC = C; eta = next;
      if (small) break;
      rc = b2_sums(ctx, q, xk, sj, n, ls, eta / delta, ws, (int)blocks, &P, &C);
      if (rc) return rc;
    }
  }
  hipLaunchKernelGGL(k_b2_final, dim3((unsigned)fblocks), dim3(256), 0, ctx->stream, y, q, xk, sj, n, ls, eta / delta,
                     delta / eta, scaled);
  SPX_LAUNCH_CHECK();
  return SPX_OK;
}
