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
// [froot < 0, froot > 0] safeguards every step.  The iteration starts from the a-priori upper bound chi sqrt(F)
// (F = sum of the far bounds squared, from the first pass): typically 4-5 passes of 24 B/element + the final 32 B/element pass.
// Round 1: the host drives the loop (one 16-byte read-back per pass) -- kept for unaligned views and as the A/B baseline
// (spx_ctx_set_tuning key 7 = 0).  Round 2: k_b2_coop runs the same iteration inside one launch.
#include <cmath>

#include "spx_common.hpp"

namespace {

constexpr int kB2Blocks = 2048;

struct B2Ws {
  double partP[kB2Blocks];
  double partC[kB2Blocks];
  double partF[kB2Blocks];
  double P, C, F;  // reduced sums of the last pass
};

__device__ __forceinline__ double b2_block_sum(double v, double* lds4) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds4[w] = v;
  __syncthreads();
  return (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
}

// sums at scale r = eta / Delta: P over the components ProjB leaves at z_i = -xk_i r, C over the clamped ones.
// FIRST (the r = 1 pass) also returns F = sum of bound_i^2 with bound_i the end of [lo_i, hi_i] that z_i runs into as
// eta -> inf: |ProjB(z)_i| <= |bound_i| for every eta >= 0, so chi_lambda sqrt(F) bounds the root from above.
// VEC: 16-byte non-temporal loads, 4 pairs of each vector in flight per lane; else 8-byte loads (unaligned views).
// ywrite != NULL (vector path, y disjoint from the inputs): the pass also stores y = ProjB(z) rinv - sj for ITS scale
// (:63, :65) -- if the iteration then stops at this very eta the separate final pass is not needed.
template <bool FIRST, bool VEC>
__global__ __launch_bounds__(256) void k_b2_pass(const double* __restrict__ q, const double* __restrict__ xk,
                                                  const double* __restrict__ sj, int64_t n, double ls, double r,
                                                  B2Ws* ws, double* ywrite, double rinv, int head) {
  // head = 1 (VEC only): the caller's vectors start 8 bytes off a 16-byte boundary (all alike); the pointers are the
  // aligned rest, n counts it, and the caller's element 0 sits at [-1] (taken along by one lane of workgroup 0)
  __shared__ double lds4[4];
  double p = 0.0, c = 0.0, f = 0.0;
  auto visit = [&](double qi, double x, double s) -> double {
    const double sq = s + qi;
    const double lo = sq - ls, hi = sq + ls;
    const double z = (-x) * r;
    const double pz = jl_min(jl_max(z, lo), hi);
    if (pz == z) p += x * x; else c += pz * pz;
    if constexpr (FIRST) {
      const double far = (x < 0.0) ? hi : (x > 0.0) ? lo : pz;  // -x > 0: z -> +inf -> hi
      f += far * far;
    }
    return pz * rinv - s;
  };
  if constexpr (VEC) {
    const f64x2* q2 = reinterpret_cast<const f64x2*>(q);
    const f64x2* x2 = reinterpret_cast<const f64x2*>(xk);
    const f64x2* s2 = reinterpret_cast<const f64x2*>(sj);
    const int64_t n2 = n >> 1;
    const int64_t ntiles = (n2 + 1023) / 1024;  // 256 lanes x 4 pairs
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const int64_t base = tile * 1024 + threadIdx.x;
      f64x2 a[4], b[4], d[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t i = (base + k * 256 < n2) ? base + k * 256 : n2 - 1;
        a[k] = __builtin_nontemporal_load(q2 + i);
        b[k] = __builtin_nontemporal_load(x2 + i);
        d[k] = __builtin_nontemporal_load(s2 + i);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (base + k * 256 < n2) {
          f64x2 o;
          o.x = visit(a[k].x, b[k].x, d[k].x);
          o.y = visit(a[k].y, b[k].y, d[k].y);
          if (ywrite) __builtin_nontemporal_store(o, reinterpret_cast<f64x2*>(ywrite) + base + k * 256);
        }
      }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
      const double o = visit(q[n - 1], xk[n - 1], sj[n - 1]);
      if (ywrite) ywrite[n - 1] = o;
    }
    if (head && blockIdx.x == 0 && threadIdx.x == 64) {
      const double o = visit(q[-1], xk[-1], sj[-1]);
      if (ywrite) ywrite[-1] = o;
    }
  } else {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) visit(q[i], xk[i], sj[i]);
  }
  p = b2_block_sum(p, lds4);
  c = b2_block_sum(c, lds4);
  if constexpr (FIRST) f = b2_block_sum(f, lds4);
  if (threadIdx.x == 0) {
    ws->partP[blockIdx.x] = p;
    ws->partC[blockIdx.x] = c;
    if constexpr (FIRST) ws->partF[blockIdx.x] = f;
  }
}

__global__ __launch_bounds__(256) void k_b2_reduce(B2Ws* ws, int nblocks, int first) {
  __shared__ double lds4[4];
  double p = 0.0, c = 0.0, f = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) {
    p += ws->partP[b];
    c += ws->partC[b];
    if (first) f += ws->partF[b];
  }
  p = b2_block_sum(p, lds4);
  c = b2_block_sum(c, lds4);
  f = b2_block_sum(f, lds4);
  if (threadIdx.x == 0) { ws->P = p; ws->C = c; ws->F = f; }
}

// y = ProjB((-xk) r) * rinv - sj     (r = eta/Delta, rinv = Delta/eta; r = rinv = 1 with scaled == 0: y = ProjB(-xk) - sj)
template <bool VEC>
__global__ __launch_bounds__(256) void k_b2_final(double* y, const double* q, const double* xk, const double* sj,
                                                   int64_t n, double ls, double r, double rinv, int scaled, int head) {
  auto out = [&](double qi, double x, double s) -> double {
    const double sq = s + qi;
    const double lo = sq - ls, hi = sq + ls;
    double t;
    if (scaled) t = jl_min(jl_max((-x) * r, lo), hi) * rinv;  // :63
    else t = jl_min(jl_max(-x, lo), hi);                      // :59
    return t - s;                                             // :65
  };
  if constexpr (VEC) {
    f64x2* y2 = reinterpret_cast<f64x2*>(y);
    const f64x2* q2 = reinterpret_cast<const f64x2*>(q);
    const f64x2* x2 = reinterpret_cast<const f64x2*>(xk);
    const f64x2* s2 = reinterpret_cast<const f64x2*>(sj);
    const int64_t n2 = n >> 1;
    const int64_t base = (int64_t)blockIdx.x * 1024 + threadIdx.x;  // one tile of 256 lanes x 4 pairs per workgroup
    f64x2 a[4], b[4], d[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int64_t i = (base + k * 256 < n2) ? base + k * 256 : n2 - 1;
      a[k] = __builtin_nontemporal_load(q2 + i);
      b[k] = __builtin_nontemporal_load(x2 + i);
      d[k] = __builtin_nontemporal_load(s2 + i);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (base + k * 256 < n2)
        __builtin_nontemporal_store(f64x2{out(a[k].x, b[k].x, d[k].x), out(a[k].y, b[k].y, d[k].y)}, y2 + base + k * 256);
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) y[n - 1] = out(q[n - 1], xk[n - 1], sj[n - 1]);
    if (head && blockIdx.x == 0 && threadIdx.x == 64) y[-1] = out(q[-1], xk[-1], sj[-1]);  // (as k_b2_pass)
  } else {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] = out(q[i], xk[i], sj[i]);
  }
}

int b2_sums(spx_ctx* ctx, const double* q, const double* xk, const double* sj, int64_t n, double ls, double r, B2Ws* ws,
            int blocks, bool vec, bool first, double* P, double* C, double* F, double* ywrite = nullptr,
            double rinv = 1.0, int head = 0) {
  const dim3 grid((unsigned)blocks), block(256);
  double* yw = vec ? ywrite : nullptr;
  if (first) {
    if (vec) hipLaunchKernelGGL((k_b2_pass<true, true>), grid, block, 0, ctx->stream, q, xk, sj, n, ls, r, ws, yw, rinv, head);
    else hipLaunchKernelGGL((k_b2_pass<true, false>), grid, block, 0, ctx->stream, q, xk, sj, n, ls, r, ws, yw, rinv, head);
  } else {
    if (vec) hipLaunchKernelGGL((k_b2_pass<false, true>), grid, block, 0, ctx->stream, q, xk, sj, n, ls, r, ws, yw, rinv, head);
    else hipLaunchKernelGGL((k_b2_pass<false, false>), grid, block, 0, ctx->stream, q, xk, sj, n, ls, r, ws, yw, rinv, head);
  }
  hipLaunchKernelGGL(k_b2_reduce, dim3(1), dim3(256), 0, ctx->stream, ws, blocks, first ? 1 : 0);
  SPX_LAUNCH_CHECK();
  double pcf[3];
  { const int rcc = spx_require_not_capturing(ctx, "the host-driven ShiftedNormL1B2 iteration (spx_ctx_set_tuning key 7 = 0 or a mixed alignment)"); if (rcc) return rcc; }
  SPX_HIP(hipMemcpyAsync(pcf, &ws->P, sizeof(pcf), hipMemcpyDeviceToHost, ctx->stream));
  SPX_HIP(hipStreamSynchronize(ctx->stream));
  *P = pcf[0];
  *C = pcf[1];
  if (F) *F = pcf[2];
  return SPX_OK;
}

// psi(y) = lambda ||xk + sj + y||_1 + IndBallL2(Delta)(sj + y)   (:32): P = sum |(xk + sj) + y|, C = sum (sj + y)^2
template <bool VEC>
__global__ __launch_bounds__(256) void k_b2_obj(const double* __restrict__ y, const double* __restrict__ xk,
                                                 const double* __restrict__ sj, int64_t n, B2Ws* ws) {
  __shared__ double lds4[4];
  double p = 0.0, c = 0.0;
  auto visit = [&](double yi, double x, double s) {
    p += fabs((x + s) + yi);
    const double t = s + yi;
    c += t * t;
  };
  if constexpr (VEC) {
    const f64x2* y2 = reinterpret_cast<const f64x2*>(y);
    const f64x2* x2 = reinterpret_cast<const f64x2*>(xk);
    const f64x2* s2 = reinterpret_cast<const f64x2*>(sj);
    const int64_t n2 = n >> 1;
    const int64_t ntiles = (n2 + 1023) / 1024;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const int64_t base = tile * 1024 + threadIdx.x;
      f64x2 a[4], b[4], d[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t i = (base + k * 256 < n2) ? base + k * 256 : n2 - 1;
        a[k] = __builtin_nontemporal_load(y2 + i);
        b[k] = __builtin_nontemporal_load(x2 + i);
        d[k] = __builtin_nontemporal_load(s2 + i);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if (base + k * 256 < n2) {
          visit(a[k].x, b[k].x, d[k].x);
          visit(a[k].y, b[k].y, d[k].y);
        }
      }
    }
    if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) visit(y[n - 1], xk[n - 1], sj[n - 1]);
  } else {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) visit(y[i], xk[i], sj[i]);
  }
  p = b2_block_sum(p, lds4);
  c = b2_block_sum(c, lds4);
  if (threadIdx.x == 0) { ws->partP[blockIdx.x] = p; ws->partC[blockIdx.x] = c; }
}

// =============================================================================================
// The whole prox in ONE launch (round 2).  The host loop above pays one stream synchronisation per reduction pass: ~100 us
// per call at n = 1e4 and n = 1e6 alike, and 4-5 round trips inside the 1.35 ms at n = 1e8.  Here the workgroups of a
// resident grid (<= number of CUs x 1024 lanes) run the same iteration themselves: every pass ends in per-workgroup partial
// sums (written, not accumulated: the order of every addition is fixed, so all workgroups form bit-identical P, C, F),
// a grid barrier, and the scalar update of eta, which every lane redoes for itself.  REG: n <= 8 Ki x number of CUs --
// xk, sj and sj + q stay in registers, the vectors are read once.  !REG: the passes stream from memory (16-byte pairs).
// =============================================================================================
// register-resident form: 512 lanes x 16 elements per workgroup (256 VGPRs per lane).  1024 lanes x 8 spill under their 128
// VGPRs (loop invariants the compiler keeps per element: the box ends, the store addresses), and the first touch of a wave's
// scratch memory costs microseconds at the start of every launch; 1024 x 16 spill 120 registers and run 2x slower.
constexpr int kB2Epl = 16;
constexpr int kB2RegThreads = 512;
constexpr int kB2RegBlock = kB2Epl * kB2RegThreads;  // elements per workgroup of the register-resident form
constexpr int kB2MaxPass = 64;
// One workgroup's partial sums of a pass: 8 words (p, c, f, p1, c1, 3 unused) in spx_ctx::sync, at a FIXED place (set, pass,
// workgroup), written once per launch.  A word is its own "ready" flag: the sync area starts out zero, a sum v is stored as
// bits(v) + 1 (never 0: NaNs are canonicalised first), and a reader polls the word until it is non-zero -- no counter, no
// rendezvous, one memory round trip per pass instead of four (store + wait, arrive, poll, load: 2.6-3.6 -> ~1.x us per
// pass, which is most of a call at n <= 1e6).  Launches alternate between two sets; a launch zeroes what the launch before
// the previous one left in the other set (the host knows how many workgroups that was).
constexpr int kB2Cols = 256;   // workgroups at most
constexpr int kB2Words = 8;
constexpr size_t kB2SetWords = (size_t)kB2MaxPass * kB2Cols * kB2Words;
constexpr size_t kB2SyncBytes = 2 * kB2SetWords * sizeof(unsigned long long);   // 2 MiB, behind the select state
__device__ __forceinline__ void b2_put(unsigned long long* slot, double v) {
  const unsigned long long b = (v != v) ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(v);
  __hip_atomic_store(slot, b + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (an atomic swap instead: no faster)
}
__device__ __forceinline__ void b2_block_sum5(double& a, double& b, double& c, double& d, double& e, double (*lds)[16]) {
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c); d = wave_sum(d); e = wave_sum(e);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { lds[0][w] = a; lds[1][w] = b; lds[2][w] = c; lds[3][w] = d; lds[4][w] = e; }
  __syncthreads();
  double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0, t4 = 0.0;
#pragma unroll
  for (int k = 0; k < 16; ++k) { t0 += lds[0][k]; t1 += lds[1][k]; t2 += lds[2][k]; t3 += lds[3][k]; t4 += lds[4][k]; }
  a = t0; b = t1; c = t2; d = t3; e = t4;
}

// REG: n <= kB2RegBlock * grid, the vectors in registers.  !REG (streaming): additionally a SAMPLE of one element per lane
// of the grid (1024 chunks of 256 = 262 144 elements on 256 CUs, kept in registers) is solved first -- a handful of
// rendezvous, no streaming -- and its root eta_s (good to ~1/sqrt(262144) = 2e-3) rides along as a SECOND TRIAL in the first
// streaming pass (the loads dominate: two sets of sums cost nothing).  The piece root of that trial is then ~1e-6 from the
// root, one more reduction pass brings ~1e-13, and the storing pass follows: 24 + 24 + 32 = 80 B/element instead of
// 4 x 24 + 32 = 128 when the iteration starts from the a-priori bound (which stays the fallback whenever the sample
// misleads).  Stopping rule: the measured quadratic constant K = step_k / step_{k-1}^2 predicts error(next) = K step_k^2.
#ifdef SPX_B2_PROFILE  // A/B builds only: time stamps of workgroup 0 (10 ns units), read with spx_debug_b2_stamps
__device__ unsigned long long g_b2_stamp[64];
__device__ int g_b2_nstamp;
#define B2_STAMP() do { if (blockIdx.x == 0 && threadIdx.x == 0 && nst < 64) g_b2_stamp[nst++] = wall_clock64(); } while (0)
extern "C" __attribute__((visibility("default"))) int spx_debug_b2_stamps(unsigned long long* out, int* count) {
  hipError_t e = hipMemcpyFromSymbol(out, HIP_SYMBOL(g_b2_stamp), sizeof(g_b2_stamp));
  if (e == hipSuccess) e = hipMemcpyFromSymbol(count, HIP_SYMBOL(g_b2_nstamp), sizeof(int));
  return (int)e;
}
#else
#define B2_STAMP() do { } while (0)
#endif

template <bool REG, int EPL, int THREADS>
__global__ __launch_bounds__(THREADS) void k_b2_coop(double* y, const double* q, const double* xk, const double* sj, int64_t n,
                                                   double ls, double delta, double chil, unsigned long long* rows,
                                                   unsigned long long* clear_rows, int clear_g, SpxSyncHeader* hdr,
                                                   int can_spec) {
  __shared__ double lds5[5][16];
  const int t = threadIdx.x;
  const int G = (int)gridDim.x;
  const int64_t NT = (int64_t)G * blockDim.x;
  const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + t;
  // the other set, for the launch after this one: passes x (workgroups of the launch that used it) x words, plain stores
  for (int64_t idx = gtid; idx < (int64_t)kB2MaxPass * clear_g * kB2Words; idx += NT) {
    const int64_t pass_i = idx / (clear_g * kB2Words), rem = idx % (clear_g * kB2Words);
    clear_rows[pass_i * kB2Cols * kB2Words + rem] = 0ull;
  }
  if (t < 80) (&lds5[0][0])[t] = 0.0;  // (workgroups of fewer than 16 wavefronts leave the upper slots alone)
  int nst = 0;
  B2_STAMP();
  const int last_scaled = hdr->b2_last_scaled;  // (written by the previous call's launch)
  // (sj itself is needed only where a y is stored: reloaded there, so that 16 elements per lane fit the 128 VGPRs)
  double X[REG ? EPL : 1], LO[REG ? EPL : 1], HI[REG ? EPL : 1];  // xk and the box (sj + q) -+ lambda sigma of each element
  if constexpr (REG) {
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      // clamped index, unconditional loads: all 3 EPL loads in flight at once (a guarded load per element compiles to EPL
      // dependent branch blocks, each waiting for its own loads: ~1.5 us apiece)
      const int64_t i = gtid + (int64_t)k * NT;
      const int64_t ic = i < n ? i : n - 1;
      const double xv = xk[ic], sv = sj[ic], qv = q[ic];
      const double sq = i < n ? (sv + qv) : 0.0;   // `sj .+ q` (:56)
      X[k] = i < n ? xv : 0.0;
      LO[k] = sq - ls;
      HI[k] = sq + ls;
    }
  }
  B2_STAMP();  // (register-resident form: the vectors are in)
  // the sample (streaming form only): one element per lane of the grid -- chunk c = 32 * workgroup + (t >> 5) of 32 G chunks
  // of 32 consecutive elements (256 bytes), element t & 31 of it (262 144 elements on 256 CUs: statistical error of its root
  // ~2e-3).  (Chunks of 256 elements at first: on SORTED input a chunk is 256 nearly equal values and the sample's root was
  // off by more than the second trial can absorb -- a fourth pass, 1.68 instead of 1.36 ms, tools/r2/b2_sorted_time.py.)
  const int kChunks = 32 * G;
  const int chunk = (int)blockIdx.x * 32 + (t >> 5);
  const int64_t nsample = (int64_t)kChunks * 32;
  const bool has_sample = !REG && n >= 4 * nsample;
  double sx = 0.0, ss = 0.0, ssq = 0.0;
  if (has_sample) {
    const int64_t i = (int64_t)((double)chunk * (double)(n - 32) / (double)(kChunks - 1)) + (t & 31);
    sx = xk[i]; ss = sj[i]; ssq = ss + q[i];
  }
  const f64x2* q2 = reinterpret_cast<const f64x2*>(q);
  const f64x2* x2 = reinterpret_cast<const f64x2*>(xk);
  const f64x2* s2 = reinterpret_cast<const f64x2*>(sj);
  f64x2* y2 = reinterpret_cast<f64x2*>(y);
  const int64_t n2 = n >> 1;
  int np = 0;
  double P = 0.0, C = 0.0, F = 0.0, P1 = 0.0, C1 = 0.0;
  // One reduction pass at scale r (= eta / Delta) and, if r1 > 0, at a second trial scale r1 (sums P1, C1).
  // store: also y = ProjB((-xk) r) rinv - sj for scale r.  sample: over the sample registers instead of the vectors.
  auto pass = [&](double r, double rinv, bool first, bool store, double r1, bool sample) {
    double p = 0.0, c = 0.0, f = 0.0, p1 = 0.0, c1 = 0.0;
    // The sums take v_max_f64 / v_min_f64 and masked fma operands (the Julia-semantics min / max cost six instructions each
    // and matter only for the bits of a STORED y: signed zeros, NaN propagation); a NaN operand -- which v_max / v_min would
    // drop -- is tracked separately and poisons P, as the reference's norm would be NaN.
    bool bad = false;
    auto visit = [&](double lo, double hi, double x, double s) -> double {
      const double z = (-x) * r;
      bad |= (z != z) | (lo != lo) | (hi != hi);
      const double pzf = fmin(fmax(z, lo), hi);
      const bool un = (pzf == z);
      const double xm = un ? x : 0.0, cm = un ? 0.0 : pzf;
      p = __builtin_fma(xm, xm, p);
      c = __builtin_fma(cm, cm, c);
      if (first) {
        const double far = (x < 0.0) ? hi : (x > 0.0) ? lo : pzf;
        f = __builtin_fma(far, far, f);
      }
      if (r1 > 0.0) {
        const double z1 = (-x) * r1;
        const double pz1 = fmin(fmax(z1, lo), hi);
        const bool un1 = (pz1 == z1);
        const double xm1 = un1 ? x : 0.0, cm1 = un1 ? 0.0 : pz1;
        p1 = __builtin_fma(xm1, xm1, p1);
        c1 = __builtin_fma(cm1, cm1, c1);
      }
      if (!store) return 0.0;
      const double pz = jl_min(jl_max(z, lo), hi);  // :56, bit-faithful for the stored value
      return pz * rinv - s;
    };
    if (sample) {
      if (has_sample) visit(ssq - ls, ssq + ls, sx, ss);
    } else if constexpr (REG) {
#pragma unroll
      for (int k = 0; k < EPL; ++k) {
        const int64_t i = gtid + (int64_t)k * NT;
        if (i < n) {
          const double o = visit(LO[k], HI[k], X[k], store ? sj[i] : 0.0);
          if (store) y[i] = o;
        }
        __builtin_amdgcn_sched_barrier(0);  // one element at a time: interleaved, the unrolled visits spill
      }
    } else {
      // software-pipelined stream: the loads of the next tile are issued before the current one is evaluated (two register
      // sets of 2 x 16-byte pairs per vector, ping-pong); a persistent lane otherwise serialises load latency and arithmetic
      constexpr int KP = 2;
      constexpr int64_t kTilePairs = 1024 * KP;
      const int64_t ntiles = (n2 + kTilePairs - 1) / kTilePairs;
      auto ld = [&](int64_t tile, f64x2* a, f64x2* b, f64x2* d) {
#pragma unroll
        for (int k = 0; k < KP; ++k) {
          int64_t i = tile * kTilePairs + t + k * 1024;
          if (i >= n2) i = n2 - 1;
          a[k] = __builtin_nontemporal_load(q2 + i);
          b[k] = __builtin_nontemporal_load(x2 + i);
          d[k] = __builtin_nontemporal_load(s2 + i);
        }
      };
      auto comp = [&](int64_t tile, const f64x2* a, const f64x2* b, const f64x2* d) {
#pragma unroll
        for (int k = 0; k < KP; ++k) {
          const int64_t i = tile * kTilePairs + t + k * 1024;
          if (i < n2) {
            f64x2 o;
            const double sq0 = d[k].x + a[k].x, sq1 = d[k].y + a[k].y;
            o.x = visit(sq0 - ls, sq0 + ls, b[k].x, d[k].x);
            o.y = visit(sq1 - ls, sq1 + ls, b[k].y, d[k].y);
            if (store) __builtin_nontemporal_store(o, y2 + i);
          }
        }
      };
      f64x2 a0[KP], b0[KP], d0[KP], a1[KP], b1[KP], d1[KP];
      int64_t tile = blockIdx.x;
      if (tile < ntiles) ld(tile, a0, b0, d0);
      while (tile < ntiles) {
        const int64_t t1 = tile + G;
        if (t1 < ntiles) ld(t1, a1, b1, d1);
        comp(tile, a0, b0, d0);
        const int64_t t2 = t1 + G;
        if (t1 < ntiles) {
          if (t2 < ntiles) ld(t2, a0, b0, d0);
          comp(t1, a1, b1, d1);
        }
        tile = t2;
      }
      if ((n & 1) && blockIdx.x == 0 && t == 0) {
        const double sql = sj[n - 1] + q[n - 1];
        const double o = visit(sql - ls, sql + ls, xk[n - 1], sj[n - 1]);
        if (store) y[n - 1] = o;
      }
    }
    B2_STAMP();  // visits done
    if (bad) p = __longlong_as_double(0x7ff8000000000000ll);
    b2_block_sum5(p, c, f, p1, c1, lds5);
    B2_STAMP();  // own sums
    if (G == 1) {  // one workgroup holds the whole vector: nothing to exchange
      P = p; C = c; P1 = p1; C1 = c1;
      if (first) F = f;
    } else {
      // the partial sums are the ONLY data the workgroups exchange: agent-scope atomic stores / loads (`sc1`, past the
      // non-coherent caches) of words that carry their own ready flag (see b2_put)
      unsigned long long* row = rows + (size_t)np * kB2Cols * kB2Words;
      if (t == 0) {
        unsigned long long* mine = row + (size_t)blockIdx.x * kB2Words;
        b2_put(mine + 0, p);
        b2_put(mine + 1, c);
        if (first) b2_put(mine + 2, f);
        if (r1 > 0.0) { b2_put(mine + 3, p1); b2_put(mine + 4, c1); }
      }
      double pp = 0.0, cc = 0.0, ff = 0.0, pp1 = 0.0, cc1 = 0.0;
      if (t < G) {
        const unsigned long long* theirs = row + (size_t)t * kB2Words;
        unsigned long long w0, w1, w2 = 1ull, w3 = 1ull, w4 = 1ull;
        unsigned int spins = 0;
        for (;;) {  // (every workgroup of the grid is resident and stores these words once per pass; bounded all the same)
          w0 = __hip_atomic_load(theirs + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          w1 = __hip_atomic_load(theirs + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (first) w2 = __hip_atomic_load(theirs + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (r1 > 0.0) {
            w3 = __hip_atomic_load(theirs + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            w4 = __hip_atomic_load(theirs + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
          if (w0 && w1 && w2 && w3 && w4) break;
          if (spx_wait_expired(spins, hdr)) break;  // (the sums come out as garbage / NaN: see kSpxPollLimit)
          __builtin_amdgcn_s_sleep(1);
        }
        pp = __longlong_as_double((long long)(w0 - 1ull));
        cc = __longlong_as_double((long long)(w1 - 1ull));
        if (first) ff = __longlong_as_double((long long)(w2 - 1ull));
        if (r1 > 0.0) { pp1 = __longlong_as_double((long long)(w3 - 1ull)); cc1 = __longlong_as_double((long long)(w4 - 1ull)); }
      }
      b2_block_sum5(pp, cc, ff, pp1, cc1, lds5);
      P = pp; C = cc; P1 = pp1; C1 = cc1;
      if (first) F = ff;
    }
    B2_STAMP();  // exchanged
    ++np;
  };
  // ---- the sample's root (streaming form): same iteration, chi scaled by sqrt(n / 65536), nothing stored
  double eta_s = -1.0;
  if (has_sample) {
    const double chis = chil * sqrt((double)n / (double)nsample);
    pass(1.0, 1.0, true, false, -1.0, true);
    if (delta <= chis * sqrt(P + C)) {
      double lo = delta, hi = INFINITY, pP = -1.0, pC = -1.0, eta = delta;
      bool exact_step = false;
      const double ub = chis * sqrt(F);
      if (ub > delta && ub < INFINITY) { eta = ub; pass(eta / delta, 1.0, false, false, -1.0, true); }
      for (int it = 0; it < 24; ++it) {
        const double r = eta / delta;
        const double f = eta - chis * sqrt(r * r * P + C);
        if (f == 0.0 || (exact_step && P == pP && C == pC)) break;
        if (f < 0.0) lo = eta; else hi = eta;
        const double den = 1.0 - chis * chis * P / (delta * delta);
        double next = (den > 0.0) ? chis * sqrt(C / den) : INFINITY;
        exact_step = (next > lo && next < hi);
        if (!exact_step) next = (hi == INFINITY) ? 2.0 * lo : 0.5 * (lo + hi);
        if (!(next > lo && next < hi)) break;
        if (fabs(next - eta) <= 1e-6 * next) { eta = next; break; }  // far below the sample's own statistical error
        pP = P; pC = C; eta = next;
        pass(eta / delta, 1.0, false, false, -1.0, true);
      }
      if (eta > delta && eta < INFINITY) eta_s = eta;
    }
#ifdef SPX_B2_DEBUG
    if (blockIdx.x == 0 && t == 0) printf("[b2] sample: chis %.17g P %.6g C %.6g F %.6g eta_s %.17g np %d\n", chis, P, C, F, eta_s, np);
#endif
  }
  // ---- y = ProjB(-xk) (:59); chi(y) = chi_lambda ||y||: at r = 1, ||y||^2 = P + C.  The sample's root rides along.
  const bool store_first = can_spec && !last_scaled;
  pass(1.0, 1.0, true, store_first, eta_s > 0.0 ? eta_s / delta : -1.0, false);
  const double chiy = chil * sqrt(P + C);
  // :61 `Delta <= chi(y)`.  With EQUALITY froot(Delta) = Delta - chi(y) is zero: find_zero returns its starting point Delta,
  // eta / Delta = 1 and the scaled branch reproduces y = ProjB(-xk) -- the unscaled result.  (The bracket below takes
  // froot(Delta) < 0 for granted: on integer lattice data, where chi(y) == Delta happens, it bisected towards a root it
  // could never accept and ran out of passes 3e-10 away: tests/test_gpu_stress.py::test_b2_integer_lattices_exact_roots.)
  const bool scaled = delta < chiy;
  if (blockIdx.x == 0 && t == 0) hdr->b2_last_scaled = scaled ? 1 : 0;
  double eta = delta;
  bool stored = !scaled && store_first;
  if (scaled) {
    double lo = delta, hi = INFINITY, pP = -1.0, pC = -1.0;
    bool exact_step = false;
    bool hi_closed = false;   // hi is the a-priori bound, NOT evaluated: froot(hi) >= 0, possibly = 0 (x = 0: the bound IS the root)
    double y_eta = -1.0;
    double prev_step = -1.0;  // relative size of the previous exact step (for the quadratic-convergence estimate)
    const double eta_ub = chil * sqrt(F);
    const bool ub_ok = eta_ub > delta && eta_ub < INFINITY;
    bool have_eval = false;
    if (eta_s > 0.0) {
      // the second trial of the first pass IS an evaluation at eta_s, on whichever side of the root it fell (the piece root
      // of an evaluation converges quadratically from either side); froot(eta_ub) >= 0 is known without evaluating it.
      // (Round 2, measured: for Delta << eta the a-priori bound is 4e-5 from the root and the sample's root, 4e-4 off, lands
      // ABOVE it in half of the draws -- discarding the trial then cost a fourth streaming pass, 1.75 ms instead of 1.36.
      // Walking alternate passes backwards to start on what the memory-side cache may still hold: no gain, tools/r2/b2_seeds.py.)
      if (ub_ok && eta_s < eta_ub) { hi = eta_ub; hi_closed = true; }
      eta = eta_s; P = P1; C = C1;
      have_eval = true;
    }
    if (!have_eval && ub_ok) {
      eta = eta_ub;
      pass(eta / delta, 1.0, false, false, -1.0, false);
    }
    for (int it = 0; it < kB2MaxPass - 34; ++it) {
      const double r = eta / delta;
      const double f = eta - chil * sqrt(r * r * P + C);
#ifdef SPX_B2_DEBUG
      if (blockIdx.x == 0 && t == 0) printf("[b2] it %d eta %.17g P %.6g C %.17g f %.6g lo %.17g hi %.17g closed %d ub %.17g\n", it, eta, P, C, f, lo, hi, (int)hi_closed, eta_ub);
#endif
      if (f == 0.0 || (exact_step && P == pP && C == pC)) break;
      if (f < 0.0) lo = eta; else { hi = eta; hi_closed = false; }
      const double den = 1.0 - chil * chil * P / (delta * delta);
      double next = (den > 0.0) ? chil * sqrt(C / den) : INFINITY;
      exact_step = (next > lo && (next < hi || (hi_closed && next == hi)));
      if (!exact_step) next = (hi == INFINITY) ? 2.0 * lo : 0.5 * (lo + hi);
      if (!(next > lo && (next < hi || (hi_closed && next == hi)))) break;
      const double step = fabs(next - eta) / next;
      if (step <= 4e-16) break;
      // The piece roots converge quadratically on generic data (measured steps 4e-2, 1e-4, 6e-10: error(next) ~ 0.06 step^2), and
      // a breakpoint between eta and the root changes the root only to second order (the pieces join continuously).  `next` is
      // taken without the pass that would only confirm it -- but on the evidence of the more pessimistic LINEAR model: with
      // rho = step / prev_step the error left after this step is at most rho step / (1 - rho); below 2e-13 (a fifth of the
      // 1e-12 bar) the iteration ends.  (Round 2 first used the quadratic estimate K step^2, K = step / prev_step^2: one pass
      // fewer at n = 1e4, 19 vs 23 us, and no failure on record -- but two steps cannot tell the two models apart.)  Without
      // a previous step there is no estimate at all: only a step at rounding level ends the iteration.
      bool done = false;
      if (exact_step) {
        if (prev_step > 0.0) {
          const double rho = step / prev_step;
          done = step <= 1e-4 && rho < 0.5 && rho * step <= 2e-13 * (1.0 - rho);
        } else {
          done = step <= 2e-13;
        }
      }
      if (done) { eta = next; y_eta = -1.0; break; }
      prev_step = exact_step ? step : -1.0;
      const bool spec = can_spec && step <= 1e-9;  // (a step this small is normally taken without a pass: see above)
      pP = P; pC = C; eta = next;
      pass(eta / delta, delta / eta, false, spec, -1.0, false);
      y_eta = spec ? eta : -1.0;
    }
    stored = (y_eta == eta);
  }
#ifdef SPX_B2_PROFILE
  if (stored && blockIdx.x == 0 && t == 0) g_b2_nstamp = nst;
#endif
  if (stored) return;  // (after the last rendezvous; every workgroup takes the same path)
  // final: y = ProjB((-xk) r) rinv - sj   (:63, :65), or ProjB(-xk) - sj (:59) when the trust region is inactive
  const double r = scaled ? eta / delta : 1.0, rinv = scaled ? delta / eta : 1.0;
#ifdef SPX_B2_DEBUG
  if (blockIdx.x == 0 && t == 0) printf("[b2] final: scaled %d eta %.17g r %.6g rinv %.6g stored %d G %d n2 %lld\n", (int)scaled, eta, r, rinv, (int)stored, G, (long long)n2);
#endif
  auto out = [&](double lo, double hi, double x, double s) -> double {
    const double tt = scaled ? jl_min(jl_max((-x) * r, lo), hi) * rinv : jl_min(jl_max(-x, lo), hi);
    return tt - s;
  };
  if constexpr (REG) {
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
      const int64_t i = gtid + (int64_t)k * NT;
      if (i < n) y[i] = out(LO[k], HI[k], X[k], sj[i]);
    }
  } else {
    // The SAME element -> lane mapping as the reduction passes (tiles of 1024 lanes x 2 pairs, workgroup-strided): a pass may
    // have stored y speculatively, and two stores to one address are ordered only when the same lane issues them -- the
    // rendezvous between the passes does no cache maintenance, so a stale line of another XCD's L2 could otherwise land last
    // (seen: 256 wrong elements at n = 2.3e6 when this loop used tiles of 4 pairs).
    constexpr int KP = 2;
    constexpr int64_t kTilePairs = 1024 * KP;
    const int64_t ntiles = (n2 + kTilePairs - 1) / kTilePairs;
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += G) {
      f64x2 a[KP], b[KP], d[KP];
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        int64_t i = tile * kTilePairs + t + k * 1024;
        if (i >= n2) i = n2 - 1;
        a[k] = __builtin_nontemporal_load(q2 + i);
        b[k] = __builtin_nontemporal_load(x2 + i);
        d[k] = __builtin_nontemporal_load(s2 + i);
      }
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        const int64_t i = tile * kTilePairs + t + k * 1024;
        if (i < n2)
        {
          const double sq0 = d[k].x + a[k].x, sq1 = d[k].y + a[k].y;
          __builtin_nontemporal_store(f64x2{out(sq0 - ls, sq0 + ls, b[k].x, d[k].x), out(sq1 - ls, sq1 + ls, b[k].y, d[k].y)}, y2 + i);
        }
      }
    }
    if ((n & 1) && blockIdx.x == 0 && t == 0) {
      const double sql = sj[n - 1] + q[n - 1];
      y[n - 1] = out(sql - ls, sql + ls, xk[n - 1], sj[n - 1]);
    }
  }
#ifdef SPX_B2_PROFILE
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  B2_STAMP();
  if (blockIdx.x == 0 && t == 0) g_b2_nstamp = nst;
#endif
}

// psi(y) from the reduced sums, on the device (spx_ctx_set_value_target)
__global__ void k_b2_obj_value(const B2Ws* ws, double lambda, double delta, double* target) {
  const double nrm = sqrt(ws->C);
  const double eps = 2.220446049250313e-16;
  const double tol = fmax(eps, sqrt(eps) * fmax(nrm, fabs(delta)));
  const bool inside = (nrm <= delta) || (fabs(nrm - delta) <= tol);
  *target = inside ? lambda * ws->P : __longlong_as_double(0x7ff0000000000000ll);
}

}  // namespace

// ShiftedNormL1B2 as a function (src/shiftedNormL1B2.jl:32).  IndBallL2(Delta)(v) [ext: ProximalOperators.jl] is 0 iff
// ||v|| <= Delta or ||v|| ~ Delta (isapprox, atol = eps, rtol = sqrt(eps)), +Inf otherwise.
SPX_EXPORT int spx_obj_l1_b2(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,
                             double lambda, double delta, double* value) {
  SPX_REQUIRE(ctx != nullptr && value != nullptr, "ctx or value is NULL");
  SPX_REQUIRE(n >= 0, "n < 0");
  *value = 0.0;
  double P = 0.0, C = 0.0;
  if (n > 0) {
    SPX_REQUIRE(y && xk && sj, "NULL vector with n > 0");
    int rc = spx_ws_reserve(ctx, sizeof(B2Ws) + 256);
    if (rc) return rc;
    SPX_ON_DEVICE(ctx);
    B2Ws* ws = reinterpret_cast<B2Ws*>(ctx->ws);
    const bool vec = n >= 2 && spx_aligned16(y) && spx_aligned16(xk) && spx_aligned16(sj);
    int64_t blocks = vec ? ((n >> 1) + 1023) / 1024 : (n + 256 * 8 - 1) / (256 * 8);
    if (blocks > kB2Blocks) blocks = kB2Blocks;
    if (blocks < 1) blocks = 1;
    if (vec) hipLaunchKernelGGL((k_b2_obj<true>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, y, xk, sj, n, ws);
    else hipLaunchKernelGGL((k_b2_obj<false>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, y, xk, sj, n, ws);
    hipLaunchKernelGGL(k_b2_reduce, dim3(1), dim3(256), 0, ctx->stream, ws, (int)blocks, 0);
    SPX_LAUNCH_CHECK();
    if (ctx->value_target) {  // device-resident value: psi(y) from (P, C) on the device, nothing read back
      hipLaunchKernelGGL(k_b2_obj_value, dim3(1), dim3(1), 0, ctx->stream, (const B2Ws*)ws, lambda, delta, ctx->value_target);
      SPX_LAUNCH_CHECK();
      *value = std::nan("");
      return SPX_OK;
    }
    double pc[2];
    { const int rcc = spx_require_not_capturing(ctx, "returning psi(y) to the host"); if (rcc) return rcc; }
    SPX_HIP(hipMemcpyAsync(pc, &ws->P, sizeof(pc), hipMemcpyDeviceToHost, ctx->stream));
    SPX_HIP(hipStreamSynchronize(ctx->stream));
    P = pc[0];
    C = pc[1];
  } else if (ctx->value_target) {
    { const int rz = spx_zero_async(ctx, ctx->value_target, sizeof(double)); if (rz) return rz; }
    return SPX_OK;
  }
  const double nrm = std::sqrt(C);
  const double eps = 2.220446049250313e-16;
  const double tol = std::fmax(eps, std::sqrt(eps) * std::fmax(nrm, std::fabs(delta)));
  const bool inside = (nrm <= delta) || (std::fabs(nrm - delta) <= tol);
  *value = inside ? lambda * P : INFINITY;
  return SPX_OK;
}

SPX_EXPORT int spx_prox_l1_b2(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                              double lambda, double sigma, double delta, double chi_lambda) {
  int rc = spx_check_common(ctx, y, q, xk, sj, n);
  if (rc) return rc;
  if (n == 0) return SPX_OK;
  rc = spx_ws_reserve(ctx, sizeof(B2Ws) + 256);
  if (rc) return rc;
  SPX_ON_DEVICE(ctx);
  B2Ws* ws = reinterpret_cast<B2Ws*>(ctx->ws);
  const double ls = lambda * sigma;  // `psi.lambda * sigma`, :56
  bool vec = n >= 2 && spx_aligned16(y) && spx_aligned16(q) && spx_aligned16(xk) && spx_aligned16(sj);
  // Residency (spx_resident_cap): the grid of a launch that synchronises inside itself never exceeds what can be resident
  // at once; the streaming form works with any grid >= 1, the register-resident one needs ceil(n / 8192) workgroups.
  const int64_t cap_reg = spx_resident_cap(ctx, reinterpret_cast<const void*>(&k_b2_coop<true, kB2Epl, kB2RegThreads>), kB2RegThreads, 0);
  const int64_t cap_mem = spx_resident_cap(ctx, reinterpret_cast<const void*>(&k_b2_coop<false, 1, 1024>), 1024, 0);
  if (cap_mem < 1) return SPX_ERR_INTERNAL;  // (message set by spx_resident_cap)
  const int64_t gmax_reg = cap_reg < kB2Cols ? cap_reg : kB2Cols;
  const int64_t gmax_mem = cap_mem < kB2Cols ? cap_mem : kB2Cols;
  if (vec || n <= (int64_t)kB2RegBlock * gmax_reg) {
    // one launch, no read-back (see k_b2_coop)
    const bool reg = n <= (int64_t)kB2RegBlock * gmax_reg;
    int64_t g = reg ? (n + kB2RegBlock - 1) / kB2RegBlock : gmax_mem;
    if (g < 1) g = 1;
    rc = spx_sync_reserve(ctx, kSpxSyncSelBytes + kB2SyncBytes);
    if (rc) return rc;
    auto disjoint = [&](const double* a) { return (y + n <= a) || (a + n <= y); };
    const int can_spec = (disjoint(q) && disjoint(xk) && disjoint(sj)) ? 1 : 0;
    SpxSyncHeader* hdr = reinterpret_cast<SpxSyncHeader*>(ctx->sync);
    unsigned long long* sets = reinterpret_cast<unsigned long long*>(static_cast<char*>(ctx->sync) + kSpxSyncSelBytes);
    int use = ctx->b2_set, other = use ^ 1;
    int clear_g = ctx->b2_dirty_g[other];
    const bool graph_safe = spx_capture_check(ctx) || ctx->graph_safe;  // (see spx_ctx::graph_safe)
    if (graph_safe) {  // set 0, its g columns zeroed by a node in front of the launch; nothing alternates
      use = 0; other = 1; clear_g = 0;
      if (g > 1)
      {
        rc = spx_zero2d_async(ctx, sets, (size_t)kB2Cols * kB2Words * sizeof(unsigned long long),
                              (size_t)g * kB2Words * sizeof(unsigned long long), (size_t)kB2MaxPass);
        if (rc) return rc;
      }
    }
    unsigned long long* rows = sets + (size_t)use * kB2SetWords;
    unsigned long long* clear_rows = sets + (size_t)other * kB2SetWords;
    {
      SpxCoopLaunchGuard guard(ctx);
      if (reg)
        hipLaunchKernelGGL((k_b2_coop<true, kB2Epl, kB2RegThreads>), dim3((unsigned)g), dim3(kB2RegThreads), 0, ctx->stream, y, q, xk, sj, n, ls, delta,
                           chi_lambda, rows, clear_rows, clear_g, hdr, can_spec);
      else
        hipLaunchKernelGGL((k_b2_coop<false, 1, 1024>), dim3((unsigned)g), dim3(1024), 0, ctx->stream, y, q, xk, sj, n, ls, delta,
                           chi_lambda, rows, clear_rows, clear_g, hdr, can_spec);
    }
    if (graph_safe) {  // both sets count as used by the widest grid from here on (a replay may have touched set 0)
      ctx->b2_dirty_g[0] = ctx->b2_dirty_g[1] = kB2Cols;
    } else {
      ctx->b2_dirty_g[use] = (g > 1) ? (int)g : 0;  // (one workgroup exchanges nothing)
      ctx->b2_dirty_g[other] = 0;
      ctx->b2_set = other;
    }
    SPX_LAUNCH_CHECK();
    return SPX_OK;
  }
  // views from an odd element on (all four vectors 8 bytes off a 16-byte boundary): the vector kernels run on the
  // aligned rest and take element 0 along (2.1 -> 1.4 ms at n = 1e8, tools/bench_misaligned.py)
  auto off8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 8u; };
  const int head = (!vec && n >= 3 && off8(y) && off8(q) && off8(xk) && off8(sj)) ? 1 : 0;
  const double* const q0 = q;
  const double* const xk0 = xk;
  const double* const sj0 = sj;
  double* const y0 = y;
  const int64_t n0 = n;
  if (head) { vec = true; ++y; ++q; ++xk; ++sj; --n; }
  int64_t blocks = vec ? ((n >> 1) + 1023) / 1024 : (n + 256 * 8 - 1) / (256 * 8);
  if (blocks > kB2Blocks) blocks = kB2Blocks;
  if (blocks < 1) blocks = 1;
  // y = ProjB(-xk); chi(y) = chi_lambda * ||y||: at r = 1,  ||y||^2 = P + C
  double P, C, F;
  // y overlaps none of the inputs: reduction passes may store y for their own scale (see below).  The first pass does:
  // if the trust region turns out to be inactive (Delta > chi(y), :61) its y = ProjB(-xk) - sj is the result and the call
  // is this one pass
  auto disjoint = [&](const double* a) { return (y0 + n0 <= a) || (a + n0 <= y0); };
  const bool can_spec = vec && disjoint(q0) && disjoint(xk0) && disjoint(sj0);
  // (a store that turns out useless costs 8 B/element: the pass only stores when the previous call on this context was
  //  unscaled too -- 0.57 ms instead of 0.85 ms for an inactive trust region, 1.35 ms unchanged for an active one)
  const bool store_first = can_spec && !ctx->b2_last_scaled;
  rc = b2_sums(ctx, q, xk, sj, n, ls, 1.0, ws, (int)blocks, vec, true, &P, &C, &F, store_first ? y : nullptr, 1.0, head);
  if (rc) return rc;
  const double chiy = chi_lambda * std::sqrt(P + C);
  // (delta == chiy: froot(Delta) == 0, find_zero returns Delta, the scaled branch is the unscaled result: see k_b2_coop)
  ctx->b2_last_scaled = (delta < chiy) ? 1 : 0;
  if (!(delta < chiy) && store_first) return SPX_OK;  // unscaled and already stored
  int scaled = 0;
  double eta = delta;
  double y_eta = -1.0;  // eta for which a reduction pass has stored y (speculatively), or -1
  if (delta < chiy) {  // :61 (equality: see above)
    scaled = 1;
    // froot(eta) = eta - chi_lambda sqrt((eta/Delta)^2 P + C); froot(Delta) <= 0 here.  Bracket lo: froot <= 0, hi: froot > 0.
    double lo = delta, hi = INFINITY;
    double pP = -1.0, pC = -1.0;
    bool exact_step = false;  // eta was set to the exact root of the piece (pP, pC)
    // a-priori upper bound of the root: froot(chi_lambda sqrt(F)) >= 0.  The iteration starts THERE: from above the
    // piece roots decrease monotonically to the root (2-3 passes), whereas from eta = Delta the first pieces have no
    // root at all (1 - chi^2 P / Delta^2 <= 0) and the bracket would have to be grown by doubling.
    const double eta_ub = chi_lambda * std::sqrt(F);
    if (eta_ub > delta && std::isfinite(eta_ub)) {
      eta = eta_ub;
      rc = b2_sums(ctx, q, xk, sj, n, ls, eta / delta, ws, (int)blocks, vec, false, &P, &C, nullptr, nullptr, 1.0, head);
      if (rc) return rc;
    }
    // y overlaps none of the inputs: a pass that is likely to be the last one (the step has become small) also stores y
    // for its eta; the iteration always ends on an eta that a pass has evaluated, so if that pass stored y the final
    // pass below is skipped (the call then costs the reduction passes only)
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
      if (std::fabs(next - eta) <= 4e-16 * next) break;  // converged: eta (evaluated) and next agree to the last bits
      const bool spec = can_spec && std::fabs(next - eta) <= 1e-3 * next;
      pP = P; pC = C; eta = next;
      rc = b2_sums(ctx, q, xk, sj, n, ls, eta / delta, ws, (int)blocks, vec, false, &P, &C, nullptr, spec ? y : nullptr,
                   delta / eta, head);
      if (rc) return rc;
      y_eta = spec ? eta : -1.0;
    }
  }
  if (scaled && y_eta == eta) return SPX_OK;  // y already holds ProjB((-xk) eta/Delta) Delta/eta - sj
  if (vec) {
    const int64_t fblocks = ((n >> 1) + 1023) / 1024;
    hipLaunchKernelGGL((k_b2_final<true>), dim3((unsigned)(fblocks < 1 ? 1 : fblocks)), dim3(256), 0, ctx->stream, y, q,
                       xk, sj, n, ls, eta / delta, delta / eta, scaled, head);
  } else {
    int64_t fblocks = (n + 255) / 256;
    if (fblocks > (int64_t)ctx->num_cu * 16) fblocks = (int64_t)ctx->num_cu * 16;
    hipLaunchKernelGGL((k_b2_final<false>), dim3((unsigned)fblocks), dim3(256), 0, ctx->stream, y, q, xk, sj, n, ls,
                       eta / delta, delta / eta, scaled, 0);
  }
  SPX_LAUNCH_CHECK();
  return SPX_OK;
}
