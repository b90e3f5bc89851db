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
// form: eta = chi_lambda sqrt(C / (1 - chi_lambda^2 P / Delta^2)).  Iteration: one reduction gives (P, C) at the
// current eta, the piece's root is taken, the next reduction verifies it (identical sums = same piece = done); a bracket
// [froot < 0, froot > 0] safeguards every step.
// Everything runs in ONE launch of a resident grid (k_b2_coop): register-resident up to 2^21 elements; beyond, two streaming
// passes (round 3) -- a sample predicts the root to ~2e-3, the first pass classifies every element against a bracket of
// +-1.5 % around the prediction (same clamp state at both ends = a fixed contribution to P or C; the ~1-2 % whose breakpoint
// lies inside are kept as candidates), the root is then found on aggregate + candidates, and the second pass stores y.
// (Rounds 1-2: a host-driven loop and three streaming passes; both gone.)
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
// The whole prox in ONE launch.  The workgroups of a resident grid (<= number of CUs) run the iteration themselves: every
// reduction ends in per-workgroup partial sums (written, not accumulated: the order of every addition is fixed, so all
// workgroups form bit-identical totals), an exchange, and the scalar update of eta, which every lane redoes for itself.
// REG: n <= 8 Ki x number of resident workgroups -- xk, sj and sj + q stay in registers, the vectors are read once.
// !REG: two streaming passes (see the head of this file); VEC = 16-byte accesses, else 8-byte (views of mixed alignment).
// =============================================================================================
// register-resident form: 512 lanes x 16 elements per workgroup (256 VGPRs per lane).  1024 lanes x 8 spill under their 128
// VGPRs (loop invariants the compiler keeps per element: the box ends, the store addresses), and the first touch of a wave's
// scratch memory costs microseconds at the start of every launch; 1024 x 16 spill 120 registers and run 2x slower.
constexpr int kB2Epl = 16;
constexpr int kB2RegThreads = 512;
constexpr int kB2RegBlock = kB2Epl * kB2RegThreads;  // elements per workgroup of the register-resident form
constexpr int kB2MaxPass = 64;
// One workgroup's partial sums of a reduction: 8 words in spx_ctx::sync, at a FIXED place (set, reduction number, workgroup),
// written once per launch.  A word is its own "ready" flag: the sync area starts out zero, a sum v is stored as
// bits(v) + 1 (never 0: NaNs are canonicalised first), and a reader polls the word until it is non-zero -- no counter, no
// rendezvous, one memory round trip per reduction instead of four (store + wait, arrive, poll, load: 2.6-3.6 -> ~1.x us,
// which is most of a call at n <= 1e6).  Launches alternate between two sets; a launch zeroes what the launch before the
// previous one left in the other set (the host knows how many workgroups that was).
constexpr int kB2Cols = 256;   // workgroups at most
constexpr int kB2Words = 8;
constexpr size_t kB2SetWords = (size_t)kB2MaxPass * kB2Cols * kB2Words;
constexpr size_t kB2SyncBytes = 2 * kB2SetWords * sizeof(unsigned long long);   // 2 MiB, behind the select state
static_assert(kB2SyncBytes == kSpxSyncB2Bytes, "spx_ctx::sync layout");
__device__ __forceinline__ void b2_put(unsigned long long* slot, double v) {
  const unsigned long long b = (v != v) ? 0x7ff8000000000000ull : (unsigned long long)__double_as_longlong(v);
  __hip_atomic_store(slot, b + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (an atomic swap instead: no faster)
}
// sums of up to 6 values over a workgroup of <= 16 wavefronts; lds = [6][16] (the slots of absent wavefronts stay zero).
// Round 4: wavefront sums by DPP (four row steps + two permutes instead of six permutes per double), and every 16-lane row reads
// the 16 slots ONCE and folds them by DPP -- every lane reading all 16 slots itself was 96 LDS instructions per wavefront and
// call, 3.5 us per call with 16 wavefronts (tools/exp/team_reduce.hip: 1.9 us this way); a call of the one-launch forms makes
// 20-30 of them.  A fixed shape: all lanes, and all workgroups on the same partial sums, form the same bits.
// NV: only the first NV of the six are formed (most reductions carry two sums, P and C: every wavefront of the workgroup does
// this work, so a sum that nobody reads costs as much as one that is read).
template <int NV>
__device__ __forceinline__ void b2_block_sum_n(double (&v)[6], double (*lds)[16]) {
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = wave_sum_dpp(v[k]);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) lds[k][w] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = fold16_sum(lds[k][threadIdx.x & 15]);
}
// mask: which of the six sums travel (3, 7, 15 or 63: the first two, three, four or all)
__device__ __forceinline__ void b2_block_sum6(double (&v)[6], double (*lds)[16], unsigned int mask = 63u) {
  if (mask <= 3u) b2_block_sum_n<2>(v, lds);
  else if (mask <= 7u) b2_block_sum_n<3>(v, lds);
  else if (mask <= 15u) b2_block_sum_n<4>(v, lds);
  else b2_block_sum_n<6>(v, lds);
}

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

// 16-byte pair p of a vector: one non-temporal 16-byte access, or two 8-byte ones when the vectors are not all 16-byte aligned
template <bool VEC>
__device__ __forceinline__ f64x2 b2_ld(const double* v, int64_t p) {
  if constexpr (VEC) return __builtin_nontemporal_load(reinterpret_cast<const f64x2*>(v) + p);
  else return f64x2{__builtin_nontemporal_load(v + 2 * p), __builtin_nontemporal_load(v + 2 * p + 1)};
}
template <bool VEC>
__device__ __forceinline__ void b2_st(double* v, int64_t p, f64x2 o) {
  if constexpr (VEC) __builtin_nontemporal_store(o, reinterpret_cast<f64x2*>(v) + p);
  else { __builtin_nontemporal_store(o.x, v + 2 * p); __builtin_nontemporal_store(o.y, v + 2 * p + 1); }
}

constexpr int kB2DmaKiB = 3;            // KiB per wavefront, vector and tile of the LDS-DMA staged streaming passes (16 waves: 144 KiB)
constexpr double kB2Bracket = 1.5e-2;   // half-width of the bracket around the sample's root (its statistical error: ~2e-3)

// LDSX (round 3; REG with 1024 lanes x 16 elements): xk sits in LDS (128 KiB), sj + q in registers and the box ends are formed
// from it on the fly -- 16 Ki elements per workgroup, 4 Mi on 256 CUs, where the streaming form pays for its sample, its
// bracket and two passes over memory (n = 4e6: 111 us per call).
template <bool REG, int EPL, int THREADS, bool VEC, bool LDSX = false>
__global__ __launch_bounds__(THREADS) void k_b2_coop(double* y, const double* q, const double* xk, const double* sj, int64_t n,
                                                   double ls, double delta, double chil, unsigned long long* rows,
                                                   unsigned long long* clear_rows, int clear_g, SpxSyncHeader* hdr,
                                                   int can_spec, f64x2* cand, unsigned int cand_cap) {
  __shared__ double lds6[6][16];
  const int t = threadIdx.x;
  const int G = (int)gridDim.x;
  const int64_t NT = (int64_t)G * blockDim.x;
  const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + t;
  // the other set, for the launch after this one: reductions x (workgroups of the launch that used it) x words, plain stores
  for (int64_t idx = gtid; idx < (int64_t)kB2MaxPass * clear_g * kB2Words; idx += NT) {
    const int64_t pass_i = idx / (clear_g * kB2Words), rem = idx % (clear_g * kB2Words);
    clear_rows[pass_i * kB2Cols * kB2Words + rem] = 0ull;
  }
  if (t < 96) (&lds6[0][0])[t] = 0.0;  // (workgroups of fewer than 16 wavefronts leave the upper slots alone)
  int nst = 0;
  B2_STAMP();
  const int last_scaled = hdr->b2_last_scaled;  // (written by the previous call's launch)
  // (sj itself is needed only where a y is stored: reloaded there, so that 16 elements per lane fit the 128 VGPRs)
  static_assert(!LDSX || REG, "LDSX is a register-resident form");
  double X[(REG && !LDSX) ? EPL : 1], LO[(REG && !LDSX) ? EPL : 1], HI[(REG && !LDSX) ? EPL : 1];  // xk and the box (sj + q) -+ lambda sigma of each element
  double SQ[LDSX ? EPL : 1];                                      // LDSX: sj + q; xk in lx
  __shared__ double lx[LDSX ? EPL * THREADS : 1];
  // element k of this lane: xk and its box
  // (lsv: lambda sigma, passed through an opaque copy per sweep -- the box ends of all 16 elements are loop invariants the
  //  compiler would otherwise keep in 64 more registers than a 1024-lane workgroup has)
  auto elem = [&](int k, double lsv, double& x, double& lo, double& hi) {
    if constexpr (LDSX) { x = lx[k * THREADS + t]; lo = SQ[k] - lsv; hi = SQ[k] + lsv; }
    else { x = X[k]; lo = LO[k]; hi = HI[k]; }
  };
  auto opaque = [](double v) -> double { if constexpr (LDSX) asm volatile("" : "+v"(v)); return v; };
  auto opaque_i = [](int64_t v) -> int64_t { if constexpr (LDSX) asm volatile("" : "+v"(v)); return v; };
  if constexpr (LDSX) {
#pragma unroll
    for (int k0 = 0; k0 < EPL; k0 += 4) {  // twelve loads in flight per lane
#pragma unroll
      for (int k = k0; k < k0 + 4; ++k) {
        const int64_t i = gtid + (int64_t)k * NT;
        const int64_t ic = i < n ? i : n - 1;
        const double xv = __builtin_nontemporal_load(xk + ic), sv = __builtin_nontemporal_load(sj + ic), qv = __builtin_nontemporal_load(q + ic);
        SQ[k] = i < n ? (sv + qv) : 0.0;   // `sj .+ q` (:56)
        lx[k * THREADS + t] = i < n ? xv : 0.0;
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (REG) {
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
  // off by more than the bracket absorbs, tools/r2/b2_sorted_time.py.)
  const int kChunks = 32 * G;
  const int chunk = (int)blockIdx.x * 32 + (t >> 5);
  const int64_t nsample = (int64_t)kChunks * 32;
  const bool has_sample = !REG && n >= 4 * nsample;
  double sx = 0.0, ss = 0.0, ssq = 0.0;
  if (has_sample) {
    const int64_t i = (int64_t)((double)chunk * (double)(n - 32) / (double)(kChunks - 1)) + (t & 31);
    sx = xk[i]; ss = sj[i]; ssq = ss + q[i];
  }
  const int64_t n2 = n >> 1;
  int np = 0;  // reductions exchanged so far (row of the exchange words)
  // One reduction: the six sums of this workgroup -> totals, identical in every workgroup.  mask: which words travel.
  auto reduce = [&](double (&v)[6], unsigned int mask) {
    b2_block_sum6(v, lds6, mask);
    if (G > 1) {
      // the partial sums are the ONLY data the workgroups exchange: agent-scope atomic stores / loads (`sc1`, past the
      // non-coherent caches) of words that carry their own ready flag (see b2_put)
      unsigned long long* row = rows + (size_t)np * kB2Cols * kB2Words;
      if (t == 0) {
        unsigned long long* mine = row + (size_t)blockIdx.x * kB2Words;
#pragma unroll
        for (int k = 0; k < 6; ++k)
          if (mask & (1u << k)) b2_put(mine + k, v[k]);
      }
      double g[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      if (t < G) {
        const unsigned long long* theirs = row + (size_t)t * kB2Words;
        unsigned long long w[6];
        unsigned int spins = 0;
        for (;;) {  // (every workgroup of the grid is resident and stores these words once per reduction; bounded all the same)
          bool all = true;
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            w[k] = (mask & (1u << k)) ? __hip_atomic_load(theirs + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 1ull;
            all = all && (w[k] != 0ull);
          }
          if (all) break;
          if (spx_wait_expired(spins, hdr)) break;  // (the sums come out as garbage / NaN: see kSpxPollLimit)
          __builtin_amdgcn_s_sleep(1);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) g[k] = (mask & (1u << k)) ? __longlong_as_double((long long)(w[k] - 1ull)) : 0.0;
      }
      b2_block_sum6(g, lds6, mask);
#pragma unroll
      for (int k = 0; k < 6; ++k) v[k] = g[k];
    }
    ++np;
  };
  // the sums at scale r (= eta / Delta) over one element; the Julia-semantics min / max cost six instructions each and matter
  // only for the bits of a STORED y (signed zeros, NaN propagation): the sums take v_max_f64 / v_min_f64 and masked fma
  // operands; a NaN operand -- which v_max / v_min would drop -- is tracked separately and poisons P, as the reference's norm would be NaN
  bool bad = false;
  auto acc = [&](double lo, double hi, double x, double r, double& p, double& c) -> double {
    const double z = (-x) * r;
    bad |= (z != z) | (lo != lo) | (hi != hi);
    const double pzf = fmin(fmax(z, lo), hi);
    const bool un = (pzf == z);
    const double xm = un ? x : 0.0, cm = un ? 0.0 : pzf;
    p = __builtin_fma(xm, xm, p);
    c = __builtin_fma(cm, cm, c);
    return pzf;
  };
  auto outv = [&](double lo, double hi, double x, double s, double r, double rinv) -> double {
    return jl_min(jl_max((-x) * r, lo), hi) * rinv - s;  // :56 / :63, :65, bit-faithful for the stored value
  };
  double P = 0.0, C = 0.0, F = 0.0;
  // ---- reduction over the SAMPLE registers / the register-resident vectors at scale r; first: also F; store (REG): also y
  auto pass_small = [&](double r, double rinv, bool first, bool store, bool sample) {
    double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    bad = false;
    auto one = [&](double lo, double hi, double x) {
      const double pz = acc(lo, hi, x, r, v[0], v[1]);
      if (first) {
        const double far = (x < 0.0) ? hi : (x > 0.0) ? lo : pz;  // -x > 0: z -> +inf -> hi
        v[2] = __builtin_fma(far, far, v[2]);
      }
    };
    if (sample) {
      if (has_sample) one(ssq - ls, ssq + ls, sx);
    } else if constexpr (REG) {
      const double lsv = opaque(ls);
      const int64_t ntv = opaque_i(NT);
#pragma unroll
      for (int k = 0; k < EPL; ++k) {
        const int64_t i = gtid + (int64_t)k * ntv;
        if (i < n) {
          double x, lo, hi;
          elem(k, lsv, x, lo, hi);
          one(lo, hi, x);
          if (store) y[i] = outv(lo, hi, x, sj[i], r, rinv);
        }
        __builtin_amdgcn_sched_barrier(0);  // one element at a time: interleaved, the unrolled visits spill
      }
    }
    if (bad) v[0] = __longlong_as_double(0x7ff8000000000000ll);
    reduce(v, first ? 7u : 3u);
    P = v[0]; C = v[1];
    if (first) F = v[2];
    B2_STAMP();
  };
  // ---- streaming passes (!REG): tiles of THREADS x KP pairs, workgroup-strided, loads of the next tile issued before the
  // current one is evaluated (ping-pong register sets; a persistent lane otherwise serialises load latency and arithmetic).
  // The SAME element -> lane mapping in every pass: two stores to one address are ordered only when the same lane issues
  // them (no cache maintenance between the passes; seen: 256 stale elements at n = 2.3e6 with differing tile shapes).
  // VEC: LDS-DMA staging as in the separable skeleton (global_load_lds ... nt: no VGPR destination, 1 KiB per wave instruction):
  // every wavefront owns 3 KiB per vector and tile, issues its nine loads, waits once, reads its own 16-byte slots back and
  // loops -- 144 KiB in flight per CU without holding registers (the register ping-pong of round 2 streamed at 5.8 TB/s, the
  // skeleton at 6.3).  !VEC (views of mixed alignment): 8-byte register loads, ping-pong.
  constexpr int KP = (VEC && !REG) ? kB2DmaKiB : 2;
  constexpr int64_t kTilePairs = (VEC && !REG) ? (int64_t)(THREADS / 64) * 64 * kB2DmaKiB : (int64_t)THREADS * KP;
  const int64_t ntiles = (n2 + kTilePairs - 1) / kTilePairs;
  __shared__ __attribute__((aligned(16))) char dma[(VEC && !REG) ? (THREADS / 64) * 3 * kB2DmaKiB * 1024 : 16];
  // dynamic (the storing pass, when it is the only pass of the launch that stores y): tiles are handed out by an atomic counter
  // instead of workgroup-strided -- a persistent grid with a static partition waits for its slowest CU (5.5 TB/s); handed out on
  // demand, with the next index fetched while the current tile is processed, the same grid streams at 6.3
  // (tools/exp/persistent_stream.hip: 5467 -> 6317 GB/s; four tiles per atomic: 5794).  Only where nothing depends on which
  // workgroup saw which element: the sums of a pass are formed per workgroup in a fixed order (results reproducible from run to
  // run), and two passes that store y must use the same element -> lane mapping -- so the reduction passes stay static.
  __shared__ unsigned int next_tile;
  unsigned int* const tile_ctr = reinterpret_cast<unsigned int*>(rows + (size_t)(kB2MaxPass - 1) * kB2Cols * kB2Words + 7);  // (row 63 is
  // never reached by a reduction; word 7 of a slot is never a partial sum; zeroed with the set like everything else in it)
  auto stream = [&](auto&& visit_pair, bool dynamic = false) {  // visit_pair(valid, pair index, q pair, xk pair, sj pair), called by every lane
    if constexpr (VEC && !REG) {
      typedef __attribute__((address_space(3))) void lds_void;
      const int wave = t >> 6, lane = t & 63;
      char* wl = dma + wave * (3 * kB2DmaKiB * 1024);
      const f64x2* q2 = reinterpret_cast<const f64x2*>(q);
      const f64x2* x2 = reinterpret_cast<const f64x2*>(xk);
      const f64x2* s2 = reinterpret_cast<const f64x2*>(sj);
      int64_t tile = blockIdx.x;
      if (dynamic) {
        if (t == 0) next_tile = __hip_atomic_fetch_add(tile_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        tile = (int64_t)next_tile;
      }
      while (tile < ntiles) {
        if (dynamic) {
          __syncthreads();  // every lane has read next_tile
          if (t == 0) next_tile = __hip_atomic_fetch_add(tile_ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (in flight during this tile)
        }
        const int64_t base = tile * kTilePairs + (int64_t)wave * (64 * kB2DmaKiB) + lane;
#pragma unroll
        for (int k = 0; k < kB2DmaKiB; ++k) {
          int64_t i = base + k * 64;
          if (i >= n2) i = n2 - 1;
          __builtin_amdgcn_global_load_lds((const void*)(q2 + i), (lds_void*)(wl + (0 * kB2DmaKiB + k) * 1024), 16, 0, 2);
          __builtin_amdgcn_global_load_lds((const void*)(x2 + i), (lds_void*)(wl + (1 * kB2DmaKiB + k) * 1024), 16, 0, 2);
          __builtin_amdgcn_global_load_lds((const void*)(s2 + i), (lds_void*)(wl + (2 * kB2DmaKiB + k) * 1024), 16, 0, 2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < kB2DmaKiB; ++k) {
          const int64_t i = base + k * 64;
          const f64x2 a = *reinterpret_cast<const f64x2*>(wl + (0 * kB2DmaKiB + k) * 1024 + lane * 16);
          const f64x2 b = *reinterpret_cast<const f64x2*>(wl + (1 * kB2DmaKiB + k) * 1024 + lane * 16);
          const f64x2 d = *reinterpret_cast<const f64x2*>(wl + (2 * kB2DmaKiB + k) * 1024 + lane * 16);
          visit_pair(i < n2, i, a, b, d);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this tile's LDS reads are done before the next tile's loads are issued
        if (dynamic) {
          __syncthreads();
          tile = (int64_t)next_tile;
        } else {
          tile += G;
        }
      }
    } else {
    auto ld = [&](int64_t tile, f64x2* a, f64x2* b, f64x2* d) {
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        int64_t i = tile * kTilePairs + t + k * THREADS;
        if (i >= n2) i = n2 - 1;
        a[k] = b2_ld<VEC>(q, i);
        b[k] = b2_ld<VEC>(xk, i);
        d[k] = b2_ld<VEC>(sj, i);
      }
    };
    auto comp = [&](int64_t tile, const f64x2* a, const f64x2* b, const f64x2* d) {
#pragma unroll
      for (int k = 0; k < KP; ++k) {
        const int64_t i = tile * kTilePairs + t + k * THREADS;
        visit_pair(i < n2, i, a[k], b[k], d[k]);
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
    }
  };
  // the storing pass: y = ProjB((-xk) r) rinv - sj
  bool y_written = false;  // (!REG) an earlier pass of this launch stored y: the storing pass must then keep the static mapping
  auto store_pass = [&](double r, double rinv) {
    if constexpr (!REG) {
      if (n2 > 0)
        stream([&](bool valid, int64_t i, f64x2 a, f64x2 b, f64x2 d) {
          if (valid) {
            const double sq0 = d.x + a.x, sq1 = d.y + a.y;
            b2_st<VEC>(y, i, f64x2{outv(sq0 - ls, sq0 + ls, b.x, d.x, r, rinv), outv(sq1 - ls, sq1 + ls, b.y, d.y, r, rinv)});
          }
        }, !y_written && G > 1 && ntiles >= 8 * (int64_t)G);  // (a few tiles per workgroup: the hand-out costs more than it balances -- n = 6e6, 4 tiles each: 128 -> 135 us; n = 1.6e7, 10 each: 220 -> 212)
      if ((n & 1) && blockIdx.x == 0 && t == 0) {
        const double sql = sj[n - 1] + q[n - 1];
        y[n - 1] = outv(sql - ls, sql + ls, xk[n - 1], sj[n - 1], r, rinv);
      }
    } else {
#pragma unroll
      for (int k = 0; k < EPL; ++k) {
        const int64_t i = gtid + (int64_t)k * NT;
        if (i < n) {
          double x, lo, hi;
          elem(k, ls, x, lo, hi);
          y[i] = outv(lo, hi, x, sj[i], r, rinv);
        }
        if constexpr (LDSX) __builtin_amdgcn_sched_barrier(0);
      }
    }
  };
  // a plain reduction pass over the vectors at scale r (the fallback iteration of the streaming form); store: also y
  auto pass_stream = [&](double r, double rinv, bool store) {
    double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    bad = false;
    if (store) y_written = true;
    if constexpr (!REG) {
      auto one = [&](double qv, double x, double s) -> double {
        const double sq = s + qv;
        acc(sq - ls, sq + ls, x, r, v[0], v[1]);
        return store ? outv(sq - ls, sq + ls, x, s, r, rinv) : 0.0;
      };
      if (n2 > 0)
        stream([&](bool valid, int64_t i, f64x2 a, f64x2 b, f64x2 d) {
          if (valid) {
            const f64x2 o{one(a.x, b.x, d.x), one(a.y, b.y, d.y)};
            if (store) b2_st<VEC>(y, i, o);
          }
        });
      if ((n & 1) && blockIdx.x == 0 && t == 0) {
        const double o = one(q[n - 1], xk[n - 1], sj[n - 1]);
        if (store) y[n - 1] = o;
      }
    }
    if (bad) v[0] = __longlong_as_double(0x7ff8000000000000ll);
    reduce(v, 3u);
    P = v[0]; C = v[1];
    B2_STAMP();
  };
  bool store_first = can_spec && !last_scaled;
  // Streaming form, the previous call found the trust region inactive: the speculative pass (y = ProjB(-xk) - sj stored, the
  // sums at r = 1 formed) takes its tiles on demand like the storing pass -- its OUTPUT does not depend on which workgroup saw
  // which element; its sums do, in the last bits, so they decide only a CLEAR `Delta > chi(y)` (margin 1e-12).  Anything else
  // (the trust region has become active, or the call is too close to say) falls through to the ordinary first pass with its
  // static partition and fixed summation order, and y counts as written: the storing pass then keeps the static mapping, and
  // every workgroup has written its speculative stores back (release fence) before any partial sum of this pass is published --
  // i.e. before any workgroup can get as far as storing the same element again.  It runs BEFORE the sample is solved (an
  // inactive call never needs the sample's root: ~25 us of exchanges).
  if constexpr (!REG && VEC) {
    if (store_first && G > 1 && ntiles >= 8 * (int64_t)G) {
      double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
      bad = false;
      auto one0 = [&](double qv, double x, double s) -> double {
        const double sq = s + qv;
        const double lo = sq - ls, hi = sq + ls;
        acc(lo, hi, x, 1.0, v[0], v[1]);
        return outv(lo, hi, x, s, 1.0, 1.0);
      };
      if (n2 > 0)
        stream([&](bool valid, int64_t i, f64x2 a, f64x2 b, f64x2 d) {
          if (valid) b2_st<VEC>(y, i, f64x2{one0(a.x, b.x, d.x), one0(a.y, b.y, d.y)});
        }, true);
      if ((n & 1) && blockIdx.x == 0 && t == 0) y[n - 1] = one0(q[n - 1], xk[n - 1], sj[n - 1]);
      // (one wavefront per workgroup issues the write-back, after all of the workgroup's stores have left: a fence per
      //  wavefront -- 4096 L2 write-back scans -- made the pass slower than the static one)
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t < 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (bad) v[0] = __longlong_as_double(0x7ff8000000000000ll);
      reduce(v, 3u);
      const double chi0 = chil * sqrt(v[0] + v[1]);
      if (delta > chi0 * (1.0 + 1e-12)) {  // (NaN sums compare false: the ordinary pass decides)
        if (blockIdx.x == 0 && t == 0) hdr->b2_last_scaled = 0;
        return;  // (after the exchange: every workgroup formed the same sums and takes the same way)
      }
      store_first = false;
      y_written = true;
      B2_STAMP();
    }
  }
  // ---- the sample's root (streaming form): same iteration, chi scaled by sqrt(n / sample size), nothing stored
  double eta_s = -1.0;
  if (has_sample) {
    const double chis = chil * sqrt((double)n / (double)nsample);
    pass_small(1.0, 1.0, true, false, true);
    if (delta <= chis * sqrt(P + C)) {
      double lo = delta, hi = INFINITY, pP = -1.0, pC = -1.0, eta = delta;
      bool exact_step = false;
      const double ub = chis * sqrt(F);
      if (ub > delta && ub < INFINITY) { eta = ub; pass_small(eta / delta, 1.0, false, false, true); }
      for (int it = 0; it < 20; ++it) {
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
        pass_small(eta / delta, 1.0, false, false, true);
      }
      if (eta > delta && eta < INFINITY) eta_s = eta;
    }
  }
  // ---- first pass over the vectors: y = ProjB(-xk) (:59), chi(y) = chi_lambda ||y|| -- at r = 1, ||y||^2 = P + C.
  // Streaming form with a sample root: every element is also classified against the bracket [eta_a, eta_b] around it.
  // Its clamp state (below lo / inside / above hi) moves one way only as r grows, so equal states at both ends mean that
  // state on the whole bracket: a fixed term of P (inside: x^2) or of C (clamped: bound^2).  The others -- a breakpoint
  // inside the bracket, 1-2 % of the vector -- are recorded (x, sj + q) in this wavefront's own region: no atomics.
  double eta_a = -1.0, eta_b = -1.0, Pf = 0.0, Cf = 0.0;
  bool have_bracket = false;
  unsigned int ncand = 0;    // (wave-uniform) candidates of this wavefront
  f64x2* myreg = nullptr;
  if constexpr (REG) {
    pass_small(1.0, 1.0, true, store_first, false);
  } else {
    have_bracket = eta_s > 0.0 && cand != nullptr;
    if (have_bracket) {
      eta_a = eta_s * (1.0 - kB2Bracket);
      eta_b = eta_s * (1.0 + kB2Bracket);
      if (eta_a < delta) eta_a = delta;   // the root is never below Delta (froot(Delta) <= 0 in the scaled branch)
    }
    const double ra = eta_a / delta, rb = eta_b / delta;
    const int lane = t & 63;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    myreg = cand + ((size_t)blockIdx.x * (THREADS / 64) + (size_t)(t >> 6)) * cand_cap;
    double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // p, c, f at r = 1; fixed P, fixed C; candidates lost to a full region
    bad = false;
    auto one = [&](bool valid, double qv, double x, double s) -> double {
      const double sq = s + qv;
      const double lo = sq - ls, hi = sq + ls;
      double o = 0.0;
      if (valid) {
        const double pz = acc(lo, hi, x, 1.0, v[0], v[1]);
        const double far = (x < 0.0) ? hi : (x > 0.0) ? lo : pz;
        v[2] = __builtin_fma(far, far, v[2]);
        if (store_first) o = outv(lo, hi, x, s, 1.0, 1.0);
      }
      if (have_bracket) {  // (wave-uniform; every lane takes part in the ballot)
        const double za = (-x) * ra, zb = (-x) * rb;
        const double pa = fmin(fmax(za, lo), hi), pb = fmin(fmax(zb, lo), hi);
        const bool ua = pa == za, ub = pb == zb;
        const bool fixp = valid && ua && ub, fixc = valid && !ua && !ub && pa == pb;
        const double xm = fixp ? x : 0.0, cm = fixc ? pa : 0.0;
        v[3] = __builtin_fma(xm, xm, v[3]);
        v[4] = __builtin_fma(cm, cm, v[4]);
        const bool is_cand = valid && !fixp && !fixc;
        const unsigned long long m = __ballot(is_cand);
        if (m) {
          const unsigned int pos = ncand + (unsigned int)__popcll(m & lt_mask);
          if (is_cand) {
            if (pos < cand_cap) myreg[pos] = f64x2{x, sq};
            else v[5] += 1.0;
          }
          ncand += (unsigned int)__popcll(m);
        }
      }
      return o;
    };
    if (store_first) y_written = true;
    if (n2 > 0)
      stream([&](bool valid, int64_t i, f64x2 a, f64x2 b, f64x2 d) {
        const f64x2 o{one(valid, a.x, b.x, d.x), one(valid, a.y, b.y, d.y)};
        if (store_first && valid) b2_st<VEC>(y, i, o);
      });
    if (n & 1) {  // the odd last element rides with wavefront 0 of workgroup 0 (all of its lanes call `one`)
      if (blockIdx.x == 0 && t < 64) {
        const double o = one(t == 0, q[n - 1], xk[n - 1], sj[n - 1]);
        if (store_first && t == 0) y[n - 1] = o;
      }
    }
    if (bad) v[0] = __longlong_as_double(0x7ff8000000000000ll);
    reduce(v, have_bracket ? 63u : 7u);
    P = v[0]; C = v[1]; F = v[2]; Pf = v[3]; Cf = v[4];
    if (v[5] != 0.0) have_bracket = false;  // a region overflowed (clustered breakpoints): the plain iteration below
    B2_STAMP();
  }
  const unsigned int ncand_mine = ncand < cand_cap ? ncand : cand_cap;
  // sums over this wavefront's candidates at scale r, added to the fixed part: the exact (P, C) of the whole vector at any
  // eta inside the bracket, for one reduction of a few per cent of the data.  Up to three scales per sweep (the bracket's two
  // ends and the sample's root in ONE reduction: r[k] <= 0 = unused); totals in Pm[k], Cm[k], the first also in P, C.
  double Pm[3] = {0.0, 0.0, 0.0}, Cm[3] = {0.0, 0.0, 0.0};
  auto pass_cand = [&](double r0, double r1, double r2) {
    double v[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    bad = false;
    if constexpr (!REG) {
      for (unsigned int e = (unsigned int)(t & 63); e < ncand_mine; e += 64) {
        const f64x2 rec = myreg[e];
        const double lo = rec.y - ls, hi = rec.y + ls;
        acc(lo, hi, rec.x, r0, v[0], v[1]);
        if (r1 > 0.0) acc(lo, hi, rec.x, r1, v[2], v[3]);
        if (r2 > 0.0) acc(lo, hi, rec.x, r2, v[4], v[5]);
      }
    }
    if (bad) v[0] = v[2] = v[4] = __longlong_as_double(0x7ff8000000000000ll);
    reduce(v, r2 > 0.0 ? 63u : (r1 > 0.0 ? 15u : 3u));
#pragma unroll
    for (int k = 0; k < 3; ++k) { Pm[k] = Pf + v[2 * k]; Cm[k] = Cf + v[2 * k + 1]; }
    P = Pm[0]; C = Cm[0];
    B2_STAMP();
  };
  const double chiy = chil * sqrt(P + C);
  // :61 `Delta <= chi(y)`.  With EQUALITY froot(Delta) = Delta - chi(y) is zero: find_zero returns its starting point Delta,
  // eta / Delta = 1 and the scaled branch reproduces y = ProjB(-xk) -- the unscaled result.  (The bracket below takes
  // froot(Delta) < 0 for granted: on integer lattice data, where chi(y) == Delta happens, it bisected towards a root it
  // could never accept and ran out of reductions 3e-10 away: tests/test_gpu_stress.py::test_b2_integer_lattices_exact_roots.)
  const bool scaled = delta < chiy;
  if (blockIdx.x == 0 && t == 0) hdr->b2_last_scaled = scaled ? 1 : 0;
  double eta = delta;
  bool stored = !scaled && store_first;
  if (scaled) {
    double lo = delta, hi = INFINITY, pP = -1.0, pC = -1.0;
    bool exact_step = false;
    bool hi_closed = false;   // hi is the a-priori bound, NOT evaluated: froot(hi) >= 0, possibly = 0 (x = 0: the bound IS the root)
    bool in_bracket = false;  // reductions run over the candidates (exact inside [eta_a, eta_b]) instead of the vectors
    const double eta_ub = chil * sqrt(F);
    const bool ub_ok = eta_ub > delta && eta_ub < INFINITY;
    bool have_eval = false;
    const double P1 = P, C1 = C;  // (the sums at r = 1)
    if (have_bracket) {
      // does the bracket hold the root?  froot at both ends and at the sample's root, exact on aggregate + candidates, one sweep
      const double eta_c = (eta_s > eta_a && eta_s < eta_b) ? eta_s : 0.5 * (eta_a + eta_b);
      pass_cand(eta_a / delta, eta_b / delta, eta_c / delta);
      auto fr = [&](double e, double Pe, double Ce) { return e - chil * sqrt((e / delta) * (e / delta) * Pe + Ce); };
      const double fa = fr(eta_a, Pm[0], Cm[0]), fb = fr(eta_b, Pm[1], Cm[1]), fc = fr(eta_c, Pm[2], Cm[2]);
      if (fa <= 0.0 && fb >= 0.0) {
        in_bracket = true;
        have_eval = true;
        // start from the evaluated point nearest the root: the sample's (2e-3 away; the piece roots converge quadratically)
        if (fa == 0.0) { lo = hi = eta_a; eta = eta_a; P = Pm[0]; C = Cm[0]; }
        else if (fc < 0.0) { lo = eta_c; hi = eta_b; eta = eta_c; P = Pm[2]; C = Cm[2]; }
        else { lo = eta_a; hi = eta_c; eta = eta_c; P = Pm[2]; C = Cm[2]; }
      // (a miss -- the sample misled by more than 1.5 % -- costs this one cheap reduction and then the plain iteration)
      } else {
        P = P1; C = C1;
      }
    }
    if (!have_eval && ub_ok) {
      eta = eta_ub;
      if constexpr (REG) pass_small(eta / delta, 1.0, false, false, false);
      else pass_stream(eta / delta, 1.0, false);
    }
    double y_eta = -1.0;
    double prev_step = -1.0;  // relative size of the previous exact step
    for (int it = 0; it < kB2MaxPass - 34 && np < kB2MaxPass - 1; ++it) {
      const double r = eta / delta;
      const double f = eta - chil * sqrt(r * r * P + C);
      if (f == 0.0 || (exact_step && P == pP && C == pC)) break;
      if (f < 0.0) lo = eta; else { hi = eta; hi_closed = false; }
      const double den = 1.0 - chil * chil * P / (delta * delta);
      double next = (den > 0.0) ? chil * sqrt(C / den) : INFINITY;
      exact_step = (next > lo && (next < hi || (hi_closed && next == hi)));
      if (!exact_step) next = (hi == INFINITY) ? 2.0 * lo : 0.5 * (lo + hi);
      if (!(next > lo && (next < hi || (hi_closed && next == hi)))) break;
      const double step = fabs(next - eta) / next;
      if (step <= 4e-16) break;
      if (in_bracket) {
        // a reduction over the candidates costs microseconds: iterate until the piece is confirmed (P, C identical) -- the
        // exact fixed point, no stopping rule
        pP = P; pC = C; eta = next;
        pass_cand(eta / delta, -1.0, -1.0);
        continue;
      }
      // Reductions over the VECTORS (register-resident form; streaming form without a usable bracket).  The piece roots converge
      // quadratically on generic data (measured steps 4e-2, 1e-4, 6e-10: error(next) ~ 0.06 step^2), and a breakpoint between
      // eta and the root changes the root only to second order (the pieces join continuously).  `next` is taken without the
      // pass that would only confirm it -- but on the evidence of the more pessimistic LINEAR model: with rho = step / prev_step
      // the error left after this step is at most rho step / (1 - rho); below 2e-13 (a fifth of the 1e-12 bar) the iteration ends.
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
      if constexpr (REG) pass_small(eta / delta, delta / eta, false, spec, false);
      else pass_stream(eta / delta, delta / eta, spec);
      y_eta = spec ? eta : -1.0;
    }
    stored = (y_eta == eta);
  }
#ifdef SPX_B2_PROFILE
  if (stored && blockIdx.x == 0 && t == 0) g_b2_nstamp = nst;
#endif
  if (stored && !spx_poisoned(hdr)) return;  // (after the last exchange; every workgroup takes the same path)
  // final: y = ProjB((-xk) r) rinv - sj   (:63, :65), or ProjB(-xk) - sj (:59) when the trust region is inactive.  A workgroup
  // that gave up waiting (spx_wait_expired) has made every sum garbage: NaN everywhere, the next libspx call reports it.
  double r = scaled ? eta / delta : 1.0, rinv = scaled ? delta / eta : 1.0;
  if (spx_poisoned(hdr)) r = rinv = __longlong_as_double(0x7ff8000000000000ll);
  store_pass(r, rinv);
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
  SPX_ON_DEVICE(ctx);
  const double ls = lambda * sigma;  // `psi.lambda * sigma`, :56
  const bool vec = n >= 2 && spx_aligned16(y) && spx_aligned16(q) && spx_aligned16(xk) && spx_aligned16(sj);
  // Residency (spx_resident_cap): the grid of a launch that synchronises inside itself never exceeds what can be resident
  // at once; the streaming form works with any grid >= 1, the register-resident one needs ceil(n / 8192) workgroups.
  const int64_t cap_reg = spx_resident_cap(ctx, reinterpret_cast<const void*>(&k_b2_coop<true, kB2Epl, kB2RegThreads, true>), kB2RegThreads, 0);
  const int64_t cap_mem = vec ? spx_resident_cap(ctx, reinterpret_cast<const void*>(&k_b2_coop<false, 1, 1024, true>), 1024, 0)
                              : spx_resident_cap(ctx, reinterpret_cast<const void*>(&k_b2_coop<false, 1, 1024, false>), 1024, 0);
  if (cap_mem < 1) return SPX_ERR_INTERNAL;  // (message set by spx_resident_cap)
  const int64_t gmax_reg = cap_reg < kB2Cols ? cap_reg : kB2Cols;
  const int64_t gmax_mem = cap_mem < kB2Cols ? cap_mem : kB2Cols;
  const bool reg = n <= (int64_t)kB2RegBlock * gmax_reg;
  // xk parked in LDS, 16 Ki elements per workgroup: between what the register form holds and 4 Mi (256 CUs); tuning key 12 = 0
  // sends those sizes to the streaming form as in round 2 (n = 4e6: 111 us per call against 61)
  bool ldsx = false;
  int64_t gmax_lds = 0;
  if (!reg && ctx->tune_b2_lds) {
    const int64_t cap_lds = spx_resident_cap(ctx, reinterpret_cast<const void*>(&k_b2_coop<true, kB2Epl, 1024, true, true>), 1024, 0);
    gmax_lds = cap_lds < kB2Cols ? cap_lds : kB2Cols;
    ldsx = n <= (int64_t)kB2Epl * 1024 * gmax_lds;
  }
  int64_t g = reg ? (n + kB2RegBlock - 1) / kB2RegBlock : gmax_mem;
  if (ldsx) g = gmax_lds;  // (n > 2 Mi here unless the grid is capped: every resident workgroup, <= 16 elements per lane)
  if (reg) {
    // the register form: 4 / 8 / 16 elements per lane by n -- more workgroups sweep fewer elements each, until their exchange
    // costs more than the sweep saves (us per call, Delta = 1, 16 / 8 / 4 / 2 per lane: n = 1e4 22.2 / 21.3 / 20.5 / 20.5;
    // n = 1e5 25.4 / 23.3 / 22.5 / 23.4; n = 3e5 26.6 / 24.8 / 25.4 / 28.3; n = 1e6 30.4 / 31.9 / 31.1 / 31.4 -- tools/r3/b2_midn.py)
    const int64_t epl = n <= 131072 ? 4 : n <= 524288 ? 8 : kB2Epl;
    int64_t gg = (n + epl * kB2RegThreads - 1) / (epl * kB2RegThreads);
    if (gg > gmax_reg) gg = gmax_reg;
    if (gg > g) g = gg;
  }
  if (g < 1) g = 1;
  // streaming form: candidate regions, one per wavefront of the grid -- room for 8 % of its share of the vector (the bracket
  // of +-1.5 % around the sample's root holds the breakpoints of 1-2 % on ordinary data; a full region = plain iteration)
  f64x2* cand = nullptr;
  unsigned int cand_cap = 0;
  if (!reg && !ldsx) {
    const int64_t waves = g * 16;
    const int64_t share = (n + waves - 1) / waves;
    int64_t cap = share / 12 + 64;
    if (cap > 0x7fffffff) cap = 0x7fffffff;
    cand_cap = (unsigned int)cap;
    rc = spx_ws_reserve(ctx, (size_t)waves * (size_t)cap * sizeof(f64x2) + 256);
    if (rc) return rc;
    cand = reinterpret_cast<f64x2*>(ctx->ws);
  }
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
    if (g > 1) {
      rc = spx_zero2d_async(ctx, sets, (size_t)kB2Cols * kB2Words * sizeof(unsigned long long),
                            (size_t)g * kB2Words * sizeof(unsigned long long), (size_t)kB2MaxPass);
      if (rc) return rc;
    }
  }
  unsigned long long* rows = sets + (size_t)use * kB2SetWords;
  unsigned long long* clear_rows = sets + (size_t)other * kB2SetWords;
  {
    SpxCoopLaunchGuard guard(ctx);
    if (ldsx)
      hipLaunchKernelGGL((k_b2_coop<true, kB2Epl, 1024, true, true>), dim3((unsigned)g), dim3(1024), 0, ctx->stream, y, q,
                         xk, sj, n, ls, delta, chi_lambda, rows, clear_rows, clear_g, hdr, can_spec, (f64x2*)nullptr, 0u);
    else if (reg)
      hipLaunchKernelGGL((k_b2_coop<true, kB2Epl, kB2RegThreads, true>), dim3((unsigned)g), dim3(kB2RegThreads), 0, ctx->stream, y, q,
                         xk, sj, n, ls, delta, chi_lambda, rows, clear_rows, clear_g, hdr, can_spec, (f64x2*)nullptr, 0u);
    else if (vec)
      hipLaunchKernelGGL((k_b2_coop<false, 1, 1024, true>), dim3((unsigned)g), dim3(1024), 0, ctx->stream, y, q, xk, sj, n, ls,
                         delta, chi_lambda, rows, clear_rows, clear_g, hdr, can_spec, cand, cand_cap);
    else
      hipLaunchKernelGGL((k_b2_coop<false, 1, 1024, false>), dim3((unsigned)g), dim3(1024), 0, ctx->stream, y, q, xk, sj, n, ls,
                         delta, chi_lambda, rows, clear_rows, clear_g, hdr, can_spec, cand, cand_cap);
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
