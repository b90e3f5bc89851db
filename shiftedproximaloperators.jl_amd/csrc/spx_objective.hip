// spx_objective.hip -- psi(y) = h(xk + sj + y) (+ trust-region / box indicator): the objective value the solvers
// evaluate next to every prox!.  Reference: generic form src/ShiftedProximalOperators.jl:51-54;
// Box forms (feasibility scan with sqrt(eps) slack) src/shiftedNormL1Box.jl:70-82, shiftedNormL0Box.jl:70-82,
// shiftedRootNormLhalfBox.jl:67-79; BInf forms (IndBallLinf(1.1 Delta) on sj + y) src/shiftedIndBallL0BInf.jl:44-49,
// shiftedGroupNormL2Binf.jl:34-39.
//
// A reduction: reads y, xk, sj (24 B/element; +16 with vector bounds, +1 with a mask), writes one double.
// Per-lane partial sums -> wavefront butterfly -> workgroup (LDS) -> one partial per workgroup in the context
// scratch -> a single workgroup adds the partials in index order: the value is reproducible run to run.  The
// summation order differs from the reference's left-to-right loop, hence a 1e-12 relative tolerance in tests;
// counts (NormL0, IndBallL0) and the +Inf decisions are exact.  The last kernel turns sum and flags into psi(y) on the
// device.  By default the value is then returned to the host (these entry points synchronise the stream); with
// spx_ctx_set_value_target(ctx, device_double) it is stored in the caller's device double instead and the call returns
// after enqueueing -- a solver's accept / reject test can consume it on the device, or read it back when it needs it.
#include <cmath>
#include <limits>

#include "spx_common.hpp"

namespace {

constexpr int kObjBlocks = 2048;

struct ObjWs {
  double partial[kObjBlocks];
  double result;
  int infeasible;
  int pad;
};

// element terms of h
// (float overloads: the Float32 forms.  The reference forms xsy and every term in Float32 -- which entries are exactly zero,
//  the rounding of each square root -- and adds them up in Float32; here the terms are Float32 values, the SUM is Float64.)
struct TermL1 {  // NormL1 [ext]
  __device__ __forceinline__ double operator()(double v) const { return fabs(v); }
  __device__ __forceinline__ double operator()(float v) const { return (double)fabsf(v); }
};
struct TermL0 {  // NormL0, IndBallL0 [ext]
  __device__ __forceinline__ double operator()(double v) const { return (v != 0.0) ? 1.0 : 0.0; }
  __device__ __forceinline__ double operator()(float v) const { return (v != 0.0f) ? 1.0 : 0.0; }
};
struct TermLhalf {  // src/rootNormLhalf.jl:27-29
  __device__ __forceinline__ double operator()(double v) const { return sqrt(fabs(v)); }
  __device__ __forceinline__ double operator()(float v) const { return (double)__builtin_sqrtf(fabsf(v)); }
};

__device__ __forceinline__ double block_sum(double v, double* lds4) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) lds4[w] = v;
  __syncthreads();
  return (lds4[0] + lds4[1]) + (lds4[2] + lds4[3]);
}

// How the reduced sum and the infeasibility flags turn into psi(y):
enum ObjRule {
  kRuleScaled = 0,   // lambda * sum                                        (generic forms)
  kRuleBox = 1,      // flag ? +Inf : lambda * sum                           (Box forms: feasibility scan)
  kRuleCount = 2,    // (flag || sum > limit) ? +Inf : 0                     (IndBallL0, IndBallL0BInf)
  kRuleGroup = 3,    // bad index ? NaN : flag ? +Inf : sum                  (GroupNormL2(Binf))
};
__device__ __forceinline__ double obj_value(int rule, int flag, double acc, double scale, double limit) {
  const double inf = __longlong_as_double(0x7ff0000000000000ll);
  if (rule == kRuleScaled) return scale * acc;
  if (rule == kRuleBox) return flag ? inf : scale * acc;
  if (rule == kRuleCount) return (flag || acc > limit) ? inf : 0.0;
  return (flag & 2) ? __longlong_as_double(0x7ff8000000000000ll) : (flag ? inf : acc);
}

// Round 4: psi(y) in ONE launch.  A call used to be three -- the flag's zero-fill, the reduction, k_obj_final -- and below
// ~1e6 elements a call costs what its launches cost (tools/r4/small_latency_all.py: 15-16 us against 9 for a prox!).  With
// `hdr` set, the workgroup that takes the last ticket (spx_fin_ticket) does k_obj_final's work itself: the partials travel as
// agent-scope atomic stores / loads, are added in the same order (the same bits), the infeasibility bits live in the
// synchronisation state (zero between launches: the last workgroup resets them) instead of library scratch that other
// operators write over.  hdr == NULL: the kernel only leaves its partials and bits in `ws` for a later k_obj_final (the
// forms whose flags have a second source: ragged Binf layouts, the chunked sums of large groups).
struct ObjFin {
  SpxSyncHeader* hdr;
  double* target;  // spx_ctx::value_target (may be NULL)
  double scale, limit;
  int rule;
};
__device__ __forceinline__ int* obj_flag_word(ObjWs* ws, const ObjFin& fin) { return fin.hdr ? &fin.hdr->fin_flag : &ws->infeasible; }
// raises `bits` from the first lane of a wavefront that saw them; the atomic has been PERFORMED when the wavefront goes on
// (its return value is waited for), so the ticket its workgroup takes later is ordered behind it
__device__ __forceinline__ void obj_raise(int* flagp, bool any, int bits) {
  if (any && (threadIdx.x & 63) == 0 && (__hip_atomic_load(flagp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & bits) != bits) {
    const int old = __hip_atomic_fetch_or(flagp, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" ::"v"(old));
  }
}
// `acc`: this workgroup's sum (block_sum: the same in every lane).  Every wavefront has raised its bits before.
__device__ __forceinline__ void obj_finish_block(double acc, ObjWs* ws, const ObjFin& fin, double* lds4) {
  if (fin.hdr == nullptr) {
    if (threadIdx.x == 0) ws->partial[blockIdx.x] = acc;
    return;
  }
  __shared__ int is_last;
  if (threadIdx.x == 0) {
    spx_atomic_store_f64(&ws->partial[blockIdx.x], acc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    is_last = spx_fin_ticket(fin.hdr) ? 1 : 0;
  }
  __syncthreads();
  if (!is_last) return;
  double a = 0.0;
  for (int b = threadIdx.x; b < (int)gridDim.x; b += 256) a += spx_atomic_load_f64(&ws->partial[b]);  // (k_obj_final's order)
  a = block_sum(a, lds4);
  if (threadIdx.x == 0) {
    const int flag = __hip_atomic_load(&fin.hdr->fin_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double v = obj_value(fin.rule, flag, a, fin.scale, fin.limit);
    ws->result = v;        // (read back by the host when there is no device target: {result, infeasible})
    ws->infeasible = flag;
    if (fin.target) *fin.target = v;
    __hip_atomic_store(&fin.hdr->fin_flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// MODE 0: xsy = (xk + sj) + y over every index                              (generic, :52)
// MODE 1: Box: the same over the selected indices, plus the feasibility scan of sj + y against
//         [l - sqrt(eps), u + sqrt(eps)] over EVERY index                  (shiftedNormL1Box.jl:70-82)
// MODE 2: BInf: xsy = (sj + y) + xk, plus |sj + y| <= 1.1 Delta (strict IndBox test) (shiftedIndBallL0BInf.jl:44-49)
// T = double, or float (Float32 forms: every element operation in Float32 as in the reference, `1.1 * Delta` and the
// comparison against it in Float64 -- Julia promotes the Float64 literal --, the sum in Float64).
template <class T, class Term, int MODE>
__global__ __launch_bounds__(256) void k_obj(const T* __restrict__ y, const T* __restrict__ xk,
                                              const T* __restrict__ sj, const T* __restrict__ lv,
                                              const T* __restrict__ uv, const uint8_t* __restrict__ mask,
                                              T ls, T us, double rad, int64_t n, Term term, ObjWs* ws, ObjFin fin) {
  __shared__ double lds4[4];
  const T slack = sizeof(T) == 8 ? (T)1.4901161193847656e-08 : (T)3.4526698300124393e-04;  // sqrt(eps(T))
  double acc = 0.0;
  bool bad = false;
  auto visit = [&](T yi, T xi, T si, T lo, T up, bool sel) {
    if constexpr (MODE == 2) {
      const T t = si + yi;
      bad |= ((double)t < -rad) || ((double)t > rad);
      acc += term((T)(t + xi));
    } else {
      if constexpr (MODE == 1) {
        const T t = si + yi;
        bad |= !(((T)(lo - slack) <= t) && (t <= (T)(up + slack)));
        if (!sel) return;
      }
      acc += term((T)((T)(xi + si) + yi));
    }
  };
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const bool vec = sizeof(T) == 8 && (((uintptr_t)y | (uintptr_t)xk | (uintptr_t)sj | (uintptr_t)lv | (uintptr_t)uv) & 15) == 0 &&
                   (((uintptr_t)mask) & 1) == 0;
  if constexpr (sizeof(T) != 8) {  // Float32: 4-byte loads, a wavefront reads 256 contiguous bytes per vector
    for (int64_t i = tid; i < n; i += stride)
      visit(y[i], xk[i], sj[i], (MODE == 1 && lv) ? lv[i] : ls, (MODE == 1 && uv) ? uv[i] : us,
            (MODE == 1 && mask) ? mask[i] != 0 : true);
  } else if (vec && n >= 2) {  // 16-byte non-temporal loads, four pairs in flight per vector and lane
    const int64_t n2 = n >> 1;
    const f64x2* y2 = reinterpret_cast<const f64x2*>(y);
    const f64x2* x2 = reinterpret_cast<const f64x2*>(xk);
    const f64x2* s2 = reinterpret_cast<const f64x2*>(sj);
    const f64x2* l2 = reinterpret_cast<const f64x2*>(lv);
    const f64x2* u2 = reinterpret_cast<const f64x2*>(uv);
    const uint16_t* m2 = reinterpret_cast<const uint16_t*>(mask);
    const int64_t ntiles = (n2 + 1023) / 1024;  // a tile = 256 lanes x 4 pairs; 12 16-byte loads in flight per lane
    for (int64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
      const int64_t base = tile * 1024 + threadIdx.x;
      f64x2 yv[4], xv[4], sv[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t i = (base + k * 256 < n2) ? base + k * 256 : n2 - 1;
        yv[k] = __builtin_nontemporal_load(y2 + i);
        xv[k] = __builtin_nontemporal_load(x2 + i);
        sv[k] = __builtin_nontemporal_load(s2 + i);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int64_t i = base + k * 256;
        if (i < n2) {
          f64x2 la = f64x2{ls, ls}, ua = f64x2{us, us};
          uint16_t ma = 0x0101;
          if constexpr (MODE == 1) {
            if (lv) la = l2[i];
            if (uv) ua = u2[i];
            if (mask) ma = m2[i];
          }
          visit(yv[k].x, xv[k].x, sv[k].x, la.x, ua.x, (ma & 0xff) != 0);
          visit(yv[k].y, xv[k].y, sv[k].y, la.y, ua.y, (ma >> 8) != 0);
        }
      }
    }
    if ((n & 1) && tid == 0) {
      const int64_t i = n - 1;
      visit(y[i], xk[i], sj[i], lv ? lv[i] : ls, uv ? uv[i] : us, mask ? mask[i] != 0 : true);
    }
  } else {
    for (int64_t i = tid; i < n; i += stride)
      visit(y[i], xk[i], sj[i], (MODE == 1 && lv) ? lv[i] : ls, (MODE == 1 && uv) ? uv[i] : us,
            (MODE == 1 && mask) ? mask[i] != 0 : true);
  }
  // an infeasible point sets the flag from EVERY wavefront: once it is up nobody needs to queue on that one address again
  // (psi at an infeasible y of 1e8 elements: 0.45 -> 0.38 ms, the time of a feasible one)
  obj_raise(obj_flag_word(ws, fin), __any(bad), 1);
  acc = block_sum(acc, lds4);
  obj_finish_block(acc, ws, fin, lds4);
}

// GroupNormL2: sum_g lambda_g ||xsy[idx_g]||_2  (src/groupNormL2.jl:33-39).  TEAM lanes of a wavefront per group, 64 / TEAM
// groups per wavefront and trip (round 3: a whole wavefront per group left 62 lanes idle on groups of two -- 1.46 ms per call
// at n = 1.6e7 against 68 us on groups of 128; TEAM follows the group size, ~4 elements per lane: run_obj_group).
// PAIRS (Float64, uniform groups of even size, 16-byte aligned vectors): a lane reads 16-byte pairs.
template <class T, int MODE, int TEAM, bool PAIRS = false>
__global__ __launch_bounds__(256) void k_obj_group(const T* __restrict__ y, const T* __restrict__ xk,
                                                    const T* __restrict__ sj, int64_t n,
                                                    const int64_t* __restrict__ offsets, int64_t gsize, int64_t ngroups,
                                                    const int64_t* __restrict__ index /* NULL: contiguous groups */,
                                                    int64_t nnz, const T* __restrict__ lambda, double rad,
                                                    ObjWs* ws, ObjFin fin) {
  __shared__ double lds4[4];
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  constexpr int GPW = 64 / TEAM;  // groups per wavefront and trip
  const int j = lane % TEAM, slot = lane / TEAM;
  double acc = 0.0;  // lane 0 of each team accumulates lambda_g * norm_g
  bool bad = false, bad_index = false;
  for (int64_t g0 = wave * GPW; g0 < ngroups; g0 += nwaves * GPW) {  // wave-uniform trip count
    const int64_t g = g0 + slot;
    const bool live = g < ngroups;
    int64_t lo = 0, hi = 0;
    if (live) {
      if (offsets) { lo = offsets[g]; hi = offsets[g + 1]; }
      else { lo = g * gsize; hi = lo + gsize; }
    }
    if (lo < 0) lo = 0;
    if (hi > (index ? nnz : n)) hi = index ? nnz : n;
    double ss = 0.0;
    if constexpr (PAIRS) {
      const f64x2* y2 = reinterpret_cast<const f64x2*>(y);
      const f64x2* x2 = reinterpret_cast<const f64x2*>(xk);
      const f64x2* s2 = reinterpret_cast<const f64x2*>(sj);
      for (int64_t p = (lo >> 1) + j; p < (hi >> 1); p += TEAM) {
        const f64x2 a = y2[p], b = x2[p], c = s2[p];
        double v0, v1;
        if constexpr (MODE == 2) {
          const double t0 = c.x + a.x, t1 = c.y + a.y;
          bad |= (t0 < -rad) || (t0 > rad) || (t1 < -rad) || (t1 > rad);
          v0 = t0 + b.x; v1 = t1 + b.y;
        } else {
          v0 = (b.x + c.x) + a.x; v1 = (b.y + c.y) + a.y;
        }
        ss += v0 * v0;
        ss += v1 * v1;
      }
    } else
    for (int64_t p = lo + j; p < hi; p += TEAM) {
      int64_t i = p;
      if (index) {
        i = index[p];
        if (i < 0 || i >= n) { bad_index = true; continue; }
      }
      T v;
      if constexpr (MODE == 2) {
        const T t = sj[i] + y[i];
        bad |= ((double)t < -rad) || ((double)t > rad);
        v = t + xk[i];
      } else {
        v = (T)(xk[i] + sj[i]) + y[i];
      }
      ss += (double)v * (double)v;
    }
#pragma unroll
    for (int off = TEAM / 2; off >= 1; off >>= 1) ss += __shfl_xor(ss, off, 64);  // (inside the team's aligned lane range)
    if (live && j == 0) acc += (double)lambda[g] * sqrt(ss);
  }
  obj_raise(obj_flag_word(ws, fin), __any(bad), 1);
  obj_raise(obj_flag_word(ws, fin), __any(bad_index), 2);
  acc = block_sum(acc, lds4);
  obj_finish_block(acc, ws, fin, lds4);
}

// LARGE contiguous groups -- first of all ONE group over the whole vector, the reference's default GroupNormL2
// (src/groupNormL2.jl:30-31; psi(y) of shifted(NormL2(lambda), xk)): k_obj_group gives a group to (at most) one wavefront,
// i.e. the whole vector to 64 lanes (n = 1e8: 380 ms, profiles/r04_big_groups_baseline.txt).  Here every group is cut into
// chunks of kObjChunk elements, one 256-lane workgroup per chunk (16 elements per lane, 24 16-byte loads in flight), the
// chunk sums land in library scratch and k_obj_chunk_groups adds the chunks of each group in index order: reproducible.
constexpr int kObjChunk = 4096;
constexpr int kObjFuseGroups = 256;   // the one-launch form of the chunked psi(y): at most this many (uniform) groups ...
constexpr int kObjFuseChunks = 4096;  // ... and chunks (16 Mi elements): the last workgroup adds them, 16 loads per lane
// chunks before group g (CSR layouts; uniform groups need no table)
__global__ __launch_bounds__(1024) void k_obj_chunk_prefix(const int64_t* __restrict__ offsets, int64_t ngroups, int64_t n,
                                                            int64_t* __restrict__ prefix /* ngroups + 1 */) {
  __shared__ long long part[1024];
  const int t = threadIdx.x;
  const int64_t per = (ngroups + 1023) / 1024;
  const int64_t g0 = (int64_t)t * per, g1 = (g0 + per < ngroups) ? g0 + per : ngroups;
  auto chunks_of = [&](int64_t g) -> int64_t {
    int64_t lo = offsets[g], hi = offsets[g + 1];
    if (lo < 0) lo = 0;
    if (hi > n) hi = n;
    return hi > lo ? (hi - lo + kObjChunk - 1) / kObjChunk : 0;
  };
  long long mine = 0;
  for (int64_t g = g0; g < g1; ++g) mine += chunks_of(g);
  part[t] = mine;
  __syncthreads();
  if (t == 0) {
    long long acc = 0;
    for (int k = 0; k < 1024; ++k) { const long long c = part[k]; part[k] = acc; acc += c; }
    prefix[ngroups] = acc;
  }
  __syncthreads();
  long long acc = part[t];
  for (int64_t g = g0; g < g1; ++g) { prefix[g] = acc; acc += chunks_of(g); }
}
// sum of squares of xsy over one chunk -> chunk_ss[k].  par: bit 3 of the vectors' addresses when they agree (16-byte pairs
// from the first element on the pair grid, the at most two others by lane 0), -1: 8-byte loads.
template <int MODE>
__global__ __launch_bounds__(256) void k_obj_chunks(const double* __restrict__ y, const double* __restrict__ xk,
                                                     const double* __restrict__ sj, int64_t n,
                                                     const int64_t* __restrict__ offsets, int64_t gsize, int64_t ngroups,
                                                     const int64_t* __restrict__ prefix, int64_t nchunks_uniform, int par,
                                                     double rad, double* __restrict__ chunk_ss, ObjWs* ws,
                                                     ObjFin fin /* hdr != NULL (uniform groups, <= kObjFuseGroups of them, <= kObjFuseChunks
                                                                   chunks): the last workgroup also does k_obj_chunk_groups' and k_obj_final's work */,
                                                     const double* __restrict__ lambda) {
  __shared__ double lds4[4];
  const int t = threadIdx.x;
  const int64_t K = prefix ? prefix[ngroups] : nchunks_uniform;
  const int64_t cpg = (gsize + kObjChunk - 1) / kObjChunk;  // (uniform groups)
  bool bad = false;
  for (int64_t k = blockIdx.x; k < K; k += gridDim.x) {
    int64_t lo, hi;
    if (prefix) {
      int64_t a = 0, b = ngroups;  // the last g with prefix[g] <= k
      while (b - a > 1) {
        const int64_t m = (a + b) >> 1;
        if (prefix[m] <= k) a = m; else b = m;
      }
      int64_t glo = offsets[a], ghi = offsets[a + 1];
      if (glo < 0) glo = 0;
      if (ghi > n) ghi = n;
      lo = glo + (k - prefix[a]) * kObjChunk;
      hi = lo + kObjChunk < ghi ? lo + kObjChunk : ghi;
    } else {
      const int64_t g = k / cpg, c = k % cpg;
      lo = g * gsize + c * kObjChunk;
      hi = (c + 1) * (int64_t)kObjChunk < gsize ? lo + kObjChunk : (g + 1) * gsize;
    }
    double ss = 0.0;
    auto visit = [&](double yi, double xi, double si) {
      double v;
      if constexpr (MODE == 2) {
        const double tt = si + yi;
        bad |= (tt < -rad) || (tt > rad);
        v = tt + xi;
      } else {
        v = (xi + si) + yi;
      }
      ss = __builtin_fma(v, v, ss);
    };
    if (par >= 0) {
      const int64_t a = lo + ((lo + par) & 1);
      const int64_t np = a < hi ? (hi - a) >> 1 : 0;
      const f64x2* y2 = reinterpret_cast<const f64x2*>(y + a);
      const f64x2* x2 = reinterpret_cast<const f64x2*>(xk + a);
      const f64x2* s2 = reinterpret_cast<const f64x2*>(sj + a);
      f64x2 yv[8], xv[8], sv[8];
      if (np > 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int64_t i = (t + j * 256 < np) ? t + j * 256 : np - 1;
          yv[j] = __builtin_nontemporal_load(y2 + i);
          xv[j] = __builtin_nontemporal_load(x2 + i);
          sv[j] = __builtin_nontemporal_load(s2 + i);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (t + j * 256 < np) {
            visit(yv[j].x, xv[j].x, sv[j].x);
            visit(yv[j].y, xv[j].y, sv[j].y);
          }
        }
      }
      if (t == 0) {
        if (lo < a && lo < hi) visit(y[lo], xk[lo], sj[lo]);
        if (a + 2 * np < hi && hi - 1 >= a) visit(y[hi - 1], xk[hi - 1], sj[hi - 1]);
      }
    } else {
      for (int64_t i = lo + t; i < hi; i += 256) visit(y[i], xk[i], sj[i]);
    }
    ss = block_sum(ss, lds4);
    if (t == 0) {
      if (fin.hdr) spx_atomic_store_f64(chunk_ss + k, ss);
      else chunk_ss[k] = ss;
    }
  }
  obj_raise(obj_flag_word(ws, fin), __any(bad), 1);
  if (fin.hdr == nullptr) return;
  // Round 4 (third session): psi(y) of ONE group over the vector -- the reference's default GroupNormL2 -- was four launches at
  // solver sizes (flag zero-fill, chunk sums, group sums, final).  With few uniform groups the workgroup that takes the last
  // ticket adds each group's chunk sums in k_obj_chunk_groups' order (lane t: chunks c0 + t, c0 + t + 256, ...; block_sum),
  // forms lambda_g sqrt(.) and adds the groups as k_obj_final does with one group per slot: the same bits as the three launches.
  __shared__ int ck_last;
  __shared__ double ck_part[kObjFuseGroups];
  __syncthreads();  // (every wavefront has raised its bits)
  if (t == 0) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    ck_last = spx_fin_ticket(fin.hdr) ? 1 : 0;
  }
  __syncthreads();
  if (!ck_last) return;
  for (int64_t g = 0; g < ngroups; ++g) {
    const int64_t c0 = g * cpg, c1 = (g + 1) * cpg;
    double gs = 0.0;
    for (int64_t c = c0 + t; c < c1; c += 256) gs += spx_atomic_load_f64(chunk_ss + c);
    gs = block_sum(gs, lds4);
    if (t == 0) ck_part[g] = 0.0 + lambda[g] * sqrt(gs);  // (k_obj_chunk_groups: acc = 0.0; acc += lambda[g] * sqrt(ss))
  }
  __syncthreads();
  double a = 0.0;
  if (t < ngroups) a += ck_part[t];                      // (k_obj_final: lane t adds the slots t, t + 256, ...: one slot here)
  a = block_sum(a, lds4);
  if (t == 0) {
    const int flag = __hip_atomic_load(&fin.hdr->fin_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double v = obj_value(fin.rule, flag, a, fin.scale, fin.limit);
    ws->result = v;
    ws->infeasible = flag;
    if (fin.target) *fin.target = v;
    __hip_atomic_store(&fin.hdr->fin_flag, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// ws->partial[b] = sum over the groups b, b + grid, ... of lambda_g sqrt(sum of the group's chunk sums, in index order)
__global__ __launch_bounds__(256) void k_obj_chunk_groups(const double* __restrict__ chunk_ss, const int64_t* __restrict__ prefix,
                                                           int64_t cpg, int64_t ngroups, const double* __restrict__ lambda,
                                                           ObjWs* ws) {
  __shared__ double lds4[4];
  double acc = 0.0;
  for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const int64_t c0 = prefix ? prefix[g] : g * cpg, c1 = prefix ? prefix[g + 1] : (g + 1) * cpg;
    double ss = 0.0;
    for (int64_t c = c0 + threadIdx.x; c < c1; c += 256) ss += chunk_ss[c];
    ss = block_sum(ss, lds4);
    if (threadIdx.x == 0) acc += lambda[g] * sqrt(ss);
  }
  if (threadIdx.x == 0) ws->partial[blockIdx.x] = acc;
}

// IndBallLinf(1.1 Delta)(sj + y) over EVERY index (src/shiftedGroupNormL2Binf.jl:35-36), for groups that need not tile 1:n
template <class T>
__global__ __launch_bounds__(256) void k_obj_linf_scan(const T* __restrict__ y, const T* __restrict__ sj,
                                                        int64_t n, double rad, ObjWs* ws) {
  bool bad = false;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const T t = sj[i] + y[i];
    bad |= ((double)t < -rad) || ((double)t > rad);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0 && (__hip_atomic_load(&ws->infeasible, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1) == 0)
    atomicOr(&ws->infeasible, 1);
}

// ... the same over the indices OUTSIDE [offsets[0], offsets[ngroups]) only (contiguous CSR groups cover that range themselves;
// the chunked form has looked at every element of it already)
__global__ __launch_bounds__(256) void k_obj_linf_uncovered(const double* __restrict__ y, const double* __restrict__ sj, int64_t n,
                                                             const int64_t* __restrict__ offsets, int64_t ngroups, double rad,
                                                             ObjWs* ws) {
  int64_t head = offsets[0], tail0 = offsets[ngroups];
  if (head < 0) head = 0;
  if (head > n) head = n;
  if (tail0 < head) tail0 = head;
  if (tail0 > n) tail0 = n;
  const int64_t total = head + (n - tail0);
  bool bad = false;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
    const int64_t i = (t < head) ? t : tail0 + (t - head);
    const double v = sj[i] + y[i];
    bad |= (v < -rad) || (v > rad);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0 && (__hip_atomic_load(&ws->infeasible, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1) == 0)
    atomicOr(&ws->infeasible, 1);
}

// one workgroup: partials in index order (pairwise inside the wave, fixed shape) -> ws->result = psi(y) by `rule`;
// target != NULL: the value also goes to the caller's device double (spx_ctx_set_value_target)
__global__ __launch_bounds__(256) void k_obj_final(ObjWs* ws, int nblocks, int rule, double scale, double limit, double* target) {
  __shared__ double lds4[4];
  double acc = 0.0;
  for (int b = threadIdx.x; b < nblocks; b += 256) acc += ws->partial[b];
  acc = block_sum(acc, lds4);
  if (threadIdx.x == 0) {
    const int flag = ws->infeasible;
    const double v = obj_value(rule, flag, acc, scale, limit);
    ws->result = v;
    if (target) *target = v;
  }
}

// *value = psi(y) (read back, stream synchronised) -- or, with a device value target on the context, NaN on the host and
// the value in the target, nothing read back.  *flags = the infeasibility bits (0 in the device-target case).
int obj_finish(spx_ctx* ctx, ObjWs* ws, int blocks, int rule, double scale, double limit, double* value, int* flags,
               bool fused = false /* the reducing kernel's last workgroup has done k_obj_final's work (ObjFin) */) {
  if (!fused) {
    hipLaunchKernelGGL(k_obj_final, dim3(1), dim3(256), 0, ctx->stream, ws, blocks, rule, scale, limit, ctx->value_target);
    SPX_LAUNCH_CHECK();
  }
  *flags = 0;
  if (ctx->value_target) {
    *value = std::numeric_limits<double>::quiet_NaN();
    return SPX_OK;
  }
  struct { double r; int f; int p; } host;
  { const int rcc = spx_require_not_capturing(ctx, "returning psi(y) to the host"); if (rcc) return rcc; }
  SPX_HIP(hipMemcpyAsync(&host, &ws->result, sizeof(host), hipMemcpyDeviceToHost, ctx->stream));
  SPX_HIP(hipStreamSynchronize(ctx->stream));
  *value = host.r;
  *flags = host.f;
  return SPX_OK;
}

// The one-launch form (ObjFin): needs the synchronisation state's header; not while a stream capture has to allocate it.
int obj_fin_prepare(spx_ctx* ctx, int rule, double scale, double limit, ObjFin* fin) {
  *fin = ObjFin{nullptr, ctx->value_target, scale, limit, rule};
  if (!ctx->tune_fewer_launches) return SPX_OK;
  const int rc = spx_sync_reserve(ctx, sizeof(SpxSyncHeader));
  if (rc) return rc;
  fin->hdr = reinterpret_cast<SpxSyncHeader*>(ctx->sync);
  return SPX_OK;
}

template <class T, class Term, int MODE>
int run_obj(spx_ctx* ctx, const T* y, const T* xk, const T* sj, int64_t n, const T* lv,
            const T* uv, T ls, T us, const uint8_t* mask, double rad, int rule, double scale, double limit,
            double* value) {
  SPX_REQUIRE(ctx != nullptr, "ctx is NULL");
  SPX_REQUIRE(n >= 0, "n < 0");
  SPX_REQUIRE(n == 0 || (y && xk && sj), "NULL vector with n > 0");
  *value = 0.0;  // h of the empty vector (count rule: 0 <= limit)
  if (n == 0) {
    if (ctx->value_target) { const int rz = spx_zero_async(ctx, ctx->value_target, sizeof(double)); if (rz) return rz; }
    return SPX_OK;
  }
  int rc = spx_ws_reserve(ctx, sizeof(ObjWs) + 256);
  if (rc) return rc;
  SPX_ON_DEVICE(ctx);
  ObjFin fin;
  rc = obj_fin_prepare(ctx, rule, scale, limit, &fin);
  if (rc) return rc;
  ObjWs* ws = reinterpret_cast<ObjWs*>(ctx->ws);
  if (!fin.hdr) { const int rz = spx_zero_async(ctx, &ws->infeasible, sizeof(int)); if (rz) return rz; }
  int64_t blocks = (n + 256 * 8 - 1) / (256 * 8);
  if (blocks > kObjBlocks) blocks = kObjBlocks;
  hipLaunchKernelGGL((k_obj<T, Term, MODE>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, y, xk, sj, lv, uv, mask, ls,
                     us, rad, n, Term{}, ws, fin);
  SPX_LAUNCH_CHECK();
  int flags;
  return obj_finish(ctx, ws, (int)blocks, rule, scale, limit, value, &flags, fin.hdr != nullptr);
}

template <class T, int MODE>
int run_obj_group(spx_ctx* ctx, const T* y, const T* xk, const T* sj, int64_t n,
                  const int64_t* offsets, int64_t gsize, int64_t ngroups, const int64_t* index, int64_t nnz,
                  const T* lambda, double rad, double* value) {
  SPX_REQUIRE(ctx != nullptr && value != nullptr, "ctx or value is NULL");
  SPX_REQUIRE(n >= 0 && ngroups >= 0 && nnz >= 0, "negative size");
  *value = 0.0;
  if (n == 0) return SPX_OK;
  SPX_REQUIRE(y && xk && sj, "NULL vector");
  if (ngroups > 0) SPX_REQUIRE(lambda != nullptr, "lambda_vec is NULL");
  if (index || nnz > 0) SPX_REQUIRE(offsets != nullptr && index != nullptr, "group_ptr or group_index is NULL");
  if (!offsets && ngroups > 0)
    SPX_REQUIRE(gsize > 0 && ngroups <= n / gsize && ngroups * gsize == n, "ngroups * group_size != n");
  int rc = spx_ws_reserve(ctx, sizeof(ObjWs) + 256);
  if (rc) return rc;
  SPX_ON_DEVICE(ctx);
  ObjWs* ws = reinterpret_cast<ObjWs*>(ctx->ws);
  if constexpr (std::is_same<T, double>::value) {
    // large contiguous groups (uniform size, or CSR offsets with a large average): chunked, the whole chip on every group
    const int64_t avg = ngroups > 0 ? n / ngroups : 0;
    if (!index && ngroups > 0 && ctx->tune_team && (offsets ? avg >= 2 * kObjChunk : gsize >= 2 * kObjChunk)) {
      const int64_t cpg = offsets ? 0 : (gsize + kObjChunk - 1) / kObjChunk;
      const int64_t kmax = offsets ? n / kObjChunk + ngroups : cpg * ngroups;  // chunks (at most, for CSR layouts)
      const size_t ss_off = (sizeof(ObjWs) + 255) & ~(size_t)255;
      const size_t pre_off = ss_off + (((size_t)kmax * sizeof(double) + 255) & ~(size_t)255);
      rc = spx_ws_reserve(ctx, pre_off + (offsets ? (size_t)(ngroups + 1) * sizeof(int64_t) : 0) + 256);
      if (rc) return rc;
      ws = reinterpret_cast<ObjWs*>(ctx->ws);
      ObjFin cfin{nullptr, ctx->value_target, 1.0, 0.0, kRuleGroup};
      if (!offsets && ngroups <= kObjFuseGroups && kmax <= kObjFuseChunks) {  // few uniform groups: one launch (k_obj_chunks)
        rc = obj_fin_prepare(ctx, kRuleGroup, 1.0, 0.0, &cfin);
        if (rc) return rc;
        ws = reinterpret_cast<ObjWs*>(ctx->ws);
      }
      if (!cfin.hdr) { const int rz = spx_zero_async(ctx, &ws->infeasible, sizeof(int)); if (rz) return rz; }
      double* chunk_ss = reinterpret_cast<double*>(static_cast<char*>(ctx->ws) + ss_off);
      int64_t* prefix = offsets ? reinterpret_cast<int64_t*>(static_cast<char*>(ctx->ws) + pre_off) : nullptr;
      if (offsets) hipLaunchKernelGGL(k_obj_chunk_prefix, dim3(1), dim3(1024), 0, ctx->stream, offsets, ngroups, n, prefix);
      const auto bit3 = [](const void* p) { return (int)((reinterpret_cast<uintptr_t>(p) >> 3) & 1u); };
      const int par = (bit3(y) == bit3(xk) && bit3(y) == bit3(sj)) ? bit3(y) : -1;
      int64_t cb = kmax < (int64_t)ctx->num_cu * 64 ? kmax : (int64_t)ctx->num_cu * 64;
      if (cb < 1) cb = 1;
      hipLaunchKernelGGL((k_obj_chunks<MODE>), dim3((unsigned)cb), dim3(256), 0, ctx->stream, (const double*)y, (const double*)xk,
                         (const double*)sj, n, offsets, gsize, ngroups, (const int64_t*)prefix, cpg * ngroups, par, rad, chunk_ss, ws,
                         cfin, (const double*)lambda);
      if (cfin.hdr) {
        SPX_LAUNCH_CHECK();
        int bad_;
        return obj_finish(ctx, ws, 0, kRuleGroup, 1.0, 0.0, value, &bad_, true);
      }
      int64_t gb = ngroups < kObjBlocks ? ngroups : kObjBlocks;
      hipLaunchKernelGGL(k_obj_chunk_groups, dim3((unsigned)gb), dim3(256), 0, ctx->stream, (const double*)chunk_ss,
                         (const int64_t*)prefix, cpg, ngroups, (const double*)lambda, ws);
      if (MODE == 2 && offsets)  // the groups need not tile 0:n: the trust-region indicator covers every index
        hipLaunchKernelGGL(k_obj_linf_uncovered, dim3(256), dim3(256), 0, ctx->stream, (const double*)y, (const double*)sj, n, offsets,
                           ngroups, rad, ws);
      SPX_LAUNCH_CHECK();
      int bad_;
      return obj_finish(ctx, ws, (int)gb, kRuleGroup, 1.0, 0.0, value, &bad_);
    }
  }
  // lanes per group: about four elements per lane of a typical group (uniform size, the caller's size bound, or the average)
  const int64_t typical = gsize > 0 ? gsize : (ngroups > 0 ? ((index ? nnz : n) + ngroups - 1) / ngroups : 1);
  int team = 1;
  while (team < 64 && (int64_t)team * 4 < typical) team *= 2;
  const int gpw = 64 / team;
  int64_t blocks = (ngroups + 4 * gpw - 1) / (4 * gpw);
  if (blocks > kObjBlocks) blocks = kObjBlocks;
  if (blocks < 1) blocks = 1;
  // one launch unless the trust-region scan of a ragged layout (below) raises bits after this kernel
  ObjFin fin{nullptr, ctx->value_target, 1.0, 0.0, kRuleGroup};
  if (!(MODE == 2 && offsets)) {
    rc = obj_fin_prepare(ctx, kRuleGroup, 1.0, 0.0, &fin);
    if (rc) return rc;
    ws = reinterpret_cast<ObjWs*>(ctx->ws);
  }
  if (!fin.hdr) { const int rz = spx_zero_async(ctx, &ws->infeasible, sizeof(int)); if (rz) return rz; }
  constexpr bool kF64 = std::is_same<T, double>::value;
  const bool pairs = kF64 && !offsets && !index && gsize > 0 && (gsize & 1) == 0 && spx_aligned16(y) && spx_aligned16(xk) && spx_aligned16(sj);
#define SPX_OBJ_GROUP(TEAM)                                                                                                  \
  do {                                                                                                                       \
    if constexpr (kF64) {                                                                                                    \
      if (pairs) {                                                                                                           \
        hipLaunchKernelGGL((k_obj_group<T, MODE, TEAM, true>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, y, xk, sj, n, \
                           offsets, gsize, ngroups, index, nnz, lambda, rad, ws, fin);                                       \
        break;                                                                                                               \
      }                                                                                                                      \
    }                                                                                                                        \
    hipLaunchKernelGGL((k_obj_group<T, MODE, TEAM>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, y, xk, sj, n, offsets, \
                       gsize, ngroups, index, nnz, lambda, rad, ws, fin);                                                    \
  } while (0)
  switch (team) {
    case 1: SPX_OBJ_GROUP(1); break;
    case 2: SPX_OBJ_GROUP(2); break;
    case 4: SPX_OBJ_GROUP(4); break;
    case 8: SPX_OBJ_GROUP(8); break;
    case 16: SPX_OBJ_GROUP(16); break;
    case 32: SPX_OBJ_GROUP(32); break;
    default: SPX_OBJ_GROUP(64); break;
  }
#undef SPX_OBJ_GROUP
  if (MODE == 2 && offsets) {  // the groups need not tile 0:n: the trust-region indicator covers every index
    int64_t sb = (n + 256 * 8 - 1) / (256 * 8);
    if (sb > kObjBlocks) sb = kObjBlocks;
    hipLaunchKernelGGL(k_obj_linf_scan<T>, dim3((unsigned)sb), dim3(256), 0, ctx->stream, y, sj, n, rad, ws);
  }
  SPX_LAUNCH_CHECK();
  int bad;
  rc = obj_finish(ctx, ws, (int)blocks, kRuleGroup, 1.0, 0.0, value, &bad, fin.hdr != nullptr);
  if (rc) return rc;
  if (bad & 2) {  // (with a device value target the value is NaN instead: nothing is read back)
    spx_set_error("invalid argument: group index outside [0, n) (BoundsError)");
    return SPX_ERR_INVALID_ARG;
  }
  return SPX_OK;
}

const double kInf = std::numeric_limits<double>::infinity();

}  // namespace

// ---- generic forms: h(xk + sj + y) ---------------------------------------------------------------
#define SPX_OBJ_PLAIN(NAME, TERM)                                                                                  \
  SPX_EXPORT int NAME(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n, double lambda, \
                      double* value) {                                                                             \
    SPX_REQUIRE(value != nullptr, "value is NULL");                                                                \
    return run_obj<double, TERM, 0>(ctx, y, xk, sj, n, nullptr, nullptr, 0.0, 0.0, nullptr, 0.0, kRuleScaled, lambda, 0.0, value); \
  }
SPX_OBJ_PLAIN(spx_obj_l1, TermL1)
SPX_OBJ_PLAIN(spx_obj_l0, TermL0)
SPX_OBJ_PLAIN(spx_obj_lhalf, TermLhalf)

// ---- Box forms -----------------------------------------------------------------------------------------
#define SPX_OBJ_BOX(NAME, TERM)                                                                                    \
  SPX_EXPORT int NAME(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n, double lambda, \
                      const double* l_vec, const double* u_vec, double l_scalar, double u_scalar,                  \
                      const uint8_t* sel_mask, double* value) {                                                    \
    SPX_REQUIRE(value != nullptr, "value is NULL");                                                                \
    return run_obj<double, TERM, 1>(ctx, y, xk, sj, n, l_vec, u_vec, l_scalar, u_scalar, sel_mask, 0.0, kRuleBox, lambda, 0.0,    \
                            value);                                                                                \
  }
SPX_OBJ_BOX(spx_obj_l1_box, TermL1)
SPX_OBJ_BOX(spx_obj_l0_box, TermL0)
SPX_OBJ_BOX(spx_obj_lhalf_box, TermLhalf)

// ---- IndBallL0: 0 if at most r nonzeros, else +Inf  [ext: ProximalOperators.IndBallL0] ---------------------
SPX_EXPORT int spx_obj_indball_l0(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,
                                  int64_t r, double* value) {
  SPX_REQUIRE(value != nullptr, "value is NULL");
  return run_obj<double, TermL0, 0>(ctx, y, xk, sj, n, nullptr, nullptr, 0.0, 0.0, nullptr, 0.0, kRuleCount, 1.0, (double)r, value);
}
SPX_EXPORT int spx_obj_indball_l0_binf(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,
                                       int64_t r, double delta, double* value) {
  SPX_REQUIRE(value != nullptr, "value is NULL");
  return run_obj<double, TermL0, 2>(ctx, y, xk, sj, n, nullptr, nullptr, 0.0, 0.0, nullptr, 1.1 * delta, kRuleCount, 1.0, (double)r,
                            value);
}

// ---- GroupNormL2 ---------------------------------------------------------------------------------------
SPX_EXPORT int spx_obj_group_l2(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,
                                const int64_t* group_offsets, int64_t group_size, int64_t ngroups,
                                const double* lambda_vec, double* value) {
  return run_obj_group<double, 0>(ctx, y, xk, sj, n, group_offsets, group_size, ngroups, nullptr, 0, lambda_vec, 0.0, value);
}
SPX_EXPORT int spx_obj_group_l2_binf(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,
                                     const int64_t* group_offsets, int64_t group_size, int64_t ngroups,
                                     const double* lambda_vec, double delta, double* value) {
  return run_obj_group<double, 2>(ctx, y, xk, sj, n, group_offsets, group_size, ngroups, nullptr, 0, lambda_vec, 1.1 * delta,
                          value);
}

// groups as arbitrary index sets (group_ptr / group_index as for spx_prox_group_l2_gather)
SPX_EXPORT int spx_obj_group_l2_gather(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,
                                       const int64_t* group_ptr, const int64_t* group_index, int64_t ngroups,
                                       int64_t nnz, const double* lambda_vec, double* value) {
  SPX_REQUIRE(group_ptr != nullptr || ngroups == 0, "group_ptr is NULL");
  return run_obj_group<double, 0>(ctx, y, xk, sj, n, group_ptr, 0, ngroups, group_index, nnz, lambda_vec, 0.0, value);
}
SPX_EXPORT int spx_obj_group_l2_binf_gather(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,
                                            const int64_t* group_ptr, const int64_t* group_index, int64_t ngroups,
                                            int64_t nnz, const double* lambda_vec, double delta, double* value) {
  SPX_REQUIRE(group_ptr != nullptr || ngroups == 0, "group_ptr is NULL");
  return run_obj_group<double, 2>(ctx, y, xk, sj, n, group_ptr, 0, ngroups, group_index, nnz, lambda_vec, 1.1 * delta, value);
}

// ---- Float32 forms (the reference is generic in R <: Real and its tests evaluate every shifted operator on Float32 vectors:
// test/runtests.jl:196-209, 268-282, 346-360, 397-412, 524-550).  Element arithmetic in Float32 as in the reference
// ((xk + sj) + y, sj + y, the box ends -+ sqrt(eps(Float32)), each sqrt), `1.1 * Delta` and the comparison against it in
// Float64 (Julia promotes the literal), sums in Float64; the value comes back as a double (round it to Float32 to compare).
#define SPX_OBJ_PLAIN_F32(NAME, TERM)                                                                              \
  SPX_EXPORT int NAME(spx_ctx* ctx, const float* y, const float* xk, const float* sj, int64_t n, float lambda,     \
                      double* value) {                                                                             \
    SPX_REQUIRE(value != nullptr, "value is NULL");                                                                \
    return run_obj<float, TERM, 0>(ctx, y, xk, sj, n, nullptr, nullptr, 0.0f, 0.0f, nullptr, 0.0, kRuleScaled,     \
                                   (double)lambda, 0.0, value);                                                    \
  }
SPX_OBJ_PLAIN_F32(spx_obj_l1_f32, TermL1)
SPX_OBJ_PLAIN_F32(spx_obj_l0_f32, TermL0)
SPX_OBJ_PLAIN_F32(spx_obj_lhalf_f32, TermLhalf)
#define SPX_OBJ_BOX_F32(NAME, TERM)                                                                                \
  SPX_EXPORT int NAME(spx_ctx* ctx, const float* y, const float* xk, const float* sj, int64_t n, float lambda,     \
                      const float* l_vec, const float* u_vec, float l_scalar, float u_scalar,                      \
                      const uint8_t* sel_mask, double* value) {                                                    \
    SPX_REQUIRE(value != nullptr, "value is NULL");                                                                \
    return run_obj<float, TERM, 1>(ctx, y, xk, sj, n, l_vec, u_vec, l_scalar, u_scalar, sel_mask, 0.0, kRuleBox,   \
                                   (double)lambda, 0.0, value);                                                    \
  }
SPX_OBJ_BOX_F32(spx_obj_l1_box_f32, TermL1)
SPX_OBJ_BOX_F32(spx_obj_l0_box_f32, TermL0)
SPX_OBJ_BOX_F32(spx_obj_lhalf_box_f32, TermLhalf)
SPX_EXPORT int spx_obj_indball_l0_f32(spx_ctx* ctx, const float* y, const float* xk, const float* sj, int64_t n, int64_t r,
                                      double* value) {
  SPX_REQUIRE(value != nullptr, "value is NULL");
  return run_obj<float, TermL0, 0>(ctx, y, xk, sj, n, nullptr, nullptr, 0.0f, 0.0f, nullptr, 0.0, kRuleCount, 1.0, (double)r, value);
}
SPX_EXPORT int spx_obj_indball_l0_binf_f32(spx_ctx* ctx, const float* y, const float* xk, const float* sj, int64_t n,
                                           int64_t r, float delta, double* value) {
  SPX_REQUIRE(value != nullptr, "value is NULL");
  return run_obj<float, TermL0, 2>(ctx, y, xk, sj, n, nullptr, nullptr, 0.0f, 0.0f, nullptr, 1.1 * (double)delta, kRuleCount, 1.0,
                                   (double)r, value);
}
SPX_EXPORT int spx_obj_group_l2_f32(spx_ctx* ctx, const float* y, const float* xk, const float* sj, int64_t n,
                                    const int64_t* group_offsets, int64_t group_size, int64_t ngroups,
                                    const float* lambda_vec, double* value) {
  return run_obj_group<float, 0>(ctx, y, xk, sj, n, group_offsets, group_size, ngroups, nullptr, 0, lambda_vec, 0.0, value);
}
SPX_EXPORT int spx_obj_group_l2_binf_f32(spx_ctx* ctx, const float* y, const float* xk, const float* sj, int64_t n,
                                         const int64_t* group_offsets, int64_t group_size, int64_t ngroups,
                                         const float* lambda_vec, float delta, double* value) {
  return run_obj_group<float, 2>(ctx, y, xk, sj, n, group_offsets, group_size, ngroups, nullptr, 0, lambda_vec,
                                 1.1 * (double)delta, value);
}
