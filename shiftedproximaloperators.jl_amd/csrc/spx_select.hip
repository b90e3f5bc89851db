// spx_select.hip -- ShiftedIndBallL0.prox! / ShiftedIndBallL0BInf.prox!: keep the r entries of
// v = (xk + sj) + q largest in magnitude, zero the others, subtract xk + sj (and clamp to +-Delta).
//
// The reference sorts: sortperm!(p, y, rev = true, by = abs) (src/shiftedIndBallL0.jl:68,
// src/shiftedIndBallL0BInf.jl:87) -- a STABLE permutation, i.e. descending |v|, ties in ascending index.
// Only the set of the first r entries matters, so this is an exact top-r SELECTION with the composite
// order (|v| descending, index ascending).  For finite doubles |v| is ordered like the 63-bit integer
// bits(v) & 0x7ff..f, so the selection is an MSD radix select on that integer key:
//
//   pass 0      v -> y (y is the reference's scratch too, :66), LDS histogram of key digit 0 (12 bits)
//   scan        one workgroup walks the 4096 bins from the top, finds the bin holding the r-th largest,
//               updates (prefix, quota) in device memory; no host round trip
//   pass 1..5   re-read y (8 B/element), histogram the next digit of the keys matching the prefix
//   tie passes  only if the threshold key T is shared by more elements than the remaining quota:
//               the same machinery over the index digits of {i : key_i == T}, ascending
//   final       y[i] = (keep_i ? v_i : 0) - (xk[i] + sj[i])   [clamped for BInf]
//               keep_i = key_i >= t_ge || (key_i == t_eq && i <= icut)
//
// Every pass after the one that resolves the selection returns immediately (phase flag in device
// memory), so the usual cost is pass 0 (24 B read + 8 B write per element), 2-3 key passes (8 B) and the
// final pass (24 B read + 8 B write).  y may alias q: pass 0 reads q[i] before it writes y[i] and q is
// not needed afterwards.
#include "spx_common.hpp"

namespace {

constexpr int kDigitBits = 12;
constexpr int kBins = 1 << kDigitBits;
constexpr uint64_t kAbsMask = 0x7fffffffffffffffull;

struct SelState {
  int phase;          // 0 = key digits, 1 = index digits, 2 = resolved
  int shift;          // low bit of the digit the next histogram pass looks at
  int width;          // its width in bits
  int pad;
  uint64_t prefix;    // phase 0: decided high key bits (>> (shift + width)); phase 1: decided high index bits
  int64_t quota;      // how many elements are still to be taken from the current bucket
  uint64_t t_ge;      // keep if key >= t_ge
  uint64_t t_eq;      // keep if key == t_eq && index <= icut
  int64_t icut;
  int idx_bits;       // number of significant bits of n - 1
  int pad2;
};

struct SelWs {
  SelState st;
  unsigned long long hist[kBins];
};

__device__ __forceinline__ uint64_t key_of(double v) { return (uint64_t)__double_as_longlong(v) & kAbsMask; }

__global__ void k_sel_init(SelWs* ws, int64_t n, int64_t r) {
  const int t = threadIdx.x;
  for (int b = t; b < kBins; b += blockDim.x) ws->hist[b] = 0ull;
  if (t == 0) {
    SelState& s = ws->st;
    int bits = 0;
    while (bits < 63 && ((int64_t)1 << bits) < n) ++bits;
    s.idx_bits = bits;
    s.prefix = 0;
    s.icut = -1;
    s.t_eq = ~0ull;
    s.pad = s.pad2 = 0;
    if (r <= 0) {            // nothing kept
      s.phase = 2; s.t_ge = ~0ull; s.quota = 0; s.shift = 0; s.width = 0;
    } else if (r >= n) {     // everything kept
      s.phase = 2; s.t_ge = 0ull; s.quota = 0; s.shift = 0; s.width = 0;
    } else {
      s.phase = 0; s.shift = 64 - kDigitBits; s.width = kDigitBits; s.quota = r; s.t_ge = ~0ull;
    }
  }
}

// flush a workgroup's LDS histogram to the global one
__device__ __forceinline__ void flush_hist(unsigned int* lh, unsigned long long* gh) {
  __syncthreads();
  for (int b = threadIdx.x; b < kBins; b += blockDim.x) {
    unsigned int c = lh[b];
    if (c) atomicAdd(&gh[b], (unsigned long long)c);
  }
}

// pass 0: v = (xk + sj) + q -> y; histogram of the top key digit.  Runs even when the selection is
// already resolved (r <= 0 or r >= n) because the final pass reads v from y.
__global__ __launch_bounds__(256) void k_sel_pass0(double* y, const double* q, const double* xk, const double* sj,
                                                    int64_t n, int vec, SelWs* ws) {
  __shared__ unsigned int lh[kBins];
  for (int b = threadIdx.x; b < kBins; b += blockDim.x) lh[b] = 0u;
  __syncthreads();
  const bool count = (ws->st.phase == 0);
  const int shift = 64 - kDigitBits;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (vec) {
    const int64_t n2 = n >> 1;
    const f64x2* q2 = reinterpret_cast<const f64x2*>(q);
    const f64x2* x2 = reinterpret_cast<const f64x2*>(xk);
    const f64x2* s2 = reinterpret_cast<const f64x2*>(sj);
    f64x2* y2 = reinterpret_cast<f64x2*>(y);
    for (int64_t i = tid; i < n2; i += stride) {
      f64x2 a = q2[i], b = x2[i], c = s2[i];
      f64x2 v;
      v.x = (b.x + c.x) + a.x;  // shiftedIndBallL0.jl:66  y .= xk .+ sj .+ q
      v.y = (b.y + c.y) + a.y;
      y2[i] = v;
      if (count) {
        atomicAdd(&lh[key_of(v.x) >> shift], 1u);
        atomicAdd(&lh[key_of(v.y) >> shift], 1u);
      }
    }
    if ((n & 1) && tid == 0) {
      const int64_t i = n - 1;
      double v = (xk[i] + sj[i]) + q[i];
      y[i] = v;
      if (count) atomicAdd(&lh[key_of(v) >> shift], 1u);
    }
  } else {
    for (int64_t i = tid; i < n; i += stride) {
      double v = (xk[i] + sj[i]) + q[i];
      y[i] = v;
      if (count) atomicAdd(&lh[key_of(v) >> shift], 1u);
    }
  }
  if (count) flush_hist(lh, ws->hist);
}

// passes >= 1: histogram the current digit of the elements that match the decided prefix.
__global__ __launch_bounds__(256) void k_sel_hist(const double* y, int64_t n, int vec, SelWs* ws) {
  const SelState st = ws->st;
  if (st.phase == 2) return;
  __shared__ unsigned int lh[kBins];
  for (int b = threadIdx.x; b < kBins; b += blockDim.x) lh[b] = 0u;
  __syncthreads();
  const int shift = st.shift;
  const int hs = st.shift + st.width;  // bits above the current digit
  const uint64_t dmask = ((uint64_t)1 << st.width) - 1;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  auto visit = [&](double v, int64_t i) {
    const uint64_t key = key_of(v);
    if (st.phase == 0) {
      if ((hs >= 64 ? 0ull : (key >> hs)) == st.prefix) atomicAdd(&lh[(key >> shift) & dmask], 1u);
    } else {
      if (key == st.t_eq && (((uint64_t)i) >> hs) == st.prefix) atomicAdd(&lh[(((uint64_t)i) >> shift) & dmask], 1u);
    }
  };
  if (vec) {
    const int64_t n2 = n >> 1;
    const f64x2* y2 = reinterpret_cast<const f64x2*>(y);
    for (int64_t i = tid; i < n2; i += stride) {
      f64x2 v = y2[i];
      visit(v.x, 2 * i);
      visit(v.y, 2 * i + 1);
    }
    if ((n & 1) && tid == 0) visit(y[n - 1], n - 1);
  } else {
    for (int64_t i = tid; i < n; i += stride) visit(y[i], i);
  }
  flush_hist(lh, ws->hist);
}

// One workgroup: locate the bucket that holds the quota-th element (from the top for keys, from the
// bottom for indices), advance the state, clear the histogram for the next pass.
__global__ __launch_bounds__(256) void k_sel_scan(SelWs* ws) {
  __shared__ unsigned long long part[256];
  __shared__ int s_bucket;
  __shared__ unsigned long long s_before, s_count;
  SelState st = ws->st;
  if (st.phase == 2) return;
  const int t = threadIdx.x;
  const int nb = 1 << st.width;
  const bool desc = (st.phase == 0);
  // thread t owns the 16 consecutive bins [16 t, 16 t + 16) of the scan order
  constexpr int PER = kBins / 256;
  unsigned long long loc[PER];
  unsigned long long sum = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int pos = t * PER + k;                       // position in scan order
    const int bin = desc ? (kBins - 1 - pos) : pos;    // descending keys / ascending indices
    loc[k] = (bin < nb) ? ws->hist[bin] : 0ull;
    sum += loc[k];
  }
  part[t] = sum;
  __syncthreads();
  if (t == 0) {
    unsigned long long run = 0;
    for (int k = 0; k < 256; ++k) { unsigned long long c = part[k]; part[k] = run; run += c; }
  }
  __syncthreads();
  unsigned long long run = part[t];
  const unsigned long long quota = (unsigned long long)st.quota;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    if (run < quota && run + loc[k] >= quota) {
      const int pos = t * PER + k;
      s_bucket = desc ? (kBins - 1 - pos) : pos;
      s_before = run;
      s_count = loc[k];
    }
    run += loc[k];
  }
  __syncthreads();
  for (int b = t; b < kBins; b += blockDim.x) ws->hist[b] = 0ull;
  if (t == 0) {
    const uint64_t bucket = (uint64_t)s_bucket;
    const int64_t left = (int64_t)(quota - s_before);   // to be taken from this bucket
    const uint64_t count = s_count;
    const uint64_t newprefix = (st.width >= 64 ? 0ull : (st.prefix << st.width)) | bucket;
    SelState& o = ws->st;
    if (st.phase == 0) {
      if ((uint64_t)left == count) {                    // the whole bucket is kept: resolved, no tie
        o.phase = 2;
        o.t_ge = newprefix << st.shift;
      } else if (st.shift == 0) {                       // full key known, more equal keys than quota: tie
        o.t_ge = newprefix + 1;                         // keys are < 2^63: no overflow
        o.t_eq = newprefix;
        o.quota = left;
        o.phase = 1;
        o.prefix = 0;
        int top = st.idx_bits;                          // index digits, most significant first
        int w = top % kDigitBits ? top % kDigitBits : kDigitBits;
        if (top == 0) { o.phase = 2; o.icut = 0; }      // n == 1 cannot get here, kept for safety
        o.shift = top - w;
        o.width = w;
      } else {
        o.prefix = newprefix;
        o.quota = left;
        int w = st.shift < kDigitBits ? st.shift : kDigitBits;
        o.shift = st.shift - w;
        o.width = w;
      }
    } else {
      if ((uint64_t)left == count || st.shift == 0) {   // all indices of this bucket are kept
        o.phase = 2;
        o.icut = (int64_t)(((newprefix + 1) << st.shift) - 1);
      } else {
        o.prefix = newprefix;
        o.quota = left;
        int w = st.shift < kDigitBits ? st.shift : kDigitBits;
        o.shift = st.shift - w;
        o.width = w;
      }
    }
  }
}

template <bool BINF>
__device__ __forceinline__ double sel_out(double v, int64_t i, double x, double s, const SelState& st, double delta) {
  const uint64_t key = key_of(v);
  const bool keep = (key >= st.t_ge) || (key == st.t_eq && i <= st.icut);
  const double kept = keep ? v : 0.0;               // shiftedIndBallL0.jl:69  y[p[r+1:end]] .= 0
  const double t = kept - (x + s);                  // :70
  if constexpr (BINF) return jl_min(jl_max(t, -delta), delta);  // shiftedIndBallL0BInf.jl:91
  else return t;
}

template <bool BINF>
__global__ __launch_bounds__(256) void k_sel_final(double* y, const double* xk, const double* sj, int64_t n, int vec,
                                                    const SelWs* ws, double delta) {
  const SelState st = ws->st;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (vec) {
    const int64_t n2 = n >> 1;
    f64x2* y2 = reinterpret_cast<f64x2*>(y);
    const f64x2* x2 = reinterpret_cast<const f64x2*>(xk);
    const f64x2* s2 = reinterpret_cast<const f64x2*>(sj);
    for (int64_t i = tid; i < n2; i += stride) {
      f64x2 v = y2[i], b = x2[i], c = s2[i];
      f64x2 r;
      r.x = sel_out<BINF>(v.x, 2 * i, b.x, c.x, st, delta);
      r.y = sel_out<BINF>(v.y, 2 * i + 1, b.y, c.y, st, delta);
      y2[i] = r;
    }
    if ((n & 1) && tid == 0) y[n - 1] = sel_out<BINF>(y[n - 1], n - 1, xk[n - 1], sj[n - 1], st, delta);
  } else {
    for (int64_t i = tid; i < n; i += stride) y[i] = sel_out<BINF>(y[i], i, xk[i], sj[i], st, delta);
  }
}

template <bool BINF>
int run_select(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n, int64_t r,
               double delta) {
  int rc = spx_check_common(ctx, y, q, xk, sj, n);
  if (rc) return rc;
  if (n == 0) return SPX_OK;
  rc = spx_ws_reserve(ctx, sizeof(SelWs) + 256);
  if (rc) return rc;
  SPX_HIP(hipSetDevice(ctx->device));
  SelWs* ws = reinterpret_cast<SelWs*>(ctx->ws);
  const int vec = (spx_aligned16(y) && spx_aligned16(q) && spx_aligned16(xk) && spx_aligned16(sj)) ? 1 : 0;
  const int64_t work = vec ? (n + 1) / 2 : n;
  int64_t blocks = (work + 255) / 256;
  const int64_t cap = (int64_t)ctx->num_cu * 8;
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  dim3 grid((unsigned)blocks), block(256);
  hipLaunchKernelGGL(k_sel_init, dim3(1), dim3(256), 0, ctx->stream, ws, n, r);
  hipLaunchKernelGGL(k_sel_pass0, grid, block, 0, ctx->stream, y, q, xk, sj, n, vec, ws);
  hipLaunchKernelGGL(k_sel_scan, dim3(1), dim3(256), 0, ctx->stream, ws);
  // remaining key digits (5 after the first 12 of 64 bits) + index digits of a possible tie
  int idx_bits = 0;
  while (idx_bits < 63 && ((int64_t)1 << idx_bits) < n) ++idx_bits;
  const int passes = (64 - kDigitBits + kDigitBits - 1) / kDigitBits + (idx_bits + kDigitBits - 1) / kDigitBits;
  if (r > 0 && r < n) {
    for (int p = 0; p < passes; ++p) {
      hipLaunchKernelGGL(k_sel_hist, grid, block, 0, ctx->stream, (const double*)y, n, vec, ws);
      hipLaunchKernelGGL(k_sel_scan, dim3(1), dim3(256), 0, ctx->stream, ws);
    }
  }
  hipLaunchKernelGGL((k_sel_final<BINF>), grid, block, 0, ctx->stream, y, xk, sj, n, vec, (const SelWs*)ws, delta);
  SPX_LAUNCH_CHECK();
  return SPX_OK;
}

}  // namespace

SPX_EXPORT int spx_prox_indball_l0(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                                   int64_t n, int64_t r) {
  return run_select<false>(ctx, y, q, xk, sj, n, r, 0.0);
}

SPX_EXPORT int spx_prox_indball_l0_binf(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                                        int64_t n, int64_t r, double delta) {
  return run_select<true>(ctx, y, q, xk, sj, n, r, delta);
}
