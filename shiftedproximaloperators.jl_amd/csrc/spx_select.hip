// spx_select.hip -- ShiftedIndBallL0.prox! / ShiftedIndBallL0BInf.prox!: keep the r entries of
// v = (xk + sj) + q largest in magnitude, zero the others, subtract xk + sj (and clamp to +-Delta).
//
// The reference sorts: sortperm!(p, y, rev = true, by = abs) (src/shiftedIndBallL0.jl:68,
// src/shiftedIndBallL0BInf.jl:87) -- a STABLE permutation, i.e. descending |v|, ties in ascending index.
// Only the set of the first r entries matters, so this is an exact top-r SELECTION with the composite
// order (|v| descending, index ascending).  For finite doubles |v| is ordered like the 63-bit integer
// bits(v) & 0x7ff..f, so the selection is an MSD radix select on that integer key (12-bit digits of key - base, then --
// only if the threshold key T is shared by more elements than the remaining quota -- the index digits of {i : key_i == T}):
//     keep_i = key_i >= t_ge || (key_i == t_eq && i <= icut)
//     y[i] = (keep_i ? v_i : 0) - (xk[i] + sj[i])   [clamped for BInf]
// Three forms, by size (run_select): one workgroup (k_sel_small), one launch of a resident grid that synchronises inside
// itself (k_sel_coop), and the sample-predicted single streaming pass (k_s2_front / k_s2_main / candidate kernels) with
// k_sel_coop queued behind it as the exact fallback.  (Round 1's multi-launch pipelines, which read a verdict back, are gone.)
#include <type_traits>

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
  uint64_t prefix;    // phase 1: decided high index bits (phase 0 carries what is decided in `base`)
  int64_t quota;      // how many elements are still to be taken from the current bucket
  uint64_t t_ge;      // keep if key >= t_ge
  uint64_t t_eq;      // keep if key == t_eq && index <= icut
  int64_t icut;
  int idx_bits;       // number of significant bits of n - 1
  int pad2;
  uint64_t t_floor;   // candidate mode: no kept key is below this (0 in the full-vector mode)
  // phase 0: the part of key space still undecided is [base, base + 2^(shift + width)); keys above it are kept, keys
  // below it dropped, and the digit of a key inside it is (key - base) >> shift.  The interval follows the DATA (first
  // candidate digit: base = low end of the band, 2^(shift + width) just above the band's span), not the bit pattern:
  // a band that straddles an exponent boundary still spreads over >= 2048 bins.  clamp != 0: the interval is open above
  // and the last bin takes everything from (nb - 1) << shift on (first candidate digit when the band has no upper end).
  uint64_t base;
  int clamp;
  int pad3;
};

// phase-0 position of a key (see SelState::base)
struct KeyPos {
  bool in, above;
  unsigned int digit;
};
__device__ __forceinline__ KeyPos key_pos(uint64_t key, uint64_t base, int shift, int width, int clamp) {
  const int hs = shift + width;
  const uint64_t rel = key - base;
  const bool ge = key >= base;
  const unsigned int last = (1u << width) - 1u;
  const uint64_t d = rel >> shift;  // shift <= 52
  KeyPos r;
  if (clamp) {
    r.in = ge;
    r.above = false;
    r.digit = d > last ? last : (unsigned int)d;
  } else {
    const bool inside = hs >= 64 || (rel >> hs) == 0;
    r.in = ge && inside;
    r.above = ge && !inside;
    r.digit = (unsigned int)d & last;
  }
  return r;
}

// A state every lane read from shared memory, moved to scalar registers (the values are the same in every lane, but a load
// leaves them in 20-odd VECTOR registers for as long as the state is in use: k_sel_lds has none to spare)
__device__ __forceinline__ int sel_uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ uint64_t sel_uni(uint64_t v) {
  return ((uint64_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)(v >> 32)) << 32) |
         (uint64_t)(unsigned int)__builtin_amdgcn_readfirstlane((int)(v & 0xffffffffu));
}
__device__ __forceinline__ int64_t sel_uni(int64_t v) { return (int64_t)sel_uni((uint64_t)v); }
__device__ __forceinline__ SelState sel_uniform(const SelState& a) {
  SelState o;
  o.phase = sel_uni(a.phase); o.shift = sel_uni(a.shift); o.width = sel_uni(a.width); o.pad = sel_uni(a.pad);
  o.prefix = sel_uni(a.prefix); o.quota = sel_uni(a.quota); o.t_ge = sel_uni(a.t_ge); o.t_eq = sel_uni(a.t_eq);
  o.icut = sel_uni(a.icut); o.idx_bits = sel_uni(a.idx_bits); o.pad2 = sel_uni(a.pad2); o.t_floor = sel_uni(a.t_floor);
  o.base = sel_uni(a.base); o.clamp = sel_uni(a.clamp); o.pad3 = sel_uni(a.pad3);
  return o;
}

// state of the sample-predicted path (see "fast path" below)
struct FastState {
  uint64_t t_hi, t_lo;            // candidate band in key space: t_lo <= key <= t_hi; "above": key > t_hi
  unsigned long long cnt_above;   // number of elements with key > t_hi
  unsigned long long cand_count;  // number of candidates appended
  int ok;                         // set by k_s2_verify: the r-th largest is provably inside the band
  int key_passes;                 // key-digit passes the candidate selection can need (host reads ok + this)
  int overflow;                   // a workgroup or the candidate buffer overflowed
  unsigned int list_count;        // entries appended to the short list by k_s2_compact
  unsigned long long smax;        // largest FINITE sample key (k_s2_sample; zeroed by k_sel_init)
  int crowded;                    // k_s2_front: a selected sample bucket is still crowded after the last digit, i.e. the band
                                  // holds tied keys -- k_s2_main then counts candidate digits in wave-uniform runs
  unsigned int ovf_count;         // candidates a wavefront could not fit into its own region: appended to the shared overflow
                                  // list behind the regions (sorted / clustered data: the band is contiguous in the vector)
  // Tie mode (round 3).  A sample bucket that is still crowded after 36 key bits is a key shared by >= 0.1 % of the vector
  // (lattice data, a constant, a sparse vector's zeros): the front kernel then resolves BOTH sample ranks to the full 64 bits,
  // t_hi / t_lo are exact keys, and the elements EQUAL to them (classes "hi" / "lo") are counted per wavefront in index
  // order (ClassCount) instead of being recorded as candidates -- a class can be the whole vector.  Only keys strictly
  // between the ends remain candidates.  If the cut falls inside a class the index of its quota-th member comes from a
  // prefix sum over the per-wave counts (k_s2_tail).  y: speculative stores in the main pass (spec_hi / spec_lo below) and a
  // rewrite of the index range where the speculation was wrong; a full second streaming pass (56 B/element in all) when y
  // aliases an input -- against ~12 passes of the exact radix select (2.6-5.8 ms at n = 1e8 in round 2).
  int tie;
  int has_hi, has_lo;             // which ends are classes (no upper end for tiny r, no lower end for r ~ n; one class if t_hi == t_lo)
  unsigned long long cls_hi, cls_lo;  // class totals (k_s2_scan_verify)
  int todo;                       // what k_s2_tail still has to do: kTodo* bits (k_s2_scan_verify, k_s2_finish)
  int tie_class;                  // kTodoTieScan: the class that holds the cut (0 = hi, 1 = lo); quota in SelState
  // Speculative stores of the class members in the single-pass form (y disjoint from the inputs): what the SAMPLE says about
  // the cut -- 0: the class is dropped, 1: kept, 2: the cut lies inside it, its members are kept up to index spec_cut
  // (= the sample's share of the class above the cut, times n: exact for ties spread evenly over the vector).  Once the true
  // cut is known k_s2_tail rewrites the index range in which the guess was wrong (a per cent of the vector on random data,
  // everything at worst): the call moves 32 B/element plus that range instead of 56.
  int spec_hi, spec_lo;
  long long spec_cut;
  // Speculation on the CANDIDATES (round 3): the main pass stores a candidate as kept when its key is >= t_mid, the middle of
  // the band -- where the sample puts the cut, +-1 sigma of a +-6 sigma band -- and as dropped below it; the candidate kernels
  // rewrite those the cut proves wrong (a sixth of what "every candidate dropped" left to rewrite: 3e5 scattered 8-byte stores
  // at n = 1e8, r = n/2, 29 us of k_s2_compact).  Any value is correct; ~0 = no candidate is stored as kept.
  uint64_t t_mid;
};
// the value of a dropped entry i (shiftedIndBallL0.jl:69-70, shiftedIndBallL0BInf.jl:91), as k_s2_main forms it
template <bool BINF>
__device__ __forceinline__ double sel_dropped(const double* xk, const double* sj, int64_t i, double delta) {
  const double d = 0.0 - (xk[i] + sj[i]);
  if constexpr (BINF) return jl_min(jl_max(d, -delta), delta);
  else return d;
}
constexpr int kTodoCandSelect = 1;  // the bucket of the first candidate digit overflows the short list: radix select over the candidates
constexpr int kTodoTieScan = 2;     // the cut lies inside a class: find the index of its quota-th member
constexpr int kTodoFinal = 4;       // tie mode: y is (re)written from q, xk, sj and the final thresholds -- where the speculative
                                    // stores of the class members were wrong (single-pass form), everywhere (y aliases an input)

// totals of the main pass: fire-and-forget atomics of its wavefronts, spread over kShards counters (wave w -> shard
// w % kShards) so that no single address serialises them; k_s2_scan_verify adds the shards up.  Round 2: 2048 shards
// instead of 64 -- atomics on ONE address retire ~12 ns apart, so the 130 208 waves of an n = 1e8 pass queued ~2000 deep
// on each of 64 addresses (~25 us of serialised traffic per address, the "13 us" the pair of atomics was measured to
// cost the pass); at 64 per address it is gone.
constexpr int kShards = 2048;
constexpr int kShardStride = 1;

struct SelWs {
  SelState st;
  FastState fs;
  unsigned long long hist[kBins];
  unsigned long long shard_above[kShards * kShardStride];
  unsigned long long shard_cand[kShards * kShardStride];
  unsigned long long shard_hi[kShards * kShardStride];   // tie mode: class totals
  unsigned long long shard_lo[kShards * kShardStride];
};

// Candidates of the main pass: every WAVEFRONT of its grid owns a fixed region of kWaveSlots entries and one count
// word {candidates, elements above the band}: no atomics, no barrier, nothing a wave has to wait for before it exits,
// and a candidate order that is the same run to run.
constexpr int kWaveSlots = 64;
struct WaveCount {
  unsigned int cand, above;
};
// Tie mode: members of the two classes per wavefront of the main pass, in index order: slot 0 = the caller's element 0 of an
// 8-byte-misaligned view, slot 1 + w = wave w (elements ioff + 768 w .. ioff + 768 w + 767), last slot = the odd last element.
struct ClassCount {
  unsigned int hi, lo;
};
// One candidate: key, index and (single-pass form) the value y[index] gets if the entry makes the cut.  An array of
// structs: the handful of entries a region holds then shares one or two cache lines, instead of one line in each of
// three arrays, when the candidate kernels walk the regions.
struct Cand {
  uint64_t key;
  int64_t idx;
  double val;
};

constexpr int kMainUnroll = 6;        // KiB per wave and vector in the main pass (as k_sep_lds)
constexpr int kMainTilePairs = 256 * kMainUnroll;  // 16-byte pairs per workgroup of the main pass
constexpr int kShortList = 4096;      // candidates left after the first digit that k_s2_finish resolves in LDS

// bits(|v|): ordered like |v| for finite values and Inf.  Every NaN maps to ONE key above Inf: the reference's sort
// compares with isless, where NaN is the largest value and all NaNs are equal (ties -> ascending index).
constexpr uint64_t kInfKey = 0x7ff0000000000000ull;
constexpr uint64_t kNanKey = 0x7ff8000000000000ull;
__device__ __forceinline__ uint64_t key_of(double v) {
  const uint64_t k = (uint64_t)__double_as_longlong(v) & kAbsMask;
  return k > kInfKey ? kNanKey : k;
}

// Histogram add of a wavefront's digits with the heavy hitters peeled off first: the digit of the first pending lane is added
// ONCE with its population count (twice over, if the first peel took a good part of the wave), the rest go lane by lane.
// Tie-heavy data (lattices, many equal entries) sends whole wavefronts to one bin, and atomics on one address are serialised
// -- in LDS ~1 per clock, in global memory ~12 ns apiece (n = 1e8 integers: 588 ms per top-r call before, ~3 ms after).  Called
// by whichever lanes are active; `in` = this lane contributes.  Ctr = unsigned int (LDS) or unsigned long long (global).
template <class Ctr>
__device__ __forceinline__ void hist_add_agg(Ctr* h, unsigned int digit, bool in) {
  unsigned long long todo = __ballot(in);
  if (todo == 0ull) return;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int rep = 0; rep < 2; ++rep) {
    const int leader = __ffsll((long long)todo) - 1;
    const unsigned int d0 = __shfl(digit, leader, 64);
    const unsigned long long same = __ballot(in && digit == d0) & todo;
    if (lane == leader) atomicAdd(&h[d0], (Ctr)__popcll(same));
    todo &= ~same;
    if (todo == 0ull || __popcll(same) < 8) break;  // (wave-uniform) no second heavy hitter in sight
  }
  if ((todo >> lane) & 1ull) atomicAdd(&h[digit], (Ctr)1);
}

// exclusive prefix sum over 256 consecutive lanes (4 wavefronts: tt = 0..255); wtot = 4 shared slots of that group
__device__ __forceinline__ unsigned long long scan256_exclusive(unsigned long long v, int tt, unsigned long long* wtot) {
  const int lane = tt & 63, w = tt >> 6;
  unsigned long long inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long up = __shfl_up(inc, off, 64);
    if (lane >= off) inc += up;
  }
  if (lane == 63) wtot[w] = inc;
  __syncthreads();
  unsigned long long base = 0;
  for (int k = 0; k < w; ++k) base += wtot[k];
  return base + inc - v;
}

// Locates the bucket that holds the quota-th element (from the top for keys, from the bottom for indices) in a
// 4096-bin histogram and advances the selection state.  Called by all lanes of a workgroup (>= 256 lanes);
// lanes 0..255 do the work.  `scratch` = 8 unsigned long long of shared memory.  The new state is returned in
// *out by lane 0 (out may be global or shared memory); no barrier after that write.
// Folded first digit of the in-launch selection (k_sel_coop, SelState::pad == 1): the top 12 key bits are sign + exponent,
// i.e. ONE bin per binade -- the bucket of the r-th largest then still holds a few per cent of the vector and two more
// digits are needed.  Folded: 62 binades around 1.0 (2^-31 .. 2^30) x 64 mantissa steps each, everything below in bin 0,
// everything above in bin 4095 -- monotone in the key, bucket = n / 64 of a binade, one more digit (bits 45..34) resolves it
// (measured at n = 1e6, r = n/100: 3 digit passes -> 2, 48 -> 3x us).  A threshold in one of the two catch-all bins restarts
// with the plain top digit (one pass lost, on data 2^31 away from 1.0).
constexpr int kFoldLoExp = 991;    // biased exponent of the last binade that falls into bin 0
constexpr int kFoldHiExp = 1054;   // ... of the first binade that falls into bin 4095
__device__ __forceinline__ unsigned int fold_digit(uint64_t key) {
  const int e = (int)(key >> 52);
  if (e <= kFoldLoExp) return 0u;
  if (e >= kFoldHiExp) return (unsigned int)kBins - 1u;
  return ((unsigned int)(e - kFoldLoExp) << 6) | (unsigned int)((key >> 46) & 63u);
}

// The same fold for the keys of Float32 values (bits(|v|) << 32; SelState::pad == 2, round 3): 62 binades around 1.0
// (2^-32 .. 2^30) x 64 mantissa steps; the next digit resolves bits 48..37, i.e. 18 of the 23 mantissa bits after two passes.
constexpr int kFoldLoExp32 = 95;    // biased exponent of the last binade that falls into bin 0
constexpr int kFoldHiExp32 = 158;   // ... of the first binade that falls into bin 4095
__device__ __forceinline__ unsigned int fold_digit_f32(uint64_t key) {
  const int e = (int)(key >> 55);
  if (e <= kFoldLoExp32) return 0u;
  if (e >= kFoldHiExp32) return (unsigned int)kBins - 1u;
  return ((unsigned int)(e - kFoldLoExp32) << 6) | (unsigned int)((key >> 49) & 63u);
}

// The state after a scan located the bucket of the quota-th element: `before` elements precede the bucket in scan order,
// `count` are in it.
__device__ __forceinline__ SelState sel_advance(const SelState st, uint64_t bucket, uint64_t before, uint64_t count) {
  const int nb = 1 << st.width;
  const unsigned long long quota = (unsigned long long)st.quota;
  const int64_t left = (int64_t)(quota - before);  // to be taken from this bucket
  const uint64_t newprefix = (st.width >= 64 ? 0ull : (st.prefix << st.width)) | bucket;  // (phase 1)
  SelState o = st;
  if (st.phase == 0 && (st.pad == 1 || st.pad == 2)) {  // folded first digit (fold_digit / fold_digit_f32)
    o.pad = 0;
    const bool f32 = st.pad == 2;
    const int ebit = f32 ? 55 : 52, mbit = f32 ? 49 : 46;  // lowest bit of the exponent field / of the six mantissa bits in the key
    const uint64_t hiexp = f32 ? (uint64_t)kFoldHiExp32 : (uint64_t)kFoldHiExp, loexp = f32 ? (uint64_t)kFoldLoExp32 : (uint64_t)kFoldLoExp;
    const bool catch_all = bucket < 64 || bucket >= (uint64_t)(kBins - 64);
    const uint64_t lowkey = bucket < 64 ? 0ull
                          : bucket >= (uint64_t)(kBins - 64) ? (hiexp << ebit)
                          : (((bucket >> 6) + loexp) << ebit) | ((bucket & 63ull) << mbit);
    if ((uint64_t)left == count) {                      // the whole bucket is kept
      o.phase = 2;
      o.t_ge = lowkey > st.t_floor ? lowkey : st.t_floor;
    } else if (!catch_all) {                            // next digit: bits 45..34 (Float32 keys: 48..37) of this bucket
      o.base = lowkey;
      o.clamp = 0;
      o.quota = left;
      o.shift = mbit - kDigitBits;
      o.width = kDigitBits;
    }                                                   // else: start over with the plain top digit (o = st, pad = 0)
    return o;
  }
  if (st.phase == 0) {
    const uint64_t lowkey = st.base + (bucket << st.shift);  // low end of the bucket (no overflow: see k_s2_pick)
    const bool open_top = st.clamp && bucket == (uint64_t)(nb - 1);  // the bucket has no upper end
    if ((uint64_t)left == count) {                    // the whole bucket is kept: resolved, no tie
      o.phase = 2;
      o.t_ge = lowkey > st.t_floor ? lowkey : st.t_floor;
    } else if (st.shift == 0 && !open_top) {          // full key known, more equal keys than quota: tie
      o.t_ge = lowkey + 1;                            // keys are < 2^63: no overflow
      o.t_eq = lowkey;
      o.quota = left;
      o.phase = 1;
      o.prefix = 0;
      int top = st.idx_bits;                          // index digits, most significant first
      int w = top % kDigitBits ? top % kDigitBits : kDigitBits;
      if (top == 0) { o.phase = 2; o.icut = 0; }      // n == 1 cannot get here, kept for safety
      o.shift = top - w;
      o.width = w;
    } else {                                          // next digit inside this bucket
      o.base = lowkey;
      o.clamp = 0;
      o.quota = left;
      const int rem = open_top ? 64 : st.shift;       // undecided bits of (key - lowkey)
      const int w = rem < kDigitBits ? rem : kDigitBits;
      o.shift = rem - w;
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
  return o;
}

template <class Hist>
__device__ __forceinline__ void sel_scan_step(const Hist& hist, const SelState st, SelState* out,
                                              unsigned long long* scratch) {
  const int t = threadIdx.x;
  const int tt = t & 255;
  const bool worker = t < 256;
  const int nb = 1 << st.width;
  const bool desc = (st.phase == 0);
  constexpr int PER = kBins / 256;  // lane tt owns PER consecutive bins of the scan order
  unsigned long long loc[PER];
  unsigned long long sum = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int pos = tt * PER + k;
    const int bin = desc ? (kBins - 1 - pos) : pos;  // descending keys / ascending indices
    loc[k] = (worker && bin < nb) ? (unsigned long long)hist[bin] : 0ull;
    sum += loc[k];
  }
  unsigned long long run = scan256_exclusive(sum, tt, scratch + (t >> 8 ? 4 : 0));
  const unsigned long long quota = (unsigned long long)st.quota;
  unsigned long long* found = scratch + 4;  // [4] bucket, [5] before, [6] count  (lanes >= 256 used [4..7] only as wtot)
  __syncthreads();
  if (worker) {
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      if (run < quota && run + loc[k] >= quota) {
        const int pos = tt * PER + k;
        found[0] = (unsigned long long)(desc ? (kBins - 1 - pos) : pos);
        found[1] = run;
        found[2] = loc[k];
      }
      run += loc[k];
    }
  }
  __syncthreads();
  if (t == 0) *out = sel_advance(st, found[0], found[1], found[2]);
}

// Float32 vectors (round 3; the reference's methods are generic in R, src/shiftedIndBallL0.jl:54-59): bits(|v|) of a Float32 moved
// into the top half of the 64-bit key -- monotone, so every digit / scan / tie-break routine below serves both element types
// (the low 32 key bits are zero: the digits there resolve at once).  All NaNs are one key above Inf, as for Float64.
__device__ __forceinline__ uint64_t key_of(float v) {
  const uint32_t k = (uint32_t)__float_as_int(v) & 0x7fffffffu;
  return (uint64_t)(k > 0x7f800000u ? 0x7fc00000u : k) << 32;
}
// xs = xk[i] + sj[i], the sum v was formed from
template <bool BINF, class T>
__device__ __forceinline__ T sel_out_xs(T v, int64_t i, T xs, const SelState& st, T delta) {
  const uint64_t key = key_of(v);
  const bool keep = (key >= st.t_ge) || (key == st.t_eq && i <= st.icut);
  const T kept = keep ? v : (T)0;                   // shiftedIndBallL0.jl:69  y[p[r+1:end]] .= 0
  const T t = kept - xs;                            // :70
  if constexpr (BINF) return jl_min(jl_max(t, -delta), delta);  // shiftedIndBallL0BInf.jl:91
  else return t;
}
template <bool BINF, class T>
__device__ __forceinline__ T sel_out(T v, int64_t i, T x, T s, const SelState& st, T delta) {
  return sel_out_xs<BINF, T>(v, i, x + s, st, delta);
}

// =============================================================================================
// Small n (<= kSmallN): the whole selection in ONE workgroup and one launch.  The multi-launch path above costs 65-70 us
// at any small size (18-20 dependent launches); here a single CU streams the vectors (L2-resident after the first
// touch): v -> y and the key range in pass 0, digits of key - min over the data's own span (SelState::base: no hot
// exponent bins for the LDS atomics), index digits for ties, final pass.  Lane t owns elements t, t + 1024, ... in
// every pass, so y needs no fence between passes.
// =============================================================================================
constexpr int64_t kSmallNCoop = 1 << 13;  // k_sel_coop serves the sizes above (see run_select)
template <bool BINF, class T = double>
__global__ __launch_bounds__(1024) void k_sel_small(T* y, const T* q, const T* xk, const T* sj, int64_t n,
                                                     int64_t r, T delta) {
  __shared__ unsigned int h[kBins];
  __shared__ unsigned long long scratch[8];
  __shared__ unsigned long long kmm[2];
  __shared__ SelState sst;
  const int t = threadIdx.x;
  if (t == 0) { kmm[0] = ~0ull; kmm[1] = 0ull; }
  __syncthreads();
  uint64_t kmin = ~0ull, kmax = 0ull;
  for (int64_t i = t; i < n; i += 1024) {
    const T v = (xk[i] + sj[i]) + q[i];  // shiftedIndBallL0.jl:66
    y[i] = v;
    const uint64_t k = key_of(v);
    kmin = k < kmin ? k : kmin;
    kmax = k > kmax ? k : kmax;
  }
  for (int off = 32; off >= 1; off >>= 1) {
    const uint64_t a = __shfl_xor((unsigned long long)kmin, off, 64), b = __shfl_xor((unsigned long long)kmax, off, 64);
    kmin = a < kmin ? a : kmin;
    kmax = b > kmax ? b : kmax;
  }
  if ((t & 63) == 0) { atomicMin(&kmm[0], (unsigned long long)kmin); atomicMax(&kmm[1], (unsigned long long)kmax); }
  __syncthreads();
  if (t == 0) {
    SelState s;
    int bits = 0;
    while (bits < 63 && ((int64_t)1 << bits) < n) ++bits;
    s.idx_bits = bits;
    s.prefix = 0;
    s.icut = -1;
    s.t_eq = ~0ull;
    s.t_ge = ~0ull;
    s.pad = s.pad2 = s.pad3 = 0;
    s.t_floor = 0;
    s.clamp = 0;
    s.quota = r;
    s.base = kmm[0];
    s.shift = 0;
    s.width = 0;
    const uint64_t span = kmm[1] - kmm[0];
    if (r <= 0) {  // nothing kept
      s.phase = 2; s.quota = 0;
    } else if (r >= n) {  // everything kept
      s.phase = 2; s.t_ge = 0ull; s.quota = 0;
    } else if (span == 0) {  // one key value: straight to the index tie-break
      s.phase = 1;
      s.t_eq = kmm[0];
      s.t_ge = kmm[0] + 1;
      const int w = bits % kDigitBits ? bits % kDigitBits : kDigitBits;
      s.shift = bits - w;
      s.width = w;
      if (bits == 0) { s.phase = 2; s.icut = 0; }
    } else {
      const int sb = 64 - __clzll((long long)span);  // span < 2^sb: every key is inside [base, base + 2^sb)
      s.phase = 0;
      s.width = sb < kDigitBits ? sb : kDigitBits;
      s.shift = sb - s.width;
    }
    sst = s;
  }
  for (int guard = 0; guard < 16; ++guard) {  // <= 6 key digits + <= 2 index digits
    __syncthreads();
    const SelState st = sst;
    if (st.phase == 2) break;
    for (int b = t; b < kBins; b += 1024) h[b] = 0u;
    __syncthreads();
    const int hs = st.shift + st.width;
    const uint64_t dmask = ((uint64_t)1 << st.width) - 1;
    for (int64_t i = t; i < n; i += 1024) {
      const uint64_t key = key_of(y[i]);
      if (st.phase == 0) {
        const KeyPos kp = key_pos(key, st.base, st.shift, st.width, st.clamp);
        if (kp.in) atomicAdd(&h[kp.digit], 1u);
      } else if (key == st.t_eq && (((uint64_t)i) >> hs) == st.prefix) {
        atomicAdd(&h[(((uint64_t)i) >> st.shift) & dmask], 1u);
      }
    }
    __syncthreads();
    sel_scan_step(h, st, &sst, scratch);
  }
  __syncthreads();
  const SelState fin = sst;
  for (int64_t i = t; i < n; i += 1024) y[i] = sel_out<BINF>(y[i], i, xk[i], sj[i], fin, delta);
}

// =============================================================================================
// Fast path (large n): sample-predicted band.
//   1. k_s2_sample   65536 samples of |v| (256 chunks of 256 consecutive elements: 1.5 MB of reads)
//   2. k_s2_pick     one workgroup: exact order statistics of the sample at ranks p*M -/+ (6 sigma + 16),
//                    p = r/n, sigma = sqrt(M p (1-p))  ->  key band [t_lo, t_hi] that contains the r-th
//                    largest |v| of the whole vector except with probability ~1e-9 per side
//   3. k_s2_main     ONE streaming pass over q, xk, sj: counts the elements above the band and writes the (key, index
//                    [, kept value]) of the ~0.5 % inside it into the candidate region of the WAVEFRONT that saw them
//                    (fixed regions + one count word per wave: no atomics, no barrier, deterministic order).
//                    If y overlaps none of the inputs the pass also stores y speculatively (above the band: kept,
//                    otherwise dropped) -- 32 B/element, the algorithmic minimum; else it writes nothing (24 B/element).
//                    The pass also histograms the first digit of the candidates (one global atomic per candidate) and
//                    adds its totals into sharded counters (fire-and-forget).
//   4. k_s2_scan_verify   the verdict  cnt_above < r <= cnt_above + candidates  and the first scan step of the
//                    selection among the candidates; k_s2_compact + k_s2_finish resolve the
//                    rest on a short list in one workgroup (regions are walked transposed: 64 regions per wavefront)
//   5a. (y disjoint) k_s2_compact / k_s2_finish also store the kept value of the candidates that make the cut
//                    (~0.25 % of y, scattered): no further pass
//   5b. k_sel_final_q (y aliases an input) y[i] from q, xk, sj and the thresholds (24 B read + 8 B written per element)
// Total 32 B/element (+ ~1 % for the candidates) in the disjoint case, 56 B/element when y aliases an input, instead
// of >= 80.  The prediction is only a performance device: if the verification fails (or a region overflows) the
// full-vector radix select above runs instead -- it recomputes everything from q, xk, sj, which the speculative stores
// cannot have touched -- so the result is exact for any input.  The host reads the 4-byte verdict back once, after
// the last kernel has been queued (steps 5a/5b return at once if the verdict is negative).
// In the aliased case y is not touched before step 5b and step 5b reads q[i] before writing y[i].
// =============================================================================================
// Main pass: one tile per workgroup, waves fully independent (no barrier, no atomics).  Counts the elements above the
// band and appends the band's elements to the wave's own candidate region.
// WRITE (y overlaps none of the inputs): the pass also stores y, speculatively -- entries above the band as kept,
// entries below it and inside it as dropped; the kept value of a band entry travels with the candidate and
// k_s2_compact / k_s2_finish store it once the cut is known.  The call then moves the algorithmic 32 B/element plus the ~0.5 % of
// candidates, instead of 56 B/element with the separate final pass (k_sel_final_q, used when y aliases an input).
// The wave totals also go into the sharded counters of SelWs (k_s2_scan_verify adds the shards up).
template <bool BINF, bool WRITE>
__global__ __launch_bounds__(256) void k_s2_main(double* y_, const double* q_, const double* xk_, const double* sj_,
                                                  int64_t n, SelWs* ws, Cand* cand, WaveCount* counts, double delta,
                                                  int ioff, int64_t ovf_base, unsigned int ovf_cap, ClassCount* cls,
                                                  int64_t nwaves_total) {
  // ioff = 1: the caller's vectors start 8 bytes off a 16-byte boundary (all four alike); the pointers passed here
  // are the aligned rest (caller's element 1 on), n counts that rest, and the caller's element 0 sits at [-1].
  // Candidate indices are the caller's.
  const uint64_t t_hi = ws->fs.t_hi, t_lo = ws->fs.t_lo;
  // first digit of the selection among the candidates (set up by k_s2_pick): histogrammed right here, one
  // fire-and-forget global atomic per candidate (~0.5 % of the elements, spread over the band's bins)
  const int d_phase = ws->st.phase, d_shift = ws->st.shift, d_width = ws->st.width, d_clamp = ws->st.clamp;
  const int d_hs = d_shift + d_width;
  const uint64_t d_mask = ((uint64_t)1 << d_width) - 1, d_prefix = ws->st.prefix, d_teq = ws->st.t_eq;
  const f64x2* q = reinterpret_cast<const f64x2*>(q_);
  const f64x2* xk = reinterpret_cast<const f64x2*>(xk_);
  const f64x2* sj = reinterpret_cast<const f64x2*>(sj_);
  f64x2* y2 = reinterpret_cast<f64x2*>(y_);
  const int64_t n2 = n >> 1;
  constexpr int UNROLL = kMainUnroll;
  __shared__ __attribute__((aligned(16))) char dma[4 * 3 * UNROLL * 1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // A band the front kernel already knows to be hopeless (its own samples put more of the vector into it than the candidate
  // regions hold: tie-heavy data -- a band of ONE key with a third of the vector in it) is not walked at all: the verdict is
  // negative whatever this pass does and the exact select recomputes everything.  (580 ms of serialised per-candidate
  // atomics on n = 1e8 integers before.)  Written by the previous launch: a plain, cached load, the same in every wave.
  const int hopeless = ws->fs.overflow;   // (tested once the staging loads have landed: a test up here would hold every
  const bool crowded = ws->fs.crowded != 0;  // wave's loads back behind its scalar loads -- measured: 5 % of the pass)
  // tie mode (FastState::tie): t_hi / t_lo are exact keys; their elements are counted (eq_hi / eq_lo), not recorded, and
  // stored on the sample's guess of the cut (FastState::spec_*); k_s2_tail rewrites the range where the guess was wrong.
  // The processing part below exists TWICE in this kernel, with the tie code compiled in and out (body(tag)), behind one
  // wave-uniform branch taken after the staging loads have landed: as a run-time `if (tie)` inside the per-element code
  // it cost the generic path 12-25 us of the pass (measured A/B: r = n/100 0.600 -> 0.588 ms, r = n/2 0.661 -> 0.636).
  const bool tie_mode = ws->fs.tie != 0;
  const bool has_hi = ws->fs.has_hi != 0, has_lo = ws->fs.has_lo != 0;
  const int spec_hi = ws->fs.spec_hi, spec_lo = ws->fs.spec_lo;
  const int64_t spec_cut = ws->fs.spec_cut;
  const uint64_t t_mid = ws->fs.t_mid;
  const int64_t gwave = (int64_t)blockIdx.x * 4 + wave;
  const int64_t rbase = gwave * kWaveSlots;  // this wave's candidate region
  const unsigned long long lt_mask = (1ull << lane) - 1;
  // each wave moves kMainUnroll KiB per vector through LDS (global_load_lds nt), waits once and reads back its own
  // 16-byte slots -- the staging of the separable skeleton (spx_separable.hip)
  char* wl = dma + wave * (3 * UNROLL * 1024);
  typedef __attribute__((address_space(3))) void lds_void;
  const int64_t wbase = gwave * (64 * UNROLL) + lane;
#pragma unroll
  for (int k = 0; k < UNROLL; ++k) {
    int64_t i = wbase + k * 64;
    if (i >= n2) i = n2 - 1;
    __builtin_amdgcn_global_load_lds((const void*)(q + i), (lds_void*)(wl + (0 * UNROLL + k) * 1024), 16, 0, 2);
    __builtin_amdgcn_global_load_lds((const void*)(xk + i), (lds_void*)(wl + (1 * UNROLL + k) * 1024), 16, 0, 2);
    __builtin_amdgcn_global_load_lds((const void*)(sj + i), (lds_void*)(wl + (2 * UNROLL + k) * 1024), 16, 0, 2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (hopeless) return;
  auto body = [&](auto tie_tag) {
  constexpr bool tie = decltype(tie_tag)::value;
  unsigned int eq_hi = 0, eq_lo = 0;      // per lane
  unsigned int seq_hi[2] = {0u, 0u}, seq_lo[2] = {0u, 0u};  // the stragglers wave 0 takes along: [0] head element, [1] odd last element
  unsigned int above = 0;   // per lane
  unsigned int ncand = 0;   // wave-uniform
  // wave-uniform: the runs of candidate digits being counted (ties, see visit)
  unsigned int run_digit[4] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu}, run_count[4] = {0u, 0u, 0u, 0u};
  int run_next = 0;
  bool full = false;        // (per lane) the overflow list had no room for this lane's candidate
  // Called by all 64 lanes together (the ballot needs them); returns the value stored speculatively (WRITE).
  auto visit = [&](bool valid, double v, int64_t i, double x, double s, int straggler = -1) -> double {
    const uint64_t key = key_of(v);
    double kept = 0.0, dropped = 0.0;
    if constexpr (WRITE) {
      const double xs = x + s;
      kept = v - xs;            // shiftedIndBallL0.jl:70 on a kept entry
      dropped = 0.0 - xs;       // :69-70 on a zeroed entry
      if constexpr (BINF) {     // shiftedIndBallL0BInf.jl:91
        kept = jl_min(jl_max(kept, -delta), delta);
        dropped = jl_min(jl_max(dropped, -delta), delta);
      }
    }
    const bool is_above = valid && key > t_hi;
    bool in_band = valid && !is_above && key >= t_lo;
    above += is_above ? 1u : 0u;
    bool spec_kept = is_above;  // what the speculative store assumes about this element
    if constexpr (tie) {  // members of the classes are counted in index order, never recorded
      const bool c_hi = in_band && has_hi && key == t_hi;
      const bool c_lo = in_band && has_lo && key == t_lo;
      if (straggler < 0) { eq_hi += c_hi ? 1u : 0u; eq_lo += c_lo ? 1u : 0u; }
      else { seq_hi[straggler & 1] += c_hi ? 1u : 0u; seq_lo[straggler & 1] += c_lo ? 1u : 0u; }
      in_band = in_band && !c_hi && !c_lo;
      if (c_hi) spec_kept = spec_hi == 1 || (spec_hi == 2 && i <= spec_cut);
      if (c_lo) spec_kept = spec_lo == 1 || (spec_lo == 2 && i <= spec_cut);
    }
    if (in_band) spec_kept = key >= t_mid;  // (FastState::t_mid)
    const unsigned long long m = __ballot(in_band);
    if (m) {
      const unsigned int pos = ncand + (unsigned int)__popcll(m & lt_mask);
      if (in_band && pos < (unsigned)kWaveSlots) {
        cand[rbase + pos].key = key;
        cand[rbase + pos].idx = i;
        if constexpr (WRITE) cand[rbase + pos].val = kept;
      }
      // more candidates than the wave's own region holds (sorted or clustered data: the band is contiguous in the vector):
      // the rest goes to the shared overflow list, one atomic per visit; only a FULL list makes the verdict negative
      const bool spill = in_band && pos >= (unsigned)kWaveSlots;
      if (ncand + (unsigned int)__popcll(m) > (unsigned)kWaveSlots) {  // (wave-uniform: only then can a lane spill)
        const unsigned long long sm = __ballot(spill);
        const int sl = __ffsll((long long)sm) - 1;
        unsigned int sbase = 0;
        if (lane == sl) sbase = atomicAdd(&ws->fs.ovf_count, (unsigned int)__popcll(sm));
        sbase = __shfl(sbase, sl, 64);
        const unsigned int slot = sbase + (unsigned int)__popcll(sm & lt_mask);
        if (spill) {
          if (slot < ovf_cap) {
            cand[ovf_base + slot].key = key;
            cand[ovf_base + slot].idx = i;
            if constexpr (WRITE) cand[ovf_base + slot].val = kept;
          } else {
            full = true;
          }
        }
      }
      {
        // Ties (FastState::crowded, decided by the front kernel from its sample): the candidates of a wave share ONE digit
        // -- the same key, or neighbouring indices of a one-key band -- and 1.7e6 of them at n = 1e8 were 1.7e6 atomics on
        // one address, 10.7 ms.  The digit of the first candidate lane is then counted in a wave-uniform RUN, added when the
        // run's digit changes or the wave ends; other digits of the same visit go lane by lane.
        bool cnt = in_band;
        unsigned int dg = 0;
        if (in_band) {
          if (d_phase == 0) {  // key digits; every band key is inside the first digit's interval (base = t_lo)
            dg = key_pos(key, t_lo, d_shift, d_width, d_clamp).digit;
          } else {             // index digits of a one-key band
            cnt = key == d_teq && (((uint64_t)i) >> d_hs) == d_prefix;
            dg = (unsigned int)((((uint64_t)i) >> d_shift) & d_mask);
          }
        }
        if (tie || crowded) {
          // tie mode: the keys between the band's ends may be shared by per cents of the vector too (a lattice: eight keys
          // holding 2.4 % of n = 1e8 were 1e6 run flushes onto eight addresses, 12 ns apiece: 1.7 ms).  Nothing is histogrammed
          // here; the radix select over the candidate records (k_s2_tail, LDS histograms per workgroup) does every digit.
        } else if (ncand < (unsigned)kWaveSlots) {  // generic data: one candidate per visit, every visit another digit
          if (cnt) atomicAdd(&ws->hist[dg], 1ull);
        } else if (const unsigned long long cm = __ballot(cnt)) {
          const int leader = __ffsll((long long)cm) - 1;
          const unsigned int d0 = __shfl(dg, leader, 64);
          const unsigned long long same = __ballot(cnt && dg == d0);
          // four runs at a time (a band of lattice data holds two or three distinct keys, interleaved in the vector: a single
          // run was flushed at every change of key -- 4.4 ms instead of 0.5)
          int hit = -1;
#pragma unroll
          for (int k = 0; k < 4; ++k) hit = (run_digit[k] == d0) ? k : hit;
          if (hit < 0) {
            hit = run_next;
            run_next = (run_next + 1) & 3;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              if (k == hit) {
                if (run_count[k] && lane == 0) atomicAdd(&ws->hist[run_digit[k]], (unsigned long long)run_count[k]);
                run_digit[k] = d0;
                run_count[k] = 0;
              }
            }
          }
#pragma unroll
          for (int k = 0; k < 4; ++k) run_count[k] += (k == hit) ? (unsigned int)__popcll(same) : 0u;
          if (cnt && dg != d0) atomicAdd(&ws->hist[dg], 1ull);
        }
      }
      ncand += (unsigned int)__popcll(m);
    }
    return spec_kept ? kept : dropped;
  };
#pragma unroll
  for (int k = 0; k < UNROLL; ++k) {
    const int64_t i = wbase + k * 64;
    const f64x2 a = *reinterpret_cast<const f64x2*>(wl + (0 * UNROLL + k) * 1024 + lane * 16);
    const f64x2 b = *reinterpret_cast<const f64x2*>(wl + (1 * UNROLL + k) * 1024 + lane * 16);
    const f64x2 c = *reinterpret_cast<const f64x2*>(wl + (2 * UNROLL + k) * 1024 + lane * 16);
    const bool valid = i < n2;
    f64x2 o;
    o.x = visit(valid, (b.x + c.x) + a.x, 2 * i + ioff, b.x, c.x);      // shiftedIndBallL0.jl:66  xk .+ sj .+ q
    o.y = visit(valid, (b.y + c.y) + a.y, 2 * i + 1 + ioff, b.y, c.y);
    if constexpr (WRITE) {
      if (valid) __builtin_nontemporal_store(o, y2 + i);
    }
  }
  if ((n & 1) && gwave == 0) {  // the odd last element rides with wave 0 (all of its lanes call visit)
    const int64_t i = n - 1;
    const double o = visit(lane == 0, (xk_[i] + sj_[i]) + q_[i], i + ioff, xk_[i], sj_[i], 1);
    if constexpr (WRITE) {
      if (lane == 0) y_[i] = o;
    }
  }
  if (ioff && gwave == 0) {  // and so does the caller's element 0 of an 8-byte-misaligned view
    const double o = visit(lane == 0, (xk_[-1] + sj_[-1]) + q_[-1], 0, xk_[-1], sj_[-1], 0);
    if constexpr (WRITE) {
      if (lane == 0) y_[-1] = o;
    }
  }
  if constexpr (tie) {
    for (int off = 32; off >= 1; off >>= 1) { eq_hi += __shfl_xor(eq_hi, off, 64); eq_lo += __shfl_xor(eq_lo, off, 64); }
    if (lane == 0) {
      cls[1 + gwave] = ClassCount{eq_hi, eq_lo};
      unsigned int th = eq_hi, tl = eq_lo;
      if (gwave == 0) {  // (only lane 0 of wave 0 visited the stragglers)
        cls[0] = ClassCount{seq_hi[0], seq_lo[0]};
        cls[1 + nwaves_total] = ClassCount{seq_hi[1], seq_lo[1]};
        th += seq_hi[0] + seq_hi[1];
        tl += seq_lo[0] + seq_lo[1];
      }
      const int shard = (int)(gwave % kShards) * kShardStride;
      if (th) atomicAdd(&ws->shard_hi[shard], (unsigned long long)th);
      if (tl) atomicAdd(&ws->shard_lo[shard], (unsigned long long)tl);
    }
  }
  for (int off = 32; off >= 1; off >>= 1) above += __shfl_xor(above, off, 64);
  if (__any(full) && lane == 0 && __hip_atomic_load(&ws->fs.overflow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
    atomicExch(&ws->fs.overflow, 1);  // (raised once, not by every wave that found the list full)
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (run_count[k]) atomicAdd(&ws->hist[run_digit[k]], (unsigned long long)run_count[k]);
    counts[gwave] = WaveCount{ncand, above};
    {
      const int shard = (int)(gwave % kShards) * kShardStride;
      // (measured on one box, interleaved: these two atomics per wave cost ~13 us of the pass; packing both totals into
      //  ONE 64-bit atomic per wave was slower still, 0.645 vs 0.622 ms per call; the per-candidate histogram atomics
      //  above cost nothing measurable)
      if (above) atomicAdd(&ws->shard_above[shard], (unsigned long long)above);
      if (ncand) atomicAdd(&ws->shard_cand[shard], (unsigned long long)ncand);
    }
  }
  };  // body
  if (tie_mode) body(std::true_type{});
  else body(std::false_type{});
}

// Candidate kernels walk the regions TRANSPOSED: a wavefront takes 64 regions at a time, lane l owns region w0 + l and
// steps through its few entries (4 loads in flight per lane), so the count words are read coalesced and no lane waits
// on a chain of dependent loads.  f(key, index, value) is called for every valid entry.
template <class F>
__device__ __forceinline__ void for_each_candidate(const WaveCount* counts, int64_t nregions, const Cand* cand, int64_t novf, F&& f) {
  {  // the shared overflow list behind the regions (usually empty)
    const Cand* ov = cand + nregions * kWaveSlots;
    const int64_t nthreads = (int64_t)gridDim.x * blockDim.x;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < novf; e += nthreads) f(ov[e].key, ov[e].idx, ov[e].val);
  }
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t w0 = wave * 64; w0 < nregions; w0 += nwaves * 64) {
    const int64_t w = w0 + lane;
    unsigned int c = 0;
    if (w < nregions) c = counts[w].cand;
    const int cnt = c < (unsigned)kWaveSlots ? (int)c : kWaveSlots;
    const int64_t e0 = (w < nregions ? w : nregions - 1) * kWaveSlots;
    constexpr int U = 8;  // entries in flight per lane (a region holds ~4 on average)
    for (int s0 = 0; __any(s0 < cnt); s0 += U) {
      Cand c[U];
      // (tried: clamped unconditional loads so that no load waits in a branch block of its own, and 12 / 16 entries in flight:
      //  k_s2_compact at n = 1e8 13.0 / 25.9 us -> 15.5 / 29.6 and 13.4 / 24.6 -- the walk is bound by its ~2e5 scattered
      //  128-byte lines, not by latency)
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (s0 + u < cnt) c[u] = cand[e0 + s0 + u];
#pragma unroll
      for (int u = 0; u < U; ++u)
        if (s0 + u < cnt) f(c[u].key, c[u].idx, c[u].val);
    }
  }
}

// One workgroup: the verdict (is the r-th largest provably inside the band?), then the first scan step.
// Tie mode: the order is  above | class hi (one key) | candidates strictly between | class lo (one key) | below; the cut
// falls into one of the three middle parts.  Inside a class: all of it kept (resolved), or FastState::todo |= kTodoTieScan
// with the quota in SelState::quota and (t_eq, t_ge) set -- k_s2_tail finds icut.  Among the candidates: as without ties.
__global__ __launch_bounds__(256) void k_s2_scan_verify(SelWs* ws, int64_t r) {
  __shared__ unsigned long long scratch[8];
  __shared__ int sok;
  __shared__ SelState sst;
  unsigned long long above = 0, cand = 0, chi = 0, clo = 0;
  {  // the 256 lanes add up the shards of the main pass
    __shared__ unsigned long long red[4][4];
    const bool tie = ws->fs.tie != 0;
    for (int k = threadIdx.x; k < kShards; k += 256) {
      above += ws->shard_above[k * kShardStride];
      cand += ws->shard_cand[k * kShardStride];
      if (tie) { chi += ws->shard_hi[k * kShardStride]; clo += ws->shard_lo[k * kShardStride]; }
    }
    for (int off = 32; off >= 1; off >>= 1) {
      above += __shfl_xor(above, off, 64);
      cand += __shfl_xor(cand, off, 64);
      chi += __shfl_xor(chi, off, 64);
      clo += __shfl_xor(clo, off, 64);
    }
    if ((threadIdx.x & 63) == 0) {
      red[0][threadIdx.x >> 6] = above; red[1][threadIdx.x >> 6] = cand;
      red[2][threadIdx.x >> 6] = chi; red[3][threadIdx.x >> 6] = clo;
    }
    __syncthreads();
    above = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    cand = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
    chi = (red[2][0] + red[2][1]) + (red[2][2] + red[2][3]);
    clo = (red[3][0] + red[3][1]) + (red[3][2] + red[3][3]);
  }
  if (threadIdx.x == 0) {
    FastState& f = ws->fs;
    SelState& s = ws->st;
    const unsigned long long ur = (unsigned long long)r;
    f.cnt_above = above;
    f.cand_count = cand;
    f.cls_hi = chi;
    f.cls_lo = clo;
    bool ok = !f.overflow && above < ur && ur <= above + chi + cand + clo;
    bool scan = false;  // the cut lies among the candidates: first scan step below
    if (ok && f.tie) {
      f.todo = kTodoFinal;
      if (ur <= above + chi) {               // inside class hi
        s.quota = (int64_t)(ur - above);
        s.phase = 2;
        if ((unsigned long long)s.quota == chi) { s.t_ge = f.t_hi; s.t_eq = ~0ull; s.icut = -1; }
        else { s.t_ge = f.t_hi + 1; s.t_eq = f.t_hi; s.icut = -1; f.todo |= kTodoTieScan; f.tie_class = 0; }
      } else if (ur <= above + chi + cand) {  // among the candidates (class hi kept, class lo dropped: t_floor = t_lo + 1)
        s.quota = (int64_t)(ur - above - chi);
        scan = true;
      } else {                               // inside class lo (everything above it is kept)
        s.quota = (int64_t)(ur - above - chi - cand);
        s.phase = 2;
        if ((unsigned long long)s.quota == clo) { s.t_ge = f.t_lo; s.t_eq = ~0ull; s.icut = -1; }
        else { s.t_ge = f.t_lo + 1; s.t_eq = f.t_lo; s.icut = -1; f.todo |= kTodoTieScan; f.tie_class = 1; }
      }
    } else if (ok) {
      s.quota = (int64_t)(ur - above);
      scan = true;
      if (f.t_lo == f.t_hi && (unsigned long long)s.quota == cand) {  // a one-key band, all of it kept
        s.phase = 2;
        s.t_ge = f.t_lo;
        s.t_eq = ~0ull;
      }
    }
    if (ok && scan && f.crowded && s.phase != 2) {  // no digit was histogrammed by the main pass (see there): k_s2_tail does them all
      f.todo |= kTodoCandSelect;
      scan = false;
    }
    f.ok = ok ? 1 : 0;
    sok = (ok && scan) ? 1 : 0;
    sst = s;
  }
  __syncthreads();
  if (sok && sst.phase != 2) sel_scan_step(ws->hist, sst, &ws->st, scratch);
  __syncthreads();
  for (int b = threadIdx.x; b < kBins; b += blockDim.x) ws->hist[b] = 0ull;
}

// after the first candidate digit: the candidates still in play (inside the chosen bin, or tied key) -> short list.
// WRITE (single-pass form): candidates that the first digit already puts above the cut get their kept value here; the
// short list carries the values of the rest and k_s2_finish stores those that make it -- no separate fix-up walk.
template <bool WRITE, bool BINF>
__global__ __launch_bounds__(256) void k_s2_compact(double* y, const Cand* cand, SelWs* ws, uint64_t* list_key,
                                                     int64_t* list_idx, double* list_val, const WaveCount* counts,
                                                     int64_t nregions, unsigned int ovf_cap, const double* xk, const double* sj,
                                                     double delta) {
  const SelState st = ws->st;
  const uint64_t t_mid = ws->fs.t_mid;
  if (!ws->fs.ok || (ws->fs.todo & kTodoCandSelect)) return;
  const bool wr = WRITE;
  if (!wr && st.phase == 2) return;
  const int hs = st.shift + st.width;
  const unsigned int novf = ws->fs.ovf_count < ovf_cap ? ws->fs.ovf_count : ovf_cap;
  for_each_candidate(counts, nregions, cand, (int64_t)novf, [&](uint64_t key, int64_t i, double val) {
    bool in = false, keep = false;
    if (st.phase == 2) {  // resolved by the first digit already
      keep = (key >= st.t_ge) || (key == st.t_eq && i <= st.icut);
    } else if (st.phase == 0) {  // key digits: compare everything decided so far
      const KeyPos kp = key_pos(key, st.base, st.shift, st.width, st.clamp);
      in = kp.in;
      keep = kp.above;
    } else {  // index digits of the tied key: lower indices are kept first
      const uint64_t itop = ((uint64_t)i) >> hs;
      in = key == st.t_eq && itop == st.prefix;
      keep = (key >= st.t_ge) || (key == st.t_eq && itop < st.prefix);
    }
    // (a list that has already overflowed is not appended to any more: a bucket of 1e4-1e6 survivors -- moderately tied
    //  data -- was tens of thousands of atomics on this one counter, 73 us of the walk; k_s2_tail takes over anyway)
    if (in && __hip_atomic_load(&ws->fs.list_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) <= (unsigned)kShortList) {
      // one atomic per wave instruction, not per survivor (they all hit one address: ~12 ns each, serialised)
      const unsigned long long am = __ballot(1);
      const int lane = threadIdx.x & 63;
      const int leader = __ffsll((long long)am) - 1;
      unsigned int base = 0;
      if (lane == leader) base = atomicAdd(&ws->fs.list_count, (unsigned int)__popcll(am));
      base = __shfl(base, leader, 64);
      const unsigned int slot = base + (unsigned int)__popcll(am & ((1ull << lane) - 1ull));
      if (slot < (unsigned)kShortList) {
        list_key[slot] = key;
        list_idx[slot] = i;
        if constexpr (WRITE) list_val[slot] = val;
      }
    }
    if constexpr (WRITE) {
      // decided by the digits so far (not inside the bucket still being refined): rewritten where the main pass guessed wrong
      const bool spec = key >= t_mid;
      if (wr && !in && keep != spec) y[i] = keep ? val : sel_dropped<BINF>(xk, sj, i, delta);
    }
  });
}

// one workgroup: finishes the selection on the short list, entirely in LDS
template <bool WRITE, bool BINF>
__global__ __launch_bounds__(1024) void k_s2_finish(double* y, SelWs* ws, const uint64_t* list_key, const int64_t* list_idx,
                                                     const double* list_val, const double* xk, const double* sj, double delta) {
  __shared__ uint64_t lk[kShortList];
  __shared__ int64_t li[kShortList];
  __shared__ unsigned int h[kBins];
  __shared__ unsigned long long scratch[8];
  __shared__ SelState sst;
  const int t = threadIdx.x;
  if (!ws->fs.ok || (ws->fs.todo & kTodoCandSelect)) return;
  if (t == 0) sst = ws->st;
  __syncthreads();
  if (sst.phase == 2) return;
  const unsigned int m = ws->fs.list_count;
  if (m > (unsigned)kShortList) {
    // too many survivors for LDS (a key shared by thousands of candidates: moderately tie-heavy data).  The state after the
    // first digit stays in ws->st; k_s2_tail continues the radix select over the candidate regions with its whole grid.
    if (t == 0) ws->fs.todo |= kTodoCandSelect;
    return;
  }
  for (unsigned int e = t; e < m; e += 1024) { lk[e] = list_key[e]; li[e] = list_idx[e]; }
  for (int guard = 0; guard < 16; ++guard) {  // <= 6 key digits + <= 6 index digits
    __syncthreads();
    const SelState st = sst;
    if (st.phase == 2) break;
    for (int b = t; b < kBins; b += 1024) h[b] = 0u;
    __syncthreads();
    const int hs = st.shift + st.width;
    const uint64_t dmask = ((uint64_t)1 << st.width) - 1;
    for (unsigned int e = t; e < m; e += 1024) {
      const uint64_t key = lk[e];
      if (st.phase == 0) {
        const KeyPos kp = key_pos(key, st.base, st.shift, st.width, st.clamp);
        if (kp.in) atomicAdd(&h[kp.digit], 1u);
      } else if (key == st.t_eq) {
        const uint64_t i = (uint64_t)li[e];
        if ((i >> hs) == st.prefix) atomicAdd(&h[(i >> st.shift) & dmask], 1u);
      }
    }
    __syncthreads();
    sel_scan_step(h, st, &sst, scratch);
  }
  __syncthreads();
  if (t == 0) ws->st = sst;
  if (WRITE) {  // the short-list entries the main pass guessed wrong (everything outside the list was settled by k_s2_compact)
    const SelState fin = sst;
    const uint64_t t_mid = ws->fs.t_mid;
    for (unsigned int e = t; e < m; e += 1024) {
      const uint64_t key = lk[e];
      const int64_t i = li[e];
      const bool keep = (key >= fin.t_ge) || (key == fin.t_eq && i <= fin.icut);
      if (keep != (key >= t_mid)) y[i] = keep ? list_val[e] : sel_dropped<BINF>(xk, sj, i, delta);
    }
  }
}

// final pass of the fast path: v recomputed from q, xk, sj (y untouched so far)
template <bool BINF>
__global__ __launch_bounds__(256) void k_sel_final_q(double* y_, const double* q_, const double* xk_, const double* sj_,
                                                      int64_t n, const SelWs* ws, double delta, int ioff) {
  // (ioff: as k_s2_main)
  if (!ws->fs.ok) return;  // prediction not verified: the host runs the full-vector path afterwards
  const bool poisoned = ws->fs.ok == 2;  // k_s2_tail gave up waiting for its own workgroups: the thresholds are garbage -> NaN
  const SelState st = ws->st;
  constexpr int UNROLL = 6;  // KiB per wave and vector, as k_sep_lds
  __shared__ __attribute__((aligned(16))) char dma[4 * 3 * UNROLL * 1024];
  typedef __attribute__((address_space(3))) void lds_void;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  char* wl = dma + wave * (3 * UNROLL * 1024);
  f64x2* y = reinterpret_cast<f64x2*>(y_);
  const f64x2* q = reinterpret_cast<const f64x2*>(q_);
  const f64x2* xk = reinterpret_cast<const f64x2*>(xk_);
  const f64x2* sj = reinterpret_cast<const f64x2*>(sj_);
  const int64_t n2 = n >> 1;
  const int64_t base = ((int64_t)blockIdx.x * 4 + wave) * (64 * UNROLL) + lane;
#pragma unroll
  for (int k = 0; k < UNROLL; ++k) {
    int64_t i = base + k * 64;
    if (i >= n2) i = n2 - 1;
    __builtin_amdgcn_global_load_lds((const void*)(q + i), (lds_void*)(wl + (0 * UNROLL + k) * 1024), 16, 0, 2);
    __builtin_amdgcn_global_load_lds((const void*)(xk + i), (lds_void*)(wl + (1 * UNROLL + k) * 1024), 16, 0, 2);
    __builtin_amdgcn_global_load_lds((const void*)(sj + i), (lds_void*)(wl + (2 * UNROLL + k) * 1024), 16, 0, 2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int k = 0; k < UNROLL; ++k) {
    const int64_t i = base + k * 64;
    const f64x2 a = *reinterpret_cast<const f64x2*>(wl + (0 * UNROLL + k) * 1024 + lane * 16);
    const f64x2 b = *reinterpret_cast<const f64x2*>(wl + (1 * UNROLL + k) * 1024 + lane * 16);
    const f64x2 c = *reinterpret_cast<const f64x2*>(wl + (2 * UNROLL + k) * 1024 + lane * 16);
    if (i < n2) {
      f64x2 r;
      r.x = sel_out<BINF>((b.x + c.x) + a.x, 2 * i + ioff, b.x, c.x, st, delta);
      r.y = sel_out<BINF>((b.y + c.y) + a.y, 2 * i + 1 + ioff, b.y, c.y, st, delta);
      if (poisoned) r = f64x2{__longlong_as_double(0x7ff8000000000000ll), __longlong_as_double(0x7ff8000000000000ll)};
      __builtin_nontemporal_store(r, y + i);
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const int64_t i = n - 1;
    const double o = sel_out<BINF>((xk_[i] + sj_[i]) + q_[i], i + ioff, xk_[i], sj_[i], st, delta);
    y_[i] = poisoned ? __longlong_as_double(0x7ff8000000000000ll) : o;
  }
  if (ioff && blockIdx.x == 0 && threadIdx.x == 64) {
    const double o = sel_out<BINF>((xk_[-1] + sj_[-1]) + q_[-1], 0, xk_[-1], sj_[-1], st, delta);
    y_[-1] = poisoned ? __longlong_as_double(0x7ff8000000000000ll) : o;
  }
}

// =============================================================================================
// In-launch synchronised kernels (round 2).  The multi-launch pipelines above pay for their serial chain of small
// dependent launches and for one host read-back per call: 116 us per call at n = 1e6 against a 5 us bandwidth floor, and
// ~90 us of the 0.62 ms at n = 1e8.  Here the workgroups of ONE launch (grid <= number of CUs, all resident) meet at grid
// barriers (spx_grid_barrier) and every workgroup redoes the cheap scalar steps (histogram scans) for itself, so that no
// step waits for a single workgroup of another launch:
//   k_sel_coop   exact MSD radix select in one launch.  REG: 65536 < n <= 8 Ki x number of CUs -- v stays in registers
//                (<= 8 elements per lane), the vectors are read once and y written once.  !REG: any n, v parked in y.
//   k_s2_front   sample + band (replaces k_sel_init, k_s2_sample, k_s2_pick): each lane keeps its sample in a register.
// The candidate kernels behind the main pass stay separate launches (k_s2_scan_verify, k_s2_compact, k_s2_finish): fused
// into one in-launch synchronised "tail" (64 x 1024 lanes, two barriers) they measured 49 us against 31 us -- the walk over
// the 130 208 per-wave candidate regions wants thousands of waves, a grid barrier wants few workgroups.  What the host no
// longer does is read the verdict back: the exact full-vector select (k_sel_coop, fallback = 1) is queued behind them
// unconditionally and returns at once (5 us) when the verdict is positive.
// State that must be zero on entry lives in spx_ctx::sync (SelSync), zeroed once; each kernel leaves clean what the next
// one expects (who clears what is noted at the fields).
// =============================================================================================
#ifdef SPX_SEL_PROFILE  // A/B builds only (csrc/build.sh -DSPX_SEL_PROFILE): phase time stamps of workgroup 0, 10 ns units
__device__ unsigned long long g_sel_stamp[64];
#define SEL_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_sel_stamp[k] = wall_clock64(); } while (0)
extern "C" __attribute__((visibility("default"))) int spx_debug_sel_stamps(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sel_stamp), sizeof(g_sel_stamp));
}
#else
#define SEL_STAMP(k) do { } while (0)
#endif
constexpr int kCoopMaxPass = 12;  // <= 6 key digits + <= 6 index digits
// The one-launch forms that keep the vector on chip (n <= 2^23) never reach their last passes (<= 7 key + 2 index digits): the
// histogram of pass kSelStatSlot, zero like every other on entry, holds their bucket statistics instead -- word 2p: max(~key),
// word 2p + 1: max(key) over the elements inside the bucket of pass p, both gathered with atomicMax, so no launch has to
// initialise them (an explicit reset by workgroup 0 at the start of every launch cost the generic path 0.4 us per call).
constexpr int kSelStatSlot = kCoopMaxPass - 1;
#ifndef SPX_COOP_FOLD
#define SPX_COOP_FOLD 1
#endif
constexpr bool kCoopFold = SPX_COOP_FOLD != 0;
constexpr int kCoopEpl = 8;       // REG: elements per lane at most (16 spill: 1024-lane workgroups leave 128 VGPRs per lane)
struct SelSync {
  SpxSyncHeader hdr;  // grid-barrier counters (a launch uses hdr.bar[parity] and clears hdr.bar[parity ^ 1])
  // One global histogram per pass.  k_sel_coop alternates between sets 0 and 1: a launch uses the clean one and clears
  // the other (dirty from the launch before it; the host keeps the flags, spx_ctx::sel_hist_*).  Set 2 belongs to the
  // fallback inside k_s2_tail, which clears it itself (one more barrier on a path that is rare and slow anyway).
  unsigned long long chist[3][kCoopMaxPass][kBins];
  unsigned long long fhist1[kBins];        // k_s2_front: top digit of the sample keys            (cleared by the fallback launch)
  unsigned long long fhist2[5][2][kBins];  // k_s2_front: digits 2..6 of the two rank selections   (cleared by the fallback launch)
  unsigned long long ftie[8];              // k_s2_front, tie mode: [2 sel + 0 / 1] = largest key / largest ~key among the samples of selection sel's bucket (cleared by the fallback launch)
  SelWs ws;                                // st, fs, hist, shards (cleared by k_s2_front; fs.smax by the fallback launch)
  // k_s2_tail, tie scan: class members per workgroup slice and the cut found by the slice that holds it -- words that are
  // their own ready flag (value + 1, zero = not yet written; cleared by the launch itself after its last reader)
  unsigned long long tie_part[256];
  unsigned long long tie_cut;
  // k_s2_tail, candidate select: smallest / largest key and number of the candidates inside the bucket of each pass (set to
  // ~0 / 0 / 0 by the launch before its first pass): a bucket of ONE key ends the key digits at once
  unsigned long long cmin[kCoopMaxPass], cmax[kCoopMaxPass], ccnt[kCoopMaxPass];
};

static_assert(sizeof(SelSync) <= kSpxSyncSelBytes, "SelSync outgrew its share of spx_ctx::sync");

struct CoopShared {
  unsigned int lh[kBins];
  unsigned long long scratch[24];
  SelState sst;
  unsigned long long kmm[2];  // register form: smallest / largest key of this workgroup's elements inside the bucket of a pass
};

// sel_scan_step for the 1024-lane workgroups of k_sel_coop: four bins per lane, read with agent-scope atomic loads -- the
// histogram is written with agent-scope atomic adds only, so the passes meet in spx_grid_rendezvous, without the L2
// write-back / invalidate of a fenced barrier (most of its cost: 6.7 -> 3.x us per pass at 123 workgroups).
// scratch = 24 words of shared memory; the new state is written to *out (shared) by lane 0, no barrier after that.
__device__ __forceinline__ void coop_scan_step(unsigned long long* hist, const SelState st, SelState* out,
                                               unsigned long long* scratch) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int nb = 1 << st.width;
  const bool desc = (st.phase == 0);
  constexpr int PER = kBins / 1024;
  unsigned long long loc[PER];
  unsigned long long sum = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int pos = t * PER + k;
    const int bin = desc ? (kBins - 1 - pos) : pos;
    loc[k] = (bin < nb) ? __hip_atomic_load(hist + bin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
    sum += loc[k];
  }
  unsigned long long inc = sum;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long up = __shfl_up(inc, off, 64);
    if (lane >= off) inc += up;
  }
  if (lane == 63) scratch[w] = inc;
  __syncthreads();
  unsigned long long run = inc - sum;
  for (int k = 0; k < w; ++k) run += scratch[k];
  const unsigned long long quota = (unsigned long long)st.quota;
  unsigned long long* found = scratch + 16;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    if (run < quota && run + loc[k] >= quota) {
      const int pos = t * PER + k;
      found[0] = (unsigned long long)(desc ? (kBins - 1 - pos) : pos);
      found[1] = run;
      found[2] = loc[k];
    }
    run += loc[k];
  }
  __syncthreads();
  if (t == 0) *out = sel_advance(st, found[0], found[1], found[2]);
}

// Called by all lanes of a workgroup right after coop_scan_step (whose lane 0 wrote *out) on a key-digit pass whose bucket
// statistics were gathered (smallest / largest key and number of elements inside the bucket, agent-scope atomics, complete
// since the pass's rendezvous).  A bucket that holds ONE key needs no further key digits: all of it kept, or the index
// tie-break on that key -- the state the remaining digits would arrive at, 2-4 sweeps later.  st = the state of the pass.
// ccnt == nullptr: the number of elements inside the bucket is known to the caller (cnt_known: the count of the bin the
// previous scan selected).
__device__ __forceinline__ void sel_single_key_shortcut(const SelState st, SelState* out, const unsigned long long* cmin,
                                                        const unsigned long long* cmax, const unsigned long long* ccnt,
                                                        unsigned long long cnt_known = 0ull, bool min_inverted = false) {
  __syncthreads();
  if (threadIdx.x != 0) return;
  // (min_inverted: the word holds max(~key) -- gathered with atomicMax into memory that starts out zero, see kSelStatSlot)
  const unsigned long long lo0 = __hip_atomic_load(cmin, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long lo = min_inverted ? ~lo0 : lo0;
  const unsigned long long hi = __hip_atomic_load(cmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned long long cnt = ccnt ? __hip_atomic_load(ccnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : cnt_known;
  if (!(lo == hi && cnt > 0 && (unsigned long long)st.quota <= cnt)) return;
  SelState o = st;
  o.pad = 0;
  if ((unsigned long long)st.quota == cnt) {   // the whole bucket is kept
    o.phase = 2;
    o.t_ge = lo > st.t_floor ? lo : st.t_floor;
  } else {                                     // more equal keys than quota: ties by ascending index
    o.t_ge = lo + 1;
    o.t_eq = lo;
    o.phase = 1;
    o.prefix = 0;
    const int top = st.idx_bits;
    const int w = top % kDigitBits ? top % kDigitBits : kDigitBits;
    if (top == 0) { o.phase = 2; o.icut = 0; }
    o.shift = top - w;
    o.width = w;
  }
  *out = o;
}

__device__ __forceinline__ void sel_state_init(SelState& s, int64_t n, int64_t r) {
  int bits = 0;
  while (bits < 63 && ((int64_t)1 << bits) < n) ++bits;
  s.idx_bits = bits;
  s.prefix = 0;
  s.icut = -1;
  s.t_eq = ~0ull;
  s.t_ge = ~0ull;
  s.pad = s.pad2 = s.pad3 = 0;
  s.t_floor = 0;
  s.base = 0;
  s.clamp = 0;
  s.quota = 0;
  s.shift = 0;
  s.width = 0;
  if (r <= 0) {            // nothing kept
    s.phase = 2;
  } else if (r >= n) {     // everything kept
    s.phase = 2; s.t_ge = 0ull;
  } else {
    s.phase = 0; s.shift = 64 - kDigitBits; s.width = kDigitBits; s.quota = r;
  }
}

// The exact selection, executed by every workgroup of a resident grid of 1024-lane workgroups.  hist[p] must be zero on
// entry for every pass p that runs.  Lane (block, t) owns the elements gtid, gtid + NT, ... in every pass (so v parked in
// y needs no fence between passes, and y may alias q: q[i] is read before y[i] is written by the same lane).
template <bool BINF, bool REG, class T = double>
__device__ __forceinline__ void coop_select(T* y, const T* q, const T* xk, const T* sj, int64_t n,
                                            int64_t r, T delta, unsigned long long (*hist)[kBins], unsigned int* bar,
                                            unsigned int& nbar, CoopShared& sh, SpxSyncHeader* hdr) {
  unsigned long long* const stat = hist[kSelStatSlot];  // (REG: bucket statistics, see kSelStatSlot)
  const int t = threadIdx.x;
  const int64_t NT = (int64_t)gridDim.x * blockDim.x;
  const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + t;
  // (form that parks v in y: 16-byte accesses when all four vectors allow them -- the exact select behind a failed prediction)
  constexpr bool kF64 = std::is_same<T, double>::value;
  const bool vec2 = kF64 && !REG && n >= 2 && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(q) |
                                        reinterpret_cast<uintptr_t>(xk) | reinterpret_cast<uintptr_t>(sj)) & 15u) == 0;
  // One element's contribution to the histogram of the pass whose state is st.  REG: plain LDS atomics -- the unrolled visits
  // are independent and overlap; wave-aggregated (hist_add_agg) each is a serial chain of ballots, 3 us per full sweep of
  // generic data at 16 waves per CU, and what it saves on one-key data (64 lanes on one LDS address: ~64 clocks per
  // instruction) is 3 us per pass of 8 elements per lane.  The form that walks the whole vector keeps the aggregation.
  struct KeyStat { unsigned long long kmin, kmax; };  // (REG, a tracked pass: keys of this lane's elements inside the bucket)
  KeyStat nostat{0ull, 0ull};
  auto visit_st = [&](const SelState& st, T vv, int64_t i, auto track_tag, KeyStat& ks) {
    constexpr bool kTrack = decltype(track_tag)::value;
    const uint64_t key = key_of(vv);
    bool in = true;
    unsigned int dg;
    if (st.phase == 0) {
      if (st.pad == 1) {
        dg = fold_digit(key);
      } else if (st.pad == 2) {
        dg = fold_digit_f32(key);
      } else {
        const KeyPos kp = key_pos(key, st.base, st.shift, st.width, st.clamp);
        in = kp.in;
        dg = kp.digit;
        if (kTrack && in) { ks.kmin = key < ks.kmin ? key : ks.kmin; ks.kmax = key > ks.kmax ? key : ks.kmax; }
      }
    } else {
      const int hs = st.shift + st.width;
      in = key == st.t_eq && (((uint64_t)i) >> hs) == st.prefix;
      dg = (unsigned int)((((uint64_t)i) >> st.shift) & (((uint64_t)1 << st.width) - 1));
    }
    if constexpr (REG) { if (in) atomicAdd(&sh.lh[dg], 1u); }
    else hist_add_agg(sh.lh, dg, in);
  };
  if (t == 0) {
    sel_state_init(sh.sst, n, r);
    if (sh.sst.phase == 0 && kCoopFold) sh.sst.pad = kF64 ? 1 : 2;  // first digit: fold_digit / fold_digit_f32 (binades)
  }
  T v[REG ? kCoopEpl : 1];
  bool fused = false;  // REG: the first digit was histogrammed while the loads were in flight (it depends on the key alone)
  if constexpr (REG) {
    for (int b = t; b < kBins; b += blockDim.x) sh.lh[b] = 0u;
    __syncthreads();
    const SelState st0 = sel_uniform(sh.sst);
    fused = st0.phase == 0;
#pragma unroll
    for (int k = 0; k < kCoopEpl; ++k) {
      const int64_t i = gtid + (int64_t)k * NT;
      const int64_t ic = i < n ? i : n - 1;            // clamped, unconditional: all the loads in flight at once
      const T xv = xk[ic], sv = sj[ic], qv = q[ic];
      v[k] = (i < n) ? (xv + sv) + qv : (T)0;          // shiftedIndBallL0.jl:66
    }
    if (fused) {
#pragma unroll
      for (int k = 0; k < kCoopEpl; ++k) {
        const int64_t i = gtid + (int64_t)k * NT;
        if (i < n) visit_st(st0, v[k], i, std::false_type{}, nostat);
      }
    }
  }
  SEL_STAMP(32);
  int p = 0;
  unsigned long long cnt_before = ~0ull;  // elements inside the bucket of the previous pass
  for (; p < kCoopMaxPass; ++p) {
    __syncthreads();
    const SelState st = sel_uniform(sh.sst);
    if (st.phase == 2) break;  // the same in every workgroup: they all computed it from the same histograms
    const bool counted = REG && p == 0 && fused;
    // REG: once a pass has failed to split its bucket (every element in one bin: tie-heavy data) the next pass also tracks the
    // smallest and largest key inside the bucket; equal ends skip the remaining key digits (as k_sel_lds; a constant vector of
    // 1e6 elements: 95 us per call for six key digits and two index digits)
    const unsigned long long cnt_in = p > 0 ? sh.scratch[18] : ~0ull - 1ull;  // (coop_scan_step: found[2] of the previous scan)
    const bool track = REG && p > 1 && p < kSelStatSlot && st.phase == 0 && st.pad == 0 && cnt_in == cnt_before;
    cnt_before = cnt_in;
    if (!counted) {
      for (int b = t; b < kBins; b += blockDim.x) sh.lh[b] = 0u;
      if (t == 0) { sh.kmm[0] = ~0ull; sh.kmm[1] = 0ull; }
      __syncthreads();
    }
    auto visit = [&](T vv, int64_t i) { visit_st(st, vv, i, std::false_type{}, nostat); };
    if constexpr (REG) {
      if (!counted) {
        if (track) {
          KeyStat ks{~0ull, 0ull};
#pragma unroll
          for (int k = 0; k < kCoopEpl; ++k) {
            const int64_t i = gtid + (int64_t)k * NT;
            if (i < n) visit_st(st, v[k], i, std::true_type{}, ks);
          }
          unsigned long long kmin = ks.kmin, kmax = ks.kmax;
          for (int off = 32; off >= 1; off >>= 1) {
            const unsigned long long a_ = __shfl_xor(kmin, off, 64), c_ = __shfl_xor(kmax, off, 64);
            kmin = a_ < kmin ? a_ : kmin;
            kmax = c_ > kmax ? c_ : kmax;
          }
          if ((t & 63) == 0 && kmin <= kmax) { atomicMin(&sh.kmm[0], kmin); atomicMax(&sh.kmm[1], kmax); }
        } else {
#pragma unroll
          for (int k = 0; k < kCoopEpl; ++k) {
            const int64_t i = gtid + (int64_t)k * NT;
            if (i < n) visit(v[k], i);
          }
        }
      }
    } else if (vec2) {
      if constexpr (kF64) {
      // 16-byte accesses (all four vectors 16-byte aligned): lane (block, t) owns the pairs gtid, gtid + NT, ... in every pass
      const int64_t n2 = n >> 1;
      f64x2* y2 = reinterpret_cast<f64x2*>(y);
      if (p == 0) {
        const f64x2* q2 = reinterpret_cast<const f64x2*>(q);
        const f64x2* x2 = reinterpret_cast<const f64x2*>(xk);
        const f64x2* s2 = reinterpret_cast<const f64x2*>(sj);
        for (int64_t pr = gtid; pr < n2; pr += NT) {
          const f64x2 a = q2[pr], b = x2[pr], c = s2[pr];
          const f64x2 vv = f64x2{(b.x + c.x) + a.x, (b.y + c.y) + a.y};
          y2[pr] = vv;
          visit(vv.x, 2 * pr);
          visit(vv.y, 2 * pr + 1);
        }
        if ((n & 1) && gtid == 0) {
          const double vv = (xk[n - 1] + sj[n - 1]) + q[n - 1];
          y[n - 1] = vv;
          visit(vv, n - 1);
        }
      } else {
        for (int64_t pr = gtid; pr < n2; pr += NT) {
          const f64x2 vv = y2[pr];
          visit(vv.x, 2 * pr);
          visit(vv.y, 2 * pr + 1);
        }
        if ((n & 1) && gtid == 0) visit(y[n - 1], n - 1);
      }
      }
    } else {
      if (p == 0) {
        for (int64_t i = gtid; i < n; i += NT) {
          const T vv = (xk[i] + sj[i]) + q[i];
          y[i] = vv;
          visit(vv, i);
        }
      } else {
        for (int64_t i = gtid; i < n; i += NT) visit(y[i], i);
      }
    }
    __syncthreads();
    if (track && t == 0 && sh.kmm[0] <= sh.kmm[1]) {  // one global atomic of each kind per workgroup
      atomicMax(&stat[2 * p], ~sh.kmm[0]); atomicMax(&stat[2 * p + 1], sh.kmm[1]);
    }
    for (int b = t; b < kBins; b += blockDim.x) {
      const unsigned int c = sh.lh[b];
      if (c) atomicAdd(&hist[p][b], (unsigned long long)c);
    }
    SEL_STAMP(33 + 3 * p);
    // the histogram atomics of every wave have been performed (vmcnt) before its workgroup arrives; nothing else is exchanged
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    spx_grid_rendezvous(bar, (++nbar) * gridDim.x, hdr);
    SEL_STAMP(34 + 3 * p);
    coop_scan_step(hist[p], st, &sh.sst, sh.scratch);
    if (track) sel_single_key_shortcut(st, &sh.sst, &stat[2 * p], &stat[2 * p + 1], nullptr, cnt_in, true);
    SEL_STAMP(35 + 3 * p);
  }
  __syncthreads();
  const SelState fin = sel_uniform(sh.sst);
  // A workgroup of this context gave up waiting (spx_wait_expired): the thresholds are not to be trusted.  Every element is
  // then stored as NaN (t_ge above every key, v replaced below) and the next libspx call reports the failure.
  const bool poisoned = spx_poisoned(hdr);
  auto P = [&](T val) -> T { return poisoned ? (T)__longlong_as_double(0x7ff8000000000000ll) : val; };  // (a select: -0.0 stays -0.0)
  if constexpr (REG) {
#pragma unroll
    for (int k = 0; k < kCoopEpl; ++k) {
      const int64_t i = gtid + (int64_t)k * NT;
      if (i < n) y[i] = P(sel_out<BINF>(v[k], i, xk[i], sj[i], fin, delta));
    }
  } else if (vec2 && p != 0) {
    if constexpr (kF64) {
    const int64_t n2 = n >> 1;
    f64x2* y2 = reinterpret_cast<f64x2*>(y);
    const f64x2* x2 = reinterpret_cast<const f64x2*>(xk);
    const f64x2* s2 = reinterpret_cast<const f64x2*>(sj);
    for (int64_t pr = gtid; pr < n2; pr += NT) {
      const f64x2 vv = y2[pr], b = x2[pr], c = s2[pr];
      y2[pr] = f64x2{P(sel_out<BINF>(vv.x, 2 * pr, b.x, c.x, fin, delta)), P(sel_out<BINF>(vv.y, 2 * pr + 1, b.y, c.y, fin, delta))};
    }
    if ((n & 1) && gtid == 0) y[n - 1] = P(sel_out<BINF>(y[n - 1], n - 1, xk[n - 1], sj[n - 1], fin, delta));
    }
  } else {
    if (p == 0) {  // resolved before any pass (r <= 0 or r >= n): v was never parked in y
      for (int64_t i = gtid; i < n; i += NT) y[i] = P(sel_out<BINF>((xk[i] + sj[i]) + q[i], i, xk[i], sj[i], fin, delta));
    } else {
      for (int64_t i = gtid; i < n; i += NT) y[i] = P(sel_out<BINF>(y[i], i, xk[i], sj[i], fin, delta));
    }
  }
  SEL_STAMP(63);
}

// use_set: the histogram set of this launch; clear_set >= 0: the set to zero for a later launch (never use_set).
template <bool BINF, bool REG, class T = double>
__global__ __launch_bounds__(1024) void k_sel_coop(T* y, const T* q, const T* xk, const T* sj, int64_t n,
                                                    int64_t r, T delta, SelSync* ss, int parity, int use_set,
                                                    int clear_set) {
  __shared__ CoopShared sh;
  spx_bar_reset(ss->hdr.bar[parity ^ 1]);
  const int64_t total = (int64_t)kCoopMaxPass * kBins;
  unsigned int nbar = 0;
  if (clear_set >= 0) {
    unsigned long long* z = &ss->chist[clear_set][0][0];
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < total; b += (int64_t)gridDim.x * blockDim.x) z[b] = 0ull;
  }
  coop_select<BINF, REG, T>(y, q, xk, sj, n, r, delta, ss->chist[use_set], ss->hdr.bar[parity], nbar, sh, &ss->hdr);
}

// ---------------------------------------------------------------------------------------------
// k_sel_lds (round 3): the one-launch exact select for 2 Mi < n <= 4 Mi (16 Ki elements per resident workgroup).  v is parked
// in LDS -- 16 x 1024 doubles, 128 KiB of the CU's 160 -- and xk + sj stays in registers, so the vectors are read ONCE and y is
// written once: the algorithmic 32 B per element, where the sample-predicted pipeline pays six launches (~50 us of fixed cost:
// 77 us per call at n = 4e6).  Sixteen elements per lane in REGISTERS spill (the 128 VGPRs of a 1024-lane workgroup hold v and
// xk + sj, but not what the unrolled digit code of 16 elements wants on top: 492 B of scratch per lane); from LDS the digit
// loop stays rolled.  Passes, rendezvous and scan are those of coop_select.
// ---------------------------------------------------------------------------------------------
constexpr int kLdsEpl = 16;      // Float64: elements per lane (128 KiB of LDS per 1024-lane workgroup)
constexpr int kLdsEpl32 = 32;    // Float32
constexpr int64_t kLdsMinN = (int64_t)1 << 20;  // k_sel_lds serves the sizes above this (and below what the resident grid holds: run_select)
template <class T> struct SelVec;   // a 16-byte vector of T
template <> struct SelVec<double> { typedef f64x2 type; };
template <> struct SelVec<float> { typedef float type __attribute__((ext_vector_type(4))); };
// VEC: all four vectors 16-byte aligned -- a lane owns 16-byte VECTORS of elements (vector gtid + k NT: pairs of doubles, quads
// of floats) and moves them with 16-byte accesses; otherwise elements gtid + k NT, one at a time.
// T = float (round 3): 32 elements per lane, 8 Mi on 256 CUs -- 16 B per element in one launch where the form that parks v in
// y moves ~44; the first digit is folded as for Float64 (fold_digit_f32).
// REGX (round 4, third session: the cliff above 2^22 elements -- the sample-predicted pipeline took over there with six launches
// and ~50 us of fixed cost: n = 6e6 86 us against 44 at 4e6): REGX MORE elements per lane whose v stays in REGISTERS through the
// digit passes (unrolled visits, as in k_sel_coop<REG>), beside the kSlots elements in LDS: 16 + 8 = 24 Ki elements per
// workgroup, 6 Mi on 256 CUs.  The registers come from xk + sj, which this form does not keep (32 of them for 16 elements): the
// storing phase re-reads xk and sj -- 16 B per element more, from the memory-side cache the load phase has just filled -- and
// forms xk + sj again (the same addition: the same bits).  48 B per element moved instead of 32, in one launch.
constexpr int kLdsRegX = 8;
template <bool BINF, bool VEC, class T = double, int REGX = 0>
__global__ __launch_bounds__(1024) void k_sel_lds(T* y, const T* q, const T* xk, const T* sj, int64_t n,
                                                   int64_t r, T delta, SelSync* ss, int parity, int use_set, int clear_set) {
  constexpr bool kF64 = std::is_same<T, double>::value;
  constexpr int kSlots = kF64 ? kLdsEpl : kLdsEpl32;   // elements per lane in LDS
  constexpr int kAll = kSlots + REGX;                  // elements per lane
  constexpr bool kKeepXs = REGX == 0;                  // xk + sj stays in registers (REGX: re-read in the storing phase)
  constexpr int W = 16 / (int)sizeof(T);               // elements per 16-byte vector
  static_assert(REGX % (2 * W) == 0, "REGX: whole load batches");
  typedef typename SelVec<T>::type VT;
  __shared__ CoopShared sh;
  __shared__ __attribute__((aligned(16))) T lv[kSlots * 1024];
  __shared__ unsigned long long kmm[3];  // smallest / largest key and number of this workgroup's elements inside the bucket of a pass
  spx_bar_reset(ss->hdr.bar[parity ^ 1]);
  if (clear_set >= 0) {
    const int64_t total = (int64_t)kCoopMaxPass * kBins;
    unsigned long long* z = &ss->chist[clear_set][0][0];
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < total; b += (int64_t)gridDim.x * blockDim.x) z[b] = 0ull;
  }
  unsigned long long (*hist)[kBins] = ss->chist[use_set];
  unsigned long long* const stat = hist[kSelStatSlot];  // bucket statistics of the tracked passes (see kSelStatSlot)
  unsigned int* bar = ss->hdr.bar[parity];
  SpxSyncHeader* hdr = &ss->hdr;
  unsigned int nbar = 0;
  const int t = threadIdx.x;
  const int nt = (int)(gridDim.x * blockDim.x), gtid = (int)(blockIdx.x * blockDim.x) + t, n32 = (int)n;  // (n <= kSlots Ki x gridDim.x)
  // slot s of this lane: element index and where its v sits in LDS
  auto index_of = [&](int s_) -> int { return VEC ? W * (gtid + (s_ / W) * nt) + (s_ % W) : gtid + s_ * nt; };
  auto lds_of = [&](int s_) -> int { return VEC ? W * ((s_ / W) * 1024 + t) + (s_ % W) : s_ * 1024 + t; };
  // first digit of a key: it depends on nothing but the key (the state sel_state_init sets up: the folded digit of a Float64,
  // the plain top digit of a Float32)
  auto digit0 = [&](uint64_t key) -> unsigned int { if constexpr (kF64) return fold_digit(key); else return fold_digit_f32(key); };
  T xs[kKeepXs ? kSlots : 1];
  T vr[REGX > 0 ? REGX : 1];  // v of the slots kSlots .. kAll - 1
  SEL_STAMP(31);
  // The first digit is histogrammed while the loads are in flight -- as its own sweep over the 16 Ki elements it was 8 us of
  // VALU work that nothing overlapped.
  for (int b = t; b < kBins; b += 1024) sh.lh[b] = 0u;
  if (t == 0) {
    sel_state_init(sh.sst, n, r);
    if (sh.sst.phase == 0 && kCoopFold) sh.sst.pad = kF64 ? 1 : 2;  // first digit: fold_digit / fold_digit_f32 (binades)
  }
  __syncthreads();
  const bool fused = kCoopFold && sh.sst.phase == 0;
  constexpr int kBatch = 2 * W;  // elements per batch: two 16-byte vectors of each input (six 16-byte or 3 x 2W narrow loads in flight)
#pragma unroll
  for (int s0 = 0; s0 < kAll; s0 += kBatch) {
    T vv[kBatch];
    T xb[kBatch];  // xk + sj of the batch
    // (small n: nothing of this batch belongs to the workgroup -- its first lane's first index is past the end)
    if ((VEC ? W * ((int)(blockIdx.x * blockDim.x) + (s0 / W) * nt) : (int)(blockIdx.x * blockDim.x) + s0 * nt) >= n32) {
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        if constexpr (kKeepXs) xs[s0 + k] = (T)0;
        if (s0 + k >= kSlots) vr[(s0 + k >= kSlots) ? s0 + k - kSlots : 0] = (T)0;
      }
      continue;
    }
    if constexpr (VEC) {
      const int nw = n32 / W, tail = n32 % W;  // whole vectors; elements behind the last whole one
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int s_ = s0 + W * h;
        const int pr = gtid + (s_ / W) * nt;
        const int pc = pr < nw ? pr : nw - 1;  // clamped, unconditional (n >= W here: n > 2^20 or a capped grid)
        const VT xv = __builtin_nontemporal_load(reinterpret_cast<const VT*>(xk) + pc);
        const VT sv = __builtin_nontemporal_load(reinterpret_cast<const VT*>(sj) + pc);
        const VT qv = __builtin_nontemporal_load(reinterpret_cast<const VT*>(q) + pc);
#pragma unroll
        for (int e = 0; e < W; ++e) {
          xb[W * h + e] = xv[e] + sv[e];
          vv[W * h + e] = xb[W * h + e] + qv[e];   // shiftedIndBallL0.jl:66
        }
        if (tail && pr == nw) {  // the elements behind the last whole vector: the first `tail` slots of the vector after it
#pragma unroll
          for (int e = 0; e < W - 1; ++e) {
            if (e < tail) {
              const int i = W * nw + e;
              xb[W * h + e] = xk[i] + sj[i];
              vv[W * h + e] = xb[W * h + e] + q[i];
            }
          }
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < kBatch; ++k) {
        const int i = index_of(s0 + k);
        const int ic = i < n32 ? i : n32 - 1;  // clamped, unconditional
        const T xv = __builtin_nontemporal_load(xk + ic), sv = __builtin_nontemporal_load(sj + ic), qv = __builtin_nontemporal_load(q + ic);
        xb[k] = xv + sv;
        vv[k] = xb[k] + qv;                    // shiftedIndBallL0.jl:66
      }
    }
#pragma unroll
    for (int k = 0; k < kBatch; ++k) {
      if constexpr (kKeepXs) xs[s0 + k] = xb[k];
      if (s0 + k < kSlots) lv[lds_of(s0 + k < kSlots ? s0 + k : 0)] = vv[k];   // (slots beyond n are never read)
      else vr[(s0 + k >= kSlots) ? s0 + k - kSlots : 0] = vv[k];
      if (fused && index_of(s0 + k) < n32) atomicAdd(&sh.lh[digit0(key_of(vv[k]))], 1u);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  SEL_STAMP(32);
  unsigned long long cnt_before = ~0ull;  // elements inside the bucket of the previous pass
  for (int p = 0; p < kCoopMaxPass; ++p) {
    __syncthreads();
    const SelState st = sel_uniform(sh.sst);
    if (st.phase == 2) break;  // the same in every workgroup: they all computed it from the same histograms
    // elements inside the bucket of THIS pass = the count of the bin the previous scan selected (coop_scan_step: found[2])
    const unsigned long long cnt_in = p > 0 ? sh.scratch[18] : ~0ull - 1ull;
    const int hs = st.shift + st.width;
    const uint64_t dmask = ((uint64_t)1 << st.width) - 1;
    // A bucket that holds ONE key (lattice data: the usual state after the first digit) needs no further key digits: the
    // passes after the first also track the smallest and largest key inside their bucket (as k_s2_tail's candidate select)
    // -- but only once a pass has failed to split its bucket at all (every element of it in one bin): on generic data the
    // bookkeeping cost 5 us per call (n = 4e6: 42 -> 48 us) for nothing; a lattice now pays one key pass more than it must.
    const bool track = p > 1 && p < kSelStatSlot && st.phase == 0 && st.pad == 0 && cnt_in == cnt_before;
    cnt_before = cnt_in;
    unsigned long long kmin = ~0ull, kmax = 0ull, kcnt = 0ull;
    if (!(p == 0 && fused)) {
      for (int b = t; b < kBins; b += 1024) sh.lh[b] = 0u;
      if (t == 0) { kmm[0] = ~0ull; kmm[1] = 0ull; kmm[2] = 0ull; }
      __syncthreads();
      // (the sweep exists twice, with and without the bucket statistics: as a run-time test per element they cost the
      //  generic path 2 us per call)
      auto sweep = [&](auto track_tag) {
        constexpr bool kTrack = decltype(track_tag)::value;
      auto visit = [&](int i, T vval) {
        if (i >= n32) return;
        const uint64_t key = key_of(vval);
        bool in = true;
        unsigned int dg;
        if (st.phase == 0) {
          if (st.pad == 1) {
            dg = fold_digit(key);
          } else if (st.pad == 2) {
            dg = fold_digit_f32(key);
          } else {
            const KeyPos kp = key_pos(key, st.base, st.shift, st.width, st.clamp);
            in = kp.in;
            dg = kp.digit;
            if (kTrack && in) { kmin = key < kmin ? key : kmin; kmax = key > kmax ? key : kmax; ++kcnt; }
          }
        } else {
          in = key == st.t_eq && (((uint64_t)i) >> hs) == st.prefix;
          dg = (unsigned int)((((uint64_t)i) >> st.shift) & dmask);
        }
        // (plain LDS atomics: the wave-aggregated form of coop_select cost this rolled loop 3 us per full sweep on generic data;
        //  on one-key data 64 lanes on one LDS address are ~64 clocks per instruction, 7 us per sweep)
        if (in) atomicAdd(&sh.lh[dg], 1u);
      };
#pragma unroll 2
      for (int s_ = 0; s_ < kSlots; ++s_) visit(index_of(s_), lv[lds_of(s_)]);
      if constexpr (REGX > 0) {
#pragma unroll
        for (int j = 0; j < REGX; ++j) {
          T vj = vr[j];
          asm volatile("" : "+v"(vj));  // (opaque per sweep: otherwise the keys and first digits of the eight are hoisted out of the pass loop and spill)
          visit(index_of(kSlots + j), vj);
        }
      }
      };
      if (track) sweep(std::true_type{}); else sweep(std::false_type{});
      if (track) {  // wave -> workgroup (LDS) -> one global atomic of each kind per workgroup (256 per address: ~3 us, under the rendezvous)
        for (int off = 32; off >= 1; off >>= 1) {
          const unsigned long long a_ = __shfl_xor(kmin, off, 64), c_ = __shfl_xor(kmax, off, 64);
          kmin = a_ < kmin ? a_ : kmin;
          kmax = c_ > kmax ? c_ : kmax;
          kcnt += __shfl_xor(kcnt, off, 64);
        }
        if ((t & 63) == 0 && kcnt) { atomicMin(&kmm[0], kmin); atomicMax(&kmm[1], kmax); atomicAdd(&kmm[2], kcnt); }
      }
    }
    if (p < 4) SEL_STAMP(23 + p);
    __syncthreads();
    if (track && t == 0 && kmm[2]) { atomicMax(&stat[2 * p], ~kmm[0]); atomicMax(&stat[2 * p + 1], kmm[1]); }
    for (int b = t; b < kBins; b += 1024) {
      const unsigned int c = sh.lh[b];
      if (c) atomicAdd(&hist[p][b], (unsigned long long)c);
    }
    // the histogram atomics of every wave have been performed (vmcnt) before its workgroup arrives; nothing else is exchanged
    SEL_STAMP(33 + 3 * p);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (p < 4) SEL_STAMP(27 + p);
    spx_grid_rendezvous(bar, (++nbar) * gridDim.x, hdr);
    SEL_STAMP(34 + 3 * p);
    coop_scan_step(hist[p], st, &sh.sst, sh.scratch);
    if (track) sel_single_key_shortcut(st, &sh.sst, &stat[2 * p], &stat[2 * p + 1], nullptr, cnt_in, true);
    SEL_STAMP(35 + 3 * p);
  }
  __syncthreads();
  const SelState fin = sel_uniform(sh.sst);
  const bool poisoned = spx_poisoned(hdr);  // (see coop_select)
  auto P = [&](T val) -> T { return poisoned ? (T)__longlong_as_double(0x7ff8000000000000ll) : val; };
  if constexpr (VEC && !kKeepXs) {
    // REGX: xk + sj is formed again from re-read vectors, four 16-byte vectors of each at a time (eight loads in flight per lane)
    const int nw = n32 / W, tail = n32 % W;
    constexpr int kVB = 4;
    static_assert((kAll / W) % kVB == 0, "whole batches of vectors");
#pragma unroll
    for (int v0 = 0; v0 < kAll / W; v0 += kVB) {
      VT xv[kVB], sv[kVB];
#pragma unroll
      for (int b = 0; b < kVB; ++b) {
        const int pr = gtid + (v0 + b) * nt;
        const int pc = pr < nw ? pr : nw - 1;
        xv[b] = __builtin_nontemporal_load(reinterpret_cast<const VT*>(xk) + pc);
        sv[b] = __builtin_nontemporal_load(reinterpret_cast<const VT*>(sj) + pc);
      }
#pragma unroll
      for (int b = 0; b < kVB; ++b) {
        const int s_ = (v0 + b) * W;
        const int pr = gtid + (v0 + b) * nt;
        VT vvv;
        if (s_ < kSlots) {
          vvv = *reinterpret_cast<const VT*>(&lv[lds_of(s_ < kSlots ? s_ : 0)]);
        } else {
#pragma unroll
          for (int e = 0; e < W; ++e) vvv[e] = vr[(s_ >= kSlots) ? s_ - kSlots + e : 0];
        }
        VT o;
#pragma unroll
        for (int e = 0; e < W; ++e) {
          T xse = xv[b][e] + sv[b][e];
          if (tail && pr == nw && e < tail) xse = xk[W * nw + e] + sj[W * nw + e];  // (the elements behind the last whole vector)
          o[e] = P(sel_out_xs<BINF>((T)vvv[e], (int64_t)(W * pr + e), xse, fin, delta));
        }
        if (pr < nw) {
          __builtin_nontemporal_store(o, reinterpret_cast<VT*>(y) + pr);
        } else if (tail && pr == nw) {
#pragma unroll
          for (int e = 0; e < W - 1; ++e)
            if (e < tail) y[W * nw + e] = o[e];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  } else if constexpr (VEC) {
    const int nw = n32 / W, tail = n32 % W;
#pragma unroll
    for (int s_ = 0; s_ < kSlots; s_ += W) {
      const int pr = gtid + (s_ / W) * nt;
      const VT vvv = *reinterpret_cast<const VT*>(&lv[lds_of(s_)]);
      VT o;
#pragma unroll
      for (int e = 0; e < W; ++e) o[e] = P(sel_out_xs<BINF>((T)vvv[e], (int64_t)(W * pr + e), xs[s_ + e], fin, delta));
      if (pr < nw) {
        __builtin_nontemporal_store(o, reinterpret_cast<VT*>(y) + pr);
      } else if (tail && pr == nw) {
#pragma unroll
        for (int e = 0; e < W - 1; ++e)
          if (e < tail) y[W * nw + e] = o[e];
      }
      __builtin_amdgcn_sched_barrier(0);  // (one vector at a time: hoisted, the LDS reads do not fit beside xs)
    }
  } else {
    static_assert(VEC || kKeepXs, "the form with register slots exists for 16-byte aligned vectors only");
#pragma unroll
    for (int s_ = 0; s_ < kSlots; ++s_) {
      const int i = index_of(s_);
      if (i < n32) __builtin_nontemporal_store(P(sel_out_xs<BINF>(lv[lds_of(s_)], (int64_t)i, xs[s_], fin, delta)), y + i);
    }
  }
  SEL_STAMP(63);
}

// ---------------------------------------------------------------------------------------------
// k_s2_tail: the last launch of a sample-predicted call, queued unconditionally (nothing is read back): as many 1024-lane
// workgroups as are resident at once.  Returns at once when the candidate kernels have finished the job (generic data: ~5 us).
// Otherwise, with its whole grid and in-launch rendezvous:
//   kTodoCandSelect  the first candidate digit left more survivors than the one-workgroup short list holds (a key shared by
//                    thousands of elements): the radix select continues over the candidate regions -- the loop of
//                    coop_select, fed from the records instead of the vector (12 us per digit at n = 1e8)
//   kTodoTieScan     tie mode, the cut lies inside a class: prefix sum over the per-wave member counts (index order) ->
//                    the wave that holds the quota-th member -> its 768 elements are re-read -> icut
//   kTodoFinal       tie mode: y = sel_out(v) for every element, one streaming pass (24 B read + 8 B written per element)
//   verdict negative the exact select over the whole vector (coop_select, v parked in y)
// ---------------------------------------------------------------------------------------------
// exclusive prefix sum over the 1024 lanes of a workgroup; *total = the sum.  lds = 17 words.
__device__ __forceinline__ unsigned long long scan1024_exclusive(unsigned long long v, unsigned long long* total, unsigned long long* lds) {
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  unsigned long long inc = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const unsigned long long up = __shfl_up(inc, off, 64);
    if (lane >= off) inc += up;
  }
  __syncthreads();  // (lds may still be read by the previous call's lanes)
  if (lane == 63) lds[w] = inc;
  __syncthreads();
  unsigned long long base = 0, all = 0;
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const unsigned long long x = lds[k];
    base += (k < w) ? x : 0ull;
    all += x;
  }
  *total = all;
  return base + inc - v;
}

template <bool BINF>
__global__ __launch_bounds__(1024) void k_s2_tail(double* y, const double* q, const double* xk, const double* sj, int64_t n,
                                                   int64_t r, double delta, SelSync* ss, int parity, const Cand* cand,
                                                   const WaveCount* counts, int64_t nregions, unsigned int ovf_cap,
                                                   const ClassCount* cls, int ioff, int write_flags) {
  const int write = write_flags & 1;
  __shared__ CoopShared sh;
  __shared__ unsigned long long tl[20];
  __shared__ unsigned long long tparts[256];
  const int t = threadIdx.x, G = (int)gridDim.x, b = (int)blockIdx.x;
  const int64_t gt = (int64_t)b * blockDim.x + t, nt = (int64_t)G * blockDim.x;
  SpxSyncHeader* hdr = &ss->hdr;
  unsigned int* bar = hdr->bar[parity];
  unsigned int nbar = 0;
  spx_bar_reset(hdr->bar[parity ^ 1]);
  {  // clean slates for the next call's k_s2_front
    unsigned long long* z1 = ss->fhist1;
    unsigned long long* z2 = &ss->fhist2[0][0][0];
    for (int64_t k = gt; k < kBins; k += nt) z1[k] = 0ull;
    for (int64_t k = gt; k < 10 * kBins; k += nt) z2[k] = 0ull;
    if (gt < 8) ss->ftie[gt] = 0ull;
    if (gt == 0) ss->ws.fs.smax = 0ull;
  }
  // written by earlier launches: the same values in every workgroup
  const int ok = ss->ws.fs.ok, todo = ss->ws.fs.todo, tie = ss->ws.fs.tie;
  if (ok && todo == 0) return;
  bool late = false;
#ifdef SPX_TEST_HOOKS
  // planted (key 101): this workgroup behaves as one that was not resident while the others waited for it -- it takes part in
  // nothing they synchronise on and only runs once they have given up (the flag), like a workgroup placed after they left
  if ((write_flags & 2) && G > 1 && b == G - 1) {
    late = true;
    for (unsigned int spins = 0; !spx_poisoned(hdr) && spins < (1u << 22); ++spins) __builtin_amdgcn_s_sleep(20);
  }
#endif
  if (t == 0) { sh.sst = ss->ws.st; tl[17] = ~0ull; tl[18] = 0ull; tl[19] = ~0ull; }
  if (!late) {
  const int64_t total_hist = (int64_t)kCoopMaxPass * kBins;
  if (!ok || (todo & kTodoCandSelect)) {  // histogram set 2 belongs to this launch: cleared here, one barrier
    unsigned long long* z = &ss->chist[2][0][0];
    for (int64_t k = gt; k < total_hist; k += nt) z[k] = 0ull;
    if (gt < kCoopMaxPass) { ss->cmin[gt] = ~0ull; ss->cmax[gt] = 0ull; ss->ccnt[gt] = 0ull; }
    spx_grid_barrier(bar, (++nbar) * G, hdr);
  }
  if (!ok) {  // prediction not verified: exact select over the whole vector (recomputes everything from q, xk, sj)
    coop_select<BINF, false>(y, q, xk, sj, n, r, delta, ss->chist[2], bar, nbar, sh, hdr);
    return;
  }
  const unsigned int novf = ss->ws.fs.ovf_count < ovf_cap ? ss->ws.fs.ovf_count : ovf_cap;
  if (todo & kTodoCandSelect) {
    int p = 0;
    for (; p < kCoopMaxPass; ++p) {
      __syncthreads();
      const SelState st = sh.sst;
      if (st.phase == 2) break;  // the same in every workgroup: they all computed it from the same histograms
      for (int k = t; k < kBins; k += 1024) sh.lh[k] = 0u;
      __syncthreads();
      const int hs = st.shift + st.width;
      const uint64_t dmask = ((uint64_t)1 << st.width) - 1;
      unsigned long long kmin = ~0ull, kmax = 0ull, kcnt = 0ull;  // (phase 0) keys of this lane's candidates inside the bucket
      for_each_candidate(counts, nregions, cand, (int64_t)novf, [&](uint64_t key, int64_t i, double) {
        bool in;
        unsigned int dg;
        if (st.phase == 0) {
          const KeyPos kp = key_pos(key, st.base, st.shift, st.width, st.clamp);
          in = kp.in;
          dg = kp.digit;
          if (in) { kmin = key < kmin ? key : kmin; kmax = key > kmax ? key : kmax; ++kcnt; }
        } else {
          in = key == st.t_eq && (((uint64_t)i) >> hs) == st.prefix;
          dg = (unsigned int)((((uint64_t)i) >> st.shift) & dmask);
        }
        hist_add_agg(sh.lh, dg, in);
      });
      if (st.phase == 0) {  // one global atomic of each kind per wavefront that saw a candidate of the bucket
        for (int off = 32; off >= 1; off >>= 1) {
          const unsigned long long a = __shfl_xor(kmin, off, 64), c = __shfl_xor(kmax, off, 64);
          kmin = a < kmin ? a : kmin;
          kmax = c > kmax ? c : kmax;
          kcnt += __shfl_xor(kcnt, off, 64);
        }
        if ((t & 63) == 0 && kcnt) {
          atomicMin(&ss->cmin[p], kmin);
          atomicMax(&ss->cmax[p], kmax);
          atomicAdd(&ss->ccnt[p], kcnt);
        }
      }
      __syncthreads();
      for (int k = t; k < kBins; k += 1024) {
        const unsigned int c = sh.lh[k];
        if (c) atomicAdd(&ss->chist[2][p][k], (unsigned long long)c);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      spx_grid_rendezvous(bar, (++nbar) * G, hdr);
      coop_scan_step(ss->chist[2][p], st, &sh.sst, sh.scratch);
      if (st.phase == 0) sel_single_key_shortcut(st, &sh.sst, &ss->cmin[p], &ss->cmax[p], &ss->ccnt[p]);
    }
    __syncthreads();
    const SelState fin = sh.sst;
    if (b == 0 && t == 0) ss->ws.st = fin;  // (every workgroup read the old state before the first rendezvous above)
    if (write && !spx_poisoned(hdr)) {
      // every candidate the main pass guessed wrong (FastState::t_mid; what k_s2_compact settled already is stored again: harmless)
      const uint64_t t_mid = ss->ws.fs.t_mid;
      for_each_candidate(counts, nregions, cand, (int64_t)novf, [&](uint64_t key, int64_t i, double val) {
        const bool keep = (key >= fin.t_ge) || (key == fin.t_eq && i <= fin.icut);
        if (keep != (key >= t_mid)) y[i] = keep ? val : sel_dropped<BINF>(xk, sj, i, delta);
      });
    }
  }
  if (todo & kTodoTieScan) {
    // members of the class (key == t_eq) per slot, slots in index order (ClassCount); keep the first `quota` of them
    const int which = ss->ws.fs.tie_class;
    const int64_t nslots = nregions + 2;
    const int64_t S = (nslots + G - 1) / G, lo = (int64_t)b * S, hi = (lo + S < nslots) ? lo + S : nslots;
    auto members = [&](int64_t w) -> unsigned long long { return which ? (unsigned long long)cls[w].lo : (unsigned long long)cls[w].hi; };
    unsigned long long mine = 0;
    for (int64_t w = lo + t; w < hi; w += 1024) mine += members(w);
    unsigned long long tot;
    (void)scan1024_exclusive(mine, &tot, tl);
    if (t == 0) __hip_atomic_store(&ss->tie_part[b], tot + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t < G) {
      unsigned long long w0;
      unsigned int spins = 0;
      for (;;) {
        w0 = __hip_atomic_load(&ss->tie_part[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w0) break;
        if (spx_wait_expired(spins, hdr)) { w0 = 1ull; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      tparts[t] = w0 - 1ull;
    }
    __syncthreads();
    const unsigned long long quota = (unsigned long long)sh.sst.quota;
    unsigned long long dummy;
    const unsigned long long before = scan1024_exclusive(t < G ? tparts[t] : 0ull, &dummy, tl);
    if (t < G && before < quota && quota <= before + tparts[t]) { tl[17] = (unsigned long long)t; tl[18] = quota - before; }
    __syncthreads();
    if (b == (int)tl[17]) {  // this slice holds the quota-th member: which slot, which element
      const unsigned long long qrem = tl[18];
      unsigned long long running = 0;
      __syncthreads();
      if (t == 0) tl[19] = ~0ull;
      for (int64_t base = lo; base < hi; base += 1024) {
        const int64_t w = base + t;
        const unsigned long long c = (w < hi) ? members(w) : 0ull;
        unsigned long long chunk;
        const unsigned long long ex = scan1024_exclusive(c, &chunk, tl);
        if (c && running + ex < qrem && qrem <= running + ex + c) { tl[19] = (unsigned long long)w; tl[18] = qrem - running - ex; }
        running += chunk;
        __syncthreads();
        if (tl[19] != ~0ull) break;
      }
      const int64_t wstar = (int64_t)tl[19];
      const unsigned long long rho = tl[18];  // the rho-th member of slot wstar, in index order
      const uint64_t t_eq = sh.sst.t_eq;
      // (q, xk, sj are the caller's pointers; the aligned rest starts at ioff; the last slot is the odd last element of the rest)
      const int64_t nrest = n - ioff, npairs2 = (nrest >> 1) << 1;
      unsigned long long icut = ~0ull;
      if (wstar == 0) {
        icut = 0ull;
      } else if (wstar == nslots - 1) {
        icut = (unsigned long long)(n - 1);
      } else if (wstar > 0) {
        const int64_t e = (wstar - 1) * (64 * kMainUnroll * 2) + t;  // element of the aligned rest
        const bool valid = t < 64 * kMainUnroll * 2 && e < npairs2;
        bool match = false;
        if (valid) {
          const int64_t i = e + ioff;
          match = key_of((xk[i] + sj[i]) + q[i]) == t_eq;
        }
        unsigned long long cnt;
        const unsigned long long ex = scan1024_exclusive(match ? 1ull : 0ull, &cnt, tl);
        if (match && ex + 1ull == rho) tl[19] = (unsigned long long)(e + ioff);
        __syncthreads();
        icut = tl[19];
      }
      if (t == 0) __hip_atomic_store(&ss->tie_cut, icut + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (t == 0) {
      unsigned long long w0;
      unsigned int spins = 0;
      for (;;) {
        w0 = __hip_atomic_load(&ss->tie_cut, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w0) break;
        if (spx_wait_expired(spins, hdr)) { w0 = 1ull; break; }
        __builtin_amdgcn_s_sleep(1);
      }
      sh.sst.icut = (int64_t)(w0 - 1ull);
    }
    // everybody has read the exchange words: the first workgroup clears them for the next call and publishes the state
    spx_grid_rendezvous(bar, (++nbar) * G, hdr);
    if (b == 0) {
      if (t < 256) ss->tie_part[t] = 0ull;
      if (t == 0) { ss->tie_cut = 0ull; ss->ws.st = sh.sst; }
    }
  }
  }  // (!late)
  __syncthreads();
  // A workgroup that gave up waiting for the others has made the counts / the cut garbage (spx_wait_expired: the candidate
  // select's histograms, the tie scan's made-up icut).  Never a plausible wrong result (include/spx.h): the single-pass form has
  // speculative values in y already -- NaN over the whole vector, by every workgroup that sees the flag (the kTodoFinal rewrite
  // below does that itself); the two-pass form has not stored anything yet -- FastState::ok = 2 makes k_sel_final_q store NaN.
  if (spx_poisoned(hdr) && !((todo & kTodoFinal) && write)) {
    if (write) {
      for (int64_t i = gt; i < n; i += nt) y[i] = __longlong_as_double(0x7ff8000000000000ll);
    } else if (t == 0) {
      ss->ws.fs.ok = 2;
    }
    return;
  }
  if ((todo & kTodoFinal) && write) {
    // The main pass stored the class members on the sample's guess of the cut (FastState::spec_*); now the cut is known.
    // Both rules are "kept up to an index c" per class (c = -1: none, n - 1: all): the guess was wrong on (min(c_spec,
    // c_true), max(..)] -- that index range is rewritten from q, xk, sj and the final thresholds, every element of it (the
    // candidates in it get the value the candidate kernels gave them).  Evenly spread ties: ~1 % of the vector.
    const SelState fin = sh.sst;
    const bool poisoned = spx_poisoned(hdr);
    auto P = [&](double val) -> double { return poisoned ? __longlong_as_double(0x7ff8000000000000ll) : val; };
    int64_t flo = n, fhi = -1;  // caller's indices, inclusive
    auto mismatch = [&](int has, uint64_t key, int spec) {
      if (!has) return;
      const int64_t ca = key >= fin.t_ge ? n - 1 : (key == fin.t_eq ? fin.icut : (int64_t)-1);
      int64_t cs = spec == 1 ? n - 1 : (spec == 2 ? (int64_t)ss->ws.fs.spec_cut : (int64_t)-1);
      cs = cs < -1 ? -1 : (cs > n - 1 ? n - 1 : cs);
      if (ca == cs) return;
      const int64_t lo = (ca < cs ? ca : cs) + 1, hi = ca < cs ? cs : ca;
      flo = lo < flo ? lo : flo;
      fhi = hi > fhi ? hi : fhi;
    };
    mismatch(ss->ws.fs.has_hi, ss->ws.fs.t_hi, ss->ws.fs.spec_hi);
    mismatch(ss->ws.fs.has_lo, ss->ws.fs.t_lo, ss->ws.fs.spec_lo);
    if (poisoned) { flo = 0; fhi = n - 1; }
    if (fhi >= flo) {
      const double* qa = q + ioff; const double* xa = xk + ioff; const double* sa = sj + ioff; double* ya = y + ioff;
      const int64_t nrest = n - ioff, n2 = nrest >> 1;
      // pairs of the aligned rest that overlap [flo, fhi]: elements 2p + ioff, 2p + 1 + ioff
      const int64_t e0 = flo - ioff < 0 ? 0 : flo - ioff, e1 = fhi - ioff;
      const int64_t plo = e0 >> 1, phi = (e1 >> 1) < n2 - 1 ? (e1 >> 1) : n2 - 1;
      const bool vec2 = ((reinterpret_cast<uintptr_t>(ya) | reinterpret_cast<uintptr_t>(qa) | reinterpret_cast<uintptr_t>(xa) |
                          reinterpret_cast<uintptr_t>(sa)) & 15u) == 0;
      if (vec2 && e1 >= 0 && phi >= plo) {
        const f64x2* q2 = reinterpret_cast<const f64x2*>(qa);
        const f64x2* x2 = reinterpret_cast<const f64x2*>(xa);
        const f64x2* s2 = reinterpret_cast<const f64x2*>(sa);
        f64x2* y2 = reinterpret_cast<f64x2*>(ya);
        constexpr int KP = 4;
        const int64_t np = phi - plo + 1;
        const int64_t ntiles = (np + 1024 * KP - 1) / (1024 * KP);
        for (int64_t tile = b; tile < ntiles; tile += G) {
          f64x2 a[KP], bb[KP], c[KP];
#pragma unroll
          for (int k = 0; k < KP; ++k) {
            int64_t i = plo + tile * (1024 * KP) + k * 1024 + t;
            if (i > phi) i = phi;
            a[k] = __builtin_nontemporal_load(q2 + i);
            bb[k] = __builtin_nontemporal_load(x2 + i);
            c[k] = __builtin_nontemporal_load(s2 + i);
          }
#pragma unroll
          for (int k = 0; k < KP; ++k) {
            const int64_t i = plo + tile * (1024 * KP) + k * 1024 + t;
            if (i <= phi) {
              f64x2 o;
              o.x = P(sel_out<BINF>((bb[k].x + c[k].x) + a[k].x, 2 * i + ioff, bb[k].x, c[k].x, fin, delta));
              o.y = P(sel_out<BINF>((bb[k].y + c[k].y) + a[k].y, 2 * i + 1 + ioff, bb[k].y, c[k].y, fin, delta));
              __builtin_nontemporal_store(o, y2 + i);
            }
          }
        }
      } else if (!vec2) {  // (the pipeline only runs on aligned vectors; kept for safety)
        for (int64_t i = flo + gt; i <= fhi; i += nt) y[i] = P(sel_out<BINF>((xk[i] + sj[i]) + q[i], i, xk[i], sj[i], fin, delta));
      }
      if (gt == 0) {  // the stragglers wave 0 of the main pass took along
        if ((nrest & 1) && fhi == n - 1) { const int64_t i = n - 1; y[i] = P(sel_out<BINF>((xk[i] + sj[i]) + q[i], i, xk[i], sj[i], fin, delta)); }
        if (ioff && flo == 0) y[0] = P(sel_out<BINF>((xk[0] + sj[0]) + q[0], 0, xk[0], sj[0], fin, delta));
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// k_s2_front: kFrontBlocks workgroups x 1024 lanes (few, fat workgroups: a grid barrier costs ~2 us with 64 arrivers,
// ~7 us with 256 -- tools/exp/grid_barrier.hip), one sample per lane, kept in a register through all phases.
// (The fenced barrier stays here: the fence-free rendezvous of coop_select needs the histograms read with agent-scope atomic
//  loads, and 16 of those per lane cost the scan more (3.6 -> 6.3 us) than the rendezvous saves (4.1 -> 3.1 us); measured.)
//   A  sample, LDS histogram of the top key digit -> fhist1; clears what the main pass / tail accumulate into
//   B  (every workgroup) scan fhist1 for the two ranks; second digit of the own sample if it sits in a selected bucket
//   C  scan; a third digit only if a selected bucket is still crowded (as k_s2_pick); workgroup 0 writes the band and
//      the digit machinery of the candidate selection (the tail of k_s2_pick, unchanged)
// ---------------------------------------------------------------------------------------------
// kFrontSpl samples per lane: the band's margin is 6 sigma of the SAMPLE rank, so four times the sample halves the candidates
// of the main pass (n = 1e8: r = n/2 670 -> 619 us, r = n/100 575 -> 568 us) for ~2 us more here -- and costs more than it
// saves while the sample is a sizeable part of the vector (n = 3e6: 63 -> 70 us): 1 below 2^23, 2 below 2^25, 4 from there on
// (tools/r2/topr_big.py).
constexpr int kFrontBlocks = 64;
constexpr int kFrontMaxDigits = 6;      // the sample ranks are bracketed to 2-3 x 12 key bits; to all 64 in tie mode
constexpr unsigned int kPickFine = 16;  // samples per selected bucket below which no further digit is resolved
template <int kFrontSpl>
__global__ __launch_bounds__(1024) void k_s2_front(const double* q, const double* xk, const double* sj, int64_t n, int64_t r,
                                                    SelSync* ss, int parity) {
  __shared__ unsigned int lh[kBins];
  __shared__ unsigned int lh1[kBins];        // (digits 2..: the second rank's histogram)
  __shared__ unsigned long long part[4][4];  // per 256-lane group
  __shared__ unsigned long long pre[2];
  __shared__ long long quo[2];
  __shared__ long long rank0[2];   // the two sample ranks as first computed (quo is consumed by the scans)
  __shared__ int active[2];
  __shared__ unsigned int bucket[2];
  const int t = threadIdx.x, c = blockIdx.x;
  unsigned int* bar = ss->hdr.bar[parity];
  unsigned int nbar = 0;
  SelWs* ws = &ss->ws;
  spx_bar_reset(ss->hdr.bar[parity ^ 1]);
  SEL_STAMP(0);
  for (int b = t; b < kBins; b += 1024) lh[b] = 0u;
  // clean slates for the main pass (it runs in a later launch)
  for (int b = c * 1024 + t; b < kBins; b += 1024 * (int)gridDim.x) ws->hist[b] = 0ull;
  for (int b = c * 1024 + t; b < kShards; b += 1024 * (int)gridDim.x) {
    ws->shard_above[b * kShardStride] = 0ull; ws->shard_cand[b * kShardStride] = 0ull;
    ws->shard_hi[b * kShardStride] = 0ull; ws->shard_lo[b * kShardStride] = 0ull;
  }
  if (t == 0) {
    bucket[0] = bucket[1] = 0u;
    constexpr int kFrontSample = kFrontBlocks * 1024 * kFrontSpl;
    const double M = (double)kFrontSample;
    const double p = (double)r / (double)n;
    const double k = p * M;
    const double margin = 6.0 * sqrt(M * p * (1.0 - p)) + 16.0;
    const long long rank_hi = (long long)floor(k - margin);  // 1-based from the largest
    const long long rank_lo = (long long)ceil(k + margin);
    pre[0] = pre[1] = 0;
    active[0] = rank_hi >= 1;
    active[1] = rank_lo <= kFrontSample;
    quo[0] = rank_hi;
    quo[1] = rank_lo;
    rank0[0] = rank_hi;
    rank0[1] = rank_lo;
  }
  __syncthreads();
  // 2048 kFrontSpl chunks of 32 consecutive elements (256 bytes: two full lines per vector), spread evenly over the vector:
  // sample s of lane t of workgroup c comes from chunk (s * 64 + c) * 32 + (t >> 5).  (Chunks of 256 elements, the first
  // layout, left 1024 chunks at n = 1e8 -- on SORTED input, where a chunk is 256 nearly equal values, the sample quantiles
  // then moved in steps as large as the band's whole margin and the verdict failed: 2.0 ms per call instead of 0.55.)
  constexpr int kChunkLen = 32;
  constexpr int kChunks = kFrontBlocks * (1024 / kChunkLen) * kFrontSpl;
  uint64_t keys[kFrontSpl];
  unsigned long long m = 0ull;
#pragma unroll
  for (int sidx = 0; sidx < kFrontSpl; ++sidx) {
    const int chunk = (sidx * kFrontBlocks + c) * (1024 / kChunkLen) + (t / kChunkLen);
    const int64_t start = (int64_t)((double)chunk * (double)(n - kChunkLen) / (double)(kChunks - 1));
    const int64_t i = start + (t % kChunkLen);
    keys[sidx] = key_of(fabs((xk[i] + sj[i]) + q[i]));
  }
#pragma unroll
  for (int sidx = 0; sidx < kFrontSpl; ++sidx) {
    // (tried in round 4: hist_add_agg here -- the top digit is sign + exponent, four or five binades hold the sample -- phase A of
    //  k_s2_front<16> 16.0 -> 19.2 us, of <4> 6.1 -> 6.3: the plain LDS atomics are not what the phase waits for)
    atomicAdd(&lh[keys[sidx] >> (64 - kDigitBits)], 1u);
    const unsigned long long fk = keys[sidx] < kInfKey ? keys[sidx] : 0ull;
    m = fk > m ? fk : m;
  }
  {
    // largest finite sample key (see k_s2_sample) -- ONE global atomic per workgroup: the 1024 per-wave atomicMax of
    // k_s2_sample on this one address were ~12 us of serialised traffic, most of that kernel's 14 us
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned long long o = __shfl_xor(m, off, 64);
      m = o > m ? o : m;
    }
    if ((t & 63) == 0) part[(t >> 6) >> 2][(t >> 6) & 3] = m;
  }
  __syncthreads();
  if (t == 0) {
    unsigned long long m = 0;
    for (int k = 0; k < 16; ++k) { const unsigned long long o = part[k >> 2][k & 3]; m = o > m ? o : m; }
    if (m) atomicMax(&ws->fs.smax, m);
  }
  for (int b = t; b < kBins; b += 1024) {
    const unsigned int cnt = lh[b];
    if (cnt) atomicAdd(&ss->fhist1[b], (unsigned long long)cnt);
  }
  SEL_STAMP(1);
  spx_grid_barrier(bar, (++nbar) * gridDim.x, &ss->hdr);
  SEL_STAMP(2);
  // scan of a 4096-bin histogram from the top: lanes 0..255 serve selection 0, lanes 256..511 selection 1 (as k_s2_pick)
  constexpr int PER = kBins / 256;
  auto scan_both = [&](const unsigned long long* h0, const unsigned long long* h1, int width) {
    const int sel = (t >> 8) & 1, tt = t & 255, grp256 = t >> 8;
    const unsigned long long* h = sel ? h1 : h0;
    unsigned long long loc[PER], sum = 0;
    if (t < 512) {
#pragma unroll
      for (int k = 0; k < PER; ++k) { loc[k] = h[kBins - 1 - (tt * PER + k)]; sum += loc[k]; }
    }
    const unsigned long long run0 = scan256_exclusive(sum, tt, part[grp256]);
    unsigned long long npre = 0, nquo = 0;
    unsigned int nb = 0;
    bool hit = false;
    if (t < 512 && active[sel]) {
      unsigned long long run = run0;
      const unsigned long long quota = (unsigned long long)quo[sel];
#pragma unroll
      for (int k = 0; k < PER; ++k) {
        if (run < quota && run + loc[k] >= quota) {
          npre = (pre[sel] << width) | (uint64_t)(kBins - 1 - (tt * PER + k));
          nquo = quota - run;
          nb = (unsigned int)loc[k];
          hit = true;
        }
        run += loc[k];
      }
    }
    __syncthreads();  // every lane has read pre / quo before the hit lanes replace them
    if (hit) { pre[sel] = npre; quo[sel] = (long long)nquo; bucket[sel] = nb; }
    __syncthreads();
  };
  scan_both(ss->fhist1, ss->fhist1, kDigitBits);
  SEL_STAMP(3);
  // Further digits.  2 and 3 as round 2: bits 51..40, then 39..28 only if a selected bucket still holds more than a handful of
  // samples.  4..6 (bits 27..16, 15..4, 3..0): TIE MODE -- a bucket that is still crowded after 36 bits is one key (or keys 2^-24
  // apart) shared by >= 0.1 % of the vector; both ranks are then resolved to the full key (see FastState::tie).
  int bits_done = kDigitBits;  // key bits resolved so far (from the top)
  int tie = 0;
  for (int digit = 1; digit < kFrontMaxDigits; ++digit) {
    // (uniform: same histograms everywhere.  The resolution asked of the band's ends is a share of the VECTOR: 16 samples of
    //  65 536, i.e. 16 kFrontSpl of this sample: the third digit -- a barrier and a scan, 10 us -- is then skipped at r = n/2
    //  as it always was at r = n/100)
    if (digit == 2 && bucket[0] <= kPickFine * kFrontSpl && bucket[1] <= kPickFine * kFrontSpl) break;
    if (digit == 3) {
      tie = (bucket[0] > 4u * kPickFine * kFrontSpl || bucket[1] > 4u * kPickFine * kFrontSpl) ? 1 : 0;
      if (!tie) break;
    }
    const int width = (64 - bits_done) < kDigitBits ? (64 - bits_done) : kDigitBits;
    const int shift = 64 - bits_done - width;
    const int hs = shift + width;
    // both ranks in the same bucket so far (the usual case: the band is narrow): ONE histogram serves both selections -- half
    // the global atomics of this phase (the barrier behind it waits for them: 9 us)
    const bool shared = active[0] && active[1] && pre[0] == pre[1];
    // (aggregated per WORKGROUP in LDS before they reach the global histogram: on tie-heavy data every sample of the selected
    //  bucket carries the same digit -- 262 144 global atomics on one address were 3.3 ms in round 2, one per wavefront still
    //  50 us per digit: atomics on one address retire ~12 ns apart; now 64 per address)
    __syncthreads();
    for (int bb = t; bb < kBins; bb += 1024) { lh[bb] = 0u; lh1[bb] = 0u; }
    __syncthreads();
    // Round 4: the first tie digit (digit 3) also tracks the smallest and the largest key among the samples of each selected
    // bucket.  A crowded bucket after 36 key bits is nearly always ONE key (a lattice, a constant, sparse zeros): equal ends
    // settle the remaining 16 bits without digits 5 and 6 -- two histogram passes, two fenced barriers and two scans, ~17 us of
    // the ~55 us the front kernel takes in tie mode.  The four words (SelSync::ftie) are zero on entry like the histograms (the
    // tail launch of the previous call clears them); the minimum is kept as the maximum of ~key.
    constexpr bool kTieShortcut = kFrontSpl <= 4;  // (k_s2_front<16> sits at its register budget: with the bookkeeping below it spills 46 registers)
    if (kTieShortcut && digit == 3) {  // (a loop of its own: inside the histogram loop below the accumulators cost registers)
      __shared__ unsigned long long tstat[16][4];  // per wavefront: {max key, max ~key} of selection 0, of selection 1
#pragma unroll
      for (int sel = 0; sel < 2; ++sel) {
        uint64_t a = 0ull, b = 0ull;  // max key, max ~key of this lane's samples inside the bucket of selection sel
        if (!(sel == 1 && shared)) {
#pragma unroll
          for (int sidx = 0; sidx < kFrontSpl; ++sidx) {
            const bool in = active[sel] && (keys[sidx] >> hs) == pre[sel];
            a = (in && keys[sidx] > a) ? keys[sidx] : a;
            b = (in && ~keys[sidx] > b) ? ~keys[sidx] : b;
          }
        }
        for (int off = 32; off >= 1; off >>= 1) {
          const uint64_t oa = __shfl_xor(a, off, 64), ob = __shfl_xor(b, off, 64);
          a = oa > a ? oa : a;
          b = ob > b ? ob : b;
        }
        if ((t & 63) == 0) { tstat[t >> 6][2 * sel + 0] = a; tstat[t >> 6][2 * sel + 1] = b; }
      }
      __syncthreads();
      // wavefronts -> workgroup -> ONE global atomicMax per word and workgroup (one per wavefront -- 1024 on each of four words of
      // one cache line -- cost this digit 34 us: atomics on one line retire ~8-12 ns apart)
      if (t < 4) {
        unsigned long long m = 0ull;
        for (int w = 0; w < 16; ++w) m = tstat[w][t] > m ? tstat[w][t] : m;
        if (m) atomicMax(&ss->ftie[t], m);
      }
    }
#pragma unroll
    for (int sidx = 0; sidx < kFrontSpl; ++sidx) {
      const uint64_t top = keys[sidx] >> hs;
      const unsigned d = (unsigned)((keys[sidx] >> shift) & (((uint64_t)1 << width) - 1));
      hist_add_agg(lh, d, active[0] && top == pre[0]);
      if (!shared) hist_add_agg(lh1, d, active[1] && top == pre[1]);
    }
    __syncthreads();
    for (int bb = t; bb < kBins; bb += 1024) {
      const unsigned int c0 = lh[bb], c1 = lh1[bb];
      if (c0) atomicAdd(&ss->fhist2[digit - 1][0][bb], (unsigned long long)c0);
      if (c1) atomicAdd(&ss->fhist2[digit - 1][1][bb], (unsigned long long)c1);
    }
    SEL_STAMP(2 + 2 * digit);
    spx_grid_barrier(bar, (++nbar) * gridDim.x, &ss->hdr);
    SEL_STAMP(3 + 2 * digit);
    scan_both(ss->fhist2[digit - 1][0], ss->fhist2[digit - 1][shared ? 0 : 1], width);
    bits_done += width;
    if (kTieShortcut && digit == 3) {
      // (the same words in every workgroup, read behind the fenced barrier: the same decision everywhere)
      bool single = true;
      uint64_t only[2] = {0ull, 0ull};
#pragma unroll
      for (int sel = 0; sel < 2; ++sel) {
        const int src = (sel == 1 && shared) ? 0 : sel;
        const uint64_t kmax = ss->ftie[2 * src + 0], knmin = ss->ftie[2 * src + 1];
        only[sel] = kmax;
        if (active[sel]) single = single && knmin != 0ull && kmax == ~knmin;
      }
      if (single) {  // every selected bucket holds one key: it IS the band's end; rank and count inside the bucket are unchanged
        __syncthreads();
        if (t == 0) {
          if (active[0]) pre[0] = only[0];
          if (active[1]) pre[1] = only[1];
        }
        __syncthreads();
        bits_done = 64;
        break;
      }
    }
  }
  // How much of the SAMPLE lies in the band (tie mode: strictly between its ends)?  Exactly known from the scans: ranks above
  // the upper bucket = rank_hi - quo[0], ranks down to the low end of the lower bucket = rank_lo - quo[1] + bucket[1].  More
  // than the candidate storage could take of the vector (6 %) -> the main pass is told not to bother.  With the heavy keys
  // counted as classes this cannot happen any more for a band of +-6 sigma (at most 2.6 % of the smallest sample); kept as
  // a guard.  (A count over one workgroup's own samples misfires on SORTED input, whose band is contiguous.)
  int hopeless = 0;
  if (c == 0 && t == 0) {
    const long long nsamp = (long long)(kFrontBlocks * 1024 * kFrontSpl);
    long long top = active[0] ? (rank0[0] - quo[0]) : 0;                                   // samples above the upper bucket
    long long bot = active[1] ? (rank0[1] - quo[1] + (long long)bucket[1]) : nsamp;          // samples down to the lower bucket's low end
    if (tie) {  // the buckets are single keys now, counted as classes
      if (active[0]) top += (long long)bucket[0];
      if (active[1] && !(active[0] && pre[0] == pre[1])) bot -= (long long)bucket[1];
    }
    const long long in_band = bot > top ? bot - top : 0;
    hopeless = (in_band * 768ll > 48ll * nsamp) ? 1 : 0;
  }
  if (c == 0 && t == 0) {
    FastState& f = ws->fs;
    const int low = 64 - bits_done;  // undecided low bits: take the whole bucket (0 in tie mode)
    const uint64_t lowmask = low > 0 ? (((uint64_t)1 << low) - 1) : 0ull;
    f.t_hi = active[0] ? ((pre[0] << low) | lowmask) : ~0ull;  // nothing is above all-ones
    f.t_lo = active[1] ? (pre[1] << low) : 0ull;
    f.cnt_above = 0;
    f.cand_count = 0;
    f.ok = 0;
    f.key_passes = 0;
    f.overflow = hopeless;
    f.ovf_count = 0;
    f.tie = tie;
    f.has_hi = (tie && active[0]) ? 1 : 0;
    f.has_lo = (tie && active[1] && !(active[0] && f.t_hi == f.t_lo)) ? 1 : 0;
    f.cls_hi = f.cls_lo = 0ull;
    f.todo = 0;
    f.tie_class = 0;
    f.spec_hi = 1;   // (without a cut inside it, class hi lies above the cut and class lo below)
    f.spec_lo = 0;
    f.spec_cut = -1;
    f.t_mid = (active[0] && active[1]) ? f.t_lo + ((f.t_hi - f.t_lo) >> 1)
            : (active[0] ? 0ull : ~0ull);  // (no lower end: r is close to n, the candidates are mostly kept; no upper end: mostly dropped)
    if (tie) {
      // where the sample puts the cut: its rank in the sample is p M; `sabove` samples lie above a class of `m` samples
      const double M = (double)(kFrontBlocks * 1024 * kFrontSpl);
      const double kcut = (double)r / (double)n * M;
      auto guess = [&](double sabove, double m, int& spec) {
        if (kcut <= sabove) spec = 0;
        else if (kcut >= sabove + m) spec = 1;
        else { spec = 2; f.spec_cut = (long long)((kcut - sabove) / m * (double)n); }
      };
      if (f.has_hi) guess((double)(rank0[0] - quo[0]), (double)bucket[0], f.spec_hi);
      if (f.has_lo) guess((double)(rank0[1] - quo[1]), (double)bucket[1], f.spec_lo);
    }
    // (generic data ends on <= kPickFine kFrontSpl samples per bucket, or far fewer after a third digit; in tie mode the keys
    //  BETWEEN the ends may be shared by many elements too: their digits are counted in runs as well)
    f.crowded = tie;
    f.list_count = 0;
    SelState& s = ws->st;
    sel_state_init(s, n, r);
    s.t_floor = f.has_lo ? f.t_lo + 1 : f.t_lo;  // (a class "lo" is not kept when the cut lies among the candidates above it)
    s.quota = 0;
    s.base = f.t_lo;
    if (!active[0]) {  // no upper end: bins scaled by the largest finite sample, last bin open above
      // (r is small against the sample's resolution): bins sized so that the largest finite sample sits in bin 512..1023 and
      // the last bin (open above) starts 4-8x as far from the band's low end; whatever lies beyond (outliers, Inf, NaN) shares
      // that last bin and is resolved on the short list
      const uint64_t smax = f.smax > f.t_lo ? f.smax : f.t_lo;
      const uint64_t dist = (smax - f.t_lo) | 1ull;
      const int bits = 64 - __clzll((long long)dist);
      int sh = bits - 10;
      if (sh < 0) sh = 0;
      if (sh > 51) sh = 51;  // t_lo + (4095 << 51) < 2^64
      s.phase = 0;
      s.shift = sh;
      s.width = kDigitBits;
      s.clamp = 1;
      f.key_passes = 6;
    } else if (f.t_lo == f.t_hi) {
      if (tie) {  // one class, no candidates at all: the state is filled in by k_s2_scan_verify
        s.phase = 0;
        s.shift = 0;
        s.width = 1;
      } else {    // one key value in the band: straight to the index tie-break
        s.t_ge = f.t_lo + 1;
        s.t_eq = f.t_lo;
        s.phase = 1;
        const int idx_bits = s.idx_bits;
        const int w = idx_bits % kDigitBits ? idx_bits % kDigitBits : kDigitBits;
        s.shift = idx_bits - w;
        s.width = w;
      }
    } else {
      const uint64_t span = f.t_hi - f.t_lo;  // > 0
      const int bits = 64 - __clzll((long long)span);
      const int width = bits < kDigitBits ? bits : kDigitBits;
      f.key_passes = (bits + kDigitBits - 1) / kDigitBits;
      s.phase = 0;
      s.shift = bits - width;
      s.width = width;
    }
  }
  SEL_STAMP(15);
}

#ifndef SPX_SEL_REG_MAX_LOG2
#define SPX_SEL_REG_MAX_LOG2 21  // largest n (log2) on the register-resident one-launch select = what 256 CUs hold at 8 elements per lane (n = 2e6: 37 us vs 57 us for the sample-predicted pipeline; beyond it the one-launch form parks v in y and is no faster: tools/r2/topr_midn.py)
#endif

template <bool BINF>
int run_select(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n, int64_t r,
               double delta) {
  int rc = spx_check_common(ctx, y, q, xk, sj, n);
  if (rc) return rc;
  if (n == 0) return SPX_OK;
  SPX_ON_DEVICE(ctx);
  // one workgroup, one launch, no scratch -- up to 8192 elements (n = 16 384: 32 us in one workgroup, 21 us on two;
  // n = 65 536: 84 vs 19 us -- tools/r2/topr_small.py)
  if (n <= kSmallNCoop && ctx->tune_sel_small) {
    hipLaunchKernelGGL((k_sel_small<BINF>), dim3(1), dim3(1024), 0, ctx->stream, y, q, xk, sj, n, r, delta);
    SPX_LAUNCH_CHECK();
    return SPX_OK;
  }
  const int vec = (spx_aligned16(y) && spx_aligned16(q) && spx_aligned16(xk) && spx_aligned16(sj)) ? 1 : 0;
  // all four vectors 8 bytes off a 16-byte boundary (a view that starts at an odd element): the sample-predicted path
  // runs on the aligned rest and its wave 0 takes element 0 along (ioff = 1)
  auto off8 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 8u; };
  const int ioff = (!vec && off8(y) && off8(q) && off8(xk) && off8(sj)) ? 1 : 0;
  // Residency of the kernels that synchronise inside one launch: a grid never exceeds what the occupancy query says can be
  // resident at once (spx_resident_cap); a form whose grid does not fit hands over to the one that works with any grid.
  const int64_t cap_reg = spx_resident_cap(ctx, reinterpret_cast<const void*>(&k_sel_coop<BINF, true>), 1024, 0);
  const int64_t cap_mem = spx_resident_cap(ctx, reinterpret_cast<const void*>(&k_sel_coop<BINF, false>), 1024, 0);
  if (cap_mem < 1) return SPX_ERR_INTERNAL;  // (message set by spx_resident_cap)
  // samples per lane of the front kernel: 1 / 2 / 4 by n (see k_s2_front); 16 for a cut in the bulk of a large vector -- the band
  // is +-6 sigma of the SAMPLE rank, sigma^2 = M p (1 - p): its share of the vector shrinks with sqrt(M), and at r = n/2 the band
  // of the 4-sample form holds 1.2 % of the vector as candidates (main pass +40 us, compaction +20 us; tuning key 10 overrides)
  const double pcut = (double)r / (double)n;
  int spl = n >= ((int64_t)1 << 25) ? 4 : (n >= ((int64_t)1 << 23) ? 2 : 1);
  if (n >= ((int64_t)1 << 26) && pcut * (1.0 - pcut) > 0.04) spl = 16;
  if (ctx->tune_front_spl == 1 || ctx->tune_front_spl == 2 || ctx->tune_front_spl == 4 || ctx->tune_front_spl == 16) spl = ctx->tune_front_spl;
  const void* front_fn = spl == 16 ? reinterpret_cast<const void*>(&k_s2_front<16>)
                       : spl == 4 ? reinterpret_cast<const void*>(&k_s2_front<4>)
                       : spl == 2 ? reinterpret_cast<const void*>(&k_s2_front<2>)
                                  : reinterpret_cast<const void*>(&k_s2_front<1>);
  const int64_t cap_front = spx_resident_cap(ctx, front_fn, 1024, 0);
  const int64_t reg_cap = (int64_t)kCoopEpl * 1024 * (cap_reg < ctx->num_cu ? cap_reg : ctx->num_cu);  // (2 Mi elements on 256 CUs)
  // ... and 4 Mi with v parked in LDS (k_sel_lds): the sample-predicted pipeline (six launches, ~50 us of them fixed cost) takes
  // over above what the resident grid holds on chip.  y must not overlap the inputs elsewhere than element for element (a lane
  // reads all its inputs before it writes: aliasing q, xk or sj exactly is fine, as in the other one-launch forms).
  int64_t lds_cap = 0;
  if (ctx->tune_sel_reg16) {
    const int64_t capl = vec ? spx_resident_cap(ctx, reinterpret_cast<const void*>(&k_sel_lds<BINF, true>), 1024, 0)
                             : spx_resident_cap(ctx, reinterpret_cast<const void*>(&k_sel_lds<BINF, false>), 1024, 0);
    lds_cap = (int64_t)kLdsEpl * 1024 * (capl < ctx->num_cu ? capl : ctx->num_cu);
  }
  // ... and 6 Mi with 8 more elements per lane in registers (k_sel_lds<.., REGX>: 16-byte aligned vectors only; tuning key 11 = 2
  // keeps the pipeline from 4 Mi on)
  int64_t hyb_cap = 0;
  if (ctx->tune_sel_reg16 == 1 && vec && lds_cap > 0) {
    const int64_t caph = spx_resident_cap(ctx, reinterpret_cast<const void*>(&k_sel_lds<BINF, true, double, kLdsRegX>), 1024, 0);
    hyb_cap = (int64_t)(kLdsEpl + kLdsRegX) * 1024 * (caph < ctx->num_cu ? caph : ctx->num_cu);
    if (hyb_cap > 0x7fffffff) hyb_cap = 0;  // (the form indexes with 32-bit integers)
  }
  int64_t fast_min = ((int64_t)1 << SPX_SEL_REG_MAX_LOG2) + 1;
  if (lds_cap >= fast_min) fast_min = lds_cap + 1;
  if (hyb_cap >= fast_min) fast_min = hyb_cap + 1;
  const bool try_fast = ctx->tune_sel_fast && (vec || ioff) && (n - ioff) >= fast_min && r > 0 && r < n &&
                        cap_front >= kFrontBlocks;  // (the front kernel's sample layout is tied to its grid)
  rc = spx_sync_reserve(ctx, sizeof(SelSync));
  if (rc) return rc;
  SelSync* ss = reinterpret_cast<SelSync*>(ctx->sync);
  const bool graph_safe = spx_capture_check(ctx) || ctx->graph_safe;  // (see spx_ctx::graph_safe)
  const int64_t g_mem = cap_mem < ctx->num_cu ? cap_mem : ctx->num_cu;  // grid of the form that parks v in y: any size >= 1 works
  const int64_t cap_tail = spx_resident_cap(ctx, reinterpret_cast<const void*>(&k_s2_tail<BINF>), 1024, 0);
  if (cap_tail < 1) return SPX_ERR_INTERNAL;
  const int64_t g_tail = cap_tail < 256 ? (cap_tail < ctx->num_cu ? cap_tail : ctx->num_cu) : (ctx->num_cu < 256 ? ctx->num_cu : 256);  // (<= 256: SelSync::tie_part)
  int tail_hook = 0;
#ifdef SPX_TEST_HOOKS  // the planted fault of tests/test_gpu_robustness.py (key 101): the last workgroup of the tail kernel leaves
  tail_hook = ctx->tune_force_tail > 0 ? 2 : 0;  // without arriving anywhere, as if it had never been placed -- the others give up waiting
#endif
  if (!try_fast) {
    // exact select in ONE launch: register-resident up to 8 Ki elements per resident workgroup, v parked in y beyond that
    // k_sel_lds from 1 Mi elements on (n = 2e6: 34 / 42 us at r = n/100 / n/2 against 39 / 45 with v in registers; n = 1e6:
    // 36 / 30 against 35 / 29 -- tools/r3/topr_small_grid.py), and wherever the register form's grid does not fit
    const bool lds = n <= lds_cap && (n > kLdsMinN || n > reg_cap);
    const bool hyb = !lds && n > lds_cap && n <= hyb_cap;  // (vec: hyb_cap is 0 otherwise)
    const bool reg = !lds && !hyb && n <= reg_cap;
    // Register form: 1 / 2 / 4 / 8 elements per lane by n -- the fewer elements a lane walks per pass the better, until the
    // workgroups are so many that their histogram flushes and arrivals cost more (us per call at r = n/100, 1 / 2 / 4 / 8 per
    // lane: n = 3e4 17.4 / 18.4 / 20.7 / 24.1; n = 1e5 19.8 / 19.3 / 20.8 / 24.4; n = 3e5 26.8 / 22.8 / 22.1 / 24.9; n = 1e6
    // 35.6 / 35.7 / 36.6 / 34.7 -- tools/r3/topr_small_grid.py)
    const int64_t epl = n <= 65536 ? 1 : n <= 196608 ? 2 : n <= 655360 ? 4 : kCoopEpl;
    int64_t g = g_mem;
    if (reg) {
      const int64_t gmax = cap_reg < ctx->num_cu ? cap_reg : ctx->num_cu;
      g = (n + epl * 1024 - 1) / (epl * 1024);
      if (g > gmax) g = gmax;  // (n <= reg_cap: 8 elements per lane always fit)
    }
    // (k_sel_lds: every CU the grid may have -- the load phase is most of the kernel and wants all of them pulling; a rendezvous
    //  costs 1.4 us with 256 workgroups since the arrivals are spread over eight counters)
    if (lds) g = lds_cap / ((int64_t)kLdsEpl * 1024);
    if (hyb) g = hyb_cap / ((int64_t)(kLdsEpl + kLdsRegX) * 1024);
#ifdef SPX_TEST_HOOKS  // the planted fault of tests/test_gpu_robustness.py: a grid that cannot be resident
    if (!reg && !lds && !hyb && ctx->tune_force_grid > 0) g = ctx->tune_force_grid;
#endif
    int use_set = ctx->sel_hist_next, other = use_set ^ 1;
    int clear_set = ctx->sel_hist_dirty[other] ? other : -1;
    int parity = ctx->coop_parity;
    if (graph_safe) {  // the counters and histogram set 0, zeroed by a node in front of the launch; nothing alternates
      rc = spx_zero_async(ctx, &ss->hdr, sizeof(ss->hdr.bar));
      if (rc) return rc;
      rc = spx_zero_async(ctx, &ss->chist[0][0][0], sizeof(ss->chist[0]));
      if (rc) return rc;
      use_set = 0; other = 1; clear_set = -1; parity = 0;
    }
    {
      SpxCoopLaunchGuard guard(ctx);
      if (reg)
        hipLaunchKernelGGL((k_sel_coop<BINF, true>), dim3((unsigned)g), dim3(1024), 0, ctx->stream, y, q, xk, sj, n, r,
                           delta, ss, parity, use_set, clear_set);
      else if (lds && vec)
        hipLaunchKernelGGL((k_sel_lds<BINF, true>), dim3((unsigned)g), dim3(1024), 0, ctx->stream, y, q, xk, sj, n, r, delta, ss,
                           parity, use_set, clear_set);
      else if (lds)
        hipLaunchKernelGGL((k_sel_lds<BINF, false>), dim3((unsigned)g), dim3(1024), 0, ctx->stream, y, q, xk, sj, n, r, delta, ss,
                           parity, use_set, clear_set);
      else if (hyb)
        hipLaunchKernelGGL((k_sel_lds<BINF, true, double, kLdsRegX>), dim3((unsigned)g), dim3(1024), 0, ctx->stream, y, q, xk, sj,
                           n, r, delta, ss, parity, use_set, clear_set);
      else
        hipLaunchKernelGGL((k_sel_coop<BINF, false>), dim3((unsigned)g), dim3(1024), 0, ctx->stream, y, q, xk, sj, n, r,
                           delta, ss, parity, use_set, clear_set);
    }
    if (graph_safe) {  // whatever the host believed about the sets no longer holds: both count as used
      ctx->sel_hist_dirty[0] = ctx->sel_hist_dirty[1] = 1;
    } else {
      ctx->coop_parity ^= 1;
      ctx->sel_hist_dirty[use_set] = 1;
      ctx->sel_hist_dirty[other] = 0;
      ctx->sel_hist_next = other;
    }
    SPX_LAUNCH_CHECK();
    return SPX_OK;
  }
  // fast path scratch: one candidate region + count word per wavefront of the main pass
  const int64_t n2 = (n - ioff) >> 1;
  const int64_t mblocks = (n2 + kMainTilePairs - 1) / kMainTilePairs;
  const int64_t nregions = mblocks * 4;
  const int64_t ccap = nregions * kWaveSlots;
  // the shared overflow list behind the per-wave regions: room for a band that is contiguous in the vector (sorted input)
  const int64_t ovf_cap64 = (n / 64 > 65536) ? n / 64 : 65536;
  const unsigned int ovf_cap = (unsigned int)(ovf_cap64 > 0x7fffffff ? 0x7fffffff : ovf_cap64);
  const size_t off_cnt = 256;
  const size_t off_cls = (off_cnt + (size_t)nregions * sizeof(WaveCount) + 255) & ~(size_t)255;   // tie mode: class members per slot
  const size_t off_ckey = (off_cls + (size_t)(nregions + 2) * sizeof(ClassCount) + 255) & ~(size_t)255;
  const size_t off_lkey = off_ckey + ((size_t)ccap + (size_t)ovf_cap) * sizeof(Cand);
  const size_t off_lidx = off_lkey + (size_t)kShortList * sizeof(uint64_t);
  const size_t off_lval = off_lidx + (size_t)kShortList * sizeof(int64_t);
  rc = spx_ws_reserve(ctx, off_lval + (size_t)kShortList * sizeof(double) + 256);
  if (rc) return rc;
  char* wsb = reinterpret_cast<char*>(ctx->ws);
  WaveCount* counts = reinterpret_cast<WaveCount*>(wsb + off_cnt);
  ClassCount* cls = reinterpret_cast<ClassCount*>(wsb + off_cls);
  Cand* cand = reinterpret_cast<Cand*>(wsb + off_ckey);
  // single-pass form when y overlaps none of the inputs (a failed prediction recomputes everything from them)
  auto disjoint = [&](const double* a) { return (y + n <= a) || (a + n <= y); };
  const bool write = ctx->tune_sel_spec && disjoint(q) && disjoint(xk) && disjoint(sj);
  uint64_t* lkey = reinterpret_cast<uint64_t*>(wsb + off_lkey);
  int64_t* lidx = reinterpret_cast<int64_t*>(wsb + off_lidx);
  double* lval = reinterpret_cast<double*>(wsb + off_lval);
  const dim3 mgrid((unsigned)mblocks);
  // front (sample + band, one in-launch synchronised kernel) -> main pass -> verdict / candidates -> fallback (exact
  // select; returns at once when the verdict is positive).  Nothing is read back.
  SelWs* sws = &ss->ws;
  if (graph_safe) {
    // counters, the front kernel's histograms and the pipeline's workspace header, zeroed by nodes in front of the
    // launches (in eager mode the previous call's last launch leaves them clean); parities 0 / 1 fixed
    rc = spx_zero_async(ctx, &ss->hdr, sizeof(ss->hdr.bar));
    if (rc) return rc;
    rc = spx_zero_async(ctx, &ss->fhist1[0], sizeof(SelSync) - offsetof(SelSync, fhist1));
    if (rc) return rc;
    ctx->coop_parity = 0;
  }
  {
    SpxCoopLaunchGuard guard(ctx);
    if (spl == 16)
      hipLaunchKernelGGL(k_s2_front<16>, dim3(kFrontBlocks), dim3(1024), 0, ctx->stream, q + ioff, xk + ioff, sj + ioff,
                         n - ioff, r, ss, ctx->coop_parity);
    else if (spl == 4)
      hipLaunchKernelGGL(k_s2_front<4>, dim3(kFrontBlocks), dim3(1024), 0, ctx->stream, q + ioff, xk + ioff, sj + ioff,
                         n - ioff, r, ss, ctx->coop_parity);
    else if (spl == 2)
      hipLaunchKernelGGL(k_s2_front<2>, dim3(kFrontBlocks), dim3(1024), 0, ctx->stream, q + ioff, xk + ioff, sj + ioff,
                         n - ioff, r, ss, ctx->coop_parity);
    else
      hipLaunchKernelGGL(k_s2_front<1>, dim3(kFrontBlocks), dim3(1024), 0, ctx->stream, q + ioff, xk + ioff, sj + ioff,
                         n - ioff, r, ss, ctx->coop_parity);
    ctx->coop_parity ^= 1;
    if (write)
      hipLaunchKernelGGL((k_s2_main<BINF, true>), mgrid, dim3(256), 0, ctx->stream, y + ioff, q + ioff, xk + ioff,
                         sj + ioff, n - ioff, sws, cand, counts, delta, ioff, ccap, ovf_cap, cls, nregions);
    else
      hipLaunchKernelGGL((k_s2_main<BINF, false>), mgrid, dim3(256), 0, ctx->stream, y + ioff, q + ioff, xk + ioff,
                         sj + ioff, n - ioff, sws, cand, counts, delta, ioff, ccap, ovf_cap, cls, nregions);
    // (tried: verdict + first digit redone by every workgroup of the candidate walk instead of the one-workgroup launch in
    //  front of it -- 598 vs 571 us per call at n = 1e8: 512 workgroups x 64 KiB of L2 reads and the scan chain cost more
    //  than the launch they save)
    hipLaunchKernelGGL(k_s2_scan_verify, dim3(1), dim3(256), 0, ctx->stream, sws, r);
    if (write) {
      hipLaunchKernelGGL((k_s2_compact<true, BINF>), dim3(512), dim3(256), 0, ctx->stream, y, (const Cand*)cand, sws, lkey, lidx,
                         lval, (const WaveCount*)counts, nregions, ovf_cap, xk, sj, delta);
      hipLaunchKernelGGL((k_s2_finish<true, BINF>), dim3(1), dim3(1024), 0, ctx->stream, y, sws, (const uint64_t*)lkey,
                         (const int64_t*)lidx, (const double*)lval, xk, sj, delta);
    } else {
      hipLaunchKernelGGL((k_s2_compact<false, BINF>), dim3(512), dim3(256), 0, ctx->stream, y, (const Cand*)cand, sws, lkey, lidx,
                         lval, (const WaveCount*)counts, nregions, ovf_cap, xk, sj, delta);
      hipLaunchKernelGGL((k_s2_finish<false, BINF>), dim3(1), dim3(1024), 0, ctx->stream, y, sws, (const uint64_t*)lkey,
                         (const int64_t*)lidx, (const double*)lval, xk, sj, delta);
    }
    hipLaunchKernelGGL((k_s2_tail<BINF>), dim3((unsigned)g_tail), dim3(1024), 0, ctx->stream, y, q, xk, sj, n, r, delta, ss,
                       ctx->coop_parity, (const Cand*)cand, (const WaveCount*)counts, nregions, ovf_cap,
                       (const ClassCount*)cls, ioff, (write ? 1 : 0) | tail_hook);
    ctx->coop_parity ^= 1;
  }
  if (!write)
    hipLaunchKernelGGL((k_sel_final_q<BINF>), dim3((unsigned)((n2 + 1535) / 1536)), dim3(256), 0, ctx->stream,
                       y + ioff, q + ioff, xk + ioff, sj + ioff, n - ioff, (const SelWs*)sws, delta, ioff);
  SPX_LAUNCH_CHECK();
  return SPX_OK;
}

// Float32 vectors (round 3): the exact select in one launch, on the same kernels -- one workgroup up to 8192 elements,
// register-resident up to 8 Ki elements per resident workgroup, v parked in y beyond.  (The sample-predicted single pass is
// Float64 only: ~44 B/element here instead of 16, 0.7 ms at n = 1e8.)  Bit-exact: v = (xk + sj) + q, the comparisons and the
// final subtraction are Float32 operations, as in the reference with R = Float32.
template <bool BINF>
int run_select_f32(spx_ctx* ctx, float* y, const float* q, const float* xk, const float* sj, int64_t n, int64_t r, float delta) {
  int rc = spx_check_common(ctx, y, q, xk, sj, n);
  if (rc) return rc;
  if (n == 0) return SPX_OK;
  SPX_ON_DEVICE(ctx);
  if (n <= kSmallNCoop && ctx->tune_sel_small) {
    hipLaunchKernelGGL((k_sel_small<BINF, float>), dim3(1), dim3(1024), 0, ctx->stream, y, q, xk, sj, n, r, delta);
    SPX_LAUNCH_CHECK();
    return SPX_OK;
  }
  const int64_t cap_reg = spx_resident_cap(ctx, reinterpret_cast<const void*>(&k_sel_coop<BINF, true, float>), 1024, 0);
  const int64_t cap_mem = spx_resident_cap(ctx, reinterpret_cast<const void*>(&k_sel_coop<BINF, false, float>), 1024, 0);
  if (cap_mem < 1) return SPX_ERR_INTERNAL;
  rc = spx_sync_reserve(ctx, sizeof(SelSync));
  if (rc) return rc;
  SelSync* ss = reinterpret_cast<SelSync*>(ctx->sync);
  const bool graph_safe = spx_capture_check(ctx) || ctx->graph_safe;
  const int64_t reg_cap = (int64_t)kCoopEpl * 1024 * (cap_reg < ctx->num_cu ? cap_reg : ctx->num_cu);
  // v parked in LDS (k_sel_lds<.., float>: 32 Ki elements per resident workgroup, 8 Mi on 256 CUs) above what the register form
  // holds (2 Mi; below, registers win: n = 2e6 39 / 41 us against 43 / 45 -- the Float32 first digit is not folded and the LDS
  // form's third pass costs more); beyond what the grid holds: the form that parks v in y (n = 8e6: 97 -> 64 us,
  // tools/r3/topr_f32_midn.py)
  const bool vec = spx_aligned16(y) && spx_aligned16(q) && spx_aligned16(xk) && spx_aligned16(sj);
  int64_t lds_cap = 0;
  if (ctx->tune_sel_reg16) {
    const int64_t capl = vec ? spx_resident_cap(ctx, reinterpret_cast<const void*>(&k_sel_lds<BINF, true, float>), 1024, 0)
                             : spx_resident_cap(ctx, reinterpret_cast<const void*>(&k_sel_lds<BINF, false, float>), 1024, 0);
    lds_cap = (int64_t)kLdsEpl32 * 1024 * (capl < ctx->num_cu ? capl : ctx->num_cu);
  }
  const bool lds = n <= lds_cap && n > reg_cap;
  const bool reg = !lds && n <= reg_cap;
  int64_t g = reg ? (n + (int64_t)kCoopEpl * 1024 - 1) / ((int64_t)kCoopEpl * 1024) : (cap_mem < ctx->num_cu ? cap_mem : ctx->num_cu);
  if (lds) g = lds_cap / ((int64_t)kLdsEpl32 * 1024);
  int use_set = ctx->sel_hist_next, other = use_set ^ 1;
  int clear_set = ctx->sel_hist_dirty[other] ? other : -1;
  int parity = ctx->coop_parity;
  if (graph_safe) {
    rc = spx_zero_async(ctx, &ss->hdr, sizeof(ss->hdr.bar));
    if (rc) return rc;
    rc = spx_zero_async(ctx, &ss->chist[0][0][0], sizeof(ss->chist[0]));
    if (rc) return rc;
    use_set = 0; other = 1; clear_set = -1; parity = 0;
  }
  {
    SpxCoopLaunchGuard guard(ctx);
    if (reg)
      hipLaunchKernelGGL((k_sel_coop<BINF, true, float>), dim3((unsigned)g), dim3(1024), 0, ctx->stream, y, q, xk, sj, n, r, delta, ss,
                         parity, use_set, clear_set);
    else if (lds && vec)
      hipLaunchKernelGGL((k_sel_lds<BINF, true, float>), dim3((unsigned)g), dim3(1024), 0, ctx->stream, y, q, xk, sj, n, r, delta, ss,
                         parity, use_set, clear_set);
    else if (lds)
      hipLaunchKernelGGL((k_sel_lds<BINF, false, float>), dim3((unsigned)g), dim3(1024), 0, ctx->stream, y, q, xk, sj, n, r, delta, ss,
                         parity, use_set, clear_set);
    else
      hipLaunchKernelGGL((k_sel_coop<BINF, false, float>), dim3((unsigned)g), dim3(1024), 0, ctx->stream, y, q, xk, sj, n, r, delta, ss,
                         parity, use_set, clear_set);
  }
  if (graph_safe) {
    ctx->sel_hist_dirty[0] = ctx->sel_hist_dirty[1] = 1;
  } else {
    ctx->coop_parity ^= 1;
    ctx->sel_hist_dirty[use_set] = 1;
    ctx->sel_hist_dirty[other] = 0;
    ctx->sel_hist_next = other;
  }
  SPX_LAUNCH_CHECK();
  return SPX_OK;
}

}  // namespace

SPX_EXPORT int spx_prox_indball_l0_f32(spx_ctx* ctx, float* y, const float* q, const float* xk, const float* sj, int64_t n,
                                       int64_t r) {
  return run_select_f32<false>(ctx, y, q, xk, sj, n, r, 0.0f);
}
SPX_EXPORT int spx_prox_indball_l0_binf_f32(spx_ctx* ctx, float* y, const float* q, const float* xk, const float* sj, int64_t n,
                                            int64_t r, float delta) {
  return run_select_f32<true>(ctx, y, q, xk, sj, n, r, delta);
}

SPX_EXPORT int spx_prox_indball_l0(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                                   int64_t n, int64_t r) {
  return run_select<false>(ctx, y, q, xk, sj, n, r, 0.0);
}

SPX_EXPORT int spx_prox_indball_l0_binf(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                                        int64_t n, int64_t r, double delta) {
  return run_select<true>(ctx, y, q, xk, sj, n, r, delta);
}
