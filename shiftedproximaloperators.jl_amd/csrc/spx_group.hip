// spx_group.hip -- ShiftedGroupNormL2.prox! and ShiftedGroupNormL2Binf.prox! (group-l2 block soft-threshold,
// with an l-infinity trust region in the Binf form).
//
// HBM layout: q, xk, sj, y contiguous fp64; groups are contiguous index ranges (CSR offsets or a uniform
// size); lambda is one fp64 per group.  Algorithmic traffic: 32 B/element + 8 B/group.
// Roofline: HBM bandwidth; the Binf form adds a per-group scalar root find that runs out of registers.
//
// Mapping.  Fast path (uniform group size LPG*EPL, 16-byte aligned): a group is owned by LPG lanes of a
// wavefront (LPG = 16 for the 128-element groups of the BASELINE config, so one wave works on 4 groups at
// once), each lane keeps EPL elements of S = (q + xk) + sj, X = xk, xk + sj and S/sigma in registers; q/xk/sj
// are read once with non-temporal 16-byte loads and y is written once.  Sums over a group are DPP
// butterflies inside a 16-lane row (no LDS).  Packing several groups into a wave amortises the
// per-group scalar arithmetic of the Binf root find (divisions, square roots), which every lane of a
// group executes redundantly.  Other shapes: a wavefront or a 256-lane workgroup per group, elements
// re-read from L1/L2 for every reduction.
#include "spx_group_common.hpp"

#ifdef SPX_DEBUG_PEEK  // diagnostic builds only: [0] last raw count the LIT launch read, [1] LIT launches that read a count
                       // outside [0, ngroups], [2] LIT launches, [3] count at the entry of the last main launch, [4] main
                       // launches, [5] main launches that found the count outside [0, ngroups] on entry
__device__ long long g_group_dbg[8];
extern "C" __attribute__((visibility("default"))) int spx_debug_group_words(long long* out8, int reset) {
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_group_dbg), sizeof(g_group_dbg));
  if (e == hipSuccess && reset) {
    long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    e = hipMemcpyToSymbol(HIP_SYMBOL(g_group_dbg), z, sizeof(z));
  }
  return (int)e;
}
#endif


// ---------------------------------------------------------------------------------------------
// fast path kernel: uniform groups of LPG*EPL elements; LPG lanes own a group, 64/LPG groups per wave;
// the group is resident in registers.  Lane j of a group owns the 16-byte pairs j, j + LPG, j + 2 LPG, ...
// ---------------------------------------------------------------------------------------------
#ifndef SPX_PADDED_CACHED
#define SPX_PADDED_CACHED 1  // A/B switch (round 4): cached accesses on partly filled tiles of <= 16 lanes per group (kPaddedCached)
#endif
#ifndef SPX_GROUP_PREFETCH
#define SPX_GROUP_PREFETCH 1  // A/B switch (round 4): Binf tiles of one / two lanes x 8 elements come in through LDS as whole kilobytes (kPre)
#endif
#ifndef SPX_GROUP_WAVES
#define SPX_GROUP_WAVES 3  // min waves/SIMD (VGPR cap) of the 8-element tiles.  Binf 1e6x128 on 16 lanes x 8: 3 (162 VGPRs, no spill) 0.84 ms; 4 (128, cold paths spill) 0.93 ms; 5: 1.49 ms
// Binf tiles with 16 elements per lane (8 x 16 for 128-element groups: 8 groups per wave) run at 2 waves/SIMD (253 VGPRs):
// the kernel is VALU-bound and the wave-uniform scalar work of the root find is shared by twice as many elements --
// 0.79 -> 0.70 ms at 1e6 x 128 in spite of the lower occupancy.
#endif
// PAIRS: the group size is even (every group starts 16-byte aligned): lane j owns the pairs j, j + LPG, ...; pairs past
// the end of the group are read as zeros (zeros are neutral in every sum of both operators).  !PAIRS: odd group size,
// lane j owns the elements j, j + LPG, ... through 8-byte loads.
// LIT (Binf only): second launch over the deferred list -- `deferred` is then read: [0] = number of groups, [1..] = their
// ids -- evaluating the reference's expressions literally on the register-resident group (binf_literal_reg).
// FULL (PAIRS only): the group size is exactly LPG * EPL -- no pair of the tile is masked, which takes the zero-fill selects
// and the clamped addresses (~7 % of the kernel's VALU instructions) out of the BASELINE shapes (128 = 8 x 16 = 16 x 8).
template <int LPG, int EPL, bool BINF, bool PAIRS, bool LIT = false, bool FULL = false>
__global__ __launch_bounds__(256, (BINF && EPL >= 16) ? 2 : SPX_GROUP_WAVES) void k_group_reg(double* y_, const double* q_, const double* xk_, const double* sj_,
                                                    int64_t ngroups, int gsize, const double* __restrict__ lambda,
                                                    double sigma, double delta,
                                                    long long* deferred /* [1..] = groups ([0]: the count, unless dcount points elsewhere) */,
                                                    const int64_t* __restrict__ offsets /* !PAIRS only: ragged groups */,
                                                    int* status /* spx_ctx::status_dev */, int pole_lit /* tuning key 9 */,
                                                    unsigned long long* dcount /* the list's count word: deferred[0], or one of SpxSyncHeader::grp_deferred */,
                                                    unsigned long long* dclear /* LIT: the count word the NEXT call uses, zeroed here (or NULL) */) {
  static_assert((EPL % 2) == 0, "EPL must be even (16-byte pairs)");
  const int64_t GS = gsize;  // <= LPG * EPL
  constexpr int GPW = 64 / LPG;  // groups per wave
  const int lane = threadIdx.x & 63;
  const int j = lane % LPG;
  const int slot = lane / LPG;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  // plain GroupNormL2 stages its loads through LDS (LDS-DMA, +2 %: 0.650-0.664 vs 0.666-0.677 ms at 1e6 x 128);
  // the Binf form is VALU-bound and loses 6 % to the lower occupancy the LDS footprint allows, so it keeps register loads
  constexpr bool kDma = !BINF && PAIRS;
  // Round 4 -- partly filled tiles (a group size that is not lanes x elements, e.g. 100 on 8 x 16): the 16-byte pairs a group's
  // lanes fetch per instruction are a run of 16 LPG bytes that starts wherever the group does -- 800-byte groups: three runs
  // of four straddle two 128-byte lines -- and the next run of the group continues in the line the last one ended in.
  // Non-temporal accesses gave that line up in between; cached ones keep it: Binf groups of 20 / 50 / 66 / 100 / 120 at
  // n = 1e8: 1001 / 809 / 936 / 708 / 607 -> 807 / 719 / 840 / 652 / 583 us (4.52 -> 4.91 TB/s on groups of 100), the plain
  // operator on groups of 10 / 66 / 100: 671 / 664 / 600 -> 611 / 585 / 567 us.  Runs of 512 bytes and more (32 and 64 lanes
  // per group) are better off streaming (groups of 300 / 500: 2-4 % slower cached).
  constexpr bool kPaddedCached = SPX_PADDED_CACHED && PAIRS && !FULL && !LIT && LPG <= 16;
  // Round 4 -- Binf on the one- and two-lane tiles of 8 elements per lane (groups of 5 .. 16): the tile comes in and goes out
  // through a buffer of the wavefront in LDS, as whole kilobytes (tile_dma / tile_regs below).
  constexpr bool kPre = BINF && PAIRS && !LIT && EPL == 8 && LPG <= 2 && !(LPG == 2 && FULL) && SPX_GROUP_PREFETCH;
  __shared__ __attribute__((aligned(16))) char dma_lds[(kDma || kPre) ? 4 * 3 * (EPL / 2) * 1024 : 16];
  const int npairs = gsize >> 1;
  // (the list can hold at most every group once: a count outside [0, ngroups] is never followed into memory -- and never
  //  skipped silently either: the context's status word is raised and every later call fails, spx_common.hpp)
  const int64_t nlist = LIT ? (int64_t)*dcount : 0;
  if (LIT && dclear != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *dclear = 0ull;  // (nobody else touches that word in this call)
  const bool bad_count = LIT && (nlist < 0 || nlist > ngroups);
  if (bad_count && blockIdx.x == 0 && threadIdx.x == 0) spx_raise_status(status, kSpxStatusCorrupt);
  const int64_t ntodo = LIT ? (bad_count ? 0 : nlist) : ngroups;
#ifdef SPX_DEBUG_PEEK  // diagnostic builds (tools/r3/graph_fault_probe.py): what the count word held when a launch read it
  if (deferred != nullptr && blockIdx.x == 0 && threadIdx.x == 0) {
    if constexpr (LIT) {
      g_group_dbg[0] = nlist;
      if (nlist < 0 || nlist > ngroups) g_group_dbg[1] += 1;
      g_group_dbg[2] += 1;
    } else {
      g_group_dbg[3] = (long long)__hip_atomic_load(dcount, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (g_group_dbg[3] < 0 || g_group_dbg[3] > ngroups) g_group_dbg[5] += 1;
      g_group_dbg[4] += 1;
    }
  }
#endif
  typedef __attribute__((address_space(3))) void lds_void;
  char* const wl = dma_lds + (threadIdx.x >> 6) * (3 * (EPL / 2) * 1024);  // (kDma / kPre) this wavefront's staging buffer
  // kPre (round 4, Binf on the one- and two-lane tiles of 8 elements per lane): the tile goes through the wavefront's buffer.
  //   tile_dma(t)  : the tile of wave iteration t -- GPW consecutive groups, GPW * npairs consecutive 16-byte pairs -- comes in
  //                  as whole kilobytes (lane l fetches pair 64 k + l) and lies linearly in the buffer (direct loads had every
  //                  lane fetch 16 bytes of its own 64-byte run per instruction: 32 lines touched four times each);
  //   tile_regs(t) : every lane picks up its own pairs (a 64-byte run per lane on the one-lane tiles: 4-way bank conflicts on
  //                  12 LDS reads, nothing next to the root find);
  // Groups of 8 at n = 1e8: 660 -> 636 us; of 6 / 12 / 14: -4 .. -7 %; y sent back the same way (parked in the buffer, stored as
  // whole kilobytes) changes nothing (634 us) and costs registers: the lanes store their own pairs.
  // (Tried on top and dropped: a resident grid whose wavefronts fetch tile i + 1 during the root find of tile i -- one buffer
  //  per wavefront is enough, the tile is copied to registers before the next DMA is issued; stores never waited for.  Bit
  //  identical and SLOWER: 665 us on groups of 8, and the second tile's registers spill on the partly filled tiles.  The
  //  hardware's own hand-out of one-tile workgroups hides the loads at least as well.)
  auto tile_dma = [&](int64_t t0) {
    const int64_t left = ntodo - t0;
    const int tile_pairs = (int)(left < GPW ? left : GPW) * npairs;  // (wave-uniform)
    const f64x2* tq = reinterpret_cast<const f64x2*>(q_ + t0 * GS);
    const f64x2* tx = reinterpret_cast<const f64x2*>(xk_ + t0 * GS);
    const f64x2* ts = reinterpret_cast<const f64x2*>(sj_ + t0 * GS);
#pragma unroll
    for (int k = 0; k < EPL / 2; ++k) {
      if (k * 64 < tile_pairs) {  // (wave-uniform)
        const int pp = (k * 64 + lane < tile_pairs) ? (k * 64 + lane) : 0;
        __builtin_amdgcn_global_load_lds((const void*)(tq + pp), (lds_void*)(wl + (0 * (EPL / 2) + k) * 1024), 16, 0, 2);
        __builtin_amdgcn_global_load_lds((const void*)(tx + pp), (lds_void*)(wl + (1 * (EPL / 2) + k) * 1024), 16, 0, 2);
        __builtin_amdgcn_global_load_lds((const void*)(ts + pp), (lds_void*)(wl + (2 * (EPL / 2) + k) * 1024), 16, 0, 2);
      }
    }
  };
  auto tile_regs = [&](int64_t t0, f64x2 (&nq)[EPL / 2], f64x2 (&nx)[EPL / 2], f64x2 (&ns)[EPL / 2]) {
    const int64_t left = ntodo - t0;
    const int slot_e = ((t0 + slot) < ntodo) ? slot : (int)(left - 1);  // idle slots shadow the last group of the tile
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < EPL / 2; ++k) {
      const int p = (FULL || k * LPG + j < npairs) ? (slot_e * npairs + k * LPG + j) : 0;  // masked pairs: zeroed below
      nq[k] = *reinterpret_cast<const f64x2*>(wl + (0 * (EPL / 2)) * 1024 + p * 16);
      nx[k] = *reinterpret_cast<const f64x2*>(wl + (1 * (EPL / 2)) * 1024 + p * 16);
      ns[k] = *reinterpret_cast<const f64x2*>(wl + (2 * (EPL / 2)) * 1024 + p * 16);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  };
  for (int64_t g0 = wave * GPW; g0 < ntodo; g0 += nwaves * GPW) {  // wave-uniform trip count
    bool valid = (g0 + slot) < ntodo;
    const int64_t gi = valid ? (g0 + slot) : (ntodo - 1);  // idle slots shadow the last group, no store
    int64_t g = gi;
    if constexpr (LIT) {
      g = (int64_t)deferred[1 + gi];
      if (g < 0 || g >= ngroups) {  // (never an id from outside the layout; reported, not skipped silently)
        if (valid && j == 0) spx_raise_status(status, kSpxStatusCorrupt);
        g = 0;
        valid = false;
      }
    }
    int64_t base = g * GS;
    int gs = gsize;  // this group's size (row-uniform)
    if constexpr (!PAIRS) {
      if (offsets) {  // ragged groups whose sizes the caller bounded by the tile (group_size hint)
        base = offsets[g];
        const int64_t sz = offsets[g + 1] - base;
        if (sz > LPG * EPL || sz < 0) {  // the hint was wrong for this group: the general kernel takes it
          if (valid && j == 0) deferred[1 + atomicAdd(dcount, 1ull)] = g;
          valid = false;
          gs = 0;
        } else {
          gs = (int)sz;
        }
      }
    }
    RegGroup<EPL, BINF && !FULL && !LIT> grp;
    if constexpr (BINF && !FULL && !LIT) {  // live slots of the tile (wave-uniform: the launch's group size / size bound)
      const int slices = PAIRS ? (npairs + LPG - 1) / LPG : ((gsize + LPG - 1) / LPG + 1) / 2;
      grp.live = (2 * slices < EPL && offsets == nullptr) ? 2 * slices : EPL;  // (ragged groups: `gsize` is only the caller's hint -- a
                                                                                 //  group above it that still fits the tile is served here)
    }
    {
      if constexpr (PAIRS) {
        const f64x2* q2 = reinterpret_cast<const f64x2*>(q_ + base);
        const f64x2* x2 = reinterpret_cast<const f64x2*>(xk_ + base);
        const f64x2* s2 = reinterpret_cast<const f64x2*>(sj_ + base);
        f64x2 vq[EPL / 2], vx[EPL / 2], vs[EPL / 2];
        if constexpr (kPre) {
          tile_dma(g0);
          tile_regs(g0, vq, vx, vs);
        } else if constexpr (kDma) {
          // LDS-DMA staging (as k_sep_lds): piece k of a wave = the k-th 16-byte pair of each lane; lane `lane` of the
          // wave lands at byte 16*lane of the piece, whichever group slot it serves
#pragma unroll
          for (int k = 0; k < EPL / 2; ++k) {
            const int p = (FULL || k * LPG + j < npairs) ? (k * LPG + j) : 0;  // masked pairs re-read pair 0, zeroed below
            __builtin_amdgcn_global_load_lds((const void*)(q2 + p), (lds_void*)(wl + (0 * (EPL / 2) + k) * 1024), 16, 0, kPaddedCached ? 0 : 2);
            __builtin_amdgcn_global_load_lds((const void*)(x2 + p), (lds_void*)(wl + (1 * (EPL / 2) + k) * 1024), 16, 0, kPaddedCached ? 0 : 2);
            __builtin_amdgcn_global_load_lds((const void*)(s2 + p), (lds_void*)(wl + (2 * (EPL / 2) + k) * 1024), 16, 0, kPaddedCached ? 0 : 2);
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int k = 0; k < EPL / 2; ++k) {
            vq[k] = *reinterpret_cast<const f64x2*>(wl + (0 * (EPL / 2) + k) * 1024 + lane * 16);
            vx[k] = *reinterpret_cast<const f64x2*>(wl + (1 * (EPL / 2) + k) * 1024 + lane * 16);
            vs[k] = *reinterpret_cast<const f64x2*>(wl + (2 * (EPL / 2) + k) * 1024 + lane * 16);
          }
        } else {
#pragma unroll
          for (int k = 0; k < EPL / 2; ++k) {
            const int p = (FULL || k * LPG + j < npairs) ? (k * LPG + j) : 0;
            if constexpr (LPG <= 2 || kPaddedCached) {  // a lane's pairs share cache lines with each other, not with its neighbours': cached accesses
              vq[k] = q2[p]; vx[k] = x2[p]; vs[k] = s2[p];
            } else {
              vq[k] = __builtin_nontemporal_load(q2 + p);
              vx[k] = __builtin_nontemporal_load(x2 + p);
              vs[k] = __builtin_nontemporal_load(s2 + p);
            }
          }
        }
#pragma unroll
        for (int k = 0; k < EPL / 2; ++k) {
          const bool in = FULL || (k * LPG + j) < npairs;
          const f64x2 zero2 = f64x2{0.0, 0.0};
          const f64x2 a = in ? vq[k] : zero2, b = in ? vx[k] : zero2, c = in ? vs[k] : zero2;
          grp.S[2 * k] = (a.x + b.x) + c.x;  // shiftedGroupNormL2.jl:65 / shiftedGroupNormL2Binf.jl:80
          grp.S[2 * k + 1] = (a.y + b.y) + c.y;
          grp.X[2 * k] = b.x;
          grp.X[2 * k + 1] = b.y;
          grp.XS[2 * k] = b.x + c.x;
          grp.XS[2 * k + 1] = b.y + c.y;
        }
      } else {
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
          const int e = k * LPG + j;
          const bool in = e < gs;
          const int64_t i = base + (in ? e : 0);
          const double a = in ? q_[i] : 0.0, b = in ? xk_[i] : 0.0, c = in ? sj_[i] : 0.0;
          grp.S[k] = (a + b) + c;
          grp.X[k] = b;
          grp.XS[k] = b + c;
        }
      }
    }
    const double lam = lambda[g];
    double out[EPL];
    if constexpr (!BINF) {
      double ss = 0.0;
#pragma unroll
      for (int k = 0; k < EPL; ++k) ss += grp.S[k] * grp.S[k];
      const double snorm = sqrt(lanes_sum<LPG>(ss));                                       // shiftedGroupNormL2.jl:69
      const double alpha = (snorm == 0.0) ? 0.0 : jl_max(1 - sigma * lam / snorm, 0.0);  // :70-73
#pragma unroll
      for (int k = 0; k < EPL; ++k) out[k] = ((snorm == 0.0) ? 0.0 : alpha * grp.S[k]) - grp.XS[k];  // :74,:77
    } else if constexpr (LIT) {
      binf_literal_reg<LPG, EPL, false>(grp, lam, sigma, delta, out);
#pragma unroll
      for (int k = 0; k < EPL; ++k) out[k] = out[k] - grp.XS[k];  // :116
    } else {
      double ru;
      const int status = binf_root<LPG>(grp, lam, sigma, delta, nullptr, ru, pole_lit != 0);
      const double sl = lam * sigma;
      if (status == BINF_LITERAL) {
        // rare (degenerate bracket / exact zero / NaN): handed to k_group_list, which evaluates the reference's
        // expressions literally; keeping that code out of this kernel saves ~40 VGPRs
        // (one atomic per group.  A data set whose groups ALL defer is bound by this counter: the hardware already merges
        //  the atomics of a wavefront into one request, ~11 ns each on the one address -- 8e6 groups of 16 = 5e5 requests
        //  = 6 ms; merging them in software changes nothing.  Sharded lists would; not needed for the cases at hand.)
        if (valid && j == 0) deferred[1 + atomicAdd(dcount, 1ull)] = g;
        valid = false;
      }
      if (status != BINF_ROOT || ru == 0.0) {  // shiftedGroupNormL2Binf.jl:102-103, :107-108
#pragma unroll
        for (int k = 0; k < EPL; ++k) out[k] = 0.0 - grp.XS[k];
      } else {
        const double tau = ru * fast_rcp(sl + ru);  // = alpha at the root (:83 with ||w|| = n)
#pragma unroll
        for (int k = 0; k < EPL; ++k) out[k] = binf_y(grp.S[k], grp.X[k], tau, delta) - grp.XS[k];  // :110-116
      }
    }
    if (valid) {
      if constexpr (PAIRS) {
        f64x2* y2 = reinterpret_cast<f64x2*>(y_ + base);
#pragma unroll
        for (int k = 0; k < EPL / 2; ++k)
          if (FULL || k * LPG + j < npairs) {
            if constexpr (LPG <= 2 || kPaddedCached) y2[k * LPG + j] = f64x2{out[2 * k], out[2 * k + 1]};  // (merged into whole lines by the L2)
            else __builtin_nontemporal_store(f64x2{out[2 * k], out[2 * k + 1]}, y2 + k * LPG + j);
          }
      } else {
#pragma unroll
        for (int k = 0; k < EPL; ++k)
          if (k * LPG + j < gs) y_[base + k * LPG + j] = out[k];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// general kernels: TEAM lanes per group, elements re-read from memory for every reduction.
//   k_group_mem    : contiguous groups (CSR offsets or uniform size)
//   k_group_gather : arbitrary index sets (the reference's idx::Vector{Vector{Int}}, src/groupNormL2.jl:30-31)
// Both run group_body on an element provider (MemGroup / GatherGroup).
// ---------------------------------------------------------------------------------------------
// Gather provider.  `sol` = (q + xk) + sj was materialised by k_gather_prepare (as the reference's psi.sol, :65 -- so
// y may alias q and groups may overlap); owner[j] = the LAST group containing j: only that group stores y[j], which is
// what the reference's sequential loop over the groups leaves behind.
template <int TEAM>
struct GatherGroup {
  static constexpr bool kReg = false;
  static constexpr int kEpl = 1;
  const double* sol;
  const double* xk;
  const double* sj;
  const int64_t* index;
  const int* owner;
  int64_t lo, hi;  // positions in `index`
  int lane;
  int g;
  template <class F>
  __device__ __forceinline__ void for_each(F&& f) const {
    for (int64_t p = lo + lane; p < hi; p += TEAM) {
      const int64_t j = index[p];
      f(sol[j], xk[j]);
    }
  }
  // y[j] = f(S, X) - (xk + sj) for every owned element
  template <class F>
  __device__ __forceinline__ void store(double* y, F&& f) const {
    for (int64_t p = lo + lane; p < hi; p += TEAM) {
      const int64_t j = index[p];
      const double x = xk[j], s = sj[j];
      const double v = f(sol[j], x) - (x + s);
      if (owner[j] == g) y[j] = v;
    }
  }
};


template <int TEAM, bool BINF>
__global__ __launch_bounds__(256) void k_group_mem(double* y, const double* q, const double* xk, const double* sj,
                                                    int64_t n, const int64_t* __restrict__ offsets, int64_t gsize,
                                                    int64_t ngroups, const double* __restrict__ lambda, double sigma,
                                                    double delta, const long long* list /* NULL, or [0] = count, [1..] */,
                                                    int* status /* spx_ctx::status_dev */, int pole_lit,
                                                    const int* big_active /* NULL, or a device word: skip groups of >= big_min elements (spx_group_team.hip has them) */,
                                                    int64_t big_min) {
  __shared__ double lds[8];
  const bool skip_big = big_active != nullptr && *big_active != 0;
  constexpr int TPB = 256 / TEAM;  // teams per block
  const int lane = threadIdx.x % TEAM;
  const int64_t team = (int64_t)blockIdx.x * TPB + threadIdx.x / TEAM;
  const int64_t nteams = (int64_t)gridDim.x * TPB;
  const int64_t nlist = list ? (int64_t)list[0] : 0;
  const bool bad_count = list && (nlist < 0 || nlist > ngroups);  // (as k_group_reg: reported through the status word)
  if (bad_count && blockIdx.x == 0 && threadIdx.x == 0) spx_raise_status(status, kSpxStatusCorrupt);
  const int64_t ntodo = list ? (bad_count ? 0 : nlist) : ngroups;
  for (int64_t t = team; t < ntodo; t += nteams) {  // for TEAM == 256 the trip count is block-uniform
    const int64_t g = list ? (int64_t)list[1 + t] : t;
    if (g < 0 || g >= ngroups) {  // (never an id from outside the layout; block-uniform for TEAM == 256)
      if (lane == 0) spx_raise_status(status, kSpxStatusCorrupt);
      continue;
    }
    int64_t lo, hi;
    if (offsets) { lo = offsets[g]; hi = offsets[g + 1]; }
    else { lo = g * gsize; hi = lo + gsize; }
    if (lo < 0) lo = 0;
    if (hi > n) hi = n;
    if (skip_big && hi - lo >= big_min) continue;  // (team- and block-uniform)
    MemGroup<TEAM> grp{q, xk, sj, lo, hi, lane};
    group_body<TEAM, BINF>(grp, y, lambda[g], sigma, delta, lds, list != nullptr, pole_lit != 0);
    if constexpr (TEAM == 256) __syncthreads();
  }
}

// Large uniform groups (512 < size <= kLdsGroupMax): one workgroup per group, S = (q + xk) + sj and X = xk staged in
// LDS once (q, xk, sj are read from HBM exactly once; the final store re-reads sj only), every reduction of the root
// find then runs out of LDS instead of re-reading the group from L2.
// Measured (1.28e8 elements, TB/s, LDS kernel vs general kernel): size 1000: 4.8 vs 3.9 (L2), 2.5 vs 1.5 (Binf);
// 2048: 5.0 / 3.3; 4096: 3.3 vs 3.9 / 2.6 vs 1.5; 8192: 2.4 vs 3.9 / 1.6 vs 1.5 -- the LDS footprint leaves one or two
// workgroups per CU, so the plain form switches back to the general kernel above 2048 and the Binf form above 4096.
constexpr int kLdsGroupMax = 4096;  // elements (Binf); 2 x 32 KiB of LDS
constexpr int kLdsGroupMaxPlain = 2048;
struct LdsGroup {
  static constexpr bool kReg = false;
  static constexpr int kEpl = 1;
  const double* S;   // LDS
  const double* X;   // LDS
  const double* sj;  // global, group base
  int64_t lo;        // first element of the group
  int m, tid;
  template <class F>
  __device__ __forceinline__ void for_each(F&& f) const {
    for (int i = tid; i < m; i += 256) f(S[i], X[i]);
  }
  template <class F>
  __device__ __forceinline__ void store(double* y, F&& f) const {
    for (int i = tid; i < m; i += 256) {
      const double x = X[i], s = sj[i];
      y[lo + i] = f(S[i], x) - (x + s);
    }
  }
};

template <bool BINF>
__global__ __launch_bounds__(256) void k_group_lds(double* y, const double* q, const double* xk, const double* sj,
                                                    int64_t n, const int64_t* __restrict__ offsets /* NULL: uniform */,
                                                    int64_t gsize /* group size, or the bound on it with offsets */,
                                                    int64_t ngroups, const double* __restrict__ lambda, double sigma,
                                                    double delta, int pole_lit) {
  extern __shared__ __attribute__((aligned(16))) double dyn[];  // S[gsize] | X[gsize]
  __shared__ double lds[8];
  double* S = dyn;
  double* X = dyn + gsize;
  const int tid = threadIdx.x;
  for (int64_t g = blockIdx.x; g < ngroups; g += gridDim.x) {
    int64_t lo = g * gsize, sz = gsize;
    if (offsets) {
      lo = offsets[g];
      int64_t hi = offsets[g + 1];
      if (lo < 0) lo = 0;
      if (hi > n) hi = n;
      sz = hi > lo ? hi - lo : 0;
      if (sz > gsize) {  // the caller's bound was wrong for this group: straight from memory, as the general kernel
        MemGroup<256> big{q, xk, sj, lo, hi, tid};
        group_body<256, BINF>(big, y, lambda[g], sigma, delta, lds, false, pole_lit != 0);
        __syncthreads();
        continue;
      }
    }
    const int m = (int)sz;
    for (int i = tid; i < m; i += 256) {
      const double x = xk[lo + i];
      S[i] = (q[lo + i] + x) + sj[lo + i];  // shiftedGroupNormL2.jl:65 / shiftedGroupNormL2Binf.jl:80
      X[i] = x;
    }
    __syncthreads();  // the group is staged (and q fully read: y may alias q)
    LdsGroup grp{S, X, sj + lo, lo, m, tid};
    group_body<256, BINF>(grp, y, lambda[g], sigma, delta, lds, false, pole_lit != 0);
    __syncthreads();  // all reads of S / X done before the next group overwrites them
  }
}

// ShiftedGroupNormL2 with CSR offsets that do not span 0:n: indices before offsets[0] / from offsets[ngroups] on keep
// the caller's y minus the shift (src/shiftedGroupNormL2.jl:77 runs over every index)
__global__ __launch_bounds__(256) void k_csr_uncovered(double* y, const double* xk, const double* sj,
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

// sol = (q + xk) + sj (:65 / :80), owner = -1
__global__ __launch_bounds__(256) void k_gather_prepare(double* __restrict__ sol, int* __restrict__ owner,
                                                         const double* q, const double* xk, const double* sj, int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    sol[i] = (q[i] + xk[i]) + sj[i];
    owner[i] = -1;
  }
}

// owner[j] = max{g : j in idx_g}; flags an index outside [0, n) (the reference: BoundsError)
__global__ __launch_bounds__(256) void k_gather_owner(int* owner, const int64_t* __restrict__ ptr,
                                                       const int64_t* __restrict__ index, int64_t ngroups, int64_t n,
                                                       int64_t nnz, int* flag) {
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int64_t nwaves = ((int64_t)gridDim.x * blockDim.x) >> 6;
  for (int64_t g = wave; g < ngroups; g += nwaves) {
    int64_t lo = ptr[g], hi = ptr[g + 1];
    if (lo < 0 || hi > nnz || lo > hi) { if (lane == 0) atomicOr(flag, 2); continue; }
    for (int64_t p = lo + lane; p < hi; p += 64) {
      const int64_t j = index[p];
      if ((j < 0 || j >= n) && (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 1) == 0) atomicOr(flag, 1);
      else atomicMax(owner + j, (int)g);
    }
  }
}

template <int TEAM, bool BINF>
__global__ __launch_bounds__(256) void k_group_gather(double* y, const double* sol, const double* xk, const double* sj,
                                                       const int* owner, const int64_t* __restrict__ ptr,
                                                       const int64_t* __restrict__ index, int64_t ngroups,
                                                       const double* __restrict__ lambda, double sigma, double delta,
                                                       int pole_lit) {
  __shared__ double lds[8];
  constexpr int TPB = 256 / TEAM;
  const int lane = threadIdx.x % TEAM;
  const int64_t team = (int64_t)blockIdx.x * TPB + threadIdx.x / TEAM;
  const int64_t nteams = (int64_t)gridDim.x * TPB;
  for (int64_t g = team; g < ngroups; g += nteams) {
    GatherGroup<TEAM> grp{sol, xk, sj, index, owner, ptr[g], ptr[g + 1], lane, (int)g};
    group_body<TEAM, BINF>(grp, y, lambda[g], sigma, delta, lds, false, pole_lit != 0);
    if constexpr (TEAM == 256) __syncthreads();
  }
}

// indices no group contains keep the caller's y (the reference never assigns them) minus the shift (:77 / :116)
__global__ __launch_bounds__(256) void k_gather_rest(double* y, const double* xk, const double* sj, const int* owner,
                                                      int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    if (owner[i] < 0) y[i] = y[i] - (xk[i] + sj[i]);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
template <bool BINF>
static int run_group(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                     const int64_t* offsets, int64_t gsize, int64_t ngroups, const double* lambda, double sigma,
                     double delta) {
  int rc = spx_check_common(ctx, y, q, xk, sj, n);
  if (rc) return rc;
  SPX_REQUIRE(ngroups >= 0, "ngroups < 0");
  if (n == 0) return SPX_OK;
  if (ngroups == 0) {  // no group at all: ShiftedGroupNormL2 still subtracts the shift everywhere (:77)
    if (!BINF && offsets) {
      SPX_ON_DEVICE(ctx);
      hipLaunchKernelGGL(k_csr_uncovered, dim3(256), dim3(256), 0, ctx->stream, y, xk, sj, offsets, ngroups, n);
      SPX_LAUNCH_CHECK();
    }
    return SPX_OK;
  }
  SPX_REQUIRE(lambda != nullptr, "lambda_vec is NULL");
  if (!offsets) {
    SPX_REQUIRE(gsize > 0, "group_size <= 0 with NULL group_offsets");
    SPX_REQUIRE(ngroups <= n / gsize && ngroups * gsize == n, "ngroups * group_size != n");
  }
  // y may alias q: every kernel finishes all reductions of a group (team barrier / wave lockstep) before
  // the group's first store, and the storing lane re-reads q[i] itself just before writing y[i].
  SPX_ON_DEVICE(ctx);
  const int64_t cap_blocks = (int64_t)ctx->num_cu * 8;
  const bool aligned = spx_aligned16(y) && spx_aligned16(q) && spx_aligned16(xk) && spx_aligned16(sj);
  const bool ragged_reg = offsets && gsize > 0 && gsize <= 512;  // ragged groups with a size bound from the caller
  if ((!offsets && gsize <= 512) || ragged_reg) {
    // register path: the smallest (LPG, EPL) tile that holds a group; partly filled tiles are padded with zeros.
    // Ragged groups (CSR offsets + an upper bound on the sizes in group_size) use the same tiles through the 8-byte
    // loads; a group that exceeds the bound after all is handed to the general kernel.
    int lpg, epl;
    // Binf: as few lanes per group as the registers allow (4 x 4/8, 8 x 8/16, 16 x 16, 32 x 16 elements) -- the wave-uniform
    // scalar work of the root find, which every lane executes, is then shared by more groups per wave
    // Small groups (round 3): tiles that FIT -- a 4 x 4 tile spent four lanes' worth of root find (Binf) or reduction on a group
    // of two, and three quarters of its loads on padding.  us per call at n = 1.6e7, old -> new tile (tools/r3/binf_small_groups.py):
    // Binf groups of 2: 1305 -> 368, of 4: 680 -> 208, of 8: 317 -> 174, of 16: 155 -> 142; plain groups of 2: 508 -> 89, of 4:
    // 283 -> 92, of 8: 162 -> 90, of 10: 132 -> 107.  (One lane per group beyond 8 elements loses more to the 64-byte strides
    // between its lanes' loads than it saves: Binf 1 x 16 on groups of 16 270 us.)
    if (BINF && gsize <= 2) { lpg = 1; epl = 2; }
    else if (BINF && gsize <= 4) { lpg = 1; epl = 4; }
    else if (BINF && gsize <= 8) { lpg = 1; epl = 8; }
    else if (BINF && gsize <= 16) { lpg = 2; epl = 8; }
    else if (BINF && gsize <= 32) { lpg = 4; epl = 8; }
    else if (BINF && gsize <= 64) { lpg = 8; epl = 8; }  // (4 x 16 is slower here: 64-byte runs per group and load)
    else if (BINF && gsize <= 128) { lpg = 8; epl = 16; }
    else if (BINF && gsize <= 256) { lpg = 16; epl = 16; }
    else if (BINF && gsize <= 512) { lpg = 32; epl = 16; }
    else if (gsize <= 2) { lpg = 1; epl = 2; }
    else if (gsize <= 4) { lpg = 2; epl = 4; }
    else if (gsize <= 12) { lpg = 4; epl = 4; }
    else if (gsize <= 32) { lpg = 16; epl = 2; }
    else if (gsize <= 64) { lpg = 16; epl = 4; }
    else if (gsize <= 128) { lpg = 16; epl = 8; }
    else if (gsize <= 256) { lpg = 32; epl = 8; }
    else if (gsize <= 384) { lpg = 64; epl = 6; }
    else { lpg = 64; epl = 8; }
    const int gpw = 64 / lpg;
    int64_t blocks = (ngroups + 4 * gpw - 1) / (4 * gpw);  // 4 waves per 256-thread block
    if (blocks > 0x7fffffff) blocks = 0x7fffffff;
    dim3 grid((unsigned)blocks), block(256);
    long long* deferred = nullptr;
    unsigned long long *dcount = nullptr, *dclear = nullptr;
    if (BINF || ragged_reg) {  // list of the groups whose bracket needs the reference's literal evaluation / oversize groups
      rc = spx_ws_reserve(ctx, (size_t)(ngroups + 1) * sizeof(long long) + 256);
      if (rc) return rc;
      deferred = reinterpret_cast<long long*>(ctx->ws);
      dcount = reinterpret_cast<unsigned long long*>(deferred);
      // Round 4: the zero-fill of the count word was a launch of its own in front of every call (a Binf call at solver sizes:
      // 17 us against 10.5 for the plain operator, tools/r4/small_latency_all.py).  Uniform Binf layouts now count in one of
      // two words of the synchronisation state (zero-initialised, never written by another operator): a call uses [set], its
      // LIT launch -- queued unconditionally behind the main one -- zeroes [set ^ 1] for the next call.  Under a stream capture
      // (one set would replay for ever) and for ragged layouts the word in front of the list and its zero-fill node stay.
      const bool graph_safe = spx_capture_check(ctx) || ctx->graph_safe;
      if (BINF && !ragged_reg && ctx->tune_fewer_launches && !graph_safe) {
        rc = spx_sync_reserve(ctx, sizeof(SpxSyncHeader));
        if (rc) return rc;
        SpxSyncHeader* hdr = reinterpret_cast<SpxSyncHeader*>(ctx->sync);
        dcount = reinterpret_cast<unsigned long long*>(&hdr->grp_deferred[ctx->grp_def_set]);
        dclear = reinterpret_cast<unsigned long long*>(&hdr->grp_deferred[ctx->grp_def_set ^ 1]);
        ctx->grp_def_set ^= 1;
      } else {
        rc = spx_zero_async(ctx, deferred, sizeof(long long)); if (rc) return rc;
      }
    }
    const bool pairs = !ragged_reg && (gsize & 1) == 0 && aligned;  // otherwise 8-byte loads
    if (!BINF && ragged_reg)  // offsets need not span 0:n (src/shiftedGroupNormL2.jl:77 runs over every index)
      hipLaunchKernelGGL(k_csr_uncovered, dim3(256), dim3(256), 0, ctx->stream, y, xk, sj, offsets, ngroups, n);
#define SPX_LAUNCH_REG(LPG, EPL)                                                                                    \
  do {                                                                                                              \
    if (pairs && gsize == (LPG) * (EPL)) {                                                                          \
      hipLaunchKernelGGL((k_group_reg<LPG, EPL, BINF, true, false, true>), grid, block, 0, ctx->stream, y, q, xk, sj, \
                         ngroups, (int)gsize, lambda, sigma, delta, deferred, (const int64_t*)nullptr, ctx->status_dev, ctx->tune_binf_literal, dcount, (unsigned long long*)nullptr); \
    } else if (pairs) {                                                                                             \
      hipLaunchKernelGGL((k_group_reg<LPG, EPL, BINF, true>), grid, block, 0, ctx->stream, y, q, xk, sj, ngroups,   \
                         (int)gsize, lambda, sigma, delta, deferred, (const int64_t*)nullptr, ctx->status_dev, ctx->tune_binf_literal, dcount, (unsigned long long*)nullptr);    \
    } else                                                                                                            \
      hipLaunchKernelGGL((k_group_reg<LPG, EPL, BINF, false>), grid, block, 0, ctx->stream, y, q, xk, sj, ngroups,  \
                         (int)gsize, lambda, sigma, delta, deferred, ragged_reg ? offsets : (const int64_t*)nullptr, ctx->status_dev, ctx->tune_binf_literal, dcount, (unsigned long long*)nullptr); \
  } while (0)
    if (lpg == 1 && epl == 2) SPX_LAUNCH_REG(1, 2);
    else if (lpg == 1 && epl == 4) SPX_LAUNCH_REG(1, 4);
    else if (lpg == 1 && epl == 8) SPX_LAUNCH_REG(1, 8);
    else if (lpg == 2 && epl == 4) SPX_LAUNCH_REG(2, 4);
    else if (lpg == 2 && epl == 8) SPX_LAUNCH_REG(2, 8);
    else if (lpg == 4 && epl == 4) SPX_LAUNCH_REG(4, 4);
    else if (lpg == 4 && epl == 8) SPX_LAUNCH_REG(4, 8);
    else if (lpg == 8 && epl == 8) { if constexpr (BINF) SPX_LAUNCH_REG(8, 8); }
    else if (lpg == 8) { if constexpr (BINF) SPX_LAUNCH_REG(8, 16); }
    else if (lpg == 16 && epl == 16) { if constexpr (BINF) SPX_LAUNCH_REG(16, 16); }
    else if (lpg == 32 && epl == 16) { if constexpr (BINF) SPX_LAUNCH_REG(32, 16); }
    else if (lpg == 16 && epl == 2) SPX_LAUNCH_REG(16, 2);
    else if (lpg == 16 && epl == 4) SPX_LAUNCH_REG(16, 4);
    else if (lpg == 16) SPX_LAUNCH_REG(16, 8);
    else if (lpg == 32) SPX_LAUNCH_REG(32, 8);
    else if (epl == 6) SPX_LAUNCH_REG(64, 6);
    else SPX_LAUNCH_REG(64, 8);
#undef SPX_LAUNCH_REG
    if constexpr (BINF) {
      if (!ragged_reg) {
        // The deferred list (usually empty: the kernel returns at once) on the same register tiles.  The literal
        // evaluation is ~60 dependent passes over a group; from memory (k_group_mem) a long list is bound by L2 misses.
        const dim3 lgrid((unsigned)(blocks < (int64_t)ctx->num_cu * 8 ? blocks : (int64_t)ctx->num_cu * 8));
#define SPX_LAUNCH_LIT(LPG, EPL)                                                                                     \
  do {                                                                                                               \
    if (pairs)                                                                                                       \
      hipLaunchKernelGGL((k_group_reg<LPG, EPL, true, true, true>), lgrid, block, 0, ctx->stream, y, q, xk, sj,      \
                         ngroups, (int)gsize, lambda, sigma, delta, deferred, (const int64_t*)nullptr, ctx->status_dev, ctx->tune_binf_literal, dcount, dclear); \
    else                                                                                                             \
      hipLaunchKernelGGL((k_group_reg<LPG, EPL, true, false, true>), lgrid, block, 0, ctx->stream, y, q, xk, sj,     \
                         ngroups, (int)gsize, lambda, sigma, delta, deferred, (const int64_t*)nullptr, ctx->status_dev, ctx->tune_binf_literal, dcount, dclear); \
  } while (0)
        // (its own tiles, by the group size: the list is short, the literal evaluation wants lanes)
        if (gsize <= 16) SPX_LAUNCH_LIT(4, 4);
        else if (gsize <= 32) SPX_LAUNCH_LIT(4, 8);
        else if (gsize <= 64) SPX_LAUNCH_LIT(8, 8);
        else if (gsize <= 128) SPX_LAUNCH_LIT(8, 16);
        else if (gsize <= 256) SPX_LAUNCH_LIT(16, 16);
        else SPX_LAUNCH_LIT(32, 16);
#undef SPX_LAUNCH_LIT
        SPX_LAUNCH_CHECK();
        return SPX_OK;
      }
    }
    if (BINF || ragged_reg) {  // usually an empty list: the kernel returns at once
      hipLaunchKernelGGL((k_group_mem<64, BINF>), dim3((unsigned)(ctx->num_cu * 2)), dim3(256), 0, ctx->stream, y, q, xk,
                         sj, n, ragged_reg ? offsets : (const int64_t*)nullptr, gsize, ngroups, lambda, sigma, delta,
                         (const long long*)deferred, ctx->status_dev, ctx->tune_binf_literal, (const int*)nullptr, (int64_t)0);
    }
    SPX_LAUNCH_CHECK();
    return SPX_OK;
  }
  // team width: wavefront per group unless groups are large on average
  if (gsize > 512 && gsize <= (BINF ? kLdsGroupMax : kLdsGroupMaxPlain)) {
    // 512 < group size (uniform) or size bound (ragged, CSR offsets): LDS-resident group per workgroup
    if (!BINF && offsets)
      hipLaunchKernelGGL(k_csr_uncovered, dim3(256), dim3(256), 0, ctx->stream, y, xk, sj, offsets, ngroups, n);
    const size_t dyn = (size_t)gsize * 2 * sizeof(double);
    // per device (the attribute belongs to the function ON the current device) and cheap: set on every call that needs it,
    // no process-wide cache that a second GPU or a second thread would find in the wrong state
    if (dyn > 48 * 1024)
      SPX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_group_lds<BINF>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kLdsGroupMax * 2 * (int)sizeof(double)));
    int64_t blocks = ngroups < cap_blocks ? ngroups : cap_blocks;
    hipLaunchKernelGGL((k_group_lds<BINF>), dim3((unsigned)blocks), dim3(256), dyn, ctx->stream, y, q, xk, sj, n, offsets,
                       gsize, ngroups, lambda, sigma, delta, ctx->tune_binf_literal);
    SPX_LAUNCH_CHECK();
    return SPX_OK;
  }
  // Large groups -- first of all ONE group over the whole vector, the reference's default GroupNormL2
  // (src/groupNormL2.jl:30-31, shifted(NormL2(lambda), xk): src/shiftedGroupNormL2.jl:34-35): a team of workgroups per
  // group (spx_group_team.hip) instead of one workgroup (n = 1e8: 384 ms plain / 1349 ms Binf that way).
  const int* big_active = nullptr;   // ragged layouts: device word of the team plan, "the large groups are taken care of"
  const int64_t big_min = (BINF ? kLdsGroupMax : kLdsGroupMaxPlain) + 1;  // ragged layouts: a group of at least this many elements is
                                                                          // a large one (what the LDS-resident kernel does not hold, as for uniform groups)
  if (ctx->tune_team) {
    if (!offsets) {
      const int tg = spx_group_team_max_grid(ctx, BINF);
      // (teams of ONE workgroup that take several groups in turn beat the one-workgroup-per-group kernels below at every count
      //  of large groups -- 1e8 elements in groups of 5000 ... 200 000: plain 0.88-1.07 -> 0.62-0.99 ms, Binf 2.0-3.4 -> 1.0-1.8 ms,
      //  tools/r4/team_crossover.py -- on chip up to 9216 elements, two passes instead of one per reduction beyond; tuning key 16
      //  = groups per workgroup up to which the team form is used, for that A/B)
      if (tg >= 1 && (ctx->tune_team_factor == 0 || ngroups < (int64_t)ctx->tune_team_factor * tg))
        return spx_group_team_launch(ctx, BINF, y, q, xk, sj, n, nullptr, gsize, ngroups, lambda, sigma, delta);
    } else if (ngroups <= 65536) {
      rc = spx_group_team_plan(ctx, BINF, y, q, xk, sj, n, offsets, ngroups, big_min, &big_active);
      if (rc) return rc;
    }
  }
  if (!BINF && offsets)
    hipLaunchKernelGGL(k_csr_uncovered, dim3(256), dim3(256), 0, ctx->stream, y, xk, sj, offsets, ngroups, n);
  const double avg = (double)n / (double)ngroups;
  if (avg <= 2048.0) {
    // lanes per group by the average size (ragged groups without a size bound from the caller, or with one above 512): a
    // whole wavefront per group of a handful of elements left most lanes idle (round 3)
    const int team = avg <= 8.0 ? 4 : avg <= 24.0 ? 8 : avg <= 64.0 ? 16 : avg <= 160.0 ? 32 : 64;
    const int tpb = 256 / team;
    int64_t blocks = (ngroups + tpb - 1) / tpb;
    if (blocks > cap_blocks) blocks = cap_blocks;
#define SPX_LAUNCH_MEM(TEAM)                                                                                              \
  hipLaunchKernelGGL((k_group_mem<TEAM, BINF>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, y, q, xk, sj, n, offsets, \
                     gsize, ngroups, lambda, sigma, delta, (const long long*)nullptr, ctx->status_dev, ctx->tune_binf_literal, \
                     big_active, big_min)
    if (team == 4) SPX_LAUNCH_MEM(4);
    else if (team == 8) SPX_LAUNCH_MEM(8);
    else if (team == 16) SPX_LAUNCH_MEM(16);
    else if (team == 32) SPX_LAUNCH_MEM(32);
    else SPX_LAUNCH_MEM(64);
#undef SPX_LAUNCH_MEM
  } else {
    int64_t blocks = ngroups < cap_blocks ? ngroups : cap_blocks;
    hipLaunchKernelGGL((k_group_mem<256, BINF>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, y, q, xk, sj, n,
                       offsets, gsize, ngroups, lambda, sigma, delta, (const long long*)nullptr, ctx->status_dev, ctx->tune_binf_literal,
                       big_active, big_min);
  }
  SPX_LAUNCH_CHECK();
  if (big_active)  // the large groups of a ragged layout (the kernel returns at once when the plan found none)
    return spx_group_team_launch(ctx, BINF, y, q, xk, sj, n, offsets, gsize, ngroups, lambda, sigma, delta);
  return SPX_OK;
}

SPX_EXPORT int spx_prox_group_l2(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                                 int64_t n, const int64_t* group_offsets, int64_t group_size, int64_t ngroups,
                                 const double* lambda_vec, double sigma) {
  return run_group<false>(ctx, y, q, xk, sj, n, group_offsets, group_size, ngroups, lambda_vec, sigma, 0.0);
}

SPX_EXPORT int spx_prox_group_l2_binf(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                                      int64_t n, const int64_t* group_offsets, int64_t group_size, int64_t ngroups,
                                      const double* lambda_vec, double sigma, double delta) {
  return run_group<true>(ctx, y, q, xk, sj, n, group_offsets, group_size, ngroups, lambda_vec, sigma, delta);
}

// ---------------------------------------------------------------------------------------------
// gather-index groups: idx_g = group_index[group_ptr[g] .. group_ptr[g+1])
// ---------------------------------------------------------------------------------------------
template <bool BINF>
static int run_group_gather(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                            const int64_t* ptr, const int64_t* index, int64_t ngroups, int64_t nnz,
                            const double* lambda, double sigma, double delta) {
  int rc = spx_check_common(ctx, y, q, xk, sj, n);
  if (rc) return rc;
  SPX_REQUIRE(ngroups >= 0 && ngroups < 0x7fffffff, "ngroups out of range");
  SPX_REQUIRE(nnz >= 0, "nnz < 0");
  if (n == 0) return SPX_OK;
  if (ngroups > 0) SPX_REQUIRE(ptr != nullptr && lambda != nullptr, "group_ptr or lambda_vec is NULL");
  if (nnz > 0) SPX_REQUIRE(index != nullptr, "group_index is NULL");
  if (ngroups > 0) {  // (the index-set validation below reads a flag back: refused before anything is enqueued)
    const int rcc = spx_require_not_capturing(ctx, "validating a group layout");
    if (rcc) return rcc;
  }
  SPX_ON_DEVICE(ctx);
  // workspace: flag (256 B) | sol (n doubles) | owner (n ints)
  const size_t sol_off = 256, own_off = sol_off + (size_t)n * sizeof(double);
  rc = spx_ws_reserve(ctx, own_off + (size_t)n * sizeof(int) + 256);
  if (rc) return rc;
  char* ws = static_cast<char*>(ctx->ws);
  int* flag = reinterpret_cast<int*>(ws);
  double* sol = reinterpret_cast<double*>(ws + sol_off);
  int* owner = reinterpret_cast<int*>(ws + own_off);
  const int64_t cap_blocks = (int64_t)ctx->num_cu * 8;
  int64_t eb = (n + 255) / 256;
  if (eb > cap_blocks) eb = cap_blocks;
  SPX_HIP(hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
  hipLaunchKernelGGL(k_gather_prepare, dim3((unsigned)eb), dim3(256), 0, ctx->stream, sol, owner, q, xk, sj, n);
  if (ngroups > 0) {
    int64_t gb = (ngroups + 3) / 4;
    if (gb > cap_blocks) gb = cap_blocks;
    hipLaunchKernelGGL(k_gather_owner, dim3((unsigned)gb), dim3(256), 0, ctx->stream, owner, ptr, index, ngroups, n, nnz,
                       flag);
    SPX_LAUNCH_CHECK();
    int hflag = 0;  // the reference throws BoundsError before touching y: check before the stores
    SPX_HIP(hipMemcpyAsync(&hflag, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    SPX_HIP(hipStreamSynchronize(ctx->stream));
    if (hflag & 2) { spx_set_error("invalid argument: group_ptr is not a non-decreasing sequence inside [0, nnz]"); return SPX_ERR_INVALID_ARG; }
    if (hflag & 1) { spx_set_error("invalid argument: group index outside [0, n) (BoundsError)"); return SPX_ERR_INVALID_ARG; }
    const double avg = (double)nnz / (double)ngroups;
    if (avg <= 2048.0) {
      const int team = avg <= 8.0 ? 4 : avg <= 24.0 ? 8 : avg <= 64.0 ? 16 : avg <= 160.0 ? 32 : 64;  // (as run_group)
      const int tpb = 256 / team;
      int64_t tb = (ngroups + tpb - 1) / tpb;
      if (tb > cap_blocks) tb = cap_blocks;
#define SPX_LAUNCH_GATHER(TEAM)                                                                                           \
  hipLaunchKernelGGL((k_group_gather<TEAM, BINF>), dim3((unsigned)tb), dim3(256), 0, ctx->stream, y, sol, xk, sj, owner, ptr, \
                     index, ngroups, lambda, sigma, delta, ctx->tune_binf_literal)
      if (team == 4) SPX_LAUNCH_GATHER(4);
      else if (team == 8) SPX_LAUNCH_GATHER(8);
      else if (team == 16) SPX_LAUNCH_GATHER(16);
      else if (team == 32) SPX_LAUNCH_GATHER(32);
      else SPX_LAUNCH_GATHER(64);
#undef SPX_LAUNCH_GATHER
    } else {
      int64_t blocks = ngroups < cap_blocks ? ngroups : cap_blocks;
      hipLaunchKernelGGL((k_group_gather<256, BINF>), dim3((unsigned)blocks), dim3(256), 0, ctx->stream, y, sol, xk, sj,
                         owner, ptr, index, ngroups, lambda, sigma, delta, ctx->tune_binf_literal);
    }
  }
  // :77 subtracts the shift at EVERY index; the Binf form does so per group (:116) and leaves the rest of y alone
  if (!BINF) hipLaunchKernelGGL(k_gather_rest, dim3((unsigned)eb), dim3(256), 0, ctx->stream, y, xk, sj, owner, n);
  SPX_LAUNCH_CHECK();
  return SPX_OK;
}

SPX_EXPORT int spx_prox_group_l2_gather(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                                        int64_t n, const int64_t* group_ptr, const int64_t* group_index,
                                        int64_t ngroups, int64_t nnz, const double* lambda_vec, double sigma) {
  return run_group_gather<false>(ctx, y, q, xk, sj, n, group_ptr, group_index, ngroups, nnz, lambda_vec, sigma, 0.0);
}

SPX_EXPORT int spx_prox_group_l2_binf_gather(spx_ctx* ctx, double* y, const double* q, const double* xk,
                                             const double* sj, int64_t n, const int64_t* group_ptr,
                                             const int64_t* group_index, int64_t ngroups, int64_t nnz,
                                             const double* lambda_vec, double sigma, double delta) {
  return run_group_gather<true>(ctx, y, q, xk, sj, n, group_ptr, group_index, ngroups, nnz, lambda_vec, sigma, delta);
}
