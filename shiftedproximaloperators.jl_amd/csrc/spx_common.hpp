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
  // Persistent device state of the multi-workgroup kernels that synchronise inside one launch (grid-barrier counters,
  // histograms that must be zero on entry): allocated and zeroed once, never shared with `ws` (which every operator
  // overwrites from offset 0).  coop_parity alternates per such launch: a launch uses counter/histogram set `parity`
  // and clears set `parity ^ 1`, which the previous launch on this stream used and has finished with.
  void* sync = nullptr;
  size_t sync_bytes = 0;
  int coop_parity = 0;
  // Graph-safe mode (sticky, set the first time a call finds its stream capturing: spx_capture_check).  The kernels that
  // synchronise inside one launch keep device state between launches (barrier counters, histogram sets, exchange words) that
  // the host tracks by alternating sets -- a captured launch would replay ONE set for ever.  In graph-safe mode every such
  // launch is preceded by a memset of exactly the state it uses (a node of the same graph) and the sets are fixed.
  int graph_safe = 0;
  int b2_set = 0;                 // ShiftedNormL1B2: the set of partial-sum words the next launch uses (spx_b2.hip, b2_put)
  int b2_dirty_g[2] = {0, 0};     // ... and how many workgroups wrote into each set (0 = clean)
  int team_set = 0;               // spx_group_team.hip: the set of exchange words the next team launch uses (it clears the other one)
  int sel_hist_next = 0;            // spx_select.hip: histogram set (0/1) the next k_sel_coop launch uses ...
  int sel_hist_dirty[2] = {0, 0};   // ... and which sets a previous launch left non-zero
  // spx_ctx_set_value_target: when non-NULL, the value-returning entry points (spx_obj_*, spx_proxval_*) store their
  // result in this DEVICE double and return after enqueueing, without the read-back / stream synchronisation
  double* value_target = nullptr;
  // tuning knobs (spx_ctx_set_tuning): per context, so that two contexts / threads never see each other's experiments
  int tune_sep_blocks_per_cu = 0;  // key 0: 0 = no cap (one tile per workgroup)
  int tune_sep_nt = 1;             // key 1: non-temporal loads / stores
  int tune_sel_fast = 1;           // key 2: sample-predicted top-r path
  int tune_sep_lds = 1;            // key 3: LDS-staged separable skeleton
  int tune_sel_spec = 1;           // key 4: single-pass (speculative store) form of the top-r fast path
  int tune_sep_xcd = 0;            // key 5: XCD-contiguous tile ranges (experiment)
  int tune_sel_small = 1;          // key 6: one-workgroup top-r for n <= 8192
  int tune_coop_cap = 0;           // key 8: pretend that only this many workgroups of an in-launch synchronised kernel are
                                   //        resident at once (0 = what the occupancy query says): exercises the smaller-grid paths
  int tune_binf_literal = 0;       // key 9: GroupNormL2Binf groups whose root sits next to the pole of step(n) (u < n/1000) take
                                   //        the reference's literal Float64 evaluation (reproduces a reference run bit pattern
                                   //        by bit pattern where the default is the more accurate side: include/spx.h)
  int tune_front_spl = 0;          // key 10: samples per lane of the top-r front kernel (1, 2, 4, 16; 0 = by n and r / n)
  int tune_sel_reg16 = 1;          // key 11: register-resident one-launch top-r at 16 elements per lane (2 Mi < n <= 4 Mi on 256 CUs)
  int tune_b2_lds = 1;             // key 12: ShiftedNormL1B2 with xk parked in LDS between the register form and the streaming form (2 Mi < n <= 4 Mi)
  int tune_team = 1;               // key 13: large contiguous groups (first of all ONE group over the whole vector) are owned by a team of
                                   //         workgroups (spx_group_team.hip); 0 = one workgroup per group as in rounds 1-3
  int tune_team_fast = 1;          // key 14: ... and the Binf form of that takes its sample-predicted two-pass path (0 = generic body: one
                                   //         streaming pass per reduction)
  int tune_team_factor = 0;        // key 16: the team form serves uniform large groups while there are fewer than this many per workgroup of its
                                   //         grid (0 = default, see run_group in spx_group.hip); an A/B knob
  int tune_fewer_launches = 1;     // key 17: psi(y) in one launch (the last workgroup finishes: spx_fin_ticket) and the Binf group operators without the
                                   //         zero-fill launch of their deferred list (count words that alternate); 0 = the launches of rounds 1-3
  int grp_def_set = 0;             // spx_group.hip: which of SpxSyncHeader::grp_deferred the next call uses (it clears the other one)
  int tune_force_grid = 0;         // key 100, test builds only (-DSPX_TEST_HOOKS): launch the one-launch top-r with THIS many workgroups,
                                   //          residency or not -- the planted fault behind tests/test_gpu_robustness.py
  int tune_force_team = 0;         // key 102, test builds only: the last workgroup of every team of k_group_team arrives late (spx_group_team.hip)
  int tune_force_tail = 0;         // key 101, test builds only: the same for the tail kernel of the sampled top-r pipeline (k_s2_tail)
  // Device-side status word in host-mapped pinned memory (spx_ctx.hip): a kernel that gives up waiting for the other
  // workgroups of its launch, or finds library-owned state outside its layout, stores a non-zero code here (system-scope
  // store); every entry point looks at it before it enqueues anything (SPX_ON_DEVICE) and fails with SPX_ERR_INTERNAL from
  // then on, until spx_sync has reported it.  No stream synchronisation and no read-back is involved.
  volatile int* status_host = nullptr;
  int* status_dev = nullptr;
  // Blocks that a captured graph may still reference (graph_safe): never freed before the context is destroyed.
  void* retired[64] = {};
  int nretired = 0;
  // ... those of them that are synchronisation-state blocks (spx_sync_reserve), with their sizes: a captured graph that timed out
  // on such a block would replay NaN for ever if spx_sync reset the current block only
  void* retired_sync[16] = {};
  size_t retired_sync_bytes[16] = {};
  int nretired_sync = 0;
  // resident workgroups per CU of the in-launch synchronised kernels (hipOccupancyMaxActiveBlocksPerMultiprocessor),
  // queried once per kernel and context: spx_resident_cap
  const void* occ_fn[32] = {};
  int occ_blocks[32] = {};
  int nocc = 0;
};

// codes of spx_ctx::status_host (bits)
constexpr int kSpxStatusTimeout = 1;    // an in-launch synchronised kernel gave up waiting for its own workgroups
constexpr int kSpxStatusCorrupt = 2;    // library-owned device state outside its layout (deferred-group list)

void spx_set_error(const char* fmt, ...);
int spx_ws_reserve(spx_ctx* ctx, size_t bytes);
int spx_sync_reserve(spx_ctx* ctx, size_t bytes);  // persistent, zero-initialised (see spx_ctx::sync)
int spx_ctx_count(int device);                     // live contexts on a device
// How many workgroups of `fn` (block_threads lanes, dyn_lds bytes of dynamic LDS) can be resident at once on the context's
// device: occupancy query x number of CUs, cached per kernel; spx_ctx_set_tuning key 8 lowers it (tests).  Every launch that
// synchronises inside itself sizes its grid with this and takes a smaller-grid form when the grid it wants does not fit --
// a workgroup must never wait for one that cannot be placed.  0 with an error set if the kernel cannot run at all.
int64_t spx_resident_cap(spx_ctx* ctx, const void* fn, int block_threads, size_t dyn_lds);
int spx_blocks_per_cu(spx_ctx* ctx, const void* fn, int block_threads, size_t dyn_lds);  // workgroups a CU holds at once; -1: error
int spx_status_report(spx_ctx* ctx);               // SPX_ERR_INTERNAL + message for a non-zero device status word
// Is ctx's stream being captured into a graph?  Sets spx_ctx::graph_safe (sticky) when it is.  Calls that would have to
// synchronise the stream or (re)allocate refuse to run while capturing (SPX_ERR_INVALID_ARG, spx_require_not_capturing).
// Zero-fill on the context's stream by a KERNEL (4-byte words): used instead of hipMemsetAsync / hipMemset2DAsync wherever
// a call may be captured into a graph -- a captured iteration replayed through torch.cuda.CUDAGraph faulted in a way that
// implicated the runtime's memset nodes (round 2; plain kernel nodes replay fine).  bytes, pitch, width: multiples of 4.
int spx_zero_async(spx_ctx* ctx, void* ptr, size_t bytes);
int spx_zero2d_async(spx_ctx* ctx, void* ptr, size_t pitch_bytes, size_t width_bytes, size_t rows);
bool spx_capture_check(spx_ctx* ctx);
int spx_require_not_capturing(spx_ctx* ctx, const char* what);

// spx_group_team.hip (large contiguous groups on teams of workgroups), called from run_group in spx_group.hip
int spx_group_team_max_grid(spx_ctx* ctx, bool binf);
int spx_group_team_plan(spx_ctx* ctx, bool binf, const double* y, const double* q, const double* xk, const double* sj,
                        int64_t n, const int64_t* offsets, int64_t ngroups, int64_t big_min, const int** active_dev);
int spx_group_team_launch(spx_ctx* ctx, bool binf, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                          const int64_t* offsets, int64_t gsize, int64_t ngroups, const double* lambda, double sigma,
                          double delta);

// Two launches that synchronise inside themselves must not run side by side on one device: each would hold CUs while it
// waits for workgroups of its own that cannot be placed.  Contexts on different streams are therefore chained through one
// event per device whenever more than one context exists on it (a stream-side dependency, the host never blocks).
// Constructed before such a launch (or a sequence that contains one), destroyed after it.  spx_ctx.hip.
struct SpxCoopLaunchGuard {
  spx_ctx* ctx;
  bool chained;
  bool capturing;
  explicit SpxCoopLaunchGuard(spx_ctx* c);
  ~SpxCoopLaunchGuard();
};

#define SPX_HIP(call)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (call);                                                                    \
    if (e_ != hipSuccess) {                                                                    \
      spx_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
      return SPX_ERR_HIP;                                                                      \
    }                                                                                          \
  } while (0)

// Entry points run on the context's device and leave the caller's current device as they found it (an array library
// working on another GPU must not have its current device changed under it).
struct SpxDeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit SpxDeviceGuard(int device) {
    if (hipGetDevice(&prev) != hipSuccess) prev = -1;
    if (prev != device) err = hipSetDevice(device);
  }
  ~SpxDeviceGuard() {
    int cur = -1;
    if (prev >= 0 && hipGetDevice(&cur) == hipSuccess && cur != prev) (void)hipSetDevice(prev);
  }
};
#define SPX_ON_DEVICE_RAW(ctx)                  \
  SpxDeviceGuard spx_device_guard((ctx)->device); \
  SPX_HIP(spx_device_guard.err)
// ... and refuse to enqueue anything on a context whose device side has reported a failure (spx_ctx::status_host)
#define SPX_ON_DEVICE(ctx)                                                       \
  if ((ctx)->status_host != nullptr && *(ctx)->status_host != 0) return spx_status_report(ctx); \
  SPX_ON_DEVICE_RAW(ctx)

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
// the same on Float32 (overloads in the SAME scope as the Float64 forms: a float overload declared in an inner namespace would
// hide these and silently round every double argument to float)
__device__ __forceinline__ float jl_min(float x, float y) {
  const float d = x - y;
  const float a = (__float_as_int(d) < 0) ? x : y;
  return (x != x || y != y) ? d : a;
}
__device__ __forceinline__ float jl_max(float x, float y) {
  const float d = x - y;
  const float a = (__float_as_int(d) < 0) ? y : x;
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

// A double moved between lanes by DPP (row operations: quad_perm, row_half_mirror, row_mirror -- inside a 16-lane row)
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
// sum / max over the 64 lanes of a wavefront, result in every lane: DPP inside the 16-lane rows, two permutes across them
// (the xor butterfly of wave_sum is six permutes per double: the reductions of a root find are latency, not bandwidth)
__device__ __forceinline__ double wave_sum_dpp(double v) {
  v += dpp_f64<0xB1>(v);
  v += dpp_f64<0x4E>(v);
  v += dpp_f64<0x141>(v);
  v += dpp_f64<0x140>(v);
  v += __shfl_xor(v, 16, 64);
  v += __shfl_xor(v, 32, 64);
  return v;
}
__device__ __forceinline__ double wave_max_dpp(double v) {
  v = fmax(v, dpp_f64<0xB1>(v));
  v = fmax(v, dpp_f64<0x4E>(v));
  v = fmax(v, dpp_f64<0x141>(v));
  v = fmax(v, dpp_f64<0x140>(v));
  v = fmax(v, __shfl_xor(v, 16, 64));
  v = fmax(v, __shfl_xor(v, 32, 64));
  return v;
}
// sum over the 16 lanes of a row, result in every lane of the row (a fixed shape: the same bits in every row that holds the same 16 values)
__device__ __forceinline__ double fold16_sum(double v) {
  v += dpp_f64<0xB1>(v);
  v += dpp_f64<0x4E>(v);
  v += dpp_f64<0x141>(v);
  v += dpp_f64<0x140>(v);
  return v;
}

// spx_ctx::sync: [0, kSpxSyncSelBytes) belongs to spx_select.hip (SelSync, whose head is the SpxSyncHeader below), the
// partial-sum words of spx_b2.hip follow.  Zero-filled when (re)allocated; the host-side flags are reset with it.
constexpr size_t kSpxSyncSelBytes = (size_t)2 << 20;  // 2 MiB >= sizeof(SelSync) (static_assert in spx_select.hip)
constexpr size_t kSpxSyncB2Bytes = (size_t)2 << 20;   // the exchange words of spx_b2.hip (static_assert there)
constexpr size_t kSpxSyncTeamOffset = kSpxSyncSelBytes + kSpxSyncB2Bytes;  // ... and behind them those of spx_group_team.hip:
constexpr size_t kSpxSyncTeamBytes = ((size_t)512 << 10) + 4096;           // two sets of kGtSetWords words + two sets of tile counters (static_assert in spx_group_common.hpp)

// Head of spx_ctx::sync, shared by every kernel that synchronises inside one launch.
constexpr int kSpxBarSplit = 8;  // arrival counters per grid barrier (see spx_grid_rendezvous)
struct SpxSyncHeader {
  unsigned int bar[2][kSpxBarSplit * 32];  // grid-barrier counters, kSpxBarSplit 128-byte lines per set: a launch uses [parity] and clears [parity ^ 1]
  int b2_last_scaled;       // ShiftedNormL1B2: did the previous call on this context find the trust region active?
  int timed_out;            // sticky: a workgroup gave up waiting for the others (kSpxPollLimit); no workgroup waits any more
  int* status;              // device view of spx_ctx::status_host (host-mapped): the failure is reported by the NEXT libspx call
  // Round 4 (launch counts at solver sizes: a call below ~1e6 elements costs what its launches cost):
  long long grp_deferred[2];  // count words of the Binf group operators' deferred list: a call uses [set] and clears [set ^ 1] (spx_group.hip)
  unsigned int fin_top;       // the objective kernels' "last workgroup" tickets (spx_fin_ticket below); zero between launches
  int fin_flag;               // ... and their infeasibility bits; zero between launches
  int pad[22];
  unsigned int fin_class[kSpxBarSplit * 32];  // first-level tickets, one 128-byte line each
};
static_assert(sizeof(SpxSyncHeader) == 2 * kSpxBarSplit * 32 * 4 + 128 + kSpxBarSplit * 32 * 4, "SpxSyncHeader layout");

// Every wait of one workgroup for others is bounded: kSpxPollLimit polls (each a memory round trip plus a short sleep: a few
// seconds in all, against microseconds of legitimate waiting).  A workgroup that gives up sets SpxSyncHeader::timed_out,
// after which no workgroup of that context waits any more -- the grid drains instead of hanging the device, the kernels
// that store a result store NaN (spx_poisoned) -- and raises the context's host-mapped status word, which makes the next
// libspx call on the context (any entry point, not only spx_sync) fail with SPX_ERR_INTERNAL.  It happens when the
// workgroups of the launch are not all resident (a grid of another process or of a graph replay holding the CUs: the
// in-process cases are excluded by spx_resident_cap and SpxCoopLaunchGuard), when the synchronisation state is corrupt (a
// context shared by two threads, device memory overwritten) -- or in tools/planted_faults.sh.
#ifndef SPX_POLL_LIMIT_LOG2
#define SPX_POLL_LIMIT_LOG2 22
#endif
constexpr unsigned int kSpxPollLimit = 1u << SPX_POLL_LIMIT_LOG2;
__device__ __forceinline__ void spx_raise_status(int* status, int code) {
  if (status != nullptr) __hip_atomic_store(status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ bool spx_wait_expired(unsigned int& spins, SpxSyncHeader* hdr) {
  if (++spins < 64u) return false;                       // (cheap path: the flag is not even read for short waits)
  if ((spins & 63u) != 0u && spins < kSpxPollLimit) return false;
  if (spins >= kSpxPollLimit) {
    __hip_atomic_store(&hdr->timed_out, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    spx_raise_status(hdr->status, kSpxStatusTimeout);
    return true;
  }
  return __hip_atomic_load(&hdr->timed_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}
// Has any workgroup of this context given up?  Asked once, before the result of an in-launch synchronised kernel is stored.
__device__ __forceinline__ bool spx_poisoned(SpxSyncHeader* hdr) {
  return __hip_atomic_load(&hdr->timed_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
}

// ---------------------------------------------------------------------------------------------
// Grid barrier for kernels whose workgroups are all resident (grid <= number of CUs, one workgroup per CU): monotonic
// counters in spx_ctx::sync.  Recipe of MI355X_MICROARCH.md ("inter-workgroup visibility"): every storing wave drains its
// stores, workgroup barrier, wave 0: agent-scope release (L2 write-back) -> arrive (agent-scope atomic add) -> relaxed polls
// -> agent-scope acquire (L1 / non-coherent L2 lines invalidated), workgroup barrier, then plain loads.
// `target` = (number of barriers passed so far in this launch + 1) * gridDim.x.  Every workgroup of the grid must call
// it the same number of times (the exit condition every wave reaches: no early return between barriers).
// Arrivals are spread over kSpxBarSplit counters (workgroup b adds to counter b % 8, each on its own 128-byte line) and
// lanes 0..7 of wave 0 poll one counter each: atomics on ONE address retire ~12 ns apart and the polls of 256 workgroups
// queue in front of the last arrivals -- measured without fences (tools/exp/grid_barrier.hip): 3.75 -> 1.44 us per
// rendezvous with 256 workgroups, 2.0 -> 1.2 with 128, 1.3 -> 1.1 with 64 (16 counters: 2.2 / 1.2 / 1.1; a two-level tree 2.4).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void spx_bar_arrive_and_wait(unsigned int* counters, unsigned int target, SpxSyncHeader* hdr) {
  // (wave 0, all 64 lanes)
  const unsigned int lane = threadIdx.x;
  const unsigned int phase = target / gridDim.x;
  if (lane == 0) __hip_atomic_fetch_add(&counters[32u * (blockIdx.x % (unsigned)kSpxBarSplit)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // counter j hears from the workgroups j, j + 8, ...
  const unsigned int want = (lane < (unsigned)kSpxBarSplit && lane < gridDim.x)
                                ? ((gridDim.x - lane + (unsigned)kSpxBarSplit - 1u) / (unsigned)kSpxBarSplit) * phase : 0u;
  unsigned int spins = 0;
  while (true) {
    const unsigned int have = lane < (unsigned)kSpxBarSplit ? __hip_atomic_load(&counters[32u * lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
    if (__ballot(have < want) == 0ull) break;
    if (spx_wait_expired(spins, hdr)) break;  // (the same in every lane)
    __builtin_amdgcn_s_sleep(1);
  }
}
// The counters of the set a LATER launch will use are cleared by workgroup 0 of this one.
__device__ __forceinline__ void spx_bar_reset(unsigned int* counters) {
  if (blockIdx.x == 0 && threadIdx.x < (unsigned)kSpxBarSplit) counters[32u * threadIdx.x] = 0u;
}
__device__ __forceinline__ void spx_grid_barrier(unsigned int* counters, unsigned int target, SpxSyncHeader* hdr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x < 64) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    spx_bar_arrive_and_wait(counters, target, hdr);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

// The same rendezvous WITHOUT cache maintenance, for kernels whose workgroups exchange nothing but words written and read
// with agent-scope atomics (spx_atomic_store_f64 / spx_atomic_load_f64 below: `sc1` accesses that bypass the non-coherent
// caches).  The release / acquire fences are most of a barrier's cost when few workgroups meet (3.5 of ~4 us).
// Every wave must have waited for its own atomic stores (s_waitcnt vmcnt(0)) before it calls this.
__device__ __forceinline__ void spx_grid_rendezvous(unsigned int* counters, unsigned int target, SpxSyncHeader* hdr) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's atomic stores have left
  __syncthreads();
  if (threadIdx.x < 64) spx_bar_arrive_and_wait(counters, target, hdr);
  __syncthreads();
}
__device__ __forceinline__ void spx_atomic_store_f64(double* p, double v) {
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                     __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double spx_atomic_load_f64(const double* p) {
  return __longlong_as_double((long long)__hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT));
}

// "Am I the last workgroup of this launch?" -- for kernels that end in a small serial step (the ordered sum of the workgroups'
// partial results) which used to be a launch of its own.  Called by ONE lane per workgroup, after that lane has waited for the
// agent-scope atomic stores of everything the last workgroup will read (s_waitcnt vmcnt(0)); the workgroup that gets `true`
// reads those words with agent-scope atomic loads.  Two levels: workgroup b takes a ticket on counter b % 8 (a 128-byte line
// each: returning atomics on ONE address retire ~12 ns apart, 2048 workgroups ending together would queue for 25 us), the last
// of a class takes one on the top counter.  Every counter is reset by the lane that takes its last ticket, so the words are
// zero between launches -- nothing to clear in front of a launch, nothing that alternates: the same node replays in a graph.
__device__ __forceinline__ bool spx_fin_ticket(SpxSyncHeader* hdr) {
  const unsigned int grid = gridDim.x, j = blockIdx.x % (unsigned)kSpxBarSplit;
  const unsigned int want = (grid - j + (unsigned)kSpxBarSplit - 1u) / (unsigned)kSpxBarSplit;  // workgroups j, j + 8, ...
  const unsigned int c = __hip_atomic_fetch_add(&hdr->fin_class[32u * j], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (c + 1u != want) return false;
  __hip_atomic_store(&hdr->fin_class[32u * j], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const unsigned int classes = grid < (unsigned)kSpxBarSplit ? grid : (unsigned)kSpxBarSplit;
  const unsigned int t = __hip_atomic_fetch_add(&hdr->fin_top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (t + 1u != classes) return false;
  __hip_atomic_store(&hdr->fin_top, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return true;
}
