// spx_ctx.hip -- context, errors, stopwatch and the construction-time helpers of the C ABI.
#include "spx_common.hpp"

#include <atomic>
#include <cstring>
#include <mutex>
static thread_local char g_err[512] = "";
static std::atomic<int> g_ctx_count[64];
int spx_ctx_count(int device) { return g_ctx_count[device & 63].load(); }

static std::mutex g_coop_mu;
static hipEvent_t g_coop_ev[64] = {};
static const spx_ctx* g_coop_last[64] = {};
SpxCoopLaunchGuard::SpxCoopLaunchGuard(spx_ctx* c) : ctx(c), chained(false), capturing(false) {
  g_coop_mu.lock();
  const int d = ctx->device & 63;
  // While the stream is being captured nothing is chained: an event recorded outside the capture cannot be waited for inside
  // it (and one recorded inside belongs to the graph).  A graph that holds such launches must not be replayed while another
  // context runs in-launch synchronised kernels on the same device (include/spx.h, "Stream capture").
  capturing = spx_capture_check(ctx);
  if (capturing) return;
  chained = (g_coop_last[d] != nullptr && g_coop_last[d] != ctx);
  if (chained && g_coop_ev[d] != nullptr) (void)hipStreamWaitEvent(ctx->stream, g_coop_ev[d], 0);
}
SpxCoopLaunchGuard::~SpxCoopLaunchGuard() {
  const int d = ctx->device & 63;
  if (capturing) {
    g_coop_mu.unlock();
    return;
  }
  // recorded after EVERY guarded eager launch (~1 us): a context created later, while this one's launch is still in flight,
  // then waits for this launch and not for some earlier one (round 2 recorded only when a second context already existed)
  if (g_coop_ev[d] == nullptr) (void)hipEventCreateWithFlags(&g_coop_ev[d], hipEventDisableTiming);
  if (g_coop_ev[d] != nullptr) (void)hipEventRecord(g_coop_ev[d], ctx->stream);
  g_coop_last[d] = ctx;
  g_coop_mu.unlock();
}

void spx_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

SPX_EXPORT int spx_abi_version(void) { return SPX_ABI_VERSION; }
SPX_EXPORT const char* spx_last_error(void) { return g_err; }

static int ctx_create_impl(int device, bool borrow, void* stream, spx_ctx** out) {
  SPX_REQUIRE(out != nullptr, "out is NULL");
  *out = nullptr;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 0) {
    spx_set_error("no HIP device visible; libspx has no CPU path");
    return SPX_ERR_NO_DEVICE;
  }
  SPX_REQUIRE(device >= 0 && device < count, "device ordinal out of range");
  SpxDeviceGuard dev_guard(device);
  SPX_HIP(dev_guard.err);
  hipDeviceProp_t prop;
  SPX_HIP(hipGetDeviceProperties(&prop, device));
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {  // the code object holds gfx950 ISA only
    spx_set_error("device %d is %s, not gfx950 (MI355X): libspx is built for gfx950 only and has no other path", device,
                  prop.gcnArchName);
    return SPX_ERR_NO_DEVICE;
  }
  spx_ctx* c = new spx_ctx();
  c->device = device;
  c->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  if (borrow) {
    c->stream = reinterpret_cast<hipStream_t>(stream);
    c->owns_stream = false;
  } else {
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
      delete c;
      spx_set_error("hipStreamCreateWithFlags failed: %s", hipGetErrorString(e));
      return SPX_ERR_HIP;
    }
    c->owns_stream = true;
  }
  if (hipEventCreate(&c->ev_start) != hipSuccess || hipEventCreate(&c->ev_stop) != hipSuccess) {
    spx_set_error("hipEventCreate failed");
    delete c;
    return SPX_ERR_HIP;
  }
  {  // the device-side status word: host-mapped pinned memory, so that the host sees a kernel's report without any copy
    void* hp = nullptr;
    void* dp = nullptr;
    if (hipHostMalloc(&hp, 64, hipHostMallocMapped) != hipSuccess || hipHostGetDevicePointer(&dp, hp, 0) != hipSuccess) {
      spx_set_error("hipHostMalloc(mapped) for the status word failed");
      if (hp) (void)hipHostFree(hp);
      (void)hipEventDestroy(c->ev_start);
      (void)hipEventDestroy(c->ev_stop);
      if (c->owns_stream) (void)hipStreamDestroy(c->stream);
      delete c;
      return SPX_ERR_ALLOC;
    }
    std::memset(hp, 0, 64);
    c->status_host = static_cast<volatile int*>(hp);
    c->status_dev = static_cast<int*>(dp);
  }
  g_ctx_count[device & 63].fetch_add(1);
  *out = c;
  return SPX_OK;
}

SPX_EXPORT int spx_ctx_create(int device, spx_ctx** out) { return ctx_create_impl(device, false, nullptr, out); }
SPX_EXPORT int spx_ctx_create_on_stream(int device, void* stream, spx_ctx** out) {
  return ctx_create_impl(device, true, stream, out);
}

SPX_EXPORT int spx_ctx_destroy(spx_ctx* ctx) {
  if (!ctx) return SPX_OK;
  SpxDeviceGuard dev_guard(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->ws) (void)hipFree(ctx->ws);
  if (ctx->sync) (void)hipFree(ctx->sync);
  if (ctx->stage) (void)hipFree(ctx->stage);
  for (int k = 0; k < ctx->nretired; ++k) (void)hipFree(ctx->retired[k]);
  if (ctx->status_host) (void)hipHostFree(const_cast<int*>(ctx->status_host));
  if (ctx->ev_start) (void)hipEventDestroy(ctx->ev_start);
  if (ctx->ev_stop) (void)hipEventDestroy(ctx->ev_stop);
  if (ctx->owns_stream) (void)hipStreamDestroy(ctx->stream);
  g_ctx_count[ctx->device & 63].fetch_sub(1);
  {
    std::lock_guard<std::mutex> lk(g_coop_mu);
    if (g_coop_last[ctx->device & 63] == ctx) g_coop_last[ctx->device & 63] = nullptr;
  }
  delete ctx;
  return SPX_OK;
}

// Per-context tuning knobs of the kernel benchmarks and A/B tests (not part of the reference's interface).  Never changes
// results, only which of several equivalent kernels runs.
SPX_EXPORT int spx_ctx_set_tuning(spx_ctx* ctx, int key, int value) {
  SPX_REQUIRE(ctx != nullptr, "ctx is NULL");
  switch (key) {
    case 0: if (value < 0 || value > 1024) break; ctx->tune_sep_blocks_per_cu = value; return SPX_OK;
    case 1: ctx->tune_sep_nt = value ? 1 : 0; return SPX_OK;
    case 2: ctx->tune_sel_fast = value ? 1 : 0; return SPX_OK;
    case 3: ctx->tune_sep_lds = value ? 1 : 0; return SPX_OK;
    case 4: ctx->tune_sel_spec = value ? 1 : 0; return SPX_OK;
    case 5: ctx->tune_sep_xcd = value ? 1 : 0; return SPX_OK;
    case 6: ctx->tune_sel_small = value ? 1 : 0; return SPX_OK;
    case 8: if (value < 0) break; ctx->tune_coop_cap = value; return SPX_OK;
    case 9: ctx->tune_binf_literal = value ? 1 : 0; return SPX_OK;
    case 10: if (value != 0 && value != 1 && value != 2 && value != 4 && value != 16) break; ctx->tune_front_spl = value; return SPX_OK;
    case 11: if (value < 0 || value > 2) break; ctx->tune_sel_reg16 = value; return SPX_OK;  // (2: v in LDS up to 4 Mi, no register slots beyond)
    case 12: ctx->tune_b2_lds = value ? 1 : 0; return SPX_OK;
    case 13: ctx->tune_team = value ? 1 : 0; return SPX_OK;
    case 14: ctx->tune_team_fast = value ? 1 : 0; return SPX_OK;
    case 16: if (value < 0 || value > 1000000) break; ctx->tune_team_factor = value; return SPX_OK;
    case 17: ctx->tune_fewer_launches = value ? 1 : 0; return SPX_OK;
#ifdef SPX_TEST_HOOKS
    case 100: if (value < 0 || value > 65535) break; ctx->tune_force_grid = value; return SPX_OK;
    case 101: if (value < 0 || value > 65535) break; ctx->tune_force_tail = value; return SPX_OK;
    case 102: if (value < 0 || value > 65535) break; ctx->tune_force_team = value; return SPX_OK;
#endif
    default: break;
  }
  spx_set_error("invalid argument: unknown tuning key/value");
  return SPX_ERR_INVALID_ARG;
}

SPX_EXPORT int spx_ctx_set_value_target(spx_ctx* ctx, double* device_value) {
  SPX_REQUIRE(ctx != nullptr, "ctx is NULL");
  ctx->value_target = device_value;
  return SPX_OK;
}

// The message for a raised status word; the word stays raised (every entry point keeps failing) until spx_sync has run.
int spx_status_report(spx_ctx* ctx) {
  const int code = ctx->status_host ? *ctx->status_host : 0;
  if (code & kSpxStatusTimeout)
    spx_set_error("internal error: a kernel that synchronises inside one launch timed out waiting for its own workgroups (are "
                  "two processes, or a graph replay and an eager call, running such kernels on this GPU at once? is the "
                  "context shared by two threads?); the results of this context since the failing call are undefined (the "
                  "failing kernel stored NaN); call spx_sync to acknowledge and reset");
  else
    spx_set_error("internal error: library-owned device state was found outside its layout (status %d: deferred-group list); "
                  "the results of this context since the failing call are undefined; call spx_sync to acknowledge and reset", code);
  return SPX_ERR_INTERNAL;
}

SPX_EXPORT int spx_sync(spx_ctx* ctx) {
  SPX_REQUIRE(ctx != nullptr, "ctx is NULL");
  SPX_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->status_host != nullptr && *ctx->status_host != 0) {  // reported here once, then the context is usable again
    const int rc = spx_status_report(ctx);
    SPX_ON_DEVICE_RAW(ctx);
    if (ctx->sync) {  // counters, histograms and exchange words of an abandoned launch: back to the initial state
      SPX_HIP(hipMemset(ctx->sync, 0, ctx->sync_bytes));
      SPX_HIP(hipMemcpy(&reinterpret_cast<SpxSyncHeader*>(ctx->sync)->status, &ctx->status_dev, sizeof(int*), hipMemcpyHostToDevice));
      ctx->coop_parity = 0;
      ctx->b2_set = 0;
      ctx->b2_dirty_g[0] = ctx->b2_dirty_g[1] = 0;
      ctx->team_set = 0;
  ctx->grp_def_set = 0;
      ctx->grp_def_set = 0;
      ctx->sel_hist_next = 0;
      ctx->sel_hist_dirty[0] = ctx->sel_hist_dirty[1] = 0;
    }
    // ... and the retired blocks a captured graph may still replay on (their sticky timed_out flag, counters, words)
    for (int k = 0; k < ctx->nretired_sync; ++k) {
      SPX_HIP(hipMemset(ctx->retired_sync[k], 0, ctx->retired_sync_bytes[k]));
      SPX_HIP(hipMemcpy(&reinterpret_cast<SpxSyncHeader*>(ctx->retired_sync[k])->status, &ctx->status_dev, sizeof(int*), hipMemcpyHostToDevice));
    }
    *ctx->status_host = 0;
    return rc;
  }
  return SPX_OK;
}

// workgroups of `fn` a CU holds at once (hipOccupancyMaxActiveBlocksPerMultiprocessor, asked once per kernel and context);
// -1 after an error (spx_set_error has the text)
int spx_blocks_per_cu(spx_ctx* ctx, const void* fn, int block_threads, size_t dyn_lds) {
  for (int k = 0; k < ctx->nocc; ++k)
    if (ctx->occ_fn[k] == fn) return ctx->occ_blocks[k];
  int nb = 0;
  const hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, block_threads, dyn_lds);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    spx_set_error("hipOccupancyMaxActiveBlocksPerMultiprocessor failed: %s", hipGetErrorString(e));
    return -1;
  }
  if (ctx->nocc < 32) { ctx->occ_fn[ctx->nocc] = fn; ctx->occ_blocks[ctx->nocc] = nb; ++ctx->nocc; }
  return nb;
}

int64_t spx_resident_cap(spx_ctx* ctx, const void* fn, int block_threads, size_t dyn_lds) {
  const int per_cu = spx_blocks_per_cu(ctx, fn, block_threads, dyn_lds);
  if (per_cu < 0) return 0;
  // The algorithms never want more than one workgroup per CU, and the query is known to promise one block per CU too many
  // for some register counts (MI355X_MICROARCH.md, "Residency and cooperative launch"): count ONE per CU, or none.
  int64_t cap = per_cu >= 1 ? (int64_t)ctx->num_cu : 0;
  if (ctx->tune_coop_cap > 0 && cap > ctx->tune_coop_cap) cap = ctx->tune_coop_cap;
  if (cap <= 0) spx_set_error("internal error: an in-launch synchronised kernel cannot be resident on this device (occupancy 0)");
  return cap;
}

SPX_EXPORT int spx_timer_start(spx_ctx* ctx) {
  SPX_REQUIRE(ctx != nullptr, "ctx is NULL");
  SPX_HIP(hipEventRecord(ctx->ev_start, ctx->stream));
  return SPX_OK;
}

SPX_EXPORT int spx_timer_stop(spx_ctx* ctx, float* elapsed_ms) {
  SPX_REQUIRE(ctx != nullptr && elapsed_ms != nullptr, "ctx or elapsed_ms is NULL");
  SPX_HIP(hipEventRecord(ctx->ev_stop, ctx->stream));
  SPX_HIP(hipEventSynchronize(ctx->ev_stop));
  SPX_HIP(hipEventElapsedTime(elapsed_ms, ctx->ev_start, ctx->ev_stop));
  return SPX_OK;
}

__global__ __launch_bounds__(256) void k_zero_words(unsigned int* p, size_t pitch_words, size_t width_words, size_t rows) {
  const size_t total = width_words * rows;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    p[(i / width_words) * pitch_words + (i % width_words)] = 0u;
}
int spx_zero2d_async(spx_ctx* ctx, void* ptr, size_t pitch_bytes, size_t width_bytes, size_t rows) {
  SPX_REQUIRE(ptr != nullptr && (reinterpret_cast<uintptr_t>(ptr) & 3u) == 0 && (pitch_bytes & 3u) == 0 && (width_bytes & 3u) == 0 &&
              width_bytes <= pitch_bytes, "spx_zero2d_async: unaligned range");
  const size_t total = (width_bytes / 4) * rows;
  if (total == 0) return SPX_OK;
#ifdef SPX_GRAPH_MEMSET_NODES  // diagnostic builds only (tools/r3/graph_fault_probe.py): the runtime's memset nodes, as in round 2
  if (rows == 1 || width_bytes == pitch_bytes) SPX_HIP(hipMemsetAsync(ptr, 0, pitch_bytes * (rows - 1) + width_bytes, ctx->stream));
  else SPX_HIP(hipMemset2DAsync(ptr, pitch_bytes, 0, width_bytes, rows, ctx->stream));
  return SPX_OK;
#endif
  size_t blocks = (total + 255) / 256;
  if (blocks > (size_t)ctx->num_cu * 8) blocks = (size_t)ctx->num_cu * 8;
  hipLaunchKernelGGL(k_zero_words, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, static_cast<unsigned int*>(ptr),
                     pitch_bytes / 4, width_bytes / 4, rows);
  SPX_LAUNCH_CHECK();
  return SPX_OK;
}
int spx_zero_async(spx_ctx* ctx, void* ptr, size_t bytes) { return spx_zero2d_async(ctx, ptr, bytes, bytes, 1); }

bool spx_capture_check(spx_ctx* ctx) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(ctx->stream, &st) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  if (st != hipStreamCaptureStatusNone) ctx->graph_safe = 1;
  return st != hipStreamCaptureStatusNone;
}
int spx_require_not_capturing(spx_ctx* ctx, const char* what) {
  if (!spx_capture_check(ctx)) return SPX_OK;
  spx_set_error("invalid argument: %s is not possible while the stream is being captured into a graph "
                "(run the same call once before capturing; keep values on the device with spx_ctx_set_value_target)", what);
  return SPX_ERR_INVALID_ARG;
}

// Grows the context scratch.  Called from entry points BEFORE any launch of that call; growing
// synchronises the stream (the old block may still be in use by earlier calls).
int spx_ws_reserve(spx_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->ws_bytes) return SPX_OK;
  { const int rc = spx_require_not_capturing(ctx, "growing the workspace"); if (rc) return rc; }
  SPX_ON_DEVICE_RAW(ctx);
  SPX_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->ws) {
    // A graph captured from this context holds the old block's address in its kernel nodes: once the context has seen a
    // capture (graph_safe, sticky) old blocks are kept until spx_ctx_destroy, so that a replay after this call still
    // finds its own scratch (it never needed more than the old size).
    if (ctx->graph_safe) {
      if (ctx->nretired >= 64) {
        spx_set_error("invalid argument: the workspace of a context that has been captured into a graph cannot grow any further "
                      "(64 retired blocks); create a new context for the larger problem");
        return SPX_ERR_INVALID_ARG;
      }
      ctx->retired[ctx->nretired++] = ctx->ws;
    } else {
      SPX_HIP(hipFree(ctx->ws));
    }
  }
  ctx->ws = nullptr;
  ctx->ws_bytes = 0;
  hipError_t e = hipMalloc(&ctx->ws, bytes);
  if (e != hipSuccess) {
    spx_set_error("workspace hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return SPX_ERR_ALLOC;
  }
  ctx->ws_bytes = bytes;
  return SPX_OK;
}

// Persistent zero-initialised device state (spx_ctx::sync).  Growing it re-zeroes everything: callers only rely on
// "zero when no launch of mine is in flight", which holds again after the stream has been drained.
int spx_sync_reserve(spx_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->sync_bytes) return SPX_OK;
  { const int rc = spx_require_not_capturing(ctx, "growing the synchronisation state"); if (rc) return rc; }
  SPX_ON_DEVICE_RAW(ctx);
  SPX_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->sync) {
    if (ctx->graph_safe) {  // (as spx_ws_reserve: a captured graph may still point into the old block)
      if (ctx->nretired >= 64) {
        spx_set_error("invalid argument: the synchronisation state of a captured context cannot grow any further");
        return SPX_ERR_INVALID_ARG;
      }
      ctx->retired[ctx->nretired++] = ctx->sync;
      if (ctx->nretired_sync < 16) {
        ctx->retired_sync[ctx->nretired_sync] = ctx->sync;
        ctx->retired_sync_bytes[ctx->nretired_sync++] = ctx->sync_bytes;
      }
    } else {
      SPX_HIP(hipFree(ctx->sync));
    }
  }
  ctx->sync = nullptr;
  ctx->sync_bytes = 0;
  hipError_t e = hipMalloc(&ctx->sync, bytes);
  if (e != hipSuccess) {
    spx_set_error("sync-state hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return SPX_ERR_ALLOC;
  }
  SPX_HIP(hipMemsetAsync(ctx->sync, 0, bytes, ctx->stream));
  SPX_HIP(hipMemcpyAsync(&reinterpret_cast<SpxSyncHeader*>(ctx->sync)->status, &ctx->status_dev, sizeof(int*), hipMemcpyHostToDevice, ctx->stream));
  SPX_HIP(hipStreamSynchronize(ctx->stream));
  ctx->sync_bytes = bytes;
  ctx->coop_parity = 0;
  ctx->b2_set = 0;
  ctx->b2_dirty_g[0] = ctx->b2_dirty_g[1] = 0;
  ctx->team_set = 0;
  ctx->grp_def_set = 0;
  ctx->sel_hist_next = 0;
  ctx->sel_hist_dirty[0] = ctx->sel_hist_dirty[1] = 0;
  return SPX_OK;
}

// ---------------------------------------------------------------------------------------------
// any(l .> u)      src/shiftedNormL1Box.jl:33-35, src/shiftedNormL0Box.jl:33-35
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_check_bounds(const double* __restrict__ l, const double* __restrict__ u,
                                                       double ls, double us, int64_t n, int* __restrict__ flag) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  bool bad = false;
  for (; i < n; i += stride) {
    double li = l ? l[i] : ls;
    double ui = u ? u[i] : us;
    bad |= (li > ui);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

SPX_EXPORT int spx_check_bounds(spx_ctx* ctx, const double* l_vec, const double* u_vec, double l_scalar,
                                double u_scalar, int64_t n, int* any_l_gt_u) {
  SPX_REQUIRE(ctx != nullptr && any_l_gt_u != nullptr, "ctx or result pointer is NULL");
  SPX_REQUIRE(n >= 0, "n < 0");
  *any_l_gt_u = 0;
  if (!l_vec && !u_vec) {  // both scalar: `any(l .> u)` on two scalars
    *any_l_gt_u = (l_scalar > u_scalar) ? 1 : 0;
    return SPX_OK;
  }
  if (n == 0) return SPX_OK;
  int rc = spx_require_not_capturing(ctx, "spx_check_bounds (synchronous)");  // (before anything is enqueued)
  if (rc) return rc;
  rc = spx_ws_reserve(ctx, 256);
  if (rc) return rc;
  SPX_ON_DEVICE(ctx);
  int* flag = reinterpret_cast<int*>(ctx->ws);
  SPX_HIP(hipMemsetAsync(flag, 0, sizeof(int), ctx->stream));
  int64_t blocks = (n + 255) / 256;
  if (blocks > 8 * (int64_t)ctx->num_cu) blocks = 8 * (int64_t)ctx->num_cu;
  hipLaunchKernelGGL(k_check_bounds, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, l_vec, u_vec, l_scalar,
                     u_scalar, n, flag);
  SPX_LAUNCH_CHECK();
  SPX_HIP(hipMemcpyAsync(any_l_gt_u, flag, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
  SPX_HIP(hipStreamSynchronize(ctx->stream));
  return SPX_OK;
}

// ---------------------------------------------------------------------------------------------
// selected -> byte mask    (`i in psi.selected`, src/shiftedNormL1Box.jl:106)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scatter_mask(uint8_t* __restrict__ mask, int64_t n,
                                                       const int64_t* __restrict__ sel, int64_t nsel) {
  int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; k < nsel; k += stride) {
    int64_t i = sel[k];
    if (i >= 0 && i < n) mask[i] = 1;  // duplicates write the same byte
  }
}

SPX_EXPORT int spx_build_mask(spx_ctx* ctx, uint8_t* mask, int64_t n, const int64_t* selected, int64_t nsel) {
  SPX_REQUIRE(ctx != nullptr, "ctx is NULL");
  SPX_REQUIRE(n >= 0 && nsel >= 0, "negative length");
  if (n == 0) return SPX_OK;
  SPX_REQUIRE(mask != nullptr, "mask is NULL");
  SPX_REQUIRE(nsel == 0 || selected != nullptr, "selected is NULL");
  SPX_ON_DEVICE(ctx);
  SPX_HIP(hipMemsetAsync(mask, 0, (size_t)n, ctx->stream));
  if (nsel > 0) {
    int64_t blocks = (nsel + 255) / 256;
    if (blocks > 8 * (int64_t)ctx->num_cu) blocks = 8 * (int64_t)ctx->num_cu;
    hipLaunchKernelGGL(k_scatter_mask, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, mask, n, selected, nsel);
    SPX_LAUNCH_CHECK();
  }
  return SPX_OK;
}

// ---------------------------------------------------------------------------------------------
// Synthetic inputs: a counter-based generator keyed by (seed, stream, i) that the host can reproduce
// bit for bit WITHOUT a GPU or torch (the checker's synth.py, numpy): splitmix64 of the counter, integer arithmetic and exact
// binary64 additions only -- no libm call whose last ulp differs between host and device.
//   kind 0: U(-1/2, 1/2)   = (top 53 bits of one draw) * 2^-53 - 1/2
//   kind 1: ~N(0, 1)       = sum of 12 such uniforms on [0, 1) minus 6 (Irwin-Hall: mean 0, variance 1, support +-6; every
//                            partial sum is a multiple of 2^-53 below 16, so the additions are exact in any order)
// out[i] = scale * value(i).  Benchmark / test plumbing, not part of the reference's interface.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t spx_splitmix64(uint64_t x) {
  x += 0x9e3779b97f4a7c15ull;
  x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ull;
  x = (x ^ (x >> 27)) * 0x94d049bb133111ebull;
  return x ^ (x >> 31);
}
__global__ __launch_bounds__(256) void k_synth_fill(double* out, int64_t n, uint64_t seed, uint64_t stream, int kind,
                                                     double scale) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const uint64_t key = spx_splitmix64(seed ^ (stream * 0xd1342543de82ef95ull));
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    double v;
    if (kind == 0) {
      v = (double)(spx_splitmix64(key + (uint64_t)i) >> 11) * 0x1.0p-53 - 0.5;
    } else {
      double acc = 0.0;
      for (int k = 0; k < 12; ++k) acc += (double)(spx_splitmix64(key + (uint64_t)i * 12ull + (uint64_t)k) >> 11) * 0x1.0p-53;
      v = acc - 6.0;
    }
    out[i] = scale * v;
  }
}

SPX_EXPORT int spx_synth_fill(spx_ctx* ctx, double* out, int64_t n, uint64_t seed, uint64_t stream, int kind, double scale) {
  SPX_REQUIRE(ctx != nullptr, "ctx is NULL");
  SPX_REQUIRE(n >= 0 && (n == 0 || out != nullptr), "bad output vector");
  SPX_REQUIRE(kind == 0 || kind == 1, "kind must be 0 (uniform) or 1 (normal-like)");
  if (n == 0) return SPX_OK;
  SPX_ON_DEVICE(ctx);
  int64_t blocks = (n + 255) / 256;
  if (blocks > (int64_t)ctx->num_cu * 32) blocks = (int64_t)ctx->num_cu * 32;
  hipLaunchKernelGGL(k_synth_fill, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, out, n, seed, stream, kind, scale);
  SPX_LAUNCH_CHECK();
  return SPX_OK;
}

#ifdef SPX_DEBUG_PEEK  // diagnostic builds only: bytes of the context's workspace (which = 0) or synchronisation state (1)
SPX_EXPORT int spx_debug_peek(spx_ctx* ctx, int which, size_t offset, size_t nbytes, void* out) {
  SPX_REQUIRE(ctx != nullptr && out != nullptr, "NULL argument");
  const char* base = static_cast<const char*>(which ? ctx->sync : ctx->ws);
  const size_t cap = which ? ctx->sync_bytes : ctx->ws_bytes;
  SPX_REQUIRE(base != nullptr && offset + nbytes <= cap, "range outside the buffer");
  SPX_ON_DEVICE_RAW(ctx);
  SPX_HIP(hipStreamSynchronize(ctx->stream));
  SPX_HIP(hipMemcpy(out, base + offset, nbytes, hipMemcpyDeviceToHost));
  return SPX_OK;
}
#endif

// ---------------------------------------------------------------------------------------------
// Strided views (the reference accepts `view(y, 1:2:10)` as xk: test/runtests.jl:196-209).  The kernels take unit-stride
// vectors; a host binding keeps a packed copy of a strided xk, refreshes it with this copy before a call and writes through
// it after shift!.  dst[i * dst_stride] = src[i * src_stride], i < n; strides in ELEMENTS (>= 1), elem_bytes 4 or 8.
// ---------------------------------------------------------------------------------------------
template <class W>
__global__ __launch_bounds__(256) void k_copy_strided(W* __restrict__ dst, int64_t ds, const W* __restrict__ src, int64_t ss,
                                                       int64_t n) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i * ds] = src[i * ss];
}
SPX_EXPORT int spx_copy_strided(spx_ctx* ctx, void* dst, int64_t dst_stride, const void* src, int64_t src_stride, int64_t n,
                                int elem_bytes) {
  SPX_REQUIRE(ctx != nullptr, "ctx is NULL");
  SPX_REQUIRE(n >= 0 && dst_stride >= 1 && src_stride >= 1, "negative length or non-positive stride");
  SPX_REQUIRE(elem_bytes == 4 || elem_bytes == 8, "elem_bytes must be 4 or 8");
  if (n == 0) return SPX_OK;
  SPX_REQUIRE(dst != nullptr && src != nullptr, "NULL vector with n > 0");
  SPX_REQUIRE(((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src)) & (uintptr_t)(elem_bytes - 1)) == 0,
              "vector not aligned to its element size");
  SPX_ON_DEVICE(ctx);
  int64_t blocks = (n + 255) / 256;
  if (blocks > (int64_t)ctx->num_cu * 16) blocks = (int64_t)ctx->num_cu * 16;
  if (elem_bytes == 8)
    hipLaunchKernelGGL(k_copy_strided<unsigned long long>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       static_cast<unsigned long long*>(dst), dst_stride, static_cast<const unsigned long long*>(src), src_stride, n);
  else
    hipLaunchKernelGGL(k_copy_strided<unsigned int>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       static_cast<unsigned int*>(dst), dst_stride, static_cast<const unsigned int*>(src), src_stride, n);
  SPX_LAUNCH_CHECK();
  return SPX_OK;
}
