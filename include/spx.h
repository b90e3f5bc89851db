/*
 * spx.h -- C ABI of libspx: MI355X (gfx950) shifted proximal operators, fp64 (NormL1 / NormL0 families also fp32).
 *
 * This is the drop-in boundary for the prox!() hot path of ShiftedProximalOperators.jl v0.2.2
 * (reference paths below are relative to the reference repository).  In the reference the path sits
 * behind Julia multiple dispatch,
 *     prox!(y::AbstractVector{R}, psi::<ShiftedType>, q::AbstractVector{R}, sigma::R) -> y
 * (src/ShiftedProximalOperators.jl:135-152; one method per operator, cited at each entry point).
 * A Julia maintainer binds these entry points with `ccall` from `prox!` methods specialised on
 * device-array-backed psi (julia/SPXShim.jl); this repository's own host-side mirror
 * of the reference API (Python, ctypes) binds exactly the same symbols.
 *
 * Conventions
 *   - Every data pointer is a DEVICE pointer to contiguous float64 (or as typed) memory on the
 *     context's device (the spx_host_* twins at the end of this file take HOST pointers and stage them).
 *     Nothing is retained past a call; the caller owns all vectors
 *     (the reference borrows xk/sj/l/u by reference too: src/shiftedNormL1Box.jl:22-47).
 *   - `y` may alias `q` exactly (test/test_allocs.jl:108-113) and may be the operator's own `sol`
 *     (prox(), src/ShiftedProximalOperators.jl:189-190).  Partial overlap is undefined.
 *   - Calls are asynchronous on the context's HIP stream and ordered on it; spx_sync() waits.  Exceptions
 *     (they return a value or a verdict to the host and therefore synchronise): spx_check_bounds, spx_obj_* and
 *     spx_proxval_* unless the context has a device value target (spx_ctx_set_value_target), the unboxed
 *     spx_iprox_* with check_d != 0, the gather-index group forms (index validation) and every spx_host_* form.
 *     (Since round 2 the top-r operators and spx_prox_l1_b2 read nothing back.)
 *     Failures of the device side (spx_status SPX_ERR_INTERNAL) are reported by the NEXT call on the context, whichever
 *     entry point it is -- see spx_status.
 *     A context is not re-entrant (neither is a reference psi: shared scratch sol/xsy/p).
 *   - Return value: 0 = SPX_OK, else an spx_status; spx_last_error() gives a thread-local message.
 *   - Indices handed over in arrays (selected sets, group offsets) are 0-BASED int64.
 *   - Floating-point semantics: no FMA contraction, reference operation order; min/max follow
 *     Julia (IEEE-754-2019 minimum/maximum).  L1/L0 families and the IndBallL0 selection are bit-exact w.r.t. the
 *     reference FORMULAS as restated in oracle/spx_oracle.c; Lhalf and group-L2 families agree with that restatement
 *     to <= 1e-12 (operand-scale relative), and where the reference's own Float64 evaluation is further than that from
 *     the exact value of its formula (roots next to the pole of step(n)) they are the closer side (binary128 arbiter,
 *     tests/arbiter.py).  What the reference's own tests pin: the Box operators, RootNormLhalf (unshifted) and
 *     GroupNormL2(Binf) (golden vectors, test/runtests.jl:113-126,449-494,587-606,658-705, test/testsbox.jl).  PARITY
 *     UNPINNED by reference vectors (its tests say `# test prox # TODO`, runtests.jl:179-180,382-383,772-773): unboxed
 *     ShiftedNormL0 / L1 / RootNormLhalf, ShiftedIndBallL0(BInf) incl. the sortperm tie-break, and every Float32 form --
 *     for these "bit-exact" means: against the literal restatement of the source text.
 */
#ifndef SPX_H
#define SPX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPX_ABI_VERSION 1

typedef enum spx_status {
  SPX_OK = 0,
  SPX_ERR_INVALID_ARG = 1, /* NULL pointer, negative length, bad group description ...            */
  SPX_ERR_BOUNDS = 2,      /* reserved for "at least one lower bound is greater than the upper"   */
  SPX_ERR_HIP = 3,         /* a HIP runtime call failed (message has the HIP error string)        */
  SPX_ERR_ALLOC = 4,       /* workspace allocation failed                                         */
  SPX_ERR_NO_DEVICE = 5,   /* no usable gfx950 device                                             */
  SPX_ERR_ASSERT = 6,      /* the reference's `@assert d[i] > 0` failed (unboxed iprox!)           */
  SPX_ERR_INTERNAL = 7     /* the device side of the context has reported a failure: a kernel that synchronises inside one
                              launch gave up waiting for its own workgroups (they were not all resident: another process or
                              a graph replay held the CUs; or a context shared by two threads), or library-owned device
                              state was found outside its layout.  The kernel raises a status word in host-mapped memory;
                              EVERY entry point looks at it before it enqueues anything, so the first libspx call after
                              the failure -- whichever it is, no spx_sync needed -- returns this code, and so does every
                              later one until spx_sync has reported it (spx_sync resets the state: the context is usable
                              again).  The failing kernel stores NaN, never a plausible wrong result.                   */
} spx_status;

typedef struct spx_ctx spx_ctx; /* opaque: device id, HIP stream, library-owned scratch */

/* ---- library / context ------------------------------------------------------------------- */
int spx_abi_version(void);
const char* spx_last_error(void);

/* device: HIP device ordinal.  spx_ctx_create: the context creates and owns a non-blocking stream.
 * spx_ctx_create_on_stream: enqueue on the caller's hipStream_t (passed as void*; NULL = the legacy
 * default stream), e.g. the array library's current stream, so that calls are ordered with the
 * caller's own kernels; the stream is borrowed, never destroyed. */
int spx_ctx_create(int device, spx_ctx** out);
int spx_ctx_create_on_stream(int device, void* stream, spx_ctx** out);
int spx_ctx_destroy(spx_ctx* ctx);
int spx_sync(spx_ctx* ctx);
/* Device-resident values.  By default the value-returning entry points (spx_obj_*, spx_proxval_*) copy their double to
 * the host and synchronise the stream.  With a non-NULL device_value they store it in that DEVICE double instead (one
 * 8-byte store by the last kernel of the call) and return after enqueueing: the host `*value` is then set to NaN and
 * nothing is read back -- a solver's accept / reject test can consume the value on the device, or copy it later.
 * NULL restores the default.  (An out-of-range gather index cannot be reported from an asynchronous call: the device
 * value is NaN then.)  Mirrors no reference function: psi(y) in the reference returns a host Float64
 * (src/ShiftedProximalOperators.jl:51-54); this is the asynchronous form of the same value. */
int spx_ctx_set_value_target(spx_ctx* ctx, double* device_value);
/* Stream capture (hipGraph): calls on a context whose stream is being captured are recorded, not run.  Capturable: every
 * prox / iprox (check = 0) / objective / prox-value entry point on device pointers, provided values go to a device double
 * (spx_ctx_set_value_target) and the same call has run once before on this context (workspaces do not grow while
 * capturing).  Not capturable (SPX_ERR_INVALID_ARG, nothing launched): host-valued results, spx_check_bounds, index-set
 * (gather) group layouts, host-pointer forms, ShiftedNormL1B2 on vectors of mixed alignment.  The first capture puts the
 * context into a graph-safe mode for good: the kernels that synchronise inside one launch are then preceded by a zero-fill of
 * the state they use (in the graph and in eager calls alike), ~2-4 us per such call, and a workspace that a LATER, larger
 * eager call outgrows is retired, not freed, until spx_ctx_destroy -- a captured graph keeps pointing at the block it was
 * captured with, so replaying it after such a call is safe (at most 64 such growths per context, then SPX_ERR_INVALID_ARG).
 * Do not replay such a graph while ANOTHER context runs top-r or ShiftedNormL1B2 calls on the same device: eager calls of
 * different contexts are chained through an event so that two resident grids never wait for CUs the other holds; a replay
 * is outside that chain.  If it happens anyway the waiting kernels give up after a bounded number of polls, store NaN and
 * the next call on the context returns SPX_ERR_INTERNAL (see spx_status) -- an error, never a silent wrong result. */

/* Strided views (the reference accepts `view(y, 1:2:10)` as xk: test/runtests.jl:196-209).  Every entry point takes
 * unit-stride vectors; a host binding keeps a packed copy of a strided xk, refreshes it with this copy before a call and
 * writes through it after shift! (8 or 16 B/element of extra traffic per call -- correctness, not speed).
 * dst[i * dst_stride] = src[i * src_stride] for i < n; strides in ELEMENTS, >= 1; elem_bytes 4 or 8; device pointers. */
int spx_copy_strided(spx_ctx* ctx, void* dst, int64_t dst_stride, const void* src, int64_t src_stride, int64_t n,
                     int elem_bytes);

/* Synthetic benchmark / test inputs: out[i] = scale * value(seed, stream, i) from a counter-based generator
 * (splitmix64, integer arithmetic and exact binary64 additions only) that the checker's synth.py reproduces on the host bit for
 * bit, so a CPU check needs no copy of the device data and no torch.  kind 0: U(-1/2, 1/2); kind 1: ~N(0, 1) as the sum of
 * 12 uniforms minus 6 (Irwin-Hall).  Not part of the reference's interface. */
int spx_synth_fill(spx_ctx* ctx, double* out, int64_t n, uint64_t seed, uint64_t stream, int kind, double scale);
/* HIP-event stopwatch on the context's stream (used by bench.py for per-launch durations). */
int spx_timer_start(spx_ctx* ctx);
int spx_timer_stop(spx_ctx* ctx, float* elapsed_ms); /* records stop, waits, returns milliseconds */

/* Kernel-benchmark / A-B knobs of ONE context (not part of the reference API; they never change results, only which of
 * several equivalent kernels runs): key 0 = workgroups per CU of the separable grid (0 = one tile per workgroup, the
 * default), key 1 = non-temporal loads/stores (0/1), key 2 = sample-predicted top-r path (1, default) or always the
 * full-vector radix select (0), key 3 = LDS-staged (1, default) or register-staged (0) separable skeleton, key 4 =
 * single-pass form of the top-r path when y overlaps no input (1, default) or always the two-pass form (0), key 5 =
 * XCD-contiguous tile ranges in the LDS-staged skeleton (0, default: tile = workgroup id), key 6 = one-workgroup top-r
 * kernel for n <= 8192 (1, default), key 8 = pretend that at most this many workgroups of a kernel that synchronises inside
 * one launch can be resident at once (0, default: what hipOccupancyMaxActiveBlocksPerMultiprocessor says; the grids of
 * top-r and ShiftedNormL1B2 are sized by it and fall back to their any-grid forms -- this key lets a test force that).
 * Key 10 = samples per lane of the top-r front kernel (0, default: 1 / 2 / 4 by n, 16 for a cut in the bulk of a vector of
 * >= 2^26 elements; 1, 2, 4, 16 force it).  Key 11 = one-launch top-r with v parked in LDS for 2^20 < n <= 16 Ki x resident
 * workgroups, and beyond that -- up to 24 Ki x resident workgroups = 6 Mi elements -- with 8 more elements per lane in
 * registers (1, default; 2: v in LDS up to 2^22, the sample-predicted path above; 0: registers up to 2^21, the sample-predicted
 * path above, as in round 2).  Key 12 = ShiftedNormL1B2
 * with xk parked in LDS for 2^21 < n <= 2^22 (1, default; 0: the two-pass streaming form from 2^21 on).  (Key 7, round 2's switch
 * to the multi-launch pipelines, is gone with those pipelines.)  Key 13 = large contiguous groups of ShiftedGroupNormL2(Binf)
 * -- first of all ONE group over the whole vector, the reference's `shifted(NormL2(lambda), xk)` -- are owned by a team of
 * workgroups and psi(y) on them is evaluated in chunks (1, default; 0: one workgroup / one wavefront per group as in rounds 1-3).
 * Key 14 = the Binf form of that takes its sample-predicted two-pass path when the group does not fit on chip (1, default;
 * 0: the generic body, one streaming pass per reduction of the root find).
 * Key 17 = launches per call at solver sizes (1, default): psi(y) (spx_obj_*) is ONE launch -- the workgroup that finishes
 * last adds the partial sums, in the order the separate final launch did -- and ShiftedGroupNormL2Binf on uniform groups
 * runs without the zero-fill launch of its deferred list (two count words that alternate between calls; under a stream
 * capture the zero-fill node stays); 0 = three launches each, as in rounds 1-3.  Same bits either way.
 * Key 9 DOES change results, within the stated tolerance: ShiftedGroupNormL2Binf, 0 (default) = the closed form at the root
 * (within ~1e-15 of the exact value of the reference's formula everywhere), 1 = groups whose root sits next to the pole of
 * step(n) (u < n / 1000) are evaluated literally, operation by operation as src/shiftedGroupNormL2Binf.jl:87-113 with
 * Roots-style bisection -- the reference's own Float64 value there, which is up to 4.5e-9 of the scale off its formula.
 * For callers who must reproduce a reference run.  Contexts are independent; a context is used by one thread at a time. */
int spx_ctx_set_tuning(spx_ctx* ctx, int key, int value);

/* ---- construction-time helpers (the reference's constructors) ---------------------------- */
/* any(l .> u) of the Box constructors (src/shiftedNormL1Box.jl:33-35, shiftedNormL0Box.jl:33-35).
 * Vector bound if the pointer is non-NULL, else the scalar.  Synchronous; *any_l_gt_u = 0/1. */
int spx_check_bounds(spx_ctx* ctx, const double* l_vec, const double* u_vec, double l_scalar,
                     double u_scalar, int64_t n, int* any_l_gt_u);
/* Byte mask of the `selected` index set (src/shiftedNormL1Box.jl:106 `i in psi.selected`): mask[i] = 1
 * iff i occurs in selected[0..nsel) (0-based, any order, duplicates allowed, out-of-range ignored). */
int spx_build_mask(spx_ctx* ctx, uint8_t* mask, int64_t n, const int64_t* selected, int64_t nsel);

/* ---- separable operators: y[i] depends on q[i], xk[i], sj[i] only ------------------------- */
/* ShiftedNormL1.prox!        src/shiftedNormL1.jl:40-54 */
int spx_prox_l1(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                int64_t n, double lambda, double sigma);
/* ShiftedNormL0.prox!        src/shiftedNormL0.jl:38-55 */
int spx_prox_l0(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                int64_t n, double lambda, double sigma);
/* ShiftedRootNormLhalf.prox! src/shiftedRootNormLhalf.jl:41-63 */
int spx_prox_lhalf(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                   int64_t n, double lambda, double sigma);

/* Box forms.  Bounds: l_vec/u_vec (length n) if non-NULL, else l_scalar/u_scalar
 * (`isa(psi.l, Real) ? psi.l : psi.l[i]`).  sel_mask: byte per index (1 = selected), NULL = all
 * selected; unselected entries get prox_zero = clamp(q, l - s, u - s)
 * (src/ShiftedProximalOperators.jl:203). */
/* ShiftedNormL1Box.prox!        src/shiftedNormL1Box.jl:89-125  (BASELINE headline operator) */
int spx_prox_l1_box(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                    int64_t n, double lambda, double sigma, const double* l_vec, const double* u_vec,
                    double l_scalar, double u_scalar, const uint8_t* sel_mask);
/* ShiftedNormL0Box.prox!        src/shiftedNormL0Box.jl:89-131 */
int spx_prox_l0_box(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                    int64_t n, double lambda, double sigma, const double* l_vec, const double* u_vec,
                    double l_scalar, double u_scalar, const uint8_t* sel_mask);
/* ShiftedRootNormLhalfBox.prox! src/shiftedRootNormLhalfBox.jl:86-120 */
int spx_prox_lhalf_box(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                       int64_t n, double lambda, double sigma, const double* l_vec,
                       const double* u_vec, double l_scalar, double u_scalar,
                       const uint8_t* sel_mask);

/* ---- Float32 forms (round 2) -------------------------------------------------------------------
 * The reference's structs and prox! methods are generic in R <: Real (`prox!(y::AbstractVector{R}, psi::ShiftedNormL1Box{R,...},
 * q::AbstractVector{R}, sigma::R)`, src/shiftedNormL1Box.jl:89-94; Float32 operators on views: test/runtests.jl:196-209).
 * With R = Float32 every operation of the NormL1 / NormL0 bodies is a Float32 operation, so these entry points reproduce the
 * reference BIT FOR BIT in fp32 (lambda and sigma are Float32 too: lambda * sigma, sqrt(2 lambda sigma), 2 lambda sigma are
 * formed in Float32).  Device pointers; any 4-byte alignment (views that start at any element); y may alias q; 16 B/element.
 * RootNormLhalf has no Float32 form: the reference's body mixes Float64 literals into it and computes in Float64. */
/* ShiftedNormL1.prox!     src/shiftedNormL1.jl:40-54 */
int spx_prox_l1_f32(spx_ctx* ctx, float* y, const float* q, const float* xk, const float* sj, int64_t n, float lambda,
                    float sigma);
/* ShiftedNormL0.prox!     src/shiftedNormL0.jl:38-55 */
int spx_prox_l0_f32(spx_ctx* ctx, float* y, const float* q, const float* xk, const float* sj, int64_t n, float lambda,
                    float sigma);
/* ShiftedNormL1Box.prox!  src/shiftedNormL1Box.jl:89-125 */
int spx_prox_l1_box_f32(spx_ctx* ctx, float* y, const float* q, const float* xk, const float* sj, int64_t n, float lambda,
                        float sigma, const float* l_vec, const float* u_vec, float l_scalar, float u_scalar,
                        const uint8_t* sel_mask);
/* ShiftedNormL0Box.prox!  src/shiftedNormL0Box.jl:89-131 */
int spx_prox_l0_box_f32(spx_ctx* ctx, float* y, const float* q, const float* xk, const float* sj, int64_t n, float lambda,
                        float sigma, const float* l_vec, const float* u_vec, float l_scalar, float u_scalar,
                        const uint8_t* sel_mask);

/* iprox! on Float32 vectors (round 3): the reference's methods are generic in R here too (src/shiftedNormL1.jl:60-75,
 * shiftedNormL0.jl:61-80, shiftedNormL1Box.jl:131-225, shiftedNormL0Box.jl:137-231; thresholds eps(R) = eps(Float32), unselected
 * entries iprox_zero, src/ShiftedProximalOperators.jl:217-236).  Bit for bit in fp32; arguments as the Float64 forms; 20 B/element. */
int spx_iprox_l1_f32(spx_ctx* ctx, float* y, const float* g, const float* d, const float* xk, const float* sj, int64_t n,
                     float lambda, int check_d);
int spx_iprox_l0_f32(spx_ctx* ctx, float* y, const float* g, const float* d, const float* xk, const float* sj, int64_t n,
                     float lambda, int check_d);
int spx_iprox_l1_box_f32(spx_ctx* ctx, float* y, const float* g, const float* d, const float* xk, const float* sj, int64_t n,
                         float lambda, const float* l_vec, const float* u_vec, float l_scalar, float u_scalar,
                         const uint8_t* sel_mask);
int spx_iprox_l0_box_f32(spx_ctx* ctx, float* y, const float* g, const float* d, const float* xk, const float* sj, int64_t n,
                         float lambda, const float* l_vec, const float* u_vec, float l_scalar, float u_scalar,
                         const uint8_t* sel_mask);

/* psi(y) on Float32 vectors, for every operator (the reference's tests evaluate each shifted operator built on Float32
 * data: test/runtests.jl:196-209, 268-282, 346-360, 397-412, 524-550, 630-646).  Every element operation is a Float32
 * operation as in the reference ((xk + sj) + y, sj + y, the box ends -+ sqrt(eps(Float32)), each square root); `1.1 * Delta`
 * and the comparison against it are Float64 (Julia promotes the literal); the SUM is formed in Float64 and *value is a
 * double -- round it to Float32 to compare with the reference's Float32 result (counts and +Inf decisions are exact, sums
 * agree to Float32 rounding of the reference's own pairwise Float32 summation).  Device value targets and the host
 * read-back behave as for the Float64 forms. */
int spx_obj_l1_f32(spx_ctx* ctx, const float* y, const float* xk, const float* sj, int64_t n, float lambda, double* value);
int spx_obj_l0_f32(spx_ctx* ctx, const float* y, const float* xk, const float* sj, int64_t n, float lambda, double* value);
int spx_obj_lhalf_f32(spx_ctx* ctx, const float* y, const float* xk, const float* sj, int64_t n, float lambda, double* value);
int spx_obj_l1_box_f32(spx_ctx* ctx, const float* y, const float* xk, const float* sj, int64_t n, float lambda,
                       const float* l_vec, const float* u_vec, float l_scalar, float u_scalar, const uint8_t* sel_mask,
                       double* value);
int spx_obj_l0_box_f32(spx_ctx* ctx, const float* y, const float* xk, const float* sj, int64_t n, float lambda,
                       const float* l_vec, const float* u_vec, float l_scalar, float u_scalar, const uint8_t* sel_mask,
                       double* value);
int spx_obj_lhalf_box_f32(spx_ctx* ctx, const float* y, const float* xk, const float* sj, int64_t n, float lambda,
                          const float* l_vec, const float* u_vec, float l_scalar, float u_scalar, const uint8_t* sel_mask,
                          double* value);
int spx_obj_indball_l0_f32(spx_ctx* ctx, const float* y, const float* xk, const float* sj, int64_t n, int64_t r, double* value);
int spx_obj_indball_l0_binf_f32(spx_ctx* ctx, const float* y, const float* xk, const float* sj, int64_t n, int64_t r,
                                float delta, double* value);
int spx_obj_group_l2_f32(spx_ctx* ctx, const float* y, const float* xk, const float* sj, int64_t n,
                         const int64_t* group_offsets, int64_t group_size, int64_t ngroups, const float* lambda_vec,
                         double* value);
int spx_obj_group_l2_binf_f32(spx_ctx* ctx, const float* y, const float* xk, const float* sj, int64_t n,
                              const int64_t* group_offsets, int64_t group_size, int64_t ngroups, const float* lambda_vec,
                              float delta, double* value);

/* ---- prox! fused with the value of h at the result ------------------------------------------ */
/* One pass instead of two for the pair every solver iteration makes (R2: `prox!(s, psi, ...)` then `h(xk + s)`):
 * y as spx_prox_X, and *value = h over the selected indices of (xk + sj) + y -- lambda * sum |v|, lambda * #nonzeros,
 * lambda * sum sqrt|v| (src/ShiftedProximalOperators.jl:51-54, Box forms src/shiftedNormL1Box.jl:70-82 without the
 * feasibility scan: a prox result lies inside the box by construction).  The prox is taken at q_scale * q[i], formed on
 * the fly (R2's `mnu_grad .= -nu .* grad` without the extra pass; pass 1.0 for q itself; bit-identical to scaling q
 * beforehand).  Synchronous: *value is written on the host. */
int spx_proxval_l1(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                   double lambda, double sigma, double q_scale, double* value);
int spx_proxval_l0(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                   double lambda, double sigma, double q_scale, double* value);
int spx_proxval_lhalf(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                      double lambda, double sigma, double q_scale, double* value);
int spx_proxval_l1_box(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                       double lambda, double sigma, const double* l_vec, const double* u_vec, double l_scalar,
                       double u_scalar, const uint8_t* sel_mask, double q_scale, double* value);
int spx_proxval_l0_box(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                       double lambda, double sigma, const double* l_vec, const double* u_vec, double l_scalar,
                       double u_scalar, const uint8_t* sel_mask, double q_scale, double* value);
int spx_proxval_lhalf_box(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                          double lambda, double sigma, const double* l_vec, const double* u_vec, double l_scalar,
                          double u_scalar, const uint8_t* sel_mask, double q_scale, double* value);

/* ---- iprox!: argmin 1/2 y'Dy + g'y + psi(y), D = diag(d)  (src/ShiftedProximalOperators.jl:154-180) ------ */
/* Separable; reads g, d, xk, sj (40 B/element).  y may alias g.
 * Unboxed forms: the reference asserts d[i] > 0.  check_d != 0: the call synchronises and returns SPX_ERR_ASSERT
 * if some d[i] <= 0 (y is then unspecified, as after the reference's exception); check_d == 0: asynchronous, no check. */
/* ShiftedNormL1.iprox!     src/shiftedNormL1.jl:60-75 */
int spx_iprox_l1(spx_ctx* ctx, double* y, const double* g, const double* d, const double* xk,
                 const double* sj, int64_t n, double lambda, int check_d);
/* ShiftedNormL0.iprox!     src/shiftedNormL0.jl:61-80 */
int spx_iprox_l0(spx_ctx* ctx, double* y, const double* g, const double* d, const double* xk,
                 const double* sj, int64_t n, double lambda, int check_d);
/* ShiftedNormL1Box.iprox!  src/shiftedNormL1Box.jl:131-225; unselected entries: iprox_zero
 * (src/ShiftedProximalOperators.jl:217-236) */
int spx_iprox_l1_box(spx_ctx* ctx, double* y, const double* g, const double* d, const double* xk,
                     const double* sj, int64_t n, double lambda, const double* l_vec,
                     const double* u_vec, double l_scalar, double u_scalar, const uint8_t* sel_mask);
/* ShiftedNormL0Box.iprox!  src/shiftedNormL0Box.jl:137-231 */
int spx_iprox_l0_box(spx_ctx* ctx, double* y, const double* g, const double* d, const double* xk,
                     const double* sj, int64_t n, double lambda, const double* l_vec,
                     const double* u_vec, double l_scalar, double u_scalar, const uint8_t* sel_mask);

/* ---- psi(y): objective value h(xk + sj + y) [+ indicator of the box / trust region] ------------------------ */
/* Generic form src/ShiftedProximalOperators.jl:51-54; *value is written on the HOST; synchronous.
 * NormL1: lambda * sum |v|, NormL0: lambda * #nonzeros, RootNormLhalf: lambda * sum sqrt|v| (src/rootNormLhalf.jl:27-29). */
int spx_obj_l1(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n, double lambda, double* value);
int spx_obj_l0(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n, double lambda, double* value);
int spx_obj_lhalf(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n, double lambda, double* value);
/* Box forms (src/shiftedNormL1Box.jl:70-82, shiftedNormL0Box.jl:70-82, shiftedRootNormLhalfBox.jl:67-79): h over the
 * selected indices; +Inf unless l - sqrt(eps) <= sj[i] + y[i] <= u + sqrt(eps) for EVERY i. */
int spx_obj_l1_box(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n, double lambda,
                   const double* l_vec, const double* u_vec, double l_scalar, double u_scalar,
                   const uint8_t* sel_mask, double* value);
int spx_obj_l0_box(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n, double lambda,
                   const double* l_vec, const double* u_vec, double l_scalar, double u_scalar,
                   const uint8_t* sel_mask, double* value);
int spx_obj_lhalf_box(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n, double lambda,
                      const double* l_vec, const double* u_vec, double l_scalar, double u_scalar,
                      const uint8_t* sel_mask, double* value);
/* IndBallL0 (0 or +Inf) and its BInf form, which adds IndBallLinf(1.1 Delta)(sj + y)
 * (src/shiftedIndBallL0BInf.jl:44-49). */
int spx_obj_indball_l0(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n, int64_t r,
                       double* value);
int spx_obj_indball_l0_binf(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,
                            int64_t r, double delta, double* value);
/* GroupNormL2: sum_g lambda_g ||v[idx_g]||_2 (src/groupNormL2.jl:33-39); Binf form src/shiftedGroupNormL2Binf.jl:34-39.
 * Group description as for spx_prox_group_l2. */
int spx_obj_group_l2(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,
                     const int64_t* group_offsets, int64_t group_size, int64_t ngroups,
                     const double* lambda_vec, double* value);
int spx_obj_group_l2_binf(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,
                          const int64_t* group_offsets, int64_t group_size, int64_t ngroups,
                          const double* lambda_vec, double delta, double* value);

/* ---- top-r selection ------------------------------------------------------------------- */
/* ShiftedIndBallL0.prox!     src/shiftedIndBallL0.jl:54-72 : keep the r entries of (xk+sj)+q largest in
 * magnitude (ties: lowest index first, = stable sortperm), zero the rest, subtract xk+sj.
 * The result never depends on the route taken, the time does: up to 2^22 elements (what 256 resident workgroups hold in
 * registers or LDS) an exact radix select in ONE launch reads the vectors once and writes y once (n = 4e6: 42 us, n = 1e5: 19 us);
 * beyond, a sample predicts a band around the r-th largest magnitude and one streaming pass settles everything outside it
 * (n = 1e8: 0.57-0.62 ms, sorted input included).
 * Keys at the threshold that are shared by per cents of the vector (lattice data, constants, a sparse vector's zeros) are
 * counted per wavefront instead of recorded, and the index tie-break (lowest index first) comes from a prefix sum over those
 * counts: 0.59-0.85 ms at n = 1e8 (round 2: 2.6-6.5 ms).  If the sample misleads, the exact radix select queued behind the
 * pass recomputes everything.  Nothing is read back by any route. */
int spx_prox_indball_l0(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                        int64_t n, int64_t r);
/* ShiftedIndBallL0BInf.prox! src/shiftedIndBallL0BInf.jl:73-95 : as above, then clamp y to [-delta, delta]. */
int spx_prox_indball_l0_binf(spx_ctx* ctx, double* y, const double* q, const double* xk,
                             const double* sj, int64_t n, int64_t r, double delta);

/* The same operators on Float32 vectors (round 3; the reference's methods are generic in R, src/shiftedIndBallL0.jl:54-59):
 * v = (xk + sj) + q, the magnitude order and the final subtraction / clamp are Float32 operations -- bit-exact in fp32, ties by
 * lowest index, NaN largest.  Exact select in one launch (one workgroup / register-resident / v parked in LDS up to 2^23
 * elements: 16 B per element moved / v parked in y beyond); the sample-predicted single pass is Float64 only. */
int spx_prox_indball_l0_f32(spx_ctx* ctx, float* y, const float* q, const float* xk, const float* sj, int64_t n, int64_t r);
int spx_prox_indball_l0_binf_f32(spx_ctx* ctx, float* y, const float* q, const float* xk, const float* sj, int64_t n,
                                 int64_t r, float delta);

/* ---- l1 norm + l2-ball trust region ------------------------------------------------------ */
/* ShiftedNormL1B2.prox!  src/shiftedNormL1B2.jl:50-67 (chi = NormL2(chi_lambda)).  All elements are coupled through
 * one scalar root (find_zero, :62).  One launch, asynchronous, nothing read back: the vectors stay on chip up to 2^22
 * elements (registers to 2^21, xk parked in LDS beyond: n = 4e6 61 us); beyond, two streaming passes (56 B/element) -- the
 * first classifies every element against a bracket around a sample's root, the root is found on the aggregate sums + the few
 * per cent of candidates, the second stores y (n = 1e8: 0.91-0.96 ms; 0.51 ms when the trust region is inactive).  Results
 * are reproducible from run to run (every sum is formed in a fixed order). */
int spx_prox_l1_b2(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                   int64_t n, double lambda, double sigma, double delta, double chi_lambda);
/* psi(y) of ShiftedNormL1B2, src/shiftedNormL1B2.jl:32: lambda ||xk + sj + y||_1 + IndBallL2(Delta)(sj + y); the value is
 * written on the host (synchronous). */
int spx_obj_l1_b2(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n, double lambda,
                  double delta, double* value);

/* ---- group operators -------------------------------------------------------------------- */
/* Groups are contiguous index ranges (the reference's `idx` entries as UnitRanges / [:]):
 *   group_offsets != NULL : CSR offsets (device, int64, length ngroups+1, 0-based, non-decreasing,
 *                           offsets[0] >= 0, offsets[ngroups] <= n); group g = [off[g], off[g+1]).
 *                           group_size is then a HINT: 0 = unknown, > 0 = an upper bound on the group sizes; a bound
 *                           <= 512 selects the register-tile kernels (a group that exceeds it is still computed
 *                           correctly, by the general kernel).
 *   group_offsets == NULL : uniform groups of group_size, ngroups * group_size == n.
 * lambda_vec: device, length ngroups (GroupNormL2.lambda, src/groupNormL2.jl:15-28).
 * ONE group over the whole vector (group_offsets == NULL, group_size == n, ngroups == 1) is what the reference's
 * `shifted(NormL2(lambda), xk[, Delta, chi])` builds (src/shiftedGroupNormL2.jl:34-35, src/shiftedGroupNormL2Binf.jl:48-49: idx = [:],
 * the default of src/groupNormL2.jl:30-31).  Groups too large for one workgroup -- that one, or a handful of ranges of
 * 1e5-1e8 elements, uniform or among the CSR ranges -- are owned by a TEAM of workgroups of one resident grid (round 4,
 * csrc/spx_group_team.hip): read once when a group fits the LDS of its team (2.36 Mi elements on 256 CUs), two streaming passes
 * beyond (56 B/element: n = 1e8 0.90 ms / 0.99 ms with the trust region); rounds 1-3 gave such a group one workgroup (384 /
 * 1349 ms).  These launches synchronise inside themselves (as top-r and ShiftedNormL1B2: bounded waits, NaN + SPX_ERR_INTERNAL
 * on the next call if a wait cannot be satisfied). */
/* ShiftedGroupNormL2.prox!     src/shiftedGroupNormL2.jl:52-79 */
int spx_prox_group_l2(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                      int64_t n, const int64_t* group_offsets, int64_t group_size, int64_t ngroups,
                      const double* lambda_vec, double sigma);
/* ShiftedGroupNormL2Binf.prox! src/shiftedGroupNormL2Binf.jl:67-119 */
int spx_prox_group_l2_binf(spx_ctx* ctx, double* y, const double* q, const double* xk,
                           const double* sj, int64_t n, const int64_t* group_offsets,
                           int64_t group_size, int64_t ngroups, const double* lambda_vec,
                           double sigma, double delta);

/* ShiftedGroupNormL2.prox! on Float32 vectors (round 3; the method is generic in R, src/shiftedGroupNormL2.jl:52-79): every
 * elementwise operation in Float32; the group norm is accumulated in Float64 and rounded once (the reference's `norm` is
 * BLAS / a generic loop: agreement to a few Float32 ulps of the operands, not bits).  Contiguous groups (uniform or CSR).
 * ShiftedGroupNormL2Binf has no Float32 form. */
int spx_prox_group_l2_f32(spx_ctx* ctx, float* y, const float* q, const float* xk, const float* sj, int64_t n,
                          const int64_t* group_offsets, int64_t group_size, int64_t ngroups, const float* lambda_vec,
                          float sigma);

/* Groups as ARBITRARY index sets -- the reference's `idx::Vector{Vector{Int}}` (src/groupNormL2.jl:30-31,
 * test/runtests.jl:290): group g = group_index[group_ptr[g] .. group_ptr[g+1]) (device int64, 0-based; group_ptr has
 * ngroups+1 entries, group_index nnz).  Literal reference semantics: an index listed by several groups keeps the value
 * of the LAST of them; an index inside no group keeps y on entry, minus xk+sj for ShiftedGroupNormL2
 * (src/shiftedGroupNormL2.jl:77) and unchanged for the Binf form (src/shiftedGroupNormL2Binf.jl:116).  An index outside
 * [0, n) returns SPX_ERR_INVALID_ARG before y is written (the reference: BoundsError).  Uses n doubles + n ints of
 * library scratch (the reference's psi.sol) and synchronises once. */
int spx_prox_group_l2_gather(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                             int64_t n, const int64_t* group_ptr, const int64_t* group_index, int64_t ngroups,
                             int64_t nnz, const double* lambda_vec, double sigma);
int spx_prox_group_l2_binf_gather(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                                  int64_t n, const int64_t* group_ptr, const int64_t* group_index, int64_t ngroups,
                                  int64_t nnz, const double* lambda_vec, double sigma, double delta);
int spx_obj_group_l2_gather(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,
                            const int64_t* group_ptr, const int64_t* group_index, int64_t ngroups, int64_t nnz,
                            const double* lambda_vec, double* value);
int spx_obj_group_l2_binf_gather(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,
                                 const int64_t* group_ptr, const int64_t* group_index, int64_t ngroups, int64_t nnz,
                                 const double* lambda_vec, double delta, double* value);

/* ---- host-pointer forms ------------------------------------------------------------------ */
/* The reference's callers (and its whole test suite, test/runtests.jl) hold plain Vector{Float64} in HOST memory.
 * spx_host_X takes exactly the arguments of spx_X with EVERY vector (y, q/g/d, xk, sj, l_vec, u_vec, sel_mask,
 * group_offsets, lambda_vec) in host memory: inputs are copied to a context-owned device staging area, the same HIP
 * kernels run on the context's stream, y is copied back and the call synchronises.  No arithmetic happens on the CPU,
 * and like the rest of the library these fail without a GPU.  y may alias q (g).  Cost: PCIe transfers of every
 * vector per call -- convenience for small problems (BASELINE config 1, n = 1e4) and for running the reference's
 * tests unchanged; solvers should keep psi on the device. */
int spx_host_prox_l1(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
    double lambda, double sigma);
int spx_host_prox_l0(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
    double lambda, double sigma);
int spx_host_prox_lhalf(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t
    n, double lambda, double sigma);
int spx_host_prox_l1_box(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t
    n, double lambda, double sigma, const double* l_vec, const double* u_vec, double l_scalar, double
    u_scalar, const uint8_t* sel_mask);
int spx_host_prox_l0_box(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t
    n, double lambda, double sigma, const double* l_vec, const double* u_vec, double l_scalar, double
    u_scalar, const uint8_t* sel_mask);
int spx_host_prox_lhalf_box(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
    int64_t n, double lambda, double sigma, const double* l_vec, const double* u_vec, double l_scalar, double
    u_scalar, const uint8_t* sel_mask);
int spx_host_iprox_l1(spx_ctx* ctx, double* y, const double* g, const double* d, const double* xk, const
    double* sj, int64_t n, double lambda, int check_d);
int spx_host_iprox_l0(spx_ctx* ctx, double* y, const double* g, const double* d, const double* xk, const
    double* sj, int64_t n, double lambda, int check_d);
int spx_host_iprox_l1_box(spx_ctx* ctx, double* y, const double* g, const double* d, const double* xk, const
    double* sj, int64_t n, double lambda, const double* l_vec, const double* u_vec, double l_scalar, double
    u_scalar, const uint8_t* sel_mask);
int spx_host_iprox_l0_box(spx_ctx* ctx, double* y, const double* g, const double* d, const double* xk, const
    double* sj, int64_t n, double lambda, const double* l_vec, const double* u_vec, double l_scalar, double
    u_scalar, const uint8_t* sel_mask);
int spx_host_obj_l1(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n, double
    lambda, double* value);
int spx_host_obj_l0(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n, double
    lambda, double* value);
int spx_host_obj_lhalf(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n, double
    lambda, double* value);
int spx_host_obj_l1_box(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n, double
    lambda, const double* l_vec, const double* u_vec, double l_scalar, double u_scalar, const uint8_t*
    sel_mask, double* value);
int spx_host_obj_l0_box(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n, double
    lambda, const double* l_vec, const double* u_vec, double l_scalar, double u_scalar, const uint8_t*
    sel_mask, double* value);
int spx_host_obj_lhalf_box(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,
    double lambda, const double* l_vec, const double* u_vec, double l_scalar, double u_scalar, const uint8_t*
    sel_mask, double* value);
int spx_host_obj_indball_l0(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,
    int64_t r, double* value);
int spx_host_obj_indball_l0_binf(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,
    int64_t r, double delta, double* value);
int spx_host_obj_group_l2(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n, const
    int64_t* group_offsets, int64_t group_size, int64_t ngroups, const double* lambda_vec, double* value);
int spx_host_obj_group_l2_binf(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,
    const int64_t* group_offsets, int64_t group_size, int64_t ngroups, const double* lambda_vec, double delta,
    double* value);
int spx_host_prox_indball_l0(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
    int64_t n, int64_t r);
int spx_host_prox_indball_l0_binf(spx_ctx* ctx, double* y, const double* q, const double* xk, const double*
    sj, int64_t n, int64_t r, double delta);
int spx_host_prox_l1_b2(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t
    n, double lambda, double sigma, double delta, double chi_lambda);
int spx_host_prox_group_l2(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
    int64_t n, const int64_t* group_offsets, int64_t group_size, int64_t ngroups, const double* lambda_vec,
    double sigma);
int spx_host_prox_group_l2_binf(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
    int64_t n, const int64_t* group_offsets, int64_t group_size, int64_t ngroups, const double* lambda_vec,
    double sigma, double delta);
int spx_host_prox_group_l2_gather(spx_ctx* ctx, double* y, const double* q, const double* xk, const double*
    sj, int64_t n, const int64_t* group_ptr, const int64_t* group_index, int64_t ngroups, int64_t nnz, const
    double* lambda_vec, double sigma);
int spx_host_prox_group_l2_binf_gather(spx_ctx* ctx, double* y, const double* q, const double* xk, const
    double* sj, int64_t n, const int64_t* group_ptr, const int64_t* group_index, int64_t ngroups, int64_t nnz,
    const double* lambda_vec, double sigma, double delta);
int spx_host_obj_group_l2_gather(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,
    const int64_t* group_ptr, const int64_t* group_index, int64_t ngroups, int64_t nnz, const double*
    lambda_vec, double* value);
int spx_host_obj_group_l2_binf_gather(spx_ctx* ctx, const double* y, const double* xk, const double* sj,
    int64_t n, const int64_t* group_ptr, const int64_t* group_index, int64_t ngroups, int64_t nnz, const
    double* lambda_vec, double delta, double* value);
int spx_host_obj_l1_b2(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n, double lambda,
    double delta, double* value);

#ifdef __cplusplus
}
#endif
#endif /* SPX_H */
