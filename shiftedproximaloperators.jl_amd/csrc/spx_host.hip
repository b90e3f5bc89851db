// spx_host.hip -- host-pointer forms of the C ABI (the reference's callers and its whole test suite
// hold plain Vector{Float64} in host memory).  Every spx_host_* entry point takes the arguments of its device
// twin with ALL vectors in host memory: it copies the inputs into a context-owned device staging area, runs the very
// same HIP kernels on the context's stream, copies y back and synchronises.  There is no CPU arithmetic here.
#include "spx_common.hpp"

namespace {

constexpr size_t kAlign = 256;
inline size_t aligned(size_t b) { return (b + kAlign - 1) / kAlign * kAlign; }

struct In {
  const void* host;
  size_t bytes;
};

int stage_reserve(spx_ctx* ctx, size_t bytes) {
  if (bytes <= ctx->stage_bytes) return SPX_OK;
  SPX_ON_DEVICE(ctx);
  SPX_HIP(hipStreamSynchronize(ctx->stream));
  if (ctx->stage) SPX_HIP(hipFree(ctx->stage));
  ctx->stage = nullptr;
  ctx->stage_bytes = 0;
  hipError_t e = hipMalloc(&ctx->stage, bytes);
  if (e != hipSuccess) {
    spx_set_error("staging hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return SPX_ERR_ALLOC;
  }
  ctx->stage_bytes = bytes;
  return SPX_OK;
}

// Copies `nin` host arrays (NULL / empty ones stay NULL) to the staging area and reserves `out_bytes` more for the
// result.  dev[i] receives the device address of input i, *dev_out the address of the output block.
template <int NIN>
int stage_in(spx_ctx* ctx, const In (&in)[NIN], const void* (&dev)[NIN], size_t out_bytes, void** dev_out) {
  SPX_REQUIRE(ctx != nullptr, "ctx is NULL");
  size_t total = aligned(out_bytes);
  for (int i = 0; i < NIN; ++i)
    if (in[i].host && in[i].bytes) total += aligned(in[i].bytes);
  int rc = spx_require_not_capturing(ctx, "a host-pointer call (it copies and synchronises)");
  if (rc) return rc;
  rc = stage_reserve(ctx, total ? total : kAlign);
  if (rc) return rc;
  SPX_ON_DEVICE(ctx);
  char* p = static_cast<char*>(ctx->stage);
  if (dev_out) *dev_out = out_bytes ? p : nullptr;
  p += aligned(out_bytes);
  for (int i = 0; i < NIN; ++i) {
    dev[i] = nullptr;
    if (in[i].host && in[i].bytes) {
      SPX_HIP(hipMemcpyAsync(p, in[i].host, in[i].bytes, hipMemcpyHostToDevice, ctx->stream));
      dev[i] = p;
      p += aligned(in[i].bytes);
    }
  }
  return SPX_OK;
}

int stage_out(spx_ctx* ctx, void* host, const void* dev, size_t bytes) {
  if (bytes) SPX_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
  SPX_HIP(hipStreamSynchronize(ctx->stream));
  return SPX_OK;
}

inline size_t vbytes(int64_t n) { return n > 0 ? (size_t)n * sizeof(double) : 0; }
inline const double* D(const void* p) { return static_cast<const double*>(p); }

int check_host(spx_ctx* ctx, const void* y, const void* a, const void* b, const void* c, int64_t n) {
  SPX_REQUIRE(ctx != nullptr, "ctx is NULL");
  SPX_REQUIRE(n >= 0, "n < 0");
  if (n > 0) SPX_REQUIRE(y && a && b && c, "NULL vector with n > 0");
  return SPX_OK;
}

// y <- f(q, xk, sj): the three-vector operators
template <class F>
int host_sep(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n, F&& f) {
  int rc = check_host(ctx, y, q, xk, sj, n);
  if (rc) return rc;
  const In in[3] = {{q, vbytes(n)}, {xk, vbytes(n)}, {sj, vbytes(n)}};
  const void* d[3];
  void* dy;
  rc = stage_in(ctx, in, d, vbytes(n), &dy);
  if (rc) return rc;
  // prox!(q, psi, q, sigma) on host vectors: the device twin must see the aliasing too (ShiftedNormL1's y === q body differs
  // from the disjoint one, src/shiftedNormL1.jl:47-51; every kernel handles y === q): y is then the staged q itself
  if (y == q && n > 0) dy = const_cast<void*>(d[0]);
  rc = f(static_cast<double*>(dy), D(d[0]), D(d[1]), D(d[2]));
  if (rc) return rc;
  return stage_out(ctx, y, dy, vbytes(n));
}

// the Box operators: + optional l, u vectors and the selection mask
template <class F>
int host_box(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
             const double* l_vec, const double* u_vec, const uint8_t* mask, F&& f) {
  int rc = check_host(ctx, y, q, xk, sj, n);
  if (rc) return rc;
  const In in[6] = {{q, vbytes(n)}, {xk, vbytes(n)}, {sj, vbytes(n)}, {l_vec, vbytes(n)}, {u_vec, vbytes(n)},
                    {mask, n > 0 ? (size_t)n : 0}};
  const void* d[6];
  void* dy;
  rc = stage_in(ctx, in, d, vbytes(n), &dy);
  if (rc) return rc;
  if (y == q && n > 0) dy = const_cast<void*>(d[0]);  // y === q, as in host_sep
  rc = f(static_cast<double*>(dy), D(d[0]), D(d[1]), D(d[2]), D(d[3]), D(d[4]), static_cast<const uint8_t*>(d[5]));
  if (rc) return rc;
  return stage_out(ctx, y, dy, vbytes(n));
}

// iprox!: g, d, xk, sj (+ Box extras)
template <class F>
int host_iprox(spx_ctx* ctx, double* y, const double* g, const double* dd, const double* xk, const double* sj,
               int64_t n, const double* l_vec, const double* u_vec, const uint8_t* mask, F&& f) {
  int rc = check_host(ctx, y, g, xk, sj, n);
  if (rc) return rc;
  if (n > 0) SPX_REQUIRE(dd != nullptr, "d is NULL");
  const In in[7] = {{g, vbytes(n)}, {dd, vbytes(n)}, {xk, vbytes(n)}, {sj, vbytes(n)}, {l_vec, vbytes(n)},
                    {u_vec, vbytes(n)}, {mask, n > 0 ? (size_t)n : 0}};
  const void* d[7];
  void* dy;
  rc = stage_in(ctx, in, d, vbytes(n), &dy);
  if (rc) return rc;
  rc = f(static_cast<double*>(dy), D(d[0]), D(d[1]), D(d[2]), D(d[3]), D(d[4]), D(d[5]),
         static_cast<const uint8_t*>(d[6]));
  if (rc) {  // SPX_ERR_ASSERT included: y is unspecified, as after the reference's exception
    (void)hipStreamSynchronize(ctx->stream);
    return rc;
  }
  return stage_out(ctx, y, dy, vbytes(n));
}

// group operators: + CSR offsets and lambda_vec
template <class F>
int host_group(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
               const int64_t* offsets, int64_t ngroups, const double* lambda_vec, F&& f) {
  int rc = check_host(ctx, y, q, xk, sj, n);
  if (rc) return rc;
  SPX_REQUIRE(ngroups >= 0, "ngroups < 0");
  const In in[5] = {{q, vbytes(n)}, {xk, vbytes(n)}, {sj, vbytes(n)},
                    {offsets, offsets ? (size_t)(ngroups + 1) * sizeof(int64_t) : 0}, {lambda_vec, vbytes(ngroups)}};
  const void* d[5];
  void* dy;
  rc = stage_in(ctx, in, d, vbytes(n), &dy);
  if (rc) return rc;
  rc = f(static_cast<double*>(dy), D(d[0]), D(d[1]), D(d[2]), static_cast<const int64_t*>(d[3]), D(d[4]));
  if (rc) return rc;
  return stage_out(ctx, y, dy, vbytes(n));
}

// gather-index groups: + group_ptr, group_index and lambda_vec
template <class F>
int host_gather(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj, int64_t n,
                const int64_t* ptr, const int64_t* index, int64_t ngroups, int64_t nnz, const double* lambda_vec,
                F&& f) {
  int rc = check_host(ctx, y, q, xk, sj, n);
  if (rc) return rc;
  SPX_REQUIRE(ngroups >= 0 && nnz >= 0, "negative size");
  // y is an input as well: indices no group contains keep the caller's value
  const In in[7] = {{q, vbytes(n)}, {xk, vbytes(n)}, {sj, vbytes(n)}, {ptr, ptr ? (size_t)(ngroups + 1) * sizeof(int64_t) : 0},
                    {index, (size_t)nnz * sizeof(int64_t)}, {lambda_vec, vbytes(ngroups)}, {y, vbytes(n)}};
  const void* d[7];
  rc = stage_in(ctx, in, d, 0, nullptr);
  if (rc) return rc;
  double* dy = const_cast<double*>(D(d[6]));
  rc = f(dy, D(d[0]), D(d[1]), D(d[2]), static_cast<const int64_t*>(d[3]), static_cast<const int64_t*>(d[4]), D(d[5]));
  if (rc) return rc;
  return stage_out(ctx, y, dy, vbytes(n));
}

// psi(y): y, xk, sj (+ extras), value returned by the device twin (which synchronises)
template <class F>
int host_obj(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n, const double* a,
             size_t a_bytes, const double* b, size_t b_bytes, const void* c, size_t c_bytes, F&& f) {
  int rc = check_host(ctx, y, y, xk, sj, n);
  if (rc) return rc;
  const In in[6] = {{y, vbytes(n)}, {xk, vbytes(n)}, {sj, vbytes(n)}, {a, a_bytes}, {b, b_bytes}, {c, c_bytes}};
  const void* d[6];
  rc = stage_in(ctx, in, d, 0, nullptr);
  if (rc) return rc;
  return f(D(d[0]), D(d[1]), D(d[2]), d[3], d[4], d[5]);
}

}  // namespace

// ---- prox! ------------------------------------------------------------------------------------
#define SPX_HOST_SEP(name)                                                                                        \
  SPX_EXPORT int spx_host_##name(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,    \
                                 int64_t n, double lambda, double sigma) {                                        \
    return host_sep(ctx, y, q, xk, sj, n, [&](double* dy, const double* dq, const double* dx, const double* ds) { \
      return spx_##name(ctx, dy, dq, dx, ds, n, lambda, sigma);                                                   \
    });                                                                                                           \
  }
SPX_HOST_SEP(prox_l1)
SPX_HOST_SEP(prox_l0)
SPX_HOST_SEP(prox_lhalf)
#undef SPX_HOST_SEP

#define SPX_HOST_BOX(name)                                                                                          \
  SPX_EXPORT int spx_host_##name(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,      \
                                 int64_t n, double lambda, double sigma, const double* l_vec, const double* u_vec,  \
                                 double l_scalar, double u_scalar, const uint8_t* sel_mask) {                       \
    return host_box(ctx, y, q, xk, sj, n, l_vec, u_vec, sel_mask,                                                   \
                    [&](double* dy, const double* dq, const double* dx, const double* ds, const double* dl,         \
                        const double* du, const uint8_t* dm) {                                                      \
                      return spx_##name(ctx, dy, dq, dx, ds, n, lambda, sigma, dl, du, l_scalar, u_scalar, dm);     \
                    });                                                                                             \
  }
SPX_HOST_BOX(prox_l1_box)
SPX_HOST_BOX(prox_l0_box)
SPX_HOST_BOX(prox_lhalf_box)
#undef SPX_HOST_BOX

SPX_EXPORT int spx_host_prox_indball_l0(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                                        int64_t n, int64_t r) {
  return host_sep(ctx, y, q, xk, sj, n, [&](double* dy, const double* dq, const double* dx, const double* ds) {
    return spx_prox_indball_l0(ctx, dy, dq, dx, ds, n, r);
  });
}
SPX_EXPORT int spx_host_prox_indball_l0_binf(spx_ctx* ctx, double* y, const double* q, const double* xk,
                                             const double* sj, int64_t n, int64_t r, double delta) {
  return host_sep(ctx, y, q, xk, sj, n, [&](double* dy, const double* dq, const double* dx, const double* ds) {
    return spx_prox_indball_l0_binf(ctx, dy, dq, dx, ds, n, r, delta);
  });
}
SPX_EXPORT int spx_host_prox_l1_b2(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                                   int64_t n, double lambda, double sigma, double delta, double chi_lambda) {
  return host_sep(ctx, y, q, xk, sj, n, [&](double* dy, const double* dq, const double* dx, const double* ds) {
    return spx_prox_l1_b2(ctx, dy, dq, dx, ds, n, lambda, sigma, delta, chi_lambda);
  });
}
SPX_EXPORT int spx_host_prox_group_l2(spx_ctx* ctx, double* y, const double* q, const double* xk, const double* sj,
                                      int64_t n, const int64_t* group_offsets, int64_t group_size, int64_t ngroups,
                                      const double* lambda_vec, double sigma) {
  return host_group(ctx, y, q, xk, sj, n, group_offsets, ngroups, lambda_vec,
                    [&](double* dy, const double* dq, const double* dx, const double* ds, const int64_t* doff,
                        const double* dlam) {
                      return spx_prox_group_l2(ctx, dy, dq, dx, ds, n, doff, group_size, ngroups, dlam, sigma);
                    });
}
SPX_EXPORT int spx_host_prox_group_l2_binf(spx_ctx* ctx, double* y, const double* q, const double* xk,
                                           const double* sj, int64_t n, const int64_t* group_offsets,
                                           int64_t group_size, int64_t ngroups, const double* lambda_vec, double sigma,
                                           double delta) {
  return host_group(ctx, y, q, xk, sj, n, group_offsets, ngroups, lambda_vec,
                    [&](double* dy, const double* dq, const double* dx, const double* ds, const int64_t* doff,
                        const double* dlam) {
                      return spx_prox_group_l2_binf(ctx, dy, dq, dx, ds, n, doff, group_size, ngroups, dlam, sigma,
                                                    delta);
                    });
}

SPX_EXPORT int spx_host_prox_group_l2_gather(spx_ctx* ctx, double* y, const double* q, const double* xk,
                                             const double* sj, int64_t n, const int64_t* group_ptr,
                                             const int64_t* group_index, int64_t ngroups, int64_t nnz,
                                             const double* lambda_vec, double sigma) {
  return host_gather(ctx, y, q, xk, sj, n, group_ptr, group_index, ngroups, nnz, lambda_vec,
                     [&](double* dy, const double* dq, const double* dx, const double* ds, const int64_t* dp,
                         const int64_t* di, const double* dlam) {
                       return spx_prox_group_l2_gather(ctx, dy, dq, dx, ds, n, dp, di, ngroups, nnz, dlam, sigma);
                     });
}
SPX_EXPORT int spx_host_prox_group_l2_binf_gather(spx_ctx* ctx, double* y, const double* q, const double* xk,
                                                  const double* sj, int64_t n, const int64_t* group_ptr,
                                                  const int64_t* group_index, int64_t ngroups, int64_t nnz,
                                                  const double* lambda_vec, double sigma, double delta) {
  return host_gather(ctx, y, q, xk, sj, n, group_ptr, group_index, ngroups, nnz, lambda_vec,
                     [&](double* dy, const double* dq, const double* dx, const double* ds, const int64_t* dp,
                         const int64_t* di, const double* dlam) {
                       return spx_prox_group_l2_binf_gather(ctx, dy, dq, dx, ds, n, dp, di, ngroups, nnz, dlam, sigma,
                                                            delta);
                     });
}

// ---- iprox! -----------------------------------------------------------------------------------
#define SPX_HOST_IPROX(name)                                                                                         \
  SPX_EXPORT int spx_host_##name(spx_ctx* ctx, double* y, const double* g, const double* d, const double* xk,        \
                                 const double* sj, int64_t n, double lambda, int check_d) {                          \
    return host_iprox(ctx, y, g, d, xk, sj, n, nullptr, nullptr, nullptr,                                            \
                      [&](double* dy, const double* dg, const double* dd, const double* dx, const double* ds,        \
                          const double*, const double*, const uint8_t*) {                                            \
                        return spx_##name(ctx, dy, dg, dd, dx, ds, n, lambda, check_d);                              \
                      });                                                                                            \
  }
SPX_HOST_IPROX(iprox_l1)
SPX_HOST_IPROX(iprox_l0)
#undef SPX_HOST_IPROX

#define SPX_HOST_IPROX_BOX(name)                                                                                     \
  SPX_EXPORT int spx_host_##name(spx_ctx* ctx, double* y, const double* g, const double* d, const double* xk,        \
                                 const double* sj, int64_t n, double lambda, const double* l_vec,                    \
                                 const double* u_vec, double l_scalar, double u_scalar, const uint8_t* sel_mask) {   \
    return host_iprox(ctx, y, g, d, xk, sj, n, l_vec, u_vec, sel_mask,                                               \
                      [&](double* dy, const double* dg, const double* dd, const double* dx, const double* ds,        \
                          const double* dl, const double* du, const uint8_t* dm) {                                   \
                        return spx_##name(ctx, dy, dg, dd, dx, ds, n, lambda, dl, du, l_scalar, u_scalar, dm);       \
                      });                                                                                            \
  }
SPX_HOST_IPROX_BOX(iprox_l1_box)
SPX_HOST_IPROX_BOX(iprox_l0_box)
#undef SPX_HOST_IPROX_BOX

// ---- psi(y) -----------------------------------------------------------------------------------
#define SPX_HOST_OBJ(name)                                                                                          \
  SPX_EXPORT int spx_host_##name(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,      \
                                 double lambda, double* value) {                                                    \
    return host_obj(ctx, y, xk, sj, n, nullptr, 0, nullptr, 0, nullptr, 0,                                          \
                    [&](const double* dy, const double* dx, const double* ds, const void*, const void*,             \
                        const void*) { return spx_##name(ctx, dy, dx, ds, n, lambda, value); });                    \
  }
SPX_HOST_OBJ(obj_l1)
SPX_HOST_OBJ(obj_l0)
SPX_HOST_OBJ(obj_lhalf)
#undef SPX_HOST_OBJ

#define SPX_HOST_OBJ_BOX(name)                                                                                       \
  SPX_EXPORT int spx_host_##name(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,       \
                                 double lambda, const double* l_vec, const double* u_vec, double l_scalar,           \
                                 double u_scalar, const uint8_t* sel_mask, double* value) {                          \
    return host_obj(ctx, y, xk, sj, n, l_vec, l_vec ? vbytes(n) : 0, u_vec, u_vec ? vbytes(n) : 0, sel_mask,         \
                    sel_mask && n > 0 ? (size_t)n : 0,                                                               \
                    [&](const double* dy, const double* dx, const double* ds, const void* dl, const void* du,        \
                        const void* dm) {                                                                            \
                      return spx_##name(ctx, dy, dx, ds, n, lambda, D(dl), D(du), l_scalar, u_scalar,                \
                                        static_cast<const uint8_t*>(dm), value);                                     \
                    });                                                                                              \
  }
SPX_HOST_OBJ_BOX(obj_l1_box)
SPX_HOST_OBJ_BOX(obj_l0_box)
SPX_HOST_OBJ_BOX(obj_lhalf_box)
#undef SPX_HOST_OBJ_BOX

SPX_EXPORT int spx_host_obj_indball_l0(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,
                                       int64_t r, double* value) {
  return host_obj(ctx, y, xk, sj, n, nullptr, 0, nullptr, 0, nullptr, 0,
                  [&](const double* dy, const double* dx, const double* ds, const void*, const void*, const void*) {
                    return spx_obj_indball_l0(ctx, dy, dx, ds, n, r, value);
                  });
}
SPX_EXPORT int spx_host_obj_indball_l0_binf(spx_ctx* ctx, const double* y, const double* xk, const double* sj,
                                            int64_t n, int64_t r, double delta, double* value) {
  return host_obj(ctx, y, xk, sj, n, nullptr, 0, nullptr, 0, nullptr, 0,
                  [&](const double* dy, const double* dx, const double* ds, const void*, const void*, const void*) {
                    return spx_obj_indball_l0_binf(ctx, dy, dx, ds, n, r, delta, value);
                  });
}
SPX_EXPORT int spx_host_obj_group_l2(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,
                                     const int64_t* group_offsets, int64_t group_size, int64_t ngroups,
                                     const double* lambda_vec, double* value) {
  SPX_REQUIRE(ngroups >= 0, "ngroups < 0");
  return host_obj(ctx, y, xk, sj, n, lambda_vec, vbytes(ngroups), nullptr, 0, group_offsets,
                  group_offsets ? (size_t)(ngroups + 1) * sizeof(int64_t) : 0,
                  [&](const double* dy, const double* dx, const double* ds, const void* dlam, const void*,
                      const void* doff) {
                    return spx_obj_group_l2(ctx, dy, dx, ds, n, static_cast<const int64_t*>(doff), group_size, ngroups,
                                            D(dlam), value);
                  });
}
SPX_EXPORT int spx_host_obj_group_l2_binf(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,
                                          const int64_t* group_offsets, int64_t group_size, int64_t ngroups,
                                          const double* lambda_vec, double delta, double* value) {
  SPX_REQUIRE(ngroups >= 0, "ngroups < 0");
  return host_obj(ctx, y, xk, sj, n, lambda_vec, vbytes(ngroups), nullptr, 0, group_offsets,
                  group_offsets ? (size_t)(ngroups + 1) * sizeof(int64_t) : 0,
                  [&](const double* dy, const double* dx, const double* ds, const void* dlam, const void*,
                      const void* doff) {
                    return spx_obj_group_l2_binf(ctx, dy, dx, ds, n, static_cast<const int64_t*>(doff), group_size,
                                                 ngroups, D(dlam), delta, value);
                  });
}

// gather objective: y, xk, sj, lambda_vec, group_ptr, group_index
template <class F>
static int host_obj_gather(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,
                           const int64_t* ptr, const int64_t* index, int64_t ngroups, int64_t nnz,
                           const double* lambda_vec, F&& f) {
  int rc = check_host(ctx, y, y, xk, sj, n);
  if (rc) return rc;
  SPX_REQUIRE(ngroups >= 0 && nnz >= 0, "negative size");
  const In in[6] = {{y, vbytes(n)}, {xk, vbytes(n)}, {sj, vbytes(n)}, {lambda_vec, vbytes(ngroups)},
                    {ptr, ptr ? (size_t)(ngroups + 1) * sizeof(int64_t) : 0}, {index, (size_t)nnz * sizeof(int64_t)}};
  const void* d[6];
  rc = stage_in(ctx, in, d, 0, nullptr);
  if (rc) return rc;
  return f(D(d[0]), D(d[1]), D(d[2]), D(d[3]), static_cast<const int64_t*>(d[4]), static_cast<const int64_t*>(d[5]));
}
SPX_EXPORT int spx_host_obj_group_l2_gather(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,
                                            const int64_t* group_ptr, const int64_t* group_index, int64_t ngroups,
                                            int64_t nnz, const double* lambda_vec, double* value) {
  return host_obj_gather(ctx, y, xk, sj, n, group_ptr, group_index, ngroups, nnz, lambda_vec,
                         [&](const double* dy, const double* dx, const double* ds, const double* dlam, const int64_t* dp,
                             const int64_t* di) {
                           return spx_obj_group_l2_gather(ctx, dy, dx, ds, n, dp, di, ngroups, nnz, dlam, value);
                         });
}
SPX_EXPORT int spx_host_obj_group_l2_binf_gather(spx_ctx* ctx, const double* y, const double* xk, const double* sj,
                                                 int64_t n, const int64_t* group_ptr, const int64_t* group_index,
                                                 int64_t ngroups, int64_t nnz, const double* lambda_vec, double delta,
                                                 double* value) {
  return host_obj_gather(ctx, y, xk, sj, n, group_ptr, group_index, ngroups, nnz, lambda_vec,
                         [&](const double* dy, const double* dx, const double* ds, const double* dlam, const int64_t* dp,
                             const int64_t* di) {
                           return spx_obj_group_l2_binf_gather(ctx, dy, dx, ds, n, dp, di, ngroups, nnz, dlam, delta,
                                                               value);
                         });
}
SPX_EXPORT int spx_host_obj_l1_b2(spx_ctx* ctx, const double* y, const double* xk, const double* sj, int64_t n,
                                  double lambda, double delta, double* value) {
  return host_obj(ctx, y, xk, sj, n, nullptr, 0, nullptr, 0, nullptr, 0,
                  [&](const double* dy, const double* dx, const double* ds, const void*, const void*, const void*) {
                    return spx_obj_l1_b2(ctx, dy, dx, ds, n, lambda, delta, value);
                  });
}
