/* A plain C host for libspx: no Python, no torch, only include/spx.h and the HIP runtime C API for device memory.
 * Built and run by tests/test_gpu_c_host.py.  Checks the headline operator (ShiftedNormL1Box.prox!,
 * src/shiftedNormL1Box.jl:89-125) through the device-pointer entry point and through the host-pointer twin against
 * the reference formula written out below, bit for bit.  Exit code 0 = pass. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "spx.h"

static double jl_min(double x, double y) { double d = x - y; return (x != x || y != y) ? d : (signbit(d) ? x : y); }
static double jl_max(double x, double y) { double d = x - y; return (x != x || y != y) ? d : (signbit(d) ? y : x); }

/* one element of src/shiftedNormL1Box.jl:99-118, all indices selected, scalar bounds */
static double ref_l1box(double q, double x, double s, double sl, double l, double u) {
  double xs = x + s, xsq = xs + q, t;
  if (xsq <= -sl) t = q + sl;
  else if (xsq >= sl) t = q - sl;
  else t = -xs;
  return jl_min(jl_max(t, l - s), u - s);
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double urand(void) { /* splitmix64 -> (-1, 1) */
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return ((double)(z >> 11) / 9007199254740992.0) * 2.0 - 1.0;
}

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, spx_last_error()); return 2; } } while (0)
#define HIPCHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 3; } } while (0)

int main(void) {
  const int64_t n = 1000003;
  const double lambda = 0.7, sigma = 1.3, delta = 0.9;
  const size_t bytes = (size_t)n * sizeof(double);
  double *q = malloc(bytes), *x = malloc(bytes), *s = malloc(bytes), *y = malloc(bytes), *yh = malloc(bytes);
  for (int64_t i = 0; i < n; ++i) { q[i] = 2.0 * urand(); x[i] = 2.0 * urand(); s[i] = 0.5 * urand(); }
  if (spx_abi_version() != SPX_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }

  spx_ctx* ctx = NULL;
  CHECK(spx_ctx_create(0, &ctx));
  double *dq, *dx, *ds, *dy;
  HIPCHECK(hipMalloc((void**)&dq, bytes)); HIPCHECK(hipMalloc((void**)&dx, bytes));
  HIPCHECK(hipMalloc((void**)&ds, bytes)); HIPCHECK(hipMalloc((void**)&dy, bytes));
  HIPCHECK(hipMemcpy(dq, q, bytes, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(dx, x, bytes, hipMemcpyHostToDevice));
  HIPCHECK(hipMemcpy(ds, s, bytes, hipMemcpyHostToDevice));

  /* device-pointer form: asynchronous on the context's stream */
  CHECK(spx_prox_l1_box(ctx, dy, dq, dx, ds, n, lambda, sigma, NULL, NULL, -delta, delta, NULL));
  CHECK(spx_sync(ctx));
  HIPCHECK(hipMemcpy(y, dy, bytes, hipMemcpyDeviceToHost));
  /* host-pointer twin: same kernels behind a staging copy */
  CHECK(spx_host_prox_l1_box(ctx, yh, q, x, s, n, lambda, sigma, NULL, NULL, -delta, delta, NULL));

  int64_t bad = 0;
  for (int64_t i = 0; i < n; ++i) {
    double r = ref_l1box(q[i], x[i], s[i], sigma * lambda, -delta, delta);
    if (memcmp(&r, &y[i], 8) != 0 || memcmp(&r, &yh[i], 8) != 0) {
      if (bad++ < 5) fprintf(stderr, "i=%lld ref=%.17g dev=%.17g host=%.17g\n", (long long)i, r, y[i], yh[i]);
    }
  }
  /* psi(y) at the prox: inside the box by construction -> finite, equal through both forms */
  double v1 = -1.0, v2 = -2.0;
  CHECK(spx_obj_l1_box(ctx, dy, dx, ds, n, lambda, NULL, NULL, -delta, delta, NULL, &v1));
  CHECK(spx_host_obj_l1_box(ctx, yh, x, s, n, lambda, NULL, NULL, -delta, delta, NULL, &v2));
  if (!(v1 == v2) || !isfinite(v1)) { fprintf(stderr, "objective mismatch %.17g vs %.17g\n", v1, v2); ++bad; }
  /* error path: a NULL vector must be refused with a message, not crash */
  if (spx_prox_l1(ctx, NULL, dq, dx, ds, n, lambda, sigma) != SPX_ERR_INVALID_ARG || strlen(spx_last_error()) == 0) ++bad;

  hipFree(dq); hipFree(dx); hipFree(ds); hipFree(dy);
  CHECK(spx_ctx_destroy(ctx));
  free(q); free(x); free(s); free(y); free(yh);
  printf("abi_driver: n=%lld mismatches=%lld psi=%.6f\n", (long long)n, (long long)bad, v1);
  return bad ? 1 : 0;
}
