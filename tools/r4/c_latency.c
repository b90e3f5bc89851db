/* Round 4: per-call latency through the C ABI alone (no Python, no torch): what a Julia `ccall` pays.  1000 back-to-back calls
 * of each entry point at solver sizes; host issue time (clock_gettime around the loop) and stream time (spx_timer_*), per call.
 * Build: gcc -O2 -std=c11 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include tools/r4/c_latency.c -o gpurun_out/c_latency \
 *        -Lshiftedproximaloperators.jl_amd/lib -lspx -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,... (tools/r4/c_latency.sh) */
#define _POSIX_C_SOURCE 200809L
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#include "spx.h"

#define CHECK(call) do { int rc_ = (call); if (rc_) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, spx_last_error()); return 2; } } while (0)
#define HIPCHECK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); return 3; } } while (0)
static double now_us(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }

int main(void) {
  const int64_t sizes[] = {10000, 100000, 1000000};
  spx_ctx* ctx = NULL;
  CHECK(spx_ctx_create(0, &ctx));
  const int64_t nmax = 1000000;
  double *q, *x, *s, *y, *lam, *val;
  HIPCHECK(hipMalloc((void**)&q, nmax * 8)); HIPCHECK(hipMalloc((void**)&x, nmax * 8)); HIPCHECK(hipMalloc((void**)&s, nmax * 8));
  HIPCHECK(hipMalloc((void**)&y, nmax * 8)); HIPCHECK(hipMalloc((void**)&lam, nmax * 8)); HIPCHECK(hipMalloc((void**)&val, 8));
  CHECK(spx_synth_fill(ctx, x, nmax, 1, 0, 1, 1.0)); CHECK(spx_synth_fill(ctx, s, nmax, 1, 1, 0, 1.0));
  CHECK(spx_synth_fill(ctx, q, nmax, 1, 2, 1, 1.0));
  {  /* group weights in [0.5, 1.5) (the constructor refuses negative ones, src/groupNormL2.jl:20-21) */
    double* hl = malloc(nmax * 8);
    for (int64_t i = 0; i < nmax; ++i) hl[i] = 0.5 + (double)(i % 1000) / 1000.0;
    HIPCHECK(hipMemcpy(lam, hl, nmax * 8, hipMemcpyHostToDevice));
    free(hl);
  }
  CHECK(spx_ctx_set_value_target(ctx, val));
  CHECK(spx_sync(ctx));
  printf("us per call through the C ABI: stream time (host issue time), 1000 back-to-back calls, best of 5\n");
  printf("%-34s %18s %18s %18s\n", "", "n = 1e4", "n = 1e5", "n = 1e6");
  const char* names[] = {"spx_prox_l1_box", "spx_obj_l1_box -> device", "spx_proxval_l1_box -> device", "spx_prox_group_l2 (groups of 8)",
                         "spx_prox_group_l2_binf (of 8)", "spx_obj_group_l2_binf (of 8)", "spx_prox_indball_l0_binf r=n/100", "spx_prox_l1_b2"};
  for (int op = 0; op < 8; ++op) {
    printf("%-34s", names[op]);
    for (int k = 0; k < 3; ++k) {
      const int64_t n = sizes[k];
      double best_ev = 1e30, best_host = 1e30, hv = 0.0;
      CHECK(spx_synth_fill(ctx, y, nmax, 1, 4, 0, 0.5)); /* the point psi is evaluated at: inside the box */
      for (int rnd = 0; rnd < 6; ++rnd) {
        float ms = 0.0f;
        const double t0 = now_us();
        CHECK(spx_timer_start(ctx));
        for (int it = 0; it < 1000; ++it) {
          switch (op) {
            case 0: CHECK(spx_prox_l1_box(ctx, y, q, x, s, n, 1.0, 1.0, NULL, NULL, -1.0, 1.0, NULL)); break;
            case 1: CHECK(spx_obj_l1_box(ctx, y, x, s, n, 1.0, NULL, NULL, -1.0, 1.0, NULL, &hv)); break;
            case 2: CHECK(spx_proxval_l1_box(ctx, y, q, x, s, n, 1.0, 1.0, NULL, NULL, -1.0, 1.0, NULL, 1.0, &hv)); break;
            case 3: CHECK(spx_prox_group_l2(ctx, y, q, x, s, n, NULL, 8, n / 8, lam, 1.0)); break;
            case 4: CHECK(spx_prox_group_l2_binf(ctx, y, q, x, s, n, NULL, 8, n / 8, lam, 1.0, 1.0)); break;
            case 5: CHECK(spx_obj_group_l2_binf(ctx, y, x, s, n, NULL, 8, n / 8, lam, 1.0, &hv)); break;
            case 6: CHECK(spx_prox_indball_l0_binf(ctx, y, q, x, s, n, n / 100, 1.0)); break;
            default: CHECK(spx_prox_l1_b2(ctx, y, q, x, s, n, 1.0, 1.0, 1.0, 1.0)); break;
          }
        }
        const double t1 = now_us();
        CHECK(spx_timer_stop(ctx, &ms));
        if (rnd == 0) continue; /* warm-up */
        if (ms < best_ev) best_ev = ms;
        if ((t1 - t0) / 1000.0 < best_host) best_host = (t1 - t0) / 1000.0;
      }
      printf("   %7.2f (%6.2f)", best_ev, best_host);
    }
    printf("\n");
  }
  CHECK(spx_ctx_set_value_target(ctx, NULL));
  CHECK(spx_ctx_destroy(ctx));
  return 0;
}
