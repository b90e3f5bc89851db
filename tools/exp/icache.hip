// Microbenchmark (round 2): what does COLD CODE cost at the start of a launch on gfx950?  The same 4096 dependent-free fp64
// FMAs per lane as a loop (small code) and fully unrolled (32 KiB of straight-line code, executed once), 1..256 workgroups.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/icache tools/exp/icache.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <bool UNROLL>
__global__ __launch_bounds__(256) void k(double* out, double a, double b, int n) {
  double x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  if (UNROLL) {
#pragma unroll
    for (int i = 0; i < 512; ++i) {
      x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b);
      x4 = __builtin_fma(x4, a, b); x5 = __builtin_fma(x5, a, b); x6 = __builtin_fma(x6, a, b); x7 = __builtin_fma(x7, a, b);
    }
  } else {
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
      x0 = __builtin_fma(x0, a, b); x1 = __builtin_fma(x1, a, b); x2 = __builtin_fma(x2, a, b); x3 = __builtin_fma(x3, a, b);
      x4 = __builtin_fma(x4, a, b); x5 = __builtin_fma(x5, a, b); x6 = __builtin_fma(x6, a, b); x7 = __builtin_fma(x7, a, b);
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = ((x0 + x1) + (x2 + x3)) + ((x4 + x5) + (x6 + x7));
}

template <bool UNROLL>
int run(const char* name, int grid) {
  double* out; CK(hipMalloc(&out, 256 * 256 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f;
  for (int rep = 0; rep < 20; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(k<UNROLL>, dim3(grid), dim3(256), 0, 0, out, 0.999, 0.001, 512);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  printf("%-28s grid %4d: %7.2f us per launch (best of 20)\n", name, grid, best * 1e3f);
  CK(hipFree(out));
  return 0;
}
int main() {
  for (int g : {1, 16, 256}) { run<false>("loop (small code)", g); run<true>("unrolled (32 KiB of code)", g); }
  return 0;
}
