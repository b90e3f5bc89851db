// Microbenchmark (round 2): time per all-to-all exchange of one word per workgroup inside a launch on gfx950, for the forms
// spx_b2.hip could use.  build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/exchange tools/exp/exchange.hip ; run on the GPU box.
//   V = 0  flag-in-data: workgroup b stores word (round, b); lanes t < G poll word t until it carries this round (sc1 load)
//   V = 1  the same, polled with an atomic RMW (fetch_add 0) instead of a load
//   V = 2  counter rendezvous (store, wait, arrive, poll the counter), then load the words
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int V>
__global__ __launch_bounds__(1024) void k(unsigned long long* words, unsigned int* counter, int rounds, unsigned long long* sink) {
  const int t = threadIdx.x, G = gridDim.x;
  unsigned long long acc = 0;
  __shared__ unsigned long long sh;
  for (int r = 1; r <= rounds; ++r) {
    unsigned long long* row = words + (size_t)(r & 63) * 512 * 8;  // 64 rows, reused every 64 rounds (values grow with r)
    if (t == 0) __hip_atomic_store(row + blockIdx.x * 8, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (V == 2) {
      __syncthreads();
      if (t == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)r * G) __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
    }
    unsigned long long w = 0;
    if (t < G) {
      for (;;) {
        if (V == 1) w = __hip_atomic_fetch_add(row + t * 8, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else w = __hip_atomic_load(row + t * 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (w >= (unsigned long long)r) break;
        __builtin_amdgcn_s_sleep(1);
      }
    }
    // workgroup-wide use of the result (as the block sum of the partials would be)
    for (int off = 32; off; off >>= 1) w += __shfl_down(w, off, 64);
    if (t == 0) sh = w;
    __syncthreads();
    acc += sh;
    __syncthreads();
  }
  if (acc == 12345 && t == 0) sink[0] = acc;
}

template <int V>
int run(const char* name, int grid, int rounds) {
  unsigned long long *words, *sink; unsigned int* c;
  CK(hipMalloc(&words, 64 * 512 * 8 * 8)); CK(hipMalloc(&c, 256)); CK(hipMalloc(&sink, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f, best0 = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    for (int pass = 0; pass < 2; ++pass) {
      CK(hipMemset(words, 0, 64 * 512 * 8 * 8)); CK(hipMemset(c, 0, 256));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k<V>, dim3(grid), dim3(1024), 0, 0, words, c, pass ? rounds : 0, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (pass) best = ms < best ? ms : best; else best0 = ms < best0 ? ms : best0;
    }
  }
  printf("%-44s grid %4d: %6.2f us per exchange\n", name, grid, (best - best0) * 1e3f / rounds);
  CK(hipFree(words)); CK(hipFree(c)); CK(hipFree(sink));
  return 0;
}

int main() {
  const int rounds = 60;  // < 64: a row is never reused inside a launch
  for (int g : {1, 2, 8, 16, 32, 64, 128, 256}) {
    run<0>("flag in data, polled with sc1 loads", g, rounds);
    run<1>("flag in data, polled with atomic RMW", g, rounds);
    run<2>("counter rendezvous, then loads", g, rounds);
  }
  return 0;
}
