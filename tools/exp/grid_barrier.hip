// Microbenchmark of in-launch grid barriers on gfx950 (round 2): time per barrier for the forms tried in spx_common.hpp.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/grid_barrier tools/exp/grid_barrier.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int V>
__device__ __forceinline__ void barrier(unsigned int* c, unsigned int target, unsigned int* xc, unsigned int* gen, unsigned int phase) {
  if (V == 0) {  // flat counter, release + acquire fences, relaxed poll
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  } else if (V == 1) {  // no fences (measurement only)
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
  } else if (V == 2) {  // flat, longer sleep
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(20);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  } else if (V == 3) {  // the arriving atomic returns the count: the LAST arriver publishes a generation word, the others poll THAT
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned int old = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == target) __hip_atomic_store(gen, phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else while (__hip_atomic_load(gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase) __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  } else if (V == 5 || V == 6) {  // NO fences, 8 / 16 split counters (one 128-byte line each), polled by 8 / 16 lanes of wave 0
    constexpr int NC = V == 5 ? 8 : 16;
    __syncthreads();
    if (threadIdx.x < 64) {
      const int lane = threadIdx.x;
      if (lane == 0) __hip_atomic_fetch_add(&xc[32 * (blockIdx.x % NC)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned int want = lane < NC ? ((gridDim.x - lane + NC - 1) / NC) * phase : 0u;
      while (true) {
        const unsigned int have = lane < NC ? __hip_atomic_load(&xc[32 * lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
        if (__ballot(have < want) == 0ull) break;
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
  } else if (V == 7) {  // NO fences, tree: 16 group counters, the last arriver of a group adds to the top counter, all poll the top
    __syncthreads();
    if (threadIdx.x == 0) {
      const unsigned int grp = blockIdx.x % 16u;
      const unsigned int members = (gridDim.x - grp + 15u) / 16u;
      const unsigned int old = __hip_atomic_fetch_add(&xc[32 * grp], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == members * phase) __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned int groups = gridDim.x < 16u ? gridDim.x : 16u;
      while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < groups * phase) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  } else if (V == 8) {  // NO fences, flat, sleep 1 (spx_grid_rendezvous as shipped)
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
  } else if (V == 4) {  // per-XCC counters (8 lines) + one top counter + per-XCC generation words
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned int xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      xcc &= 7u;
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // xc[32*x]: arrivals on XCC x (monotonic); xc[32*(8+x)]: workgroups resident on XCC x (counted in phase 0)
      const unsigned int mine = __hip_atomic_load(&xc[32 * (8 + xcc)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned int old = __hip_atomic_fetch_add(&xc[32 * xcc], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (old + 1 == mine * phase) {  // last arriver of this XCC
        const unsigned int t = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned int nx = __hip_atomic_load(&xc[32 * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // XCCs in use
        if (t + 1 == nx * phase) {
          for (int x = 0; x < 8; ++x) __hip_atomic_store(&gen[32 * x], phase, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      while (__hip_atomic_load(&gen[32 * xcc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase) __builtin_amdgcn_s_sleep(2);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
  }
}

template <int V>
__global__ void k(unsigned int* c, unsigned int* xc, unsigned int* gen, int nb, double* sink) {
  if (V == 4) {  // census: workgroups per XCC (one flat barrier on the side counter c[64])
    if (threadIdx.x == 0) {
      unsigned int xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      xcc &= 7u;
      const unsigned int o = __hip_atomic_fetch_add(&xc[32 * (8 + xcc)], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (o == 0) __hip_atomic_fetch_add(&xc[32 * 16], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_fetch_add(&c[64], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      while (__hip_atomic_load(&c[64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < gridDim.x) __builtin_amdgcn_s_sleep(2);
    }
    __syncthreads();
  }
  double acc = 0;
  for (int b = 1; b <= nb; ++b) {
    barrier<V>(c, (unsigned)b * gridDim.x, xc, gen, (unsigned)b);
    acc += b;
  }
  if (acc < 0) sink[0] = acc;
}

template <int V>
int run(const char* name, int grid, int block, int nb) {
  unsigned int *c, *xc, *gen; double* sink;
  CK(hipMalloc(&c, 4096)); CK(hipMalloc(&xc, 4096 * 4)); CK(hipMalloc(&gen, 4096)); CK(hipMalloc(&sink, 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f, best0 = 1e30f;
  for (int rep = 0; rep < 6; ++rep) {
    for (int pass = 0; pass < 2; ++pass) {
      const int nbr = pass ? nb : 0;
      CK(hipMemset(c, 0, 4096)); CK(hipMemset(xc, 0, 4096 * 4)); CK(hipMemset(gen, 0, 4096));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k<V>, dim3(grid), dim3(block), 0, 0, c, xc, gen, nbr, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (pass) best = ms < best ? ms : best; else best0 = ms < best0 ? ms : best0;
    }
  }
  printf("%-34s grid %4d x %4d: %6.2f us per barrier (launch alone %.1f us)\n", name, grid, block, (best - best0) * 1e3f / nb, best0 * 1e3f);
  return 0;
}

int main() {
  const int nb = 200;
  for (int block : {256, 1024}) {
    run<0>("flat, fences, sleep 2", 256, block, nb);
    run<1>("flat, NO fences", 256, block, nb);
    run<2>("flat, fences, sleep 20", 256, block, nb);
    run<3>("last arriver publishes a flag", 256, block, nb);
    run<4>("per-XCC hierarchical", 256, block, nb);
  }
  for (int grid : {256, 128, 64}) {
    run<8>("NO fences, flat (shipped)", grid, 1024, nb);
    run<5>("NO fences, 8 split counters", grid, 1024, nb);
    run<6>("NO fences, 16 split counters", grid, 1024, nb);
    run<7>("NO fences, tree 16 + top", grid, 1024, nb);
  }
  run<0>("flat, fences, sleep 2", 64, 1024, nb);
  run<3>("last arriver publishes a flag", 64, 1024, nb);
  return 0;
}
