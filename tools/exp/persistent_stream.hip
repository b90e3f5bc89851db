// Microbenchmark (round 3): does a PERSISTENT grid (256 workgroups x 1024 lanes, the shape of k_b2_coop's streaming passes)
// reach the rate of one-tile workgroups once its tiles are handed out dynamically?  3 reads + 1 write of fp64 (ShiftedNormL1Box
// arithmetic), LDS-DMA staged, 3 KiB per wavefront, vector and tile (a tile = 16 waves x 3 KiB = 48 KiB per vector).
//   static    tile = blockIdx + k * gridDim              (k_b2_coop today)
//   dynamic   tile = atomicAdd(counter, 1), the next index fetched while the current tile is processed
//   chunked   four tiles per atomicAdd
//   onetile   one 256-lane workgroup per 12 KiB tile, no loop (the separable skeleton's shape)
// build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/exp/persistent_stream tools/exp/persistent_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef double f64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__device__ __forceinline__ double op(double q, double x, double s) {
  const double xs = x + s, xsq = xs + q;
  double t = (xsq <= -1.0) ? (q + 1.0) : ((xsq >= 1.0) ? (q - 1.0) : -xs);
  return fmin(fmax(t, -1.0 - s), 1.0 - s);
}
constexpr int KP = 3;  // KiB per wavefront, vector and tile
typedef __attribute__((address_space(3))) void lds_void;
template <int WAVES>
__device__ __forceinline__ void do_tile(f64x2* y, const f64x2* q, const f64x2* x, const f64x2* s, long n2, long tile, char* wl) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long base = (tile * WAVES + wave) * 64 * KP + lane;
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    long i = base + k * 64; if (i >= n2) i = n2 - 1;
    __builtin_amdgcn_global_load_lds((const void*)(q + i), (lds_void*)(wl + (0 * KP + k) * 1024), 16, 0, 2);
    __builtin_amdgcn_global_load_lds((const void*)(x + i), (lds_void*)(wl + (1 * KP + k) * 1024), 16, 0, 2);
    __builtin_amdgcn_global_load_lds((const void*)(s + i), (lds_void*)(wl + (2 * KP + k) * 1024), 16, 0, 2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int k = 0; k < KP; ++k) {
    const long i = base + k * 64;
    const f64x2 a = *reinterpret_cast<const f64x2*>(wl + (0 * KP + k) * 1024 + lane * 16);
    const f64x2 b = *reinterpret_cast<const f64x2*>(wl + (1 * KP + k) * 1024 + lane * 16);
    const f64x2 c = *reinterpret_cast<const f64x2*>(wl + (2 * KP + k) * 1024 + lane * 16);
    if (i < n2) __builtin_nontemporal_store(f64x2{op(a.x, b.x, c.x), op(a.y, b.y, c.y)}, y + i);
  }
}
template <int MODE>  // 0 static, 1 dynamic, 2 chunked
__global__ __launch_bounds__(1024) void k_persistent(f64x2* y, const f64x2* q, const f64x2* x, const f64x2* s, long n2, long ntiles,
                                                      unsigned int* counter) {
  __shared__ __attribute__((aligned(16))) char dma[16 * 3 * KP * 1024];
  __shared__ unsigned int next_tile;
  char* wl = dma + (threadIdx.x >> 6) * (3 * KP * 1024);
  if (MODE == 0) {
    for (long t = blockIdx.x; t < ntiles; t += gridDim.x) do_tile<16>(y, q, x, s, n2, t, wl);
  } else {
    constexpr unsigned int CH = MODE == 2 ? 4u : 1u;
    if (threadIdx.x == 0) next_tile = atomicAdd(counter, CH);
    __syncthreads();
    unsigned int t0 = next_tile;
    while ((long)t0 < ntiles) {
      __syncthreads();  // everybody has read next_tile
      if (threadIdx.x == 0) next_tile = atomicAdd(counter, CH);  // (in flight while this chunk is processed)
      for (unsigned int c = 0; c < CH && (long)(t0 + c) < ntiles; ++c) do_tile<16>(y, q, x, s, n2, (long)(t0 + c), wl);
      __syncthreads();
      t0 = next_tile;
    }
  }
}
__global__ __launch_bounds__(256) void k_onetile(f64x2* y, const f64x2* q, const f64x2* x, const f64x2* s, long n2) {
  __shared__ __attribute__((aligned(16))) char dma[4 * 3 * KP * 1024];
  do_tile<4>(y, q, x, s, n2, (long)blockIdx.x, dma + (threadIdx.x >> 6) * (3 * KP * 1024));
}
int main() {
  const long n = 100000000, n2 = n / 2;
  f64x2 *y, *q, *x, *s; unsigned int* ctr;
  CK(hipMalloc(&y, n * 8)); CK(hipMalloc(&q, n * 8)); CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&s, n * 8)); CK(hipMalloc(&ctr, 4096));
  CK(hipMemset(q, 0, n * 8)); CK(hipMemset(x, 0, n * 8)); CK(hipMemset(s, 0, n * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const long tile16 = 16L * 64 * KP, tile4 = 4L * 64 * KP;
  const long nt16 = (n2 + tile16 - 1) / tile16, nt4 = (n2 + tile4 - 1) / tile4;
  for (int mode = 0; mode < 4; ++mode) {
    std::vector<float> ts;
    for (int rep = 0; rep < 12; ++rep) {
      CK(hipMemset(ctr, 0, 4096));
      CK(hipEventRecord(e0));
      if (mode == 0) hipLaunchKernelGGL(k_persistent<0>, dim3(256), dim3(1024), 0, 0, y, q, x, s, n2, nt16, ctr);
      else if (mode == 1) hipLaunchKernelGGL(k_persistent<1>, dim3(256), dim3(1024), 0, 0, y, q, x, s, n2, nt16, ctr);
      else if (mode == 2) hipLaunchKernelGGL(k_persistent<2>, dim3(256), dim3(1024), 0, 0, y, q, x, s, n2, nt16, ctr);
      else hipLaunchKernelGGL(k_onetile, dim3((unsigned)nt4), dim3(256), 0, 0, y, q, x, s, n2);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep >= 2) ts.push_back(ms);
    }
    std::sort(ts.begin(), ts.end());
    const char* names[4] = {"persistent, static tiles", "persistent, dynamic tiles", "persistent, 4 tiles per atomic", "one tile per 256-lane workgroup"};
    printf("%-34s median %.4f ms  min %.4f ms  -> %.0f GB/s\n", names[mode], ts[ts.size() / 2], ts[0], 32.0 * n / ts[ts.size() / 2] / 1e6);
  }
  return 0;
}
