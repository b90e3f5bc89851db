// Does a captured hipMemsetAsync / hipMemset2DAsync node do what the eager call does?  (round 2, graph-safe mode of libspx)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/exp/graph_memset tools/exp/graph_memset.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void bump(unsigned long long* buf, unsigned long long* seen, int slot) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { seen[slot] = buf[0]; buf[0] += 5; }
}
__global__ void probe2d(const unsigned long long* rows, unsigned long long* seen) {  // sums what a 2D memset should have zeroed / kept
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    unsigned long long in = 0, out = 0;
    for (int r = 0; r < 64; ++r) for (int c = 0; c < 2048; ++c) { if (c < 24) in += rows[r * 2048 + c]; else out += rows[r * 2048 + c]; }
    seen[0] = in; seen[1] = out;
  }
}
__global__ void fill(unsigned long long* p, size_t n, unsigned long long v) {
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
int main() {
  hipStream_t st; CK(hipStreamCreate(&st));
  unsigned long long *buf, *seen, *rows, h[8];
  CK(hipMalloc(&buf, 4096)); CK(hipMalloc(&seen, 4096)); CK(hipMalloc(&rows, 64 * 2048 * 8));
  CK(hipMemset(buf, 0xff, 4096)); CK(hipMemset(seen, 0, 4096));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  CK(hipMemsetAsync(buf, 0, 8, st));
  hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, st, buf, seen, 0);
  hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, st, buf, seen, 1);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h, seen, 16, hipMemcpyDeviceToHost));
    printf("1D memset of 8 bytes, replay %d: kernel 1 saw %llu (want 0), kernel 2 saw %llu (want 5)\n", rep, h[0], h[1]);
  }
  // 2D: 64 rows of 2048 words, zero the first 24 words of every row
  hipGraph_t g2; hipGraphExec_t ge2;
  hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, st, rows, (size_t)64 * 2048, 1ull);
  CK(hipStreamSynchronize(st));
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  CK(hipMemset2DAsync(rows, 2048 * 8, 0, 24 * 8, 64, st));
  hipLaunchKernelGGL(probe2d, dim3(1), dim3(64), 0, st, rows, seen);
  CK(hipStreamEndCapture(st, &g2));
  CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, st, rows, (size_t)64 * 2048, 1ull);
    CK(hipGraphLaunch(ge2, st)); CK(hipStreamSynchronize(st));
    CK(hipMemcpy(h, seen, 16, hipMemcpyDeviceToHost));
    printf("2D memset, replay %d: inside the window %llu (want 0), outside %llu (want %d)\n", rep, h[0], h[1], 64 * (2048 - 24));
  }
  // eager 2D for comparison
  hipLaunchKernelGGL(fill, dim3(64), dim3(256), 0, st, rows, (size_t)64 * 2048, 1ull);
  CK(hipMemset2DAsync(rows, 2048 * 8, 0, 24 * 8, 64, st));
  hipLaunchKernelGGL(probe2d, dim3(1), dim3(64), 0, st, rows, seen);
  CK(hipStreamSynchronize(st)); CK(hipMemcpy(h, seen, 16, hipMemcpyDeviceToHost));
  printf("2D memset, eager: inside %llu (want 0), outside %llu (want %d)\n", h[0], h[1], 64 * (2048 - 24));
  return 0;
}
