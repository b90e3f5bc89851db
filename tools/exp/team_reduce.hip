// Microbenchmark (round 4): time per grid_team_reduce<NV> (spx_group_common.hpp) inside one launch: the reduction primitive of
// spx_group_team.hip.  build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Ishiftedproximaloperators.jl_amd/csrc \
//   -o tools/exp/team_reduce tools/exp/team_reduce.hip ; run on the GPU box.
#include "spx_group_common.hpp"
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
void spx_set_error(const char*, ...) {}

template <int NV>
__global__ __launch_bounds__(1024) void k(unsigned long long* rows, SpxSyncHeader* hdr, int rounds, int W, double* sink) {
  __shared__ GridTeam gt;
  const int first = ((int)blockIdx.x / W) * W;
  if (threadIdx.x == 0) { gt.rows = rows; gt.hdr = hdr; gt.first = first; gt.W = W; }
  if (threadIdx.x < 16) { gt.wnp[threadIdx.x] = 0; gt.wcalls[threadIdx.x] = 0; }
  __syncthreads();
  double acc = 0.0;
  for (int r = 0; r < rounds; ++r) {
    double v[NV];
#pragma unroll
    for (int k2 = 0; k2 < NV; ++k2) v[k2] = 1.0 + k2 + acc * 1e-30;
    grid_team_reduce<NV>(&gt, v, NV >= 10 ? 0x300u : 0u);
    acc += v[0];
  }
  if (acc == 12345.0 && threadIdx.x == 0) sink[0] = acc;
}

template <int NV>
int run(int grid, int W, int rounds) {
  unsigned long long* rows; SpxSyncHeader* hdr; double* sink;
  CK(hipMalloc(&rows, kGtSetWords * 8)); CK(hipMalloc(&hdr, sizeof(SpxSyncHeader))); CK(hipMalloc(&sink, 8));
  CK(hipMemset(hdr, 0, sizeof(SpxSyncHeader)));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e30f, best0 = 1e30f;
  for (int rep = 0; rep < 6; ++rep)
    for (int pass = 0; pass < 2; ++pass) {
      CK(hipMemset(rows, 0, kGtSetWords * 8));
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k<NV>, dim3(grid), dim3(1024), 0, 0, rows, hdr, pass ? rounds : 0, W, sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (pass) best = ms < best ? ms : best; else best0 = ms < best0 ? ms : best0;
    }
  printf("NV %2d  grid %3d  team %3d : %.2f us per reduction (empty launch %.1f us)\n", NV, grid, W, (best - best0) * 1e3 / rounds, best0 * 1e3);
  CK(hipFree(rows)); CK(hipFree(hdr)); CK(hipFree(sink));
  return 0;
}

int main() {
  const int rounds = 200;
  for (int W : {1, 2, 36, 256}) {
    const int grid = W == 36 ? 252 : (W == 1 ? 256 : (W == 2 ? 256 : 256));
    if (run<2>(grid, W, rounds)) return 1;
    if (run<6>(grid, W, rounds)) return 1;
    if (run<10>(grid, W, rounds)) return 1;
  }
  return 0;
}
