// Microbenchmark: variants of the 3-read/1-write fp64 streaming skeleton (ShiftedNormL1Box arithmetic) on MI355X.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/exp/exp_stream.hip -o tools/exp/exp_stream
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <functional>
#include <string>
typedef double f64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ double jl_min(double x, double y) { double d = x - y; double a = (__double_as_longlong(d) < 0) ? x : y; return (x != x || y != y) ? d : a; }
__device__ __forceinline__ double jl_max(double x, double y) { double d = x - y; double a = (__double_as_longlong(d) < 0) ? y : x; return (x != x || y != y) ? d : a; }
__device__ __forceinline__ double op(double q, double x, double s, double sl, double l, double u) {
  double xs = x + s, xsq = xs + q;
  double t = (xsq <= -sl) ? (q + sl) : ((xsq >= sl) ? (q - sl) : -xs);
  return jl_min(jl_max(t, l - s), u - s);
}
__device__ __forceinline__ f64x2 op2(f64x2 a, f64x2 b, f64x2 c) { f64x2 r; r.x = op(a.x, b.x, c.x, 1.0, -1.0, 1.0); r.y = op(a.y, b.y, c.y, 1.0, -1.0, 1.0); return r; }

template <int THREADS, int UNROLL, int NTL, int NTS>
__global__ __launch_bounds__(THREADS) void k_reg(f64x2* y, const f64x2* q, const f64x2* x, const f64x2* s, long n2) {
  const long base = (long)blockIdx.x * THREADS * UNROLL + threadIdx.x;
  f64x2 a[UNROLL], b[UNROLL], c[UNROLL];
#pragma unroll
  for (int k = 0; k < UNROLL; ++k) {
    long i = base + k * THREADS; if (i >= n2) i = n2 - 1;
    if (NTL) { a[k] = __builtin_nontemporal_load(q + i); b[k] = __builtin_nontemporal_load(x + i); c[k] = __builtin_nontemporal_load(s + i); }
    else { a[k] = q[i]; b[k] = x[i]; c[k] = s[i]; }
  }
#pragma unroll
  for (int k = 0; k < UNROLL; ++k) {
    long i = base + k * THREADS;
    if (i < n2) { f64x2 r = op2(a[k], b[k], c[k]); if (NTS) __builtin_nontemporal_store(r, y + i); else y[i] = r; }
  }
}

// LDS-DMA variant: each wave stages its q/x/s tile through LDS with global_load_lds_dwordx4 (no VGPR destination),
// then reads it back with ds_read_b128, computes, stores.  One tile = UNROLL KiB per array per wave.
template <int UNROLL, int AUX>
__global__ __launch_bounds__(256) void k_lds(f64x2* y, const f64x2* q, const f64x2* x, const f64x2* s, long n2) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  char* wl = lds + wave * (3 * UNROLL * 1024);
  const long base = ((long)blockIdx.x * 4 + wave) * 64 * UNROLL;  // in pairs
#pragma unroll
  for (int k = 0; k < UNROLL; ++k) {
    long i = base + k * 64 + lane; if (i >= n2) i = n2 - 1;
    __builtin_amdgcn_global_load_lds((const void*)(q + i), (__attribute__((address_space(3))) void*)(wl + (0 * UNROLL + k) * 1024), 16, 0, AUX);
    __builtin_amdgcn_global_load_lds((const void*)(x + i), (__attribute__((address_space(3))) void*)(wl + (1 * UNROLL + k) * 1024), 16, 0, AUX);
    __builtin_amdgcn_global_load_lds((const void*)(s + i), (__attribute__((address_space(3))) void*)(wl + (2 * UNROLL + k) * 1024), 16, 0, AUX);
  }
  __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) lgkmcnt(0): all DMA pieces landed
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
  for (int k = 0; k < UNROLL; ++k) {
    long i = base + k * 64 + lane;
    f64x2 a = *reinterpret_cast<f64x2*>(wl + (0 * UNROLL + k) * 1024 + lane * 16);
    f64x2 b = *reinterpret_cast<f64x2*>(wl + (1 * UNROLL + k) * 1024 + lane * 16);
    f64x2 c = *reinterpret_cast<f64x2*>(wl + (2 * UNROLL + k) * 1024 + lane * 16);
    if (i < n2) __builtin_nontemporal_store(op2(a, b, c), y + i);
  }
}

// pipelined: piece k (q, x, s) is consumed as soon as its three DMA loads have landed; younger loads and the
// stores already issued stay in flight (vmcnt counts loads, LDS-DMA and stores in issue order)
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int UNROLL, int K>
struct Consume {
  static __device__ __forceinline__ void run(char* wl, f64x2* y, long base, int lane, long n2) {
    wait_vmcnt<3 * (UNROLL - 1 - K) + K>();
    long i = base + K * 64 + lane;
    f64x2 a = *reinterpret_cast<f64x2*>(wl + (3 * K + 0) * 1024 + lane * 16);
    f64x2 b = *reinterpret_cast<f64x2*>(wl + (3 * K + 1) * 1024 + lane * 16);
    f64x2 c = *reinterpret_cast<f64x2*>(wl + (3 * K + 2) * 1024 + lane * 16);
    f64x2 r = op2(a, b, c);
    if (i < n2) __builtin_nontemporal_store(r, y + i);
    else asm volatile("s_nop 0");  // keep the store count uniform only when in range: tail tiles use vmcnt(0) path
    if constexpr (K + 1 < UNROLL) Consume<UNROLL, K + 1>::run(wl, y, base, lane, n2);
  }
};
template <int THREADS, int UNROLL>
__global__ __launch_bounds__(THREADS) void k_ldsp(f64x2* y, const f64x2* q, const f64x2* x, const f64x2* s, long n2) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  char* wl = lds + wave * (3 * UNROLL * 1024);
  const long base = ((long)blockIdx.x * (THREADS / 64) + wave) * 64 * UNROLL;  // in pairs
#pragma unroll
  for (int k = 0; k < UNROLL; ++k) {
    long i = base + k * 64 + lane; if (i >= n2) i = n2 - 1;
    __builtin_amdgcn_global_load_lds((const void*)(q + i), (__attribute__((address_space(3))) void*)(wl + (3 * k + 0) * 1024), 16, 0, 2);
    __builtin_amdgcn_global_load_lds((const void*)(x + i), (__attribute__((address_space(3))) void*)(wl + (3 * k + 1) * 1024), 16, 0, 2);
    __builtin_amdgcn_global_load_lds((const void*)(s + i), (__attribute__((address_space(3))) void*)(wl + (3 * k + 2) * 1024), 16, 0, 2);
  }
  if (base + 64 * UNROLL <= n2) {
    Consume<UNROLL, 0>::run(wl, y, base, lane, n2);
  } else {
    wait_vmcnt<0>();
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) {
      long i = base + k * 64 + lane;
      f64x2 a = *reinterpret_cast<f64x2*>(wl + (3 * k + 0) * 1024 + lane * 16);
      f64x2 b = *reinterpret_cast<f64x2*>(wl + (3 * k + 1) * 1024 + lane * 16);
      f64x2 c = *reinterpret_cast<f64x2*>(wl + (3 * k + 2) * 1024 + lane * 16);
      if (i < n2) __builtin_nontemporal_store(op2(a, b, c), y + i);
    }
  }
}

__global__ void k_copy(f64x2* y, const f64x2* q, long n2) {  // reference: 1 read + 1 write
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n2) __builtin_nontemporal_store(__builtin_nontemporal_load(q + i), y + i);
}
__global__ void k_fill(double* p, long n, unsigned long long seed) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { unsigned long long z = (i + seed) * 0x9E3779B97F4A7C15ull; z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29;
    p[i] = ((double)(z >> 11) * (1.0 / 9007199254740992.0) - 0.5) * 4.0; }
}

int main(int argc, char** argv) {
  long n = argc > 1 ? atol(argv[1]) : 100000000L;
  long n2 = n / 2;
  double *q, *x, *s, *y;
  long skew = argc > 2 ? atol(argv[2]) : 0;  // bytes of extra offset between consecutive arrays
  char* big; CK(hipMalloc(&big, 4 * (n * 8 + (1 << 21)) + 4 * skew + (1 << 21)));
  size_t stride = ((size_t)n * 8 + (1 << 21) - 1) / (1 << 21) * (1 << 21);
  q = (double*)(big); x = (double*)(big + stride + skew); s = (double*)(big + 2 * (stride + skew)); y = (double*)(big + 3 * (stride + skew));
  printf("n=%ld skew=%ld\n", n, skew);
  k_fill<<<(n + 255) / 256, 256>>>(q, n, 1); k_fill<<<(n + 255) / 256, 256>>>(x, n, 2); k_fill<<<(n + 255) / 256, 256>>>(s, n, 3);
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  struct V { std::string name; std::function<void()> run; double bytes; std::vector<float> t; };
  std::vector<V> vs;
  auto Q = (const f64x2*)q; auto X = (const f64x2*)x; auto S = (const f64x2*)s; auto Y = (f64x2*)y;
#define REG(T, U, L, W) vs.push_back({"reg thr=" #T " unroll=" #U " ntl=" #L " nts=" #W, [=]() { hipLaunchKernelGGL((k_reg<T, U, L, W>), dim3((unsigned)((n2 + (long)T * U - 1) / ((long)T * U))), dim3(T), 0, 0, Y, Q, X, S, n2); }, 32.0 * n, {}})
  REG(256, 4, 1, 1);
#define LDS(U, A) vs.push_back({"ldsdma unroll=" #U " aux=" #A, [=]() { hipLaunchKernelGGL((k_lds<U, A>), dim3((unsigned)((n2 + 256L * U - 1) / (256L * U))), dim3(256), 4 * 3 * U * 1024, 0, Y, Q, X, S, n2); }, 32.0 * n, {}})
  LDS(5, 2); LDS(6, 2);
#define LDSP(T, U) vs.push_back({"ldsdma-pipe thr=" #T " unroll=" #U, [=]() { hipLaunchKernelGGL((k_ldsp<T, U>), dim3((unsigned)((n2 + (long)T * U - 1) / ((long)T * U))), dim3(T), (T / 64) * 3 * U * 1024, 0, Y, Q, X, S, n2); }, 32.0 * n, {}})
  LDSP(256, 4); LDSP(256, 5); LDSP(256, 6); LDSP(128, 6); LDSP(128, 8); LDSP(128, 12); LDSP(512, 5); LDSP(64, 12); LDSP(64, 16);
  vs.push_back({"copy 1r1w nt", [=]() { k_copy<<<(unsigned)((n2 + 255) / 256), 256>>>(Y, Q, n2); }, 16.0 * n, {}});
  {  // validate every 32 B/element variant against the register baseline
    std::vector<double> ref(n), got(n);
    vs[0].run(); CK(hipDeviceSynchronize()); CK(hipMemcpy(ref.data(), y, n * 8, hipMemcpyDeviceToHost));
    for (size_t k = 1; k < vs.size(); ++k) {
      if (vs[k].bytes != 32.0 * n) continue;
      CK(hipMemset(y, 0xff, n * 8)); vs[k].run(); CK(hipDeviceSynchronize());
      CK(hipMemcpy(got.data(), y, n * 8, hipMemcpyDeviceToHost));
      long bad = 0; for (long i = 0; i < n; ++i) bad += (memcmp(&ref[i], &got[i], 8) != 0);
      printf("check %-36s mismatches %ld\n", vs[k].name.c_str(), bad);
    }
  }
  const int rounds = 7, iters = 10;
  for (int r = 0; r < rounds; ++r)
    for (auto& v : vs) {
      if (r == 0) { v.run(); CK(hipDeviceSynchronize()); }
      CK(hipEventRecord(e0));
      for (int i = 0; i < iters; ++i) v.run();
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); v.t.push_back(ms / iters);
    }
  CK(hipGetLastError());
  for (auto& v : vs) { std::sort(v.t.begin(), v.t.end()); float med = v.t[v.t.size() / 2];
    printf("%-40s median %.4f ms  min %.4f ms  -> %.0f GB/s\n", v.name.c_str(), med, v.t[0], v.bytes / med / 1e6); }
  return 0;
}
