// Microbenchmark: what does it cost a wave to ISSUE LDS-DMA (global_load_lds_dwordx4) vs
// plain global_load_dwordx4, with 2 workgroups x 4 waves per CU all streaming rows?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

template <bool DMA, int PIECES>
__global__ __launch_bounds__(256, 2) void k(const double* X, long rows_per_wg, int K, int iters,
                                            unsigned long long* t_issue, unsigned long long* t_total, double* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long base = (long)blockIdx.x * rows_per_wg;
  unsigned long long ti = 0, tt = 0;
  d2 acc = {0, 0};
  for (int it = 0; it < iters; ++it) {
    unsigned long long a = __builtin_amdgcn_s_memtime();
    d2 v[PIECES];
#pragma unroll
    for (int p = 0; p < PIECES; ++p) {
      const long row = base + (long)(it * PIECES * 4 + p * 4 + wave) % rows_per_wg;
      const double* src = X + row * K + lane * 2;
      if (DMA) {
        char* dst = smem + ((it & 1) * PIECES * 4 + p * 4 + wave) * 1152;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
      } else {
        v[p] = *(const d2*)src;
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long b = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
    if (!DMA) {
#pragma unroll
      for (int p = 0; p < PIECES; ++p) *(d2*)(smem + ((it & 1) * PIECES * 4 + p * 4 + wave) * 1152 + lane * 16) = v[p];
    }
    __syncthreads();
    acc += *(d2*)(smem + ((it & 1) * PIECES * 4 + wave) * 1152 + lane * 16);
    unsigned long long c = __builtin_amdgcn_s_memtime();
    ti += b - a; tt += c - a;
  }
  if (lane == 0) { t_issue[blockIdx.x * 4 + wave] = ti; t_total[blockIdx.x * 4 + wave] = tt; }
  sink[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1];
}

template <bool DMA, int PIECES> int run(const double* X, long N, int K, const char* name) {
  const int wgs = 512, iters = 200;
  unsigned long long *ti, *tt; double* sink;
  CK(hipMalloc(&ti, wgs * 4 * 8)); CK(hipMalloc(&tt, wgs * 4 * 8)); CK(hipMalloc(&sink, wgs * 256 * 8));
  const size_t lds = 2 * PIECES * 4 * 1152;
  CK(hipFuncSetAttribute((const void*)k<DMA, PIECES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<DMA, PIECES>), dim3(wgs), dim3(256), lds, 0, X, N / wgs, K, iters, ti, tt, sink);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> hi(wgs * 4), ht(wgs * 4);
  CK(hipMemcpy(hi.data(), ti, wgs * 4 * 8, hipMemcpyDeviceToHost));
  CK(hipMemcpy(ht.data(), tt, wgs * 4 * 8, hipMemcpyDeviceToHost));
  double si = 0, st = 0; for (int i = 0; i < wgs * 4; ++i) { si += hi[i]; st += ht[i]; }
  const double bytes = (double)wgs * 4 * iters * PIECES * 1024.0;
  printf("%-28s pieces/wave %d: issue %.0f cyc/piece, %.0f cyc/iter total, %.3f ms, %.2f TB/s, %.1f GB/s/CU\n", name, PIECES,
         si / (wgs * 4) / iters / PIECES * 1.0, st / (wgs * 4) / iters, ms, bytes / ms / 1e9, bytes / ms / 1e6 / 256);
  return 0;
}
int main() {
  const int K = 512; const long N = 100000;
  double* X; CK(hipMalloc(&X, (size_t)N * K * 8)); CK(hipMemset(X, 0, (size_t)N * K * 8));
  run<true, 8>(X, N, K, "LDS-DMA, HBM stream");
  run<false, 8>(X, N, K, "reg-staged, HBM stream");
  run<true, 4>(X, N, K, "LDS-DMA, HBM stream");
  run<false, 4>(X, N, K, "reg-staged, HBM stream");
  run<true, 8>(X, 512 * 64, K, "LDS-DMA, L2/MALL resident");
  run<false, 8>(X, 512 * 64, K, "reg-staged, L2/MALL resident");
  return 0;
}
