// Microbenchmark: run-ahead LDS-DMA loader waves (like wgram4's waves 4-6): NW loader waves
// per workgroup (1 workgroup per CU), each issues P x 1 KiB pieces per step and keeps D
// steps in flight behind a counted vmcnt.  Reports issue cycles per piece and GB/s per CU.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int P, int D>
__global__ __launch_bounds__(512, 2) void k(const double* X, long rows_per_wg, int K, int iters, int nw,
                                            unsigned long long* t_issue, unsigned long long* t_wait) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave >= nw) return;
  const long base = (long)blockIdx.x * rows_per_wg;
  unsigned long long ti = 0, tw = 0;
  for (int it = 0; it < iters; ++it) {
    unsigned long long a = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int p = 0; p < P; ++p) {
      const long row = base + (long)((it * P + p) * nw + wave) % rows_per_wg;
      const double* src = X + row * K + lane * 2;
      char* dst = smem + (((it % (D + 1)) * P + p) * 8 + wave) * 1152;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    unsigned long long b = __builtin_amdgcn_s_memtime();
    if (D == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(P * D) : "memory");
    unsigned long long c = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ti += b - a; tw += c - b;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) { t_issue[blockIdx.x * 8 + wave] = ti; t_wait[blockIdx.x * 8 + wave] = tw; }
}

template <int P, int D> int run(const double* X, long N, int K, int nw, const char* name) {
  const int wgs = 256, iters = 400;
  unsigned long long *ti, *tw;
  CK(hipMalloc(&ti, wgs * 8 * 8)); CK(hipMalloc(&tw, wgs * 8 * 8));
  CK(hipMemset(ti, 0, wgs * 64)); CK(hipMemset(tw, 0, wgs * 64));
  const size_t lds = (size_t)(D + 1) * P * 8 * 1152;
  CK(hipFuncSetAttribute((const void*)k<P, D>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<P, D>), dim3(wgs), dim3(512), lds, 0, X, N / wgs, K, iters, nw, ti, tw);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> hi(wgs * 8), hw(wgs * 8);
  CK(hipMemcpy(hi.data(), ti, wgs * 64, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hw.data(), tw, wgs * 64, hipMemcpyDeviceToHost));
  double si = 0, sw = 0; for (int i = 0; i < wgs * 8; ++i) { si += hi[i]; sw += hw[i]; }
  const double bytes = (double)wgs * nw * iters * P * 1024.0;
  printf("%-14s waves %d P %2d depth %d: issue %4.0f cyc/piece, wait %5.0f cyc/step, %.1f GB/s/CU (%.2f TB/s)\n", name, nw, P, D,
         si / (wgs * nw) / iters / P, sw / (wgs * nw) / iters, bytes / ms / 1e6 / 256, bytes / ms / 1e9);
  return 0;
}
int main() {
  const int K = 512; const long N = 100000;
  double* X; CK(hipMalloc(&X, (size_t)N * K * 8)); CK(hipMemset(X, 0, (size_t)N * K * 8));
  run<11, 2>(X, N, K, 3, "HBM stream");
  run<11, 0>(X, N, K, 3, "HBM stream");
  run<4, 2>(X, N, K, 8, "HBM stream");
  run<8, 1>(X, N, K, 4, "HBM stream");
  run<11, 2>(X, 256 * 64, K, 3, "L2-resident");
  run<4, 2>(X, 256 * 64, K, 8, "L2-resident");
  run<11, 2>(X, N, K, 1, "HBM stream");
  return 0;
}
