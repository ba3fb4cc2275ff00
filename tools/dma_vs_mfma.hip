// Microbenchmark: cost for a loader wave to issue LDS-DMA / SALU / VALU instructions while
// the compute wave of the same SIMD streams back-to-back v_mfma_f64_16x16x4_f64.
// 1 workgroup per CU: waves 0-3 MFMA (one per SIMD), waves 4-7 loaders.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>  // 0: DMA pieces (scalar asm), 1: VALU v_add_u32 chain, 2: SALU chain
__global__ __launch_bounds__(512, 2) void k(const double* X, int K, int iters, int mfma_on,
                                            unsigned long long* t_ld, unsigned long long* t_mf, double* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave < 4) {
    d4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (d4){0, 0, 0, 0};
    double a = X[lane], b = X[lane + 64];
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (mfma_on)
      for (int it = 0; it < iters * 8; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
      }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0; for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    sink[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) t_mf[blockIdx.x * 4 + wave] = t1 - t0;
    return;
  }
  const int d = wave - 4;
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char*)smem);
  const unsigned voff = lane * 16;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned vacc = lane; unsigned sacc = wave;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int p = 0; p < 12; ++p) {
        const long row = ((long)blockIdx.x * 97 + it * 48 + p * 4 + d) % 4096;
        const char* base = (const char*)(X + row * K);
        const unsigned dst = lds0 + ((it % 3) * 48 + p * 4 + d) * 1024;
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(base), "s"(dst) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    } else if (MODE == 1) {
#pragma unroll
      for (int p = 0; p < 12; ++p) asm volatile("v_add_u32 %0, %0, %1" : "+v"(vacc) : "v"(voff));
    } else {
#pragma unroll
      for (int p = 0; p < 12; ++p) asm volatile("s_add_u32 %0, %0, 7" : "+s"(sacc));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) t_ld[blockIdx.x * 4 + d] = t1 - t0;
  sink[blockIdx.x * 512 + threadIdx.x] = (double)(vacc + sacc);
}

template <int MODE> int run(const double* X, int K, int mfma_on, const char* name) {
  const int wgs = 256, iters = 200;
  unsigned long long *tl, *tm; double* sink;
  CK(hipMalloc(&tl, wgs * 4 * 8)); CK(hipMalloc(&tm, wgs * 4 * 8)); CK(hipMalloc(&sink, wgs * 512 * 8));
  const size_t lds = 3 * 48 * 1024;
  CK(hipFuncSetAttribute((const void*)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((k<MODE>), dim3(wgs), dim3(512), lds, 0, X, K, iters, mfma_on, tl, tm, sink);
    CK(hipDeviceSynchronize());
  }
  std::vector<unsigned long long> hl(wgs * 4), hm(wgs * 4);
  CK(hipMemcpy(hl.data(), tl, wgs * 32, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hm.data(), tm, wgs * 32, hipMemcpyDeviceToHost));
  double sl = 0, sm = 0; for (int i = 0; i < wgs * 4; ++i) { sl += hl[i]; sm += hm[i]; }
  printf("%-22s mfma %d: loader %6.0f cyc per 12 instr (%4.0f each); mfma wave %5.1f cyc per MFMA\n", name, mfma_on,
         sl / (wgs * 4) / iters, sl / (wgs * 4) / iters / 12, mfma_on ? sm / (wgs * 4) / (iters * 8.0 * 16) : 0.0);
  return 0;
}
int main() {
  const int K = 512;
  double* X; CK(hipMalloc(&X, (size_t)4096 * K * 8)); CK(hipMemset(X, 0, (size_t)4096 * K * 8));
  run<0>(X, K, 0, "LDS-DMA (L2 rows)");
  run<0>(X, K, 1, "LDS-DMA (L2 rows)");
  run<1>(X, K, 0, "VALU v_add_u32");
  run<1>(X, K, 1, "VALU v_add_u32");
  run<2>(X, K, 0, "SALU s_add_u32");
  run<2>(X, K, 1, "SALU s_add_u32");
  return 0;
}
