// Measures the sustained MFMA rate on this chip for the two instructions the Gram kernel
// uses (v_mfma_f64_16x16x4_f64, v_mfma_f32_16x16x4_f32): bare loops, operands in registers,
// 8 independent accumulators per wave, 1 or 2 waves per SIMD, every CU busy, random data.
// Round 6: every workgroup also stamps {s_memtime, s_memrealtime} around its loop, so that the run prints the
// shader clock the chip holds under the bare loop and the CYCLES per MFMA (= clock x time / MFMAs per SIMD):
// the calibration of the product kernel's clock probe (cvm_clock_probe) -- 64 cycles per v_mfma_f64_16x16x4_f64
// and 32 per v_mfma_f32_16x16x4_f32 say that an s_memtime tick is a shader cycle.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ unsigned long long g_clk[1024 * 4];
#define CLK0() unsigned long long c0_ = 0, q0_ = 0; if (threadIdx.x == 0) { c0_ = __builtin_amdgcn_s_memtime(); q0_ = __builtin_amdgcn_s_memrealtime(); }
#define CLK1() if (threadIdx.x == 0) { unsigned long long c1_ = __builtin_amdgcn_s_memtime(), q1_ = __builtin_amdgcn_s_memrealtime(); \
    g_clk[4 * blockIdx.x] = c0_; g_clk[4 * blockIdx.x + 1] = q0_; g_clk[4 * blockIdx.x + 2] = c1_; g_clk[4 * blockIdx.x + 3] = q1_; }
template <int NACC> __global__ void k64(const double* in, double* out, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = in[threadIdx.x + 64 * i];
  for (int i = 0; i < 2; ++i) b[i] = in[threadIdx.x + 64 * (4 + i)];
  CLK0();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
        acc[(m * 2 + n) % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[(m * 2 + n) % NACC], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
  asm volatile("" :: "v"(s));
  CLK1();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> __global__ void k32(const float* in, float* out, int iters) {
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f4){0, 0, 0, 0};
  float a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = in[threadIdx.x + 64 * i];
  for (int i = 0; i < 2; ++i) b[i] = in[threadIdx.x + 64 * (4 + i)];
  CLK0();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
        acc[(m * 2 + n) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[n], acc[(m * 2 + n) % NACC], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
  asm volatile("" :: "v"(s));
  CLK1();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
static void clock_report(int threads, int iters, const char *what) {
  static unsigned long long h[1024 * 4];
  hipMemcpyFromSymbol(h, HIP_SYMBOL(g_clk), sizeof(h));
  double cyc = 0, tk = 0;
  for (int b = 0; b < 256; ++b) { cyc += (double)(h[4 * b + 2] - h[4 * b]); tk += (double)(h[4 * b + 3] - h[4 * b + 1]); }
  const double mhz = cyc / tk * 100.0;
  // MFMAs issued per SIMD by the waves of one workgroup: waves per SIMD x iters x 8
  const double per_simd = (threads / 256) * (double)iters * 8.0;
  printf("    %s: shader clock %.0f MHz (s_memtime / s_memrealtime), %.2f cycles per MFMA and SIMD\n", what, mhz, cyc / 256.0 / per_simd);
}
int main(int argc, char **argv) {
  // (long enough for the clock to settle: argv[1] = iterations, default 20000 = ~4 ms per launch; the clock
  //  report of a run with 2000000 iterations is the one to read)
  const int iters = argc > 1 ? atoi(argv[1]) : 20000;
  double* din; double* dout; float* fin; float* fout;
  hipMalloc(&din, 1024 * 8 * 8); hipMalloc(&dout, 256 * 1024 * 8);
  hipMalloc(&fin, 1024 * 8 * 4); hipMalloc(&fout, 256 * 1024 * 4);
  double h[8192]; float hf[8192];
  for (int i = 0; i < 8192; ++i) { h[i] = (rand() / (double)RAND_MAX) - 0.5; hf[i] = (float)h[i]; }
  hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  hipMemcpy(fin, hf, sizeof(hf), hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int threads : {256, 512}) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(k64<8>, dim3(256), dim3(threads), 0, 0, din, dout, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      double fl = 256.0 * (threads / 64) * iters * 8.0 * 2048.0;
      if (rep) { printf("f64 16x16x4  %d waves/SIMD: %.3f ms  %.2f TFLOP/s\n", threads / 256, ms, fl / ms / 1e9); clock_report(threads, iters, "f64"); }
      hipEventRecord(e0);
      hipLaunchKernelGGL(k32<8>, dim3(256), dim3(threads), 0, 0, fin, fout, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      if (rep) { printf("f32 16x16x4  %d waves/SIMD: %.3f ms  %.2f TFLOP/s\n", threads / 256, ms, fl / ms / 1e9); clock_report(threads, iters, "f32"); }
    }
  }
  return 0;
}
