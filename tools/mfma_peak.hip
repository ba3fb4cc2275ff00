// Measures the sustained MFMA rate on this chip for the two instructions the Gram kernel
// uses (v_mfma_f64_16x16x4_f64, v_mfma_f32_16x16x4_f32): bare loops, operands in registers,
// 8 independent accumulators per wave, 1 or 2 waves per SIMD, every CU busy, random data.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC> __global__ void k64(const double* in, double* out, int iters) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = in[threadIdx.x + 64 * i];
  for (int i = 0; i < 2; ++i) b[i] = in[threadIdx.x + 64 * (4 + i)];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
        acc[(m * 2 + n) % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[m], b[n], acc[(m * 2 + n) % NACC], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC> __global__ void k32(const float* in, float* out, int iters) {
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = (f4){0, 0, 0, 0};
  float a[4], b[2];
  for (int i = 0; i < 4; ++i) a[i] = in[threadIdx.x + 64 * i];
  for (int i = 0; i < 2; ++i) b[i] = in[threadIdx.x + 64 * (4 + i)];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
        acc[(m * 2 + n) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], b[n], acc[(m * 2 + n) % NACC], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 4; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main() {
  const int iters = 20000;
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
      if (rep) printf("f64 16x16x4  %d waves/SIMD: %.3f ms  %.2f TFLOP/s\n", threads / 256, ms, fl / ms / 1e9);
      hipEventRecord(e0);
      hipLaunchKernelGGL(k32<8>, dim3(256), dim3(threads), 0, 0, fin, fout, iters);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("f32 16x16x4  %d waves/SIMD: %.3f ms  %.2f TFLOP/s\n", threads / 256, ms, fl / ms / 1e9);
    }
  }
  return 0;
}
