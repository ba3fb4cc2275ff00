// Which linear fill reaches the write ceiling of the box?  hipcc --offload-arch=gfx950 -O3 -o tools/fill_variants tools/fill_variants.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float v4 __attribute__((ext_vector_type(4)));
template <bool NT> __device__ __forceinline__ void st(v4 *p, v4 v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }
template <bool NT> __global__ __launch_bounds__(256) void stride_fill(v4 *p, size_t n) {
  const v4 z = {0, 0, 0, 0};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) st<NT>(p + i, z);
}
template <bool NT, int IT> __global__ __launch_bounds__(256) void chunk_fill(v4 *p, size_t n) {
  const v4 z = {0, 0, 0, 0};
  size_t i = (size_t)blockIdx.x * 256 * IT + threadIdx.x;
#pragma unroll
  for (int k = 0; k < IT; ++k, i += 256) if (i < n) st<NT>(p + i, z);
}
template <typename F> void timeit(const char *name, size_t bytes, F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 20; ++i) f();
  hipEventRecord(a);
  for (int i = 0; i < 10; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 10;
  printf("%-34s %7.3f ms  %6.2f TB/s\n", name, ms, bytes / ms / 1e9);
}
int main() {
  for (size_t bytes : {(size_t)3221225472ull, (size_t)6442450944ull}) {
    v4 *p; if (hipMalloc(&p, bytes) != hipSuccess) return 1;
    const size_t n = bytes / 16;
    printf("%zu MiB\n", bytes >> 20);
    for (int g : {1024, 2048, 4096, 16384}) {
      char nm[64]; snprintf(nm, 64, "stride nt grid %d", g);
      timeit(nm, bytes, [&] { hipLaunchKernelGGL(stride_fill<true>, dim3(g), dim3(256), 0, 0, p, n); });
      snprintf(nm, 64, "stride plain grid %d", g);
      timeit(nm, bytes, [&] { hipLaunchKernelGGL(stride_fill<false>, dim3(g), dim3(256), 0, 0, p, n); });
    }
    timeit("chunk 4 KiB nt", bytes, [&] { hipLaunchKernelGGL((chunk_fill<true, 1>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, p, n); });
    timeit("chunk 16 KiB nt", bytes, [&] { hipLaunchKernelGGL((chunk_fill<true, 4>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, p, n); });
    timeit("chunk 64 KiB nt", bytes, [&] { hipLaunchKernelGGL((chunk_fill<true, 16>), dim3((unsigned)((n + 4095) / 4096)), dim3(256), 0, 0, p, n); });
    timeit("chunk 16 KiB plain", bytes, [&] { hipLaunchKernelGGL((chunk_fill<false, 4>), dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, 0, p, n); });
    timeit("chunk 64 KiB plain", bytes, [&] { hipLaunchKernelGGL((chunk_fill<false, 16>), dim3((unsigned)((n + 4095) / 4096)), dim3(256), 0, 0, p, n); });
    hipFree(p);
  }
  return 0;
}
