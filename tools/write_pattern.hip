// Microbenchmark: HBM write throughput for the output patterns of the small-fold kernels.
//   hipcc --offload-arch=gfx950 -O3 -o tools/write_pattern tools/write_pattern.hip && tools/write_pattern
// A workgroup of 256 threads stores TR x TC doubles (16-byte stores, a row segment of TC*8 bytes is
// contiguous, rows are K*8 bytes apart) of a [F][K][K] array; all tiles of all F matrices are written.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double v2 __attribute__((ext_vector_type(2)));
template <int TR, int TC> __global__ __launch_bounds__(256) void wr(double *out, int K, int tiles_c, int tiles_r) {
  const int t = blockIdx.x, f = blockIdx.y;
  const int tr = t / tiles_c, tc = t - tr * tiles_c;
  double *o = out + (size_t)f * K * K + (size_t)tr * TR * K + (size_t)tc * TC;
  constexpr int LPR = TC / 2;
  for (int q = threadIdx.x; q < TR * LPR; q += 256) {
    const int r = q / LPR, c = (q - r * LPR) * 2;
    if (tr * TR + r < K && tc * TC + c < K) *reinterpret_cast<v2 *>(o + (size_t)r * K + c) = (v2){(double)q, (double)f};
  }
}
template <int TR, int TC> void run(const char *name, double *buf, int K, int F) {
  const int tiles_c = (K + TC - 1) / TC, tiles_r = (K + TR - 1) / TR;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((wr<TR, TC>), dim3(tiles_c * tiles_r, F), dim3(256), 0, 0, buf, K, tiles_c, tiles_r);
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((wr<TR, TC>), dim3(tiles_c * tiles_r, F), dim3(256), 0, 0, buf, K, tiles_c, tiles_r);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  printf("K=%d F=%d %-22s %8.3f ms  %7.0f GB/s\n", K, F, name, ms, (double)F * K * K * 8 / ms / 1e6);
}
int main() {
  double *buf; const size_t bytes = (size_t)2000 * 512 * 512 * 8;
  if (hipMalloc(&buf, bytes) != hipSuccess) return 1;
  for (int K : {500, 512}) {
    run<64, 64>("64 x 64 (512 B rows)", buf, K, 2000);
    run<32, 128>("32 x 128 (1 KB rows)", buf, K, 2000);
    run<16, 256>("16 x 256 (2 KB rows)", buf, K, 2000);
    run<8, 512>("8 x 512 (4 KB rows)", buf, K, 2000);
  }
  run<64, 64>("64 x 64 (512 B rows)", buf, 4096, 30);
  run<8, 512>("8 x 512 (4 KB rows)", buf, 4096, 30);
  // round 3: the float32 K = 4096 tile kernel writes 64 rows x 256 B, 16 KB apart (= 2048 doubles per row)
  run<64, 32>("64 x 256 B, rows 16 KB apart", buf, 2048, 120);
  run<64, 64>("64 x 512 B, rows 16 KB apart", buf, 2048, 120);
  run<32, 128>("32 x 1 KB, rows 16 KB apart", buf, 2048, 120);
  run<128, 32>("128 x 256 B, rows 16 KB apart", buf, 2048, 120);
  run<128, 64>("128 x 512 B, rows 16 KB apart", buf, 2048, 120);
  hipFree(buf);
  return 0;
}
