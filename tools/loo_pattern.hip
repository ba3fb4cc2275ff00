// Pure-store ceilings of the leave-one-out output pattern (K = 500 / 512 float64, 2000 matrices of K x K, whole rows by
// nontemporal 16-byte stores, 8-row panels like small_rows_kernel).  hipcc --offload-arch=gfx950 -O3 -o tools/loo_pattern tools/loo_pattern.hip
//   mode 0  one workgroup per (fold, panel), fold-major
//   mode 1  small_rows_kernel's walk: a workgroup = (group of FPW consecutive folds, panel), XCD-contiguous item ranges
//   mode 2  persistent: panels x G workgroups, workgroup (g, panel) walks the folds g, g + G, g + 2 G, ... (all workgroups of the
//           launch resident: the stores in flight span G matrices and move through memory in order)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v2 __attribute__((ext_vector_type(2)));
__device__ int g_map = 0;
// map 0: thread t stores piece t of every row (the kernel's mapping: a wave's 1 KiB store starts where the row starts, mod 128 B)
// map 1: the waves' boundaries inside a row are moved to 128-byte lines (row pitch 4000 B: the row starts 32 (row % 4) bytes into
//        a line): wave 0 takes the pieces up to the first line boundary + 56, waves 1-3 whole runs of 64 from there
// map 2: the panel as ONE flat run of ROWS x 250 pieces: piece t + 256 p, every wave store 1 KiB on a 1 KiB boundary
template <int ROWS> __device__ __forceinline__ void store_panel(double *m, int K, int panel, int tid, double tag, int map) {
  const int ppr = K / 2;                                 // pieces per row
  if (map == 2) {
    v2 *base = reinterpret_cast<v2 *>(m + (size_t)panel * ROWS * K);
    const int total = (panel * ROWS + ROWS <= K ? ROWS : K - panel * ROWS) * ppr;
#pragma unroll
    for (int p = 0; p < ROWS; ++p) {
      const int q = tid + 256 * p;
      if (q < total) __builtin_nontemporal_store((v2){tag, (double)p}, base + q);
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int row = panel * ROWS + r;
    if (row >= K) continue;
    int piece = tid;
    if (map == 1) {
      const int off16 = (int)(((size_t)row * K * 8 / 16) & 7);      // pieces into a 128-byte line at the row's start
      const int h = (8 - off16) & 7;                                  // pieces up to the first line boundary
      const int w = tid >> 6, lane = tid & 63;
      if (w == 0) piece = lane < h + 56 ? lane : -1;
      else piece = h + 56 + 64 * (w - 1) + lane;
    }
    if (piece >= 0 && piece < ppr) __builtin_nontemporal_store((v2){tag, (double)r}, reinterpret_cast<v2 *>(m + (size_t)row * K + 2 * piece));
  }
}
template <int ROWS> __global__ __launch_bounds__(256) void loo(double *out, int K, int F, int mode, int fpw, int G, int map) {
  const int panels = (K + ROWS - 1) / ROWS, tid = threadIdx.x;
  const size_t mat = (size_t)K * K;
  if (mode == 0) {
    const int f = blockIdx.x / panels, p = blockIdx.x % panels;
    store_panel<ROWS>(out + f * mat, K, p, tid, (double)f, map);
  } else if (mode == 1) {
    const unsigned groups = (F + fpw - 1) / fpw, tot = groups * panels, per = (tot + 7) / 8;
    const unsigned item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (item >= tot) return;
    const int p = item % panels, g = item / panels;
    for (int i = 0; i < fpw; ++i) {
      const int f = g * fpw + i;
      if (f < F) store_panel<ROWS>(out + f * mat, K, p, tid, (double)f, map);
    }
  } else {
    const unsigned tot = (unsigned)G * panels, per = (tot + 7) / 8;
    const unsigned item = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (item >= tot) return;
    const int p = item % panels, g = item / panels;
    for (int f = g; f < F; f += G) store_panel<ROWS>(out + f * mat, K, p, tid, (double)f, map);
  }
}
template <typename F> float timeit(F f) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) f();
  hipEventRecord(a);
  for (int i = 0; i < 5; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 5;
}
int main() {
  const int F = 2000;
  double *buf; if (hipMalloc(&buf, (size_t)F * 512 * 512 * 8) != hipSuccess) return 1;
  for (int K : {500, 512}) {
    const int panels = (K + 7) / 8;
    const double by = (double)F * K * K * 8;
    float ms = timeit([&] { hipLaunchKernelGGL(loo<8>, dim3(F * panels), dim3(256), 0, 0, buf, K, F, 0, 1, 1, 0); });
    printf("K=%d one workgroup per (fold, panel), fold-major:        %6.3f ms %5.2f TB/s\n", K, ms, by / ms / 1e9);
    for (int fpw : {2, 4, 8, 16}) {
      const unsigned tot = ((F + fpw - 1) / fpw) * panels;
      ms = timeit([&] { hipLaunchKernelGGL(loo<8>, dim3(8 * ((tot + 7) / 8)), dim3(256), 0, 0, buf, K, F, 1, fpw, 1, 0); });
      printf("K=%d (group of %2d consecutive folds, panel):             %6.3f ms %5.2f TB/s\n", K, fpw, ms, by / ms / 1e9);
    }
    for (int map : {0, 1, 2})
      for (int fpw : {1, 2, 8}) {
        const unsigned tot = ((F + fpw - 1) / fpw) * panels;
        ms = timeit([&] { hipLaunchKernelGGL(loo<8>, dim3(8 * ((tot + 7) / 8)), dim3(256), 0, 0, buf, K, F, 1, fpw, 1, map); });
        printf("K=%d lane -> piece map %d, groups of %d consecutive folds:        %6.3f ms %5.2f TB/s\n", K, map, fpw, ms, by / ms / 1e9);
      }
    for (int G : {8, 16, 32, 48, 64, 128}) {
      const unsigned tot = (unsigned)G * panels;
      ms = timeit([&] { hipLaunchKernelGGL(loo<8>, dim3(8 * ((tot + 7) / 8)), dim3(256), 0, 0, buf, K, F, 2, 1, G, 0); });
      printf("K=%d persistent, %3d workgroup sets x %d panels (%5u wgs):  %6.3f ms %5.2f TB/s\n", K, G, panels, tot, ms, by / ms / 1e9);
    }
    for (int G : {16, 32, 64}) {
      const int p4 = (K + 3) / 4; const unsigned tot = (unsigned)G * p4;
      ms = timeit([&] { hipLaunchKernelGGL(loo<4>, dim3(8 * ((tot + 7) / 8)), dim3(256), 0, 0, buf, K, F, 2, 1, G, 0); });
      printf("K=%d persistent, 4-row panels, %3d sets (%5u wgs):         %6.3f ms %5.2f TB/s\n", K, G, tot, ms, by / ms / 1e9);
    }
  }
  return 0;
}
