// Microbenchmark: HBM read throughput of the row gather of colstats_kernel (statistics-only fold
// stage): a workgroup streams R rows of a row-major [N][K] fp64 matrix, rows chosen by an index
// list, a thread owns two adjacent columns and keeps U rows of loads in flight.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/gather_probe tools/gather_probe.hip
// Patterns: seq (unit = R consecutive rows), mod P (unit = R rows of the fold `row % P`).
// Variants: U rows per group; PIPE = the next group's loads are issued before this group is summed.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef double v2 __attribute__((ext_vector_type(2)));

template <int U, bool PIPE, int NT>
__global__ __launch_bounds__(NT) void gat(const double *X, const double *w, const long *idx, double *out, int K, int R) {
  const long u = blockIdx.x;
  const long *ix = idx + u * R;
  constexpr int CPT = 2;
  for (int c0 = threadIdx.x * CPT; c0 < K; c0 += NT * CPT) {
    double s0 = 0, s1 = 0, q0 = 0, q1 = 0, sw = 0;
    auto acc = [&](v2 x, double wr) {
      const double p0 = x[0] * wr, p1 = x[1] * wr;
      s0 += p0; s1 += p1; q0 += p0 * x[0]; q1 += p1 * x[1]; sw += wr;
    };
    if (!PIPE) {
      long rows[U], nrows[U];
#pragma unroll
      for (int j = 0; j < U; ++j) rows[j] = ix[j];
      for (int r = 0; r + U <= R; r += U) {
        v2 x[U]; double wr[U];
        const bool more = r + 2 * U <= R;
#pragma unroll
        for (int j = 0; j < U; ++j) nrows[j] = more ? ix[r + U + j] : 0;
#pragma unroll
        for (int j = 0; j < U; ++j) { x[j] = *(const v2 *)(X + rows[j] * K + c0); wr[j] = w[rows[j]]; }
#pragma unroll
        for (int j = 0; j < U; ++j) acc(x[j], wr[j]);
#pragma unroll
        for (int j = 0; j < U; ++j) rows[j] = nrows[j];
      }
    } else {
      // two register buffers of U rows: group g+1 is requested before group g is summed
      v2 xa[U], xb[U]; double wa[U], wb[U];
      long rows[U];
#pragma unroll
      for (int j = 0; j < U; ++j) rows[j] = ix[j];
#pragma unroll
      for (int j = 0; j < U; ++j) { xa[j] = *(const v2 *)(X + rows[j] * K + c0); wa[j] = w[rows[j]]; }
#pragma unroll
      for (int j = 0; j < U; ++j) rows[j] = (U + j < R) ? ix[U + j] : 0;
      for (int r = 0; r + U <= R; r += 2 * U) {
        const bool m1 = r + 2 * U <= R, m2 = r + 3 * U <= R;
        if (m1) {
#pragma unroll
          for (int j = 0; j < U; ++j) { xb[j] = *(const v2 *)(X + rows[j] * K + c0); wb[j] = w[rows[j]]; }
#pragma unroll
          for (int j = 0; j < U; ++j) rows[j] = m2 ? ix[r + 2 * U + j] : 0;
        }
#pragma unroll
        for (int j = 0; j < U; ++j) acc(xa[j], wa[j]);
        if (m1) {
          const bool m3 = r + 4 * U <= R;
          if (m2) {
#pragma unroll
            for (int j = 0; j < U; ++j) { xa[j] = *(const v2 *)(X + rows[j] * K + c0); wa[j] = w[rows[j]]; }
#pragma unroll
            for (int j = 0; j < U; ++j) rows[j] = m3 ? ix[r + 3 * U + j] : 0;
          }
#pragma unroll
          for (int j = 0; j < U; ++j) acc(xb[j], wb[j]);
        }
      }
    }
    double *o = out + u * (2 * K + 8);
    o[c0] = s0; o[c0 + 1] = s1; o[K + c0] = q0; o[K + c0 + 1] = q1;
    if (c0 == 0) o[2 * K] = sw;
  }
}

// wave-per-row-group variant: a wave owns all K columns of its rows (lane: CPL columns in pieces
// of 2), the four waves of a workgroup take rows r % 4 == wave; per-wave results summed in LDS order
template <int U, int NT>
__global__ __launch_bounds__(NT) void gatw(const double *X, const double *w, const long *idx, double *out, int K, int R) {
  (void)X; (void)w; (void)idx; (void)out; (void)K; (void)R;
}

struct Case { const char *name; int P; int R; };

template <int U, bool PIPE, int NT> float run(const double *X, const double *w, const long *idx, double *out, int K, int R, long units) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((gat<U, PIPE, NT>), dim3(units), dim3(NT), 0, 0, X, w, idx, out, K, R);
  hipEventRecord(a);
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gat<U, PIPE, NT>), dim3(units), dim3(NT), 0, 0, X, w, idx, out, K, R);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main(int argc, char **argv) {
  const long N = argc > 1 ? atol(argv[1]) : 100000;
  const int K = argc > 2 ? atoi(argv[2]) : 512;
  double *X, *w, *out; long *idx;
  hipMalloc(&X, (size_t)N * K * 8); hipMalloc(&w, N * 8); hipMalloc(&idx, N * 8);
  hipMalloc(&out, (size_t)N * (2 * K + 8) * 8 / 16 + (1 << 20));
  hipMemset(X, 0, (size_t)N * K * 8); hipMemset(w, 0, N * 8);
  std::vector<long> h(N);
  for (int P : {1, 10, 1000}) {
    for (int R : {80, 100, 400, 2000}) {
      if (N / P < R || (N / P) % R || (P < 1000 && R == 100)) continue;
      // unit (f, sp) = rows f + P * (sp * R + i)
      long k = 0;
      for (int f = 0; f < P; ++f)
        for (long i = 0; i < N / P; ++i) h[k++] = f + (long)P * i;
      hipMemcpy(idx, h.data(), N * 8, hipMemcpyHostToDevice);
      const long units = N / R;
      const double mb = (double)N * (K + 1) * 8 / 1e6 + N * 8 / 1e6;
      float t;
#define RUN(U, PIPE, NT) t = run<U, PIPE, NT>(X, w, idx, out, K, R, units); \
      printf("P=%-5d R=%-4d units=%-6ld U=%-2d pipe=%d nt=%d  %7.1f us  %6.0f GB/s\n", P, R, units, U, (int)PIPE, NT, t * 1e3, mb / t / 1e3);
      RUN(4, false, 256) RUN(8, false, 256) RUN(16, false, 256)
      RUN(4, true, 256) RUN(8, true, 256) RUN(16, true, 256)
      RUN(8, false, 128) RUN(8, true, 128) RUN(8, true, 64)
    }
  }
  return 0;
}
