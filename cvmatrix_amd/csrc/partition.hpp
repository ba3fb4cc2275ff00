// partition.hpp -- part of libcvmhip.so (included by cvmhip.hip inside its anonymous namespace).
// Device-side Partitioner: integer fold labels in [0, L) -> validation indices grouped by label
// (ascending inside a group, like cvmatrix/partitioner.py:101-107) + offsets: a stable counting
// sort in three small kernels.  Integer atomics only (histogram, first appearance): the result
// does not depend on their order.
#pragma once

constexpr int PART_THREADS = 256;
constexpr int PART_MAXL = 4096;     // labels per call (LDS histogram); more: use the host Partitioner

struct PartArgs {
  const int64_t *labels;
  int64_t N, chunk;                 // rows, rows per block
  int L, nb;                        // labels, blocks
  int *blockhist;                   // [nb][L]
  int64_t *blockoff;                // [nb][L]
  unsigned long long *first;        // [L] first row of each label (N: label absent)
  int64_t *offsets;                 // [L+1]
  int64_t *idx_out;                 // [N]
  int *err;                         // set to 1 if a label is outside [0, L)
};

__global__ void part_init_kernel(const PartArgs a) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < a.L) a.first[t] = (unsigned long long)a.N;
  if (t == 0) *a.err = 0;
}

__global__ __launch_bounds__(PART_THREADS) void part_hist_kernel(const PartArgs a) {
  extern __shared__ int hist[];
  for (int l = threadIdx.x; l < a.L; l += PART_THREADS) hist[l] = 0;
  __syncthreads();
  const int64_t r0 = (int64_t)blockIdx.x * a.chunk;
  const int64_t r1 = r0 + a.chunk < a.N ? r0 + a.chunk : a.N;
  for (int64_t i = r0 + threadIdx.x; i < r1; i += PART_THREADS) {
    const int64_t l = a.labels[i];
    if (l < 0 || l >= a.L) { *a.err = 1; continue; }
    atomicAdd(&hist[l], 1);
    atomicMin(&a.first[l], (unsigned long long)i);
  }
  __syncthreads();
  for (int l = threadIdx.x; l < a.L; l += PART_THREADS) a.blockhist[(size_t)blockIdx.x * a.L + l] = hist[l];
}

// offsets[l] = rows with a smaller label; blockoff[b][l] = where block b's rows of label l start
__global__ __launch_bounds__(PART_THREADS) void part_scan_kernel(const PartArgs a) {
  for (int l = threadIdx.x; l < a.L; l += PART_THREADS) {
    int64_t tot = 0;
    for (int b = 0; b < a.nb; ++b) tot += a.blockhist[(size_t)b * a.L + l];
    a.offsets[l + 1] = tot;          // totals first, prefix below
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int64_t run = 0;
    a.offsets[0] = 0;
    for (int l = 0; l < a.L; ++l) { run += a.offsets[l + 1]; a.offsets[l + 1] = run; }
  }
  __syncthreads();
  for (int l = threadIdx.x; l < a.L; l += PART_THREADS) {
    int64_t run = a.offsets[l];
    for (int b = 0; b < a.nb; ++b) {
      a.blockoff[(size_t)b * a.L + l] = run;
      run += a.blockhist[(size_t)b * a.L + l];
    }
  }
}

// stable scatter: blocks own consecutive row chunks, tiles of 256 rows inside a chunk and the
// four waves of a tile go in order, lanes of a wave rank themselves among equal labels
__global__ __launch_bounds__(PART_THREADS) void part_scatter_kernel(const PartArgs a) {
  extern __shared__ long long cursor[];
  for (int l = threadIdx.x; l < a.L; l += PART_THREADS) cursor[l] = a.blockoff[(size_t)blockIdx.x * a.L + l];
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t r0 = (int64_t)blockIdx.x * a.chunk;
  const int64_t r1 = r0 + a.chunk < a.N ? r0 + a.chunk : a.N;
  for (int64_t t0 = r0; t0 < r1; t0 += PART_THREADS) {
    for (int w = 0; w < PART_THREADS / 64; ++w) {
      if (wave == w) {
        const int64_t i = t0 + 64 * w + lane;
        long long l = -1;
        if (i < r1) { l = a.labels[i]; if (l < 0 || l >= a.L) l = -1; }
        int lower = 0, total = 0;
        for (int j = 0; j < 64; ++j) {
          const long long lj = __shfl(l, j);
          if (lj == l) { ++total; if (j < lane) ++lower; }
        }
        if (l >= 0) {
          a.idx_out[cursor[l] + lower] = i;
          if (lower == total - 1) cursor[l] += total;   // the group's last lane moves the cursor on
        }
      }
      __syncthreads();
    }
  }
}

inline size_t partition_workspace_bytes(int64_t N, int L) {
  (void)N;
  const size_t nb = 256;
  return align_up(nb * (size_t)L * sizeof(int), 256) + align_up(nb * (size_t)L * sizeof(int64_t), 256) +
         align_up((size_t)L * 8, 256) + 256;
}

inline int partition_impl(const int64_t *labels, int64_t N, int L, int64_t *idx_out, int64_t *offsets,
                          int64_t *first, int32_t *err, void *ws, size_t ws_bytes, hipStream_t st) {
  if (L < 1 || L > PART_MAXL) return fail(CVM_EINVAL, "cvm_partition_labels: 1 <= n_labels <= 4096%s");
  if (ws_bytes < partition_workspace_bytes(N, L)) return fail(CVM_EWORKSPACE, "cvm_partition_labels: workspace too small%s");
  PartArgs a;
  int64_t nb = (N + 1023) / 1024;
  if (nb > 256) nb = 256;
  if (nb < 1) nb = 1;
  a.labels = labels; a.N = N; a.L = L; a.nb = (int)nb;
  a.chunk = ((N + nb - 1) / nb + PART_THREADS - 1) / PART_THREADS * PART_THREADS;
  char *p = (char *)ws;
  a.blockhist = (int *)p; p += align_up((size_t)256 * L * sizeof(int), 256);
  a.blockoff = (int64_t *)p; p += align_up((size_t)256 * L * sizeof(int64_t), 256);
  a.first = (unsigned long long *)first;
  a.offsets = offsets; a.idx_out = idx_out; a.err = (int *)err;
  hipLaunchKernelGGL(part_init_kernel, dim3((L + 255) / 256), dim3(256), 0, st, a);
  hipLaunchKernelGGL(part_hist_kernel, dim3((unsigned)nb), dim3(PART_THREADS), (size_t)L * sizeof(int), st, a);
  hipLaunchKernelGGL(part_scan_kernel, dim3(1), dim3(PART_THREADS), 0, st, a);
  hipLaunchKernelGGL(part_scatter_kernel, dim3((unsigned)nb), dim3(PART_THREADS), (size_t)L * sizeof(long long), st, a);
  HIP_OK(hipGetLastError());
  return CVM_OK;
}
