// What does the compute wave's k-step cost on its own?  One wave per SIMD (4 waves per CU,
// one WG per CU), a 64x64 block per wave (16 accumulators), 16-row "stages" of 4 k-steps, as in
// wgram4_kernel, with the pieces added one at a time:
//   0  64 MFMAs per loop iteration, operands fixed in registers
//   1  + the 9 LDS fragment reads of the next k-step (double-buffered registers)
//   2  + the 4 weighting multiplies in the middle of each k-step
//   3  + one s_barrier per stage
//   4  as 0 but 8 MFMAs + branch per iteration (the shape of tools/mfma_peak.hip)
//   5  as 3, but the LDS reads and the multiplies spread between the MFMAs
//      (sched_group_barrier pipeline) instead of issued in two clumps
//   6  as 3, but the fragments come by ds_read_b128: a lane takes columns 2j, 2j+1 of a 32-column
//      group, i.e. the operands of two MFMA tiles (even / odd columns) in one instruction:
//      4 + 1 LDS instructions per k-step instead of 9 (the same bytes)
//   7  the diagonal tile's wave 0: 11 MFMAs per k-step fed by 8 + 1 + 1 b64 reads
//   8  as 7 with b128 reads (4 + 1 + 1)
//   9-11  variants 1-3 with inline-asm MFMAs on AccVGPR accumulators (the compiler pads each: 41-53 TFLOP/s)
//   12/13  one LDS read / multiply pinned behind each MFMA, with / without the stage barrier (72.0: what
//          wgram4_kernel does since round 2)
//   14  as 12 in a function the compiler treats as an AccVGPR user: it then puts the accumulators
//       into AccVGPRs itself -- 50.6 TFLOP/s: not the way on gfx950
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/mfma_mix.hip -o tools/mfma_mix
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int PITCH = 144, ROWS = 16, PANEL = ROWS * PITCH, BUF = 2 * PANEL + 16;

// V = 9, 10, 11: variants 1, 2, 3 with the accumulators in AccVGPRs (inline-asm MFMA, "+a")
template <bool AG> __device__ __forceinline__ void mfma_acc(d4& acc, double a, double b) {
  if (AG) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
  else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
}
template <int V> __global__ __launch_bounds__(512, 2) void kern(const double* in, double* out, int stages) {
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 4 * BUF; i += 512) smem[i] = in[i % 4096];
  __syncthreads();
  if (wave >= 4) return;   // (the real kernel's loader waves)
  d4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (d4){0, 0, 0, 0};
  const int lk = lane >> 4, lc = lane & 15;
  const int a_off = 64 * (wave >> 1) + lc, b_off = PANEL + 64 * (wave & 1) + lc;
  double af[2][4], bf[2][4], wv[2];
  auto read_frags = [&](const double* buf, int ks, int slot) {
    const int r = 4 * ks + lk;
#pragma unroll
    for (int m = 0; m < 4; ++m) af[slot][m] = buf[a_off + r * PITCH + 16 * m];
#pragma unroll
    for (int n = 0; n < 4; ++n) bf[slot][n] = buf[b_off + r * PITCH + 16 * n];
    wv[slot] = buf[2 * PANEL + r];
  };
  read_frags(smem, 0, 0);
  read_frags(smem, 1, 1);
  if (V == 6) {
    typedef double d2 __attribute__((ext_vector_type(2)));
    const int a2 = 64 * (wave >> 1) + 2 * lc, b2 = PANEL + 64 * (wave & 1) + 2 * lc;
    d2 ap[2][2], bp[2][2];
    auto read2 = [&](const double* buf, int ks, int slot) {
      const int r = 4 * ks + lk;
#pragma unroll
      for (int g = 0; g < 2; ++g) ap[slot][g] = *reinterpret_cast<const d2*>(&buf[a2 + r * PITCH + 32 * g]);
#pragma unroll
      for (int g = 0; g < 2; ++g) bp[slot][g] = *reinterpret_cast<const d2*>(&buf[b2 + r * PITCH + 32 * g]);
      wv[slot] = buf[2 * PANEL + r];
    };
    read2(smem, 0, 0);
#pragma unroll 1
    for (int s = 0; s < stages; ++s) {
      const double* buf = smem + (s & 3) * BUF;
      const double* nbuf = smem + ((s + 1) & 3) * BUF;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int c = ks & 1;
        if (ks < 3) read2(buf, ks + 1, c ^ 1); else read2(nbuf, 0, c ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n)
            acc[m * 4 + n] = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[c][m >> 1][m & 1], bp[c][n >> 1][n & 1], acc[m * 4 + n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 2; ++g) { ap[c ^ 1][g][0] *= wv[c ^ 1]; ap[c ^ 1][g][1] *= wv[c ^ 1]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 2; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n)
            acc[m * 4 + n] = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[c][m >> 1][m & 1], bp[c][n >> 1][n & 1], acc[m * 4 + n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
  } else if (V == 7 || V == 8) {
    // diagonal tile, wave W = 0: B fragments of column tiles 0..7, rows 0 and 7, one Y fragment
    typedef double d2 __attribute__((ext_vector_type(2)));
    double bfd[2][8], yf[2], aw[2][2];
    auto readd = [&](const double* buf, int ks, int slot) {
      const int r = 4 * ks + lk;
      if (V == 7) {
#pragma unroll
        for (int j = 0; j < 8; ++j) bfd[slot][j] = buf[r * PITCH + 16 * j + lc];
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const d2 t = *reinterpret_cast<const d2*>(&buf[r * PITCH + 32 * g + 2 * lc]);
          bfd[slot][2 * g] = t[0]; bfd[slot][2 * g + 1] = t[1];
        }
      }
      yf[slot] = buf[PANEL + r * 48 + lc];
      wv[slot] = buf[2 * PANEL + r];
    };
    double st_s[2] = {0, 0}, st_q[2] = {0, 0};
    auto prep = [&](int c) {
      const double x0 = bfd[c][0], x1 = bfd[c][7];
      const double p0 = x0 * wv[c], p1 = x1 * wv[c];
      st_s[0] += p0; st_q[0] += p0 * x0; st_s[1] += p1; st_q[1] += p1 * x1;
      aw[c][0] = p0; aw[c][1] = p1;
    };
    readd(smem, 0, 0);
    prep(0);
#pragma unroll 1
    for (int s = 0; s < stages; ++s) {
      const double* buf = smem + (s & 3) * BUF;
      const double* nbuf = smem + ((s + 1) & 3) * BUF;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int c = ks & 1;
        if (ks < 3) readd(buf, ks + 1, c ^ 1); else readd(nbuf, 0, c ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(aw[c][0], bfd[c][j], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        prep(c ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        acc[8] = __builtin_amdgcn_mfma_f64_16x16x4f64(aw[c][1], bfd[c][7], acc[8], 0, 0, 0);
        acc[9] = __builtin_amdgcn_mfma_f64_16x16x4f64(aw[c][0], yf[c], acc[9], 0, 0, 0);
        acc[10] = __builtin_amdgcn_mfma_f64_16x16x4f64(aw[c][1], yf[c], acc[10], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
    acc[11][0] = st_s[0] + st_q[0] + st_s[1] + st_q[1];
  } else if (V == 12 || V == 13 || V == 14) {
    if (V == 14) { double dummy = in[0]; asm volatile("; agpr hint %0" : "+a"(dummy)); acc[0][0] += dummy; }
    // one LDS read (or one multiply) right behind each MFMA, the order pinned by a scheduling barrier
    // after every pair: the wave's other instructions issue while the matrix pipe is busy
    // (V == 13: without the stage barrier)
#pragma unroll 1
    for (int s = 0; s < stages; ++s) {
      const double* buf = smem + (s & 3) * BUF;
      const double* nbuf = smem + ((s + 1) & 3) * BUF;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int c = ks & 1;
        const double* rb = ks < 3 ? buf : nbuf;
        const int r = 4 * (ks < 3 ? ks + 1 : 0) + lk;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int m = i >> 2, n = i & 3;
          acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[c][m], bf[c][n], acc[i], 0, 0, 0);
          if (i < 4) af[c ^ 1][i] = rb[a_off + r * PITCH + 16 * i];
          else if (i < 8) bf[c ^ 1][i - 4] = rb[b_off + r * PITCH + 16 * (i - 4)];
          else if (i == 8) wv[c ^ 1] = rb[2 * PANEL + r];
          else if (i >= 11 && i < 15) af[c ^ 1][i - 11] *= wv[c ^ 1];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (V == 12 || V == 14) __syncthreads();
    }
  } else if (V == 4) {
    for (int s = 0; s < stages * 8; ++s) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n)
          acc[m * 4 + n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[0][m], bf[0][n], acc[m * 4 + n], 0, 0, 0);
    }
  } else if (V == 5) {
#pragma unroll 1
    for (int s = 0; s < stages; ++s) {
      const double* buf = smem + (s & 3) * BUF;
      const double* nbuf = smem + ((s + 1) & 3) * BUF;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int c = ks & 1;
        if (ks < 3) read_frags(buf, ks + 1, c ^ 1); else read_frags(nbuf, 0, c ^ 1);
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n)
            acc[m * 4 + n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[c][m], bf[c][n], acc[m * 4 + n], 0, 0, 0);
#pragma unroll
        for (int m = 0; m < 4; ++m) af[c ^ 1][m] *= wv[c ^ 1];
        // pipeline: MFMA, DS read, MFMA, DS read ... then MFMAs, then MFMA, VALU ...
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // address VALU for the read
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 5, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
  } else {
    constexpr bool AG = V >= 9;
    constexpr int W = AG ? V - 8 : V;
#pragma unroll 1
    for (int s = 0; s < stages; ++s) {
      const double* buf = smem + (s & 3) * BUF;
      const double* nbuf = smem + ((s + 1) & 3) * BUF;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int c = ks & 1;
        if (W >= 1) { if (ks < 3) read_frags(buf, ks + 1, c ^ 1); else read_frags(nbuf, 0, c ^ 1); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) mfma_acc<AG>(acc[m * 4 + n], af[c][m], bf[c][n]);
        __builtin_amdgcn_sched_barrier(0);
        if (W >= 2) {
#pragma unroll
          for (int m = 0; m < 4; ++m) af[c ^ 1][m] *= wv[c ^ 1];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 2; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) mfma_acc<AG>(acc[m * 4 + n], af[c][m], bf[c][n]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (W >= 3) __syncthreads();
    }
  }
  double sacc = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) sacc += acc[i][j];
  out[blockIdx.x * 512 + tid] = sacc;
}

// 8 compute waves per CU (two per SIMD), a 64x32 block each: the same work per CU and stage.
//   V = 0: reads + weighting (of the 2 B fragments) + barrier, in two clumps per k-step
template <int V> __global__ __launch_bounds__(768, 1) void kern8(const double* in, double* out, int stages) {
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int i = tid; i < 4 * BUF; i += 768) smem[i] = in[i % 4096];
  __syncthreads();
  if (wave >= 8) return;   // (loader waves)
  d4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = (d4){0, 0, 0, 0};
  const int lk = lane >> 4, lc = lane & 15;
  const int a_off = 64 * (wave >> 2) + lc, b_off = PANEL + 32 * (wave & 3) + lc;
  double af[2][4], bf[2][2], wv[2];
  auto read_frags = [&](const double* buf, int ks, int slot) {
    const int r = 4 * ks + lk;
#pragma unroll
    for (int m = 0; m < 4; ++m) af[slot][m] = buf[a_off + r * PITCH + 16 * m];
#pragma unroll
    for (int n = 0; n < 2; ++n) bf[slot][n] = buf[b_off + r * PITCH + 16 * n];
    wv[slot] = buf[2 * PANEL + r];
  };
  read_frags(smem, 0, 0);
#pragma unroll 1
  for (int s = 0; s < stages; ++s) {
    const double* buf = smem + (s & 3) * BUF;
    const double* nbuf = smem + ((s + 1) & 3) * BUF;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = ks & 1;
      if (ks < 3) read_frags(buf, ks + 1, c ^ 1); else read_frags(nbuf, 0, c ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
          acc[m * 2 + n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[c][m], bf[c][n], acc[m * 2 + n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int n = 0; n < 2; ++n) bf[c ^ 1][n] *= wv[c ^ 1];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int m = 2; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
          acc[m * 2 + n] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[c][m], bf[c][n], acc[m * 2 + n], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (V == 0) __syncthreads();
  }
  double sacc = 0;
  for (int i = 0; i < 8; ++i) for (int j = 0; j < 4; ++j) sacc += acc[i][j];
  out[blockIdx.x * 768 + tid] = sacc;
}

template <int V> void run8(const double* din, double* dout, const char* what) {
  const int stages = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)kern8<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * BUF * 8);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern8<V>, dim3(256), dim3(768), 4 * BUF * 8, 0, din, dout, stages);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double mf = 256.0 * 4 * stages * 64.0;
  printf("%-58s %.3f ms  %.2f TFLOP/s  %.1f ns/stage\n", what, ms, mf * 2048.0 / ms / 1e9, ms * 1e6 / stages);
}

template <int V> void run(const double* din, double* dout, const char* what, double mfma_per_stage = 64.0) {
  const int stages = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipFuncSetAttribute((const void*)kern<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 4 * BUF * 8);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern<V>, dim3(256), dim3(512), 4 * BUF * 8, 0, din, dout, stages);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  const double mf = 256.0 * 4 * stages * mfma_per_stage;
  printf("%-58s %.3f ms  %.2f TFLOP/s  %.1f ns/stage\n", what, ms, mf * 2048.0 / ms / 1e9, ms * 1e6 / stages);
}

int main() {
  double* din; double* dout;
  hipMalloc(&din, 4096 * 8); hipMalloc(&dout, 256 * 768 * 8);
  double h[4096];
  for (int i = 0; i < 4096; ++i) h[i] = (rand() / (double)RAND_MAX) - 0.5;
  hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
  run<4>(din, dout, "8 MFMAs per branch, registers only");
  run<0>(din, dout, "64 MFMAs per branch, registers only");
  run<1>(din, dout, "+ 9 LDS fragment reads per k-step");
  run<2>(din, dout, "+ 4 weighting multiplies per k-step");
  run<3>(din, dout, "+ one barrier per stage (4 waves)");
  run<9>(din, dout, "AccVGPR accumulators: + 9 LDS fragment reads per k-step");
  run<10>(din, dout, "AccVGPR accumulators: + 4 weighting multiplies");
  run<11>(din, dout, "AccVGPR accumulators: + one barrier per stage");
  run<5>(din, dout, "same, reads and multiplies spread between the MFMAs");
  run<12>(din, dout, "one read / multiply behind each MFMA, order pinned, + barrier");
  run<13>(din, dout, "one read / multiply behind each MFMA, order pinned, no barrier");
  run<14>(din, dout, "the same + barrier, function marked as an AccVGPR user");
  run<6>(din, dout, "as the barrier line, fragments by ds_read_b128");
  run<7>(din, dout, "diagonal tile wave 0: 11 MFMAs, 10 b64 reads per k-step", 44.0);
  run<8>(din, dout, "diagonal tile wave 0: 11 MFMAs, 6 reads (b128)", 44.0);
  run8<0>(din, dout, "8 waves x 64x32 (2 per SIMD): reads + weighting + barrier");
  run8<1>(din, dout, "8 waves x 64x32 (2 per SIMD): reads + weighting, no barrier");
  return 0;
}
