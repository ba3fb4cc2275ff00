// The float32 Gram loop of wgram4_kernel in isolation, WITH its loader waves: 4 compute waves (one per SIMD, a
// 64 x 64 block each) + 4 loader waves that stream 16-row stages global -> LDS by LDS-DMA (12 instructions per
// loader and stage, three stages ahead through a ring of four buffers, one barrier per stage) -- and the compute
// waves' k-steps in two shapes of the same arithmetic:
//   V = 0   v_mfma_f32_16x16x4_f32 : 4 x 4 tiles of 16 x 16, 16 MFMAs of 32 cycles per 4 rows (what the product runs)
//   V = 1   v_mfma_f32_32x32x2_f32 : 2 x 2 tiles of 32 x 32,  8 MFMAs of 64 cycles per 4 rows
//   V = 2   as 1, the two fragments of a side by ONE ds_read_b64 (tile m covers the columns 2 j + m)
//   V = 3   as 0, the four fragments of a side by ONE ds_read_b128 (tile m covers the columns 4 j + m): 3 LDS reads
//           per k-step instead of 9 (round 5's f32_wide_fragments.patch, which did not move the product's launch)
//   V = 4   as 0, but the wave's block is 32 x 128 (2 A-side fragments x 8 B-side fragments) instead of 64 x 64
//           (4 x 4): the A side is the weighted one, so 2 weighting multiplies per k-step instead of 4 -- the probe's
//           own ablation says each costs the MFMA stream ~12 cycles -- for 11 LDS reads instead of 9
// Same LDS bytes, same weighting multiplies, same 64 accumulator registers; half the matrix instructions.
// The question: does the float32 loop (0.79-0.85 of the peak inside the loop, C5 0.81 over the launch) gain from the
// longer MFMA -- fewer issue slots taken from the loader waves that share the SIMDs?
// No results are checked (the sums only keep the work alive); X is random data larger than the caches.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/f32_loop_probe.hip -o tools/f32_loop_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int PITCH = 144, ROWS = 16, PANEL = ROWS * PITCH, BUF = 2 * PANEL + 16, NBUF = 4;

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }

template <int V, bool LOADERS, int ABL = 0>
__global__ __launch_bounds__(512, 2) void kern(const float *X, const float *w, int K, int nstages, float *out,
                                               unsigned long long *clk) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float *smem = reinterpret_cast<float *>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
  const int np = K / 128;
  const int colA0 = (blockIdx.x % np) * 128, colB0 = ((blockIdx.x / np) % np) * 128;
  const long row0 = (long)(blockIdx.x / (np * np)) * nstages * ROWS;
  unsigned long long c0 = 0, q0 = 0;
  if (tid == 0) { c0 = __builtin_amdgcn_s_memtime(); q0 = __builtin_amdgcn_s_memrealtime(); }
  if (wave >= 4) {
    // ---- loaders: the product's scalar-address LDS-DMA, 12 per stage and wave -------------------------------
    const int d = wave - 4;
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)smem_raw);
    const unsigned va = 16u * (unsigned)lane, vw = 4u * (unsigned)lane;
    auto dma16_lo32 = [&](const char *sbase, unsigned voff, unsigned lds_addr) {
      unsigned keep; unsigned long long ex;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b64 exec, 0xffffffff\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep), "=&s"(ex) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    };
    auto dma4_lo1 = [&](const char *sbase, unsigned voff, unsigned lds_addr) {
      unsigned keep; unsigned long long ex;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                   "global_load_lds_dword %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep), "=&s"(ex) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    };
    auto issue = [&](int t) {
      const unsigned bufb = lds0 + (unsigned)((t % NBUF) * BUF) * 4u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int lrow = d + 4 * j;
        const long rn = row0 + (long)t * ROWS + lrow;
        const char *xrow = reinterpret_cast<const char *>(X + rn * (long)K);
        dma16_lo32(xrow + 4l * colA0, va, bufb + (unsigned)(lrow * PITCH) * 4u);
        dma16_lo32(xrow + 4l * colB0, va, bufb + (unsigned)(PANEL + lrow * PITCH) * 4u);
        dma4_lo1(reinterpret_cast<const char *>(w + rn), vw, bufb + (unsigned)(2 * PANEL + lrow) * 4u);
      }
    };
    if (LOADERS) {
      for (int t = 0; t < 3; ++t) issue(t);
      asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_barrier();
#pragma unroll 1
    for (int s = 0; s < nstages; ++s) {
      if (LOADERS) {
        issue(s + 3);
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    return;
  }
  // ---- compute waves ------------------------------------------------------------------------------------------
  if (!LOADERS) {     // (no loaders: some data in the ring so that the operands are not all zero)
    for (int i = tid; i < NBUF * BUF; i += 256) smem[i] = X[i % 4096] - 0.5f;
  }
  const int wr = wave >> 1, wc = wave & 1;
  __syncthreads();
  __syncthreads();
  float total = 0.f;
  if (V == 4) {
    f4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f4){0, 0, 0, 0};
    const int lk = lane >> 4, lc = lane & 15;
    const int a_off = 32 * wave + lc, b_off = PANEL + lc;
    float af[2][2], bf[2][8], wv[2];
    {
      const int r = lk;
#pragma unroll
      for (int m = 0; m < 2; ++m) af[0][m] = smem[a_off + r * PITCH + 16 * m];
#pragma unroll
      for (int n = 0; n < 8; ++n) bf[0][n] = smem[b_off + r * PITCH + 16 * n];
      wv[0] = smem[2 * PANEL + r];
      af[0][0] *= wv[0]; af[0][1] *= wv[0];
    }
#pragma unroll 1
    for (int s = 0; s < nstages; ++s) {
      const float *buf = smem + (s % NBUF) * BUF, *nbuf = smem + ((s + 1) % NBUF) * BUF;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int c = ks & 1;
        const float *rb = ks < 3 ? buf : nbuf;
        const int r = 4 * (ks < 3 ? ks + 1 : 0) + lk;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c][i >> 3], bf[c][i & 7], acc[i], 0, 0, 0);
          if (i == 0) wv[c ^ 1] = rb[2 * PANEL + r];
          else if (i < 3) af[c ^ 1][i - 1] = rb[a_off + r * PITCH + 16 * (i - 1)];
          else if (i < 11) bf[c ^ 1][i - 3] = rb[b_off + r * PITCH + 16 * (i - 3)];
          if ((ABL & 128) && i == 12) __builtin_amdgcn_s_waitcnt(0xC87F);      // lgkmcnt(8): the weight and the A side are back
          if ((ABL & 128) && i == 15) __builtin_amdgcn_s_waitcnt(0xC07F);      // lgkmcnt(0): ONE wait per k-step for the B side
          if ((ABL & 16) && i >= 13 && i < 15) asm volatile("" : "+v"(af[c ^ 1][i - 13]) : "v"(wv[c ^ 1]));   // (the wait for the fragments, no instruction)
          else if ((ABL & 32) && i >= 13 && i < 15) af[c ^ 1][i - 13] = __int_as_float(__float_as_int(af[c ^ 1][i - 13]) ^ (__float_as_int(wv[c ^ 1]) & 1));   // (two integer instructions instead of a float multiply)
          else if ((ABL & 64) && i >= 13 && i < 15) asm volatile("v_mov_b32 %0, %0" : "+v"(af[c ^ 1][i - 13]));      // (a move)
          else if (!(ABL & 1) && !(ABL & 8) && !(ABL & 240) && i >= 13 && i < 15) af[c ^ 1][i - 13] *= wv[c ^ 1];
          if ((ABL & 128) && !(ABL & 1) && i >= 13 && i < 15) af[c ^ 1][i - 13] *= wv[c ^ 1];
          else if ((ABL & 8) && i == 13) { af[c ^ 1][0] *= wv[c ^ 1]; af[c ^ 1][1] *= wv[c ^ 1]; }      // (both behind ONE MFMA)
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (!(ABL & 2)) __syncthreads();
    }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) total += acc[i][j];
  } else if (V == 0 || V == 3) {
    f4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (f4){0, 0, 0, 0};
    const int lk = lane >> 4, lc = lane & 15;
    const int a_off = 64 * wr + (V == 3 ? 4 * lc : lc), b_off = PANEL + 64 * wc + (V == 3 ? 4 * lc : lc);
    constexpr int CS = V == 3 ? 1 : 16;      // column stride between a lane's fragments of one side
    float af[2][4], bf[2][4], wv[2];
    {
      const int r = lk;
#pragma unroll
      for (int m = 0; m < 4; ++m) { af[0][m] = smem[a_off + r * PITCH + CS * m]; bf[0][m] = smem[b_off + r * PITCH + CS * m]; }
      wv[0] = smem[2 * PANEL + r];
#pragma unroll
      for (int m = 0; m < 4; ++m) af[0][m] *= wv[0];
    }
#pragma unroll 1
    for (int s = 0; s < nstages; ++s) {
      const float *buf = smem + (s % NBUF) * BUF, *nbuf = smem + ((s + 1) % NBUF) * BUF;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int c = ks & 1;
        const float *rb = ks < 3 ? buf : nbuf;
        const int r = 4 * (ks < 3 ? ks + 1 : 0) + lk;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[c][i >> 2], bf[c][i & 3], acc[i], 0, 0, 0);
          if (ABL & 4) {
            if (!(ABL & 1) && i >= 11 && i < 15) af[c ^ 1][i - 11] *= wv[c ^ 1];
          } else if (V == 3) {
            if (i == 0) { const f4 t = *reinterpret_cast<const f4 *>(&rb[a_off + r * PITCH]); af[c ^ 1][0] = t[0]; af[c ^ 1][1] = t[1]; af[c ^ 1][2] = t[2]; af[c ^ 1][3] = t[3]; }
            else if (i == 2) { const f4 t = *reinterpret_cast<const f4 *>(&rb[b_off + r * PITCH]); bf[c ^ 1][0] = t[0]; bf[c ^ 1][1] = t[1]; bf[c ^ 1][2] = t[2]; bf[c ^ 1][3] = t[3]; }
            else if (i == 4) wv[c ^ 1] = rb[2 * PANEL + r];
            else if (!(ABL & 1) && i >= 11 && i < 15) af[c ^ 1][i - 11] *= wv[c ^ 1];
          } else if (i < 4) af[c ^ 1][i] = rb[a_off + r * PITCH + 16 * i];
          else if (i < 8) bf[c ^ 1][i - 4] = rb[b_off + r * PITCH + 16 * (i - 4)];
          else if (i == 8) wv[c ^ 1] = rb[2 * PANEL + r];
          else if (!(ABL & 1) && i >= 11 && i < 15) af[c ^ 1][i - 11] *= wv[c ^ 1];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      if (!(ABL & 2)) __syncthreads();
    }
    for (int i = 0; i < 16; ++i) for (int j = 0; j < 4; ++j) total += acc[i][j];
    if (ABL & 4) for (int m = 0; m < 4; ++m) total += af[1][m] + bf[1][m];
  } else {
    f16v acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[i][j] = 0.f;
    const int lk = lane >> 5, lc = lane & 31;
    // V == 1: tile m of a side covers the columns 32 m + lc; V == 2: the columns 2 lc + m (one 8-byte read for both)
    const int a_off = 64 * wr + (V == 2 ? 2 * lc : lc), b_off = PANEL + 64 * wc + (V == 2 ? 2 * lc : lc);
    float af[2][2], bf[2][2], wv[2];
    auto rd = [&](const float *b_, int r, int slot, int part) {
      // part 0: the weight; 1: the A side; 2: the B side
      if (part == 0) wv[slot] = b_[2 * PANEL + r];
      else if (part == 1) {
        if (V == 2) { const f2 t = *reinterpret_cast<const f2 *>(&b_[a_off + r * PITCH]); af[slot][0] = t[0]; af[slot][1] = t[1]; }
        else { af[slot][0] = b_[a_off + r * PITCH]; af[slot][1] = b_[a_off + r * PITCH + 32]; }
      } else {
        if (V == 2) { const f2 t = *reinterpret_cast<const f2 *>(&b_[b_off + r * PITCH]); bf[slot][0] = t[0]; bf[slot][1] = t[1]; }
        else { bf[slot][0] = b_[b_off + r * PITCH]; bf[slot][1] = b_[b_off + r * PITCH + 32]; }
      }
    };
    rd(smem, lk, 0, 0); rd(smem, lk, 0, 1); rd(smem, lk, 0, 2);
    af[0][0] *= wv[0]; af[0][1] *= wv[0];
#pragma unroll 1
    for (int s = 0; s < nstages; ++s) {
      const float *buf = smem + (s % NBUF) * BUF, *nbuf = smem + ((s + 1) % NBUF) * BUF;
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2) {          // 2 rows per MFMA k-step, 16 rows per stage
        const int c = k2 & 1;
        const float *rb = k2 < 7 ? buf : nbuf;
        const int r = 2 * (k2 < 7 ? k2 + 1 : 0) + lk;
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][0], bf[c][0], acc[0], 0, 0, 0);
        rd(rb, r, c ^ 1, 0);
        rd(rb, r, c ^ 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][0], bf[c][1], acc[1], 0, 0, 0);
        rd(rb, r, c ^ 1, 2);
        __builtin_amdgcn_sched_barrier(0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][1], bf[c][0], acc[2], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][1], bf[c][1], acc[3], 0, 0, 0);
        af[c ^ 1][0] *= wv[c ^ 1]; af[c ^ 1][1] *= wv[c ^ 1];
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 16; ++j) total += acc[i][j];
  }
  out[(size_t)blockIdx.x * 256 + tid] = total;
  if (tid == 0) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
    clk[4 * blockIdx.x] = c0; clk[4 * blockIdx.x + 1] = q0; clk[4 * blockIdx.x + 2] = c1; clk[4 * blockIdx.x + 3] = q1;
  }
}

template <int V, bool LOADERS, int ABL = 0>
void run(const float *X, const float *w, int K, int nstages, float *out, unsigned long long *clk, const char *what) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds = (size_t)NBUF * BUF * 4;
  hipFuncSetAttribute((const void *)kern<V, LOADERS, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  float best = 1e30f, ms = 0;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((kern<V, LOADERS, ABL>), dim3(256), dim3(512), lds, 0, X, w, K, nstages, out, clk);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 2 && ms < best) best = ms;
  }
  static unsigned long long h[256 * 4];
  hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost);
  double cyc = 0, tk = 0;
  for (int b = 0; b < 256; ++b) { cyc += (double)(h[4 * b + 2] - h[4 * b]); tk += (double)(h[4 * b + 3] - h[4 * b + 1]); }
  // flops: 256 workgroups x 128 x 128 x 2 per row
  const double fl = 256.0 * (double)nstages * ROWS * 128.0 * 128.0 * 2.0;
  printf("%-64s %8.3f ms  %7.2f TFLOP/s = %.3f of 157.3   %6.0f cycles per stage (2048 = MFMA-bound)  clock %.0f MHz\n", what, best,
         fl / best / 1e9, fl / best / 1e9 / 157.3, cyc / 256.0 / nstages, cyc / tk * 100.0);
}

int main(int argc, char **argv) {
  const int K = 4096, nstages = argc > 1 ? atoi(argv[1]) : 3000;      // 48 000 rows per workgroup
  const int np = K / 128;
  const long groups = (256 + np * np - 1) / (np * np);
  const long rows = groups * (long)(nstages + 4) * ROWS;
  float *X, *w, *out; unsigned long long *clk;
  hipMalloc(&X, (size_t)rows * K * 4); hipMalloc(&w, (size_t)rows * 4); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&clk, 256 * 4 * 8);
  {
    const size_t n = (size_t)rows * K;
    float *h = (float *)malloc(n * 4);
    unsigned s = 12345u;
    for (size_t i = 0; i < n; ++i) { s = s * 1664525u + 1013904223u; h[i] = (float)(s >> 8) * (1.0f / 16777216.0f); }
    hipMemcpy(X, h, n * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, h, (size_t)rows * 4, hipMemcpyHostToDevice);
    free(h);
  }
  printf("float32 Gram loop probe: K = %d, %d stages of 16 rows per workgroup, 256 workgroups, X = %.1f MB\n", K, nstages,
         (double)rows * K * 4 / 1e6);
  run<0, false>(X, w, K, nstages, out, clk, "16x16x4, no loaders (LDS holds fixed data)");
  run<1, false>(X, w, K, nstages, out, clk, "32x32x2, no loaders");
  run<2, false>(X, w, K, nstages, out, clk, "32x32x2 + 8-byte fragment reads, no loaders");
  run<3, false>(X, w, K, nstages, out, clk, "16x16x4 + 16-byte fragment reads, no loaders");
  run<0, true>(X, w, K, nstages, out, clk, "16x16x4 with the loader waves (the product's loop)");
  run<1, true>(X, w, K, nstages, out, clk, "32x32x2 with the loader waves");
  run<2, true>(X, w, K, nstages, out, clk, "32x32x2 + 8-byte fragment reads with the loader waves");
  run<3, true>(X, w, K, nstages, out, clk, "16x16x4 + 16-byte fragment reads with the loader waves");
  run<0, true>(X, w, K, nstages, out, clk, "16x16x4 with the loader waves (again)");
  run<4, false>(X, w, K, nstages, out, clk, "16x16x4, 32 x 128 wave blocks (2 multiplies per k-step), no loaders");
  run<4, true>(X, w, K, nstages, out, clk, "16x16x4, 32 x 128 wave blocks, with the loader waves");
  run<0, true>(X, w, K, nstages, out, clk, "16x16x4, 64 x 64 wave blocks, with the loader waves (again)");
  run<4, true, 1>(X, w, K, nstages, out, clk, "16x16x4, 32 x 128 wave blocks, loaders, no multiplies");
  run<4, true, 8>(X, w, K, nstages, out, clk, "16x16x4, 32 x 128 wave blocks, loaders, both multiplies behind one MFMA");
  run<4, true, 128>(X, w, K, nstages, out, clk, "32 x 128, loaders: explicit waits, one lgkmcnt(8) + one lgkmcnt(0) per k-step");
  run<4, true, 129>(X, w, K, nstages, out, clk, "32 x 128, loaders: explicit waits, no multiplies");
  run<4, true, 16>(X, w, K, nstages, out, clk, "32 x 128, loaders: the multiplies' WAIT for their operands only");
  run<4, true, 32>(X, w, K, nstages, out, clk, "32 x 128, loaders: v_and + v_xor in place of each multiply");
  run<4, true, 64>(X, w, K, nstages, out, clk, "32 x 128, loaders: v_mov in place of each multiply");
  run<4, true>(X, w, K, nstages, out, clk, "16x16x4, 32 x 128 wave blocks, with the loader waves (again)");
  // where the ~300 cycles per stage beyond the 2048 of the MFMAs go (16x16x4, no loaders: nothing else on the CU)
  run<0, false, 1>(X, w, K, nstages, out, clk, "16x16x4, no loaders, no weighting multiplies");
  run<0, false, 2>(X, w, K, nstages, out, clk, "16x16x4, no loaders, no stage barrier");
  run<0, false, 3>(X, w, K, nstages, out, clk, "16x16x4, no loaders, no multiplies, no barrier");
  run<0, false, 4>(X, w, K, nstages, out, clk, "16x16x4, no loaders, no LDS reads (fragments stay in registers)");
  run<0, false, 7>(X, w, K, nstages, out, clk, "16x16x4, no loaders: MFMAs alone");
  run<0, true, 1>(X, w, K, nstages, out, clk, "16x16x4 with the loader waves, no weighting multiplies");
  return 0;
}
