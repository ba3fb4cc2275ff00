// Round 6: what a CHIP-RESIDENT G could reach in the HBM regime.  hipcc --offload-arch=gfx950 -O3 -o tools/resident_probe tools/resident_probe.hip
//
// The small-fold kernels keep a 64 x 64 tile of G in registers for eight folds and write it direct + mirrored; their store
// pattern tops out at 5.1-5.4 (float32) / 6.0 (float64) TB/s (xcd_stack_probe.hip).  The register files of the whole chip
// hold 128 MiB: a K = 4096 float32 G (64 MiB) fits at 128 registers per lane and two workgroups per CU.  A launch of 512
// persistent workgroups that each own a 32-row x 1024-column block of G for ALL folds reads G once per launch, computes
// both triangles (the matrix cores idle two thirds of a 16-row fold anyway), has every workgroup of the chip inside the
// SAME output matrix at any time, and stores whole 128-byte lines straight from the MFMA accumulators (32x32x2: 32 lanes
// on a row).  This probe measures that pattern: stores alone, then + the fold's rows through LDS, + the MFMAs, + the
// finishing arithmetic, for several block shapes / workgroup -> block maps / store orders, next to a linear fill.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float v4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef double d4v __attribute__((ext_vector_type(4)));

enum { F_READ = 1, F_MFMA = 2, F_FIN = 4, F_SYNC = 8, F_G = 16 };

__device__ __forceinline__ unsigned map_block(unsigned b, unsigned W, unsigned nbc, unsigned map) {
  // b -> XCD b % 8.  0: as dispatched; 1: XCD-contiguous ranges; 2: the chunks of a band on one XCD, bands round-robin
  if (map == 1) return (b & 7) * (W / 8) + (b >> 3);
  if (map == 2) { const unsigned x = b & 7, j = b >> 3; const unsigned band = (j / nbc) * 8 + x, ch = j % nbc; return band * nbc + ch; }
  return b;
}

// float32: block = 32 rows x (4 waves x NT tiles of 32 columns); tile registers v = 0..15: row (v % 4) + 8 (v / 4) + 4 (lane / 32),
// column lane % 32 (v_mfma_f32_32x32x2f32).  N = rows of a fold (multiple of 2).
template <int NT, int N, unsigned order, unsigned flags> __global__ __launch_bounds__(256, NT == 8 ? 2 : 4) void res_f32(float *__restrict__ out, const float *__restrict__ G,
    const float *__restrict__ xv, const float *__restrict__ stats, unsigned K, unsigned nmat, unsigned map, unsigned *cnt) {
  constexpr unsigned BC = 4 * NT * 32, PITCH = BC + 32;
  extern __shared__ float lds[];                           // N x PITCH: the fold's rows at the block's columns
  const unsigned nbc = K / BC, W = gridDim.x;
  const unsigned blk = map_block(blockIdx.x, W, nbc, map);
  const unsigned band = blk / nbc, ch = blk % nbc;
  const unsigned r0 = band * 32, c0 = ch * BC;
  const unsigned tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l32 = lane & 31, lh = lane >> 5;
  const unsigned cw = c0 + wave * NT * 32;                 // first column of this wave
  unsigned voff[16];                                        // byte offsets of this lane's 16 rows from (r0, cw)
#pragma unroll
  for (int v = 0; v < 16; ++v) voff[v] = 4 * (((v & 3) + 8 * (v >> 2) + 4 * lh) * K + l32);   // in bytes: saddr + 32-bit voffset + immediate
  f16v g[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int v = 0; v < 16; ++v) {
      const unsigned row = r0 + (v & 3) + 8 * (v >> 2) + 4 * lh;
      g[t][v] = (flags & F_G) ? G[(size_t)row * K + cw + 32 * t + l32] : (float)(row + t);
    }
  const size_t mat = (size_t)K * K;
  for (unsigned f = 0; f < nmat; ++f) {
    char *o = reinterpret_cast<char *>(out + f * mat + (size_t)r0 * K + cw);
    const float *x = xv + (size_t)f * N * K;
    float af[N / 2];
    if (flags & F_READ) {
      __syncthreads();
      // the fold's N rows at the block's BC columns -> LDS (16-byte pieces, contiguous per row)
#pragma unroll 4
      for (unsigned e = tid; e < N * (BC / 4); e += 256) {
        const unsigned r = e / (BC / 4), c4 = e % (BC / 4);
        const v4 val = *reinterpret_cast<const v4 *>(x + (size_t)r * K + c0 + 4 * c4);
        *reinterpret_cast<v4 *>(lds + r * PITCH + 4 * c4) = val;
      }
#pragma unroll
      for (int kk = 0; kk < N / 2; ++kk) af[kk] = x[(size_t)(2 * kk + lh) * K + r0 + l32] * 0.5f;   // "weighted" A side
      __syncthreads();
    } else {
#pragma unroll
      for (int kk = 0; kk < N / 2; ++kk) af[kk] = (float)(f + kk);
    }
    if (flags & F_SYNC) {
      // soft lock-step: do not start fold f before every workgroup has finished fold f - 2 (bounded spin)
      if (f >= 2 && tid == 0) {
        unsigned spins = 0;
        while (__hip_atomic_load(cnt + f - 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < W && ++spins < 100000) __builtin_amdgcn_s_sleep(2);
      }
      __syncthreads();
    }
    // the row side of the two outer products of the finish (means, reciprocal stds): MFMA A operands, k = 0 only
    float amu = 0.f, asd = 0.f;
    if (flags & F_FIN) {
      amu = lh ? 0.f : stats[(size_t)f * 2 * K + r0 + l32];
      asd = lh ? 0.f : stats[(size_t)f * 2 * K + K + r0 + l32];
    }
    auto tile = [&](int t, f16v &res) {
      f16v acc;
      if (flags & F_MFMA) {
        // the chain starts from the tile of G (C operand) with a negated A side: G - sum (w x_r) x_c, then one more k-step for
        // the centring term (sqrt(sw) mu on both sides: symmetric), then ONE multiply per element by the outer product of the
        // reciprocal stds, itself an MFMA
        acc = g[t];
#pragma unroll
        for (int kk = 0; kk < N / 2; ++kk) {
          const float bf = (flags & F_READ) ? lds[(2 * kk + lh) * PITCH + wave * NT * 32 + 32 * t + l32] : (float)(kk + t);
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[kk], bf, acc, 0, 0, 0);
        }
      } else if (flags & F_READ) {
        float s = 0.f;
#pragma unroll
        for (int kk = 0; kk < N / 2; ++kk) s += lds[(2 * kk + lh) * PITCH + wave * NT * 32 + 32 * t + l32] * af[kk];
        acc = g[t];
        acc[0] = s;
      } else { acc = g[t]; acc[0] = (float)f; }
      if (flags & F_FIN) {
        const float muc = lh ? 0.f : stats[(size_t)f * 2 * K + cw + 32 * t + l32], sdc = lh ? 0.f : stats[(size_t)f * 2 * K + K + cw + 32 * t + l32];
        f16v ps;
#pragma unroll
        for (int v = 0; v < 16; ++v) ps[v] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(amu, muc, acc, 0, 0, 0);
        ps = __builtin_amdgcn_mfma_f32_32x32x2f32(asd, sdc, ps, 0, 0, 0);
#pragma unroll
        for (int v = 0; v < 16; ++v) res[v] = acc[v] * ps[v];
      } else res = acc;
    };
    if (order == 0) {
      // tile-major: a tile's sixteen 2 x 128-byte stores one after the other
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        f16v res; tile(t, res);
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          __builtin_nontemporal_store(res[v], reinterpret_cast<float *>(o + voff[v] + 128 * t));
        }
        __builtin_amdgcn_sched_barrier(0);                  // one tile's temporaries at a time
      }
    } else {
      // row-major in groups of TG tiles: TG x 128 contiguous bytes per row and store burst
      constexpr int TG = NT == 8 ? 2 : (NT < 4 ? NT : 4);
#pragma unroll
      for (int t0 = 0; t0 < NT; t0 += TG) {
        f16v res[TG];
#pragma unroll
        for (int t = 0; t < TG; ++t) tile(t0 + t, res[t]);
#pragma unroll
        for (int v = 0; v < 16; ++v) {
#pragma unroll
          for (int t = 0; t < TG; ++t) __builtin_nontemporal_store(res[t][v], reinterpret_cast<float *>(o + voff[v] + 128 * (t0 + t)));
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (flags & F_SYNC) {
      __syncthreads();
      if (tid == 0) __hip_atomic_fetch_add(cnt + f, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// float64: block = BRT x 16 rows x (4 waves x NT tiles of 16 columns); tile registers r = 0..3: row (lane / 16) + 4 r, column lane % 16
// (v_mfma_f64_16x16x4f64).  Rows [row0, row0 + rows) of the matrix only (a K = 4096 float64 G is resident one half at a time).
template <int NT, int BRT, int N> __global__ __launch_bounds__(256, 2) void res_f64(double *__restrict__ out, const double *__restrict__ xv,
    unsigned K, unsigned row0, unsigned nmat, unsigned map, unsigned order, unsigned flags) {
  constexpr unsigned BC = 4 * NT * 16, PITCH = BC + 16;
  extern __shared__ double ldsd[];
  const unsigned nbc = K / BC, W = gridDim.x;
  const unsigned blk = map_block(blockIdx.x, W, nbc, map);
  const unsigned band = blk / nbc, ch = blk % nbc;
  const unsigned r0 = row0 + band * 16 * BRT, c0 = ch * BC;
  const unsigned tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, l16 = lane & 15, lq = lane >> 4;
  const unsigned cw = c0 + wave * NT * 16;
  d4v g[BRT][NT];
#pragma unroll
  for (int b = 0; b < BRT; ++b)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) g[b][t][r] = (double)(r0 + 16 * b + lq + 4 * r + t);
  const size_t mat = (size_t)K * K;
  for (unsigned f = 0; f < nmat; ++f) {
    double *o = out + f * mat;
    const double *x = xv + (size_t)f * N * K;
    double af[BRT][N / 4];
    if (flags & F_READ) {
      __syncthreads();
      for (unsigned e = tid; e < N * (BC / 2); e += 256) {
        const unsigned r = e / (BC / 2), c2 = e % (BC / 2);
        const v4 val = *reinterpret_cast<const v4 *>(x + (size_t)r * K + c0 + 2 * c2);
        *reinterpret_cast<v4 *>(ldsd + r * PITCH + 2 * c2) = val;
      }
#pragma unroll
      for (int b = 0; b < BRT; ++b)
#pragma unroll
        for (int kk = 0; kk < N / 4; ++kk) af[b][kk] = x[(size_t)(4 * kk + lq) * K + r0 + 16 * b + l16] * 0.5;
      __syncthreads();
    } else {
#pragma unroll
      for (int b = 0; b < BRT; ++b)
#pragma unroll
        for (int kk = 0; kk < N / 4; ++kk) af[b][kk] = (double)(f + kk);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      double bf[N / 4];
#pragma unroll
      for (int kk = 0; kk < N / 4; ++kk) bf[kk] = (flags & F_READ) ? ldsd[(4 * kk + lq) * PITCH + wave * NT * 16 + 16 * t + l16] : (double)(kk + t);
#pragma unroll
      for (int b = 0; b < BRT; ++b) {
        d4v acc = {0, 0, 0, 0};
        if (flags & F_MFMA) {
#pragma unroll
          for (int kk = 0; kk < N / 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[b][kk], bf[kk], acc, 0, 0, 0);
        } else acc[0] = bf[0] * af[b][0];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const unsigned row = r0 + 16 * b + lq + 4 * r;
          double val = g[b][t][r] - acc[r];
          if (flags & F_FIN) { val -= 3.0 * (af[b][0] * bf[0]); val = val * (af[b][1] * bf[1]); }
          __builtin_nontemporal_store(val, o + (size_t)row * K + cw + 16 * t + l16);
        }
      }
    }
    (void)order;
  }
}

// 16 bytes per lane, whole rows: workgroup w owns R rows of K elements (ES bytes each); inter = 0: rows w R ... w R + R - 1,
// inter = 1: rows w, w + W, w + 2 W, ...  (every workgroup of the chip writes neighbouring rows at the same time)
__global__ __launch_bounds__(256, 2) void res_lin(v4 *out, unsigned row16, unsigned R, unsigned nmat, unsigned inter, unsigned map) {
  const unsigned W = gridDim.x, w = map_block(blockIdx.x, W, 1, map), tid = threadIdx.x;
  const size_t mat16 = (size_t)row16 * R * W;
  for (unsigned f = 0; f < nmat; ++f) {
    v4 *o = out + f * mat16;
    const v4 v = {(float)f, 1, 2, 3};
    for (unsigned i = 0; i < R; ++i) {
      const size_t row = inter ? (size_t)w + (size_t)W * i : (size_t)w * R + i;
      for (unsigned c = tid; c < row16; c += 256) __builtin_nontemporal_store(v, o + row * row16 + c);
    }
  }
}
__global__ __launch_bounds__(256) void chunk_fill(v4 *p) {
  __builtin_nontemporal_store((v4){0, 0, 0, 0}, p + (size_t)blockIdx.x * 256 + threadIdx.x);
}

template <typename F> float timeit(F f, int rep = 5) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) f();
  hipEventRecord(a);
  for (int i = 0; i < rep; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return ms / rep;
}
static const char *flagname(unsigned fl) {
  static char s[64];
  snprintf(s, 64, "%s%s%s%s%s", fl & F_G ? "G " : "", fl & F_READ ? "rows " : "", fl & F_MFMA ? "mfma " : "", fl & F_FIN ? "finish " : "", fl & F_SYNC ? "sync " : "");
  return fl ? s : "stores only";
}

int main() {
  const size_t bytes = (size_t)3 << 30;
  char *p; if (hipMalloc(&p, bytes + (64u << 20)) != hipSuccess) return 1;
  float *xv; hipMalloc(&xv, (size_t)96 * 32 * 4096 * 8);
  hipMemset(xv, 0, (size_t)96 * 32 * 4096 * 8);
  float *stats; hipMalloc(&stats, (size_t)96 * 2 * 4096 * 4); hipMemset(stats, 0, (size_t)96 * 2 * 4096 * 4);
  unsigned *cnt; hipMalloc(&cnt, 4096);
  float *G = (float *)(p + bytes);
  hipMemset(G, 0, 64u << 20);
  {
    const float ms = timeit([&] { hipLaunchKernelGGL(chunk_fill, dim3((unsigned)(bytes / 4096)), dim3(256), 0, 0, (v4 *)p); });
    printf("linear fill, one workgroup per 4 KiB: %5.2f TB/s\n", bytes / ms / 1e9);
  }
  // ---- float32, K = 4096, 48 matrices ----
  {
    const unsigned K = 4096, nmat = 48;
    const double by = (double)nmat * K * K * 4;
    printf("float32 K = 4096, 48 folds of 16 rows; persistent workgroups owning 32-row blocks of G (MFMA 32x32x2 layout: 2 x 128 B per store)\n");
    for (unsigned R : {8u, 4u})
      for (unsigned inter = 0; inter < 2; ++inter)
        for (unsigned map : {0u, 1u}) {
          const unsigned W = K / R;
          const float ms = timeit([&] { hipLaunchKernelGGL(res_lin, dim3(W), dim3(256), 0, 0, (v4 *)p, K / 4, R, nmat, inter, map); });
          printf("  16 B per lane, %u workgroups x %u %s rows, map %u: %5.2f TB/s\n", W, R, inter ? "interleaved" : "consecutive", map, by / ms / 1e9);
        }
    auto run32 = [&](auto kern, int NT, int N, unsigned map, unsigned order, unsigned flags) {
      const unsigned BC = 4 * NT * 32, W = (K / 32) * (K / BC);
      const size_t lds = (size_t)N * (BC + 32) * 4;
      hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      const float ms = timeit([&] {
        if (flags & F_SYNC) hipMemsetAsync(cnt, 0, 4096, 0);
        hipLaunchKernelGGL(kern, dim3(W), dim3(256), lds, 0, (float *)p, G, xv, stats, K, nmat, map, cnt);
      });
      printf("  blocks 32 x %4u (%4u workgroups), n = %2d, map %u, %s, %-28s: %6.3f ms  %5.2f TB/s\n", BC, W, N, map, order ? "row-major " : "tile-major", flagname(flags), ms, by / ms / 1e9);
    };
#define RUN32(NT, N, MAP, ORDER, FL) run32(res_f32<NT, N, ORDER, FL>, NT, N, MAP, ORDER, FL)
    for (unsigned map : {0u, 1u, 2u}) { RUN32(8, 16, map, 0, 0); RUN32(8, 16, map, 1, 0); }
    for (unsigned map : {0u, 1u, 2u}) { RUN32(4, 16, map, 0, 0); RUN32(4, 16, map, 1, 0); }
#define LADDER(NT, ORDER) RUN32(NT, 16, 0, ORDER, F_READ); RUN32(NT, 16, 0, ORDER, F_READ | F_MFMA); RUN32(NT, 16, 0, ORDER, F_G | F_READ | F_MFMA | F_FIN); \
    RUN32(NT, 16, 0, ORDER, F_G | F_READ | F_MFMA | F_FIN | F_SYNC)
    LADDER(8, 0); LADDER(8, 1); LADDER(4, 0); LADDER(4, 1);
    RUN32(8, 32, 0, 0, F_G | F_READ | F_MFMA | F_FIN);
    RUN32(4, 32, 0, 0, F_G | F_READ | F_MFMA | F_FIN);
    RUN32(8, 16, 1, 0, F_G | F_READ | F_MFMA | F_FIN);
    RUN32(4, 16, 1, 0, F_G | F_READ | F_MFMA | F_FIN);
  }
  // ---- float64 ----
  {
    printf("float64 (MFMA 16x16x4 layout: 4 x 128 B per store)\n");
    auto run64 = [&](auto kern, int NT, int BRT, int N, unsigned K, unsigned rows, unsigned nmat, unsigned map, unsigned flags) {
      const unsigned BC = 4 * NT * 16, W = (rows / (16 * BRT)) * (K / BC);
      const size_t lds = (size_t)N * (BC + 16) * 8;
      hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      const double by = (double)nmat * rows * K * 8;
      const float ms = timeit([&] { hipLaunchKernelGGL(kern, dim3(W), dim3(256), lds, 0, (double *)p, (const double *)xv, K, 0u, nmat, map, 0u, flags); });
      printf("  K = %u rows %u: blocks %2d x %4u (%4u workgroups), n = %2d, map %u, %-24s: %6.3f ms  %5.2f TB/s\n", K, rows, 16 * BRT, BC, W, N, map, flagname(flags), ms, by / ms / 1e9);
    };
    for (unsigned fl : {0u, (unsigned)F_READ, (unsigned)(F_READ | F_MFMA | F_FIN)})
      for (unsigned map : {0u, 1u}) {
        run64(res_f64<8, 2, 16>, 8, 2, 16, 4096, 2048, 24, map, fl);      // half of a K = 4096 matrix: 64 MiB resident, 512 workgroups
        run64(res_f64<8, 2, 16>, 8, 2, 16, 2048, 2048, 96, map, fl);      // K = 2048: 256 workgroups x 128 KB
        run64(res_f64<8, 1, 16>, 8, 1, 16, 2048, 2048, 96, map, fl);      // K = 2048: 512 workgroups x 64 KB
        run64(res_f64<4, 2, 16>, 4, 2, 16, 2048, 2048, 96, map, fl);      // K = 2048: 512 workgroups, blocks 32 x 256
      }
  }
  hipFree(p);
  return 0;
}
