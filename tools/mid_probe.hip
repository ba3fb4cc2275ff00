// mid_probe.hip -- the mid-size-fold kernels of the library in a standalone harness (measurements only).
//
// Launches mid_tile_kernel (and whatever experimental kernel a round adds below) exactly as the library does
// (launch_mid, host.hpp) on synthetic data of the C3 shape -- N x K float64 rows, P strided folds of N / P rows,
// random statistics -- with the ablation bits of MidArgs::dbg (built with -DCVM_MID_ABLATE) and prints the
// launch time per variant and, optionally, cycle stamps of sampled workgroups.  No result is checked here: the
// library's own tests do that; ablated runs are wrong by design.
//
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DCVM_MID_ABLATE tools/mid_probe.hip -o tools/mid_probe
//   tools/mid_probe [P=1000] [K=512] [N=100000] [reps=20]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <functional>
#include <map>
#include <mutex>
#include <queue>
#include <tuple>
#include <type_traits>
#include <vector>

#include "../include/cvmhip.h"

namespace {
#include "../cvmatrix_amd/csrc/geometry.hpp"
#include "../cvmatrix_amd/csrc/wgram_fallback.hpp"
#include "../cvmatrix_amd/csrc/wgram4.hpp"
#include "../cvmatrix_amd/csrc/finalize.hpp"
#include "../cvmatrix_amd/csrc/colstats.hpp"
#include "../cvmatrix_amd/csrc/small_folds.hpp"
#include "../cvmatrix_amd/csrc/mid_tile.hpp"
#include "experiments/mid_chain.hpp"
#include "experiments/mid128.hpp"
}  // namespace

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); }  \
  } while (0)

__global__ void symmetrize(double *G, int K) {       // (the library's G is exactly symmetric: the kernels rely on it)
  const int i = blockIdx.x, j = threadIdx.x + blockIdx.y * blockDim.x;
  if (j < K && j > i) G[(size_t)j * K + i] = G[(size_t)i * K + j];
}
__global__ void fill_rand(double *p, size_t n, unsigned seed, double lo, double hi) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned long long z = (i + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull;
    z ^= z >> 31; z *= 0xBF58476D1CE4E5B9ull; z ^= z >> 29; z *= 0x94D049BB133111EBull; z ^= z >> 32;
    p[i] = lo + (hi - lo) * (double)(z >> 11) * (1.0 / 9007199254740992.0);
  }
}

int main(int argc, char **argv) {
  const int P = argc > 1 ? atoi(argv[1]) : 1000;
  const int K = argc > 2 ? atoi(argv[2]) : 512;
  const int64_t N = argc > 3 ? atoll(argv[3]) : 100000;
  const int reps = argc > 4 ? atoi(argv[4]) : 20;
  const int M = 16;
  const int n = (int)(N / P);
  printf("mid_probe: N=%lld K=%d M=%d P=%d n=%d\n", (long long)N, K, M, P, n);
  double *X, *Y, *w, *G, *H, *fst, *oX, *oY;
  int64_t *idx, *offs;
  unsigned long long *stamps;
  const size_t fl = fstat_len(K, M);
  CK(hipMalloc(&X, N * K * 8)); CK(hipMalloc(&Y, N * M * 8)); CK(hipMalloc(&w, N * 8));
  CK(hipMalloc(&G, (size_t)K * K * 8)); CK(hipMalloc(&H, (size_t)K * M * 8));
  CK(hipMalloc(&fst, (size_t)P * fl * 8));
  CK(hipMalloc(&oX, (size_t)P * K * K * 8)); CK(hipMalloc(&oY, (size_t)P * K * M * 8));
  CK(hipMalloc(&idx, (size_t)P * n * 8)); CK(hipMalloc(&offs, (size_t)(P + 1) * 8));
  CK(hipMalloc(&stamps, 512 * 8 * 8));
  fill_rand<<<1024, 256>>>(X, (size_t)N * K, 1, 0.0, 1.0);
  fill_rand<<<256, 256>>>(Y, (size_t)N * M, 2, 0.0, 1.0);
  fill_rand<<<64, 256>>>(w, (size_t)N, 3, 0.1, 1.0);
  fill_rand<<<256, 256>>>(G, (size_t)K * K, 4, 0.0, 1.0);
  symmetrize<<<dim3(K, (K + 255) / 256), 256>>>(G, K);
  fill_rand<<<64, 256>>>(H, (size_t)K * M, 5, 0.0, 1.0);
  fill_rand<<<256, 256>>>(fst, (size_t)P * fl, 6, 0.5, 1.5);
  std::vector<int64_t> hidx((size_t)P * n), hoffs(P + 1);
  for (int f = 0; f < P; ++f) {
    hoffs[f] = (int64_t)f * n;
    for (int r = 0; r < n; ++r) hidx[(size_t)f * n + r] = (int64_t)f + (int64_t)r * P;      // the reference's folds: f, f + P, ...
  }
  hoffs[P] = (int64_t)P * n;
  CK(hipMemcpy(idx, hidx.data(), hidx.size() * 8, hipMemcpyHostToDevice));
  CK(hipMemcpy(offs, hoffs.data(), hoffs.size() * 8, hipMemcpyHostToDevice));
  CK(hipDeviceSynchronize());

  MidArgs m;
  memset(&m, 0, sizeof(m));
  m.X = X; m.Y = Y; m.w = w; m.idx = idx; m.offs = offs; m.seg0 = 0;
  m.fstats = fst; m.G = G; m.H = H; m.out_XTX = oX; m.out_XTY = oY;
  m.K = K; m.M = M; m.flags = 0x3F;
  m.nt = (K + 63) / 64; m.n_xtx = m.nt * (m.nt + 1) / 2; m.yextra = 0; m.ipf = m.n_xtx;
  m.n_items = (long long)P * m.ipf; m.per_xcd = (m.n_items + 7) / 8;
  m.maxn = (n + 15) / 16 * 16;
  m.nb = P;
  const size_t lds = mid_lds_bytes<double>(m.maxn);
  const dim3 grid((unsigned)(m.per_xcd * 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](int dbg, const char *label, bool stamp) {
    m.dbg = dbg; m.stamps = stamp ? stamps : nullptr;
    if (stamp) CK(hipMemset(stamps, 0, 512 * 8 * 8));
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((mid_tile_kernel<double, true>), grid, dim3(MID_THREADS), lds, 0, m);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((mid_tile_kernel<double, true>), grid, dim3(MID_THREADS), lds, 0, m);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("mid_tile dbg=%2d %-44s %8.4f ms\n", dbg, label, ms / reps);
    if (stamp) {
      std::vector<unsigned long long> h(512 * 8);
      CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
      double d[6] = {0, 0, 0, 0, 0, 0};
      int cnt = 0;
      for (int i = 0; i < 512; ++i) {
        const unsigned long long *s = &h[(size_t)i * 8];
        if (!s[0] || !s[6]) continue;       // (off-diagonal tiles only: they reach stamp 6)
        for (int k = 0; k < 6; ++k) d[k] += (double)(s[k + 1] - s[k]);
        ++cnt;
      }
      if (cnt)
        printf("   stamps (%d off-diagonal workgroups, mean cycles): row numbers %.0f | stats + first stage %.0f | loop %.0f | "
               "G preload + dump %.0f | direct finish %.0f | mirror %.0f | total %.0f\n",
               cnt, d[0] / cnt, d[1] / cnt, d[2] / cnt, d[3] / cnt, d[4] / cnt, d[5] / cnt,
               (d[0] + d[1] + d[2] + d[3] + d[4] + d[5]) / cnt);
      double e[5] = {0, 0, 0, 0, 0};
      int dc = 0;
      for (int i = 0; i < 512; ++i) {
        const unsigned long long *s = &h[(size_t)i * 8];
        if (!s[0] || !s[5] || s[6]) continue;       // (diagonal tiles: they stop at stamp 5)
        for (int k = 0; k < 5; ++k) e[k] += (double)(s[k + 1] - s[k]);
        ++dc;
      }
      if (dc)
        printf("   stamps (%d diagonal workgroups): row numbers %.0f | stats + first stage %.0f | loop %.0f | XTY piece + dump %.0f | "
               "direct finish %.0f | total %.0f\n", dc, e[0] / dc, e[1] / dc, e[2] / dc, e[3] / dc, e[4] / dc,
               (e[0] + e[1] + e[2] + e[3] + e[4]) / dc);
    }
  };
  // warm the device (clocks) before anything is read off
  for (int i = 0; i < 30; ++i) hipLaunchKernelGGL((mid_tile_kernel<double, true>), grid, dim3(MID_THREADS), lds, 0, m);
  CK(hipDeviceSynchronize());
  run(0, "as shipped", true);
#ifdef CVM_MID_PROLOGUE_ABL   // (with tools/experiments/mid_prologue_ablation.patch applied to mid_tile.hpp: bits 32 / 64 / 128)
  run(32, "row numbers computed, not loaded", true);
  run(96, "+ no statistics loads", true);
  run(224, "+ no weight gather", true);
  run(226, "+ no G loads", true);
  run(0, "as shipped (again)", true);
  return 0;
#endif
  run(1, "no output stores", false);
  run(2, "no G loads", false);
  run(3, "no stores, no G loads", false);
  run(4, "no LDS-DMA after the first stage", false);
  run(8, "no MFMA", false);
  run(12, "no LDS-DMA, no MFMA", false);
  run(7, "no stores, no G, no DMA (MFMA + finish VALU)", false);
  run(11, "no stores, no G, no MFMA (DMA + finish VALU)", false);
  run(16, "return after the loop", false);
  run(20, "loop only, no DMA", false);
  run(24, "loop only, no MFMA", false);
  run(0, "as shipped (again)", false);
  // occupancy sensitivity: the same kernel with dynamic LDS padded so that only 3 / 2 workgroups fit a CU
  for (size_t pad : {(size_t)52 * 1024, (size_t)78 * 1024}) {
    hipFuncSetAttribute((const void *)mid_tile_kernel<double, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)pad);
    m.dbg = 0; m.stamps = nullptr;
    for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((mid_tile_kernel<double, true>), grid, dim3(MID_THREADS), pad, 0, m);
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((mid_tile_kernel<double, true>), grid, dim3(MID_THREADS), pad, 0, m);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    CK(hipGetLastError());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("mid_tile with %zu KB of LDS per workgroup (%d workgroups per CU)  %8.4f ms\n", pad / 1024, (int)(160 * 1024 / pad), ms / reps);
  }
  // ---- mid_chain_kernel: the same tiles in chains of up to chmax per workgroup ------------------------------------
  double *rX, *rY;                               // mid_tile_kernel's outputs, for the bit-for-bit comparison
  CK(hipMalloc(&rX, (size_t)P * K * K * 8)); CK(hipMalloc(&rY, (size_t)P * K * M * 8));
  m.dbg = 0; m.stamps = nullptr;
  hipLaunchKernelGGL((mid_tile_kernel<double, true>), grid, dim3(MID_THREADS), lds, 0, m);
  CK(hipMemcpy(rX, oX, (size_t)P * K * K * 8, hipMemcpyDeviceToDevice));
  CK(hipMemcpy(rY, oY, (size_t)P * K * M * 8, hipMemcpyDeviceToDevice));
  std::vector<double> h0((size_t)4 * K * K), h1((size_t)4 * K * K);
  // ---- mid128_kernel: 128 x 128 items, eight computing waves, two workgroups per CU -----------------------------------
  {
    MidArgs c = m;
    c.nt = (K + 127) / 128;
    c.ipf = c.nt * (c.nt + 1) / 2;
    c.n_items = (long long)P * c.ipf; c.per_xcd = (c.n_items + 7) / 8;
    const dim3 cg((unsigned)(c.per_xcd * 8));
    CK(hipFuncSetAttribute((const void *)mid128_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)M128_LDS_BYTES));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)mid128_kernel<true>, M128_THREADS, M128_LDS_BYTES));
    printf("mid128: %zu bytes of LDS, %d workgroups per CU by the occupancy API, %d items per fold\n", (size_t)M128_LDS_BYTES, occ, c.ipf);
    auto run128 = [&](int dbg, const char *label) {
      c.dbg = dbg;
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((mid128_kernel<true>), cg, dim3(M128_THREADS), M128_LDS_BYTES, 0, c);
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((mid128_kernel<true>), cg, dim3(M128_THREADS), M128_LDS_BYTES, 0, c);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("mid128 dbg=%2d %-44s %8.4f ms\n", dbg, label, ms / reps);
    };
    CK(hipMemset(oX, 0xff, (size_t)P * K * K * 8)); CK(hipMemset(oY, 0xff, (size_t)P * K * M * 8));
    run128(0, "as built");
    {
      size_t bad = 0, tot = 0;
      const int fl[4] = {0, P / 2, P - 2, P - 1};
      for (int k = 0; k < 4; ++k) {
        CK(hipMemcpy(h0.data() + (size_t)k * K * K, rX + (size_t)fl[k] * K * K, (size_t)K * K * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h1.data() + (size_t)k * K * K, oX + (size_t)fl[k] * K * K, (size_t)K * K * 8, hipMemcpyDeviceToHost));
      }
      size_t first = (size_t)-1;
      for (size_t i = 0; i < h0.size(); ++i) { ++tot; if (memcmp(&h0[i], &h1[i], 8)) { if (first == (size_t)-1) first = i; ++bad; } }
      std::vector<double> y0((size_t)P * K * M), y1((size_t)P * K * M);
      CK(hipMemcpy(y0.data(), rY, y0.size() * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(y1.data(), oY, y1.size() * 8, hipMemcpyDeviceToHost));
      size_t bady = 0;
      for (size_t i = 0; i < y0.size(); ++i) if (memcmp(&y0[i], &y1[i], 8)) ++bady;
      printf("   against mid_tile_kernel: %zu of %zu XTX elements of 4 folds differ, %zu of %zu XTY elements\n", bad, tot, bady, y0.size());
      if (bad) {
        const size_t e = first % ((size_t)K * K);
        printf("   first difference: fold slot %zu row %zu col %zu: %.17g vs %.17g\n", first / ((size_t)K * K), e / K, e % K, h0[first], h1[first]);
      }
    }
#ifdef CVM_MID_ABLATE
    run128(1, "no output stores");
    run128(3, "no stores, no G loads");
    run128(4, "no LDS-DMA after the first stage");
    run128(8, "no MFMA");
    run128(16, "loops only");
#endif
  }
  for (int chmax : {1, 3}) {
    MidArgs c = m;
    c.chmax = chmax;
    c.ipf = chain_items_per_fold(c.nt, chmax);
    c.n_items = (long long)P * c.ipf; c.per_xcd = (c.n_items + 7) / 8;
    const size_t cl = chain_lds_bytes<double>(c.maxn);
    const dim3 cg((unsigned)(c.per_xcd * 8));
    auto crun = [&](int dbg, const char *label, bool stamp) {
      c.dbg = dbg; c.stamps = stamp ? stamps : nullptr;
      if (stamp) CK(hipMemset(stamps, 0, 512 * 8 * 8));
      for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((mid_chain_kernel<double, true>), cg, dim3(MID_THREADS), cl, 0, c);
      CK(hipEventRecord(e0, 0));
      for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((mid_chain_kernel<double, true>), cg, dim3(MID_THREADS), cl, 0, c);
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      printf("mid_chain chmax=%d dbg=%2d %-34s %8.4f ms  (lds %zu, %d chains per fold)\n", chmax, dbg, label, ms / reps, cl, c.ipf);
      if (stamp) {
        std::vector<unsigned long long> h(512 * 8);
        CK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
        double d[6] = {0, 0, 0, 0, 0, 0};
        int cnt = 0;
        for (int i = 0; i < 512; ++i) {
          const unsigned long long *s = &h[(size_t)i * 8];
          if (!s[0] || !s[6] || !s[5] || !s[4]) continue;
          for (int k = 0; k < 6; ++k) d[k] += (double)(s[k + 1] - s[k]);
          ++cnt;
        }
        if (cnt)
          printf("   stamps of a chain's FIRST tile (%d workgroups, mean cycles): row numbers %.0f | stats + first stage %.0f | loop %.0f | "
                 "dump of half 0 %.0f | direct half 0 %.0f | mirror half 0 %.0f\n",
                 cnt, d[0] / cnt, d[1] / cnt, d[2] / cnt, d[3] / cnt, d[4] / cnt, d[5] / cnt);
      }
    };
    CK(hipMemset(oX, 0xff, (size_t)P * K * K * 8)); CK(hipMemset(oY, 0xff, (size_t)P * K * M * 8));
    crun(0, "as built", true);
    // bit-for-bit against mid_tile_kernel: the first, a middle and the last two folds
    {
      size_t bad = 0, tot = 0;
      const int fl[4] = {0, P / 2, P - 2, P - 1};
      for (int k = 0; k < 4; ++k) {
        CK(hipMemcpy(h0.data() + (size_t)k * K * K, rX + (size_t)fl[k] * K * K, (size_t)K * K * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h1.data() + (size_t)k * K * K, oX + (size_t)fl[k] * K * K, (size_t)K * K * 8, hipMemcpyDeviceToHost));
      }
      for (size_t i = 0; i < h0.size(); ++i) { ++tot; if (memcmp(&h0[i], &h1[i], 8)) ++bad; }
      std::vector<double> y0((size_t)P * K * M), y1((size_t)P * K * M);
      CK(hipMemcpy(y0.data(), rY, y0.size() * 8, hipMemcpyDeviceToHost));
      CK(hipMemcpy(y1.data(), oY, y1.size() * 8, hipMemcpyDeviceToHost));
      size_t bady = 0;
      for (size_t i = 0; i < y0.size(); ++i) if (memcmp(&y0[i], &y1[i], 8)) ++bady;
      printf("   against mid_tile_kernel: %zu of %zu XTX elements of 4 folds differ, %zu of %zu XTY elements\n", bad, tot, bady, y0.size());
    }
    if (chmax == 3 || chmax == 4) {
      crun(1, "no output stores", false);
      crun(3, "no stores, no G loads", false);
      crun(4, "no LDS-DMA after the first stage", false);
      crun(8, "no MFMA", false);
      crun(16, "loops only", false);
    }
  }
  return 0;
}
