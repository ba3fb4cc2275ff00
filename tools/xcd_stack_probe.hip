// Does it matter WHICH XCD writes WHICH part of memory?  hipcc --offload-arch=gfx950 -O3 -o tools/xcd_stack_probe tools/xcd_stack_probe.hip
// Workgroup b of a 1-D launch runs on XCD b % 8 (tools/dispatch_probe.hip).  A plain linear fill with one workgroup per
// 4 KiB therefore pairs XCD x with the 4-KiB blocks x, x + 8, x + 16, ... of the buffer -- and it is the only store
// pattern of this project that reaches 6.7-7 TB/s; everything that spreads an XCD's stores over all of memory ends at
// 4.7-5.3.  This probe rotates that pairing:
//   rot g s    the workgroup on XCD x writes block slot (x + s) % 8 of its group of 8 slots, a slot being g * 4 KiB
//   pair x s   only the workgroups of XCD x work, and write the blocks of slot s: one XCD against one eighth of memory
//   tile ...   64-row x 256-byte tiles with 16-KiB row pitch (float32 K = 4096): XCD by dispatch order against XCD chosen
//              from the tile's column (address bits 12..14)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float v4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void rot_fill(v4 *p, unsigned g, unsigned s) {
  const unsigned b = blockIdx.x, x = b & 7, j = (b >> 3) % g, grp = b / (8 * g);
  const size_t blk = (size_t)grp * 8 * g + (size_t)((x + s) & 7) * g + j;
  __builtin_nontemporal_store((v4){1, 2, 3, 4}, p + blk * 256 + threadIdx.x);
}
__global__ __launch_bounds__(256) void pair_fill(v4 *p, unsigned x, unsigned s, unsigned rep) {
  const unsigned b = blockIdx.x;
  if ((b & 7) != x) return;
  // `rep` consecutive groups per workgroup: enough bytes per launch without more empty workgroups
  for (unsigned r = 0; r < rep; ++r) {
    const size_t blk = ((size_t)(b >> 3) * rep + r) * 8 + s;
    __builtin_nontemporal_store((v4){1, 2, 3, 4}, p + blk * 256 + threadIdx.x);
  }
}
// tiles of 64 rows x 256 B, row pitch 16 KiB, matrices of 64 MiB: item -> (matrix, tile row, tile column)
//   mode 0: item = b (dispatch order: XCD = b % 8, neighbours in a row on different XCDs)
//   mode 1: XCD-contiguous ranges (the library's order): item = (b % 8) * per + b / 8
//   mode 2: the XCD follows the tile's column: 4-KiB address block of a row segment = 4 r + tc / 16 -> slot (4 r + tc / 16) % 8;
//           even rows of tile column tc go to slot tc / 16 % 8 (tc / 16 in 0..3), odd rows to that + 4: a workgroup on XCD x
//           takes the rows of ONE parity: 32 rows of two tiles (tc and tc + 1... see below)
__global__ __launch_bounds__(256) void tile_fill(v4 *p, unsigned mode, unsigned shift, unsigned n_items) {
  const unsigned b = blockIdx.x;
  unsigned item = b;
  if (mode == 1) { const unsigned per = (n_items + 7) / 8; item = (b & 7) * per + (b >> 3); if (item >= n_items) return; }
  const unsigned lane16 = threadIdx.x & 15, r0 = threadIdx.x >> 4;       // 16 lanes x 16 B = one 256-B segment; 16 rows per pass
  if (mode <= 1) {
    const unsigned mat = item / 4096, t = item % 4096, tr = t / 64, tc = t % 64;
    v4 *base = p + (size_t)mat * (64u << 20) / 16 + (size_t)tr * 64 * 1024 + tc * 16;
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) __builtin_nontemporal_store((v4){1, 2, 3, 4}, base + (size_t)(r0 + 16 * k) * 1024 + lane16);
  } else {
    // an item is 32 rows of one parity of TWO neighbouring tile rows' worth?  Keep it simple: the item covers rows of
    // parity q of a 128-row x 256-byte strip (64 rows written), all of them in slot (tc / 16 + 4 q) % 8 =: sl; the
    // workgroup must sit on XCD (sl + shift) % 8: b % 8 = x -> choose (q, tc / 16) from x, the rest from b / 8
    const unsigned x = b & 7, sl = (x + 8 - shift) & 7, q = sl >> 2, c4 = sl & 3;      // tc / 16 = c4
    const unsigned rest = b >> 3;                        // (matrix, strip of 128 rows, tc % 16)
    const unsigned tcl = rest % 16, strip = (rest / 16) % 32, mat = rest / 512;
    const unsigned tc = c4 * 16 + tcl;
    v4 *base = p + (size_t)mat * (64u << 20) / 16 + (size_t)strip * 128 * 1024 + tc * 16;
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) __builtin_nontemporal_store((v4){1, 2, 3, 4}, base + (size_t)(2 * (r0 + 16 * k) + q) * 1024 + lane16);
  }
}

// the same tiles, `nf` of them per workgroup one after the other (XCD-contiguous ranges of items like the library):
//   along = 0: the same tile in nf consecutive matrices (64 MiB apart) -- what a workgroup that keeps G in registers does
//   along = 1: nf neighbouring tiles of one tile row of one matrix
//   spin: cycles of s_sleep-free arithmetic between two tiles (a stand-in for the per-fold work)
__global__ __launch_bounds__(256) void tile_loop_fill(v4 *p, unsigned nf, unsigned along, unsigned n_items, unsigned spin) {
  const unsigned b = blockIdx.x, per = (n_items + 7) / 8;
  const unsigned item = (b & 7) * per + (b >> 3);
  if (item >= n_items) return;
  const unsigned lane16 = threadIdx.x & 15, r0 = threadIdx.x >> 4;
  float acc = (float)threadIdx.x;
  for (unsigned f = 0; f < nf; ++f) {
    unsigned mat, tr, tc;
    if (along == 0) { const unsigned grp = item / 4096, t = item % 4096; mat = grp * nf + f; tr = t / 64; tc = t % 64; }
    else { const unsigned t = item * nf + f; mat = t / 4096; tr = (t % 4096) / 64; tc = t % 64; }
    v4 *base = p + (size_t)mat * (64u << 20) / 16 + (size_t)tr * 64 * 1024 + tc * 16;
    for (unsigned k = 0; k < spin; ++k) acc = acc * 1.0001f + 0.5f;
    const v4 v = {acc, 2, 3, 4};
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) __builtin_nontemporal_store(v, base + (size_t)(r0 + 16 * k) * 1024 + lane16);
  }
}

// A stand-in for small_apply_kernel's store behaviour (float32, K = 4096): an upper-triangle tile (tr <= tc) of nf matrices
// one after the other, per matrix 4 direct + 4 mirrored 16-byte stores per thread (32 KiB per workgroup), `sleep` x 64 cycles
// of s_sleep between two matrices (the per-fold work) and, drain = 1, `s_waitcnt vmcnt(0)` before the next matrix's stores
// (what the compiler puts at the top of the fold loop: the rows of the next fold were loaded before the stores, and a wait
// for them is a wait for everything).  Dynamic LDS sets the number of workgroups a CU holds.
template <int SLEEP> __global__ __launch_bounds__(256) void apply_like_fill(v4 *p, unsigned nf, unsigned n_items, unsigned drain) {
  extern __shared__ char pad[];
  const unsigned b = blockIdx.x, per = (n_items + 7) / 8;
  const unsigned item = (b & 7) * per + (b >> 3);
  if (item >= n_items) return;
  const unsigned lane16 = threadIdx.x & 15, r0 = threadIdx.x >> 4;
  const unsigned grp = item / 2080;
  unsigned t = item % 2080, tr = 0;
  while (t >= 64 - tr) { t -= 64 - tr; ++tr; }
  const unsigned tc = tr + t;
  if (threadIdx.x == 0) pad[0] = 1;
  for (unsigned f = 0; f < nf; ++f) {
    v4 *m = p + (size_t)(grp * nf + f) * (64u << 20) / 16;
    v4 *d = m + (size_t)tr * 64 * 1024 + tc * 16, *mi = m + (size_t)tc * 64 * 1024 + tr * 16;
    if (drain) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
    const v4 v = {(float)f, 2, 3, 4};
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) __builtin_nontemporal_store(v, d + (size_t)(r0 + 16 * k) * 1024 + lane16);
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) __builtin_nontemporal_store(v, mi + (size_t)(r0 + 16 * k) * 1024 + lane16);
  }
}

// ONE fold per workgroup: upper-triangle tile (tr, tc) of matrix f, direct + mirrored, optionally behind a read of the tile of
// a 49th matrix ("G", added to the stored values).  order 0: fold-major (all tiles of matrix 0, then matrix 1, ...);
// order g > 0: tile-major in groups of g matrices (the g folds of a tile are neighbours in an XCD's range: G from L2)
__global__ __launch_bounds__(256) void oneshot_fill(v4 *p, const v4 *G, unsigned order, unsigned nmat, unsigned readg) {
  const unsigned n_items = nmat * 2080, b = blockIdx.x, per = (n_items + 7) / 8;
  const unsigned item = (b & 7) * per + (b >> 3);
  if (item >= n_items) return;
  unsigned f, t;
  if (order == 0) { f = item / 2080; t = item % 2080; }
  else { const unsigned grp = item / (2080 * order), r = item % (2080 * order); t = r / order; f = grp * order + r % order; }
  unsigned tr = 0;
  while (t >= 64 - tr) { t -= 64 - tr; ++tr; }
  const unsigned tc = tr + t;
  const unsigned lane16 = threadIdx.x & 15, r0 = threadIdx.x >> 4;
  v4 *m = p + (size_t)f * (64u << 20) / 16;
  v4 *d = m + (size_t)tr * 64 * 1024 + tc * 16, *mi = m + (size_t)tc * 64 * 1024 + tr * 16;
  v4 v[4];
#pragma unroll
  for (unsigned k = 0; k < 4; ++k) v[k] = (v4){(float)f, 2, 3, 4};
  if (readg) {
    const v4 *g = G + (size_t)tr * 64 * 1024 + tc * 16;
#pragma unroll
    for (unsigned k = 0; k < 4; ++k) v[k] += g[(size_t)(r0 + 16 * k) * 1024 + lane16];
  }
#pragma unroll
  for (unsigned k = 0; k < 4; ++k) __builtin_nontemporal_store(v[k], d + (size_t)(r0 + 16 * k) * 1024 + lane16);
#pragma unroll
  for (unsigned k = 0; k < 4; ++k) __builtin_nontemporal_store(v[k], mi + (size_t)(r0 + 16 * k) * 1024 + lane16);
}

// apply-like with other tile shapes: square tiles of R x R elements of ES bytes of K = 4096 matrices (upper triangle, direct + mirrored,
// nf matrices per workgroup one after the other, XCD-contiguous ranges)
template <int R, int ES> __global__ __launch_bounds__(256) void apply_shape_fill(v4 *p, unsigned nf, unsigned n_items) {
  constexpr unsigned T = 4096 / R, NT = T * (T + 1) / 2;          // tiles per row, upper-triangle tiles
  constexpr unsigned SEG16 = R * ES / 16;                          // 16-byte pieces per row segment
  constexpr unsigned ROWS_PASS = 256 / SEG16, PASSES = R / ROWS_PASS;
  constexpr size_t PITCH16 = (size_t)4096 * ES / 16, MAT16 = PITCH16 * 4096;
  const unsigned b = blockIdx.x, per = (n_items + 7) / 8;
  const unsigned item = (b & 7) * per + (b >> 3);
  if (item >= n_items) return;
  const unsigned grp = item / NT;
  unsigned t = item % NT, tr = 0;
  while (t >= T - tr) { t -= T - tr; ++tr; }
  const unsigned tc = tr + t;
  const unsigned lane = threadIdx.x % SEG16, r0 = threadIdx.x / SEG16;
  for (unsigned f = 0; f < nf; ++f) {
    v4 *m = p + (size_t)(grp * nf + f) * MAT16;
    v4 *d = m + (size_t)tr * R * PITCH16 + tc * SEG16, *mi = m + (size_t)tc * R * PITCH16 + tr * SEG16;
    const v4 v = {(float)f, 2, 3, 4};
#pragma unroll
    for (unsigned k = 0; k < PASSES; ++k) __builtin_nontemporal_store(v, d + (size_t)(r0 + ROWS_PASS * k) * PITCH16 + lane);
#pragma unroll
    for (unsigned k = 0; k < PASSES; ++k) __builtin_nontemporal_store(v, mi + (size_t)(r0 + ROWS_PASS * k) * PITCH16 + lane);
  }
}
template <int R, int ES> void run_shape(v4 *p, const char *name);
template <typename F> float timeit(F f, int rep = 10) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) f();
  hipEventRecord(a);
  for (int i = 0; i < rep; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  hipEventDestroy(a); hipEventDestroy(b);
  return ms / rep;
}

template <int R, int ES> void run_shape(v4 *p, const char *name) {
  constexpr unsigned T = 4096 / R, NT = T * (T + 1) / 2;
  const unsigned nmat = ES == 4 ? 48 : 24;
  printf("apply-like, %-34s 1 / 2 / 4 / 8 folds per workgroup:", name);
  for (unsigned nf : {1u, 2u, 4u, 8u}) {
    const unsigned items = nmat / nf * NT;
    const float ms = timeit([&] { hipLaunchKernelGGL((apply_shape_fill<R, ES>), dim3(8 * ((items + 7) / 8)), dim3(256), 0, 0, p, nf, items); });
    printf(" %5.2f", (double)items * nf * 2 * R * R * ES / ms / 1e9);
  }
  printf("  TB/s\n");
}
int main(int argc, char **argv) {
  const size_t bytes = (size_t)3 << 30;
  v4 *p; if (hipMalloc(&p, bytes) != hipSuccess) return 1;
  printf("buffer %p (%zu MiB)\n", (void *)p, bytes >> 20);
  const unsigned nblk = (unsigned)(bytes / 4096);
  for (unsigned g : {1u, 2u, 4u, 16u, 64u, 512u}) {
    printf("rot  slot = %4u KiB:", 4 * g);
    for (unsigned s = 0; s < 8; ++s) {
      const float ms = timeit([&] { hipLaunchKernelGGL(rot_fill, dim3(nblk), dim3(256), 0, 0, p, g, s); });
      printf(" %5.2f", bytes / ms / 1e9);
    }
    printf("   TB/s for s = 0..7\n");
  }
  printf("pair (XCD x alone -> slot s of 4 KiB), GB/s:\n");
  for (unsigned x = 0; x < 8; ++x) {
    printf("  x=%u:", x);
    for (unsigned s = 0; s < 8; ++s) {
      const unsigned rep = 4, grid = nblk / rep;        // grid / 8 active workgroups x rep blocks x 4 KiB
      const float ms = timeit([&] { hipLaunchKernelGGL(pair_fill, dim3(grid), dim3(256), 0, 0, p, x, s, rep); }, 5);
      printf(" %6.0f", (double)(grid / 8) * rep * 4096 / ms / 1e6);
    }
    printf("\n");
  }
  const unsigned n_items = 48 * 4096;
  for (unsigned mode = 0; mode < 2; ++mode) {
    const float ms = timeit([&] { hipLaunchKernelGGL(tile_fill, dim3(8 * ((n_items + 7) / 8)), dim3(256), 0, 0, p, mode, 0u, n_items); });
    printf("tile mode %u: %5.2f TB/s\n", mode, (double)n_items * 16384 / ms / 1e9);
  }
  printf("tile mode 2 (XCD from the column), shift 0..7:");
  for (unsigned s = 0; s < 8; ++s) {
    const float ms = timeit([&] { hipLaunchKernelGGL(tile_fill, dim3(n_items), dim3(256), 0, 0, p, 2u, s, n_items); });
    printf(" %5.2f", (double)n_items * 16384 / ms / 1e9);
  }
  printf("  TB/s\n");
  for (unsigned along = 0; along < 2; ++along)
    for (unsigned spin : {0u, 300u}) {
      printf("tile loop, %s, spin %3u, 1/2/4/8/16 tiles per workgroup:", along ? "neighbouring tiles  " : "same tile, 64 MiB on", spin);
      for (unsigned nf : {1u, 2u, 4u, 8u, 16u}) {
        const unsigned items = n_items / nf;
        const float ms = timeit([&] { hipLaunchKernelGGL(tile_loop_fill, dim3(8 * ((items + 7) / 8)), dim3(256), 0, 0, p, nf, along, items, spin); });
        printf(" %5.2f", (double)n_items * 16384 / ms / 1e9);
      }
      printf("  TB/s\n");
    }
  {
    const unsigned nfa = 8, items = 6 * 2080;           // 48 matrices, upper triangle: 6 groups x 2080 tiles
    const double by = (double)items * nfa * 32768;
    hipFuncSetAttribute((const void *)apply_like_fill<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    hipFuncSetAttribute((const void *)apply_like_fill<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    hipFuncSetAttribute((const void *)apply_like_fill<48>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 256);
    for (unsigned wg : {8u, 5u, 4u, 3u, 2u}) {
      const size_t lds = wg == 8 ? 16 : (size_t)(160 * 1024 / wg - 512);
      printf("apply-like, %u workgroups per CU: ", wg);
      for (unsigned drain = 0; drain < 2; ++drain) {
        float m0 = timeit([&] { hipLaunchKernelGGL(apply_like_fill<0>, dim3(8 * ((items + 7) / 8)), dim3(256), lds, 0, p, nfa, items, drain); });
        float m1 = timeit([&] { hipLaunchKernelGGL(apply_like_fill<16>, dim3(8 * ((items + 7) / 8)), dim3(256), lds, 0, p, nfa, items, drain); });
        float m2 = timeit([&] { hipLaunchKernelGGL(apply_like_fill<48>, dim3(8 * ((items + 7) / 8)), dim3(256), lds, 0, p, nfa, items, drain); });
        printf(" %s sleep 0 / 1k / 3k cycles: %5.2f %5.2f %5.2f ", drain ? "| drain," : "no drain,", by / m0 / 1e9, by / m1 / 1e9, by / m2 / 1e9);
      }
      printf(" TB/s\n");
    }
  }
  {
    const unsigned nmat = 47;                              // matrix 47 of the buffer plays G
    const v4 *G = p + (size_t)47 * (64u << 20) / 16;
    const unsigned n_items = nmat * 2080;
    const double by = (double)n_items * 32768;
    for (unsigned readg = 0; readg < 2; ++readg) {
      printf("one fold per workgroup, %s: order fold-major / tile-major in groups of 2, 4, 8, 16, 47:", readg ? "G read   " : "no G read");
      for (unsigned order : {0u, 2u, 4u, 8u, 16u, 47u}) {
        const unsigned nm = order ? nmat / order * order : nmat;
        const float ms = timeit([&] { hipLaunchKernelGGL(oneshot_fill, dim3(8 * ((nm * 2080 + 7) / 8)), dim3(256), 0, 0, p, G, order, nm, readg); });
        printf(" %5.2f", (double)nm * 2080 * 32768 / ms / 1e9);
      }
      printf("  TB/s of stores\n");
    }
    (void)by;
  }
  run_shape<64, 4>(p, "float32 64 x 64 (256-B segments)");
  run_shape<128, 4>(p, "float32 128 x 128 (512-B segments)");
  run_shape<256, 4>(p, "float32 256 x 256 (1-KiB segments)");
  run_shape<64, 8>(p, "float64 64 x 64 (512-B segments)");
  run_shape<128, 8>(p, "float64 128 x 128 (1-KiB segments)");
  hipFree(p);
  return 0;
}
