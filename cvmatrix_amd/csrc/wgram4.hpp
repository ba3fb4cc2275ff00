// wgram4.hpp -- part of libcvmhip.so (included by cvmhip.hip inside its anonymous namespace).
// wgram4_kernel<T>: the LDS-DMA Gram kernel (4 compute + 4 loader waves) for float64 and float32,
// with the fused single-split epilogue (float64).
#pragma once

// ----------------------------------------------------------------------------------
// wgram4_kernel: the fast path (16-byte aligned rows; float64: even M; float32: K % 4 == 0).
//
// Same work decomposition, LDS stage image and partial layout as wgram_kernel, but the
// eight waves of a workgroup (one workgroup per CU) are specialised:
//   waves 0-3  COMPUTE, one per SIMD.  They never touch global memory inside the loop, so no
//              vector-memory instruction ever blocks their issue (a 1 KiB load costs its
//              wave 200-450 cycles of issue on a busy CU: tools/dma_issue.hip).
//              Off-diagonal tile (wgram4_body): wave (wr,wc) owns the 64x64 block (wr,wc) of
//              the 128x128 tile: 4x4 MFMA tiles, 16 accumulators (128 VGPRs), 16 MFMAs per 8
//              LDS fragment reads.
//              Diagonal tile, first Y chunk (wgram4_diag_body): the upper triangle of the
//              tile's 8x8 grid of MFMA tiles shared out evenly, 9 tiles + the XTY tiles and
//              the column sums of two row-tiles per wave: 11 MFMAs per k-step.
//              XTY-only items (further Y chunks, or calls that want no XTX) keep the older
//              roles of wgram4_body: wave 2 computes panel_i^T W Y[:, 32c..32c+32) (8x2 MFMA
//              tiles), waves 0 and 3 sum the X columns, wave 1 (panel 0) the Y columns.
//   waves 4-7  LOADERS.  Loader d owns stage rows d, d+4, d+8, d+12 and moves, per row, the
//              X panel rows, the Y tile row (diagonal tiles) and the weight global -> LDS
//              by LDS-DMA (global_load_lds: one wave instruction = one 1 KiB panel row,
//              gathered by row number; rows past the end read a zero line).  They run
//              THREE stages ahead of the compute waves through a ring of four LDS stage
//              buffers behind a hand-counted s_waitcnt vmcnt, and contain no VALU
//              instruction at all (see the loader section for why).
// One s_barrier per 16-row stage joins all eight waves.
// ----------------------------------------------------------------------------------
constexpr int NT4 = 512;
// Partial tiles go to the workspace once and are read once, by another kernel: nontemporal stores (experiment
// CVM_PARTIAL_NT; 0 = plain stores, whose dirty lines wait in the L2s for the end-of-kernel write-back)
#ifndef CVM_PARTIAL_NT
#define CVM_PARTIAL_NT 0
#endif
template <typename T> __device__ __forceinline__ void partial_store(T *p, T v) {
#if CVM_PARTIAL_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}
#ifndef CVM_INTERLEAVE
#define CVM_INTERLEAVE 1
#endif
// Shape of a compute wave's block of a plain off-diagonal tile (round 6).  64 x 64 (4 x 4 MFMA tiles: 4 A-side and
// 4 B-side fragments per k-step) was the shape of rounds 1-5; 32 x 128 (2 x 8: wave w owns rows 32 w .. 32 w + 31 of
// the tile and all of its columns) reads 10 fragments instead of 8 but WEIGHTS only 2 instead of 4 -- the A side is
// the weighted one, and every element of the A panel is then multiplied by exactly one wave, the minimum.
// tools/f32_loop_probe.hip: a vector instruction inside the MFMA stream costs the matrix pipe ~12 cycles in float32
// (the float32 MFMA runs on the vector unit's own multipliers: the two are one resource), the 16 weighting
// multiplies of a stage 8 % of the loop; with 8 of them 0.862 -> 0.890 of the peak in the probe.  Per element type:
#ifndef CVM_WIDE_F32
#define CVM_WIDE_F32 1
#endif
#ifndef CVM_WIDE_F64
#define CVM_WIDE_F64 1
#endif
constexpr int NBUF4 = 4;        // LDS stage buffers
constexpr size_t LDS4_BYTES = (size_t)NBUF4 * BUF_ELEMS * 8;   // float64; float32 uses half of it
template <typename T> constexpr size_t lds4_bytes() { return (size_t)NBUF4 * BUF_ELEMS * sizeof(T); }

// The body is instantiated once per wave role and kept out of line: inlined together, the
// register allocator has to give all roles one common assignment of the 128 accumulator
// registers and spills hundreds of values; as separate functions every role fits.
//   ROLER 0/1/2: compute wave without sums / with X column sums / with Y column sums
//   ROLER 3: loader wave
__host__ __device__ inline size_t fstat_len(int K, int M);
template <typename T> struct FusedPre;                                          // finalize.hpp
template <typename T>
__device__ __forceinline__ void fused_g_preload32(FusedPre<T> &p, int a0, int b0, int K, const T *Gt, int lane, int row_lo);
template <typename T, int TP>
__device__ __forceinline__ void fused_finish_direct32(T (*Ts)[TP], const double *rs, int a0, int b0, int K, T *out,
                                                      double swt, bool cX, bool sX, int lane, int row_lo,
                                                      const FusedPre<T> &p);
template <typename T, int TP>
__device__ __forceinline__ void fused_finish_block(T (*Ts)[TP], const double *rs, bool diagb, int a0,
                                                   int b0, int K, const T *Gt, T *out,
                                                   double swt, bool cX, bool sX, int lane);
template <typename T, int TP>
__device__ __forceinline__ void fused_finish_direct(T (*Ts)[TP], const double *rs, bool diagb, int a0,
                                                    int b0, int K, const T *Gt, T *out,
                                                    double swt, bool cX, bool sX, int lane, int row_lo, int row_hi);
template <typename T, int TP>
__device__ __forceinline__ void fused_finish_mirror(T (*Ts)[TP], int a0, int b0, int K, T *out, int lane,
                                                    int row_lo, int row_hi);
// LDS of the fused epilogue (it reuses the stage ring): per compute wave a 64x64 block in the
// accumulators' type (pitch 65) + 256 float64 row/column means and reciprocal stds; a diagonal
// tile: one 128x128 image (pitch 129) + three such statistics blocks
template <typename T> constexpr size_t wave_tile_bytes() { return ((size_t)64 * 65 * sizeof(T) + 7) / 8 * 8; }
template <typename T> constexpr size_t wave_lds_bytes() { return wave_tile_bytes<T>() + 256 * 8; }
template <typename T> constexpr size_t diag_tile_bytes() { return ((size_t)TILE * (TILE + 1) * sizeof(T) + 7) / 8 * 8; }
template <typename T> constexpr size_t fused_lds_bytes() {
  // (a diagonal item that forms its statistics itself keeps 324 more float64 behind its three blocks)
  return 4 * wave_lds_bytes<T>() > diag_tile_bytes<T>() + (3 * 256 + 328) * 8 ? 4 * wave_lds_bytes<T>()
                                                                             : diag_tile_bytes<T>() + (3 * 256 + 328) * 8;
}
// a fused launch's dynamic LDS: the stage ring or the epilogue's images, whichever is larger, and 16 bytes
// behind them that neither touches (the mark of fused_wait_flag)
template <typename T> constexpr size_t fused_launch_lds_bytes() {
  return (fused_lds_bytes<T>() > lds4_bytes<T>() ? fused_lds_bytes<T>() : lds4_bytes<T>()) + 16;
}

// Fused epilogue of a DIAGONAL 128x128 tile, shared by all eight waves of the workgroup (the four
// loader waves are idle by then): the tile's raw update is in Td (LDS), the statistics of block b in
// rs_b = rs0 + 256 b.  Blocks: 0 = (0,0) and 2 = (1,1) on the diagonal, 1 = (0,1) off it (stored
// mirrored too).  Phase 0: twelve units of sixteen rows (block u / 4, rows 16 (u % 4) ..), wave v
// takes units v and v + 8; phase 1 (after a barrier): the mirrored store of block 1, waves 0..3.
// (a unit = sixteen rows)
template <typename T, int TP>
__device__ __forceinline__ void diag_tile_finish(T (*Td)[TP], double *rs0, int v, int phase, int ti, int K,
                                                 const T *Gt, T *out, double swt, bool cX, bool sX,
                                                 int lane) {
  if (phase == 0) {
    for (int u = v; u < 12; u += 8) {
      const int b = u >> 2, si = b == 2, sj = b >= 1;
      const int a0 = ti * TILE + 64 * si, b0 = ti * TILE + 64 * sj;
      if (a0 >= K || b0 >= K) continue;
      fused_finish_direct<T, TP>(reinterpret_cast<T (*)[TP]>(&Td[64 * si][64 * sj]), rs0 + 256 * b, si == sj, a0,
                                 b0, K, Gt, out, swt, cX, sX, lane, (u & 3) * 16, (u & 3) * 16 + 16);
    }
  } else if (v < 4) {
    const int a0 = ti * TILE, b0 = ti * TILE + 64;
    if (b0 < K)
      fused_finish_mirror<T, TP>(reinterpret_cast<T (*)[TP]>(&Td[0][64]), a0, b0, K, out, lane, 16 * v, 16 * v + 16);
  }
}

// (the role functions are called once per work item from the persistent loop of wgram4_kernel)
#define ROLE_EXIT() return
#define ROLE_ATTR
// Statistics formed inside a fused launch (WgramArgs::stat_flags).  fused_wait_flag: ONE lane of the first
// loader wave waits until the diagonal items of its two panels have published.  They precede the waiter in its
// own list (fold-major lists, geometry.hpp) and wait for nobody, so the flags are normally up long before they
// are asked for and the wait is bounded by one item's duration whatever else shares the device.  The spin limit
// is the safety net under that argument (HIP promises no dispatch order): a wait that gives up leaves mark 0 in
// the last words of the dynamic LDS -- every wave of the item then writes NOTHING -- and appends the item's list
// position to WgramArgs::retry_items; the host's second launch of this kernel over that list (retry_mode)
// recomputes the item when every flag of the first launch is up.  Giving up THERE (it cannot) is counted in
// fused_status[1] and mark 2 poisons the item's results with NaN; cvm_fold_update_ex reports either through
// *status.  Marks: 1 = go, 0 = skip (recomputed later), 2 = poison.
constexpr int FUSED_GO = 1, FUSED_SKIP = 0, FUSED_POISON = 2;
template <typename T>
__device__ __forceinline__ void fused_wait_flag(const int *f0, const int *f1, char *smem_raw, int lane, int seg,
                                                int xcd_q, int slot_q, int *fused_status,
                                                unsigned long long *retry_items, int retry_mode, int test_mode) {
  if (lane != 0) return;
  int ok = 1;
  const long limit = test_mode == 2 ? 4 : (1L << 22);
  const bool feign = (test_mode == 1 || test_mode == 3) && seg % 3 == 0 && (!retry_mode || test_mode == 3);
  for (int w = 0; w < 2 && ok; ++w) {
    const int *f = w ? f1 : f0;
    long spins = 0;
    while (feign || __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
      __builtin_amdgcn_s_sleep(2);
      if (feign || ++spins > limit) { ok = 0; break; }
    }
  }
  int mark = FUSED_GO;
  if (!ok) {
    if (!retry_mode && fused_status && retry_items) {
      const int pos = __hip_atomic_fetch_add(fused_status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      retry_items[pos] = ((unsigned long long)(unsigned)xcd_q << 32) | (unsigned)slot_q;
      mark = FUSED_SKIP;
    } else {
      if (fused_status) __hip_atomic_fetch_add(fused_status + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      mark = FUSED_POISON;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the count is at the memory before this workgroup can leave)
  }
  *reinterpret_cast<volatile int *>(smem_raw + fused_launch_lds_bytes<T>() - 16) = mark;
}
// the mark of this item (off-diagonal items of a launch that forms its statistics; every wave, behind the barrier
// that follows fused_wait_flag)
template <typename T> __device__ __forceinline__ int fused_mark(const int *stat_flags, const char *smem_raw) {
  if (!stat_flags) return FUSED_GO;
  return uni(*reinterpret_cast<const volatile int *>(smem_raw + fused_launch_lds_bytes<T>() - 16));
}

template <typename T, bool WEIGHTED, bool GATHER, bool HWR, bool MFMR, int ROLER, bool FUSEDR = false>
__device__ __noinline__ ROLE_ATTR void wgram4_body(kargs_ptr<T> kargs, int xcd_q, int slot_q) {
  typedef typename MF<T>::acc_t acc_t;
#ifdef CVM_STAMPS
  const unsigned long long c_entry = __builtin_amdgcn_s_memtime();
#endif
  const WgramArgs<T> a = kernel_args<T>(kargs);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T *smem = reinterpret_cast<T *>(smem_raw);
  const Geom &g = a.g;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wave_all & 3;

  Item wi;
  if (!decode_slot(a, uni(xcd_q), (long)uni(slot_q), wi)) ROLE_EXIT();
  const long u = wi.u;
  const int it = wi.it, seg = wi.seg, sp = wi.sp, ti = wi.ti, tj = wi.tj, yc = wi.yc;
  const bool diag = (ti == tj);
  const int wr = wave >> 1, wc = wave & 1;
  const bool h_wave = diag && wave == 2;
  const bool do_g = !g.diag_only && yc == 0;

  int64_t seg_begin, seg_rows;
  if (a.offs) { seg_begin = a.offs[a.seg0 + seg]; seg_rows = a.offs[a.seg0 + seg + 1] - seg_begin; }
  else { seg_begin = 0; seg_rows = a.N; }
  int64_t r0, r1;
  split_range(seg_rows, wi.nsp, sp, r0, r1);
  // wave-uniform by construction; the 64-bit division above runs on the VALU, so say so
  r0 = uni64(r0); r1 = uni64(r1); seg_begin = uni64(seg_begin);
  const int nstages = uni((int)((r1 - r0 + STAGE_ROWS - 1) / STAGE_ROWS));
  const int colA0 = uni(ti * TILE), colB0 = uni(tj * TILE);

  if (ROLER == 3) {
    // ---- loader waves 4..7 ----------------------------------------------------------------
    // Loader d owns stage rows d, d+4, d+8, d+12 and issues, per row, three LDS-DMA
    // instructions: the X panel A row (1 KiB), the X panel B row (off-diagonal tile) or the
    // Y tile row (diagonal tile, 16 lanes), and the row's weight (2 lanes x 4 B): exactly 12
    // per stage, whatever the tile.  Everything per piece is SCALAR (row number by s_load,
    // row base by SALU, LDS address in M0) plus a loop-invariant per-lane VGPR offset: while
    // the compute wave of the same SIMD streams f64 MFMAs a VALU instruction of another
    // wave waits up to a whole MFMA (64 cycles) for an issue slot (measured: 580 cycles per
    // piece with ~8 VALU instructions in it, 180 without the MFMAs running).  The loads are
    // inline asm (saddr form) so that no vector instruction and no compiler-chosen wait
    // enters the loop and the vmcnt count below is exact.
    // Columns past K (or M) are clamped to the last valid pair: they only feed output
    // columns >= K that nothing reads.  Rows past the end read a zero line.
    const int d = wave_all - 4;
    const char *zero_src = reinterpret_cast<const char *>(unip(g_zero_line));
    constexpr unsigned ES = sizeof(T);
    constexpr int EPL = 16 / (int)ES;          // elements per lane of a 16-byte piece
    const char *one_src = reinterpret_cast<const char *>(ES == 8 ? (const void *)unip(g_one_line)
                                                                : (const void *)unip(g_one_line_f));
    // float64: 64 lanes x 16 B = one 1 KiB panel row, the Y tile row by 16 lanes x 16 B (M even),
    // the weight by 2 lanes x 4 B; float32: 32 lanes x 16 B = one 512 B panel row, the Y tile row
    // by 32 lanes x 4 B (any M), the weight by 1 lane x 4 B
    int oa = EPL * lane, ob = EPL * lane, oy = ES == 8 ? 2 * (lane & 15) : (lane & 31);
    if (colA0 + oa > g.K - EPL) oa = g.K - EPL - colA0;
    if (colB0 + ob > g.K - EPL) ob = g.K - EPL - colB0;
    if (oa < 0) oa = 0;
    if (ob < 0) ob = 0;
    const int ycol0 = yc * YT;
    const int ylast = ES == 8 ? g.M - 2 : g.M - 1;
    if (g.M > 0) { if (ycol0 + oy > ylast) oy = ylast - ycol0; if (oy < 0) oy = 0; } else oy = 0;
    const unsigned va = ES * (unsigned)oa, vb = ES * (unsigned)ob, vy = ES * (unsigned)oy, vw = 4u * (unsigned)lane;
    const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)smem_raw);
    auto dma16_all = [&](const char *sbase, unsigned voff, unsigned lds_addr) {
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    };
    auto dma16_lo16 = [&](const char *sbase, unsigned voff, unsigned lds_addr) {
      unsigned keep; unsigned long long ex;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b64 exec, 0xffff\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep), "=&s"(ex) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    };
    auto dma16_lo32 = [&](const char *sbase, unsigned voff, unsigned lds_addr) {
      unsigned keep; unsigned long long ex;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b64 exec, 0xffffffff\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                   "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep), "=&s"(ex) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    };
    auto dma4_lo32 = [&](const char *sbase, unsigned voff, unsigned lds_addr) {
      unsigned keep; unsigned long long ex;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b64 exec, 0xffffffff\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                   "global_load_lds_dword %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep), "=&s"(ex) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    };
    auto dma4_lo1 = [&](const char *sbase, unsigned voff, unsigned lds_addr) {
      unsigned keep; unsigned long long ex;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b64 exec, 1\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                   "global_load_lds_dword %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep), "=&s"(ex) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    };
    auto dma4_lo2 = [&](const char *sbase, unsigned voff, unsigned lds_addr) {
      unsigned keep; unsigned long long ex;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b64 exec, 3\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                   "global_load_lds_dword %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep), "=&s"(ex) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
    };
    // wave-uniform row numbers of stage t, row slots d + 4j.  Gathered ones come by four
    // scalar loads in ONE asm statement that also waits for them (an asm load's destination
    // counts as written when the statement ends; a later, separate wait would let the
    // compiler copy the registers before the data has landed).
    // (32-bit row positions: a 64-bit compare would be a VALU instruction, and a VALU
    //  instruction of this wave waits ~700 cycles for a slot between the other wave's MFMAs:
    //  tools/dma_vs_mfma.hip)
    const int r0i = uni((int)r0), r1i = uni((int)r1);
    auto row_numbers = [&](int t, int64_t (&rn)[4], bool (&ok)[4]) {
      int rr[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        rr[j] = r0i + t * STAGE_ROWS + d + 4 * j;
        ok[j] = rr[j] < r1i;
      }
      if (!GATHER) {
#pragma unroll
        for (int j = 0; j < 4; ++j) rn[j] = seg_begin + rr[j];
        return;
      }
      const int64_t *p0 = a.idx + seg_begin + (ok[0] ? rr[0] : 0);
      const int64_t *p1 = a.idx + seg_begin + (ok[1] ? rr[1] : 0);
      const int64_t *p2 = a.idx + seg_begin + (ok[2] ? rr[2] : 0);
      const int64_t *p3 = a.idx + seg_begin + (ok[3] ? rr[3] : 0);
      int64_t v0 = 0, v1 = 0, v2 = 0, v3 = 0;
      if (r1i > 0) {   // (segment not empty: the clamped addresses are valid)
        asm volatile("s_load_dwordx2 %0, %4, 0x0\n\ts_load_dwordx2 %1, %5, 0x0\n\t"
                     "s_load_dwordx2 %2, %6, 0x0\n\ts_load_dwordx2 %3, %7, 0x0\n\t"
                     "s_waitcnt lgkmcnt(0)"
                     : "=&s"(v0), "=&s"(v1), "=&s"(v2), "=&s"(v3)
                     : "s"(p0), "s"(p1), "s"(p2), "s"(p3) : "memory");
      }
      rn[0] = v0; rn[1] = v1; rn[2] = v2; rn[3] = v3;
    };
    auto issue_stage = [&](int t, const int64_t (&rn)[4], const bool (&ok)[4]) {
      const unsigned bufb = lds0 + (unsigned)((t % NBUF4) * BUF_ELEMS) * ES;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int lrow = d + 4 * j;
        const bool valid = ok[j];
        const char *xrow = reinterpret_cast<const char *>(a.X + rn[j] * (int64_t)g.K);
        const char *pa = valid ? xrow + (int64_t)ES * colA0 : zero_src;
        if (ES == 8) dma16_all(pa, va, bufb + (unsigned)(lrow * PITCH) * ES);
        else dma16_lo32(pa, va, bufb + (unsigned)(lrow * PITCH) * ES);
        if (!diag) {
          const char *pb = valid ? xrow + (int64_t)ES * colB0 : zero_src;
          if (ES == 8) dma16_all(pb, vb, bufb + (unsigned)(PANEL_ELEMS + lrow * PITCH) * ES);
          else dma16_lo32(pb, vb, bufb + (unsigned)(PANEL_ELEMS + lrow * PITCH) * ES);
        } else {
          const char *yrow = (valid && g.M > 0)
              ? reinterpret_cast<const char *>(a.Y + rn[j] * (int64_t)g.M + ycol0) : zero_src;
          if (ES == 8) dma16_lo16(yrow, vy, bufb + (unsigned)(PANEL_ELEMS + lrow * YPITCH) * ES);
          else dma4_lo32(yrow, vy, bufb + (unsigned)(PANEL_ELEMS + lrow * YPITCH) * ES);
        }
        const char *wsrc = valid ? (WEIGHTED ? reinterpret_cast<const char *>(a.w + rn[j]) : one_src) : zero_src;
        if (ES == 8) dma4_lo2(wsrc, vw, bufb + (unsigned)(2 * PANEL_ELEMS + lrow) * ES);
        else dma4_lo1(wsrc, vw, bufb + (unsigned)(2 * PANEL_ELEMS + lrow) * ES);
      }
    };
    // 12 LDS-DMA instructions per stage; ONE stage may stay in flight across a barrier, so
    // that at barrier B_s stage s+2 is in LDS: the compute waves may then read the first
    // fragments of stage s+1 before they reach B_s
    auto wait_one_stage_in_flight = [&]() { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); };
    int64_t rn[4];
    bool ok[4];
    // (The loaders used to raise their wave priority to 3 -- a round-1 choice.  Measured in round 2
    //  (tools/exp_prio.sh): any setting in which the loaders rank ABOVE the compute waves costs the
    //  MFMA stream issue slots -- C3 Gram launch 0.469 ms at loader 3 / compute 0 and at 1 / 0, 0.463 at
    //  3 / 2, 0.452-0.453 at 0 / 0, 0 / 1..3, 1 / 3, 2 / 3; C4 16.89 -> 16.29 ms, C5 28.56 -> 26.96 ms.
    //  Everybody stays at the default priority 0.)
#ifdef CVM_LOADER_PRIO
    __builtin_amdgcn_s_setprio(CVM_LOADER_PRIO);
#endif
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      row_numbers(t, rn, ok);
      issue_stage(t, rn, ok);
    }
    wait_one_stage_in_flight();                       // stages 0 and 1 have landed
    __builtin_amdgcn_s_barrier();                     // B_a (two barriers in every role's prologue)
    __builtin_amdgcn_s_barrier();                     // B_-1
#ifdef CVM_STAMPS
    unsigned long long t_a = 0, t_b = 0, t_c = 0, t0, t1, t2, t3;
#endif
#pragma unroll 1
    for (int s = 0; s < nstages; ++s) {
#ifdef CVM_STAMPS
      STAMP(t0);
#endif
      row_numbers(s + 3, rn, ok);
      issue_stage(s + 3, rn, ok);                     // buffer (s+3)%4 was last read in stage s-1
#ifdef CVM_STAMPS
      STAMP(t1);
#endif
      wait_one_stage_in_flight();                     // stage s+2 has landed
#ifdef CVM_STAMPS
      STAMP(t2);
#endif
#ifdef CVM_STAMPS
      if (!(a.dbg & 4))
#endif
      __builtin_amdgcn_s_barrier();                   // B_s
#ifdef CVM_STAMPS
      STAMP(t3);
      t_a += t1 - t0; t_b += t2 - t1; t_c += t3 - t2;
#endif
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // nothing in flight at wave exit
    if constexpr (FUSEDR) {
      // statistics formed inside the launch: an off-diagonal item needs those of its two panels, which the
      // diagonal items of (fold, ti) and (fold, tj) publish (they come first in every list and wait for
      // nobody).  The first loader wave -- idle by now -- polls the two flags before the barrier that ends
      // the loop, so that every wave's (device-coherent) loads of the statistics come after the match.
      if (a.stat_flags && do_g && !diag && d == 0) {
        const int *fl = a.stat_flags + (size_t)seg * g.P;
        fused_wait_flag<T>(fl + ti, fl + tj, smem_raw, lane, seg, uni(xcd_q), uni(slot_q), a.fused_status,
                           a.retry_items, a.retry_mode, a.test_mode);
      }
    }
    if (FUSEDR) lds_barrier();                      // the compute waves reuse the ring in their epilogue (and the
                                                    // mark of fused_wait_flag has landed in LDS: lgkmcnt(0) first)
    if constexpr (FUSEDR) {
      if (do_g && diag && a.stat_flags) {           // the diagonal item forms its statistics first: two more barriers
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_s_barrier();
        if (d == 0) {
          // publish: the panel's statistics, the Y statistics, sw_T -> fstats (what the off-diagonal items of
          // this fold read) with device-coherent stores, by ONE wave that waits for its own stores and then
          // raises the flag
          const int K = g.K, M = g.M;
          double *fsw = const_cast<double *>(a.fstats) + (size_t)seg * fstat_len(K, M);
          const double *stl = reinterpret_cast<const double *>(smem_raw + diag_tile_bytes<T>()) + 768;
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            const int lcol = 64 * h + lane, col = ti * TILE + lcol;
            if (col < K) { stc(fsw + col, stl[lcol]); stc(fsw + K + col, stl[128 + lcol]); }
          }
          if (lane < 32 && lane < M) { stc(fsw + 2 * K + lane, stl[256 + lane]); stc(fsw + 2 * K + M + lane, stl[288 + lane]); }
          if (lane == 0) stc(fsw + 2 * K + 2 * M, stl[320]);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (lane == 0)
            __hip_atomic_store(a.stat_flags + (size_t)seg * g.P + ti, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      if (do_g && diag && a.out_XTX) {
        // diagonal tile of the fused route: share the epilogue (diag_tile_finish), waves 4..7
        constexpr int TP = TILE + 1;
        const int K = g.K, M = g.M;
        const double *fs = a.fstats + (size_t)seg * fstat_len(K, M);
        const double swt = a.stat_flags ? (reinterpret_cast<const double *>(smem_raw + diag_tile_bytes<T>()) + 768)[320]
                                        : ldc(fs + 2 * K + 2 * M);
        const bool cX = a.flags & CVM_CENTER_X, sX = a.flags & CVM_SCALE_X;
        T (*Td)[TP] = reinterpret_cast<T (*)[TP]>(smem_raw);
        double *rs0 = reinterpret_cast<double *>(smem_raw + diag_tile_bytes<T>());
        T *outp = (T *)a.out_XTX + (size_t)(a.seg0 + seg) * (size_t)K * K;
        lds_barrier();     // B_dump
        diag_tile_finish<T, TP>(Td, rs0, wave_all, 0, ti, K, (const T *)a.G, outp, swt, cX, sX, lane);
        lds_barrier();     // B_parked (LDS only: the direct stores need not be acknowledged first)
        diag_tile_finish<T, TP>(Td, rs0, wave_all, 1, ti, K, (const T *)a.G, outp, swt, cX, sX, lane);
      }
      const int mark = (do_g && !diag) ? fused_mark<T>(a.stat_flags, smem_raw) : FUSED_GO;
      if (do_g && !diag && mark != FUSED_SKIP) {
        // off-diagonal tile of the fused route: help compute wave d with the second half of its
        // 64x64 block (see the compute role's epilogue; same two barriers)
        const int K = g.K, M = g.M;
        const double *fs = a.fstats + (size_t)seg * fstat_len(K, M);
        const double swt = mark == FUSED_POISON ? __builtin_nan("") : ldc(fs + 2 * K + 2 * M);
        const bool cX = a.flags & CVM_CENTER_X, sX = a.flags & CVM_SCALE_X;
        const size_t fo = (size_t)(a.seg0 + seg);
        const int a0 = ti * TILE + 64 * (d >> 1), b0 = tj * TILE + 64 * (d & 1);
        const bool active = a.out_XTX && a0 < K && b0 < K;
        char *slice = smem_raw + (size_t)d * wave_lds_bytes<T>();
        T (*Ts)[65] = reinterpret_cast<T (*)[65]>(slice);
        const double *rs = reinterpret_cast<const double *>(slice + wave_tile_bytes<T>());
        T *outp = (T *)a.out_XTX + fo * (size_t)K * K;
        FusedPre<T> gp;
        if (active) fused_g_preload32<T>(gp, a0, b0, K, (const T *)a.G, lane, 32);   // in flight across B_dump
        lds_barrier();     // B_dump
        if (active)
          fused_finish_direct32<T, 65>(Ts, rs, a0, b0, K, outp, swt, cX, sX, lane, 32, gp);
        lds_barrier();     // B_parked (LDS only: the direct stores need not be acknowledged first)
        if (active) fused_finish_mirror<T, 65>(Ts, a0, b0, K, outp, lane, 32, 64);
      }
    }
#ifdef CVM_STAMPS
    if (lane == 0 && blockIdx.x < 1024) {
      unsigned long long *o = g_stamps + ((size_t)blockIdx.x * 8 + wave_all) * 4;
      o[0] = t_a; o[1] = t_b; o[2] = t_c; o[3] = (unsigned long long)nstages;
    }
#endif
    ROLE_EXIT();
  }

  // ---- compute waves ----------------------------------------------------------------------
  const int stat_role = ROLER;
  acc_t acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (acc_t){0, 0, 0, 0};
  // float32: two-level sums (round 4).  The MFMA accumulates a chain of float32 fmas over the rows; over
  // thousands of rows that chain loses what a blocked GEMM keeps (the reference's sgemm sums in blocks:
  // an unweighted, uncentred XTY of a 6000-row fold came out 8.7 roundings off against NumPy's 2.4).
  // Every FOLD_STAGES stages (1024 rows) the accumulators are folded into a second set and restart from
  // zero, so a chain is at most that long whatever the row-split plan; float64 keeps the single chain
  // (1e-16 * rows is far below the bar).  Cost, same-box A/B at C5 (tools/exp_fold_stages.sh): a fold is 32
  // packed adds + the moves that zero / carry the sets, in front of idle matrix cores -- every 256 rows
  // +2.3 % on the Gram launch, every 1024 rows +0.5 %; a fold test INSIDE the stage loop cost 4 % by itself,
  // hence the two nested loops below.
  constexpr bool TWO_LEVEL = CVM_TWO_LEVEL && sizeof(T) == 4;
  acc_t acc2[TWO_LEVEL ? 16 : 1];
  if (TWO_LEVEL) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc2[i] = (acc_t){0, 0, 0, 0};
  }
  double st_s[4] = {0, 0, 0, 0}, st_q[4] = {0, 0, 0, 0};

  const int lk = lane >> 4, lc = lane & 15;
  // (the plain off-diagonal wave of the two-stage / sweep routes: a 32 x 128 block, see CVM_WIDE_*)
  constexpr bool WIDE = !HWR && MFMR && ROLER == 0 && !FUSEDR && (sizeof(T) == 4 ? CVM_WIDE_F32 : CVM_WIDE_F64);
  const int a_col = h_wave ? 0 : (WIDE ? 32 * wave : 64 * wr);
  const int b_col = h_wave ? 0 : (WIDE ? 0 : 64 * wc);
  const int a_off = a_col + lc;
  const int b_off = h_wave ? PANEL_ELEMS + lc : (diag ? 0 : PANEL_ELEMS) + b_col + lc;

#ifdef CVM_COMPUTE_PRIO
  __builtin_amdgcn_s_setprio(CVM_COMPUTE_PRIO);
#endif
  __syncthreads();   // B_a
  __syncthreads();   // B_-1: stage 0 is in buffer 0

#ifdef CVM_STAMPS
  unsigned long long t_a = 0, t_b = 0, t_c = 0, t0, t1, t2, t3;
  unsigned long long c_loop0, c_loop1;
  STAMP(c_loop0);
#endif
  constexpr bool HW = HWR, MFM = MFMR;
  constexpr int ROLE = ROLER;
  // the X-summing waves are exactly the diagonal 64x64 blocks of a diagonal tile: nothing reads
  // the strictly-lower 16x16 tiles of such a block (the finalize kernels mirror the upper ones),
  // so they are not computed -- 10 MFMAs per k-step instead of 16.  (Not wall time: the block's
  // wave waits for the others at the stage barrier; about 1 % measured in round 1.  The kernel is
  // NOT clock- or power-limited: cvm_clock_probe reads 2.37-2.39 GHz under it, profiles/r6/clock_product.txt.
  // Dropping the padded second column tile of the H wave the same way made the gathered variant
  // 1.7 % slower -- code placement -- and was not kept.)
  constexpr bool TRI = (ROLE == 1) && !HW;
  constexpr int NA = HW ? 8 : (WIDE ? 2 : 4), NB = HW ? 2 : (WIDE ? 8 : 4);
  // Fragments of the NEXT k-step are read while the current one computes, across the
  // stage barrier too (the loaders guarantee stage s+1 is in LDS before stage s starts);
  // the next k-step's weighting (and column sums) sit in the middle of the current
  // k-step's MFMAs, so no MFMA ever waits for LDS or for a VALU result.
  T af[2][NA], bf[2][NB], yf[2][2], wv[2], raw[4];
  auto read_frags = [&](const T *buf, int ks, int slot) {
    const int r = 4 * ks + lk;
    if (MFM || ROLE == 1) {
#pragma unroll
      for (int m = 0; m < NA; ++m) af[slot][m] = buf[a_off + r * PITCH + 16 * m];
    }
    if (MFM) {
#pragma unroll
      for (int n = 0; n < NB; ++n) bf[slot][n] = buf[b_off + r * (HW ? YPITCH : PITCH) + 16 * n];
    }
    if (ROLE == 2) {
#pragma unroll
      for (int n = 0; n < 2; ++n) yf[slot][n] = buf[PANEL_ELEMS + r * YPITCH + 16 * n + lc];
    }
    wv[slot] = buf[2 * PANEL_ELEMS + r];
  };
  // column sums and weighting of one k-step's fragments (slot c); see wgram_kernel for the
  // summation order (same row classes, same combine)
  auto prepare = [&](int c) {
    if (ROLE == 1) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        raw[m] = af[c][m];
        const T pv = WEIGHTED ? (T)(af[c][m] * wv[c]) : af[c][m];
        st_s[m] += pv; st_q[m] += (T)(pv * raw[m]);
        af[c][m] = pv;
      }
    } else {
      if (ROLE == 2) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const T yv = yf[c][n];
          const T pv = WEIGHTED ? (T)(yv * wv[c]) : yv;
          st_s[n] += pv; st_q[n] += (T)(pv * yv);
        }
        st_s[2] += wv[c];
        st_s[3] += (wv[c] != (T)0) ? 1.0 : 0.0;
        st_q[3] += (wv[c] < (T)0) ? 1.0 : 0.0;
      }
      if (MFM && WEIGHTED) {
        if (HW) {   // H wave: 2 Y fragments instead of 8 X fragments
#pragma unroll
          for (int n = 0; n < NB; ++n) bf[c][n] *= wv[c];
        } else {
#pragma unroll
          for (int m = 0; m < NA; ++m) af[c][m] *= wv[c];
        }
      }
    }
  };
  if (MFM || ROLE != 0) {
    read_frags(smem, 0, 0);
    prepare(0);
  }
  // (unrolling this loop over the four LDS buffers to make every LDS address an immediate
  //  was tried: the role functions grow to 11-15 KB each, the instruction cache thrashes and
  //  the kernel loses 25 %)
  // (two loops: the stage loop itself is the loop of rounds 1-3, untouched -- a fold test inside it cost
  //  the float32 kernel 4 % whatever the fold interval -- and the float32 fold sits between two runs of it)
#pragma unroll 1
  for (int sb = 0; sb < nstages; sb += (TWO_LEVEL ? FOLD_STAGES : (1 << 30))) {
  const int se = (TWO_LEVEL && sb + FOLD_STAGES < nstages) ? sb + FOLD_STAGES : nstages;
#pragma unroll 1
  for (int s = sb; s < se; ++s) {
#ifdef CVM_STAMPS
    STAMP(t0);
    STAMP(t1);
    if (a.dbg & 2) { __syncthreads(); continue; }   // diagnostic: loaders alone
#endif
    const T *buf = smem + (s % NBUF4) * BUF_ELEMS;
    const T *nbuf = smem + ((s + 1) % NBUF4) * BUF_ELEMS;
    if constexpr (CVM_INTERLEAVE && MFM && !HW && ROLE == 0) {
      // the plain off-diagonal wave (16 MFMAs per k-step, no column sums): ONE other instruction
      // -- an LDS fragment read of the next k-step, then its four weighting multiplies -- right
      // behind each MFMA, the order pinned by a scheduling barrier after every pair.  The wave's
      // other instructions then issue while the matrix pipe is busy with the MFMA in front of
      // them; issued in two clumps per k-step they hold up the next MFMA (tools/mfma_mix.hip:
      // 70.5 -> 72.0 TFLOP/s in isolation).
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int c = ks & 1;
        const T *rb = ks < 3 ? buf : nbuf;
        const int r = 4 * (ks < 3 ? ks + 1 : 0) + lk;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          acc[i] = MF<T>::mfma(af[c][i / NB], bf[c][i % NB], acc[i]);
          if (WIDE) {
            // the weight first, then the two A-side fragments it multiplies, then the eight of the B side
            if (i == 0) wv[c ^ 1] = rb[2 * PANEL_ELEMS + r];
            else if (i < 3) af[c ^ 1][i - 1] = rb[a_off + r * PITCH + 16 * (i - 1)];
            else if (i < 11) bf[c ^ 1][i - 3] = rb[b_off + r * PITCH + 16 * (i - 3)];
            else if (WEIGHTED && i >= 13 && i < 15) af[c ^ 1][i - 13] *= wv[c ^ 1];
          } else if (i < 4) af[c ^ 1][i] = rb[a_off + r * PITCH + 16 * i];
          else if (i < 8) bf[c ^ 1][i - 4] = rb[b_off + r * PITCH + 16 * (i - 4)];
          else if (i == 8) wv[c ^ 1] = rb[2 * PANEL_ELEMS + r];
          else if (WEIGHTED && i >= 11 && i < 15) af[c ^ 1][i - 11] *= wv[c ^ 1];
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    } else if (MFM || ROLE != 0) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int c = ks & 1;
        if (ks < 3) read_frags(buf, ks + 1, c ^ 1); else read_frags(nbuf, 0, c ^ 1);
        __builtin_amdgcn_sched_barrier(0);
        if (MFM) {
#pragma unroll
          for (int m = 0; m < NA / 2; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n)
              if (!TRI || m <= n) acc[m * NB + n] = MF<T>::mfma(af[c][m], bf[c][n], acc[m * NB + n]);
        }
        __builtin_amdgcn_sched_barrier(0);
        prepare(c ^ 1);   // the other slot: its LDS reads were issued half a k-step ago
        __builtin_amdgcn_sched_barrier(0);
        if (MFM) {
#pragma unroll
          for (int m = NA / 2; m < NA; ++m)
#pragma unroll
            for (int n = 0; n < NB; ++n)
              if (!TRI || m <= n) acc[m * NB + n] = MF<T>::mfma(af[c][m], bf[c][n], acc[m * NB + n]);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#ifdef CVM_STAMPS
    STAMP(t2);
    if (!(a.dbg & 4))   // diagnostic: free-running waves (wrong results)
#endif
    __syncthreads();   // B_s
#ifdef CVM_STAMPS
    STAMP(t3);
    t_a += t1 - t0; t_b += t2 - t1; t_c += t3 - t2;
#endif
  }
  if constexpr (TWO_LEVEL) {
    if (se < nstages) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { acc2[i] += acc[i]; acc[i] = (acc_t){0, 0, 0, 0}; }
    }
  }
  }
  if constexpr (TWO_LEVEL) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = acc2[i] + acc[i];
  }
#ifdef CVM_STAMPS
  STAMP(c_loop1);
  if (lane == 0 && blockIdx.x < 1024) {
    unsigned long long *o = g_stamps + ((size_t)blockIdx.x * 8 + wave) * 4;
    o[0] = t_a; o[1] = t_b; o[2] = t_c; o[3] = (unsigned long long)nstages;
  }
#endif

  if constexpr (FUSEDR) {
#ifdef CVM_STAMPS
    unsigned long long f0, f1, f2, f3, f4, f5;
    STAMP(f0);
#endif
    // ---- fused single-split epilogue: no partials, no apply kernel --------------------------
    // The fold's statistics are already in a.fstats (colstats_kernel + fold_stats_kernel ran
    // first); every wave finishes its own block: total - update, rank-1 centring, outer-std
    // scaling (cvmatrix.py:1001-1010), mirrored store through the wave's slice of the ring.
    __syncthreads();   // all loaders have drained their LDS-DMA (and, statistics formed in the launch: polled the flags)
    const int K = g.K, M = g.M;
    const double *fs = a.fstats + (size_t)seg * fstat_len(K, M);
    const int mark = (do_g && !diag) ? fused_mark<T>(a.stat_flags, smem_raw) : FUSED_GO;
    if (mark == FUSED_SKIP) ROLE_EXIT();      // its wait gave up: nothing is written, the retry launch recomputes the item
    const double swt = mark == FUSED_POISON ? __builtin_nan("") : ldc(fs + 2 * K + 2 * M);
    const bool cX = a.flags & CVM_CENTER_X, cY = a.flags & CVM_CENTER_Y;
    const bool sX = a.flags & CVM_SCALE_X, sY = a.flags & CVM_SCALE_Y;
    const size_t fo = (size_t)(a.seg0 + seg);
    if (h_wave) {
      if (a.out_XTY && M > 0) {
        T *out = (T *)a.out_XTY + fo * (size_t)K * M;
        const T *Ht = (const T *)a.H;
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = ti * TILE + 16 * m + MF<T>::drow(lane, r), col = yc * YT + 16 * n + lc;
              if (row < K && col < M) {
                double v = (double)Ht[(size_t)row * M + col] - (double)acc[m * 2 + n][r];
                if (cX || cY) v -= swt * (ldc(fs + row) * ldc(fs + 2 * K + col));
                if (sX && sY) v = v * (ldc(fs + K + row) * ldc(fs + 2 * K + M + col));
                else if (sX) v = v * ldc(fs + K + row);
                else if (sY) v = v * ldc(fs + 2 * K + M + col);
                out[(size_t)row * M + col] = (T)v;
              }
            }
      }
    } else if (do_g && MFM) {
      // off-diagonal tile: wave w finishes rows 0..31 of its 64x64 block, the idle loader wave
      // w + 4 rows 32..63 (no MFMA is running any more, so its VALU work costs nothing extra):
      // two workgroup barriers, after the accumulators are in LDS and after the finished values
      // are parked for the mirrored store
      const int a0 = ti * TILE + 64 * wr, b0 = tj * TILE + 64 * wc;
      const bool active = a.out_XTX && a0 < K && b0 < K;
      char *slice = smem_raw + (size_t)wave * wave_lds_bytes<T>();
      T (*Ts)[65] = reinterpret_cast<T (*)[65]>(slice);
      double *rs = reinterpret_cast<double *>(slice + wave_tile_bytes<T>());
      T *outp = (T *)a.out_XTX + fo * (size_t)K * K;
      FusedPre<T> gp;
      if (active) {
        // the statistics first, then the G pieces of this wave's 32 rows: both in flight while the
        // accumulators go to LDS (the statistics are waited for alone: they were issued first)
        double r0v = (cX && a0 + lane < K) ? ldc(fs + a0 + lane) : 0.0;
        double r1v = (sX && a0 + lane < K) ? ldc(fs + K + a0 + lane) : 1.0;
        double r2v = (cX && b0 + lane < K) ? ldc(fs + b0 + lane) : 0.0;
        double r3v = (sX && b0 + lane < K) ? ldc(fs + K + b0 + lane) : 1.0;
        // (a poisoned item: the reciprocal stds too, so that scaling without centring cannot pass a number either)
        if (mark == FUSED_POISON) r0v = r1v = r2v = r3v = __builtin_nan("");
        fused_g_preload32<T>(gp, a0, b0, K, (const T *)a.G, lane, 0);
        rs[lane] = r0v; rs[64 + lane] = r1v; rs[128 + lane] = r2v; rs[192 + lane] = r3v;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              Ts[16 * m + MF<T>::drow(lane, r)][16 * n + lc] = acc[m * 4 + n][r];
      }
#ifdef CVM_STAMPS
      STAMP(f1);
#endif
      lds_barrier();     // B_dump
#ifdef CVM_STAMPS
      STAMP(f2);
#endif
      if (active)
        fused_finish_direct32<T, 65>(Ts, rs, a0, b0, K, outp, swt, cX, sX, lane, 0, gp);
#ifdef CVM_STAMPS
      STAMP(f3);
#endif
      lds_barrier();     // B_parked (LDS only: the direct stores need not be acknowledged first)
#ifdef CVM_STAMPS
      STAMP(f4);
#endif
      if (active) fused_finish_mirror<T, 65>(Ts, a0, b0, K, outp, lane, 0, 32);
#ifdef CVM_STAMPS
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      STAMP(f5);
      if (lane == 0 && wave == 0 && blockIdx.x < 1024) {
        unsigned long long *o = g_stamps4 + (size_t)blockIdx.x * 8;
        o[0] = c_loop0 - c_entry; o[1] = c_loop1 - c_loop0; o[2] = f1 - f0; o[3] = f2 - f1; o[4] = f3 - f2;
        o[5] = f4 - f3; o[6] = f5 - f4; o[7] = f5 - c_entry;
      }
#endif
    }
    ROLE_EXIT();
  }

  auto comb = [&](double v) -> double {
    const double v1 = __shfl(v, lc + 16), v2 = __shfl(v, lc + 32), v3 = __shfl(v, lc + 48);
    return ((v + v1) + v2) + v3;
  };
  if (stat_role == 1) {
    double *st = unit_stats<T>(a.ws, g, u);
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const double sv = comb(st_s[m]), qv = comb(st_q[m]);
      if (lk == 0) {
        st[ti * TILE + a_col + 16 * m + lc] = sv;
        st[g.Kp + ti * TILE + a_col + 16 * m + lc] = qv;
      }
    }
  } else if (stat_role == 2) {
    double *st = unit_stats<T>(a.ws, g, u);
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const double sv = comb(st_s[n]), qv = comb(st_q[n]);
      if (lk == 0) {
        st[2 * g.Kp + yc * YT + 16 * n + lc] = sv;
        st[2 * g.Kp + g.Mp + yc * YT + 16 * n + lc] = qv;
      }
    }
    const double swv = comb(st_s[2]), nzv = comb(st_s[3]), ngv = comb(st_q[3]);
    if (yc == 0 && lane == 0) {
      st[2 * g.Kp + 2 * g.Mp + 0] = swv;
      st[2 * g.Kp + 2 * g.Mp + 1] = nzv;
      st[2 * g.Kp + 2 * g.Mp + 2] = ngv;
    }
  }
  if (h_wave) {
    if (g.M > 0) {
      T *hp = unit_h<T>(a.ws, g, u) + (size_t)ti * TILE * g.Mp + yc * YT;
#pragma unroll
      for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            hp[(size_t)(16 * m + MF<T>::drow(lane, r)) * g.Mp + 16 * n + lc] = acc[m * 2 + n][r];
    }
  } else if (do_g) {
    T *tp = unit_tiles<T>(a.ws, g, u) + (size_t)it * TILE * TILE;
    constexpr int SA = (HW || !MFM) ? 4 : NA, SB = (HW || !MFM) ? 4 : NB;      // (the block's MFMA tiles: 4 x 4, or 2 x 8)
#pragma unroll
    for (int m = 0; m < SA; ++m)
#pragma unroll
      for (int n = 0; n < SB; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          partial_store(&tp[(a_col + 16 * m + MF<T>::drow(lane, r)) * TILE + b_col + 16 * n + lc], acc[m * SB + n][r]);
  }
#ifdef CVM_STAMPS
  {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long c_exit = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x < 1024) {
      unsigned long long *o = g_stamps3 + ((size_t)blockIdx.x * 8 + wave) * 2;
      o[0] = c_loop0 - c_entry; o[1] = c_exit - c_loop1;
    }
  }
#endif
  ROLE_EXIT();
}

// ----------------------------------------------------------------------------------
// Compute waves of a DIAGONAL tile (panel x same panel, first Y chunk), balanced: the upper
// triangle of the tile's 8x8 grid of 16x16 MFMA tiles has 36 tiles, 9 + 2 NBY MFMAs per k-step
// and wave with the XTY tiles (11 with M <= 16) instead of 16, nothing below the diagonal tiles
// (the finalize kernels mirror the upper ones).  Who does what is a table (DiagTab):
//   * "rows W and 7-W" (rounds 1-2; kept for two Y tiles, M > 16): wave W takes grid rows W and
//     7-W (8-W and W+1 tiles), the XTY tiles and the column sums of those two row-tiles; it holds
//     the raw fragments of column tiles W..7 -- 8, 7, 6, 5 of them, the busiest wave reads 10
//     fragments per k-step for its 11 MFMAs;
//   * "two triangles, two strips" (round 3, one Y tile): waves 0 and 3 take the upper triangles of
//     the grid's 4x4 corner blocks (10 tiles, fragments of 4 column tiles), waves 1 and 2 the rows
//     0-1 and 2-3 of the off-diagonal 4x4 block (8 tiles, fragments of 6 column tiles); the XTY
//     tiles go where they even the MFMAs out (1 + 3 + 3 + 1: a wave that holds x_t forms w x_t
//     with one multiply) -- 11 MFMAs each as before, 6 / 8 / 8 / 6 fragment reads.  The same
//     MFMA sequence per accumulator whoever runs it: the same bits.  (Measured: no faster than
//     the first, DESIGN.md section 7 -- the fragment reads are not what a diagonal stage waits for.)
//   YSTAT: this wave also sums the Y columns, sw and nz (wave 3 of panel 0)
// ----------------------------------------------------------------------------------
struct DiagTab {
  int nf, na, ng, nx;
  int ft[8];            // column tiles whose raw fragments the wave holds (the B side)
  int at[4];            // the weighted (A side) fragments it forms: indices into ft
  int ga[10], gb[10];   // G tiles: A index (into at), B index (into ft)
  int xa[3];            // XTY row tiles: A index
  int ca[2];            // column sums of two tiles: A index
  int rd[8];            // the order the fragments of the next k-step are read in: A sides first
};
constexpr DiagTab diag_tab_fill(int W, bool strips) {
  DiagTab t{};
  if (!strips) {
    const int R1 = 7 - W;
    t.nf = 8 - W;
    for (int j = 0; j < t.nf; ++j) t.ft[j] = W + j;
    t.na = 2; t.at[0] = 0; t.at[1] = R1 - W;
    t.ng = 9;
    for (int j = 0; j < t.nf; ++j) { t.ga[j] = 0; t.gb[j] = j; }
    for (int j = 0; j < W + 1; ++j) { t.ga[t.nf + j] = 1; t.gb[t.nf + j] = R1 - W + j; }
    t.nx = 2; t.xa[0] = 0; t.xa[1] = 1;
    t.ca[0] = 0; t.ca[1] = 1;
    return t;
  }
  if (W == 0 || W == 3) {                 // a 4x4 corner block's upper triangle
    const int b = W == 0 ? 0 : 4;
    t.nf = 4; t.na = 4;
    for (int j = 0; j < 4; ++j) { t.ft[j] = b + j; t.at[j] = j; }
    t.ng = 0;
    for (int i = 0; i < 4; ++i)
      for (int j = i; j < 4; ++j) { t.ga[t.ng] = i; t.gb[t.ng] = j; ++t.ng; }
    t.nx = 1; t.xa[0] = W == 0 ? 0 : 3;   // XTY of tile 0 / tile 7
    t.ca[0] = W == 0 ? 0 : 2; t.ca[1] = W == 0 ? 1 : 3;   // column sums of tiles 0, 1 / 6, 7
    return t;
  }
  // W == 1: rows 0, 1 x columns 4..7; W == 2: rows 2, 3 x columns 4..7
  const int r = W == 1 ? 0 : 2;
  t.nf = 6;
  t.ft[0] = r; t.ft[1] = r + 1;
  for (int j = 0; j < 4; ++j) t.ft[2 + j] = 4 + j;
  t.ng = 0;
  for (int i = 0; i < 2; ++i)
    for (int j = 0; j < 4; ++j) { t.ga[t.ng] = i; t.gb[t.ng] = 2 + j; ++t.ng; }
  if (W == 1) {                           // w x of tiles 0, 1, 4, 5; XTY 1, 4, 5; column sums 4, 5
    t.na = 4; t.at[0] = 0; t.at[1] = 1; t.at[2] = 2; t.at[3] = 3;
    t.nx = 3; t.xa[0] = 1; t.xa[1] = 2; t.xa[2] = 3;
    t.ca[0] = 2; t.ca[1] = 3;
  } else {                                // w x of tiles 2, 3, 6; XTY 2, 3, 6; column sums 2, 3
    t.na = 3; t.at[0] = 0; t.at[1] = 1; t.at[2] = 4;
    t.nx = 3; t.xa[0] = 0; t.xa[1] = 1; t.xa[2] = 2;
    t.ca[0] = 0; t.ca[1] = 1;
  }
  return t;
}
constexpr DiagTab diag_tab(int W, bool strips) {
  DiagTab t = diag_tab_fill(W, strips);
  int n = 0;
  for (int q = 0; q < t.na; ++q) t.rd[n++] = t.at[q];
  for (int j = 0; j < t.nf; ++j) {
    bool is_a = false;
    for (int q = 0; q < t.na; ++q) is_a = is_a || t.at[q] == j;
    if (!is_a) t.rd[n++] = j;
  }
  return t;
}
#ifndef CVM_DIAG_STRIPS
#define CVM_DIAG_STRIPS 1
#endif

template <typename T, bool WEIGHTED, bool GATHER, int W, int NBY, bool YSTAT, bool FUSEDR = false>
__device__ __noinline__ ROLE_ATTR void wgram4_diag_body(kargs_ptr<T> kargs, int xcd_q, int slot_q) {
  typedef typename MF<T>::acc_t acc_t;
  const WgramArgs<T> a = kernel_args<T>(kargs);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T *smem = reinterpret_cast<T *>(smem_raw);
  const Geom &g = a.g;
  const int lane = threadIdx.x & 63;
  Item wi;
  decode_slot(a, uni(xcd_q), (long)uni(slot_q), wi);   // diagonal tile, Y chunk 0 (the kernel checked)
  const long u = wi.u;
  const int it = wi.it, seg = wi.seg, sp = wi.sp, ti = wi.ti;
  int64_t seg_begin, seg_rows;
  if (a.offs) { seg_begin = a.offs[a.seg0 + seg]; seg_rows = a.offs[a.seg0 + seg + 1] - seg_begin; }
  else { seg_begin = 0; seg_rows = a.N; }
  int64_t r0, r1;
  split_range(seg_rows, wi.nsp, sp, r0, r1);
  r0 = uni64(r0); r1 = uni64(r1);
  const int nstages = uni((int)((r1 - r0 + STAGE_ROWS - 1) / STAGE_ROWS));

  constexpr DiagTab P = diag_tab(W, CVM_DIAG_STRIPS && NBY == 1);
  constexpr int NF = P.nf, NA = P.na, NG = P.ng, NX = P.nx;
  acc_t acc[NG], acch[NX * NBY];
#pragma unroll
  for (int i = 0; i < NG; ++i) acc[i] = (acc_t){0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < NX * NBY; ++i) acch[i] = (acc_t){0, 0, 0, 0};
  constexpr bool TWO_LEVEL = CVM_TWO_LEVEL && sizeof(T) == 4;      // float32: two-level sums, see wgram4_body
  acc_t acc2[TWO_LEVEL ? NG : 1], acch2[TWO_LEVEL ? NX * NBY : 1];
  if (TWO_LEVEL) {
#pragma unroll
    for (int i = 0; i < NG; ++i) acc2[i] = (acc_t){0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < NX * NBY; ++i) acch2[i] = (acc_t){0, 0, 0, 0};
  }
  double st_s[2] = {0, 0}, st_q[2] = {0, 0};
  double sy[NBY], qy[NBY], sw_ = 0, nz_ = 0, ng_ = 0;
#pragma unroll
  for (int n = 0; n < NBY; ++n) sy[n] = qy[n] = 0;

  const int lk = lane >> 4, lc = lane & 15;
  // statistics formed inside the launch (fused route, WgramArgs::stat_flags): this wave's two column tiles
  // are summed like in the unfused kernel and finished in the epilogue; the full-data sums they are
  // subtracted from are requested now and arrive during the loop
  const bool ink = FUSEDR && a.stat_flags != nullptr;
  double ink_gs[2] = {0, 0}, ink_gq[2] = {0, 0}, ink_tot[2] = {0, 0};
  if (ink) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      int col = ti * TILE + 16 * P.ft[P.at[P.ca[q]]] + lc;
      if (col >= g.K) col = g.K - 1;
      ink_gs[q] = a.gstats[col]; ink_gq[q] = a.gstats[g.K + col];
    }
    ink_tot[0] = a.gstats[2 * g.K + 2 * g.M]; ink_tot[1] = a.gstats[2 * g.K + 2 * g.M + 1];
  }
#ifdef CVM_COMPUTE_PRIO
  __builtin_amdgcn_s_setprio(CVM_COMPUTE_PRIO);
#endif
  __syncthreads();   // B_a
  __syncthreads();   // B_-1: stage 0 is in buffer 0

  T bf[2][NF], aw[2][NA], yf[2][NBY], wv[2];
  auto read_frags = [&](const T *buf, int ks, int slot) {
    const int r = 4 * ks + lk;
#pragma unroll
    for (int j = 0; j < NF; ++j) bf[slot][j] = buf[r * PITCH + 16 * P.ft[j] + lc];
#pragma unroll
    for (int n = 0; n < NBY; ++n) yf[slot][n] = buf[PANEL_ELEMS + r * YPITCH + 16 * n + lc];
    wv[slot] = buf[2 * PANEL_ELEMS + r];
  };
  // weighting and column sums of one k-step's fragments: same row classes and combine order
  // as everywhere else (a constant-one column gives s == q == sw bit for bit)
  auto prepare = [&](int c) {
#pragma unroll
    for (int q = 0; q < NA; ++q) {
      const T x = bf[c][P.at[q]];
      aw[c][q] = WEIGHTED ? (T)(x * wv[c]) : x;
    }
    if (!FUSEDR || ink) {   // (the fused route: from colstats_kernel, or -- ink -- formed here like everywhere else)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const T x = bf[c][P.at[P.ca[q]]], pv = aw[c][P.ca[q]];
        st_s[q] += pv; st_q[q] += (T)(pv * x);
      }
    }
    if (YSTAT && (!FUSEDR || ink)) {
#pragma unroll
      for (int n = 0; n < NBY; ++n) {
        const T yv = yf[c][n];
        const T pv = WEIGHTED ? (T)(yv * wv[c]) : yv;
        sy[n] += pv; qy[n] += (T)(pv * yv);
      }
      sw_ += wv[c];
      nz_ += (wv[c] != (T)0) ? 1.0 : 0.0;
      ng_ += (wv[c] < (T)0) ? 1.0 : 0.0;
    }
  };
  // MFMA i of a k-step: the G tiles in table order, then the XTY tiles
  auto mfma_i = [&](int i, int c) {
    if (i < NG) acc[i] = MF<T>::mfma(aw[c][P.ga[i]], bf[c][P.gb[i]], acc[i]);
    else {
      const int h = i - NG, x = h / NBY, n = h - x * NBY;
      acch[x * NBY + n] = MF<T>::mfma(aw[c][P.xa[x]], yf[c][n], acch[x * NBY + n]);
    }
  };
  read_frags(smem, 0, 0);
  prepare(0);
#pragma unroll 1
  for (int sb = 0; sb < nstages; sb += (TWO_LEVEL ? FOLD_STAGES : (1 << 30))) {
  const int se = (TWO_LEVEL && sb + FOLD_STAGES < nstages) ? sb + FOLD_STAGES : nstages;
#pragma unroll 1
  for (int s = sb; s < se; ++s) {
    const T *buf = smem + (s % NBUF4) * BUF_ELEMS;
    const T *nbuf = smem + ((s + 1) % NBUF4) * BUF_ELEMS;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int c = ks & 1;
      constexpr int NM = NG + NX * NBY;
#if CVM_INTERLEAVE
      // one LDS read of the next k-step right behind each MFMA, the order pinned (see the
      // off-diagonal wave in wgram4_body): first what the weighting / column sums need (w, the
      // A-side fragments, the Y fragments), then the other fragments (DiagTab::rd); the weighting
      // and the sums of the next k-step behind the last MFMA.  NR = NF + NBY + 1 reads for NM MFMAs.
      constexpr int NR = NF + NBY + 1;
      static_assert(NR <= NM, "more fragment reads than MFMAs in a k-step");
      const T *rb = ks < 3 ? buf : nbuf;
      const int r = 4 * (ks < 3 ? ks + 1 : 0) + lk;
#pragma unroll
      for (int i = 0; i < NM; ++i) {
        mfma_i(i, c);
        if (i == 0) wv[c ^ 1] = rb[2 * PANEL_ELEMS + r];
        else if (i <= NA) bf[c ^ 1][P.rd[i - 1]] = rb[r * PITCH + 16 * P.ft[P.rd[i - 1]] + lc];
        else if (i <= NA + NBY) yf[c ^ 1][i - NA - 1] = rb[PANEL_ELEMS + r * YPITCH + 16 * (i - NA - 1) + lc];
        else if (i < NR) bf[c ^ 1][P.rd[i - 1 - NBY]] = rb[r * PITCH + 16 * P.ft[P.rd[i - 1 - NBY]] + lc];
        if (i == NM - 1) prepare(c ^ 1);
        __builtin_amdgcn_sched_barrier(0);
      }
#else
      if (ks < 3) read_frags(buf, ks + 1, c ^ 1); else read_frags(nbuf, 0, c ^ 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < NM / 2; ++i) mfma_i(i, c);
      __builtin_amdgcn_sched_barrier(0);
      prepare(c ^ 1);   // the other slot: its LDS reads were issued most of a k-step ago
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = NM / 2; i < NM; ++i) mfma_i(i, c);
      __builtin_amdgcn_sched_barrier(0);
#endif
    }
    __syncthreads();   // B_s
  }
  if constexpr (TWO_LEVEL) {
    if (se < nstages) {
#pragma unroll
      for (int i = 0; i < NG; ++i) { acc2[i] += acc[i]; acc[i] = (acc_t){0, 0, 0, 0}; }
#pragma unroll
      for (int i = 0; i < NX * NBY; ++i) { acch2[i] += acch[i]; acch[i] = (acc_t){0, 0, 0, 0}; }
    }
  }
  }
  if constexpr (TWO_LEVEL) {
#pragma unroll
    for (int i = 0; i < NG; ++i) acc[i] = acc2[i] + acc[i];
#pragma unroll
    for (int i = 0; i < NX * NBY; ++i) acch[i] = acch2[i] + acch[i];
  }

  if constexpr (FUSEDR) {
    // ---- fused epilogue (one unit per fold): the four waves put their tiles of the upper
    // triangle into one 128 x 128 image in the LDS ring, then waves 0, 1 and 3 finish the
    // 64 x 64 blocks (0,0), (0,1) and (1,1) as in wgram4_body; every wave finishes the XTY
    // rows of its row-tiles itself.
    __syncthreads();   // all loaders have drained their LDS-DMA
    const int K = g.K, M = g.M;
    double *fsw = const_cast<double *>(a.fstats) + (size_t)seg * fstat_len(K, M);
    const double *fs = fsw;
    const bool cX = a.flags & CVM_CENTER_X, cY = a.flags & CVM_CENTER_Y;
    const bool sX = a.flags & CVM_SCALE_X, sY = a.flags & CVM_SCALE_Y;
    const size_t fo = (size_t)(a.seg0 + seg);
    // statistics in LDS (ink): [0,128) means of the panel's columns, [128,256) reciprocal stds, [256,288) /
    // [288,320) the same for the Y chunk, [320] sw_T -- behind the tile image and its statistics blocks
    double *stl = reinterpret_cast<double *>(smem_raw + diag_tile_bytes<T>()) + 768;
    if (ink) {
      // ---- the statistics of this panel (and, wave 3: of Y and the fold's totals), formed here ----------
      // fold_stats_kernel's arithmetic (cvmatrix.py:612-620, 1020, 1043, 1079, 1119-1128) on the sums this
      // item has just taken; every item of a fold sums the same rows in the same order, so every panel's
      // statistics -- and the Y statistics and totals every diagonal item of the fold writes -- are the same
      // bits whoever writes them.  They go to LDS for this item's own epilogue; the first loader wave then
      // copies them to fstats with device-coherent stores, waits for ITS stores alone and raises the flag,
      // while the other waves are already finishing the tile (wgram4_body, loader role).
      auto comb = [&](double v) -> double {
        const double v1 = __shfl(v, lc + 16), v2 = __shfl(v, lc + 32), v3 = __shfl(v, lc + 48);
        return ((v + v1) + v2) + v3;
      };
      double *tot = stl + 322;                                      // [0] sw_V, [1] nz_V
      if (YSTAT) {
        const double swv = comb(sw_), nzv = comb(nz_);
        if (lane == 0) { tot[0] = swv; tot[1] = nzv; }
      }
      lds_barrier();
      const double swv = tot[0], nzv = tot[1];
      const double swt_ = ink_tot[0] - swv, nzt = ink_tot[1] - nzv;
      const double divisor = (nzt - a.ddof) * swt_ / nzt;
      const bool rXTY = a.flags & CVM_RET_XTY;
      const bool want_muX = cX || sX || (rXTY && cY), want_sdX = sX;
      const bool want_muY = rXTY && (cX || cY || sY), want_sdY = rXTY && sY;
      auto finish_col = [&](double sv, double qv, double gs, double gq, bool want_sd, double &mu, double &isd, double &sd) {
        const double st_ = gs - sv;          // cvmatrix.py:1020
        mu = st_ / swt_;                     // cvmatrix.py:1043
        sd = 1.0;
        if (want_sd) {
          const double qt = gq - qv;
          double var = (-2 * mu * st_ + swt_ * (mu * mu) + qt) / divisor;   // 1119-1123
          var = (var < 0) ? 0.0 : var;       // np.maximum(var, 0): NaN stays NaN
          sd = sqrt(var);
          if (sd <= a.resolution) sd = 1.0;  // 1128
        }
        isd = 1.0 / sd;
      };
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const double sv = comb(st_s[q]), qv = comb(st_q[q]);
        const int lcol = 16 * P.ft[P.at[P.ca[q]]] + lc, col = ti * TILE + lcol;
        if (lk == 0) {
          double mu = 0.0, isd = 1.0, sd = 1.0;
          if (col < K && want_muX) {
            finish_col(sv, qv, ink_gs[q], ink_gq[q], want_sdX, mu, isd, sd);
            const size_t o = fo * K + col;
            if (a.out_muX) ((T *)a.out_muX)[o] = (T)mu;
            if (a.out_sdX && want_sdX) ((T *)a.out_sdX)[o] = (T)sd;
          }
          stl[lcol] = mu; stl[128 + lcol] = isd;
        }
      }
      if (YSTAT) {
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int col = 16 * n + lc;
          double mu = 0.0, isd = 1.0, sd = 1.0;
          if (n < NBY) {
            const double sv = comb(sy[n < NBY ? n : 0]), qv = comb(qy[n < NBY ? n : 0]);
            if (lk == 0 && col < M && want_muY) {
              finish_col(sv, qv, a.gstats[2 * K + col], a.gstats[2 * K + M + col], want_sdY, mu, isd, sd);
              if (ti == 0) {
                const size_t o = fo * M + col;
                if (a.out_muY) ((T *)a.out_muY)[o] = (T)mu;
                if (a.out_sdY && want_sdY) ((T *)a.out_sdY)[o] = (T)sd;
              }
            }
          }
          if (lk == 0) { stl[256 + col] = mu; stl[288 + col] = isd; }
        }
        if (lane == 0) {
          stl[320] = swt_;
          if (a.out_fold && ti == 0) { double *o = a.out_fold + 4 * fo; o[0] = swt_; o[1] = nzt; o[2] = swv; o[3] = nzv; }
        }
      }
      lds_barrier();
      // (the first loader wave publishes them: it has half a compute wave's share of the tile finish)
    }
    // the statistics this item's own epilogue uses: from LDS when they were formed here, else from fstats
    auto st_mu = [&](int col) -> double { return ink ? stl[col - ti * TILE] : ldc(fs + col); };
    auto st_isd = [&](int col) -> double { return ink ? stl[128 + col - ti * TILE] : ldc(fs + K + col); };
    auto st_muY = [&](int col) -> double { return ink ? stl[256 + col] : ldc(fs + 2 * K + col); };
    auto st_isdY = [&](int col) -> double { return ink ? stl[288 + col] : ldc(fs + 2 * K + M + col); };
    const double swt = ink ? stl[320] : ldc(fs + 2 * K + 2 * M);
    if (a.out_XTY && M > 0) {
      T *out = (T *)a.out_XTY + fo * (size_t)K * M;
      const T *Ht = (const T *)a.H;
#pragma unroll
      for (int i = 0; i < NX; ++i)
#pragma unroll
        for (int n = 0; n < NBY; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = ti * TILE + 16 * P.ft[P.at[P.xa[i]]] + MF<T>::drow(lane, r), col = 16 * n + lc;
            if (row < K && col < M) {
              double v = (double)Ht[(size_t)row * M + col] - (double)acch[i * NBY + n][r];
              if (cX || cY) v -= swt * (st_mu(row) * st_muY(col));
              if (sX && sY) v = v * (st_isd(row) * st_isdY(col));
              else if (sX) v = v * st_isd(row);
              else if (sY) v = v * st_isdY(col);
              out[(size_t)row * M + col] = (T)v;
            }
          }
    }
    if (a.out_XTX) {
      constexpr int TP = TILE + 1;
      T (*Td)[TP] = reinterpret_cast<T (*)[TP]>(smem_raw);
#pragma unroll
      for (int j = 0; j < NG; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Td[16 * P.ft[P.at[P.ga[j]]] + MF<T>::drow(lane, r)][16 * P.ft[P.gb[j]] + lc] = acc[j][r];
      double *rs0 = reinterpret_cast<double *>(smem_raw + diag_tile_bytes<T>());
      if (W != 2) {
        // waves 0, 1, 3 set up the statistics of blocks 0, 1, 2
        const int bb = W == 3 ? 2 : W, si = bb == 2, sj = bb >= 1;
        const int a0 = ti * TILE + 64 * si, b0 = ti * TILE + 64 * sj;
        double *rs = rs0 + 256 * bb;
        rs[lane] = (cX && a0 + lane < K) ? st_mu(a0 + lane) : 0.0;
        rs[64 + lane] = (sX && a0 + lane < K) ? st_isd(a0 + lane) : 1.0;
        rs[128 + lane] = (cX && b0 + lane < K) ? st_mu(b0 + lane) : 0.0;
        rs[192 + lane] = (sX && b0 + lane < K) ? st_isd(b0 + lane) : 1.0;
      }
      lds_barrier();     // B_dump: the tile and the statistics are in LDS (all eight waves)
      T *outp = (T *)a.out_XTX + fo * (size_t)K * K;
      diag_tile_finish<T, TP>(Td, rs0, W, 0, ti, K, (const T *)a.G, outp, swt, cX, sX, lane);
      lds_barrier();     // B_parked (LDS only: the direct stores need not be acknowledged first)
      diag_tile_finish<T, TP>(Td, rs0, W, 1, ti, K, (const T *)a.G, outp, swt, cX, sX, lane);
    }
    ROLE_EXIT();
  }

  auto comb = [&](double v) -> double {
    const double v1 = __shfl(v, lc + 16), v2 = __shfl(v, lc + 32), v3 = __shfl(v, lc + 48);
    return ((v + v1) + v2) + v3;
  };
  double *st = unit_stats<T>(a.ws, g, u);
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const double sv = comb(st_s[q]), qv = comb(st_q[q]);
    const int tile = P.ft[P.at[P.ca[q]]];
    if (lk == 0) { st[ti * TILE + 16 * tile + lc] = sv; st[g.Kp + ti * TILE + 16 * tile + lc] = qv; }
  }
  if (YSTAT) {
#pragma unroll
    for (int n = 0; n < NBY; ++n) {
      const double sv = comb(sy[n]), qv = comb(qy[n]);
      if (lk == 0) { st[2 * g.Kp + 16 * n + lc] = sv; st[2 * g.Kp + g.Mp + 16 * n + lc] = qv; }
    }
    if (NBY == 1 && lk == 0) { st[2 * g.Kp + 16 + lc] = 0.0; st[2 * g.Kp + g.Mp + 16 + lc] = 0.0; }
    const double swv = comb(sw_), nzv = comb(nz_), ngv = comb(ng_);
    if (lane == 0) {
      st[2 * g.Kp + 2 * g.Mp + 0] = swv;
      st[2 * g.Kp + 2 * g.Mp + 1] = nzv;
      st[2 * g.Kp + 2 * g.Mp + 2] = ngv;
    }
  }
  if (g.M > 0) {
    T *hp = unit_h<T>(a.ws, g, u) + (size_t)ti * TILE * g.Mp;
#pragma unroll
    for (int i = 0; i < NX; ++i)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          hp[(size_t)(16 * P.ft[P.at[P.xa[i]]] + MF<T>::drow(lane, r)) * g.Mp + 16 * n + lc] =
              (n < NBY) ? acch[i * NBY + (n < NBY ? n : 0)][r] : 0.0;
  }
  T *tp = unit_tiles<T>(a.ws, g, u) + (size_t)it * TILE * TILE;
#pragma unroll
  for (int j = 0; j < NG; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      partial_store(&tp[(16 * P.ft[P.at[P.ga[j]]] + MF<T>::drow(lane, r)) * TILE + 16 * P.ft[P.gb[j]] + lc], acc[j][r]);
  ROLE_EXIT();
}

// One wave's part of one work item: pick the role function of (tile kind, wave).
template <typename T, bool WEIGHTED, bool GATHER, bool FUSED>
__device__ __forceinline__ void wgram4_item(const WgramArgs<T> &a, kargs_ptr<T> kargs, const Item &wi, int wave_all,
                                            int xq, int sq) {
  const Geom &g = a.g;
  const int wave = wave_all & 3;
  if (wave_all >= 4) { wgram4_body<T, WEIGHTED, GATHER, false, false, 3, FUSED>(kargs, xq, sq); return; }
  const int ti = wi.ti, tj = wi.tj, yc = wi.yc;
  const bool diag = (ti == tj);
  const bool do_g = !g.diag_only && yc == 0;
  if (FUSED && diag && do_g) {
    // diagonal tile of the fused route: the balanced waves with the fused epilogue
    const bool wide = g.M > 16;
#define CVM_DIAGF(WV)                                                                        \
    do {                                                                                     \
      if (wide) wgram4_diag_body<T, WEIGHTED, GATHER, WV, 2, false, true>(kargs, xq, sq);       \
      else wgram4_diag_body<T, WEIGHTED, GATHER, WV, 1, false, true>(kargs, xq, sq);            \
    } while (0)
    if (wave == 0) CVM_DIAGF(0);
    else if (wave == 1) CVM_DIAGF(1);
    else if (wave == 2) CVM_DIAGF(2);
    else if (!a.stat_flags) CVM_DIAGF(3);
    else if (wide) wgram4_diag_body<T, WEIGHTED, GATHER, 3, 2, true, true>(kargs, xq, sq);     // (+ the Y columns, sw, nz)
    else wgram4_diag_body<T, WEIGHTED, GATHER, 3, 1, true, true>(kargs, xq, sq);
#undef CVM_DIAGF
    return;
  }
  if (FUSED) {   // statistics come from colstats_kernel: no summing roles
    if (diag && wave == 2) wgram4_body<T, WEIGHTED, GATHER, true, true, 0, true>(kargs, xq, sq);
    else if (do_g) wgram4_body<T, WEIGHTED, GATHER, false, true, 0, true>(kargs, xq, sq);
    else wgram4_body<T, WEIGHTED, GATHER, false, false, 0, true>(kargs, xq, sq);
    return;
  }
  if (diag && do_g) {
    // diagonal tile, first Y chunk: the four balanced waves (see wgram4_diag_body)
    const bool wide = g.M > 16, ys = (ti == 0);
#define CVM_DIAG(WV)                                                                         \
    do {                                                                                     \
      if (wide) wgram4_diag_body<T, WEIGHTED, GATHER, WV, 2, false>(kargs, xq, sq);             \
      else wgram4_diag_body<T, WEIGHTED, GATHER, WV, 1, false>(kargs, xq, sq);                  \
    } while (0)
    if (wave == 0) CVM_DIAG(0);
    else if (wave == 1) CVM_DIAG(1);
    else if (wave == 2) CVM_DIAG(2);
    else if (!ys) CVM_DIAG(3);
    else if (wide) wgram4_diag_body<T, WEIGHTED, GATHER, 3, 2, true>(kargs, xq, sq);
    else wgram4_diag_body<T, WEIGHTED, GATHER, 3, 1, true>(kargs, xq, sq);
#undef CVM_DIAG
    return;
  }
  const int role = !diag ? 0 : ((yc == 0 && (wave == 0 || wave == 3)) ? 1 : ((ti == 0 && wave == 1) ? 2 : 0));
  if (diag && wave == 2) wgram4_body<T, WEIGHTED, GATHER, true, true, 0>(kargs, xq, sq);
  else if (role == 1) { if (do_g) wgram4_body<T, WEIGHTED, GATHER, false, true, 1>(kargs, xq, sq); else wgram4_body<T, WEIGHTED, GATHER, false, false, 1>(kargs, xq, sq); }
  else if (role == 2) { if (do_g) wgram4_body<T, WEIGHTED, GATHER, false, true, 2>(kargs, xq, sq); else wgram4_body<T, WEIGHTED, GATHER, false, false, 2>(kargs, xq, sq); }
  else { if (do_g) wgram4_body<T, WEIGHTED, GATHER, false, true, 0>(kargs, xq, sq); else wgram4_body<T, WEIGHTED, GATHER, false, false, 0>(kargs, xq, sq); }
}

// PERSISTENT workgroups, one per CU, that pull work items from per-XCD queues.
// Why not one workgroup per item: the hardware deals the workgroups of a launch to the XCDs and,
// inside an XCD, to its four shader engines strictly round-robin and IN ORDER -- a workgroup
// whose turn falls on a full shader engine holds up every later one of its XCD even while CUs of
// the other engines idle (tools/dispatch_probe.hip: 240 long + 280 short workgroups take three
// rounds, not two).  With items of two durations (off-diagonal and diagonal tiles) that costs
// 15-40 % of a launch; equal durations would need split counts the tile counts do not divide.
// Here a workgroup reads the XCD it runs on (XCC_ID), takes the next position of THAT XCD's list
// with one integer atomic (the lists: geometry.hpp -- contiguous ranges, longest items first, the
// tiles that share rows next to each other) and, when its own list is empty, helps the other
// XCDs.  Results do not depend on who computes an item (every partial has its own slot, sums are
// taken in slot order).  The next position is fetched by an idle loader wave while the compute
// waves store the current item's partials.
__device__ __forceinline__ int fetch_position(unsigned *queue, int xcd) {
  return (int)__hip_atomic_fetch_add(queue + xcd * QUEUE_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T, bool WEIGHTED, bool GATHER, bool FUSED = false>
__global__ __launch_bounds__(NT4, 2) void wgram4_kernel(const WgramArgs<T> a) {
  __shared__ int s_next[2][2];                   // [parity]: {xcd, position} of the next item, xcd < 0: none
  const kargs_ptr<T> kargs = kernarg_address<T>();
  const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const int home = (int)(xcc & 7u);
  const long per_xcd = a.ipx0 + a.ipx1;
  // next real item of the lists, starting with the home XCD's: {xcd, position} or xcd = -1
  auto fetch = [&](int &fx, int &fq, int &probe) {
    Item tmp;
    if (FUSED && a.retry_mode) {
      // the retry launch: the items whose wait gave up in the first launch, in the order they were listed
      const int q = fetch_position(a.queue, 0);
      const int n = __hip_atomic_load(a.fused_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (q < n) {
        const unsigned long long it = __hip_atomic_load(a.retry_items + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        fx = (int)(it >> 32); fq = (int)(it & 0xffffffffu);
      } else { fx = -1; fq = 0; }
      return;
    }
    while (probe < 8) {
      const int x = (home + probe) & 7;
      const int q = fetch_position(a.queue, x);
      if (q >= per_xcd) { ++probe; continue; }        // that list is exhausted: next XCD's
      if (decode_slot(a, x, (long)q, tmp)) { fx = x; fq = q; return; }
      // (a padding position of a list shorter than the others: take the next one)
    }
    fx = -1; fq = 0;
  };
  // clock probe (off unless cvm_clock_probe handed a buffer): the workgroup's first clock pair
  const bool clk = a.clock_stamps && threadIdx.x == 0 && (int)blockIdx.x < a.clock_wgs;
  if (clk) {
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();
    a.clock_stamps[4 * blockIdx.x + 0] = c0; a.clock_stamps[4 * blockIdx.x + 1] = q0;
  }
  int probe = 0;                                 // lists already found empty (kept by the fetching thread)
  if (threadIdx.x == 4 * 64) {
    int fx, fq;
    fetch(fx, fq, probe);
    s_next[0][0] = fx; s_next[0][1] = fq;
  }
  __syncthreads();
#ifdef CVM_STAMPS
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int n = 0;; ++n) {
    const int xq = uni(s_next[n & 1][0]), sq = uni(s_next[n & 1][1]);
    if (xq < 0) break;
    Item wi;
    decode_slot(a, xq, (long)sq, wi);
    wgram4_item<T, WEIGHTED, GATHER, FUSED>(a, kargs, wi, wave_all, xq, sq);
    if (threadIdx.x == 4 * 64) {                 // loader wave 4 is done first: it fetches
      int fx, fq;
      fetch(fx, fq, probe);
      s_next[(n + 1) & 1][0] = fx; s_next[(n + 1) & 1][1] = fq;
    }
    // the item is finished with the LDS ring; s_next is visible.  LDS only: the item's output / partial
    // stores need not be acknowledged before the next item's loaders start (a __syncthreads() here
    // made every wave wait for them -- several microseconds per item with stores streaming to HBM)
    lds_barrier();
  }
  if (clk) {                                     // ... and its second, behind its last item (its stores may still be in flight)
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
    a.clock_stamps[4 * blockIdx.x + 2] = c1; a.clock_stamps[4 * blockIdx.x + 3] = q1;
  }
  // the queue block is the library's (host.hpp: queue pool) and must be all zero again for its
  // next launch: the last workgroup to leave -- every other one has made its last fetch -- clears it
  if (threadIdx.x == 0) {
    const unsigned left = __hip_atomic_fetch_add(a.queue + QUEUE_DONE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (left == gridDim.x - 1) {
#pragma unroll
      for (int x = 0; x < 8; ++x)
        __hip_atomic_store(a.queue + x * QUEUE_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(a.queue + QUEUE_DONE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (FUSED && a.retry_mode && a.status_out) {
        // what the caller may read behind the call (cvm_fold_update_ex): every workgroup of this launch has
        // made its last fetch, and an item that gave up here counted itself before its workgroup's exit add
        const int redone = __hip_atomic_load(a.fused_status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int lost = __hip_atomic_load(a.fused_status + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int st = lost ? 1 : (redone ? 2 : 0);
        if (st) {
          // (1 outranks 2: a poisoned batch stays poisoned whatever the other batches of the call report)
          int cur = __hip_atomic_load(a.status_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          while (cur != 1 && cur != st) {
            if (__hip_atomic_compare_exchange_strong(a.status_out, &cur, st, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
          }
        }
      }
    }
  }
#ifdef CVM_STAMPS
  {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) {
      unsigned long long *o = g_stamps2 + ((size_t)blockIdx.x * 8 + wave_all) * 4;
      o[0] = c1 - c0; o[1] = q1 - q0; o[2] = q0; o[3] = q1;
    }
  }
#endif
}
