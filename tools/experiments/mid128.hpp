// mid128.hpp -- an EXPERIMENT of round 5 (built by tools/mid_probe.hip; not part of libcvmhip.so): measured, bit-identical to
// mid_tile_kernel, and not faster where it would have mattered (profiles/r5/mid_tile/mid128_variant.txt).
// mid128_kernel (round 5, float64): training matrices of mid-size folds with 128 x 128 work items, eight waves that
// ALL compute, and TWO workgroups per CU.
//
// Why.  The two kernels the regime had are each bounded by their shape.  wgram4_kernel<.., FUSED> has the right
// tile -- 128 x 128: half the operand bytes per flop of a 64 x 64 tile, its stage loop runs at 0.87 of the MFMA
// peak -- but ONE workgroup owns the CU (128 accumulator registers per compute wave, a 148 KB ring), and an item's
// 8.8 k-cycle prologue and 23 k-cycle finish run with the matrix cores idle (profiles/r3/fused_epilogue_stamps.txt).
// mid_tile_kernel has the overlap -- four small workgroups per CU, one's finish under another's MFMAs -- but its
// 64 x 64 tiles need twice the LDS-DMA bytes and twice the barriers per flop, and its loop runs at 0.39-0.45 of the
// peak whatever the fold size (tools/mid_probe.hip, profiles/r5/mid_tile/).  Here:
//   * an item is one 128 x 128 upper tile (ti, tj) of one fold; eight waves, wave w = (wr, wc) owns the 32 x 64 block
//     (rows 32 wr, columns 64 wc): 2 x 4 MFMA tiles, 64 accumulator registers -- under 128 registers per wave, so four
//     waves per SIMD: two workgroups per CU, and 80 KB of LDS each (two 16-row stage buffers of [A panel | B panel |
//     weights], the tile's statistics) keep it so;
//   * no loader waves: every wave fetches two rows of a stage by LDS-DMA with SCALAR row bases (row numbers by scalar
//     loads, one loop-invariant register of column offsets: no vector instruction per row), one stage ahead;
//   * diagonal tile: the lower-left 64 x 64 block is the mirror image of the upper-right one -- its two waves compute
//     the panel's 128 x 16 piece of XTY instead (four MFMA tiles each, like everybody's k-step);
//   * finish: in four quarters of 32 rows through ONE stage buffer (pitch 129): dump, direct half (G pieces requested a
//     quarter ahead, fused_finish_direct's arithmetic, 1 KiB row stores), mirrored half; a diagonal tile takes what
//     lies below its diagonal from the transposed upper blocks, so the output is symmetric to the bit.
// The same MFMA sequence per output element and the same finishing arithmetic as mid_tile_kernel and the fused
// route: the same bits.  Limits: float64, rows of X in whole 16-byte pieces, M <= 16 and even when XTY is wanted,
// statistics from the pre-pass, 32-bit row numbers.
#pragma once

constexpr int M128_THREADS = 512;
constexpr size_t M128_BUF_BYTES = (size_t)BUF_ELEMS * 8;                       // one stage: A panel, B panel | Y tile, weights
constexpr size_t M128_LDS_BYTES = 2 * M128_BUF_BYTES + 512 * 8 + 256 * 8;       // + row / column statistics, the XTY block
static_assert(32 * 129 * 8 <= M128_BUF_BYTES, "a quarter of the tile fits one stage buffer");

// one LDS-DMA instruction with a scalar base, `lanes` lanes of 16 bytes (8: a 16-column Y row; 64: a panel row)
__device__ __forceinline__ void m128_dma16(const char *sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void m128_dma16_lo8(const char *sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  unsigned long long ex;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b64 exec, 0xff\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep), "=&s"(ex) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void m128_dma4_lo2(const char *sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  unsigned long long ex;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b64 exec, 3\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
               "global_load_lds_dword %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
               : "=&s"(keep), "=&s"(ex) : "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}

template <bool WEIGHTED>
__global__ __launch_bounds__(M128_THREADS, 4) void mid128_kernel(const MidArgs a) {
  typedef double T;
  typedef MF<T>::acc_t acc_t;
  typedef T vt __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int bid = blockIdx.x;
  const long long item = (long long)(bid & 7) * a.per_xcd + (bid >> 3);
  if (item >= a.n_items) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
  const int dbg = MID_DBG(a);
  const int K = a.K, M = a.M;
  const int f = (int)(item / a.ipf);
  int ti = 0, tj;
  {
    int rem = (int)(item - (long long)f * a.ipf);
    while (rem >= a.nt - ti) { rem -= a.nt - ti; ++ti; }
    tj = ti + rem;
  }
  ti = uni(ti); tj = uni(tj);
  const bool diag = ti == tj;
  const int wr = wave >> 1, wc = wave & 1;
  const bool want_xty = a.out_XTY != nullptr && M > 0;
  const int a0 = ti * TILE, b0 = tj * TILE;
  // what this wave computes: 1 = its 32 x 64 block, 2 = rows 64 (wr - 2) .. of the panel's 128 x 16 piece of XTY, 0 = nothing
  int role = 1;
  if (diag && wr >= 2 && wc == 0) role = want_xty ? 2 : 0;
  if (role == 1 && (a0 + 32 * wr >= K || b0 + 64 * wc >= K)) role = 0;
  if (role == 2 && a0 + 64 * (wr - 2) >= K) role = 0;
  role = uni(role);

  const int64_t rbeg = a.offs[a.seg0 + f];
  const int n = (int)(a.offs[a.seg0 + f + 1] - rbeg);
  const int nks = (n + 3) >> 2, nst = (nks + 3) >> 2;
  T *ring = reinterpret_cast<T *>(smem_raw);
  // rs: [0,128) row means, [128,256) row reciprocal stds, [256,384) / [384,512) the same for the columns
  // sq: the XTY block of a diagonal tile: [0,128) row means, [128,144) response means, [144,160) response reciprocal stds
  double *rs = reinterpret_cast<double *>(smem_raw + 2 * M128_BUF_BYTES);
  double *sq = rs + 512;
  const double *fs = a.fstats + (size_t)f * fstat_len(K, M);
  const bool cX = a.flags & CVM_CENTER_X, cY = a.flags & CVM_CENTER_Y;
  const bool sX = a.flags & CVM_SCALE_X, sY = a.flags & CVM_SCALE_Y;
  const size_t fo = (size_t)(a.seg0 + f);
  {
    const int which = tid >> 7, i = tid & 127, col = ((which < 2) ? a0 : b0) + i;
    double v = (which & 1) ? 1.0 : 0.0;
    if (col < K) {
      if (!(which & 1) && cX) v = fs[col];
      if ((which & 1) && sX) v = fs[K + col];
    }
    rs[tid] = v;
    if (diag && want_xty) {
      if (tid < 128) sq[tid] = ((cX || cY) && a0 + tid < K) ? fs[a0 + tid] : 0.0;
      else if (tid < 144) sq[tid] = ((cX || cY) && tid - 128 < M) ? fs[2 * K + tid - 128] : 0.0;
      else if (tid < 160) sq[tid] = (sY && tid - 144 < M) ? fs[2 * K + M + tid - 144] : 1.0;
    }
  }
  const double swt = fs[2 * K + 2 * M];

  // ---- LDS-DMA of one stage: this wave's two rows (A piece, B piece or Y row, weight) ------------------------------
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)smem_raw);
  const char *zero = reinterpret_cast<const char *>(unip(g_zero_line));
  const char *one_src = reinterpret_cast<const char *>(unip(g_one_line));
  const T *Xp = reinterpret_cast<const T *>(a.X);
  const T *Yp = reinterpret_cast<const T *>(a.Y);
  const T *Wp = reinterpret_cast<const T *>(a.w);
  unsigned va, vb, vy;
  {
    int oa = 2 * lane, ob = 2 * lane, oy = 2 * (lane & 7);
    if (a0 + oa > K - 2) oa = K - 2 - a0;
    if (b0 + ob > K - 2) ob = K - 2 - b0;
    if (oa < 0) oa = 0;
    if (ob < 0) ob = 0;
    if (oy > M - 2) oy = M - 2;
    if (oy < 0) oy = 0;
    va = 8u * (unsigned)oa; vb = 8u * (unsigned)ob; vy = 8u * (unsigned)oy;
  }
  const unsigned vw = 4u * (unsigned)lane;
  const bool loadY = diag && want_xty;
  // (the row numbers of a stage are scalar loads made a stage BEFORE they are used: no load latency between the
  //  barrier and the LDS-DMAs)
  int64_t rnx[2] = {0, 0};
  auto row_numbers = [&](int s) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = 16 * s + 2 * wave + j;
      rnx[j] = r < n ? a.idx[rbeg + r] : 0;               // (wave-uniform: a scalar load)
    }
  };
  auto issue = [&](int s) {
    if ((dbg & 4) && s > 0) return;
    const unsigned bufb = lds0 + (unsigned)((s & 1) * (int)M128_BUF_BYTES);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int lrow = 2 * wave + j, r = 16 * s + lrow;
      if (16 * s + (lrow & ~3) >= n) continue;            // its k-step holds no row: never read
      const bool valid = r < n;
      const int64_t rn = rnx[j];
      const char *xrow = reinterpret_cast<const char *>(Xp + rn * (int64_t)K);
      m128_dma16(valid ? xrow + 8 * (int64_t)a0 : zero, va, bufb + (unsigned)(lrow * PITCH) * 8u);
      if (!diag) m128_dma16(valid ? xrow + 8 * (int64_t)b0 : zero, vb, bufb + (unsigned)(PANEL_ELEMS + lrow * PITCH) * 8u);
      else if (loadY)
        m128_dma16_lo8(valid ? reinterpret_cast<const char *>(Yp + rn * (int64_t)M) : zero, vy,
                       bufb + (unsigned)(PANEL_ELEMS + lrow * YPITCH) * 8u);
      m128_dma4_lo2(valid ? (WEIGHTED ? reinterpret_cast<const char *>(Wp + rn) : one_src) : zero, vw,
                    bufb + (unsigned)(2 * PANEL_ELEMS + lrow) * 8u);
    }
  };

  const int lk = lane >> 4, lc = lane & 15;
  acc_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (acc_t){0, 0, 0, 0};
  row_numbers(0);
  issue(0);
  row_numbers(1);
  const int a_off = 32 * wr + lc;
  const int b_off = (diag ? 0 : PANEL_ELEMS) + 64 * wc + lc;
  const int x_off = 64 * (wr - 2) + lc;                    // (XTY waves: the panel's columns 64 (wr - 2) ..)
#pragma unroll 1
  for (int s = 0; s < nst; ++s) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // stage s (requested a stage ago) is all this wave has in flight
    lds_barrier();
    if (s + 1 < nst) issue(s + 1);
    row_numbers(s + 2);
    if (dbg & 8) continue;
    const T *buf = ring + (size_t)(s & 1) * BUF_ELEMS;
    const int nk = nks - 4 * s;
    if (role == 1) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks >= nk) break;                               // (wave-uniform)
        const int r = 4 * ks + lk;
        const T wv = buf[2 * PANEL_ELEMS + r];
        const T f0 = buf[r * PITCH + a_off], f1 = buf[r * PITCH + a_off + 16];
        T g[4];
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) g[nn] = buf[r * PITCH + b_off + 16 * nn];
        const T a0v = f0 * wv, a1v = f1 * wv;
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) acc[nn] = MF<T>::mfma(a0v, g[nn], acc[nn]);
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) acc[4 + nn] = MF<T>::mfma(a1v, g[nn], acc[4 + nn]);
      }
    } else if (role == 2) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        if (ks >= nk) break;
        const int r = 4 * ks + lk;
        T af[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) af[m] = buf[r * PITCH + x_off + 16 * m];
        const T yv = buf[PANEL_ELEMS + r * YPITCH + lc] * buf[2 * PANEL_ELEMS + r];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = MF<T>::mfma(af[m], yv, acc[m]);
      }
    }
  }
  if (dbg & 16) return;
#ifdef CVM_M128_FINISH_PRIO
  __builtin_amdgcn_s_setprio(CVM_M128_FINISH_PRIO);
#endif
  // ---- finish -----------------------------------------------------------------------------------------------------
  const T *Gt = reinterpret_cast<const T *>(a.G);
  T *outp = reinterpret_cast<T *>(a.out_XTX) + fo * (size_t)K * K;
  vt gv[4];
  // (the lane number through an opaque move at the head of every phase: what is derived from it is a few integer
  //  operations, and hoisted to the top of the kernel it is registers held across the stage loop -- the kernel spills)
  auto g_request = [&](int q) {
    int lg = lane;
    asm volatile("" : "+v"(lg));
    const int lcc = 2 * lg, gc = b0 + lcc;
    const bool col_ok = gc < K;
    if (dbg & 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) gv[j] = (vt)(T)0;
      return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gr = a0 + 32 * q + 4 * wave + j;
      gv[j] = *reinterpret_cast<const vt *>(Gt + (size_t)(gr < K ? gr : 0) * K + (col_ok ? gc : 0));
    }
  };
  g_request(0);
  lds_barrier();                                           // every wave has left the loop: both stage buffers are free
  if (role == 2) {
    // XTY piece straight from the accumulators (cvmatrix.py:1001-1010 for XTY); all sixteen pieces of H first
    T *out = reinterpret_cast<T *>(a.out_XTY) + fo * (size_t)K * M;
    const T *Ht = reinterpret_cast<const T *>(a.H);
    int lo = lane;
    asm volatile("" : "+v"(lo));
    const int col = lo & 15, rb = 64 * (wr - 2);
    T hv[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = a0 + rb + 16 * m + MF<T>::drow(lo, r);
        hv[m][r] = Ht[(size_t)(row < K ? row : 0) * M + (col < M ? col : 0)];
      }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lr = rb + 16 * m + MF<T>::drow(lo, r), row = a0 + lr;
        if (row < K && col < M && !(dbg & 1)) {
          double vv = (double)hv[m][r] - (double)acc[m][r];
          if (cX || cY) vv -= swt * (sq[lr] * sq[128 + col]);
          if (sX && sY) vv = vv * (rs[128 + lr] * sq[144 + col]);
          else if (sX) vv = vv * rs[128 + lr];
          else if (sY) vv = vv * sq[144 + col];
          out[(size_t)row * M + col] = (T)vv;
        }
      }
  }
  if (!a.out_XTX) return;
  T (*Th)[129] = reinterpret_cast<T (*)[129]>(smem_raw);
#pragma unroll 1
  for (int q = 0; q < 4; ++q) {
    int lf = lane;
    asm volatile("" : "+v"(lf));
    const int lcc = 2 * lf, gc = b0 + lcc;
    const bool col_ok = gc < K;
    const int msub = lf >> 4, mcc = 2 * (lf & 15);         // mirror: row within the instruction, first column of the quarter
    const int lcq = lf & 15;
    // the quarter's raw update into LDS (rows 32 q ..): its owners' blocks; a diagonal tile's columns below 32 q -- and the
    // block nobody computed -- are the transposes of the upper blocks that hold them
    if (role == 1 && wr == q) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
          if (diag && 64 * wc + 16 * nn < 32 * q) continue;      // (wave-uniform: filled from the transposes)
#pragma unroll
          for (int r = 0; r < 4; ++r) Th[16 * m + MF<T>::drow(lf, r)][64 * wc + 16 * nn + lcq] = acc[m * 4 + nn][r];
        }
    }
    if (diag && role == 1 && wr < q && wc == (q >> 1)) {
      // (its MFMA tile columns that lie in [32 q, 32 q + 32): tiles 2 (q & 1), 2 (q & 1) + 1 -- static register indices:
      //  a run-time index would put the accumulators in private memory)
      if (q & 1) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              Th[16 * nn + lcq][32 * wr + 16 * m + MF<T>::drow(lf, r)] = acc[m * 4 + 2 + nn][r];
      } else {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              Th[16 * nn + lcq][32 * wr + 16 * m + MF<T>::drow(lf, r)] = acc[m * 4 + nn][r];
      }
    }
    lds_barrier();
    // direct half: wave w the rows 4 w .. of the quarter, one 1 KiB row per instruction (fused_finish_direct's arithmetic)
    {
      double muc[2], sdc[2];
#pragma unroll
      for (int e = 0; e < 2; ++e) { muc[e] = rs[256 + lcc + e]; sdc[e] = rs[384 + lcc + e]; }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int hl = 4 * wave + j, lr = 32 * q + hl, gr = a0 + lr;
        if (!(col_ok && gr < K)) continue;
        const double mur = rs[lr], sdr = rs[128 + lr];
        vt vv;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int c = lcc + e;
          const double u = (double)((diag && lr > c && c >= 32 * q) ? Th[c - 32 * q][lr] : Th[hl][c]);
          double x = (double)gv[j][e] - u;
          if (cX) x -= swt * (mur * muc[e]);
          if (sX) x = x * (sdr * sdc[e]);
          vv[e] = (T)x;
        }
        if (!(dbg & 1)) out_store(reinterpret_cast<vt *>(outp + (size_t)gr * K + gc), vv);
        if (!diag) { Th[hl][lcc] = vv[0]; Th[hl][lcc + 1] = vv[1]; }
      }
    }
    if (q < 3) g_request(q + 1);
    lds_barrier();                                         // the finished values are parked (LDS only: the stores stay in flight)
    if (!diag) {
      // mirrored half: out[b0 + c][a0 + 32 q + r] = finished[r][c]; wave w the mirror rows 16 w ..
      const int gc2 = a0 + 32 * q + mcc;
      if (gc2 < K && !(dbg & 1)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = 16 * wave + 4 * j + msub, gr = b0 + c;
          if (gr >= K) continue;
          vt vv;
          vv[0] = Th[mcc][c]; vv[1] = Th[mcc + 1][c];
          out_store(reinterpret_cast<vt *>(outp + (size_t)gr * K + gc2), vv);
        }
      }
    }
    lds_barrier();                                         // the image is free for the next quarter
  }
}
