// mid_chain.hpp -- part of libcvmhip.so (included by cvmhip.hip inside its anonymous namespace).
// mid_chain_kernel (round 5): mid_tile_kernel's work, with a workgroup walking a CHAIN of tiles of one fold.
//
// What the stamps of mid_tile_kernel said (tools/mid_probe.hip, profiles/r5/mid_tile/probe_ablation.txt; C3 rows in
// 1000 folds of 100 rows, cycles of one workgroup, four workgroups per CU): row numbers 5.4 k | statistics, first
// stage, weights 12.1 k | stage loop 21.8 k | G + dump 5.0 k | finish 8.2 k | mirror 3.0 k = 55.5 k, of which the
// matrix cores work 6.4 k per wave.  A third of a workgroup's life is a chain of three dependent memory round
// trips (offsets -> row numbers -> rows / weights) before the first MFMA, and the first stage of a tile is never
// ahead of its use.  Here a work item is up to `chmax` tiles (ti, tj0 .. tj0 + cnt - 1) of one 64-column row
// panel of one fold:
//   * the prologue -- the fold's row numbers and weights, the row panel's statistics -- is paid once per chain;
//   * the stage stream runs THROUGH the tile boundaries: the first stage of the next tile is requested at the top
//     of the current tile's last stage, into the stage buffer that stage does not read, and lands while the
//     current tile is finished;
//   * so that it can, the finish needs only ONE stage buffer: the tile is dumped and finished in two halves of 32
//     rows (pitch 65: 16.6 KB, the size a stage buffer is padded to), each half direct + mirrored;
//   * the G pieces of a half are requested before the accumulators that need them are dumped (the first half's
//     in front of the last stage's MFMAs).
// Everything else is mid_tile_kernel's: items dealt to the XCDs in contiguous ranges of folds, operands by LDS-DMA
// with per-lane addresses, 16-row stages through two buffers, one LDS-only barrier per stage, the same MFMA
// sequence per accumulator and the same finishing arithmetic -- the same bits as mid_tile_kernel.
// Limits (else mid_tile_kernel): XTX wanted, M <= 16 when XTY is wanted (no XTY-only items), statistics from the
// pre-pass.
#pragma once

template <typename T> constexpr size_t chain_buf_bytes() {
  const size_t stage = (size_t)MID_STAGE_ELEMS * sizeof(T);          // 16 rows x (64 + 64) columns
  const size_t half = (size_t)32 * 65 * sizeof(T);                   // half a tile, pitch 65
  return ((stage > half ? stage : half) + 15) / 16 * 16;
}
template <typename T> inline size_t chain_lds_bytes(int maxn) {
  return 2 * chain_buf_bytes<T>() + 2 * 256 * 8 + (size_t)maxn * sizeof(T) + (size_t)maxn * 4;
}
// chains of one fold: row panel ti has nt - ti tiles, cut into ceil((nt - ti) / chmax) chains of even length
inline int chain_items_per_fold(int nt, int chmax) {
  int n = 0;
  for (int ti = 0; ti < nt; ++ti) n += (nt - ti + chmax - 1) / chmax;
  return n;
}

// all but (at most) the n youngest vector-memory operations of this wave are done, n wave-uniform: in steps of
// four (a twelve-way branch on the scalar unit; up to three operations more than asked for are waited for)
__device__ __forceinline__ void chain_wait_vmcnt_le(int n) {
#define CVM_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n & ~3) {
    CVM_W(0) CVM_W(4) CVM_W(8) CVM_W(12) CVM_W(16) CVM_W(20) CVM_W(24) CVM_W(28) CVM_W(32) CVM_W(36) CVM_W(40)
    default: asm volatile("s_waitcnt vmcnt(44)" ::: "memory"); break;
  }
#undef CVM_W
}

template <typename T, bool WEIGHTED>
__global__ __launch_bounds__(MID_THREADS, CVM_MID_WPE) void mid_chain_kernel(const MidArgs a) {
  typedef typename MF<T>::acc_t acc_t;
  constexpr int ES = (int)sizeof(T), EPL = 16 / ES;
  constexpr int SR = 16, KPS = SR / 4;
  constexpr int LPR = 64 / EPL;      // lanes per row of a 64-column panel (32 / 16)
  constexpr int RPI = 64 / LPR;      // panel rows per DMA instruction (2 / 4)
  constexpr int RPW = SR / 4;        // panel rows per wave and stage
  constexpr int IPW = RPW / RPI;     // instructions per wave, panel and stage (2 / 1)
  constexpr int LY = 16 / EPL;       // lanes per row of a 16-column Y tile (8 / 4)
  constexpr int RPY = 64 / LY;       // Y tile rows per instruction (8 / 16)
  constexpr int IY = SR >= RPY ? SR / RPY : 1;
  constexpr int VW = EPL;            // elements of a 16-byte piece
  constexpr int LPD = 64 / VW;       // lanes per tile row in the direct finish (32 / 16): VW rows per instruction
  constexpr int JD = 8 / VW;         // direct-finish instructions per wave and half (8 rows per wave): 4 / 2
  constexpr int LPM = 32 / VW;       // lanes per mirrored row of a half (16 / 8)
  constexpr int RPM = 64 / LPM;      // mirrored rows per instruction (4 / 8)
  constexpr int JM = 16 / RPM;       // mirror instructions per wave and half (16 mirrored rows per wave): 4 / 2
  static_assert(MID_SR == 16, "mid_chain_kernel stages sixteen rows");
  typedef T vt __attribute__((ext_vector_type(VW)));
  constexpr size_t BUFB = chain_buf_bytes<T>();
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int bid = blockIdx.x;
  const long long item = (long long)(bid & 7) * a.per_xcd + (bid >> 3);
  if (item >= a.n_items) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
  const int dbg = MID_DBG(a);
  MID_STAMP(0);
  const int K = a.K, M = a.M;
  const int f = (int)(item / a.ipf);
  int q = (int)(item - (long long)f * a.ipf);
  // the chain: row panel ti, tiles tj0 .. tj0 + cnt - 1
  int ti = 0, tj0 = 0, cnt = 0;
  for (;; ++ti) {
    const int len = a.nt - ti, ch = (len + a.chmax - 1) / a.chmax;
    if (q < ch) {
      const int base = len / ch, extra = len - base * ch;
      cnt = base + (q < extra ? 1 : 0);
      tj0 = ti + q * base + (q < extra ? q : extra);
      break;
    }
    q -= ch;
  }
  ti = uni(ti); tj0 = uni(tj0); cnt = uni(cnt);
  const int wr = wave >> 1, wc = wave & 1;
  const bool want_xty = a.out_XTY != nullptr && M > 0;
  const int a0 = ti * 64;

  const int64_t rbeg = a.offs[a.seg0 + f];
  const int n = (int)(a.offs[a.seg0 + f + 1] - rbeg);
  const int nks = (n + 3) >> 2, nst = (nks + KPS - 1) / KPS;
  // LDS: two stage buffers (a half tile fits one) | rs: [0,64) row means, [64,128) row reciprocal stds, [128,192) /
  // [192,256) the same for the current tile's columns | sq: the XTY block of a diagonal tile: [0,64) row means,
  // [64,128) response means, [128,192) response reciprocal stds | weights | row numbers
  double *rs = reinterpret_cast<double *>(smem_raw + 2 * BUFB);
  double *sq = rs + 256;
  T *wl = reinterpret_cast<T *>(sq + 256);
  int *rowl = reinterpret_cast<int *>(wl + a.maxn);
  const double *fs = a.fstats + (size_t)f * fstat_len(K, M);
  const bool cX = a.flags & CVM_CENTER_X, cY = a.flags & CVM_CENTER_Y;
  const bool sX = a.flags & CVM_SCALE_X, sY = a.flags & CVM_SCALE_Y;
  const size_t fo = (size_t)(a.seg0 + f);
  const int which = tid >> 6, wc64 = tid & 63;
  const bool head_diag = tj0 == ti;                    // only the first tile of a chain can be a diagonal tile
  const int ny = (want_xty && head_diag) ? (M < 16 ? M : 16) : 0;

  // ---- prologue: the fold's row numbers, the row panel's statistics (the weights: behind the first stage's DMAs) ----
  const int npad = nst * SR;
  for (int r = tid; r < npad; r += MID_THREADS) rowl[r] = r < n ? (int)a.idx[rbeg + r] : 0;
  MID_STAMP(1);
  {
    const int xrow = a0 + wc64;
    if (which < 2) {
      double v = which ? 1.0 : 0.0;
      if (xrow < K) {
        if (which == 0 && cX) v = fs[xrow];
        if (which == 1 && sX) v = fs[K + xrow];
      }
      rs[tid] = v;
    }
    if (which == 0) sq[wc64] = ((cX || cY) && xrow < K) ? fs[xrow] : 0.0;
    if (which == 2) sq[64 + wc64] = ((cX || cY) && wc64 < ny) ? fs[2 * K + wc64] : 0.0;
    if (which == 3) sq[128 + wc64] = (sY && wc64 < ny) ? fs[2 * K + M + wc64] : 1.0;
  }
  const double swt = fs[2 * K + 2 * M];
  __syncthreads();

  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)smem_raw);
  const char *zero = reinterpret_cast<const char *>(g_zero_line);
  const T *Xp = reinterpret_cast<const T *>(a.X);
  const T *Yp = reinterpret_cast<const T *>(a.Y);
  // LDS-DMA of stage s of the tile with column panel tj into buffer `buf`; the Y tile of a diagonal tile by wave 2
  auto issue = [&](int tj, int s, int buf) {
    if ((dbg & 4) && (s > 0 || tj > tj0)) return;
    const bool dg = tj == ti;
    const int b0 = tj * 64;
    const unsigned bufb = lds0 + (unsigned)(buf * (int)BUFB);
    // (the lane number through an opaque move, here and in the finish: what is derived from it is a handful of
    //  integer operations, and hoisted out of the tile loop it is registers held for the whole chain -- the
    //  kernel then spills, and a reload inside the stage loop waits for every LDS-DMA in flight)
    int ln = lane;
    asm volatile("" : "+v"(ln));
#pragma unroll
    for (int qi = 0; qi < IPW; ++qi) {
      const int lrow = RPW * wave + RPI * qi;            // first stage row of this instruction
      if (SR * s + (lrow & ~3) >= n) continue;           // its k-step holds no row: never read
      const int rl = lrow + ln / LPR, gr = SR * s + rl, piece = ln % LPR;
      const bool valid = gr < n;
      const int64_t rn = valid ? (int64_t)rowl[gr] : 0;
      int ca = a0 + EPL * piece;
      if (ca > K - EPL) ca = K - EPL;
      const char *srcA = valid ? reinterpret_cast<const char *>(Xp + rn * (int64_t)K + ca) : zero + 16 * piece;
      dma16_lanes(srcA, (unsigned)uni((int)(bufb + (unsigned)(lrow * 64 * ES))));
      if (!dg) {
        int cb = b0 + EPL * piece;
        if (cb > K - EPL) cb = K - EPL;
        const char *srcB = valid ? reinterpret_cast<const char *>(Xp + rn * (int64_t)K + cb) : zero + 16 * piece;
        dma16_lanes(srcB, (unsigned)uni((int)(bufb + (unsigned)((SR * 64 + lrow * 64) * ES))));
      }
    }
    if (dg && wave == 2 && want_xty) {   // (a Y tile is loaded by the wave that multiplies with it)
#pragma unroll
      for (int qi = 0; qi < IY; ++qi) {
        const int lrow0 = RPY * qi;
        if (SR * s + (lrow0 & ~3) >= n) continue;
        const int rl = lrow0 + ln / LY, gr = SR * s + rl, piece = ln % LY;
        const bool valid = gr < n && rl < SR;
        const int64_t rn = valid ? (int64_t)rowl[gr] : 0;
        int cy = EPL * piece;
        if (cy > M - EPL) cy = M - EPL;
        const char *src = valid ? reinterpret_cast<const char *>(Yp + rn * (int64_t)M + cy) : zero + 16 * piece;
        dma16_lanes(src, (unsigned)uni((int)(bufb + (unsigned)((SR * 64 + lrow0 * 16) * ES))));
      }
    }
  };

  const int lk = lane >> 4, lc = lane & 15;
  const T *Gt = reinterpret_cast<const T *>(a.G);
  T *outp = reinterpret_cast<T *>(a.out_XTX) + fo * (size_t)K * K;

  issue(tj0, 0, 0);
  for (int r = tid; r < npad; r += MID_THREADS)
    wl[r] = r < n ? (WEIGHTED ? reinterpret_cast<const T *>(a.w)[rowl[r]] : (T)1) : (T)0;

  int g = 0;                                    // stages done so far: stage g lies in buffer g & 1
  int young = 0;                                // vector-memory operations this wave has issued since its last LDS-DMA
#pragma unroll 1
  for (int t = 0; t < cnt; ++t) {
    const int tj = tj0 + t, b0 = tj * 64;
    const bool diag = tj == ti;
    // what this wave computes: 1 = a 32 x 32 block of the tile, 2 = the panel's 64 x 16 piece of XTY, 0 = nothing
    int role = 1;
    if (diag && wave == 2) role = want_xty ? 2 : 0;
    if (role == 1 && (a0 + 32 * wr >= K || b0 + 32 * wc >= K)) role = 0;
    role = uni(role);
    // the column panel's statistics: requested now, put into LDS behind the loop
    double colv = (which & 1) ? 1.0 : 0.0;
    if (which >= 2 && b0 + wc64 < K) {
      if (which == 2 && cX) colv = fs[b0 + wc64];
      if (which == 3 && sX) colv = fs[K + b0 + wc64];
    }
    acc_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (acc_t){0, 0, 0, 0};
    vt gv[JD];
    auto g_request = [&](int h) {
      int lf = lane;
      asm volatile("" : "+v"(lf));
      const int sub = lf / LPD, lcc = VW * (lf - sub * LPD);
      const int gc = b0 + lcc;
      const bool col_ok = gc < K;
      if (dbg & 2) {
#pragma unroll
        for (int j = 0; j < JD; ++j) gv[j] = (vt)(T)0;
        return;
      }
      young += JD;
#pragma unroll
      for (int j = 0; j < JD; ++j) {
        const int gr = a0 + 32 * h + 8 * wave + VW * j + sub;
        gv[j] = *reinterpret_cast<const vt *>(Gt + (size_t)(gr < K ? gr : 0) * K + (col_ok ? gc : 0));
      }
    };
    const int a_off = 32 * wr + lc;
    const int b_off = (diag ? 0 : SR * 64) + 32 * wc + lc;
    const int y_off = SR * 64 + lc;
#pragma unroll 1
    for (int s = 0; s < nst; ++s, ++g) {
      // stage s of this tile was requested a stage ago: it is all this wave has in flight -- but for the first
      // stage of a later tile of the chain, which was requested in front of the last tile's finish: the G
      // pieces and output stores of that finish (`young` of them, counted where they were issued) are younger
      // and need not be waited for
      if (s == 0 && t > 0) chain_wait_vmcnt_le(young);
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
      {
        // the next stage of the stream: of this tile, or the first of the chain's next tile
        const bool last = s + 1 >= nst;
        if (!last || t + 1 < cnt) issue(last ? tj + 1 : tj, last ? 0 : s + 1, (g + 1) & 1);
      }
      young = 0;
      if (t == 0 && s == 0) MID_STAMP(2);
      const T *buf = reinterpret_cast<const T *>(smem_raw + (size_t)(g & 1) * BUFB);
      const T *wst = wl + SR * s;
      const int nk = nks - KPS * s;
      if (dbg & 8) continue;
      if (role == 1) {
#pragma unroll
        for (int ks = 0; ks < KPS; ++ks) {
          if (ks >= nk) break;                           // (wave-uniform)
          const int r = 4 * ks + lk;
          const T f0 = buf[r * 64 + a_off], f1 = buf[r * 64 + a_off + 16];
          const T g0 = buf[r * 64 + b_off], g1 = buf[r * 64 + b_off + 16];
          const T wv = WEIGHTED ? wst[r] : (T)1;
          const T a0v = WEIGHTED ? (T)(f0 * wv) : f0, a1v = WEIGHTED ? (T)(f1 * wv) : f1;
          acc[0] = MF<T>::mfma(a0v, g0, acc[0]);
          acc[1] = MF<T>::mfma(a0v, g1, acc[1]);
          acc[2] = MF<T>::mfma(a1v, g0, acc[2]);
          acc[3] = MF<T>::mfma(a1v, g1, acc[3]);
        }
      } else if (role == 2) {
#pragma unroll
        for (int ks = 0; ks < KPS; ++ks) {
          if (ks >= nk) break;
          const int r = 4 * ks + lk;
          T af[4];
#pragma unroll
          for (int m = 0; m < 4; ++m) af[m] = buf[r * 64 + 16 * m + lc];
          const T yr = buf[r * 16 + y_off];
          const T yv = WEIGHTED ? (T)(yr * wst[r]) : yr;
#pragma unroll
          for (int m = 0; m < 4; ++m) acc[m] = MF<T>::mfma(af[m], yv, acc[m]);
        }
      }
    }
    g_request(0);                                        // (in flight across the barrier and the dump)
    if (t == 0) MID_STAMP(3);
    if (dbg & 16) continue;
    // ---- finish: the last stage's buffer is free once every wave has left the loop ----------------------------
    lds_barrier();
    T (*Th)[65] = reinterpret_cast<T (*)[65]>(smem_raw + (size_t)((g - 1) & 1) * BUFB);
    if (which >= 2) rs[tid] = colv;                      // (read behind the next barrier)
    if (role == 2) {
      // XTY piece straight from the accumulators (cvmatrix.py:1001-1010 for XTY); the block's statistics have
      // been in LDS since the prologue
      T *out = reinterpret_cast<T *>(a.out_XTY) + fo * (size_t)K * M;
      const T *Ht = reinterpret_cast<const T *>(a.H);
      // (the lane number through an opaque move: the sixteen address pairs below are invariant over the tiles of
      //  the chain, and hoisted out of the tile loop they cost 64 registers for its whole length)
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      const int col = lane_o & 15;
      // (all sixteen pieces of H requested before the first is used: one memory round trip, not sixteen)
      T hv[4][4];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = a0 + 16 * m + MF<T>::drow(lane_o, r);
          hv[m][r] = Ht[(size_t)(row < K ? row : 0) * M + (col < M ? col : 0)];
        }
      young += 16;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int lr = 16 * m + MF<T>::drow(lane_o, r), row = a0 + lr;
          const bool ok = row < K && col < M && !(dbg & 1);
          if (!__ballot(ok)) continue;
          ++young;
          if (ok) {
            double vv = (double)hv[m][r] - (double)acc[m][r];
            if (cX || cY) vv -= swt * (sq[lr] * sq[64 + col]);
            if (sX && sY) vv = vv * (rs[64 + lr] * sq[128 + col]);
            else if (sX) vv = vv * rs[64 + lr];
            else if (sY) vv = vv * sq[128 + col];
            out[(size_t)row * M + col] = (T)vv;
          }
        }
    }
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      int lf = lane;
      asm volatile("" : "+v"(lf));
      const int sub = lf / LPD, lcc = VW * (lf - sub * LPD);      // direct finish: row within the instruction, first column
      const int msub = lf / LPM, mcc = VW * (lf - msub * LPM);    // mirror: row within the instruction, first column of the half
      const int gc = b0 + lcc;
      const bool col_ok = gc < K;
      // the half's raw update into LDS: rows 32 h .. of the tile are held by the waves of block row h; the lower
      // left block of a DIAGONAL tile is nobody's (its wave computed XTY): it is the transpose of block (0, 1)
      if (role == 1 && wr == h) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              Th[16 * m + MF<T>::drow(lane, r)][32 * wc + 16 * nn + lc] = acc[m * 2 + nn][r];
      }
      if (diag && h == 1 && wave == 1 && role == 1) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              Th[16 * nn + lc][16 * m + MF<T>::drow(lane, r)] = acc[m * 2 + nn][r];
      }
      lds_barrier();
      if (t == 0 && h == 0) MID_STAMP(4);
      // (fused_finish_direct's arithmetic, finalize.hpp, on the requested pieces): wave w finishes rows 8 w .. of the half
      {
        double muc[VW], sdc[VW];
#pragma unroll
        for (int e = 0; e < VW; ++e) { muc[e] = rs[128 + lcc + e]; sdc[e] = rs[192 + lcc + e]; }
#pragma unroll
        for (int j = 0; j < JD; ++j) {
          const int hl = 8 * wave + VW * j + sub, lr = 32 * h + hl, gr = a0 + lr;
          const bool ok = col_ok && gr < K;
          if (!__ballot(ok)) continue;
          if (!(dbg & 1)) ++young;
          if (!ok) continue;
          const double mur = rs[lr], sdr = rs[64 + lr];
          vt vv;
#pragma unroll
          for (int e = 0; e < VW; ++e) {
            // (a diagonal tile: below the diagonal the value of the mirrored position -- the same half holds it --
            //  so that the output is symmetric to the bit, like mid_tile_kernel's)
            const int c = lcc + e;
            const double u = (double)((diag && lr > c && c >= 32 * h) ? Th[c - 32 * h][lr] : Th[hl][c]);
            double x = (double)gv[j][e] - u;
            if (cX) x -= swt * (mur * muc[e]);
            if (sX) x = x * (sdr * sdc[e]);
            vv[e] = (T)x;
          }
          if (!(dbg & 1)) out_store(reinterpret_cast<vt *>(outp + (size_t)gr * K + gc), vv);
          if (!diag) {
#pragma unroll
            for (int e = 0; e < VW; ++e) Th[hl][lcc + e] = vv[e];
          }
        }
      }
      if (h == 0) g_request(1);                          // (the first half's pieces are used up)
      lds_barrier();                                     // the finished values are parked (LDS only: the stores stay in flight)
      if (t == 0 && h == 0) MID_STAMP(5);
      if (!diag) {
        // mirrored store of the half: out[b0 + c][a0 + 32 h + r] = finished[r][c]; wave w the rows 16 w .. of the mirror
        const int gc2 = a0 + 32 * h + mcc;
        if (gc2 < K && !(dbg & 1)) {
#pragma unroll
          for (int j = 0; j < JM; ++j) {
            const int c = 16 * wave + RPM * j + msub, gr = b0 + c;
            if (!__ballot(gr < K)) continue;
            ++young;
            if (gr >= K) continue;
            vt vv;
#pragma unroll
            for (int e = 0; e < VW; ++e) vv[e] = Th[mcc + e][c];
            out_store(reinterpret_cast<vt *>(outp + (size_t)gr * K + gc2), vv);
          }
        }
        lds_barrier();                                   // (the half image is free: the next dump, or the next tile's second stage)
      }
      if (t == 0 && h == 0) MID_STAMP(6);
    }
  }
}
