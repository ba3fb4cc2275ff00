// mid_tile.hpp -- part of libcvmhip.so (included by cvmhip.hip inside its anonymous namespace).
// mid_tile_kernel (round 4): training matrices of MID-SIZE folds (8 to ~300 validation rows, float64 and float32) with
// the Gram stage and the finishing step of DIFFERENT work items overlapping on every CU.
//
// Why another kernel.  The fused route of wgram4_kernel (one persistent 8-wave workgroup per CU, 128 x 128
// tiles) runs an item's phases one after the other: a cold prologue (8.5 k cycles), the MFMA stage loop
// (4.7 k per 16 rows), then an epilogue that reads G and stores the tile twice (22 k cycles) while the
// matrix cores idle -- profiles/r3/fused_epilogue_stamps.txt.  With folds of 100 rows the loop is half of
// an item's 64 k cycles, and nothing else is resident on the CU to fill the gaps (248 registers, 148 KB of
// LDS).  Here a work item is small -- one 64 x 64 tile of one fold, four waves, 85-127 registers, ~39 KB of LDS
// (float32: 93 / 21 KB) -- so that FOUR (float32: five) workgroups share a CU and the hardware interleaves one item's stores and G loads with
// another's MFMAs.  (What bounds it -- measured with five structures of this kernel, DESIGN.md section 4.4 -- is the
// CU's vector-memory pipe: X panels, G and both halves of the output, ~196 KB per tile at 100 rows, pass through it
// at ~27 GB/s per CU whatever issues them.)
//
//   * Items: for every fold the upper-triangle 64 x 64 tiles (i <= j) of XTX in row-major order (tiles that
//     share the row panel are neighbours), then -- only when M > 16 -- XTY-only items for the response
//     columns past the first sixteen.  Workgroup b takes item (b % 8) * per_xcd + b / 8: the hardware deals
//     workgroups to the XCDs round-robin, so an XCD works on a contiguous range of folds and its L2 serves
//     the 2 nt panel reads of every validation row and the re-reads of G (measured: 0.45 GB fetched from HBM
//     at P = 1000 where the fused route fetches 1.59 GB).
//   * Off-diagonal tile: the four waves form a 2 x 2 grid of 32 x 32 blocks (four MFMA tiles each).
//     Diagonal tile: the waves of blocks (0,0), (0,1), (1,1) do the same with the row panel on both sides;
//     the fourth wave -- its block is the mirror image of (0,1) -- computes the panel's 64 x 16 piece of
//     XTY instead (four MFMA tiles as well).  Every wave of every item issues 4 MFMAs per k-step.
//   * Operands go global -> LDS by LDS-DMA with per-lane source addresses (a gathered row is 512 bytes of
//     a 64-column panel: two rows per wave instruction), SR rows per stage through a ring of NBUF stage
//     buffers, one workgroup barrier per stage (LDS only: `s_waitcnt vmcnt(<DMAs issued since>)`, hand-counted).
//     The row numbers and weights of the whole fold are staged in LDS once (one level of dependent loads).
//     Only the k-steps that hold rows are computed (a 100-row fold: 25, not 28).
//   * Finish: the accumulators go to LDS (64 x 65, over the ring), each wave finishes sixteen rows of the
//     tile with fused_finish_direct's arithmetic (finalize.hpp: total - update, rank-1 centring,
//     reciprocal-std scaling, 16-byte nontemporal stores) and fused_finish_mirror -- the same arithmetic on
//     the same MFMA sums as the fused route.
//   * The per-fold statistics come from colstats_kernel + fold_stats_kernel (the pre-pass of host.hpp).  (Two
//     routes without the pre-pass were built, were green and measured SLOWER, and left the product in round 6 --
//     every item summing its own 128 staged columns on the vector units, 22 % slower at 1000 folds; the diagonal
//     tiles forming the statistics and handing them to the others behind flags, 7 % slower: the code is
//     tools/experiments/pruned_r6_routes.patch, the numbers profiles/r4/mid_tile/own_statistics.txt and
//     profiles/r5/mid_tile/inlaunch_statistics.txt.)
#pragma once

struct MidArgs {
  const void *X, *Y, *w;
  const int64_t *idx, *offs;
  int64_t seg0;                  // first fold of this batch in offs / the outputs
  const double *fstats;          // [fold of batch][fstat_len] from the pre-pass
  const void *G, *H;
  void *out_XTX, *out_XTY;
  long long n_items, per_xcd;    // work items; workgroups per XCD
  int K, M;
  int nt;                        // 64-column panels of X
  int n_xtx;                     // nt (nt + 1) / 2
  int yextra;                    // XTY-only items per panel (64 response columns each, past the first 16)
  int ipf;                       // items per fold
  int maxn;                      // rows the LDS lists hold (a multiple of 16, >= the longest fold)
  int chmax;                     // (tools/experiments/mid_chain.hpp: tiles per chain at most)
  unsigned flags;
  int nb;                        // folds of the batch
  // measurements only (tools/mid_probe.hip builds with -DCVM_MID_ABLATE; the library never sets them):
  // dbg bits: 1 no output stores, 2 no G loads, 4 no LDS-DMA after the first stages, 8 no MFMA, 16 return after the loop
  int dbg;
  unsigned long long *stamps;    // [512][8] cycle stamps of sampled workgroups
};
#ifdef CVM_MID_ABLATE
#define MID_DBG(a) ((a).dbg)
#define MID_STAMP(i)                                                                                  \
  do {                                                                                                \
    if (a.stamps && tid == 0 && (bid & 63) == 5 && (bid >> 6) < 512) {                                 \
      __builtin_amdgcn_sched_barrier(0);                                                              \
      a.stamps[(size_t)(bid >> 6) * 8 + (i)] = __builtin_amdgcn_s_memtime();                          \
      __builtin_amdgcn_sched_barrier(0);                                                              \
    }                                                                                                 \
  } while (0)
#else
#define MID_DBG(a) 0
#define MID_STAMP(i)
#endif
constexpr int MID_THREADS = 256;
#ifndef CVM_MID_SR
#define CVM_MID_SR 16            // rows per stage
#endif
#ifndef CVM_MID_NBUF
#define CVM_MID_NBUF 2           // stage buffers
#endif
#ifndef CVM_MID_WPE
#define CVM_MID_WPE 4            // workgroups per CU the registers are cut for
#endif
constexpr int MID_SR = CVM_MID_SR, MID_NBUF = CVM_MID_NBUF;
constexpr int MID_STAGE_ELEMS = MID_SR * 128;      // SR rows x (64 + 64) columns
template <typename T> constexpr size_t mid_region_bytes() {
  const size_t tile = ((size_t)64 * 65 * sizeof(T) + 15) / 16 * 16, ring = (size_t)MID_NBUF * MID_STAGE_ELEMS * sizeof(T);
  return tile > ring ? tile : ring;
}
template <typename T> inline size_t mid_lds_bytes(int maxn) {
  return mid_region_bytes<T>() + 2 * 256 * 8 + (size_t)maxn * sizeof(T) + (size_t)maxn * 4;
}

// one work item: fold f of the batch, item q of the fold
template <typename T, bool WEIGHTED>
__device__ __forceinline__ void mid_tile_item(const MidArgs &a, const int f, const int q, const int bid, char *smem_raw) {
  typedef typename MF<T>::acc_t acc_t;
  constexpr int ES = (int)sizeof(T), EPL = 16 / ES;
  constexpr int SR = MID_SR, NBUF = MID_NBUF, KPS = SR / 4;   // k-steps per stage
  constexpr int LPR = 64 / EPL;      // lanes per row of a 64-column panel (32 / 16)
  constexpr int RPI = 64 / LPR;      // panel rows per DMA instruction (2 / 4)
  constexpr int RPW = SR / 4;        // panel rows per wave and stage
  static_assert(RPW % RPI == 0 || RPI % RPW == 0, "stage rows");
  constexpr int IPW = RPW >= RPI ? RPW / RPI : 1;   // instructions per wave, panel and stage
  constexpr int LY = 16 / EPL;       // lanes per row of a 16-column Y tile (8 / 4)
  constexpr int RPY = 64 / LY;       // Y tile rows per instruction (8 / 16)
  constexpr int IY = SR >= RPY ? SR / RPY : 1;      // instructions per Y tile and stage
  constexpr int VW = 16 / ES, LPRO = 64 / VW, JB = 16 / VW;
  static_assert(RPW >= RPI, "a wave's rows of a stage fill whole DMA instructions");
  typedef T vt __attribute__((ext_vector_type(VW)));
  const int tid = threadIdx.x, lane = tid & 63, wave = uni(tid >> 6);
  const int dbg = MID_DBG(a);
  MID_STAMP(0);
  const int K = a.K, M = a.M;
  // kind 0: XTX tile (ti, tj); kind 2: XTY-only item (panel ti, response columns 16 + 64 yc ..)
  int ti = 0, tj = 0, yc = 0, kind = 0;
  if (q < a.n_xtx) {
    int rem = q;
    while (rem >= a.nt - ti) { rem -= a.nt - ti; ++ti; }
    tj = ti + rem;
  } else {
    kind = 2;
    ti = (q - a.n_xtx) / a.yextra;
    yc = (q - a.n_xtx) - ti * a.yextra;
    tj = ti;
  }
  const bool diag = kind == 0 && ti == tj;
  const int wr = wave >> 1, wc = wave & 1;
  const bool want_xty = a.out_XTY != nullptr && M > 0;
  // what this wave computes: 1 = a 32 x 32 block of the XTX tile, 2 = a 64 x 16 piece of XTY, 0 = nothing
  int role = 1, ycol0 = 0, yslot = 0;
  if (kind == 2) { role = 2; ycol0 = 16 + 64 * yc + 16 * wave; yslot = wave; }
  else if (diag && wave == 2) role = 2;
  if (role == 2 && (!want_xty || ycol0 >= M)) role = 0;
  const int a0 = ti * 64, b0 = tj * 64;
  if (role == 1 && (a0 + 32 * wr >= K || b0 + 32 * wc >= K)) role = 0;
  role = uni(role);

  const int64_t rbeg = a.offs[a.seg0 + f];
  const int n = (int)(a.offs[a.seg0 + f + 1] - rbeg);
  const int nks = (n + 3) >> 2, nst = (nks + KPS - 1) / KPS;
  T *ring = reinterpret_cast<T *>(smem_raw);
  // statistics blocks: rs = fused_finish_direct's ([0,64) row means, [64,128) row reciprocal stds, [128,256) the
  // same for the columns); sq = the column sums while they are being formed, then the XTY block:
  // [0,64) row means for XTY, [64,128) response means, [128,192) response reciprocal stds
  double *rs = reinterpret_cast<double *>(smem_raw + mid_region_bytes<T>());
  double *sq = rs + 256;
  T *wl = reinterpret_cast<T *>(sq + 256);
  int *rowl = reinterpret_cast<int *>(wl + a.maxn);
  T (*Ts)[65] = reinterpret_cast<T (*)[65]>(smem_raw);
  // the fold's statistics: from the pre-pass
  const double *fs = a.fstats + (size_t)f * fstat_len(K, M);
  const bool cX = a.flags & CVM_CENTER_X, cY = a.flags & CVM_CENTER_Y;
  const bool sX = a.flags & CVM_SCALE_X, sY = a.flags & CVM_SCALE_Y;
  const size_t fo = (size_t)(a.seg0 + f);
  const bool finish_xtx = kind == 0 && a.out_XTX != nullptr;
  // the response columns this item holds in LDS: a diagonal tile the first sixteen, an XTY-only item up to 64
  const int ybase = kind == 2 ? 16 + 64 * yc : 0;
  const int ny = !want_xty ? 0 : (kind == 2 ? (M - ybase < 64 ? M - ybase : 64) : (diag ? (M < 16 ? M : 16) : 0));
  const int which = tid >> 6, wc64 = tid & 63;

  // ---- the fold's row numbers, the tile's statistics (the weights: behind the first stages' DMAs) ----
  const int npad = nst * SR;
  for (int r = tid; r < npad; r += MID_THREADS) rowl[r] = r < n ? (int)a.idx[rbeg + r] : 0;
  MID_STAMP(1);
  double swt = 0.0;
  const int xcol = ((which < 2) ? a0 : b0) + wc64;        // (diagonal tile: b0 == a0)
  const int ycol = ybase + wc64;
  {
    double v = (which & 1) ? 1.0 : 0.0;
    if (xcol < K && (kind == 0 || which < 2)) {
      if (!(which & 1) && cX) v = fs[xcol];
      if ((which & 1) && sX) v = fs[K + xcol];
    }
    rs[tid] = v;
    if (which == 0) sq[wc64] = ((cX || cY) && xcol < K) ? fs[xcol] : 0.0;
    if (which == 2) sq[64 + wc64] = ((cX || cY) && wc64 < ny) ? fs[2 * K + ycol] : 0.0;
    if (which == 3) sq[128 + wc64] = (sY && wc64 < ny) ? fs[2 * K + M + ycol] : 1.0;
    swt = fs[2 * K + 2 * M];
  }
  __syncthreads();

  // ---- LDS-DMA of one stage; returns the number of instructions this wave issued ---------------------
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)smem_raw);
  const char *zero = reinterpret_cast<const char *>(g_zero_line);
  const T *Xp = reinterpret_cast<const T *>(a.X);
  const T *Yp = reinterpret_cast<const T *>(a.Y);
  const bool loadB = kind == 0 && !diag;
  auto issue = [&](int s) -> int {
    int cnt = 0;
    if ((dbg & 4) && s >= NBUF - 1) return 0;
    const unsigned bufb = lds0 + (unsigned)((s % NBUF) * MID_STAGE_ELEMS * ES);
#pragma unroll
    for (int qi = 0; qi < IPW; ++qi) {
      const int lrow = RPW * wave + RPI * qi;            // first stage row of this instruction
      if (SR * s + (lrow & ~3) >= n) continue;           // its k-step holds no row: never read
      const int rl = lrow + lane / LPR, gr = SR * s + rl, piece = lane % LPR;
      const bool valid = gr < n;
      const int64_t rn = valid ? (int64_t)rowl[gr] : 0;
      int ca = a0 + EPL * piece;
      if (ca > K - EPL) ca = K - EPL;
      const char *srcA = valid ? reinterpret_cast<const char *>(Xp + rn * (int64_t)K + ca) : zero + 16 * piece;
      dma16_lanes(srcA, (unsigned)uni((int)(bufb + (unsigned)(lrow * 64 * ES))));
      ++cnt;
      if (loadB) {
        int cb = b0 + EPL * piece;
        if (cb > K - EPL) cb = K - EPL;
        const char *srcB = valid ? reinterpret_cast<const char *>(Xp + rn * (int64_t)K + cb) : zero + 16 * piece;
        dma16_lanes(srcB, (unsigned)uni((int)(bufb + (unsigned)((SR * 64 + lrow * 64) * ES))));
        ++cnt;
      }
    }
    if (role == 2) {   // (a Y tile is loaded by the wave that multiplies with it)
#pragma unroll
      for (int qi = 0; qi < IY; ++qi) {
        const int lrow0 = RPY * qi;
        if (SR * s + (lrow0 & ~3) >= n) continue;
        const int rl = lrow0 + lane / LY, gr = SR * s + rl, piece = lane % LY;
        const bool valid = gr < n && rl < SR;
        const int64_t rn = valid ? (int64_t)rowl[gr] : 0;
        int cy = ycol0 + EPL * piece;
        if (cy > M - EPL) cy = M - EPL;
        const char *src = valid ? reinterpret_cast<const char *>(Yp + rn * (int64_t)M + cy) : zero + 16 * piece;
        dma16_lanes(src, (unsigned)uni((int)(bufb + (unsigned)((SR * 64 + yslot * SR * 16 + lrow0 * 16) * ES))));
        ++cnt;
      }
    }
    return uni(cnt);
  };

  const int lk = lane >> 4, lc = lane & 15;
  const int sub = lane / LPRO, lcc = VW * (lane - sub * LPRO), gc = b0 + lcc;
  const bool col_ok = gc < K;
  const T *Gt = reinterpret_cast<const T *>(a.G);
  T *outp = reinterpret_cast<T *>(a.out_XTX) + fo * (size_t)K * K;
  vt gv[JB];
  // the G pieces of this wave's sixteen rows of the tile: requested before the accumulators go to LDS
  auto g_preload = [&]() {
    if (!finish_xtx) return;
    if (dbg & 2) {
#pragma unroll
      for (int j = 0; j < JB; ++j) gv[j] = (vt)(T)0;
      return;
    }
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      const int gr = a0 + 16 * wave + VW * j + sub;
      gv[j] = *reinterpret_cast<const vt *>(Gt + (size_t)(gr < K ? gr : 0) * K + (col_ok ? gc : 0));
    }
  };
  // the first NBUF - 1 stages, then (one more level of dependent loads, in flight together with them) the weights
  int cnt[NBUF];
#pragma unroll
  for (int t = 0; t < NBUF; ++t) cnt[t] = 0;
#pragma unroll
  for (int t = 0; t < NBUF - 1; ++t)
    if (t < nst) cnt[t] = issue(t);
  for (int r = tid; r < npad; r += MID_THREADS)
    wl[r] = r < n ? (WEIGHTED ? reinterpret_cast<const T *>(a.w)[rowl[r]] : (T)1) : (T)0;
  // stage s: wait for its DMAs (the younger ones -- stages s + 1 .. s + NBUF - 2 -- may stay in flight), join,
  // refill the buffer stage s - 1 was read from
  auto stage_head = [&](int s) {
    int younger = 0;
#pragma unroll
    for (int t = 1; t < NBUF - 1; ++t) younger += cnt[(s + t) % NBUF];      // (0 for stages past the end)
    if (s == nst - 1) younger = 0;
    wait_vmcnt_le(younger);
    lds_barrier();
    cnt[(s + NBUF - 1) % NBUF] = (s + NBUF - 1 < nst) ? issue(s + NBUF - 1) : 0;
  };
  if (role == 2) {
    acc_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (acc_t){0, 0, 0, 0};
    const int y_off = SR * 64 + yslot * SR * 16 + lc;
    auto ksteps = [&](const T *buf, const T *wst, int nk) {
      T af[2][4], yf[2], wv[2] = {(T)1, (T)1};
      auto rd = [&](int ks, int c) {
        const int r = 4 * ks + lk;
#pragma unroll
        for (int m = 0; m < 4; ++m) af[c][m] = buf[r * 64 + 16 * m + lc];
        yf[c] = buf[r * 16 + y_off];
        if (WEIGHTED) wv[c] = wst[r];
      };
      rd(0, 0);
#pragma unroll
      for (int ks = 0; ks < KPS; ++ks) {
        const int c = ks & 1;
        if (ks >= nk) break;                           // (wave-uniform)
        if (ks + 1 < KPS) rd(ks + 1, c ^ 1);           // (past the fold's last k-step: read, never used)
        const T yv = WEIGHTED ? (T)(yf[c] * wv[c]) : yf[c];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = MF<T>::mfma(af[c][m], yv, acc[m]);
      }
    };
#pragma unroll 1
    for (int s = 0; s < nst; ++s) {
      stage_head(s);
      if (s == 0) MID_STAMP(2);
      if (!(dbg & 8)) ksteps(ring + (s % NBUF) * MID_STAGE_ELEMS, wl + SR * s, nks - KPS * s);
    }
    MID_STAMP(3);
    if (dbg & 16) return;
    g_preload();
    lds_barrier();                                       // the ring is free
    // XTY piece straight from the accumulators (cvmatrix.py:1001-1010 for XTY)
    T *out = reinterpret_cast<T *>(a.out_XTY) + fo * (size_t)K * M;
    const T *Ht = reinterpret_cast<const T *>(a.H);
    const int col = ycol0 + lc, yl = col - ybase;
    // (all sixteen pieces of H requested before the first is used: one memory round trip, not sixteen -- the other
    //  waves of a diagonal tile wait for this one at the next barrier)
    T hv[4][4];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = a0 + 16 * m + MF<T>::drow(lane, r);
        hv[m][r] = Ht[(size_t)(row < K ? row : 0) * M + (col < M ? col : 0)];
      }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lr = 16 * m + MF<T>::drow(lane, r), row = a0 + lr;
        if (row < K && col < M && !(dbg & 1)) {
          double vv = (double)hv[m][r] - (double)acc[m][r];
          if (cX || cY) vv -= swt * (sq[lr] * sq[64 + yl]);
          if (sX && sY) vv = vv * (rs[64 + lr] * sq[128 + yl]);
          else if (sX) vv = vv * rs[64 + lr];
          else if (sY) vv = vv * sq[128 + yl];
          out[(size_t)row * M + col] = (T)vv;
        }
      }
  } else {
    acc_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (acc_t){0, 0, 0, 0};
    const int a_off = 32 * wr + lc;
    const int b_off = (diag ? 0 : SR * 64) + 32 * wc + lc;
    auto ksteps = [&](const T *buf, const T *wst, int nk) {
      T af[2][2], bf[2][2], wv[2] = {(T)1, (T)1};
      auto rd = [&](int ks, int c) {
        const int r = 4 * ks + lk;
        af[c][0] = buf[r * 64 + a_off]; af[c][1] = buf[r * 64 + a_off + 16];
        bf[c][0] = buf[r * 64 + b_off]; bf[c][1] = buf[r * 64 + b_off + 16];
        if (WEIGHTED) wv[c] = wst[r];
      };
      rd(0, 0);
#pragma unroll
      for (int ks = 0; ks < KPS; ++ks) {
        const int c = ks & 1;
        if (ks >= nk) break;                           // (wave-uniform)
        if (ks + 1 < KPS) rd(ks + 1, c ^ 1);           // (past the fold's last k-step: read, never used)
        const T a0v = WEIGHTED ? (T)(af[c][0] * wv[c]) : af[c][0], a1v = WEIGHTED ? (T)(af[c][1] * wv[c]) : af[c][1];
        acc[0] = MF<T>::mfma(a0v, bf[c][0], acc[0]);
        acc[1] = MF<T>::mfma(a0v, bf[c][1], acc[1]);
        acc[2] = MF<T>::mfma(a1v, bf[c][0], acc[2]);
        acc[3] = MF<T>::mfma(a1v, bf[c][1], acc[3]);
      }
    };
#pragma unroll 1
    for (int s = 0; s < nst; ++s) {
      stage_head(s);
      if (s == 0) MID_STAMP(2);
      if (role == 1 && !(dbg & 8)) ksteps(ring + (s % NBUF) * MID_STAGE_ELEMS, wl + SR * s, nks - KPS * s);
    }
    MID_STAMP(3);
    if (dbg & 16) return;
    g_preload();
    lds_barrier();                                       // the ring is free: the tile goes over it
    if (finish_xtx) {
      if (role == 1) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int nn = 0; nn < 2; ++nn)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              Ts[32 * wr + 16 * m + MF<T>::drow(lane, r)][32 * wc + 16 * nn + lc] = acc[m * 2 + nn][r];
      }
    }
  }
  if (!finish_xtx) return;
  lds_barrier();                                         // the tile is in LDS
  MID_STAMP(4);
  // (fused_finish_direct's arithmetic, finalize.hpp, on the preloaded pieces)
  {
    double muc[VW], sdc[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) { muc[e] = rs[128 + lcc + e]; sdc[e] = rs[192 + lcc + e]; }
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      const int lr = 16 * wave + VW * j + sub, gr = a0 + lr;
      if (!(col_ok && gr < K)) continue;
      const double mur = rs[lr], sdr = rs[64 + lr];
      vt vv;
#pragma unroll
      for (int e = 0; e < VW; ++e) {
        const double u = (double)((diag && lr > lcc + e) ? Ts[lcc + e][lr] : Ts[lr][lcc + e]);
        double x = (double)gv[j][e] - u;
        if (cX) x -= swt * (mur * muc[e]);
        if (sX) x = x * (sdr * sdc[e]);
        vv[e] = (T)x;
      }
      if (!(dbg & 1)) out_store(reinterpret_cast<vt *>(outp + (size_t)gr * K + gc), vv);
      if (!diag) {
#pragma unroll
        for (int e = 0; e < VW; ++e) Ts[lr][lcc + e] = vv[e];
      }
    }
  }
  MID_STAMP(5);
  if (diag) return;
  lds_barrier();                                         // the finished values are parked in Ts (LDS only: the stores stay in flight)
  if (!(dbg & 1)) fused_finish_mirror<T, 65>(Ts, a0, b0, K, outp, lane, 16 * wave, 16 * wave + 16);
  MID_STAMP(6);
}

template <typename T, bool WEIGHTED>
__global__ __launch_bounds__(MID_THREADS, CVM_MID_WPE) void mid_tile_kernel(const MidArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int bid = blockIdx.x;
  const long long item = (long long)(bid & 7) * a.per_xcd + (bid >> 3);
  if (item >= a.n_items) return;
  const int f = (int)(item / a.ipf);
  const int q = (int)(item - (long long)f * a.ipf);
  mid_tile_item<T, WEIGHTED>(a, f, q, bid, smem_raw);
}
