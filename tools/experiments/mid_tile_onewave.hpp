// EXPERIMENT (not compiled into the library): the one-wave-per-tile variant of mid_tile_kernel with G / H travelling
// through the LDS ring and hand-counted store waits -- see tools/README.md (round 4) for what it measured.
// mid_tile.hpp -- part of libcvmhip.so (included by cvmhip.hip inside its anonymous namespace).
// mid_tile_kernel (round 4): training matrices of MID-SIZE folds (a few dozen to a few hundred validation
// rows) with the Gram stage and the finishing step of DIFFERENT work items overlapping on every CU.
//
// Why another kernel.  The fused route of wgram4_kernel (one persistent 8-wave workgroup per CU, 128 x 128
// tiles) runs an item's phases one after the other: a cold prologue (8.5 k cycles), the MFMA stage loop
// (4.7 k per 16 rows), then an epilogue that reads G and stores the tile twice (22 k cycles) while the
// matrix cores idle -- profiles/r3/fused_epilogue_stamps.txt.  With folds of 100 rows the loop is half of
// an item's 64 k cycles, and nothing else is resident on the CU to fill the gaps (248 registers, 148 KB of
// LDS).  Here ONE WAVE is a workgroup and computes one 64 x 64 tile of one fold by itself:
//
//   * no workgroup barrier anywhere -- a wave waits for its own LDS-DMA pieces only (hand-counted vmcnt);
//     a first version of this kernel with four waves per tile and a barrier per 16-row stage spent half its
//     time in those joins (same speed with 2, 3 or 4 stage buffers: not the DMA latency), and per item it
//     had a quarter of the MFMA work to set against the same fixed latencies (row numbers -> weights ->
//     first rows -> G -> stores);
//   * 16 MFMAs per k-step and wave against nine LDS fragment reads and four DMA instructions (4 rows x
//     (64 + 64) columns = 4 KB per k-step through a ring of four k-step slots);
//   * ~18 KB of LDS and < 256 registers per wave: eight independent tiles in flight per CU, two per SIMD --
//     one wave's finish (G loads, 160 stores) runs under the other's MFMAs.
//
//   * Items: for every fold the upper-triangle 64 x 64 tiles (i <= j) of XTX in row-major order (tiles that
//     share the row panel are neighbours), then -- only when M > 16 -- XTY-only items of up to 64 response
//     columns past the first sixteen.  Workgroup b takes item (b % 8) * per_xcd + b / 8: the hardware deals
//     workgroups to the XCDs round-robin, so an XCD works on a contiguous range of folds and its L2 serves
//     the 2 nt panel reads of every validation row and the re-reads of G.
//   * Off-diagonal tile: 4 x 4 MFMA tiles.  Diagonal tile: the ten MFMA tiles on and above the diagonal plus
//     the panel's 64 x 16 piece of XTY (four MFMA tiles): 14 MFMAs per k-step.
//   * Finish straight from the accumulators, with NO global load in it: the tile of G (and of H) travels
//     through the same LDS ring as eight more "k-steps" (8 rows x 64 columns each) whose DMAs are issued
//     while the last k-steps still compute -- a dependent global round trip costs 4-7 k cycles under this
//     load (tools/mid_stamps.py: four rounds of G loads were 35 k of an item's 106 k cycles, ten rounds 80 k
//     on a diagonal tile), a wait for data requested three steps ago costs nothing.  Neighbouring lanes
//     swap one value (DPP) so that a direct store carries 16 bytes per lane (eight rows x 128 bytes per
//     instruction); the mirrored half goes through a wave-private 16 x 16 LDS transposition (128-byte row
//     segments again).  Stores are never waited for: every wait is `s_waitcnt vmcnt(<DMAs and stores
//     issued since>)`, hand-counted.  Same arithmetic as fused_finish_direct (finalize.hpp).
//   * The per-fold statistics come from colstats_kernel + fold_stats_kernel (the pre-pass of host.hpp).
#pragma once

struct MidArgs {
  const void *X, *Y, *w;
  const int64_t *idx, *offs;
  int64_t seg0;                  // first fold of this batch in offs / the outputs
  const double *fstats;          // [fold of batch][fstat_len]
  const void *G, *H;
  void *out_XTX, *out_XTY;
  long long n_items, per_xcd;    // work items; workgroups per XCD
  int K, M;
  int nt;                        // 64-column panels of X
  int n_xtx;                     // nt (nt + 1) / 2
  int yextra;                    // XTY-only items per panel (64 response columns each, past the first 16)
  int ipf;                       // items per fold
  int maxn;                      // rows the LDS lists hold (a multiple of 4, >= the longest fold)
  unsigned flags;
};
constexpr int MID_THREADS = 64;
#ifndef CVM_MID_DEPTH
#define CVM_MID_DEPTH 3          // ring steps requested ahead of the one being consumed
#endif
constexpr int MID_DEPTH = CVM_MID_DEPTH, MID_SLOTS = MID_DEPTH + 1;
constexpr int MID_SLOT_ELEMS = 512;        // 4 rows x (64 + 64) columns, or 8 rows x 64 columns of G / H
constexpr int MID_SCR_BYTES = 16 * 17 * 8;
// LDS of a wave: the ring, the transposition scratch, then the fold's weights and row numbers -- and, over
// those two lists once the k-loop is over, three 64-entry blocks of row statistics
inline size_t mid_list_bytes(int maxn) { const size_t l = (size_t)maxn * 12; return l > 1536 ? l : 1536; }
template <typename T> inline size_t mid_lds_bytes(int maxn) {
  return (size_t)MID_SLOTS * MID_SLOT_ELEMS * sizeof(T) + MID_SCR_BYTES + mid_list_bytes(maxn);
}

// all but the n youngest vector-memory operations of this wave are done (n wave-uniform, 0..63)
__device__ __noinline__ void wait_vmcnt_le63(int n) {
#define CVM_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
#define CVM_W8(b) CVM_W(b) CVM_W(b + 1) CVM_W(b + 2) CVM_W(b + 3) CVM_W(b + 4) CVM_W(b + 5) CVM_W(b + 6) CVM_W(b + 7)
  switch (n) {
    CVM_W(0) CVM_W(1) CVM_W(2) CVM_W(3) CVM_W(4) CVM_W(5) CVM_W(6) CVM_W(7)
    CVM_W(8) CVM_W(9) CVM_W(10) CVM_W(11) CVM_W(12) CVM_W(13) CVM_W(14) CVM_W(15)
    CVM_W(16) CVM_W(17) CVM_W(18) CVM_W(19) CVM_W(20) CVM_W(21) CVM_W(22) CVM_W(23)
    CVM_W(24) CVM_W(25) CVM_W(26) CVM_W(27) CVM_W(28) CVM_W(29) CVM_W(30) CVM_W(31)
    CVM_W(32) CVM_W(33) CVM_W(34) CVM_W(35) CVM_W(36) CVM_W(37) CVM_W(38) CVM_W(39)
    CVM_W(40) CVM_W(41) CVM_W(42) CVM_W(43) CVM_W(44) CVM_W(45) CVM_W(46) CVM_W(47)
    CVM_W(48) CVM_W(49) CVM_W(50) CVM_W(51) CVM_W(52) CVM_W(53) CVM_W(54) CVM_W(55)
    CVM_W(56) CVM_W(57) CVM_W(58) CVM_W(59) CVM_W(60) CVM_W(61) CVM_W(62)
    default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
  }
#undef CVM_W8
#undef CVM_W
}
// the value of the neighbouring lane (l ^ 1)
__device__ __forceinline__ double mid_swap1(double v) {
  const long long b = __builtin_bit_cast(long long, v);
  const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffLL), 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
  const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), 0xB1, 0xF, 0xF, true);
  return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned)lo);
}

template <typename T, bool WEIGHTED>
__global__ __launch_bounds__(MID_THREADS, 2) void mid_tile_kernel(const MidArgs a) {
  typedef typename MF<T>::acc_t acc_t;
  static_assert(sizeof(T) == 8, "float64 (the DMA piece maps below are written for 8-byte elements)");
  constexpr int ES = 8;
  constexpr int D = MID_DEPTH, NS = MID_SLOTS;
  typedef T vt __attribute__((ext_vector_type(2)));
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int bid = blockIdx.x;
  const long long item = (long long)(bid & 7) * a.per_xcd + (bid >> 3);
  if (item >= a.n_items) return;
  const int lane = threadIdx.x;
#ifdef CVM_STAMPS
  unsigned long long tq[6];
  tq[0] = __builtin_amdgcn_s_memtime();
#define MID_STAMP(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tq[i] = __builtin_amdgcn_s_memtime(); } while (0)
#define MID_STAMP_OUT() do { tq[4] = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); tq[5] = __builtin_amdgcn_s_memtime(); \
    if (lane == 0 && (bid & 63) == 0 && (bid >> 6) < 1024) { unsigned long long *o = g_stamps4 + (size_t)(bid >> 6) * 8; \
      for (int i = 0; i < 6; ++i) o[i] = tq[i]; o[6] = (unsigned long long)(diag ? 1 : 0) + 2 * kind; o[7] = (unsigned long long)n; } } while (0)
#else
#define MID_STAMP(i) do {} while (0)
#define MID_STAMP_OUT() do {} while (0)
#endif
  const int K = a.K, M = a.M;
  const int f = (int)(item / a.ipf);
  const int q = (int)(item - (long long)f * a.ipf);
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
  const bool want_xty = a.out_XTY != nullptr && M > 0;
  const bool dxty = diag && want_xty;      // this item also computes the panel's first sixteen XTY columns
  const int a0 = ti * 64, b0 = tj * 64;
  const int ycol0 = kind == 2 ? 16 + 64 * yc : 0;

  const int64_t rbeg = a.offs[a.seg0 + f];
  const int n = (int)(a.offs[a.seg0 + f + 1] - rbeg);
  const int nks = (n + 3) >> 2;
  T *ring = reinterpret_cast<T *>(smem_raw);
  T (*scr)[17] = reinterpret_cast<T (*)[17]>(smem_raw + (size_t)NS * MID_SLOT_ELEMS * ES);
  char *lists = smem_raw + (size_t)NS * MID_SLOT_ELEMS * ES + MID_SCR_BYTES;
  T *wl = reinterpret_cast<T *>(lists);
  int *rowl = reinterpret_cast<int *>(wl + a.maxn);
  double *rs = reinterpret_cast<double *>(lists);    // (after the k-loop) [0,64) XTX row means, [64,128) row reciprocal stds, [128,192) XTY row means
  const double *fs = a.fstats + (size_t)f * fstat_len(K, M);
  const bool cX = a.flags & CVM_CENTER_X, cY = a.flags & CVM_CENTER_Y;
  const bool sX = a.flags & CVM_SCALE_X, sY = a.flags & CVM_SCALE_Y;
  const size_t fo = (size_t)(a.seg0 + f);
  const int lk = lane >> 4, lc = lane & 15;

  // ---- the fold's row numbers; this lane's statistics ----------------------------------------------------
  const int npad = 4 * nks;
  for (int r = lane; r < npad; r += 64) rowl[r] = r < n ? (int)a.idx[rbeg + r] : 0;
  double rsv[3] = {0.0, 1.0, 0.0};         // of row a0 + lane: XTX mean, reciprocal std, XTY mean
  if (a0 + lane < K) {
    if (cX) rsv[0] = fs[a0 + lane];
    if (sX) rsv[1] = fs[K + a0 + lane];
    if (cX || cY) rsv[2] = fs[a0 + lane];
  }
  double muc[4], sdc[4];                   // of this lane's column in each of the four column tiles
#pragma unroll
  for (int nn = 0; nn < 4; ++nn) {
    muc[nn] = 0.0; sdc[nn] = 1.0;
    if (kind == 0) {
      const int col = b0 + 16 * nn + lc;
      if (col < K) { if (cX) muc[nn] = fs[col]; if (sX) sdc[nn] = fs[K + col]; }
    } else {
      const int col = ycol0 + 16 * nn + lc;
      if (col < M) { if (cX || cY) muc[nn] = fs[2 * K + col]; if (sY) sdc[nn] = fs[2 * K + M + col]; }
    }
  }
  double muy = 0.0, sdy = 1.0;             // of response column lc (a diagonal item's XTY piece)
  if (dxty && lc < M) { if (cX || cY) muy = fs[2 * K + lc]; if (sY) sdy = fs[2 * K + M + lc]; }
  const double swt = fs[2 * K + 2 * M];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  MID_STAMP(1);

  // ---- the ring: steps 0 .. nks - 1 are k-steps of X (and Y), then come the parts of G and H -------------
  //   kind 0: 8 parts of G (8 rows x 64 columns), a diagonal item with XTY: + 2 parts of H (32 rows x 16 columns)
  //   kind 2: 8 parts of H (8 rows x 64 columns)
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)smem_raw);
  const char *zero = reinterpret_cast<const char *>(g_zero_line);
  const T *Xp = reinterpret_cast<const T *>(a.X);
  const T *Yp = reinterpret_cast<const T *>(a.Y);
  const T *Gt = reinterpret_cast<const T *>(a.G);
  const T *Ht = reinterpret_cast<const T *>(a.H);
  const bool bX = kind == 0 && !diag;      // the B side is a second panel of X
  const int IPK = 2 + (bX ? 2 : 0) + (kind == 2 ? 2 : 0) + (dxty ? 1 : 0);
  const int NV = nks + 8 + (dxty ? 2 : 0);
  auto cnt_of = [&](int v) { return v < nks ? IPK : 4; };
  auto issue = [&](int v) {
    const unsigned slot = lds0 + (unsigned)((v % NS) * MID_SLOT_ELEMS * ES);
    const int piece = lane & 31;
    if (v < nks) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {              // rows 2 h, 2 h + 1 of the k-step
        const int gr = 4 * v + 2 * h + (lane >> 5);
        const bool valid = gr < n;
        const int64_t rn = valid ? (int64_t)rowl[gr] : 0;
        int ca = a0 + 2 * piece;
        if (ca > K - 2) ca = K - 2;
        const char *srcA = valid ? reinterpret_cast<const char *>(Xp + rn * (int64_t)K + ca) : zero + 16 * piece;
        dma16_lanes(srcA, (unsigned)uni((int)(slot + (unsigned)(h * 128 * ES))));
        if (bX) {
          int cb = b0 + 2 * piece;
          if (cb > K - 2) cb = K - 2;
          const char *srcB = valid ? reinterpret_cast<const char *>(Xp + rn * (int64_t)K + cb) : zero + 16 * piece;
          dma16_lanes(srcB, (unsigned)uni((int)(slot + (unsigned)((256 + h * 128) * ES))));
        } else if (kind == 2) {                  // 64 response columns per row, clamped like the X columns
          int cy = ycol0 + 2 * piece;
          if (cy > M - 2) cy = M - 2;
          const char *srcY = valid ? reinterpret_cast<const char *>(Yp + rn * (int64_t)M + cy) : zero + 16 * piece;
          dma16_lanes(srcY, (unsigned)uni((int)(slot + (unsigned)((256 + h * 128) * ES))));
        }
      }
      if (dxty) {                                // the first sixteen response columns: 4 rows x 128 bytes, lanes 0..31
        const int gr = 4 * v + (lane >> 3);      // (lanes 32..63: the zero line, into the slot's spare half)
        const bool valid = gr < n && lane < 32;
        const int64_t rn = valid ? (int64_t)rowl[gr] : 0;
        int cy = 2 * (lane & 7);
        if (cy > M - 2) cy = M - 2;
        const char *srcY = valid ? reinterpret_cast<const char *>(Yp + rn * (int64_t)M + cy) : zero + 16 * (lane & 7);
        dma16_lanes(srcY, (unsigned)uni((int)(slot + (unsigned)(256 * ES))));
      }
      return;
    }
    const int p = v - nks;
    if (p < 8) {
      // part p of G (kind 0) or of H (kind 2): rows a0 + 8 p .. + 7, 64 columns from b0 / ycol0
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int row = a0 + 8 * p + 2 * h + (lane >> 5);
        const char *src;
        if (kind == 0) {
          int cb = b0 + 2 * piece;
          if (cb > K - 2) cb = K - 2;
          src = row < K ? reinterpret_cast<const char *>(Gt + (size_t)row * K + cb) : zero + 16 * piece;
        } else {
          int cy = ycol0 + 2 * piece;
          if (cy > M - 2) cy = M - 2;
          src = row < K ? reinterpret_cast<const char *>(Ht + (size_t)row * M + cy) : zero + 16 * piece;
        }
        dma16_lanes(src, (unsigned)uni((int)(slot + (unsigned)(h * 128 * ES))));
      }
    } else {
      // part p - 8 of the diagonal item's H piece: rows a0 + 32 (p - 8) .. + 31, sixteen columns
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int row = a0 + 32 * (p - 8) + 8 * h + (lane >> 3);
        int cy = 2 * (lane & 7);
        if (cy > M - 2) cy = M - 2;
        const char *src = row < K ? reinterpret_cast<const char *>(Ht + (size_t)row * M + cy) : zero + 16 * (lane & 7);
        dma16_lanes(src, (unsigned)uni((int)(slot + (unsigned)(h * 128 * ES))));
      }
    }
  };
#pragma unroll 1
  for (int t = 0; t < D; ++t)
    if (t < NV) issue(t);
  // (one more level of dependent loads -- w[row] -- in flight together with the first steps)
  for (int r = lane; r < npad; r += 64)
    wl[r] = r < n ? (WEIGHTED ? reinterpret_cast<const T *>(a.w)[rowl[r]] : (T)1) : (T)0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  MID_STAMP(2);

  // k-step t: its DMAs have landed when at most the younger ones (steps t + 1 .. t + D - 1) are outstanding
  auto kstep_head = [&](int t) {
    int young = 0;
#pragma unroll
    for (int u = 1; u < D; ++u)
      if (t + u < NV) young += cnt_of(t + u);
    wait_vmcnt_le63(uni(young));
    if (t + D < NV) issue(t + D);                        // (into the slot of step t - 1: read and consumed)
  };
  // The finish consumes the parts two at a time (sixteen rows of the tile = one row of MFMA tiles): step m
  // first requests the parts up to 2 m + 3 (into the slots of the two parts step m - 1 consumed), then waits
  // for parts 2 m and 2 m + 1 -- younger than those are the DMAs of the parts requested after them and the
  // stores of step m - 1.  (A store counts when any lane takes part; an instruction that issues with no lane
  // active makes the wait longer than needed, never shorter.)
  const int NP = NV - nks;
  int next_p = D < NP ? D : NP;                          // parts 0 .. D - 1 were requested by the k-loop
  int prev_stores = 0;
  auto finish_head = [&](int m) {
    const int upto = 2 * m + 3 < NP - 1 ? 2 * m + 3 : NP - 1;
#pragma unroll 1
    while (next_p <= upto) { issue(nks + next_p); ++next_p; }
    int young = 4 * (next_p - (2 * m + 2)) + prev_stores;
    if (young < prev_stores) young = prev_stores;
    wait_vmcnt_le63(uni(young));
  };
  T *outp = reinterpret_cast<T *>(a.out_XTX) + fo * (size_t)K * K;
  T *outy = reinterpret_cast<T *>(a.out_XTY) + fo * (size_t)K * M;

  // the row statistics go over the two lists once the k-loop is over
  auto rs_to_lds = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int i = 0; i < 3; ++i) rs[64 * i + lane] = rsv[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  };
  // finished values fin[r] of MFMA tile (m, nn) (rows drow(r), column lc): out[b0 + 16 nn + c][a0 + 16 m + r']
  auto mirror_tile = [&](const T (&fin)[4], int m, int nn) -> int {
#pragma unroll
    for (int r = 0; r < 4; ++r) scr[MF<T>::drow(lane, r)][lc] = fin[r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int cnt = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int c = (lane >> 3) + 8 * j, e0 = 2 * (lane & 7);
      const int grow = b0 + 16 * nn + c, gcol = a0 + 16 * m + e0;
      vt vv;
      vv[0] = scr[e0][c]; vv[1] = scr[e0 + 1][c];
      const bool ok = grow < K && gcol < K;
      if (ok) out_store(reinterpret_cast<vt *>(outp + (size_t)grow * K + gcol), vv);
      cnt += __ballot(ok) != 0ull ? 1 : 0;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    return cnt;
  };
  // two finished values of one lane (rows r0 < r1 of the same MFMA tile, column lc) -> one 16-byte store per
  // lane: even lanes take row r0, odd lanes row r1, columns (lc & ~1), + 1
  auto store_pair = [&](T *base, int ld, int row0, int row1, int col, int rows_end, int cols_end, T v0, T v1) -> int {
    const bool odd = lane & 1;
    const T got = mid_swap1(odd ? v0 : v1);              // even lanes receive the neighbour's v0, odd lanes its v1
    vt vv;
    vv[0] = odd ? got : v0; vv[1] = odd ? v1 : got;
    const int row = odd ? row1 : row0, c0 = col & ~1;
    const bool ok = row < rows_end && c0 < cols_end;
    if (ok) out_store(reinterpret_cast<vt *>(base + (size_t)row * ld + c0), vv);
    return __ballot(ok) != 0ull ? 1 : 0;
  };
  auto finish_val = [&](double g, double u, double mur, double sdr, double mc, double sc_) -> T {
    double x = g - u;
    if (cX) x -= swt * (mur * mc);
    if (sX) x = x * (sdr * sc_);
    return (T)x;
  };
  auto finish_xty = [&](double h, double u, double mur, double sdr, double mc, double sc_) -> T {
    double x = h - u;
    if (cX || cY) x -= swt * (mur * mc);
    if (sX && sY) x = x * (sdr * sc_);
    else if (sX) x = x * sdr;
    else if (sY) x = x * sc_;
    return (T)x;
  };

  if (!diag) {
    // ---- off-diagonal XTX tile, or 64 x 64 of XTY: 4 x 4 MFMA tiles --------------------------------------
    acc_t acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (acc_t){0, 0, 0, 0};
#pragma unroll 1
    for (int t = 0; t < nks; ++t) {
      kstep_head(t);
      const T *slot = ring + (t % NS) * MID_SLOT_ELEMS;
      T af[4], bf[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) af[m] = slot[lk * 64 + 16 * m + lc];
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) bf[nn] = slot[256 + lk * 64 + 16 * nn + lc];
      if (WEIGHTED) {
        const T wv = wl[4 * t + lk];
#pragma unroll
        for (int m = 0; m < 4; ++m) af[m] = (T)(af[m] * wv);
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) acc[m * 4 + nn] = MF<T>::mfma(af[m], bf[nn], acc[m * 4 + nn]);
    }
    MID_STAMP(3);
    rs_to_lds();
    const bool xty = kind == 2;
    // finish step m: tile row m (rows 16 m ..) -- always in acc[0..3]: the rows below move up after every step
#pragma unroll 1
    for (int m = 0; m < 4; ++m) {
      finish_head(m);
      int cnt = 0;
      T fin[4][4];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const T *slot = ring + ((nks + 2 * m + half) % NS) * MID_SLOT_ELEMS;
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
          T val[2];
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * half + rr;
            const int lr = 16 * m + MF<T>::drow(lane, r);            // = 16 m + 8 half + lk + 4 rr
            const T g = slot[(lk + 4 * rr) * 64 + 16 * nn + lc];
            val[rr] = xty ? finish_xty((double)g, (double)acc[nn][r], rs[128 + lr], rs[64 + lr], muc[nn], sdc[nn])
                          : finish_val((double)g, (double)acc[nn][r], rs[lr], rs[64 + lr], muc[nn], sdc[nn]);
            fin[nn][r] = val[rr];
          }
          const int row0 = a0 + 16 * m + MF<T>::drow(lane, 2 * half);
          if (xty) cnt += store_pair(outy, M, row0, row0 + 4, ycol0 + 16 * nn + lc, K, M, val[0], val[1]);
          else cnt += store_pair(outp, K, row0, row0 + 4, b0 + 16 * nn + lc, K, K, val[0], val[1]);
        }
      }
      if (!xty && a0 + 16 * m < K) {
#pragma unroll
        for (int nn = 0; nn < 4; ++nn)
          if (b0 + 16 * nn < K) cnt += mirror_tile(fin[nn], m, nn);
      }
      prev_stores = cnt;
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[i] = acc[i + 4];
    }
    MID_STAMP_OUT();
    return;
  }

  // ---- diagonal XTX tile: the ten MFMA tiles on and above the diagonal + 64 x 16 of XTY ---------------
  {
    acc_t acc[16], yacc[4];                  // (tile (m, nn) in acc[4 m + nn]; those below the diagonal stay zero)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = (acc_t){0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i) yacc[i] = (acc_t){0, 0, 0, 0};
#pragma unroll 1
    for (int t = 0; t < nks; ++t) {
      kstep_head(t);
      const T *slot = ring + (t % NS) * MID_SLOT_ELEMS;
      T xf[4], af[4], yf = (T)0;
#pragma unroll
      for (int m = 0; m < 4; ++m) xf[m] = slot[lk * 64 + 16 * m + lc];
      if (dxty) yf = slot[256 + lk * 16 + lc];
      if (WEIGHTED) {
        const T wv = wl[4 * t + lk];
#pragma unroll
        for (int m = 0; m < 4; ++m) af[m] = (T)(xf[m] * wv);
        yf = (T)(yf * wv);
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) af[m] = xf[m];
      }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int nn = m; nn < 4; ++nn) acc[4 * m + nn] = MF<T>::mfma(af[m], xf[nn], acc[4 * m + nn]);
      if (dxty) {
#pragma unroll
        for (int m = 0; m < 4; ++m) yacc[m] = MF<T>::mfma(xf[m], yf, yacc[m]);
      }
    }
    MID_STAMP(3);
    rs_to_lds();
#pragma unroll 1
    for (int m = 0; m < 4; ++m) {
      finish_head(m);
      int cnt = 0;
      T fin[4][4];
      // the diagonal MFMA tile of this row (acc[m] of the row in acc[0..3]): the values below its diagonal are
      // the mirror images of those above (the same bits in both places, like fused_finish_direct)
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) {
        if (nn != m) continue;                           // (wave-uniform)
#pragma unroll
        for (int r = 0; r < 4; ++r) scr[MF<T>::drow(lane, r)][lc] = acc[nn][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int tr = MF<T>::drow(lane, r);
          if (tr > lc) acc[nn][r] = scr[lc][tr];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const T *slot = ring + ((nks + 2 * m + half) % NS) * MID_SLOT_ELEMS;
#pragma unroll
        for (int nn = 0; nn < 4; ++nn) {
          if (nn < m) continue;                          // (wave-uniform)
          T val[2];
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const int r = 2 * half + rr;
            const int lr = 16 * m + MF<T>::drow(lane, r);
            const T g = slot[(lk + 4 * rr) * 64 + 16 * nn + lc];
            val[rr] = finish_val((double)g, (double)acc[nn][r], rs[lr], rs[64 + lr], muc[nn], sdc[nn]);
            fin[nn][r] = val[rr];
          }
          const int row0 = a0 + 16 * m + MF<T>::drow(lane, 2 * half);
          cnt += store_pair(outp, K, row0, row0 + 4, a0 + 16 * nn + lc, K, K, val[0], val[1]);
        }
      }
      if (a0 + 16 * m < K) {
#pragma unroll
        for (int nn = 1; nn < 4; ++nn)
          if (nn > m && a0 + 16 * nn < K) cnt += mirror_tile(fin[nn], m, nn);
      }
      prev_stores = cnt;
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[i] = acc[i + 4];
    }
    if (dxty) {
      // the XTY piece: two parts of H (32 rows x 16 columns each) = rows of MFMA tiles 2 pp, 2 pp + 1
      finish_head(4);
      int cnt = 0;
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        const T *slot = ring + ((nks + 8 + pp) % NS) * MID_SLOT_ELEMS;
#pragma unroll
        for (int mm = 0; mm < 2; ++mm) {
          const int m = 2 * pp + mm;
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            T val[2];
#pragma unroll
            for (int rr = 0; rr < 2; ++rr) {
              const int r = 2 * half + rr;
              const int tr = MF<T>::drow(lane, r), lr = 16 * m + tr;
              const T h = slot[(16 * mm + tr) * 16 + lc];
              val[rr] = finish_xty((double)h, (double)yacc[m][r], rs[128 + lr], rs[64 + lr], muy, sdy);
            }
            const int row0 = a0 + 16 * m + MF<T>::drow(lane, 2 * half);
            cnt += store_pair(outy, M, row0, row0 + 4, lc, K, M, val[0], val[1]);
          }
        }
      }
    }
    MID_STAMP_OUT();
  }
}
