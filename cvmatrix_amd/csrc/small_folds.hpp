// small_folds.hpp -- part of libcvmhip.so (included by cvmhip.hip inside its anonymous namespace).
// Direct HBM-bound kernels for folds of few rows (at most SMALL_MAXN, staged SMALL_ROWS at a time).
#pragma once

// ----------------------------------------------------------------------------------
// Small folds (at most SMALL_ROWS validation rows: leave-one-out and its neighbours).
// The Gram of a handful of rows is no MFMA problem: the fold update is a stream over
// K x (K+M) outputs (read G, H once, write XTX, XTY once) -- HBM-bound.  Two kernels, no
// partials in between:
//   small_stats_kernel  column sums of the fold's rows, sequential in row order for sw, sX,
//                       qX, sY, qY alike (constant-one columns stay exact), then the same
//                       mean / std arithmetic as fold_stats_kernel
//   small_apply_kernel  one 64x64 upper tile per workgroup: the fold's rows (A side weighted)
//                       go to LDS, every thread accumulates a 4x4 block of the rank-n update
//                       in float64, applies total - update, centring and scaling in the
//                       reference's order (cvmatrix.py:1001-1010) and the tile is written
//                       twice -- as is and transposed through LDS -- with coalesced stores.
// ----------------------------------------------------------------------------------
constexpr int SMALL_ROWS = 32;          // rows staged at a time (one chunk); folds of at most so many take one pass
constexpr int SMALL_MAXN = 128;         // the kernels take folds of up to so many rows, SMALL_ROWS at a time
struct SmallArgs {
  const void *X, *Y, *w;
  const int64_t *idx, *offs;
  int64_t seg0;
  int K, M;
  const void *G, *H;
  const double *gstats;
  double *fstats;                      // [fold of batch][fstat_len]
  void *out_XTX, *out_XTY, *out_muX, *out_sdX, *out_muY, *out_sdY;
  double *out_fold;
  double ddof, resolution;
  unsigned flags;
  int P64, nT64;
  // one small fold whose indices came from the host (CVM_IDX_HOST): they travel in the kernel
  // arguments, no device copy of the index array is needed; inl_n < 0: idx / offs are used
  int inl_n;
  int64_t inl[SMALL_ROWS];
  int nb, fpb;                         // folds of this launch, folds per workgroup of small_apply_kernel
  int rshift;                          // small_apply_kernel: 1 << rshift row slots per fold (32, 64, 128); fpb << rshift <= 256
  int gx, gy;                          // small_apply_kernel: tiles + panels, fold groups
  int x0;                              // small_apply_kernel: first tile / panel number of this launch
  int noremap;                         // 1: workgroup b works on item b (no XCD-contiguous ranges)
};

template <typename T, bool WEIGHTED> __global__ __launch_bounds__(256) void small_stats_kernel(const SmallArgs a) {
  const int f = blockIdx.x;
  const int K = a.K, M = a.M;
  const bool inl = a.inl_n >= 0;
  const int64_t o0 = inl ? 0 : a.offs[a.seg0 + f];
  const int n = inl ? a.inl_n : (int)(a.offs[a.seg0 + f + 1] - o0);
  const T *X = (const T *)a.X, *Y = (const T *)a.Y, *W = (const T *)a.w;
  __shared__ int64_t rows[SMALL_MAXN];
  __shared__ double wl[SMALL_MAXN];
  if ((int)threadIdx.x < n) {
    const int64_t r = inl ? a.inl[threadIdx.x] : a.idx[o0 + threadIdx.x];
    rows[threadIdx.x] = r;
    wl[threadIdx.x] = WEIGHTED ? (double)W[r] : 1.0;
  }
  __syncthreads();
  double swv = 0, nzv = 0;
  for (int r = 0; r < n; ++r) { swv += wl[r]; nzv += (wl[r] != 0.0) ? 1.0 : 0.0; }
  const double gsw = a.gstats[2 * K + 2 * M], gnz = a.gstats[2 * K + 2 * M + 1];
  const double swt = gsw - swv, nzt = gnz - nzv;
  const double divisor = (nzt - a.ddof) * swt / nzt;
  double *fs = a.fstats + (size_t)f * fstat_len(K, M);
  if (threadIdx.x == 0 && blockIdx.y == 0) {
    fs[2 * K + 2 * M] = swt;
    if (a.out_fold) {
      double *o = a.out_fold + 4 * (a.seg0 + f);
      o[0] = swt; o[1] = nzt; o[2] = swv; o[3] = nzv;
    }
  }
  const bool cX = a.flags & CVM_CENTER_X, cY = a.flags & CVM_CENTER_Y;
  const bool sX = a.flags & CVM_SCALE_X, sY = a.flags & CVM_SCALE_Y;
  const bool rXTY = a.flags & CVM_RET_XTY;
  const bool want_muX = cX || sX || (rXTY && cY), want_sdX = sX;
  const bool want_muY = rXTY && (cX || cY || sY), want_sdY = rXTY && sY;
  for (int c = blockIdx.y * blockDim.x + threadIdx.x; c < K + M; c += gridDim.y * blockDim.x) {
    const bool isX = c < K;
    const int cc = isX ? c : c - K;
    if (isX ? !(want_muX) : !(want_muY)) continue;
    double sv = 0, qv = 0;
    // eight rows requested at a time (a row past the fold's end re-reads its last row: branch-free, so
    // that the loads of a batch are in flight together), summed in row order
    const T *col = isX ? X + cc : Y + cc;
    const int64_t ld = isX ? K : M;
    for (int r0 = 0; r0 < n; r0 += 8) {
      T xb[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) xb[u] = col[rows[r0 + u < n ? r0 + u : n - 1] * ld];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (r0 + u >= n) break;
        const T xv = xb[u];
        const int r = r0 + u;
        if (sizeof(T) == 8) {
          const T pv = WEIGHTED ? (T)((T)wl[r] * xv) : xv;
          sv += (double)pv; qv += (double)(pv * xv);
        } else {
          const double pv = wl[r] * (double)xv;
          sv += pv; qv += pv * (double)xv;
        }
      }
    }
    const double gs = isX ? a.gstats[cc] : a.gstats[2 * K + cc];
    const double gq = isX ? a.gstats[K + cc] : a.gstats[2 * K + M + cc];
    const double st_ = gs - sv;          // cvmatrix.py:1020
    const double mu = st_ / swt;         // cvmatrix.py:1043
    double sd = 1.0;
    if (isX ? want_sdX : want_sdY) {
      const double qt = gq - qv;
      double var = (-2 * mu * st_ + swt * (mu * mu) + qt) / divisor;   // 1119-1123
      var = (var < 0) ? 0.0 : var;       // np.maximum(var, 0): NaN stays NaN
      sd = sqrt(var);
      if (sd <= a.resolution) sd = 1.0;  // 1128
    }
    fs[isX ? cc : 2 * K + cc] = mu;
    fs[isX ? K + cc : 2 * K + M + cc] = 1.0 / sd;     // reciprocal: the finish multiplies (finalize.hpp)
    T *omu = (T *)(isX ? a.out_muX : a.out_muY), *osd = (T *)(isX ? a.out_sdX : a.out_sdY);
    const size_t o = (size_t)(a.seg0 + f) * (isX ? K : M) + cc;
    if (omu) omu[o] = (T)mu;
    if (osd && (isX ? want_sdX : want_sdY)) osd[o] = (T)sd;
  }
}

// small_apply_kernel, round 3.  What the round-2 kernel spent its time on (tools/exp_small_apply.py,
// K = 4096, 16-row folds): with NO rows at all (only G - 0, finish, stores) it moved 4.7 TB/s of
// float32 outputs and 5.5 TB/s of float64 -- every fold of a workgroup's group started with two
// dependent global latencies (offsets, then the statistics), and under a memory system saturated
// with writes a read takes several microseconds -- and the rank-n update itself, 16 mul + 16 add per
// row and thread with four LDS reads, cost another 0.2 ms per 16 rows (3.5 / 4.0 TB/s), serial with
// the stores.  Now:
//   * the update runs on the matrix cores: wave w owns row tile w of the 64 x 64 tile, 4 MFMA
//     16x16x4 tiles, one A and four B fragment reads per 4 rows (LDS row pitch 80: conflict-free);
//   * the indices, weights and row counts of ALL folds of the workgroup's group are read once, up
//     front; the next fold's rows and statistics are requested before the current fold's
//     arithmetic and stores and arrive in registers while those run: no global latency is left
//     inside the per-fold chain.
constexpr int SA_PITCH = 80;            // LDS row pitch of As / Bs in elements (f32: 320 B, f64: 640 B)
constexpr int SA_FPB = 8;               // folds per workgroup at most (host: fpb <= 8)
// (at least four waves per SIMD: float32 then fits 128 registers -- four workgroups per CU instead of
//  three -- and gains 3-7 %; float64 is not bound by occupancy)
#ifndef CVM_SMALL_WPE
#define CVM_SMALL_WPE 4
#endif
template <typename T, bool WEIGHTED> __global__ __launch_bounds__(256, CVM_SMALL_WPE) void small_apply_kernel(const SmallArgs a) {
  // Workgroups go to the 8 XCDs round-robin by their linear number; neighbouring tiles of one
  // output matrix share cache lines when its rows are not whole lines, and only one L2 can merge
  // the two halves before they go to HBM: give every XCD a contiguous range of (fold group, tile).
  // (a 1-D launch of 8 * ceil(gx * gy / 8) workgroups; gx tiles and panels, gy fold groups)
  const unsigned lin = blockIdx.x, tot = (unsigned)a.gx * (unsigned)a.gy;
  const unsigned per = (tot + 7) / 8;
  const unsigned item = a.noremap ? lin : (lin & 7) * per + (lin >> 3);
  if (item >= tot) return;
  const int x = a.x0 + (int)(item % (unsigned)a.gx), by = (int)(item / (unsigned)a.gx);
  const int K = a.K, M = a.M;
  const int tid = threadIdx.x;
  const T *X = (const T *)a.X, *Y = (const T *)a.Y, *W = (const T *)a.w;
  const bool cX = a.flags & CVM_CENTER_X, cY = a.flags & CVM_CENTER_Y;
  const bool sX = a.flags & CVM_SCALE_X, sY = a.flags & CVM_SCALE_Y;
  const bool inl = a.inl_n >= 0;
  // As: w * x, columns of the tile's rows; Bs: x (or y), columns of the tile's columns (SMALL_ROWS
  // rows of SA_PITCH each); Ts: the tile for the finish and the transposed store -- it reuses the
  // As / Bs space.  Elements are T: float32 problems stage and accumulate in float32.
  typedef T TS;
  constexpr int AB_ELEMS = 2 * SMALL_ROWS * SA_PITCH, TS_ELEMS = ST * (ST + 1);
  constexpr int SLOT = AB_ELEMS > TS_ELEMS ? AB_ELEMS : TS_ELEMS;
  constexpr int NSLOT = SA_FPB * SMALL_ROWS;     // row slots of a workgroup: fpb folds x (1 << rshift) rows
  __shared__ __attribute__((aligned(16))) TS sm[SLOT];
  __shared__ int64_t rows_all[NSLOT];
  __shared__ T wl_all[NSLOT];
  __shared__ int n_all[SA_FPB];
  __shared__ double st_lds[2][4 * ST];           // the fold's means / stds of the tile's rows and columns
  const bool xtx_part = x < a.nT64;
  if (xtx_part ? !a.out_XTX : (!a.out_XTY || M == 0)) return;
  int ti = 0, tj = 0;
  if (xtx_part) decode_tile(x, a.P64, ti, tj);
  const int a0 = (xtx_part ? ti : x - a.nT64) * ST, b0 = tj * ST;
  const int f_first = by * a.fpb;
  const int nf = (a.nb - f_first < a.fpb) ? a.nb - f_first : a.fpb;      // folds of this workgroup
  const int rshift = a.rshift;
  // ---- every fold's row numbers, weights and row count, once ----------------------------------
  {
    const int ff = tid >> rshift, j = tid & ((1 << rshift) - 1);         // fpb folds x row slots = 256 threads at most
    if (ff < nf) {
      const int f = f_first + ff;
      const int64_t o0 = inl ? 0 : a.offs[a.seg0 + f];
      const int n = inl ? a.inl_n : (int)(a.offs[a.seg0 + f + 1] - o0);
      int64_t r = 0;
      T wv = (T)0;
      if (j < n) {
        r = inl ? a.inl[j] : a.idx[o0 + j];
        wv = WEIGHTED ? W[r] : (T)1;
      }
      rows_all[tid] = r;
      wl_all[tid] = wv;
      if (j == 0) n_all[ff] = n;
    }
  }
  lds_barrier();
  if (xtx_part) {
    constexpr int VWG = 16 / (int)sizeof(T);
    constexpr int NQG = ST * (ST / VWG) / 256;
    T gpre[NQG][VWG];
    finish_tile_preload<T, NQG>(gpre, a0, b0, K, (const T *)a.G, (const T *)a.out_XTX, tid, 256);
    // staging element j of this thread: row sr0 + 4 j of the chunk, column sc of the tile; up to
    // SMALL_ROWS * 64 / 256 = 8 elements, each an A-side and a B-side value
    constexpr int NE = SMALL_ROWS * ST / 256;
    const int sc = tid & (ST - 1), sr0 = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform: scalar branches below)
    const bool ca_ok = a0 + sc < K, cb_ok = b0 + sc < K;
    T pa[NE], pb[NE];
    double pst = 0.0, pswt = 0.0;
    // rows of chunk ch of fold ff (and, with its first chunk, the fold's statistics) -> registers
    auto request = [&](int ff, int ch) {
      const int nc = n_all[ff] - SMALL_ROWS * ch;              // rows left from this chunk on
      const int base = (ff << rshift) + SMALL_ROWS * ch;
#pragma unroll
      for (int j = 0; j < NE; ++j) {
        const int r = sr0 + 4 * j;
        pa[j] = (T)0; pb[j] = (T)0;
        if (r < nc) {
          const int64_t row = rows_all[base + r];
          if (ca_ok) pa[j] = X[row * (int64_t)K + a0 + sc];
          if (cb_ok) pb[j] = X[row * (int64_t)K + b0 + sc];
        }
      }
      if (ch == 0) {
        const double *fs = a.fstats + (size_t)(f_first + ff) * fstat_len(K, M);
        pswt = fs[2 * K + 2 * M];
        const int part = tid / ST, l = tid - part * ST;        // stage_tile_stats, one value per thread
        const int col = ((part < 2) ? a0 : b0) + l;
        pst = (col < K) ? fs[((part & 1) ? K : 0) + col] : ((part & 1) ? 1.0 : 0.0);
      }
    };
    request(0, 0);
    const int wave = tid >> 6, lane = tid & 63, lk = lane >> 4, lc = lane & 15;
    // Fast finish of a tile that lies wholly inside the matrix, off the diagonal, with 16-byte
    // accessible rows (all but the edge and diagonal tiles of a large K): piece j of this thread is
    // row fr0 + FSTEP j, columns fc .. fc + VWG - 1 of the tile; its two output offsets are computed
    // once per workgroup, the per-fold work is arithmetic, LDS and stores only (finish_store_tile
    // re-derives rows, columns, bounds and addresses per piece and fold: most of its instructions)
    constexpr int LPRF = ST / VWG, FSTEP = 256 / LPRF;
    const int fr0 = tid / LPRF, fc = (tid - fr0 * LPRF) * VWG;
    const bool fast = ti != tj && a0 + ST <= K && b0 + ST <= K && ((size_t)K * sizeof(T)) % 16 == 0 &&
                      ((uintptr_t)a.out_XTX % 16 == 0) && ((uintptr_t)a.G % 16 == 0);
    const size_t off_d0 = (size_t)(a0 + fr0) * K + b0 + fc, off_m0 = (size_t)(b0 + fr0) * K + a0 + fc;
    const size_t off_step = (size_t)FSTEP * K;
    typedef T vst_t __attribute__((ext_vector_type(VWG)));
    typedef typename MF<T>::acc_t acc_t;
    TS *As = sm, *Bs = sm + SMALL_ROWS * SA_PITCH;
    TS (*Ts)[ST + 1] = reinterpret_cast<TS (*)[ST + 1]>(sm);
    // One pass = one chunk of at most SMALL_ROWS rows of one fold; the accumulators live across the
    // chunks of a fold (one chunk for folds of at most SMALL_ROWS rows) and the tile is finished after the last.  (Two folds per pass were
    // measured for float32 -- the tile kernel is bound by instructions per byte, and a pass pays
    // barriers, loop control and output offsets once: SLOWER, 16-row folds 0.761 -> 0.831 ms, three
    // workgroups per CU instead of five.)
    int pp = 0;
    for (int ff = 0; ff < nf; ++ff, pp ^= 1) {
      const int n = n_all[ff];
      const int nch = n > SMALL_ROWS ? (n + SMALL_ROWS - 1) / SMALL_ROWS : 1;
      double swt = 0.0;
      acc_t acc[4];
#pragma unroll
      for (int nn = 0; nn < 4; ++nn) acc[nn] = (acc_t){0, 0, 0, 0};
      for (int ch = 0; ch < nch; ++ch) {
        const int left = n - SMALL_ROWS * ch;
        const int n4 = ((left < SMALL_ROWS ? left : SMALL_ROWS) + 3) & ~3;   // zero rows up to a multiple of 4
        const int base = (ff << rshift) + SMALL_ROWS * ch;
        // the chunk's rows (A side weighted) and, with the first chunk, the fold's statistics -> LDS
#pragma unroll
        for (int j = 0; j < NE; ++j) {
          const int r = sr0 + 4 * j;
          if (r < n4) {
            As[r * SA_PITCH + sc] = WEIGHTED ? (T)(wl_all[base + r] * pa[j]) : pa[j];
            Bs[r * SA_PITCH + sc] = pb[j];
          }
        }
        if (ch == 0) {
          st_lds[pp][tid] = pst;
          swt = pswt;
        }
        lds_barrier();
        {                                        // in flight during the arithmetic and the stores below
          const int nff = ch + 1 < nch ? ff : ff + 1, nchk = ch + 1 < nch ? ch + 1 : 0;
          if (nff < nf) request(nff, nchk);
        }
        // rank-n update of the 64 x 64 tile on the matrix cores: wave w -> row tile w, 4 column tiles
        for (int k0 = 0; k0 < n4; k0 += 4) {
          const T af = As[(k0 + lk) * SA_PITCH + 16 * wave + lc];
          T bf[4];
#pragma unroll
          for (int nn = 0; nn < 4; ++nn) bf[nn] = Bs[(k0 + lk) * SA_PITCH + 16 * nn + lc];
#pragma unroll
          for (int nn = 0; nn < 4; ++nn) acc[nn] = MF<T>::mfma(af, bf[nn], acc[nn]);
        }
        lds_barrier();   // every wave is done with As/Bs: the next chunk, or Ts, may overwrite them
      }
#pragma unroll
      for (int nn = 0; nn < 4; ++nn)
#pragma unroll
        for (int r = 0; r < 4; ++r) Ts[16 * wave + MF<T>::drow(lane, r)][16 * nn + lc] = acc[nn][r];
      lds_barrier();
      const size_t fo = (size_t)(a.seg0 + f_first + ff) * (size_t)K * K;
      if (fast) {
        const double *st = st_lds[pp];
        double muc[VWG], sdc[VWG];
#pragma unroll
        for (int e = 0; e < VWG; ++e) { muc[e] = st[2 * ST + fc + e]; sdc[e] = st[3 * ST + fc + e]; }
        T *od = (T *)a.out_XTX + fo + off_d0;
#pragma unroll
        for (int j = 0; j < NQG; ++j) {
          const int lr = fr0 + FSTEP * j;
          const double mur = st[lr], sdr = st[ST + lr];
          vst_t vv;
#pragma unroll
          for (int e = 0; e < VWG; ++e) {
            double v = (double)gpre[j][e] - (double)Ts[lr][fc + e];
            if (cX) v -= swt * (mur * muc[e]);
            if (sX) v = v * (sdr * sdc[e]);
            vv[e] = (T)v;
          }
          out_store(reinterpret_cast<vst_t *>(od), vv);
          od += off_step;
#pragma unroll
          for (int e = 0; e < VWG; ++e) Ts[lr][fc + e] = vv[e];       // parked for the mirrored pass
        }
        lds_barrier();
        T *om = (T *)a.out_XTX + fo + off_m0;
#pragma unroll
        for (int j = 0; j < NQG; ++j) {
          const int lr = fr0 + FSTEP * j;
          vst_t vv;
#pragma unroll
          for (int e = 0; e < VWG; ++e) vv[e] = Ts[fc + e][lr];          // finished, transposed
          out_store(reinterpret_cast<vst_t *>(om), vv);
          om += off_step;
        }
        lds_barrier();
      } else {
        finish_store_tile<T, true, TS>(Ts, ti == tj, a0, b0, K, (const T *)a.G, (T *)a.out_XTX + fo, st_lds[pp], swt, cX, sX,
                                       tid, 256, gpre);
      }
      // (both ways end with a barrier: As / Bs are free for the next pass)
    }
  } else {
    // XTY panel: 64 rows of XTY x up to 64 responses at a time; output e = tid + 256 i of the panel
    // accumulates over the chunks of the fold in row order
    TS *As = sm, *Bs = sm + SMALL_ROWS * SA_PITCH;
    constexpr int NO = ST * ST / 256;
    for (int ff = 0; ff < nf; ++ff) {
      const int f = f_first + ff;
      const int n = n_all[ff];
      const double *fs = a.fstats + (size_t)f * fstat_len(K, M);
      const double swt = fs[2 * K + 2 * M];
      const size_t fo = (size_t)(a.seg0 + f);
      const T *Ht = (const T *)a.H;
      T *out = (T *)a.out_XTY + fo * (size_t)K * M;
      for (int m0 = 0; m0 < M; m0 += ST) {
        const int mw = (M - m0 < ST) ? M - m0 : ST;
        TS acc[NO];
#pragma unroll
        for (int i = 0; i < NO; ++i) acc[i] = 0;
        for (int c0 = 0; c0 < n; c0 += SMALL_ROWS) {
          const int nc = n - c0 < SMALL_ROWS ? n - c0 : SMALL_ROWS;
          const int base = (ff << rshift) + c0;
          lds_barrier();                         // the previous chunk / panel / fold is done with As / Bs
          for (int e = tid; e < nc * ST; e += 256) {
            const int r = e / ST, c = e - r * ST;
            const int64_t row = rows_all[base + r];
            const T xa = (a0 + c < K) ? X[row * (int64_t)K + a0 + c] : (T)0;
            As[r * SA_PITCH + c] = WEIGHTED ? (T)(wl_all[base + r] * xa) : xa;
            Bs[r * SA_PITCH + c] = (m0 + c < M) ? Y[row * (int64_t)M + m0 + c] : (T)0;
          }
          lds_barrier();
#pragma unroll
          for (int i = 0; i < NO; ++i) {
            const int e = tid + 256 * i;
            if (e < ST * mw) {
              const int la = e / mw, lm = e - la * mw;
              TS s_ = acc[i];
              for (int r = 0; r < nc; ++r) s_ += As[r * SA_PITCH + la] * Bs[r * SA_PITCH + lm];
              acc[i] = s_;
            }
          }
        }
#pragma unroll
        for (int i = 0; i < NO; ++i) {
          const int e = tid + 256 * i;
          if (e >= ST * mw) continue;
          const int la = e / mw, lm = e - la * mw;
          const int ga = a0 + la, gm = m0 + lm;
          if (ga >= K) continue;
          double v = (double)Ht[(size_t)ga * M + gm] - (double)acc[i];
          if (cX || cY) v -= swt * (fs[ga] * fs[2 * K + gm]);
          if (sX && sY) v = v * (fs[K + ga] * fs[2 * K + M + gm]);
          else if (sX) v = v * fs[K + ga];
          else if (sY) v = v * fs[2 * K + M + gm];
          out[(size_t)ga * M + gm] = (T)v;
        }
      }
    }
  }
}


// The same fold update as full ROWS of the output, nothing transposed: a workgroup owns 8 rows x
// (256 x 16 bytes) columns of the K x K result (all columns when K <= 512 in float64), a thread
// one 16-byte column piece of each of the 8 rows.  The update of element (a, b) is
// sum_i w_i * (x_ia * x_ib) -- the product of the two x commutes, so both triangles come out
// bit-identical without mirroring -- and every store is part of a contiguous run of whole rows:
// a 64x64 tile of a matrix whose rows are not multiples of 128 bytes (K = 500) is written in
// partial cache lines, which the memory system sustains at 2.8 TB/s; whole rows at 5.5 TB/s
// (tools/write_pattern.hip).  G is read for both triangles, so this kernel is for matrices that
// stay in L2 / MALL across the folds of a batch.  One barrier per fold; the column data (x of the
// validation rows, means, stds) goes straight from global memory to registers.
// Rows of the output a workgroup takes per fold (the panel): 32 KB of output either way -- 8 rows of float64, 16
// rows of float32 (round 4, same box: K = 500 float32 one-row folds 1.305 -> 0.990 ms with 16 rows; float64
// with 16 rows 0.845 -> 1.225 ms, with 4 rows the same as 8; tools/exp_rows_panel.sh).
#ifndef CVM_SR_ROWS
#define CVM_SR_ROWS 8
#endif
#ifndef CVM_SR_ROWS_F32
#define CVM_SR_ROWS_F32 16
#endif
// (float32 rows of more than 128 pieces -- K > 512 -- keep 8: sixteen passes per thread leave one workgroup per CU)
template <typename T> constexpr int sr_rows(int lpr) { return (sizeof(T) == 4 && lpr <= 128) ? CVM_SR_ROWS_F32 : CVM_SR_ROWS; }
// LPR: 16-byte pieces per output row handled by a workgroup (64, 128 or 256: the smallest that
// covers K keeps the threads busy); 256 / LPR rows go in one pass, SR_ROWS rows per workgroup.
template <typename T, bool WEIGHTED, int LPR>
__global__ __launch_bounds__(256) void small_rows_kernel(const SmallArgs a) {
  const int K = a.K, M = a.M;
  const int tid = threadIdx.x;
  const T *X = (const T *)a.X, *Y = (const T *)a.Y, *W = (const T *)a.w;
  const bool cX = a.flags & CVM_CENTER_X, cY = a.flags & CVM_CENTER_Y;
  const bool sX = a.flags & CVM_SCALE_X, sY = a.flags & CVM_SCALE_Y;
  const bool inl = a.inl_n >= 0;
  typedef typename std::conditional<sizeof(T) == 8, double, float>::type TS;
  constexpr int VW = 16 / (int)sizeof(T);
  constexpr int TC = LPR * VW;
  constexpr int SR_ROWS = sr_rows<T>(LPR);
  constexpr int RSTEP = 256 / LPR;                 // rows per pass
  constexpr int NP = SR_ROWS / RSTEP;              // passes = pieces per thread
  constexpr int NV = 2;                            // validation rows per fold this kernel takes
  typedef T vec_t __attribute__((ext_vector_type(VW)));
  typedef double dvec_t __attribute__((ext_vector_type(VW)));
  // two LDS buffers: fold ff computes from buffer ff & 1 while the data of fold ff + 1 is on its way
  __shared__ TS xr[2][NV][SR_ROWS];                // x of the validation rows at the panel's 8 rows
  __shared__ TS wxr[2][NV][SR_ROWS];               // w * x there (rounded like the reference's WX)
  __shared__ double wl[2][NV];
  __shared__ double str[2][2][SR_ROWS];            // mean, std of the panel's rows
  // XCD-contiguous ranges of (fold group, panel), as in small_apply_kernel
  const unsigned lin = blockIdx.x, tot = (unsigned)a.gx * (unsigned)a.gy;
  const unsigned per = (tot + 7) / 8;
  const unsigned item = a.noremap ? lin : (lin & 7) * per + (lin >> 3);
  if (item >= tot) return;
  const int bx = (int)(item % (unsigned)a.gx), by = (int)(item / (unsigned)a.gx);
  const int ncc = (K + TC - 1) / TC;               // column chunks
  const int rp = bx / ncc, cc = bx - rp * ncc;
  const int a0 = rp * SR_ROWS, b0 = cc * TC;
  const int rr = tid / LPR;                        // this thread's rows: rr + p * RSTEP
  const int gc = b0 + (tid - rr * LPR) * VW;       // K % VW == 0: a piece is inside or outside
  const bool col_ok = gc < K;
  const bool do_xx = a.out_XTX && col_ok;
  const T *Gt = (const T *)a.G;
  T gpre[NP][VW];
  if (do_xx) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
      if (a0 + rr + p * RSTEP < K) {
        const vec_t t = *reinterpret_cast<const vec_t *>(Gt + (size_t)(a0 + rr + p * RSTEP) * K + gc);
#pragma unroll
        for (int e = 0; e < VW; ++e) gpre[p][e] = t[e];
      }
  }
  // ---- what a fold needs, fetched one fold ahead (registers; the staged part goes to LDS at the
  //      top of its own iteration): the loads of fold ff + 1 are issued before the stores of fold ff
  struct Pre {
    int n;
    vec_t xc[NV];        // x of the validation rows at this thread's columns
    dvec_t muc, sdc;     // mean, std of this thread's columns
    T sx, sw;            // staged by thread r * 8 + i: x[row r][a0 + i], w[row r]
    double sst;          // staged by threads 0..15: mean / std of panel row
    double swt;          // the fold's training weight sum
    // XTY element (row i, column m) of thread i * M + m (8 M <= 256): y of the validation rows,
    // the four statistics it needs
    T yv[NV];
    double ya, yb, yc, yd;
  };
  // (every per-fold load sits in `fetch`, one fold ahead: a load issued and consumed inside an
  //  iteration would make the wave wait for everything older in the in-order memory counter --
  //  the next fold's prefetch and the previous stores -- and serialise the folds)
  const bool xty_fast = a.out_XTY && M > 0 && SR_ROWS * M <= 256;
  const int yi = tid / (M > 0 ? M : 1), ym = tid - yi * (M > 0 ? M : 1);
  const bool y_mine = xty_fast && cc == 0 && tid < SR_ROWS * M && a0 + yi < K;
  T y_h = (T)0;
  if (y_mine) y_h = ((const T *)a.H)[(size_t)(a0 + yi) * M + ym];
  const int f_lo = by * a.fpb;
  const int f_hi = (f_lo + a.fpb < a.nb) ? f_lo + a.fpb : a.nb;
  // the row numbers and row counts of ALL folds of this workgroup's group, once (round 3): inside the
  // per-fold prefetch a global load that depends on another global load (offsets -> indices -> x)
  // makes the wave wait for everything older in the in-order memory counter, the previous fold's
  // stores included, three times per fold
  __shared__ int64_t ridx_all[SA_FPB][NV];
  __shared__ int n_grp[SA_FPB];
  if (tid < SA_FPB * NV) {
    const int ff = tid / NV, u = tid - ff * NV;
    if (f_lo + ff < f_hi) {
      const int64_t o0 = inl ? 0 : a.offs[a.seg0 + f_lo + ff];
      const int n = inl ? a.inl_n : (int)(a.offs[a.seg0 + f_lo + ff + 1] - o0);
      ridx_all[ff][u] = u < n ? (inl ? a.inl[u] : a.idx[o0 + u]) : 0;
      if (u == 0) n_grp[ff] = n;
    }
  }
  lds_barrier();
  auto fetch = [&](int f, Pre &p) {
    const int64_t *ridx_f = ridx_all[f - f_lo];
    p.n = __builtin_amdgcn_readfirstlane(n_grp[f - f_lo]);
    const double *fs = a.fstats + (size_t)f * fstat_len(K, M);
#pragma unroll
    for (int e = 0; e < VW; ++e) { p.muc[e] = 0.0; p.sdc[e] = 1.0; }
    if (do_xx) {
#pragma unroll
      for (int u = 0; u < NV; ++u)
        if (u < p.n) {
          const int64_t ridx = ridx_f[u];
          p.xc[u] = *reinterpret_cast<const vec_t *>(X + ridx * (int64_t)K + gc);
        }
      if (cX) p.muc = *reinterpret_cast<const dvec_t *>(fs + gc);
      if (sX) p.sdc = *reinterpret_cast<const dvec_t *>(fs + K + gc);
    }
    p.sx = (T)0; p.sw = (T)1; p.sst = 0.0;
    if (tid < p.n * SR_ROWS) {
      const int r = tid / SR_ROWS, i = tid - r * SR_ROWS;
      const int64_t ridx = ridx_f[r];
      p.sw = WEIGHTED ? W[ridx] : (T)1;
      p.sx = (a0 + i < K) ? X[ridx * (int64_t)K + a0 + i] : (T)0;
    }
    if (tid < 2 * SR_ROWS) {
      const int which = tid / SR_ROWS, i = tid - which * SR_ROWS;
      double v = which ? 1.0 : 0.0;
      if (a0 + i < K) {
        if (!which && cX) v = fs[a0 + i];
        if (which && sX) v = fs[K + a0 + i];
      }
      p.sst = v;
    }
    p.swt = fs[2 * K + 2 * M];
    p.ya = p.yb = p.yc = p.yd = 0.0;
#pragma unroll
    for (int u = 0; u < NV; ++u) p.yv[u] = (T)0;
    if (y_mine) {
#pragma unroll
      for (int u = 0; u < NV; ++u)
        if (u < p.n) {
          const int64_t ridx = ridx_f[u];
          p.yv[u] = Y[ridx * (int64_t)M + ym];
        }
      if (cX || cY) { p.ya = fs[a0 + yi]; p.yb = fs[2 * K + ym]; }
      if (sX) p.yc = fs[K + a0 + yi];
      if (sY) p.yd = fs[2 * K + M + ym];
    }
  };
  Pre cur, nxt;
  if (f_lo < f_hi) fetch(f_lo, cur);
  for (int f = f_lo; f < f_hi; ++f) {
    const int b = (f - f_lo) & 1;
    const int n = cur.n;
    // stage this fold's small shared data (buffer b was last read two folds ago: one barrier per fold)
    if (tid < n * SR_ROWS) {
      const int r = tid / SR_ROWS, i = tid - r * SR_ROWS;
      xr[b][r][i] = (TS)cur.sx;
      wxr[b][r][i] = (TS)(WEIGHTED ? (T)(cur.sw * cur.sx) : cur.sx);
      if (i == 0) wl[b][r] = (double)cur.sw;
    }
    if (tid < 2 * SR_ROWS) str[b][tid / SR_ROWS][tid % SR_ROWS] = cur.sst;
    lds_barrier();                                // (LDS only: the previous fold's stores need no acknowledgement)
    if (f + 1 < f_hi) fetch(f + 1, nxt);          // in flight during the arithmetic and the stores below
    const double *fs = a.fstats + (size_t)f * fstat_len(K, M);
    const double swt = cur.swt;
    const size_t fo = (size_t)(a.seg0 + f);
    if (do_xx) {
      TS acc[NP][VW];
#pragma unroll
      for (int p = 0; p < NP; ++p)
#pragma unroll
        for (int e = 0; e < VW; ++e) acc[p][e] = 0;
#pragma unroll
      for (int u = 0; u < NV; ++u)
        if (u < n) {
          const TS wr = (TS)(T)wl[b][u];
#pragma unroll
          for (int p = 0; p < NP; ++p) {
            const TS rv = xr[b][u][rr + p * RSTEP];
#pragma unroll
            for (int e = 0; e < VW; ++e) acc[p][e] += WEIGHTED ? wr * (rv * (TS)cur.xc[u][e]) : rv * (TS)cur.xc[u][e];
          }
        }
      T *out = (T *)a.out_XTX + fo * (size_t)K * K;
#pragma unroll
      for (int p = 0; p < NP; ++p) {
        const int i = rr + p * RSTEP;
        if (a0 + i >= K) continue;
        const double mur = str[b][0][i], sdr = str[b][1][i];
        vec_t vv;
#pragma unroll
        for (int e = 0; e < VW; ++e) {
          double v = (double)gpre[p][e] - (double)acc[p][e];
          if (cX) v -= swt * (mur * cur.muc[e]);
          if (sX) v = v * (sdr * cur.sdc[e]);
          vv[e] = (T)v;
        }
        out_store(reinterpret_cast<vec_t *>(out + (size_t)(a0 + i) * K + gc), vv);
      }
    }
    // the panel's rows of XTY (first column chunk only)
    if (xty_fast) {
      if (y_mine) {
        TS sacc = 0;
#pragma unroll
        for (int u = 0; u < NV; ++u)
          if (u < n) sacc += wxr[b][u][yi] * (TS)cur.yv[u];
        double v = (double)y_h - (double)sacc;
        if (cX || cY) v -= swt * (cur.ya * cur.yb);
        if (sX && sY) v = v * (cur.yc * cur.yd);
        else if (sX) v = v * cur.yc;
        else if (sY) v = v * cur.yd;
        ((T *)a.out_XTY)[fo * (size_t)K * M + (size_t)(a0 + yi) * M + ym] = (T)v;
      }
    } else if (a.out_XTY && M > 0 && cc == 0) {
      const T *Ht = (const T *)a.H;
      T *out = (T *)a.out_XTY + fo * (size_t)K * M;
      const int64_t o0 = inl ? 0 : a.offs[a.seg0 + f];
      for (int e = tid; e < SR_ROWS * M; e += 256) {
        const int i = e / M, m = e - i * M;
        const int ga = a0 + i;
        if (ga >= K) continue;
        TS s = 0;
        for (int r = 0; r < n; ++r) {
          const int64_t ridx = inl ? a.inl[r] : a.idx[o0 + r];
          s += wxr[b][r][i] * (TS)Y[ridx * (int64_t)M + m];
        }
        double v = (double)Ht[(size_t)ga * M + m] - (double)s;
        if (cX || cY) v -= swt * (fs[ga] * fs[2 * K + m]);
        if (sX && sY) v = v * (fs[K + ga] * fs[2 * K + M + m]);
        else if (sX) v = v * fs[K + ga];
        else if (sY) v = v * fs[2 * K + M + m];
        out[(size_t)ga * M + m] = (T)v;
      }
    }
    cur = nxt;
  }
}
