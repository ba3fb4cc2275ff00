// colstats.hpp -- part of libcvmhip.so (included by cvmhip.hip inside its anonymous namespace).
// colstats_kernel: streaming column statistics of the validation rows (statistics-only calls and
// the fused route).
#pragma once

// ----------------------------------------------------------------------------------
// colstats_kernel: statistics-only fold stage (training_statistics, cvmatrix.py:519-574;
// SURVEY.md 8f-3).  HBM-bound: the validation rows are streamed once, nothing else is read.
//   unit = (fold, row split).  One-dimensional grid: first the Y workgroups, then one workgroup
//   per (unit, block of 256*VEC columns of X), the column blocks of a unit adjacent (one row of
//   X is one contiguous stretch for them).
//   X workgroup: a thread owns VEC = 16/sizeof(T) adjacent columns and walks the unit's rows in
//   order with COL_UNROLL rows in flight all the time (a register slot requests its next row as
//   soon as it has been summed), so the memory pipe never drains (tools/gather_probe.hip: 250
//   units of 400 rows 105 -> 70 us for 410 MB).  s += w x, q += (w x) x, sw += w all in that one row order, so
//   a constant-one column gives s == q == sw bit for bit (as in the Gram kernels).  The row
//   numbers and weights are wave-uniform (scalar loads).
//   Y workgroup (one per unit): chunks of rows staged through LDS by all threads, then thread c
//   sums column c in the same sequential row order (Mq = M rounded up to a power of two, <= 256).
// Output: the unit's statistics vector in the layout fold_stats_kernel reads
//   [ sX(Kp) | qX(Kp) | sY(Mp) | qY(Mp) | sw nz neg - ].
// ----------------------------------------------------------------------------------
struct ColArgs {
  const void *X, *Y, *w;
  const int64_t *idx, *offs;
  int64_t seg0;
  int splits;
  int nxb;              // column blocks of X
  int ywgs;             // Y workgroups (one per unit); mq = M rounded up to a power of two
  int mq;
  int64_t units;
  Geom g;               // tile_elems = h_elems = 0: a unit is its statistics vector
  char *ws;
};
constexpr int COL_THREADS = 256;
#ifndef CVM_COL_UNROLL
#define CVM_COL_UNROLL 8
#endif
constexpr int COL_UNROLL = CVM_COL_UNROLL;
constexpr int Y_ELEMS = 2048;                      // elements of Y staged per chunk
constexpr int Y_PASSES = Y_ELEMS / COL_THREADS;

template <typename T, bool WEIGHTED, bool ALIGNED>
__global__ __launch_bounds__(COL_THREADS) void colstats_kernel(const ColArgs a) {
  constexpr int VEC = 16 / (int)sizeof(T);
  constexpr int U = COL_UNROLL;
  typedef T vec_t __attribute__((ext_vector_type(VEC)));
  const Geom &g = a.g;
  const int K = g.K, M = g.M;
  const T *wp = (const T *)a.w;
  const int tid = threadIdx.x;
  const long b = blockIdx.x;
  if (b >= a.ywgs) {
    const long xu = b - a.ywgs;
    const long u = xu / a.nxb;
    const int xb = (int)(xu - u * a.nxb);
    const int seg = (int)(u / a.splits), sp = (int)(u - (long)seg * a.splits);
    const int64_t seg_begin = a.offs[a.seg0 + seg];
    const int64_t seg_rows = a.offs[a.seg0 + seg + 1] - seg_begin;
    int64_t r0, r1;
    split_range(seg_rows, a.splits, sp, r0, r1);
    const int64_t *idx = a.idx + seg_begin;
    double *st = unit_stats<T>(a.ws, g, u);
    const int c0 = (xb * COL_THREADS + tid) * VEC;
    const bool live = c0 < K;
    const T *Xp = (const T *)a.X + (live ? c0 : 0);
    double s[VEC], q[VEC], sw = 0, nz = 0, ng = 0;
#pragma unroll
    for (int v = 0; v < VEC; ++v) s[v] = q[v] = 0;
    // branch-free loads (a per-thread branch around a load costs a wait for everything in
    // flight): threads past the last column read column 0 and store nothing; unaligned rows go
    // by elements, the elements past K re-read the thread's first column and count as zero
    int cv[VEC];
#pragma unroll
    for (int v = 0; v < VEC; ++v) cv[v] = (live && c0 + v < K) ? v : 0;
    auto load = [&](int64_t row) -> vec_t {
      vec_t x;
      const T *p = Xp + row * (int64_t)K;
      if (ALIGNED) x = *reinterpret_cast<const vec_t *>(p);
      else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) { const T e = p[cv[v]]; x[v] = (c0 + v < K) ? e : (T)0; }
      }
      return x;
    };
    auto acc1 = [&](const vec_t &x, T wr) {
#pragma unroll
      for (int v = 0; v < VEC; ++v) {
        const T pv = WEIGHTED ? (T)(x[v] * wr) : x[v];
        s[v] += (double)pv; q[v] += (double)(T)(pv * x[v]);
      }
      if (WEIGHTED) { sw += (double)wr; nz += (wr != (T)0) ? 1.0 : 0.0; ng += (wr < (T)0) ? 1.0 : 0.0; }
    };
    // U rows in flight all the time: slot j holds row base + j; as soon as it has been summed
    // the slot requests row base + U + j.  The row numbers (wave-uniform, scalar loads) are
    // requested one round ahead of their rows, so a row load never waits for an index load.
    // Requests past the unit's end repeat its last row (a valid address) and are not summed.
    const int64_t n = r1 - r0;
    const int64_t last = r1 - 1;
    if (n > 0) {
      vec_t x[U];
      T wr[U];
      int64_t nxt[U];
      auto numbers = [&](int64_t rb) {
#pragma unroll
        for (int j = 0; j < U; ++j) nxt[j] = idx[rb + j < last ? rb + j : last];
      };
      numbers(r0);
#pragma unroll
      for (int j = 0; j < U; ++j) { x[j] = load(nxt[j]); wr[j] = WEIGHTED ? wp[nxt[j]] : (T)1; }
      numbers(r0 + U);
      int64_t base = 0;
      for (; base + U <= n; base += U) {
#pragma unroll
        for (int j = 0; j < U; ++j) {
          const vec_t xv = x[j];
          const T wv = wr[j];
          acc1(xv, wv);
          x[j] = load(nxt[j]);
          wr[j] = WEIGHTED ? wp[nxt[j]] : (T)1;
          // (the scheduler would otherwise gather the eight sums, wait for every row, and only
          // then issue the eight loads)
          __builtin_amdgcn_sched_barrier(0);
        }
        numbers(r0 + base + 2 * U);
      }
#pragma unroll
      for (int j = 0; j < U; ++j)
        if (base + j < n) acc1(x[j], wr[j]);
    }
    if (live) {
#pragma unroll
      for (int v = 0; v < VEC; ++v)
        if (c0 + v < K) { st[c0 + v] = s[v]; st[g.Kp + c0 + v] = q[v]; }
    }
    if (xb == 0 && tid == 0) {       // (the Y threads walk the rows in the same order: one sw serves both)
      if (!WEIGHTED) { sw = nz = (double)n; }
      st[2 * g.Kp + 2 * g.Mp + 0] = sw; st[2 * g.Kp + 2 * g.Mp + 1] = nz; st[2 * g.Kp + 2 * g.Mp + 2] = ng;
    }
    return;
  }
  // Y workgroup: unit b.  Chunks of Y_ELEMS / mq rows are staged through LDS by all 256 threads
  // (thread = (row of a pass, column); the next chunk's loads are in flight while this one is
  // summed), then thread c sums column c down the chunk in row order.
  const long u = b;
  const int seg = (int)(u / a.splits), sp = (int)(u - (long)seg * a.splits);
  const int64_t seg_begin = a.offs[a.seg0 + seg];
  const int64_t seg_rows = a.offs[a.seg0 + seg + 1] - seg_begin;
  int64_t r0, r1;
  split_range(seg_rows, a.splits, sp, r0, r1);
  const int64_t n = r1 - r0, last = r1 - 1;
  const int64_t *idx = a.idx + seg_begin;
  double *st = unit_stats<T>(a.ws, g, u);
  const T *Yp = (const T *)a.Y;
  __shared__ T ys[Y_ELEMS];
  __shared__ T wsm[Y_ELEMS];
  const int mq = a.mq;
  const int rp = COL_THREADS / mq;            // rows per pass
  const int rc = Y_ELEMS / mq;                // rows per chunk = Y_PASSES passes
  const int cl = tid & (mq - 1), rl = tid / mq;
  for (int cb = 0; cb < M; cb += COL_THREADS) {
    const int mc = M - cb < COL_THREADS ? M - cb : COL_THREADS;
    const int col = cb + (cl < mc ? cl : 0);
    double s = 0, q = 0;
    if (n > 0) {
      T yv[Y_PASSES], wv[Y_PASSES];
      auto request = [&](int64_t k) {
#pragma unroll
        for (int p = 0; p < Y_PASSES; ++p) {
          const int64_t r = r0 + k * rc + p * rp + rl;
          const int64_t row = idx[r < last ? r : last];
          yv[p] = Yp[row * (int64_t)M + col];
          wv[p] = WEIGHTED ? wp[row] : (T)1;
        }
      };
      const int64_t chunks = (n + rc - 1) / rc;
      request(0);
      for (int64_t k = 0; k < chunks; ++k) {
#pragma unroll
        for (int p = 0; p < Y_PASSES; ++p) {
          ys[(p * rp + rl) * mq + cl] = yv[p];
          if (cl == 0) wsm[p * rp + rl] = wv[p];
        }
        __syncthreads();
        if (k + 1 < chunks) request(k + 1);
        const int cnt = (int)(n - k * rc < rc ? n - k * rc : rc);
        if (tid < mc) {
#pragma unroll 8
          for (int r = 0; r < cnt; ++r) {
            const T y = ys[r * mq + tid], wr = wsm[r];
            const T pv = WEIGHTED ? (T)(y * wr) : y;
            s += (double)pv; q += (double)(T)(pv * y);
          }
        }
        __syncthreads();
      }
    }
    if (tid < mc) { st[2 * g.Kp + cb + tid] = s; st[2 * g.Kp + g.Mp + cb + tid] = q; }
  }
}

// launch geometry of colstats_kernel for nb folds of `splits` units
template <typename T> inline void colstats_shape(ColArgs &c, int K, int M, int64_t nb, int splits) {
  constexpr int VEC = 16 / (int)sizeof(T);
  c.splits = splits;
  c.nxb = (K + COL_THREADS * VEC - 1) / (COL_THREADS * VEC);
  int mq = 1;
  while (mq < M && mq < COL_THREADS) mq *= 2;
  c.mq = mq;
  c.units = nb * splits;
  c.ywgs = M > 0 ? (int)c.units : 0;
}
inline dim3 colstats_grid(const ColArgs &c, int64_t nb) {
  return dim3((unsigned)((int64_t)c.ywgs + nb * c.splits * c.nxb));
}

// Units per fold of a statistics-only launch.  Measured (tools/exp_colstats.sh, MI355X): the
// X workgroups of a launch all start at once (up to about five per CU are resident) and a CU
// streams at a fixed rate, so the launch lasts as long as the CU with the most workgroups:
// 250 workgroups of 400 rows 70 us, 260 of 385 rows 89 us (410 MB); past the resident limit the
// dispatcher balances.  Every unit also costs a statistics vector (written, then read by
// fold_stats_kernel: about 24 rows' worth) and 0.2 us of fold_stats_kernel's serial sum.
#ifndef CVM_COL_MIN_ROWS
#define CVM_COL_MIN_ROWS 32
#endif
inline int64_t colstats_splits(int64_t max_rows, int64_t n_folds, int K, size_t elem, int cu_count) {
  static const char *force = getenv("CVM_COL_SPLITS");     // experiments
  if (force) return atol(force) < 1 ? 1 : (atol(force) > 1024 ? 1024 : atol(force));
  const int vec = 16 / (int)elem;
  const int nxb = (K + COL_THREADS * vec - 1) / (COL_THREADS * vec);
  const double cus = (double)cu_count;
  int64_t cap = (max_rows + CVM_COL_MIN_ROWS - 1) / CVM_COL_MIN_ROWS;
  if (cap > 1024) cap = 1024;
  if (cap < 1) cap = 1;
  int64_t best = 1;
  double best_cost = 0;
  for (int64_t s = 1; s <= cap; ++s) {
    int64_t per = (max_rows + s - 1) / s;
    per = (per + STAGE_ROWS - 1) / STAGE_ROWS * STAGE_ROWS;      // (split_range)
    const double wgs = (double)n_folds * nxb * (double)s;
    const double rounds = wgs <= 5 * cus ? ceil(wgs / cus) : wgs / cus + 0.5;
    const double cost = rounds * ((double)per + 24.0) + 1.2 * (double)s;
    if (s == 1 || cost < best_cost) { best = s; best_cost = cost; }
  }
  return best;
}
