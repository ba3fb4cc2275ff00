// colstats.hpp -- part of libcvmhip.so (included by cvmhip.hip inside its anonymous namespace).
// colstats_kernel: streaming column statistics of the validation rows (statistics-only calls and
// the fused route).
#pragma once

// ----------------------------------------------------------------------------------
// colstats_kernel: statistics-only fold stage (training_statistics, cvmatrix.py:519-574;
// SURVEY.md 8f-3).  HBM-bound: the validation rows are streamed once, nothing else is read.
//   grid (column blocks of X + 1 block for Y, units); unit = (fold, row split)
//   a thread owns VEC = 16/sizeof(T) adjacent columns and walks the unit's rows in order,
//   eight rows in flight; s += w x, q += (w x) x, sw += w all in that one row order, so a
//   constant-one column gives s == q == sw bit for bit (as in the Gram kernels).
// Output: the unit's statistics vector in the layout fold_stats_kernel reads
//   [ sX(Kp) | qX(Kp) | sY(Mp) | qY(Mp) | sw nz neg - ].
// ----------------------------------------------------------------------------------
struct ColArgs {
  const void *X, *Y, *w;
  const int64_t *idx, *offs;
  int64_t seg0;
  int splits;
  Geom g;               // tile_elems = h_elems = 0: a unit is its statistics vector
  char *ws;
};
constexpr int COL_THREADS = 256;
#ifndef CVM_COL_UNROLL
#define CVM_COL_UNROLL 8
#endif
#ifndef CVM_COL_ROWS
#define CVM_COL_ROWS 256
#endif
constexpr int COL_UNROLL = CVM_COL_UNROLL;

template <typename T, bool WEIGHTED, bool ALIGNED>
__global__ __launch_bounds__(COL_THREADS) void colstats_kernel(const ColArgs a) {
  constexpr int VEC = 16 / (int)sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(VEC)));
  const Geom &g = a.g;
  const int K = g.K, M = g.M;
  const long u = blockIdx.y;
  const int seg = (int)(u / a.splits), sp = (int)(u - (long)seg * a.splits);
  const int64_t seg_begin = a.offs[a.seg0 + seg];
  const int64_t seg_rows = a.offs[a.seg0 + seg + 1] - seg_begin;
  int64_t r0, r1;
  split_range(seg_rows, a.splits, sp, r0, r1);
  const int64_t *idx = a.idx + seg_begin;
  const T *wp = (const T *)a.w;
  double *st = unit_stats<T>(a.ws, g, u);
  const int nxb = (K + COL_THREADS * VEC - 1) / (COL_THREADS * VEC);
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < nxb) {
    const int c0 = ((int)blockIdx.x * COL_THREADS + tid) * VEC;
    const bool live = c0 < K;
    const T *Xp = (const T *)a.X;
    double s[VEC], q[VEC], sw = 0, nz = 0, ng = 0;
#pragma unroll
    for (int v = 0; v < VEC; ++v) s[v] = q[v] = 0;
    auto load = [&](int64_t row) -> vec_t {
      vec_t x;
      const T *p = Xp + row * (int64_t)K + c0;
      if (ALIGNED && c0 + VEC <= K) x = *reinterpret_cast<const vec_t *>(p);
      else {
#pragma unroll
        for (int v = 0; v < VEC; ++v) x[v] = (c0 + v < K) ? p[v] : (T)0;
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
    int64_t r = r0;
    if (live) {
      // the row numbers of the NEXT group are requested before this group's rows: the row loads
      // then never wait for an index load (two dependent latencies per group otherwise)
      int64_t rows[COL_UNROLL], nrows[COL_UNROLL];
      if (r + COL_UNROLL <= r1) {
#pragma unroll
        for (int j = 0; j < COL_UNROLL; ++j) rows[j] = idx[r + j];
      }
      for (; r + COL_UNROLL <= r1; r += COL_UNROLL) {
        T wr[COL_UNROLL];
        vec_t x[COL_UNROLL];
        const bool more = r + 2 * COL_UNROLL <= r1;
#pragma unroll
        for (int j = 0; j < COL_UNROLL; ++j) nrows[j] = more ? idx[r + COL_UNROLL + j] : 0;
#pragma unroll
        for (int j = 0; j < COL_UNROLL; ++j) { x[j] = load(rows[j]); wr[j] = WEIGHTED ? wp[rows[j]] : (T)1; }
#pragma unroll
        for (int j = 0; j < COL_UNROLL; ++j) acc1(x[j], wr[j]);
#pragma unroll
        for (int j = 0; j < COL_UNROLL; ++j) rows[j] = nrows[j];
      }
      for (; r < r1; ++r) {
        const int64_t row = idx[r];
        acc1(load(row), WEIGHTED ? wp[row] : (T)1);
      }
#pragma unroll
      for (int v = 0; v < VEC; ++v)
        if (c0 + v < K) { st[c0 + v] = s[v]; st[g.Kp + c0 + v] = q[v]; }
    }
    if (blockIdx.x == 0 && tid == 0 && M == 0) {
      if (!WEIGHTED) { sw = nz = (double)(r1 - r0); }
      st[2 * g.Kp + 2 * g.Mp + 0] = sw; st[2 * g.Kp + 2 * g.Mp + 1] = nz; st[2 * g.Kp + 2 * g.Mp + 2] = ng;
    }
    return;
  }
  // the Y block: one column per thread (M is small), the same row order
  const T *Yp = (const T *)a.Y;
  for (int cb = 0; cb < M || cb == 0; cb += COL_THREADS) {
    const int c = cb + tid;
    const bool live = c < M;
    double s = 0, q = 0, sw = 0, nz = 0, ng = 0;
    int64_t r = r0;
    auto acc1 = [&](T y, T wr) {
      const T pv = WEIGHTED ? (T)(y * wr) : y;
      s += (double)pv; q += (double)(T)(pv * y);
      if (WEIGHTED) { sw += (double)wr; nz += (wr != (T)0) ? 1.0 : 0.0; ng += (wr < (T)0) ? 1.0 : 0.0; }
    };
    for (; r + COL_UNROLL <= r1; r += COL_UNROLL) {
      int64_t rows[COL_UNROLL];
      T wr[COL_UNROLL], y[COL_UNROLL];
#pragma unroll
      for (int j = 0; j < COL_UNROLL; ++j) rows[j] = idx[r + j];
#pragma unroll
      for (int j = 0; j < COL_UNROLL; ++j) {
        y[j] = live ? Yp[rows[j] * (int64_t)M + c] : (T)0;
        wr[j] = WEIGHTED ? wp[rows[j]] : (T)1;
      }
#pragma unroll
      for (int j = 0; j < COL_UNROLL; ++j) acc1(y[j], wr[j]);
    }
    for (; r < r1; ++r) {
      const int64_t row = idx[r];
      acc1(live ? Yp[row * (int64_t)M + c] : (T)0, WEIGHTED ? wp[row] : (T)1);
    }
    if (live) { st[2 * g.Kp + c] = s; st[2 * g.Kp + g.Mp + c] = q; }
    if (cb == 0 && tid == 0) {
      if (!WEIGHTED) { sw = nz = (double)(r1 - r0); }
      st[2 * g.Kp + 2 * g.Mp + 0] = sw; st[2 * g.Kp + 2 * g.Mp + 1] = nz; st[2 * g.Kp + 2 * g.Mp + 2] = ng;
    }
    if (M == 0) break;
  }
}
