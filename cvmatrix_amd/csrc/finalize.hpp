// finalize.hpp -- part of libcvmhip.so (included by cvmhip.hip inside its anonymous namespace).
// Finalize kernels: ordered reduction of the row-split partials, fold statistics, the
// subtract-and-correct update with mirrored stores.
#pragma once

// ----------------------------------------------------------------------------------
// finalize kernels
// ----------------------------------------------------------------------------------
// Output matrices are written once and never read again by the library: their 16-byte stores are
// nontemporal (-DCVM_NT_STORES=0 builds the plain-store variant).  Measured (tools/bench_small.py,
// same box): leave-one-out K=500 1.71 -> 2.06 M folds/s, K=512 n=8 1.25 -> 1.53 M, K=4096 float32
// n=16 4.55 -> 5.16 TB/s algorithmic, 3000 folds of 33 rows through the fused epilogue +8 %.
#ifndef CVM_NT_STORES
#define CVM_NT_STORES 1
#endif
template <typename V> __device__ __forceinline__ void out_store(V *p, V v) {
#if CVM_NT_STORES
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

// the write ceiling of the device with this kind of store (cvm_fill_probe): one 16-byte piece per thread, one
// workgroup per 4 KiB, in linear order.  (tools/fill_variants.hip: this shape reaches 6.7 TB/s where a
// grid-stride loop of the same stores reaches 4.6-5.3 and 64 KiB per workgroup 5.2 -- the memory system wants
// the stores that are in flight at one time to lie close together.)
__global__ __launch_bounds__(256) void fill_probe_kernel(float *buf, size_t pieces) {
  typedef float v4 __attribute__((ext_vector_type(4)));
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < pieces) out_store(reinterpret_cast<v4 *>(buf) + i, (v4){0.f, 0.f, 0.f, 0.f});
}

// Scaling by the training-set standard deviations (cvmatrix.py:1007-1010: XTX / (sd_a sd_b), XTY /
// (sd_a sd_y)): the per-fold statistics vector `fstats` holds the RECIPROCAL stds, one division per
// column and fold (fold_stats_kernel / small_stats_kernel), and every finish computes
// v * (isd_a * isd_b).  A float64 division per output element costs a dozen instructions at the
// float64 vector rate, and with K x K outputs per fold of a few rows that arithmetic -- not the
// memory system -- bounded the small-fold kernels (K = 4096, 16-row folds: 3.4 TB/s of stores with
// the division, 4.0 without any scaling).  Against the reference's v / (sd_a * sd_b) the result
// differs by at most ~2 ulp (three roundings instead of two); the 1e-10 parity bar is six orders
// above that, symmetry stays exact (the same two factors commute).
struct FinArgs {
  Geom g;
  int splits;           // slot stride of the partial workspace: unit (seg, sp) = seg * splits + sp
  int s_off, s_diag;    // row splits of the off-diagonal tiles / of everything else (WgramArgs)
  int n_sum;            // fit mode: segments whose partials are summed (1; the sweep: every fold)
  int n_seg;            // segments (folds) in this batch
  int64_t seg0;         // first fold of the batch (for output addressing)
  const char *ws;       // unit partials
  double *fstats;       // per fold of the batch: [muX(K) sdX(K) muY(M) sdY(M) swT pad..]
  const int64_t *offs;  // device offsets (fold sizes) or nullptr
  const void *w;        // non-null: weighted
  const void *G, *H;    // global Gram (fold mode)
  const double *gstats;
  void *out_XTX, *out_XTY, *out_muX, *out_sdX, *out_muY, *out_sdY;
  double *out_fold;
  double ddof, resolution;
  unsigned flags;
  int32_t *neg_flag;
  int gx, gy;           // apply_kernel<.., true>: sub-tiles + panels, folds of the launch
  int inline_stats;     // apply_kernel<.., true>: no fold_stats_kernel ran -- every workgroup derives the
                        // fold statistics its tile needs from the partials' column sums itself (same
                        // chains, same bits) and designated workgroups write the statistics outputs: one
                        // launch and one launch gap less per call of cvm_sweep_folds / cvm_sweep_fold_range
  int compact;          // fit mode over several segments (the one-sweep path, float64): every segment's
                        // subtotal -- the fold's raw update U_f = sum_sp partial, exactly the chain the fold
                        // stage forms -- is written back to the segment's slot 0, so that the fold stage
                        // reads ONE partial per fold and tile instead of s_off / s_diag (same bits: 0 + U_f)
};
__host__ __device__ inline size_t fstat_len(int K, int M) { return 2 * (size_t)K + 2 * (size_t)M + 4; }

// q-th partial of a class with `nsp` splits per segment, segment-major: its unit slot
__device__ __forceinline__ long sum_unit(int q, int nsp, int stride) {
  const int sg = q / nsp;
  return (long)sg * stride + (q - sg * nsp);
}
// the same slots visited in order, one step at a time (no division in the summing loops)
struct UnitCursor {
  int sp, nsp;
  long base, stride;     // current slot = base + sp
  __device__ __forceinline__ UnitCursor(int nsp_, int stride_) : sp(0), nsp(nsp_), base(0), stride(stride_) {}
  __device__ __forceinline__ long next() {
    const long u = base + sp;
    if (++sp == nsp) { sp = 0; base += stride; }
    return u;
  }
};

// column chunks (grid.y) of fold_stats_kernel: enough workgroups to fill the chip when there are
// few folds and many columns, one when there are many folds
inline int fold_stats_chunks(int K, int M, int64_t n_folds) {
  int c = (K + M + 255) / 256;
  const int64_t cap = n_folds >= 512 ? 1 : (512 + n_folds - 1) / n_folds;
  if (c > cap) c = (int)cap;
  return c < 1 ? 1 : c;
}

// fit: gstats = ordered sum of the split partials
template <typename T>
__device__ __forceinline__ void fit_stats_columns(const FinArgs &a, double *gstats, int c0, int stride) {
  const Geom &g = a.g;
  const int total = 2 * g.K + 2 * g.M + 3;
  for (int c = c0; c < total; c += stride) {
    int src;
    if (c < g.K) src = c;
    else if (c < 2 * g.K) src = g.Kp + (c - g.K);
    else if (c < 2 * g.K + g.M) src = 2 * g.Kp + (c - 2 * g.K);
    else if (c < 2 * g.K + 2 * g.M) src = 2 * g.Kp + g.Mp + (c - 2 * g.K - g.M);
    else src = 2 * g.Kp + 2 * g.Mp + (c - 2 * g.K - 2 * g.M);
    // segment by segment, in split order: s = sum_seg (sum_sp partial) -- the one-sweep path sums
    // the folds' own sums (one segment: the plain chain); sixteen loads in flight at a time
    double s = 0, us = 0;
    int p = 0, kk = 0;
    long segc = 0;                         // segments closed so far
    const int np = a.n_sum * a.s_diag;     // (the column sums come from the diagonal items)
    UnitCursor cur(a.s_diag, a.splits);
    auto close_seg = [&]() {
      if (a.compact) unit_stats<T>((char *)a.ws, g, segc * a.splits)[src] = us;
      s += us; us = 0; kk = 0; ++segc;
    };
    for (; p + 16 <= np; p += 16) {
      double v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = unit_stats<T>((char *)a.ws, g, cur.next())[src];
#pragma unroll
      for (int u = 0; u < 16; ++u) { us += v[u]; if (++kk == a.s_diag) close_seg(); }
    }
    for (; p < np; ++p) {
      us += unit_stats<T>((char *)a.ws, g, cur.next())[src];
      if (++kk == a.s_diag) close_seg();
    }
    if (c < total - 1) gstats[c] = s;
    else if (a.neg_flag) *a.neg_flag = (s > 0) ? 1 : 0;
  }
}
template <typename T> __global__ void fit_stats_kernel(const FinArgs a, double *gstats) {
  fit_stats_columns<T>(a, gstats, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// value of `v` in lane `j` (wave-uniform j) as a double
__device__ __forceinline__ double lane_value(double v, int j) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), j);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
  return __hiloint2double(hi, lo);
}

// One column of one fold: training-set mean and std from the fold's column sums (sv, qv), the
// full-data sums (gs, gq), the training-set weight sum and divisor; reference operation order
// (cvmatrix.py:1020, 1043, 1119-1128).  Writes the fold's statistics vector and the outputs.
template <typename T>
__device__ __forceinline__ void fold_column_finish(const FinArgs &a, int f, bool isX, int cc, double sv, double qv,
                                                   double gs, double gq, double swt, double divisor,
                                                   bool want_sd, double *fs) {
  const int K = a.g.K, M = a.g.M;
  const double st_ = gs - sv;          // cvmatrix.py:1020
  const double mu = st_ / swt;         // cvmatrix.py:1043
  double sd = 1.0;
  if (want_sd) {
    const double qt = gq - qv;
    double var = (-2 * mu * st_ + swt * (mu * mu) + qt) / divisor;   // 1119-1123
    var = (var < 0) ? 0.0 : var;       // np.maximum(var, 0): NaN stays NaN
    sd = sqrt(var);
    if (sd <= a.resolution) sd = 1.0;  // 1128
  }
  fs[isX ? cc : 2 * K + cc] = mu;
  fs[isX ? K + cc : 2 * K + M + cc] = 1.0 / sd;     // the finish MULTIPLIES by reciprocal stds (below)
  T *omu = (T *)(isX ? a.out_muX : a.out_muY), *osd = (T *)(isX ? a.out_sdX : a.out_sdY);
  const size_t o = (size_t)(a.seg0 + f) * (isX ? K : M) + cc;
  if (omu) omu[o] = (T)mu;
  if (osd && want_sd) osd[o] = (T)sd;
}

// The same arithmetic for workgroups that derive the statistics they need themselves
// (FinArgs::inline_stats): the fold's weight sum / non-zero count (every thread, the same loads),
template <typename T>
__device__ __forceinline__ void inline_fold_totals(const FinArgs &a, int f, double &swv, double &nzv, double &swt,
                                                   double &nzt, double &divisor) {
  const Geom &g = a.g;
  const int K = g.K, M = g.M;
  const long u0 = (long)f * a.splits;
  swv = 0; nzv = 0;
  if (a.w != nullptr) {
    for (int p = 0; p < a.s_diag; ++p) {               // split order, like fold_stats_kernel's lane sums
      const double *st = unit_stats<T>((char *)a.ws, g, u0 + p);
      swv += st[2 * g.Kp + 2 * g.Mp + 0]; nzv += st[2 * g.Kp + 2 * g.Mp + 1];
    }
  } else {
    swv = nzv = (double)(a.offs[a.seg0 + f + 1] - a.offs[a.seg0 + f]);
  }
  const double gsw = a.gstats[2 * K + 2 * M], gnz = a.gstats[2 * K + 2 * M + 1];
  swt = gsw - swv; nzt = gnz - nzv;
  divisor = (nzt - a.ddof) * swt / nzt;
}
// and one column: mean and RECIPROCAL std (std itself in `sd`), fold_column_finish's formulas
template <typename T>
__device__ __forceinline__ void inline_column_stat(const FinArgs &a, int f, bool isX, int cc, double swt, double divisor,
                                                   bool want_sd, double &mu, double &isd, double &sd) {
  const Geom &g = a.g;
  const int K = g.K, M = g.M;
  const long u0 = (long)f * a.splits;
  const int s_src = isX ? cc : 2 * g.Kp + cc;
  const int q_src = isX ? g.Kp + cc : 2 * g.Kp + g.Mp + cc;
  double sv = 0, qv = 0;
  for (int p = 0; p < a.s_diag; ++p) {
    const double *st = unit_stats<T>((char *)a.ws, g, u0 + p);
    sv += st[s_src]; qv += st[q_src];
  }
  const double gs = isX ? a.gstats[cc] : a.gstats[2 * K + cc];
  const double gq = isX ? a.gstats[K + cc] : a.gstats[2 * K + M + cc];
  const double st_ = gs - sv;          // cvmatrix.py:1020
  mu = st_ / swt;                      // cvmatrix.py:1043
  sd = 1.0;
  if (want_sd) {
    const double qt = gq - qv;
    double var = (-2 * mu * st_ + swt * (mu * mu) + qt) / divisor;   // 1119-1123
    var = (var < 0) ? 0.0 : var;
    sd = sqrt(var);
    if (sd <= a.resolution) sd = 1.0;  // 1128
  }
  isd = 1.0 / sd;
}

// fold: training-set mean / std of every column; reference operation order
// (cvmatrix.py:612-620, 709-745, 1043, 1079, 1119-1128)
template <typename T> __global__ __launch_bounds__(256) void fold_stats_kernel(const FinArgs a) {
  const Geom &g = a.g;
  const int f = blockIdx.x;
  const int K = g.K, M = g.M;
  const bool weighted = a.w != nullptr;
  const long u0 = (long)f * a.splits;
  double swv = 0, nzv = 0;
  if (weighted) {
    // lane l fetches partial p0 + l, the sums then run over the lanes in split order (one load
    // round per 64 partials instead of a chain of dependent scalar loads)
    const int lane = threadIdx.x & 63;
    for (int p0 = 0; p0 < a.s_diag; p0 += 64) {
      const int cnt = a.s_diag - p0 < 64 ? a.s_diag - p0 : 64;
      const double *st = unit_stats<T>((char *)a.ws, g, u0 + p0 + (lane < cnt ? lane : 0));
      const double sl = st[2 * g.Kp + 2 * g.Mp + 0], nl = st[2 * g.Kp + 2 * g.Mp + 1];
      for (int j = 0; j < cnt; ++j) { swv += lane_value(sl, j); nzv += lane_value(nl, j); }
    }
  } else {
    swv = nzv = (double)(a.offs[a.seg0 + f + 1] - a.offs[a.seg0 + f]);
  }
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
    const int s_src = isX ? cc : 2 * g.Kp + cc;
    const int q_src = isX ? g.Kp + cc : 2 * g.Kp + g.Mp + cc;
    // in split order; sixteen partials (32 loads) in flight, the adds stay one chain
    double sv = 0, qv = 0;
    int p = 0;
    for (; p + 16 <= a.s_diag; p += 16) {
      double s16[16], q16[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const double *st = unit_stats<T>((char *)a.ws, g, u0 + p + j);
        s16[j] = st[s_src]; q16[j] = st[q_src];
      }
      __builtin_amdgcn_sched_barrier(0);     // (all 32 loads issued before the first add waits)
#pragma unroll
      for (int j = 0; j < 16; ++j) { sv += s16[j]; qv += q16[j]; }
    }
#pragma unroll 4
    for (; p < a.s_diag; ++p) {
      const double *st = unit_stats<T>((char *)a.ws, g, u0 + p);
      sv += st[s_src]; qv += st[q_src];
    }
    const double gs = isX ? a.gstats[cc] : a.gstats[2 * K + cc];
    const double gq = isX ? a.gstats[K + cc] : a.gstats[2 * K + M + cc];
    fold_column_finish<T>(a, f, isX, cc, sv, qv, gs, gq, swt, divisor, isX ? want_sdX : want_sdY, fs);
  }
}

// Finish one 64x64 tile whose raw update (sum over the fold's rows of w*x_a*x_b) sits in
// Ts, and store it twice: as rows a / columns b and, off the diagonal, mirrored as rows b /
// columns a.  Row-contiguous mapping: a lane owns 16 contiguous bytes of one row, 64/VW
// lanes cover a 64-column row segment, so every wave instruction reads G and writes XTX in
// whole contiguous row segments.  Pass 0 finishes the tile in the reference's order
// (cvmatrix.py:1001-1010: total - update, - sw_T*(mu_a*mu_b), / (sd_a*sd_b)), parks the
// finished values in Ts and stores them; pass 1 stores the transposed Ts.  On a diagonal tile
// the lower triangle takes the update of its mirror element, so the result is exactly
// symmetric (G is, and the corrections are products of the same two factors).
constexpr int ST = 64;                 // tile edge of the finishing code
// `gpre`: the tile's 16-byte pieces of G already in registers, piece j = q-th with q = tid +
// j * nthreads (finish_tile_preload below) -- used where `full`; nullptr: load them here.
// `st`: the fold's statistics of this tile in LDS (stage_tile_stats below): [0,64) means of the
// tile's rows, [64,128) their stds, [128,192) means of its columns, [192,256) their stds.
template <typename T, bool FOLD, typename TS = double>
__device__ __forceinline__ void finish_store_tile(TS (*Ts)[ST + 1], bool diag, int a0, int b0, int K,
                                                  const T *Gt, T *out, const double *st, double swt,
                                                  bool cX, bool sX, int tid, int nthreads,
                                                  const T (*gpre)[16 / sizeof(T)] = nullptr) {
  constexpr int VW = 16 / sizeof(T);            // elements per 16-byte access
  constexpr int LPR = ST / VW;                  // lanes per row segment
  typedef T vst_t __attribute__((ext_vector_type(VW)));
  const bool vec_ok = ((size_t)K * sizeof(T)) % 16 == 0 && ((uintptr_t)out % 16 == 0) &&
                      (!FOLD || (uintptr_t)Gt % 16 == 0);
  for (int pass = 0; pass < (diag ? 1 : 2); ++pass) {
    const int r0g = pass ? b0 : a0, c0g = pass ? a0 : b0;
    int jq = 0;
    for (int q = tid; q < ST * LPR; q += nthreads, ++jq) {
      const int lr = q / LPR, lc = (q - lr * LPR) * VW;
      const int gr = r0g + lr, gc = c0g + lc;
      if (gr >= K || gc >= K) continue;
      const bool full = vec_ok && gc + VW <= K;
      T vals[VW];
      if (pass == 0) {
        T gvv[VW];
        if (FOLD) {
          if (full && gpre) {
#pragma unroll
            for (int e = 0; e < VW; ++e) gvv[e] = gpre[jq][e];
          } else if (full) {
            const vst_t t = *reinterpret_cast<const vst_t *>(Gt + (size_t)gr * K + gc);
#pragma unroll
            for (int e = 0; e < VW; ++e) gvv[e] = t[e];
          } else {
#pragma unroll
            for (int e = 0; e < VW; ++e) gvv[e] = (gc + e < K) ? Gt[(size_t)gr * K + gc + e] : (T)0;
          }
        }
        const double mur = (FOLD && cX) ? st[lr] : 0.0, sdr = (FOLD && sX) ? st[ST + lr] : 1.0;
#pragma unroll
        for (int e = 0; e < VW; ++e) {
          const int cc = lc + e, gce = gc + e;
          double v = 0;
          if (gce < K) {
            const double upd = (diag && lr > cc) ? Ts[cc][lr] : Ts[lr][cc];
            if (FOLD) {
              v = (double)gvv[e] - upd;
              if (cX) v -= swt * (mur * st[2 * ST + cc]);
              if (sX) v = v * (sdr * st[3 * ST + cc]);
            } else {
              v = upd;
            }
          }
          vals[e] = (T)v;
        }
      } else {
#pragma unroll
        for (int e = 0; e < VW; ++e) vals[e] = (T)Ts[lc + e][lr];   // finished, transposed
      }
      T *dst = out + (size_t)gr * K + gc;
      if (full) {
        vst_t vv;
#pragma unroll
        for (int e = 0; e < VW; ++e) vv[e] = vals[e];
        out_store(reinterpret_cast<vst_t *>(dst), vv);
      } else {
#pragma unroll
        for (int e = 0; e < VW; ++e) if (gc + e < K) dst[e] = vals[e];
      }
      if (pass == 0 && !diag) {
        // park the finished values in place (off the diagonal every raw element is read by
        // this thread only) for the mirrored pass
#pragma unroll
        for (int e = 0; e < VW; ++e) Ts[lr][lc + e] = (TS)vals[e];
      }
    }
    lds_barrier();     // (LDS only: the stores above need not be acknowledged before the mirrored pass)
  }
}

// The statistics a tile's finish needs, from the fold's vector `fs` ([muX(K) sdX(K) ...]) into LDS:
// one global load per thread instead of two per finished element.  (Callers put a barrier between
// this and finish_store_tile.)
__device__ __forceinline__ void stage_tile_stats(double *st, const double *fs, int a0, int b0, int K, int tid) {
  if (tid < 4 * ST) {
    const int part = tid / ST, l = tid - part * ST;
    const int col = ((part < 2) ? a0 : b0) + l;
    st[tid] = (col < K) ? fs[((part & 1) ? K : 0) + col] : ((part & 1) ? 1.0 : 0.0);
  }
}

// Issue the loads of a tile's G pieces early (before the work that produces the update): piece
// j of this thread as finish_store_tile numbers them; pieces outside the matrix or not
// 16-byte accessible are left for finish_store_tile to load.
template <typename T, int NQ>
__device__ __forceinline__ void finish_tile_preload(T (&gpre)[NQ][16 / sizeof(T)], int a0, int b0, int K,
                                                    const T *Gt, const T *out, int tid, int nthreads) {
  constexpr int VW = 16 / sizeof(T);
  constexpr int LPR = ST / VW;
  typedef T vst_t __attribute__((ext_vector_type(VW)));
  const bool vec_ok = ((size_t)K * sizeof(T)) % 16 == 0 && ((uintptr_t)out % 16 == 0) && ((uintptr_t)Gt % 16 == 0);
#pragma unroll
  for (int j = 0; j < NQ; ++j) {
    const int q = tid + j * nthreads;
    const int lr = q / LPR, lc = (q - lr * LPR) * VW;
    const int gr = a0 + lr, gc = b0 + lc;
    if (vec_ok && gr < K && gc + VW <= K) {
      const vst_t t = *reinterpret_cast<const vst_t *>(Gt + (size_t)gr * K + gc);
#pragma unroll
      for (int e = 0; e < VW; ++e) gpre[j][e] = t[e];
    }
  }
}

// The same finishing step for ONE WAVE inside wgram4_kernel<.., FUSED> (rows of K elements 16-byte
// aligned): the raw update of a 64x64 block is in Ts (in the accumulators' type), the row/column
// means and reciprocal stds in rs[0..255].  A wave has no other wave to hide its latency behind, so
// the G loads go out sixteen rows at a time.  A lane owns 16 bytes of a row (VW = 2 float64 /
// 4 float32 columns), the wave covers VW rows per instruction.
// Direct half: rows [row_lo, row_hi) of the 64-row block (multiples of 16).  Finished values are
// parked in Ts for the mirrored store (off the diagonal).
template <typename T, int TP>
__device__ __forceinline__ void fused_finish_direct(T (*Ts)[TP], const double *rs, bool diagb, int a0,
                                                    int b0, int K, const T *Gt, T *out,
                                                    double swt, bool cX, bool sX, int lane, int row_lo, int row_hi) {
  constexpr int VW = 16 / (int)sizeof(T);          // columns per lane
  constexpr int LPR = 64 / VW;                     // lanes per row
  constexpr int JB = 16 / VW;                      // wave instructions per 16 rows
  typedef T vt __attribute__((ext_vector_type(VW)));
  const int sub = lane / LPR, lc = VW * (lane - sub * LPR);
  const int gc = b0 + lc;
  const bool col_ok = gc < K;                      // K is a multiple of VW: the whole piece is inside
  double muc[VW], sdc[VW];
#pragma unroll
  for (int e = 0; e < VW; ++e) { muc[e] = rs[128 + lc + e]; sdc[e] = rs[192 + lc + e]; }
#pragma unroll 1
  for (int r0 = row_lo; r0 < row_hi; r0 += 16) {
    vt gv[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      const int lr = r0 + VW * j + sub, gr = a0 + lr;
      vt z;
#pragma unroll
      for (int e = 0; e < VW; ++e) z[e] = (T)0;
      gv[j] = (col_ok && gr < K) ? *reinterpret_cast<const vt *>(Gt + (size_t)gr * K + gc) : z;
    }
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      const int lr = r0 + VW * j + sub, gr = a0 + lr;
      if (!(col_ok && gr < K)) continue;
      const double mur = rs[lr], sdr = rs[64 + lr];
      vt vv;
#pragma unroll
      for (int e = 0; e < VW; ++e) {
        const double u = (double)((diagb && lr > lc + e) ? Ts[lc + e][lr] : Ts[lr][lc + e]);
        double v = (double)gv[j][e] - u;
        if (cX) v -= swt * (mur * muc[e]);
        if (sX) v = v * (sdr * sdc[e]);
        vv[e] = (T)v;
      }
      out_store(reinterpret_cast<vt *>(out + (size_t)gr * K + gc), vv);
      if (!diagb) {                                // parked for the mirrored store
#pragma unroll
        for (int e = 0; e < VW; ++e) Ts[lr][lc + e] = vv[e];
      }
    }
  }
}
// The same direct half for 32 rows of an OFF-DIAGONAL block with the G pieces already in registers:
// the loads are issued by fused_g_preload32 before the accumulators are dumped to LDS and before the
// workgroup barrier that follows (round 3: the two dependent rounds of G loads were 7.9 k cycles of
// an off-diagonal item's epilogue, profiles/r3/fused_epilogue_stamps.txt)
template <typename T> struct FusedPre {
  static constexpr int VW = 16 / (int)sizeof(T);
  static constexpr int JB = 16 / VW;
  typedef T vt __attribute__((ext_vector_type(VW)));
  vt gv[2][JB];
};
template <typename T>
__device__ __forceinline__ void fused_g_preload32(FusedPre<T> &p, int a0, int b0, int K, const T *Gt, int lane, int row_lo) {
  constexpr int VW = FusedPre<T>::VW, LPR = 64 / VW, JB = FusedPre<T>::JB;
  typedef typename FusedPre<T>::vt vt;
  const int sub = lane / LPR, lc = VW * (lane - sub * LPR);
  const int gc = b0 + lc;
  const int gcc = gc < K ? gc : 0;                 // (branch-free loads; pieces outside the matrix are never stored)
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      const int gr = a0 + row_lo + 16 * b + VW * j + sub;
      p.gv[b][j] = *reinterpret_cast<const vt *>(Gt + (size_t)(gr < K ? gr : 0) * K + gcc);
    }
}
template <typename T, int TP>
__device__ __forceinline__ void fused_finish_direct32(T (*Ts)[TP], const double *rs, int a0, int b0, int K, T *out,
                                                      double swt, bool cX, bool sX, int lane, int row_lo,
                                                      const FusedPre<T> &p) {
  constexpr int VW = FusedPre<T>::VW, LPR = 64 / VW, JB = FusedPre<T>::JB;
  typedef typename FusedPre<T>::vt vt;
  const int sub = lane / LPR, lc = VW * (lane - sub * LPR);
  const int gc = b0 + lc;
  const bool col_ok = gc < K;
  double muc[VW], sdc[VW];
#pragma unroll
  for (int e = 0; e < VW; ++e) { muc[e] = rs[128 + lc + e]; sdc[e] = rs[192 + lc + e]; }
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int j = 0; j < JB; ++j) {
      const int lr = row_lo + 16 * b + VW * j + sub, gr = a0 + lr;
      if (!(col_ok && gr < K)) continue;
      const double mur = rs[lr], sdr = rs[64 + lr];
      vt vv;
#pragma unroll
      for (int e = 0; e < VW; ++e) {
        double v = (double)p.gv[b][j][e] - (double)Ts[lr][lc + e];
        if (cX) v -= swt * (mur * muc[e]);
        if (sX) v = v * (sdr * sdc[e]);
        vv[e] = (T)v;
      }
      out_store(reinterpret_cast<vt *>(out + (size_t)gr * K + gc), vv);
#pragma unroll
      for (int e = 0; e < VW; ++e) Ts[lr][lc + e] = vv[e];          // parked for the mirrored store
    }
}
// Mirrored half: rows b0 + r for r in [row_lo, row_hi), columns a0..; out[b0 + r][a0 + c] = finished[c][r]
template <typename T, int TP>
__device__ __forceinline__ void fused_finish_mirror(T (*Ts)[TP], int a0, int b0, int K, T *out, int lane,
                                                    int row_lo, int row_hi) {
  constexpr int VW = 16 / (int)sizeof(T);
  constexpr int LPR = 64 / VW;
  typedef T vt __attribute__((ext_vector_type(VW)));
  const int sub = lane / LPR, lc = VW * (lane - sub * LPR);
  const int gc2 = a0 + lc;
  if (gc2 >= K) return;
#pragma unroll 4
  for (int r = row_lo; r < row_hi; r += VW) {
    const int lr = r + sub, gr = b0 + lr;
    if (gr >= K) continue;
    vt vv;
#pragma unroll
    for (int e = 0; e < VW; ++e) vv[e] = Ts[lc + e][lr];
    out_store(reinterpret_cast<vt *>(out + (size_t)gr * K + gc2), vv);
  }
}
template <typename T, int TP>
__device__ __forceinline__ void fused_finish_block(T (*Ts)[TP], const double *rs, bool diagb, int a0,
                                                   int b0, int K, const T *Gt, T *out,
                                                   double swt, bool cX, bool sX, int lane) {
  fused_finish_direct<T, TP>(Ts, rs, diagb, a0, b0, K, Gt, out, swt, cX, sX, lane, 0, 64);
  if (diagb) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  fused_finish_mirror<T, TP>(Ts, a0, b0, K, out, lane, 0, 64);
}

// One 64x64 sub-tile of a 128x128 upper tile (or one 128 x M panel of H) of one segment:
// ordered sum of the split partials (16-byte loads, four in flight, added in split order)
// staged in LDS, then finish_store_tile: total - partial, rank-1 centring, outer-std scaling
// and the mirrored store.  HBM-bound.
constexpr int APPLY_THREADS = 256;        // fold mode: one workgroup per (fold, sub-tile), plenty of them
constexpr int APPLY_THREADS_FIT = 1024;   // fit mode: 44 workgroups at K = 512 -- more threads each
constexpr int APPLY_SUB = 4;   // 64x64 sub-tiles per 128x128 tile
constexpr int APPLY_INLINE_MAXM = 512;   // responses up to which a workgroup keeps the Y statistics itself (inline_stats)
template <typename T, bool FOLD>
__global__ __launch_bounds__(FOLD ? APPLY_THREADS : APPLY_THREADS_FIT) void apply_kernel(const FinArgs a) {
  constexpr int NTHR = FOLD ? APPLY_THREADS : APPLY_THREADS_FIT;
  const Geom &g = a.g;
  // ONE block of LDS for both kinds of workgroup (static arrays of different scopes are not overlaid by the
  // compiler: the XTY workgroups' statistics buffers used to sit on top of the tile buffer and took every
  // launch from four resident workgroups per CU to three): a tile workgroup uses [tile | its statistics], an
  // XTY workgroup [xs_l | ys_l] in the same bytes
  static_assert(2 * TILE + 2 * APPLY_INLINE_MAXM <= ST * (ST + 1) + 4 * ST, "the XTY workgroups' buffers fit the tile's");
  __shared__ __attribute__((aligned(16))) double apply_lds[ST * (ST + 1) + 4 * ST];
  // fold mode: a 1-D launch of 8 * ceil(gx * gy / 8) workgroups; the 8 XCDs take workgroups
  // round-robin, so workgroup `lin` works on item (lin % 8) * per + lin / 8: every XCD gets a
  // contiguous range of (fold, sub-tile) -- neighbouring sub-tiles and the G tiles they read meet
  // in one L2
  int f, x;
  if (FOLD) {
    const unsigned lin = blockIdx.x, tot = (unsigned)a.gx * (unsigned)a.gy;
    const unsigned per = (tot + 7) / 8;
    const unsigned item = (lin & 7) * per + (lin >> 3);
    if (item >= tot) return;
    x = (int)(item % (unsigned)a.gx);
    f = (int)(item / (unsigned)a.gx);
  } else {
    f = blockIdx.y;
    x = blockIdx.x;
  }
  const int K = g.K, M = g.M;
  const long u0 = (long)f * a.splits;
  const bool inl = FOLD && a.inline_stats;
  const double *fs = (FOLD && !inl) ? a.fstats + (size_t)f * fstat_len(K, M) : nullptr;
  double swt = (FOLD && !inl) ? fs[2 * K + 2 * M] : 0.0;
  double divisor = 0, swv = 0, nzv = 0, nzt = 0;
  if (inl) inline_fold_totals<T>(a, f, swv, nzv, swt, nzt, divisor);
  const bool cX = a.flags & CVM_CENTER_X, cY = a.flags & CVM_CENTER_Y;
  const bool sX = a.flags & CVM_SCALE_X, sY = a.flags & CVM_SCALE_Y;
  const bool rXTY = a.flags & CVM_RET_XTY;
  const bool want_muX = cX || sX || (rXTY && cY), want_sdX = sX;
  const bool want_muY = rXTY && (cX || cY || sY), want_sdY = rXTY && sY;
  // who writes the statistics outputs (inline_stats): the X statistics of a 128-column panel by the
  // panel's XTY workgroup when XTY is produced, else (XTX only) by the diagonal 64 x 64 sub-tiles;
  // the Y statistics and the per-fold diagnostics by the first XTY workgroup / sub-tile 0
  const bool xty_wgs = a.out_XTY && M > 0;
  const size_t fo = (size_t)(a.seg0 + f);
  const char *ws0 = a.ws + (size_t)u0 * g.unit_bytes;
  if (inl && x == 0 && threadIdx.x == 0 && a.out_fold) {
    double *o = a.out_fold + 4 * (a.seg0 + f);
    o[0] = swt; o[1] = nzt; o[2] = swv; o[3] = nzv;
  }
  if (x < g.nTiles * APPLY_SUB) {
    if (!a.out_XTX) return;
    const int t = x / APPLY_SUB, sub = x - t * APPLY_SUB;
    int ti, tj;
    decode_tile(t, g.P, ti, tj);
    const int si = sub >> 1, sj = sub & 1;
    if (ti == tj && si > sj) return;                 // strictly lower: mirror of sub-tile (0,1)
    const int a0 = ti * TILE + si * ST, b0 = tj * TILE + sj * ST;
    if (a0 >= K || b0 >= K) return;
    double *sm = apply_lds, *st_lds = apply_lds + ST * (ST + 1);
    double (*Ts)[ST + 1] = reinterpret_cast<double (*)[ST + 1]>(sm);
    constexpr int VW = 16 / sizeof(T);
    constexpr int LPR = ST / VW;
    typedef T vld_t __attribute__((ext_vector_type(VW)));
    const int tid = threadIdx.x;
    if (FOLD && !inl) stage_tile_stats(st_lds, fs, a0, b0, K, tid);   // (visible after the barrier below)
    if (inl && tid < 4 * ST) {
      const int part = tid / ST, l = tid - part * ST;
      const int col = ((part < 2) ? a0 : b0) + l;
      double v = (part & 1) ? 1.0 : 0.0;
      if (col < K && want_muX) {
        double mu, isd, sd;
        inline_column_stat<T>(a, f, true, col, swt, divisor, want_sdX, mu, isd, sd);
        v = (part & 1) ? isd : mu;
        if (!xty_wgs && ti == tj && si == sj && part < 2) {            // this sub-tile owns columns a0 .. a0 + 63
          const size_t o = fo * K + col;
          if (part == 0 && a.out_muX) ((T *)a.out_muX)[o] = (T)mu;
          if (part == 1 && a.out_sdX && want_sdX) ((T *)a.out_sdX)[o] = (T)sd;
        }
      }
      st_lds[tid] = v;
    }
    // every thread owns NQ 16-byte pieces of the sub-tile; the splits are summed in order, the
    // pieces of one split loaded together (NQ independent loads in flight: a fit with 25 splits
    // is otherwise one long chain of dependent latencies on 44 workgroups)
    constexpr int NQ = ST * LPR / NTHR;
    double v[NQ][VW];
    const char *pp[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int q = tid + j * NTHR;
      const int lr = q / LPR, lc = (q - lr * LPR) * VW;
      const size_t off = (size_t)t * TILE * TILE + (size_t)(si * ST + lr) * TILE + sj * ST + lc;
      pp[j] = ws0 + off * sizeof(T);
#pragma unroll
      for (int e = 0; e < VW; ++e) v[j][e] = 0;
    }
    // UP splits' pieces requested before the first is added (fit mode runs on few workgroups:
    // it needs the loads of several splits in flight per thread)
    constexpr int UP = FOLD ? 2 : 8;
    // the partials of this tile: one per row split of its class (fold mode: of this fold; fit
    // mode: of every summed segment, segment-major)
    const int nsp = (ti == tj) ? a.s_diag : a.s_off;
    const int np = FOLD ? nsp : a.n_sum * nsp;
    auto slot = [&](int q) -> size_t { return (size_t)(FOLD ? (long)q : sum_unit(q, nsp, a.splits)) * g.unit_bytes; };
    // fit mode over several segments (the one-sweep path): segment sums first, then their sum
    // (v = sum_seg (sum_sp partial)); one segment or fold mode: the plain chain in uu, moved to v
    double uu[NQ][VW];
#pragma unroll
    for (int j = 0; j < NQ; ++j)
#pragma unroll
      for (int e = 0; e < VW; ++e) uu[j][e] = 0;
    int kk = 0;
    auto close_segment = [&]() {
#pragma unroll
      for (int j = 0; j < NQ; ++j)
#pragma unroll
        for (int e = 0; e < VW; ++e) { v[j][e] += uu[j][e]; uu[j][e] = 0; }
      kk = 0;
    };
    int p = 0;
    for (; p + UP <= np; p += UP) {
      vld_t qv[UP][NQ];
#pragma unroll
      for (int u = 0; u < UP; ++u) {
        const size_t so = slot(p + u);
#pragma unroll
        for (int j = 0; j < NQ; ++j) qv[u][j] = *reinterpret_cast<const vld_t *>(pp[j] + so);
      }
#pragma unroll
      for (int u = 0; u < UP; ++u) {
#pragma unroll
        for (int j = 0; j < NQ; ++j)
#pragma unroll
          for (int e = 0; e < VW; ++e) uu[j][e] += (double)qv[u][j][e];
        if (++kk == nsp) close_segment();
      }
    }
    for (; p < np; ++p) {
      vld_t qv[NQ];
      const size_t so = slot(p);
#pragma unroll
      for (int j = 0; j < NQ; ++j) qv[j] = *reinterpret_cast<const vld_t *>(pp[j] + so);
#pragma unroll
      for (int j = 0; j < NQ; ++j)
#pragma unroll
        for (int e = 0; e < VW; ++e) uu[j][e] += (double)qv[j][e];
      if (++kk == nsp) close_segment();
    }
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int q = tid + j * NTHR;
      const int lr = q / LPR, lc = (q - lr * LPR) * VW;
#pragma unroll
      for (int e = 0; e < VW; ++e) Ts[lr][lc + e] = v[j][e];
    }
    __syncthreads();
    T *out = (T *)a.out_XTX + (FOLD ? fo * (size_t)K * K : 0);
    finish_store_tile<T, FOLD>(Ts, ti == tj && si == sj, a0, b0, K, (const T *)a.G, out, st_lds, swt, cX, sX,
                               tid, NTHR);
  } else {
    if (!a.out_XTY || M == 0) return;
    const int ti = x - g.nTiles * APPLY_SUB;
    T *out = (T *)a.out_XTY + (FOLD ? fo * (size_t)K * M : 0);
    const T *Ht = (const T *)a.H;
    const size_t hoff = (g.tile_elems * sizeof(T) + 255) / 256 * 256;
    // inline_stats: this workgroup derives (and writes out) the X statistics of its 128 rows and, panel
    // 0, the Y statistics; M <= APPLY_INLINE_MAXM (the host checks)
    double *xs_l = apply_lds, *ys_l = apply_lds + 2 * TILE;
    if (inl) {
      const int tid = threadIdx.x;
      for (int q = tid; q < 2 * TILE; q += NTHR) {
        const int part = q / TILE, l = q - part * TILE, col = ti * TILE + l;
        double v = part ? 1.0 : 0.0;
        if (col < K && want_muX) {
          double mu, isd, sd;
          inline_column_stat<T>(a, f, true, col, swt, divisor, want_sdX, mu, isd, sd);
          v = part ? isd : mu;
          const size_t o = fo * K + col;
          if (part == 0 && a.out_muX) ((T *)a.out_muX)[o] = (T)mu;
          if (part == 1 && a.out_sdX && want_sdX) ((T *)a.out_sdX)[o] = (T)sd;
        }
        xs_l[q] = v;
      }
      for (int q = tid; q < 2 * M; q += NTHR) {
        const int part = q / M, m = q - part * M;
        double v = part ? 1.0 : 0.0;
        if (want_muY) {
          double mu, isd, sd;
          inline_column_stat<T>(a, f, false, m, swt, divisor, want_sdY, mu, isd, sd);
          v = part ? isd : mu;
          if (ti == 0) {
            const size_t o = fo * M + m;
            if (part == 0 && a.out_muY) ((T *)a.out_muY)[o] = (T)mu;
            if (part == 1 && a.out_sdY && want_sdY) ((T *)a.out_sdY)[o] = (T)sd;
          }
        }
        ys_l[part * APPLY_INLINE_MAXM + m] = v;
      }
      __syncthreads();
    }
    for (int e = threadIdx.x; e < TILE * M; e += NTHR) {
      const int ra = e / M, m = e - ra * M;
      const int ga = ti * TILE + ra;
      if (ga >= K) continue;
      double v = 0;
      const char *pp = ws0 + hoff + ((size_t)ga * g.Mp + m) * sizeof(T);
      const int np = FOLD ? a.s_diag : a.n_sum * a.s_diag;
      double us = 0;
      int kk = 0;
#pragma unroll 4
      for (int p = 0; p < np; ++p) {
        us += (double)*reinterpret_cast<const T *>(pp + (size_t)(FOLD ? (long)p : sum_unit(p, a.s_diag, a.splits)) * g.unit_bytes);
        if (++kk == a.s_diag) { v += us; us = 0; kk = 0; }     // (segment sums, then their sum)
      }
      if (FOLD) {
        const double mx = inl ? xs_l[ra] : fs[ga], ix = inl ? xs_l[TILE + ra] : fs[K + ga];
        const double my = inl ? ys_l[m] : fs[2 * K + m], iy = inl ? ys_l[APPLY_INLINE_MAXM + m] : fs[2 * K + M + m];
        v = (double)Ht[(size_t)ga * M + m] - v;
        if (cX || cY) v -= swt * (mx * my);
        if (sX && sY) v = v * (ix * iy);
        else if (sX) v = v * ix;
        else if (sY) v = v * iy;
      }
      out[(size_t)ga * M + m] = (T)v;
    }
  }
}

// Fit mode on many workgroups: the full-data matrices as the ordered sum of the units' partials.
// A 64x64 sub-tile is cut into FIT_RC row chunks, one 256-thread workgroup each (176 workgroups
// at K = 512 instead of 44: the sum of 25-50 partials per element is bound by what a CU can pull).
// Rows are stored as they are summed; the mirrored block goes through a [rows][65] LDS transpose.
// On a diagonal sub-tile only the upper triangle is stored, twice (as is and mirrored), so the
// result is exactly symmetric.  Needs 16-byte aligned rows (K * sizeof(T) % 16 == 0).
// (8-row chunks for float64, 16-row chunks for float32: one 16-byte piece per thread, sixteen
//  partials requested before the first is added -- the sweep sums 40-70 partials per element and a
//  workgroup is bound by the round trips, not by the bytes)
template <typename T> constexpr int fit_rc() { return sizeof(T) == 8 ? 8 : 4; }
constexpr int FIT_THREADS = 256;
constexpr int FIT_PCH = 4;             // row chunks of a 128-row XTY panel
constexpr int FIT_STAT_WGS = 8;        // workgroups that sum the column statistics (a.gstats: output)
template <typename T>
__global__ __launch_bounds__(FIT_THREADS) void fit_apply_kernel(const FinArgs a) {
  const Geom &g = a.g;
  const int K = g.K, M = g.M;
  const int x = blockIdx.x, tid = threadIdx.x;
  constexpr int FIT_RC = fit_rc<T>(), FIT_RH = ST / FIT_RC;
  const int n_sub = g.nTiles * APPLY_SUB * FIT_RC;
  constexpr int VW = 16 / sizeof(T);
  typedef T vld_t __attribute__((ext_vector_type(VW)));
  if (x < n_sub) {
    if (!a.out_XTX) return;
    const int t = x / (APPLY_SUB * FIT_RC);
    const int rem = x - t * (APPLY_SUB * FIT_RC);
    const int sub = rem / FIT_RC, ch = rem - sub * FIT_RC;
    int ti, tj;
    decode_tile(t, g.P, ti, tj);
    const int si = sub >> 1, sj = sub & 1;
    if (ti == tj && si > sj) return;                 // strictly lower: mirror of sub-tile (0,1)
    const int a0 = ti * TILE + si * ST + ch * FIT_RH, b0 = tj * TILE + sj * ST;
    if (a0 >= K || b0 >= K) return;
    const bool diag = (ti == tj && si == sj);
    __shared__ double Ts[FIT_RH][ST + 1];
    constexpr int LPR = ST / VW;
    constexpr int NQ = FIT_RH * LPR / FIT_THREADS;   // one 16-byte piece per thread
    static_assert(NQ >= 1, "chunk smaller than the workgroup");
    double v[NQ][VW];
    const char *pp[NQ];
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int q = tid + j * FIT_THREADS;
      const int lr = q / LPR, lc = (q - lr * LPR) * VW;
      const size_t off = (size_t)t * TILE * TILE + (size_t)(si * ST + ch * FIT_RH + lr) * TILE + sj * ST + lc;
      pp[j] = a.ws + off * sizeof(T);
#pragma unroll
      for (int e = 0; e < VW; ++e) v[j][e] = 0;
    }
    constexpr int UP = 16;                           // splits requested before the first is added
    const int nsp = (ti == tj) ? a.s_diag : a.s_off;
    const int np = a.n_sum * nsp;                    // segment-major, split order within a segment
    UnitCursor cur(nsp, a.splits);
    // segment sums first, then their sum (v = sum_seg (sum_sp partial): the one-sweep path's
    // G = sum_f G_Vf with G_Vf exactly the update the fold stage subtracts; one segment: the chain)
    double uu[NQ][VW];
#pragma unroll
    for (int j = 0; j < NQ; ++j)
#pragma unroll
      for (int e = 0; e < VW; ++e) uu[j][e] = 0;
    int kk = 0;
    long segc = 0;                                   // segments closed so far
    auto close_segment = [&]() {
      if (a.compact) {                               // the fold's raw update goes back to its slot 0
        const size_t so = (size_t)(segc * a.splits) * g.unit_bytes;
#pragma unroll
        for (int j = 0; j < NQ; ++j) {
          vld_t uv;
#pragma unroll
          for (int e = 0; e < VW; ++e) uv[e] = (T)uu[j][e];
          *reinterpret_cast<vld_t *>(const_cast<char *>(pp[j]) + so) = uv;
        }
      }
#pragma unroll
      for (int j = 0; j < NQ; ++j)
#pragma unroll
        for (int e = 0; e < VW; ++e) { v[j][e] += uu[j][e]; uu[j][e] = 0; }
      kk = 0; ++segc;
    };
    int p = 0;
    for (; p + UP <= np; p += UP) {
      vld_t qv[UP][NQ];
#pragma unroll
      for (int u = 0; u < UP; ++u) {
        const size_t so = (size_t)cur.next() * g.unit_bytes;
#pragma unroll
        for (int j = 0; j < NQ; ++j) qv[u][j] = *reinterpret_cast<const vld_t *>(pp[j] + so);
      }
#pragma unroll
      for (int u = 0; u < UP; ++u) {
#pragma unroll
        for (int j = 0; j < NQ; ++j)
#pragma unroll
          for (int e = 0; e < VW; ++e) uu[j][e] += (double)qv[u][j][e];
        if (++kk == nsp) close_segment();
      }
    }
    for (; p < np; ++p) {
      const size_t so = (size_t)cur.next() * g.unit_bytes;
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const vld_t qv = *reinterpret_cast<const vld_t *>(pp[j] + so);
#pragma unroll
        for (int e = 0; e < VW; ++e) uu[j][e] += (double)qv[e];
      }
      if (++kk == nsp) close_segment();
    }
    T *out = (T *)a.out_XTX;
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int q = tid + j * FIT_THREADS;
      const int lr = q / LPR, lc = (q - lr * LPR) * VW;
#pragma unroll
      for (int e = 0; e < VW; ++e) Ts[lr][lc + e] = v[j][e];
      const int gr = a0 + lr, gc = b0 + lc;
      if (gr >= K || gc >= K) continue;
      T *dst = out + (size_t)gr * K + gc;
      if (!diag && gc + VW <= K) {
        vld_t vv;
#pragma unroll
        for (int e = 0; e < VW; ++e) vv[e] = (T)v[j][e];
        *reinterpret_cast<vld_t *>(dst) = vv;
      } else {
#pragma unroll
        for (int e = 0; e < VW; ++e)
          if (gc + e < K && (!diag || gc + e >= gr)) dst[e] = (T)v[j][e];
      }
    }
    __syncthreads();
    // mirrored: rows b0 + c, columns a0 + rr (contiguous over rr)
    constexpr int PPR = FIT_RH / VW;                 // 16-byte pieces per mirrored row
    for (int q = tid; q < ST * PPR; q += FIT_THREADS) {
      const int c = q / PPR, rr = (q - c * PPR) * VW;
      const int gr = b0 + c, gc = a0 + rr;
      if (gr >= K || gc >= K) continue;
      T *dst = out + (size_t)gr * K + gc;
      if (!diag && gc + VW <= K) {
        vld_t vv;
#pragma unroll
        for (int e = 0; e < VW; ++e) vv[e] = (T)Ts[rr + e][c];
        *reinterpret_cast<vld_t *>(dst) = vv;
      } else {
#pragma unroll
        for (int e = 0; e < VW; ++e)
          if (gc + e < K && (!diag || gr > gc + e)) dst[e] = (T)Ts[rr + e][c];
      }
    }
  } else if (x >= n_sub + g.P * FIT_PCH) {
    // the column statistics ride in the same launch (FIT_STAT_WGS workgroups): one kernel less
    // between the Gram kernel and the fold stage
    const int b = x - (n_sub + g.P * FIT_PCH);
    fit_stats_columns<T>(a, const_cast<double *>(a.gstats), b * FIT_THREADS + tid, FIT_STAT_WGS * FIT_THREADS);
  } else {
    if (!a.out_XTY || M == 0) return;
    const int x2 = x - n_sub;
    const int ti = x2 / FIT_PCH, pc = x2 - ti * FIT_PCH;
    constexpr int PR = TILE / FIT_PCH;
    T *out = (T *)a.out_XTY;
    const size_t hoff = (g.tile_elems * sizeof(T) + 255) / 256 * 256;
    for (int e = tid; e < PR * M; e += FIT_THREADS) {
      const int ra = e / M, m = e - ra * M;
      const int ga = ti * TILE + pc * PR + ra;
      if (ga >= K) continue;
      double s = 0, us = 0;
      const char *pp = a.ws + hoff + ((size_t)ga * g.Mp + m) * sizeof(T);
      int p = 0, kk = 0;
      long segc = 0;
      const int np = a.n_sum * a.s_diag;
      UnitCursor cur(a.s_diag, a.splits);
      auto close_seg = [&]() {
        if (a.compact) *reinterpret_cast<T *>(const_cast<char *>(pp) + (size_t)(segc * a.splits) * g.unit_bytes) = (T)us;
        s += us; us = 0; kk = 0; ++segc;
      };
      for (; p + 16 <= np; p += 16) {
        T t16[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t16[u] = *reinterpret_cast<const T *>(pp + (size_t)cur.next() * g.unit_bytes);
#pragma unroll
        for (int u = 0; u < 16; ++u) { us += (double)t16[u]; if (++kk == a.s_diag) close_seg(); }
      }
      for (; p < np; ++p) {
        us += (double)*reinterpret_cast<const T *>(pp + (size_t)cur.next() * g.unit_bytes);
        if (++kk == a.s_diag) close_seg();
      }
      out[(size_t)ga * M + m] = (T)s;
    }
  }
}


// ----------------------------------------------------------------------------------
// One-sweep path with few folds (<= SWF_MAX), everything in two launches after the Gram kernel:
//   sweep_stats_kernel   full-data column sums (gstats) AND every fold's statistics
//   sweep_finish_kernel  full-data G, H AND every fold's training matrices, each partial read once
// (fit_apply_kernel + fold_stats_kernel + apply_kernel read the folds' partials twice: 68 MB of the
// 91 MB they move at C3.)  A fold's raw update U_f = sum_sp partial stays in registers; G = sum_f
// U_f in fold order -- the very sums fit_apply_kernel and apply_kernel form, so the results are
// bit-identical to cvm_sweep_fit + cvm_sweep_folds.
// ----------------------------------------------------------------------------------
constexpr int SWF_MAX = 16;    // folds whose updates a thread holds
#ifndef CVM_SWF_R
#define CVM_SWF_R 16
#endif
constexpr int SWF_R = CVM_SWF_R;      // block rows of sweep_finish_kernel (its columns: 256 bytes)
constexpr int SWF_T = 16 * SWF_R;     // its threads: one 16-byte piece each
constexpr int SWF_EPW = SWF_T / SWF_MAX;   // XTY elements per workgroup
#ifndef CVM_SWF_GROUP
#define CVM_SWF_GROUP 8
#endif
constexpr int SWF_GROUP = CVM_SWF_GROUP;   // output matrices finished per barrier round

// thread = (column cl of the block's 16, fold fl): the fold's column sums from its s_diag partials,
// the full-data sums over the folds through LDS, then the fold's mean / std
template <typename T> __global__ __launch_bounds__(256) void sweep_stats_kernel(const FinArgs a, double *gstats) {
  const Geom &g = a.g;
  const int K = g.K, M = g.M, P = a.n_seg;
  const int tid = threadIdx.x, cl = tid & 15, fl = tid >> 4;
  const int c = blockIdx.x * 16 + cl;
  const bool cvalid = c < K + M, fvalid = fl < P;
  const bool isX = c < K;
  const int cc = isX ? c : c - K;
  const int s_src = isX ? cc : 2 * g.Kp + cc;
  const int q_src = isX ? g.Kp + cc : 2 * g.Kp + g.Mp + cc;
  __shared__ double ss[SWF_MAX][17], qq[SWF_MAX][17], tot[3][SWF_MAX];
  const bool weighted = a.w != nullptr;
  double sv = 0, qv = 0, tv = 0;
  const long u0 = (long)fl * a.splits;
  if (fvalid) {
    const int t_src = (int)(2 * g.Kp + 2 * g.Mp) + (cl < 3 ? cl : 0);    // sw, nz, neg of the fold (cl 0..2)
    for (int p0 = 0; p0 < a.s_diag; p0 += 8) {
      const int cnt = a.s_diag - p0 < 8 ? a.s_diag - p0 : 8;
      double s8[8], q8[8], t8[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const double *st = unit_stats<T>((char *)a.ws, g, u0 + p0 + (j < cnt ? j : 0));
        s8[j] = cvalid ? st[s_src] : 0.0; q8[j] = cvalid ? st[q_src] : 0.0; t8[j] = st[t_src];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < cnt) { sv += s8[j]; qv += q8[j]; tv += t8[j]; }
    }
    ss[fl][cl] = sv; qq[fl][cl] = qv;
    if (cl < 3) tot[cl][fl] = tv;
  }
  __syncthreads();
  double gs = 0, gq = 0, gsw = 0, gnz = 0, gng = 0;
  for (int f = 0; f < P; ++f) { gs += ss[f][cl]; gq += qq[f][cl]; gsw += tot[0][f]; gnz += tot[1][f]; gng += tot[2][f]; }
  if (fl == 0 && cvalid) {
    gstats[isX ? cc : 2 * K + cc] = gs;
    gstats[isX ? K + cc : 2 * K + M + cc] = gq;
  }
  if (blockIdx.x == 0 && tid == 0) {
    gstats[2 * K + 2 * M] = gsw; gstats[2 * K + 2 * M + 1] = gnz;
    if (a.neg_flag) *a.neg_flag = (gng > 0) ? 1 : 0;
  }
  if (!fvalid || !a.fstats) return;
  double swv, nzv;
  if (weighted) { swv = tot[0][fl]; nzv = tot[1][fl]; }
  else swv = nzv = (double)(a.offs[a.seg0 + fl + 1] - a.offs[a.seg0 + fl]);
  const double swt = gsw - swv, nzt = gnz - nzv;
  const double divisor = (nzt - a.ddof) * swt / nzt;
  double *fs = a.fstats + (size_t)fl * fstat_len(K, M);
  if (blockIdx.x == 0 && cl == 0) {
    fs[2 * K + 2 * M] = swt;
    if (a.out_fold) {
      double *o = a.out_fold + 4 * (a.seg0 + fl);
      o[0] = swt; o[1] = nzt; o[2] = swv; o[3] = nzv;
    }
  }
  const bool cX = a.flags & CVM_CENTER_X, cY = a.flags & CVM_CENTER_Y;
  const bool sX = a.flags & CVM_SCALE_X, sY = a.flags & CVM_SCALE_Y;
  const bool rXTY = a.flags & CVM_RET_XTY;
  const bool want_muX = cX || sX || (rXTY && cY), want_sdX = sX;
  const bool want_muY = rXTY && (cX || cY || sY), want_sdY = rXTY && sY;
  if (!cvalid || (isX ? !want_muX : !want_muY)) return;
  fold_column_finish<T>(a, fl, isX, cc, sv, qv, gs, gq, swt, divisor, isX ? want_sdX : want_sdY, fs);
}

// Blocks of SWF_R rows x (256 bytes of) columns over the upper triangle of the K x K matrices (a
// block is inside one 128 x 128 tile of the partials), then ceil(K M / 256) workgroups for XTY.
// A thread owns one 16-byte piece: the folds' updates of it in registers, their sum = its piece of
// G; elements below the diagonal are masked (the partials hold nothing there), every finished
// matrix is stored as rows a / columns b and, through an LDS transpose, as rows b / columns a.
// Needs 16-byte aligned rows (K * sizeof(T) % 16 == 0) and P <= SWF_MAX.
template <typename T> __global__ __launch_bounds__(SWF_T) void sweep_finish_kernel(const FinArgs a, T *Gout, T *Hout) {
  const Geom &g = a.g;
  const int K = g.K, M = g.M, P = a.n_seg;
  const int tid = threadIdx.x;
  constexpr int VW = 16 / (int)sizeof(T);
  constexpr int C = 16 * VW;                 // block columns
  typedef T vld_t __attribute__((ext_vector_type(VW)));
  const int nrb = (K + SWF_R - 1) / SWF_R, ncb = (K + C - 1) / C;
  const bool cX = a.flags & CVM_CENTER_X, cY = a.flags & CVM_CENTER_Y;
  const bool sX = a.flags & CVM_SCALE_X, sY = a.flags & CVM_SCALE_Y;
  const int x = blockIdx.x;
  if (x >= nrb * ncb) {
    // ---- XTY: thread = (element el of the workgroup's 16, fold fl): the fold's update from its
    // s_diag partials, H = their sum over the folds through LDS, then the fold's own result
    if (M == 0 || !Hout) return;
    const int el = tid % SWF_EPW, fl = tid / SWF_EPW;
    const int e = (x - nrb * ncb) * SWF_EPW + el;
    const bool evalid = e < K * M, fvalid = fl < P;
    const int ga = evalid ? e / M : 0, m = evalid ? e - ga * M : 0;
    const size_t hoff = (g.tile_elems * sizeof(T) + 255) / 256 * 256;
    __shared__ double uh[SWF_MAX][17];
    double u = 0;
    if (fvalid) {
      const char *pf = a.ws + hoff + ((size_t)ga * g.Mp + m) * sizeof(T) + (size_t)fl * a.splits * g.unit_bytes;
      for (int p0 = 0; p0 < a.s_diag; p0 += 8) {
        const int cnt = a.s_diag - p0 < 8 ? a.s_diag - p0 : 8;
        T t8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t8[j] = *reinterpret_cast<const T *>(pf + (size_t)(p0 + (j < cnt ? j : 0)) * g.unit_bytes);
#pragma unroll
        for (int j = 0; j < 8; ++j) if (j < cnt) u += (double)t8[j];
      }
      uh[fl][el] = u;
    }
    lds_barrier();
    double h = 0;
    for (int f = 0; f < P; ++f) h += uh[f][el];
    const T hT = (T)h;
    if (!evalid) return;
    if (fl == 0) Hout[(size_t)ga * M + m] = hT;
    if (!a.out_XTY || !fvalid) return;
    {
      const double *fs = a.fstats + (size_t)fl * fstat_len(K, M);
      const double swt = fs[2 * K + 2 * M];
      double v = (double)hT - u;
      if (cX || cY) v -= swt * (fs[ga] * fs[2 * K + m]);
      if (sX && sY) v = v * (fs[K + ga] * fs[2 * K + M + m]);
      else if (sX) v = v * fs[K + ga];
      else if (sY) v = v * fs[2 * K + M + m];
      ((T *)a.out_XTY)[((size_t)(a.seg0 + fl) * K + ga) * M + m] = (T)v;
    }
    return;
  }
  const int rb = x / ncb, cb = x - rb * ncb;
  const int a0 = rb * SWF_R, b0 = cb * C;
  if (b0 + C - 1 < a0) return;               // the whole block is below the diagonal
  const int lr = tid >> 4, lc = (tid & 15) * VW;
  const int gr = a0 + lr, gc = b0 + lc;
  const int ti = a0 / TILE, tj = b0 / TILE;
  const int nsp = (ti == tj) ? a.s_diag : a.s_off;
  const size_t off = (size_t)tile_id(ti, tj, g.P) * TILE * TILE + (size_t)(a0 - ti * TILE + lr) * TILE + (b0 - tj * TILE + lc);
  const char *pp = a.ws + off * sizeof(T);
  // the folds' statistics of this block: [f][0..15] row means, [16..31] row 1/sd, [32..32+C) column
  // means, [32+C..32+2C) column 1/sd, [32+2C] sw_T
  constexpr int SL = 32 + 2 * C + 1;
  __shared__ double stl[SWF_MAX][SL];
  __shared__ T tm[SWF_GROUP][SWF_R][C + 1];
  if (a.out_XTX) {
    for (int q = tid; q < P * SL; q += SWF_T) {
      const int f = q / SL, i = q - f * SL;
      const double *fs = a.fstats + (size_t)f * fstat_len(K, M);
      double v;
      if (i < 16) v = (cX && a0 + i < K) ? fs[a0 + i] : 0.0;
      else if (i < 32) v = (sX && a0 + i - 16 < K) ? fs[K + a0 + i - 16] : 1.0;
      else if (i < 32 + C) v = (cX && b0 + i - 32 < K) ? fs[b0 + i - 32] : 0.0;
      else if (i < 32 + 2 * C) v = (sX && b0 + i - 32 - C < K) ? fs[K + b0 + i - 32 - C] : 1.0;
      else v = fs[2 * K + 2 * M];
      stl[f][i] = v;
    }
  }
  // the folds' updates of this thread's piece, in split order (up to 8 partials requested at a
  // time), and their sum in fold order
  double U[SWF_MAX][VW], gsum[VW];
#pragma unroll
  for (int e = 0; e < VW; ++e) gsum[e] = 0;
  vld_t qv[2][8];
  auto request = [&](int f, int p0, vld_t (&q)[8]) {
    const char *pf = pp + (size_t)f * a.splits * g.unit_bytes;
    const int cnt = nsp - p0 < 8 ? nsp - p0 : 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) q[j] = *reinterpret_cast<const vld_t *>(pf + (size_t)(p0 + (j < cnt ? j : 0)) * g.unit_bytes);
  };
  auto add = [&](int p0, const vld_t (&q)[8], double (&u)[VW]) {
    const int cnt = nsp - p0 < 8 ? nsp - p0 : 8;
#pragma unroll
    for (int j = 0; j < 8; ++j)
      if (j < cnt) {
#pragma unroll
        for (int e = 0; e < VW; ++e) u[e] += (double)q[j][e];
      }
  };
  request(0, 0, qv[0]);
#pragma unroll
  for (int f = 0; f < SWF_MAX; ++f) {
#pragma unroll
    for (int e = 0; e < VW; ++e) U[f][e] = 0;
    if (f < P) {
      // (requesting fold f + 1 BEFORE fold f is added was measured slower, 25 -> 49 us: the unrolled
      //  code with its masked tails grows past the instruction cache; tools/exp_sweepfin.sh)
      add(0, qv[f & 1], U[f]);
      for (int p0 = 8; p0 < nsp; p0 += 8) { request(f, p0, qv[f & 1]); add(p0, qv[f & 1], U[f]); }
      if (f + 1 < P) request(f + 1, 0, qv[(f + 1) & 1]);
#pragma unroll
      for (int e = 0; e < VW; ++e) gsum[e] += U[f][e];
    }
  }
  // element (gr, gc + e) is this thread's to finish when it is on or above the diagonal
  bool ok[VW];
  const bool all_ok = gr < K && gc + VW <= K && gr <= gc;
#pragma unroll
  for (int e = 0; e < VW; ++e) ok[e] = gr < K && gc + e < K && gr <= gc + e;
  T gT[VW];
#pragma unroll
  for (int e = 0; e < VW; ++e) gT[e] = (T)gsum[e];
  // output matrix o: 0 = G, o >= 1 = fold o - 1.  SWF_GROUP matrices per round: the finished
  // pieces are stored as rows a / columns b and parked in LDS; after one barrier the images
  // are stored transposed as rows b0 + c / columns a0 + r (strictly below the diagonal).
  const int mc = tid / (SWF_R / VW), mr = (tid - mc * (SWF_R / VW)) * VW;
  const int n_out = a.out_XTX ? P + 1 : 1;
  lds_barrier();                                   // (stl)
  for (int o0 = 0; o0 < n_out; o0 += SWF_GROUP) {
#pragma unroll
    for (int k = 0; k < SWF_GROUP; ++k) {
      const int o = o0 + k;
      if (o < n_out) {
        T vals[VW];
        T *out;
        if (o == 0) {
#pragma unroll
          for (int e = 0; e < VW; ++e) vals[e] = gT[e];
          out = Gout;
        } else {
          // (o0 is 0 or a multiple of SWF_GROUP: the fold number is known at compile time in
          //  each of the unrolled cases below, so U stays in registers)
          double uf[VW];
#pragma unroll
          for (int e = 0; e < VW; ++e) uf[e] = 0;
#pragma unroll
          for (int f = 0; f < SWF_MAX; ++f)
            if (f == o - 1) {
#pragma unroll
              for (int e = 0; e < VW; ++e) uf[e] = U[f][e];
            }
          const double *st = stl[o - 1];
          const double swt = st[32 + 2 * C], mur = st[lr], sdr = st[16 + lr];
#pragma unroll
          for (int e = 0; e < VW; ++e) {
            double v = (double)gT[e] - uf[e];
            if (cX) v -= swt * (mur * st[32 + lc + e]);
            if (sX) v = v * (sdr * st[32 + C + lc + e]);
            vals[e] = (T)v;
          }
          out = (T *)a.out_XTX + (size_t)(a.seg0 + o - 1) * K * K;
        }
        T *dst = out + (size_t)gr * K + gc;
        if (all_ok) {
          vld_t vv;
#pragma unroll
          for (int e = 0; e < VW; ++e) vv[e] = vals[e];
          if (o) out_store(reinterpret_cast<vld_t *>(dst), vv); else *reinterpret_cast<vld_t *>(dst) = vv;
        } else {
#pragma unroll
          for (int e = 0; e < VW; ++e) if (ok[e]) dst[e] = vals[e];
        }
#pragma unroll
        for (int e = 0; e < VW; ++e) tm[k][lr][lc + e] = vals[e];
      }
    }
    lds_barrier();
    const int orow = b0 + mc, ocol = a0 + mr;
    if (orow < K) {
#pragma unroll
      for (int k = 0; k < SWF_GROUP; ++k) {
        const int o = o0 + k;
        if (o < n_out) {
          T *out = o ? (T *)a.out_XTX + (size_t)(a.seg0 + o - 1) * K * K : Gout;
          T *md = out + (size_t)orow * K + ocol;
          if (ocol + VW <= K && ocol + VW - 1 < orow) {
            vld_t vv;
#pragma unroll
            for (int e = 0; e < VW; ++e) vv[e] = tm[k][mr + e][mc];
            if (o) out_store(reinterpret_cast<vld_t *>(md), vv); else *reinterpret_cast<vld_t *>(md) = vv;
          } else {
#pragma unroll
            for (int e = 0; e < VW; ++e) if (ocol + e < K && ocol + e < orow) md[e] = tm[k][mr + e][mc];
          }
        }
      }
    }
    lds_barrier();
  }
}

// ----------------------------------------------------------------------------------
// sweep_finish4_kernel (round 6): the same blocks, sums and arithmetic as sweep_finish_kernel, with the folds' partial
// loads dealt over FOUR groups of 256 threads.  The 256-thread kernel walks its <= 16 folds one after the other --
// request up to 8 partials of a fold, wait, add, next fold: ten dependent memory round trips per thread at C3, with
// barely more than one workgroup per CU (272 blocks): 25 us for 84 MB, 3.4 TB/s, latency-bound (profiles/r6/
// pmc_derived.json).  Here thread (piece, group g) loads the folds f = g, g + 4, ... only (2-3 round trips), leaves
// each fold's update U_f in LDS as float64 -- exactly the value the other kernel keeps in a register --, and after one
// barrier every thread forms G = sum_f U_f in fold order from LDS (the same chain: the same bits) and group g finishes
// and stores the outputs o = g, g + 4, ... (o = 0: the full-data matrix; o >= 1: fold o - 1), four matrices per
// barrier round through four transposition buffers.  Dynamic LDS: P x 256 pieces x 16 or 32 bytes + the statistics
// + the buffers (65 KB at C3); the host falls back to sweep_finish_kernel when that does not fit.
// ----------------------------------------------------------------------------------
// PPR = 16-byte pieces per block row: 16 (blocks of 16 rows x 256 bytes, 1024 threads, 272 blocks at K = 512) or 8 (16 rows x
// 128 bytes, 512 threads, 528 blocks: two to four workgroups per CU, so that one's loads overlap another's stores).
constexpr int SWF4_FG = 4;
#ifndef CVM_SWF4_PPR
#define CVM_SWF4_PPR 8
#endif
template <int PPR> constexpr int swf4_threads() { return 16 * PPR * SWF4_FG; }
template <int PPR> constexpr int swf4_epw() { return swf4_threads<PPR>() / SWF_MAX; }      // XTY elements per workgroup
template <typename T, int PPR> constexpr size_t swf4_lds_bytes(int P) {
  constexpr int VW = 16 / (int)sizeof(T), C = PPR * VW, SL = 32 + 2 * C + 1, NP = 16 * PPR;
  const size_t xtx = (size_t)P * NP * VW * 8 + (size_t)P * SL * 8 + (size_t)SWF4_FG * 16 * (C + 1) * sizeof(T) + 16;
  const size_t xty = (size_t)SWF_MAX * (swf4_epw<PPR>() + 1) * 8;
  return xtx > xty ? xtx : xty;
}
template <typename T, int PPR> __global__ __launch_bounds__(16 * PPR * SWF4_FG) void sweep_finish4_kernel(const FinArgs a, T *Gout, T *Hout) {
  constexpr int SWF4_T = swf4_threads<PPR>(), SWF4_EPW = swf4_epw<PPR>(), NP = 16 * PPR;
  static_assert(SWF_R == 16, "blocks of 16 rows x 256 bytes");
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const Geom &g = a.g;
  const int K = g.K, M = g.M, P = a.n_seg;
  const int tid = threadIdx.x;
  constexpr int VW = 16 / (int)sizeof(T);
  constexpr int C = PPR * VW;                // block columns
  typedef T vld_t __attribute__((ext_vector_type(VW)));
  const int nrb = (K + SWF_R - 1) / SWF_R, ncb = (K + C - 1) / C;
  const bool cX = a.flags & CVM_CENTER_X, cY = a.flags & CVM_CENTER_Y;
  const bool sX = a.flags & CVM_SCALE_X, sY = a.flags & CVM_SCALE_Y;
  const int x = blockIdx.x;
  if (x >= nrb * ncb) {
    // ---- XTY: thread = (element el of the workgroup's EPW, fold fl): as in sweep_finish_kernel
    if (M == 0 || !Hout) return;
    const int el = tid % SWF4_EPW, fl = tid / SWF4_EPW;
    const int e = (x - nrb * ncb) * SWF4_EPW + el;
    const bool evalid = e < K * M, fvalid = fl < P;
    const int ga = evalid ? e / M : 0, m = evalid ? e - ga * M : 0;
    const size_t hoff = (g.tile_elems * sizeof(T) + 255) / 256 * 256;
    double (*uh)[SWF4_EPW + 1] = reinterpret_cast<double (*)[SWF4_EPW + 1]>(dsm);     // (SWF_MAX x (EPW + 1) doubles)
    double u = 0;
    if (fvalid) {
      const char *pf = a.ws + hoff + ((size_t)ga * g.Mp + m) * sizeof(T) + (size_t)fl * a.splits * g.unit_bytes;
      for (int p0 = 0; p0 < a.s_diag; p0 += 8) {
        const int cnt = a.s_diag - p0 < 8 ? a.s_diag - p0 : 8;
        T t8[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) t8[j] = *reinterpret_cast<const T *>(pf + (size_t)(p0 + (j < cnt ? j : 0)) * g.unit_bytes);
#pragma unroll
        for (int j = 0; j < 8; ++j) if (j < cnt) u += (double)t8[j];
      }
      uh[fl][el] = u;
    }
    lds_barrier();
    double h = 0;
    for (int f = 0; f < P; ++f) h += uh[f][el];
    const T hT = (T)h;
    if (!evalid) return;
    if (fl == 0) Hout[(size_t)ga * M + m] = hT;
    if (!a.out_XTY || !fvalid) return;
    {
      const double *fs = a.fstats + (size_t)fl * fstat_len(K, M);
      const double swt = fs[2 * K + 2 * M];
      double v = (double)hT - u;
      if (cX || cY) v -= swt * (fs[ga] * fs[2 * K + m]);
      if (sX && sY) v = v * (fs[K + ga] * fs[2 * K + M + m]);
      else if (sX) v = v * fs[K + ga];
      else if (sY) v = v * fs[2 * K + M + m];
      ((T *)a.out_XTY)[((size_t)(a.seg0 + fl) * K + ga) * M + m] = (T)v;
    }
    return;
  }
  const int rb = x / ncb, cb = x - rb * ncb;
  const int a0 = rb * SWF_R, b0 = cb * C;
  if (b0 + C - 1 < a0) return;               // the whole block is below the diagonal
  const int pid = tid % NP, fg = __builtin_amdgcn_readfirstlane(tid / NP);      // piece of the block, fold group
  const int lr = pid / PPR, lc = (pid % PPR) * VW;
  const int gr = a0 + lr, gc = b0 + lc;
  const int ti = a0 / TILE, tj = b0 / TILE;
  const int nsp = (ti == tj) ? a.s_diag : a.s_off;
#ifdef CVM_SWF_CONTIG_PROBE
  // (timing probe only, WRONG results: what the kernel would take if a block's pieces of a partial tile were 4 KB in a row)
  const size_t off = (size_t)tile_id(ti, tj, g.P) * TILE * TILE +
                     (size_t)(((a0 - ti * TILE) / SWF_R) * (TILE / C) + (b0 - tj * TILE) / C) * (SWF_R * C) + (size_t)lr * C + lc;
#else
  const size_t off = (size_t)tile_id(ti, tj, g.P) * TILE * TILE + (size_t)(a0 - ti * TILE + lr) * TILE + (b0 - tj * TILE + lc);
#endif
  const char *pp = a.ws + off * sizeof(T);
  constexpr int SL = 32 + 2 * C + 1;
  // dynamic LDS: [P][NP][VW] float64 updates | [P][SL] float64 statistics | [FG][16][C + 1] T transposition buffers
  double *Us = reinterpret_cast<double *>(dsm);
  double (*stl)[SL] = reinterpret_cast<double (*)[SL]>(dsm + (size_t)P * NP * VW * 8);
  T (*tm)[SWF_R][C + 1] = reinterpret_cast<T (*)[SWF_R][C + 1]>(dsm + (size_t)P * NP * VW * 8 + (size_t)P * SL * 8);
  if (a.out_XTX) {
    for (int q = tid; q < P * SL; q += SWF4_T) {
      const int f = q / SL, i = q - f * SL;
      const double *fs = a.fstats + (size_t)f * fstat_len(K, M);
      double v;
      if (i < 16) v = (cX && a0 + i < K) ? fs[a0 + i] : 0.0;
      else if (i < 32) v = (sX && a0 + i - 16 < K) ? fs[K + a0 + i - 16] : 1.0;
      else if (i < 32 + C) v = (cX && b0 + i - 32 < K) ? fs[b0 + i - 32] : 0.0;
      else if (i < 32 + 2 * C) v = (sX && b0 + i - 32 - C < K) ? fs[K + b0 + i - 32 - C] : 1.0;
      else v = fs[2 * K + 2 * M];
      stl[f][i] = v;
    }
  }
  // ---- this group's folds: the update of this thread's piece, in split order (up to 8 partials in flight) -> LDS
  // (the NEXT fold of the group requested before the current one is added -- two folds' loads in flight -- was measured
  //  SLOWER, 22.7 -> 26.9 us: like the 256-thread kernel's same experiment in round 2, more loads in flight per thread do
  //  not help a kernel whose CUs' memory pipes are already full)
  for (int f = fg; f < P; f += SWF4_FG) {
    double u[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) u[e] = 0;
    const char *pf = pp + (size_t)f * a.splits * g.unit_bytes;
    for (int p0 = 0; p0 < nsp; p0 += 8) {
      const int cnt = nsp - p0 < 8 ? nsp - p0 : 8;
      vld_t q[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) q[j] = *reinterpret_cast<const vld_t *>(pf + (size_t)(p0 + (j < cnt ? j : 0)) * g.unit_bytes);
#pragma unroll
      for (int j = 0; j < 8; ++j)
        if (j < cnt) {
#pragma unroll
          for (int e = 0; e < VW; ++e) u[e] += (double)q[j][e];
        }
    }
#pragma unroll
    for (int e = 0; e < VW; ++e) Us[((size_t)f * NP + pid) * VW + e] = u[e];
  }
  lds_barrier();
  // ---- G = sum of the folds' updates in fold order (every group forms it for itself: the same chain)
  double gsum[VW];
#pragma unroll
  for (int e = 0; e < VW; ++e) gsum[e] = 0;
  for (int f = 0; f < P; ++f) {
#pragma unroll
    for (int e = 0; e < VW; ++e) gsum[e] += Us[((size_t)f * NP + pid) * VW + e];
  }
  bool ok[VW];
  const bool all_ok = gr < K && gc + VW <= K && gr <= gc;
#pragma unroll
  for (int e = 0; e < VW; ++e) ok[e] = gr < K && gc + e < K && gr <= gc + e;
  T gT[VW];
#pragma unroll
  for (int e = 0; e < VW; ++e) gT[e] = (T)gsum[e];
  // ---- outputs: o = 0 the full-data matrix, o >= 1 fold o - 1; group fg takes o = fg, fg + 4, ...: finished pieces are
  // stored as rows a / columns b and parked in the group's buffer; after one barrier the image is stored transposed
  const int mc = pid / (SWF_R / VW), mr = (pid - mc * (SWF_R / VW)) * VW;
  const int n_out = a.out_XTX ? P + 1 : 1;
  for (int o0 = 0; o0 < n_out; o0 += SWF4_FG) {
    const int o = o0 + fg;
    T *out = nullptr;
    if (o < n_out) {
      T vals[VW];
      if (o == 0) {
#pragma unroll
        for (int e = 0; e < VW; ++e) vals[e] = gT[e];
        out = Gout;
      } else {
        const double *st = stl[o - 1];
        const double swt = st[32 + 2 * C], mur = st[lr], sdr = st[16 + lr];
#pragma unroll
        for (int e = 0; e < VW; ++e) {
          double v = (double)gT[e] - Us[((size_t)(o - 1) * NP + pid) * VW + e];
          if (cX) v -= swt * (mur * st[32 + lc + e]);
          if (sX) v = v * (sdr * st[32 + C + lc + e]);
          vals[e] = (T)v;
        }
        out = (T *)a.out_XTX + (size_t)(a.seg0 + o - 1) * K * K;
      }
      T *dst = out + (size_t)gr * K + gc;
#ifdef CVM_SWF_NO_DIRECT_PROBE
      if (false) {
#else
      if (all_ok) {
#endif
        vld_t vv;
#pragma unroll
        for (int e = 0; e < VW; ++e) vv[e] = vals[e];
        if (o) out_store(reinterpret_cast<vld_t *>(dst), vv); else *reinterpret_cast<vld_t *>(dst) = vv;
      } else {
#ifndef CVM_SWF_NO_DIRECT_PROBE
#pragma unroll
        for (int e = 0; e < VW; ++e) if (ok[e]) dst[e] = vals[e];
#endif
      }
#pragma unroll
      for (int e = 0; e < VW; ++e) tm[fg][lr][lc + e] = vals[e];
    }
    lds_barrier();
    const int orow = b0 + mc, ocol = a0 + mr;
#ifdef CVM_SWF_NO_MIRROR_PROBE
    if (o < n_out && orow < K && tm[fg][0][0] == (T)12345.678) {      // (timing probe: never true)
#else
    if (o < n_out && orow < K) {
#endif
      T *md = out + (size_t)orow * K + ocol;
      if (ocol + VW <= K && ocol + VW - 1 < orow) {
        vld_t vv;
#pragma unroll
        for (int e = 0; e < VW; ++e) vv[e] = tm[fg][mr + e][mc];
        if (o) out_store(reinterpret_cast<vld_t *>(md), vv); else *reinterpret_cast<vld_t *>(md) = vv;
      } else {
#pragma unroll
        for (int e = 0; e < VW; ++e) if (ocol + e < K && ocol + e < orow) md[e] = tm[fg][mr + e][mc];
      }
    }
    lds_barrier();
  }
}
