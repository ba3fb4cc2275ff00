// cvmhip.hip -- MI355X (gfx950 / CDNA4) kernels + C ABI for the cvmatrix hot path.
//
// Two stages of the reference are replaced (see include/cvmhip.h for the mapping):
//   fit stage   cvmatrix/cvmatrix.py:1193-1243   G = X^T W X, H = X^T W Y, column sums
//   fold stage  cvmatrix/cvmatrix.py:589-1129    per fold: G - G_V, rank-1 centring,
//                                                outer-std scaling
// Both are the same contraction  A^T diag(w) [A | B]  reduced over ROWS, so one MFMA kernel
// (`wgram_kernel`) serves both: the fit stage runs it over all rows, the fold stage over
// the validation rows of every fold of a batch gathered by index.  Small deterministic
// finalize kernels then reduce the row-split partials in a fixed order and apply the
// reference's correction arithmetic in the reference's operation order.
//
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -shared -fPIC
// (-ffp-contract=off: the finalize arithmetic must round like NumPy's separate ufuncs,
//  and the VALU column sums must round like the MFMA's A operand (w*x rounded first).)

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <type_traits>

#include "../../include/cvmhip.h"

namespace {

// ----------------------------------------------------------------------------------
// geometry
// ----------------------------------------------------------------------------------
constexpr int TILE = 128;     // columns per panel; a workgroup owns a TILE x TILE output tile
constexpr int STAGE_ROWS = 16;  // rows staged in LDS per pipeline stage (4 MFMA k-steps)
constexpr int PITCH = 144;    // LDS row pitch of a panel, in elements (see bank note below)
constexpr int YT = 32;        // Y columns handled per diagonal work item (2 MFMA col tiles)
constexpr int YPITCH = 48;    // LDS row pitch of the Y tile, in elements
constexpr int NTHREADS = 512; // 8 waves, two per SIMD
constexpr int PANEL_ELEMS = STAGE_ROWS * PITCH;             // 2304
constexpr int BUF_ELEMS = 2 * PANEL_ELEMS + STAGE_ROWS;     // A panel, B panel | Y tile, w
constexpr int TARGET_WG_1 = 256;  // resident workgroups (both Gram kernels: one 8-wave workgroup per CU)
constexpr int TARGET_WG_2 = 256;
// LDS bank note.  MFMA 16x16x4 operand reads: lane l reads row k0+(l>>4), column c0+(l&15).
// f64 / ds_read_b64 (64 banks of 4 B): lanes 0-15 cover 128 B = 32 banks; lanes 16-31 read
// the next row, so the pitch must be = 128 B mod 256 B: 144*8 = 1152 = 4*256+128.  f32 /
// ds_read_b32 (32 banks): lanes 0-15 cover 64 B; pitch must be = 64 mod 128: 144*4 = 576.
// Same for the Y tile: 48*8 = 384 = 256+128, 48*4 = 192 = 128+64.

template <typename T> struct MF;
template <> struct MF<double> {
  typedef double acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  // v_mfma_f64_16x16x4_f64 C/D map: col = lane&15, row = (lane>>4) + 4*reg
  static __device__ __forceinline__ int drow(int lane, int r) { return (lane >> 4) + 4 * r; }
};
template <> struct MF<float> {
  typedef float acc_t __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  // v_mfma_f32_16x16x4_f32 C/D map: col = lane&15, row = 4*(lane>>4) + reg
  static __device__ __forceinline__ int drow(int lane, int r) { return 4 * (lane >> 4) + r; }
};

struct Geom {
  int K, M;
  int P;        // column panels = ceil(K/128)
  int Kp;       // P*128
  int Yc;       // Y chunks of 32 columns (>= 1 even when M == 0)
  int Mp;       // Yc*32
  int nTiles;   // P(P+1)/2 upper-triangular tiles
  int nT;       // work items per unit
  int diag_only;  // 1: only diagonal items (XTY / statistics only), no G tiles
  size_t tile_elems, h_elems;   // per unit, in elements of T
  size_t stat_len;              // per unit, float64 entries
  size_t unit_bytes;
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

Geom make_geom(int K, int M, int esize, int diag_only) {
  Geom g;
  g.K = K; g.M = M;
  g.P = (K + TILE - 1) / TILE;
  g.Kp = g.P * TILE;
  g.Yc = M > 0 ? (M + YT - 1) / YT : 1;
  g.Mp = g.Yc * YT;
  g.nTiles = g.P * (g.P + 1) / 2;
  g.diag_only = diag_only;
  g.nT = diag_only ? g.P * g.Yc : g.nTiles + g.P * (g.Yc - 1);
  g.tile_elems = diag_only ? 0 : (size_t)g.nTiles * TILE * TILE;
  g.h_elems = (size_t)g.P * TILE * g.Mp;
  g.stat_len = 2 * (size_t)g.Kp + 2 * (size_t)g.Mp + 4;
  g.unit_bytes = align_up(g.tile_elems * esize, 256) + align_up(g.h_elems * esize, 256) +
                 align_up(g.stat_len * 8, 256);
  return g;
}

template <typename T> struct WgramArgs {
  const T *X, *Y, *w;
  const int64_t *idx;   // nullptr: rows are offs[seg]..offs[seg+1] themselves
  const int64_t *offs;  // device; nullptr: one segment [0, N)
  int64_t N;
  int64_t seg0;         // first segment of this batch
  int n_seg, splits;
  Geom g;
  long n_items, items_per_xcd;
  char *ws;             // unit u at ws + u*unit_bytes
  // fused single-split fold update (wgram4_kernel<.., FUSED>): finish in the epilogue
  const double *fstats; // per fold of the batch: means / stds / sw_train (fold_stats_kernel)
  const void *G, *H;    // full-data matrices
  void *out_XTX, *out_XTY;
  unsigned flags;
  int dbg;              // diagnostic ablations (env CVM_DEBUG): 1 no global loads after the
                        // first stage, 2 no MFMA, 4 no VALU column sums; results are wrong
};

template <typename T> __device__ __forceinline__ T *unit_tiles(char *ws, const Geom &g, long u) {
  return (T *)(ws + (size_t)u * g.unit_bytes);
}
template <typename T> __device__ __forceinline__ T *unit_h(char *ws, const Geom &g, long u) {
  return (T *)(ws + (size_t)u * g.unit_bytes + ((g.tile_elems * sizeof(T) + 255) / 256 * 256));
}
template <typename T> __device__ __forceinline__ double *unit_stats(char *ws, const Geom &g, long u) {
  return (double *)(ws + (size_t)u * g.unit_bytes + ((g.tile_elems * sizeof(T) + 255) / 256 * 256) +
                    ((g.h_elems * sizeof(T) + 255) / 256 * 256));
}

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long uni64(long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffll));
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}
template <typename P> __device__ __forceinline__ P *unip(P *p) { return (P *)uni64((long long)p); }
// by-value copy of the launch arguments with every field forced into scalar registers; a
// body function that reads them through the caller's reference reloads them with flat
// loads (and a full vmcnt wait) at every use
template <typename T> __device__ __forceinline__ WgramArgs<T> scalarize(const WgramArgs<T> &r) {
  WgramArgs<T> a;
  a.X = unip(r.X); a.Y = unip(r.Y); a.w = unip(r.w); a.idx = unip(r.idx); a.offs = unip(r.offs);
  a.N = uni64(r.N); a.seg0 = uni64(r.seg0); a.n_seg = uni(r.n_seg); a.splits = uni(r.splits);
  a.g.K = uni(r.g.K); a.g.M = uni(r.g.M); a.g.P = uni(r.g.P); a.g.Kp = uni(r.g.Kp);
  a.g.Yc = uni(r.g.Yc); a.g.Mp = uni(r.g.Mp); a.g.nTiles = uni(r.g.nTiles); a.g.nT = uni(r.g.nT);
  a.g.diag_only = uni(r.g.diag_only);
  a.g.tile_elems = (size_t)uni64((long long)r.g.tile_elems);
  a.g.h_elems = (size_t)uni64((long long)r.g.h_elems);
  a.g.stat_len = (size_t)uni64((long long)r.g.stat_len);
  a.g.unit_bytes = (size_t)uni64((long long)r.g.unit_bytes);
  a.n_items = uni64(r.n_items); a.items_per_xcd = uni64(r.items_per_xcd);
  a.ws = unip(r.ws); a.dbg = uni(r.dbg);
  a.fstats = unip(r.fstats); a.G = unip(r.G); a.H = unip(r.H);
  a.out_XTX = unip(r.out_XTX); a.out_XTY = unip(r.out_XTY); a.flags = (unsigned)uni((int)r.flags);
  return a;
}

__device__ __forceinline__ void decode_tile(int t, int P, int &ti, int &tj) {
  // row-major upper triangle: (0,0),(0,1)..(0,P-1),(1,1)...
  int i = 0, rem = t;
  while (rem >= P - i) { rem -= P - i; ++i; }
  ti = i; tj = i + rem;
}
__host__ __device__ __forceinline__ int tile_id(int i, int j, int P) {
  return i * P - i * (i - 1) / 2 + (j - i);
}

// rows of segment `seg` handled by split `sp`
__device__ __forceinline__ void split_range(int64_t n, int splits, int sp, int64_t &r0, int64_t &r1) {
  int64_t per = (n + splits - 1) / splits;
  per = (per + STAGE_ROWS - 1) / STAGE_ROWS * STAGE_ROWS;
  r0 = (int64_t)sp * per; if (r0 > n) r0 = n;
  r1 = r0 + per; if (r1 > n) r1 = n;
}

// ----------------------------------------------------------------------------------
// wgram_kernel: partial  P[a][b] = sum_{rows r in split} w_r * X[r][a] * [X|Y][r][b]
// plus the weighted column sums of the same rows.
//
// Workgroup = 8 waves (2 per SIMD) = one 128x128 tile (i,j), i <= j, of one
// (segment, split) unit.  Wave (wr,wc), wr in 0..1, wc in 0..3, owns the 64x32 block
// rows 64wr.., cols 32wc.. of the tile: 4x2 MFMA 16x16 tiles, 8 accumulators.
//   diagonal tile: blocks (1,0),(1,1) lie strictly below the diagonal (mirror of data the
//     other waves produce) and are not computed.  The two freed waves ("H waves") compute
//     panel_i[:, 64h..64h+64)^T W Y[:, 32c..32c+32)  (again 4x2 MFMA tiles) and, on the
//     VALU, the column sums sX,qX of their 64 columns (+ sY,qY,sw,nz on panel 0).
// Every wave therefore runs the same loop: 4 A fragments x 2 B fragments per k-step.
// Rows reach LDS through registers (global_load_dwordx4 -> ds_write_b128), one 16-row stage
// ahead of the MFMAs (loads issued before the stage's MFMAs, LDS written after them, one
// barrier per stage).  Row numbers come from a 3-slot LDS ring filled three stages ahead,
// so no global load in the loop depends on another one.
// ----------------------------------------------------------------------------------
__device__ double g_zero_line[128];  // zero-initialised at code-object load
__device__ double g_one_line[2] = {1.0, 1.0};
#ifdef CVM_STAMPS
// diagnostic build only: per (workgroup, wave) cycle sums of the three phases of a stage
__device__ unsigned long long g_stamps[1024 * 8 * 4];
__device__ unsigned long long g_stamps2[1024 * 8 * 4];
__device__ unsigned long long g_stamps3[1024 * 8 * 2];   // per compute wave: prologue, epilogue cycles   // per wave: shader cycles, 100 MHz ticks, start tick
#define STAMP(v) do { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); } while (0)
#endif

template <typename T, bool WEIGHTED, bool GATHER, bool ALIGNED>
__global__ __launch_bounds__(NTHREADS, 2) void wgram_kernel(const WgramArgs<T> a) {
  typedef typename MF<T>::acc_t acc_t;
  constexpr int VEC = 16 / sizeof(T);               // elements per 16-byte chunk
  constexpr int CPR = TILE / VEC;                   // chunks per panel row
  constexpr int NCH = STAGE_ROWS * CPR / NTHREADS;  // chunks per thread per panel (2 / 1)
  typedef T vec_t __attribute__((ext_vector_type(VEC)));

  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T *smem = reinterpret_cast<T *>(smem_raw);
  int64_t *ring = reinterpret_cast<int64_t *>(smem_raw + 2 * BUF_ELEMS * sizeof(T));

  const Geom &g = a.g;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  // ---- which work item: contiguous ranges of the (unit, tile) list per XCD ----------
  const long b = blockIdx.x;
  const long item = (b & 7) * a.items_per_xcd + (b >> 3);
  if ((b >> 3) >= a.items_per_xcd || item >= a.n_items) return;
  const long u = item / g.nT;
  const int it = (int)(item - u * g.nT);
  const int seg = (int)(u / a.splits);
  const int sp = (int)(u - (long)seg * a.splits);
  int ti, tj, yc;
  if (g.diag_only) { ti = tj = it / g.Yc; yc = it - ti * g.Yc; }
  else if (it < g.nTiles) { decode_tile(it, g.P, ti, tj); yc = 0; }
  else { int e = it - g.nTiles; ti = tj = e / (g.Yc - 1); yc = 1 + e - ti * (g.Yc - 1); }
  const bool diag = (ti == tj);
  const int wr = wave >> 2, wc = wave & 3;
  const bool h_wave = diag && wr == 1 && wc < 2;
  const bool do_g = !g.diag_only && yc == 0;       // the G tile of this item is wanted
  const bool mfma_wave = h_wave ? (g.M > 0 || yc == 0) : do_g;   // H waves also feed the X column sums

  int64_t seg_begin, seg_rows;
  if (a.offs) { seg_begin = a.offs[a.seg0 + seg]; seg_rows = a.offs[a.seg0 + seg + 1] - seg_begin; }
  else { seg_begin = 0; seg_rows = a.N; }
  int64_t r0, r1;
  split_range(seg_rows, a.splits, sp, r0, r1);
  const int nstages = (int)((r1 - r0 + STAGE_ROWS - 1) / STAGE_ROWS);

  // ---- per-thread staging coordinates -----------------------------------------------
  const int st_row0 = tid / CPR, st_col = (tid % CPR) * VEC;   // chunk j: row st_row0 + j*(512/CPR)
  constexpr int ST_ROW_STEP = NTHREADS / CPR;
  const int colA0 = ti * TILE, colB0 = tj * TILE;
  const int y_row = tid >> 5, y_m = tid & 31;       // Y tile: one element per thread
  const int y_col = yc * YT + y_m;

  vec_t ra[NCH], rb[NCH];
  T ry = 0, rw = 0;

  // threads 0..15: row number of stage s, row tid (or -1 past the end).  The global load
  // is issued early (ring_load) and parked in LDS after the stage's MFMAs (ring_store).
  auto ring_load = [&](int s) -> int64_t {
    int64_t row = -1;
    if (tid < STAGE_ROWS) {
      int64_t r = r0 + (int64_t)s * STAGE_ROWS + tid;
      if (r < r1) row = GATHER ? a.idx[seg_begin + r] : seg_begin + r;
    }
    return row;
  };
  auto ring_store = [&](int s, int64_t row) {
    if (tid < STAGE_ROWS) ring[(s % 3) * STAGE_ROWS + tid] = row;
  };
  auto load_panel = [&](vec_t *dst, int col0, int s, int region, int only_j) {
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
      if (only_j >= 0 && j != only_j) continue;
      const int64_t row = ring[(s % 3) * STAGE_ROWS + st_row0 + j * ST_ROW_STEP];
      const int col = col0 + st_col;
      vec_t v;
#pragma unroll
      for (int e = 0; e < VEC; ++e) v[e] = 0;
      if (row >= 0) {
        const T *src = a.X + row * (int64_t)g.K + col;
        if (ALIGNED) {
          if (col < g.K) v = *reinterpret_cast<const vec_t *>(src);
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) if (col + e < g.K) v[e] = src[e];
        }
      }
      dst[j] = v;
    }
  };
  auto issue_loads = [&](int s, bool panels) {
    if (panels) {
      load_panel(ra, colA0, s, 0, -1);
      if (!diag) load_panel(rb, colB0, s, 1, -1);
    }
    if (diag) {
      const int64_t row = ring[(s % 3) * STAGE_ROWS + y_row];
      T v = 0;
      if (row >= 0 && y_col < g.M) v = a.Y[row * (int64_t)g.M + y_col];
      ry = v;
    }
    if (tid < STAGE_ROWS) {
      const int64_t row = ring[(s % 3) * STAGE_ROWS + tid];
      T v = 0;
      if (row >= 0) v = WEIGHTED ? a.w[row] : (T)1;
      rw = v;
    }
  };
  auto write_lds = [&](int buf) {
    T *base = smem + buf * BUF_ELEMS;
#pragma unroll
    for (int j = 0; j < NCH; ++j)
      *reinterpret_cast<vec_t *>(base + (st_row0 + j * ST_ROW_STEP) * PITCH + st_col) = ra[j];
    if (!diag) {
#pragma unroll
      for (int j = 0; j < NCH; ++j)
        *reinterpret_cast<vec_t *>(base + PANEL_ELEMS + (st_row0 + j * ST_ROW_STEP) * PITCH + st_col) = rb[j];
    } else {
      base[PANEL_ELEMS + y_row * YPITCH + y_m] = ry;
    }
    if (tid < STAGE_ROWS) base[2 * PANEL_ELEMS + tid] = rw;
  };

  // ---- accumulators -------------------------------------------------------------------
  acc_t acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = (acc_t){0, 0, 0, 0};
  // column-sum accumulators; meaning depends on the wave's role (see the k-step loop)
  double st_s[4] = {0, 0, 0, 0}, st_q[4] = {0, 0, 0, 0};
  const int stat_role = (h_wave && yc == 0) ? 1 : ((diag && ti == 0 && wave == 0) ? 2 : 0);

  // wave -> operand blocks inside the LDS stage buffer
  const int lk = lane >> 4, lc = lane & 15;
  int a_off, b_off, b_pitch;     // element offsets of this lane's A / B fragment, row 0
  int a_col, b_col;              // block origin inside the tile (for the store)
  if (h_wave) { a_col = 64 * wc; b_col = 0; a_off = a_col + lc; b_off = PANEL_ELEMS + lc; b_pitch = YPITCH; }
  else {
    a_col = 64 * wr; b_col = 32 * wc;
    a_off = a_col + lc; b_off = (diag ? 0 : PANEL_ELEMS) + b_col + lc; b_pitch = PITCH;
  }

  // ---- prologue -------------------------------------------------------------------------
  if (nstages > 0) {
    ring_store(0, ring_load(0));
    ring_store(1, ring_load(1));
    ring_store(2, ring_load(2));
    __syncthreads();
    issue_loads(0, true);
    write_lds(0);
    __syncthreads();
  }
  // the second-dispatched half of the workgroup loses issue arbitration to the older
  // half on every k-step (MI355X_MICROARCH "Two waves per SIMD" item 4): static priority
  if (wave >= 4) __builtin_amdgcn_s_setprio(1);

#ifdef CVM_STAMPS
  unsigned long long t_a = 0, t_b = 0, t_c = 0, t0, t1, t2, t3;
#endif

  // One pipeline stage, specialised at compile time on the wave's role so that the body is
  // straight-line code (branches inside it make hipcc drain lgkmcnt at every block edge):
  //   MFM   the wave issues MFMAs;  ROLE 0 none / 1 X column sums / 2 Y column sums
  //   LD    a next stage exists (loads for it are issued here)
  auto stage = [&](auto MFMc, auto ROLEc, auto LDc, int s) {
    constexpr bool MFM = decltype(MFMc)::value;
    constexpr int ROLE = decltype(ROLEc)::value;
    constexpr bool LD = decltype(LDc)::value;
#ifdef CVM_STAMPS
    STAMP(t0);
#endif
    if (LD) issue_loads(s + 1, true);
    const int64_t ring_next = ring_load(s + 3);
    const T *buf = smem + (s & 1) * BUF_ELEMS;
    const T *wb = buf + 2 * PANEL_ELEMS;
#ifdef CVM_STAMPS
    STAMP(t1);
#endif
    if (MFM || ROLE != 0) {
      // fragments of k-step ks+1 are read from LDS before the MFMAs of k-step ks issue
      T af[2][4], bf[2][2], yf[2][2], wv[2];
      auto read_frags = [&](int ks, int slot) {
        const int r = 4 * ks + lk;
        if (MFM || ROLE == 1) {
#pragma unroll
          for (int m = 0; m < 4; ++m) af[slot][m] = buf[a_off + r * PITCH + 16 * m];
        }
        if (MFM) {
#pragma unroll
          for (int n = 0; n < 2; ++n) bf[slot][n] = buf[b_off + r * b_pitch + 16 * n];
        }
        if (ROLE == 2) {
#pragma unroll
          for (int n = 0; n < 2; ++n) yf[slot][n] = buf[PANEL_ELEMS + r * YPITCH + 16 * n + lc];
        }
        wv[slot] = wb[r];   // 0 on rows past the end of the split, 1 if unweighted
      };
      read_frags(0, 0);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int c = ks & 1;
        if (ks < 3) read_frags(ks + 1, c ^ 1);
        // Column sums on the VALU, in the shadow of this k-step's MFMAs, from fragments
        // in registers: lane (lk,lc) owns rows = lk (mod 4) of column lc of each 16-column
        // group.  p = w*x is rounded like the MFMA A operand; s += p, q += p*x.  sw, sX,
        // sY use the same row classes and the same final combine, so a column of ones
        // gets s == q == sw bit for bit (variance exactly 0).
        if (ROLE == 1) {          // H wave: its 64 X columns
#pragma unroll
          for (int m = 0; m < 4; ++m) {
            if (sizeof(T) == 8) {
              const T pv = WEIGHTED ? (T)(af[c][m] * wv[c]) : af[c][m];
              st_s[m] += (double)pv; st_q[m] += (double)(pv * af[c][m]);
            } else {
              const double pv = (double)wv[c] * (double)af[c][m];
              st_s[m] += pv; st_q[m] += pv * (double)af[c][m];
            }
          }
        } else if (ROLE == 2) {   // wave 0 of panel 0: Y columns, sw, nz
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            const T yv = yf[c][n];
            if (sizeof(T) == 8) {
              const T pv = WEIGHTED ? (T)(yv * wv[c]) : yv;
              st_s[n] += (double)pv; st_q[n] += (double)(pv * yv);
            } else {
              const double pv = (double)wv[c] * (double)yv;
              st_s[n] += pv; st_q[n] += pv * (double)yv;
            }
          }
          st_s[2] += (double)wv[c];                         // sw
          st_s[3] += (wv[c] != (T)0) ? 1.0 : 0.0;           // nz
          st_q[3] += (wv[c] < (T)0) ? 1.0 : 0.0;            // any negative weight
        }
        if (MFM) {
          if (WEIGHTED) {
#pragma unroll
            for (int m = 0; m < 4; ++m) af[c][m] *= wv[c];
          }
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
              acc[m * 2 + n] = MF<T>::mfma(af[c][m], bf[c][n], acc[m * 2 + n]);
        }
      }
    }
#ifdef CVM_STAMPS
    STAMP(t2);
#endif
    if (LD) write_lds((s + 1) & 1);
    ring_store(s + 3, ring_next);   // slot (s%3) was last read for stage s, one barrier ago
    __syncthreads();
#ifdef CVM_STAMPS
    STAMP(t3);
    t_a += t1 - t0; t_b += t2 - t1; t_c += t3 - t2;
#endif
  };
  auto run = [&](auto MFMc, auto ROLEc) {
#pragma unroll 1
    for (int s = 0; s + 1 < nstages; ++s) stage(MFMc, ROLEc, std::true_type{}, s);
    if (nstages > 0) stage(MFMc, ROLEc, std::false_type{}, nstages - 1);
  };
  typedef std::integral_constant<int, 0> R0;
  typedef std::integral_constant<int, 1> R1;
  typedef std::integral_constant<int, 2> R2;
  if (!diag) run(std::true_type{}, R0{});
  else if (h_wave) { if (yc == 0) run(std::true_type{}, R1{}); else run(std::true_type{}, R0{}); }
  else if (stat_role == 2) { if (do_g) run(std::true_type{}, R2{}); else run(std::false_type{}, R2{}); }
  else { if (do_g) run(std::true_type{}, R0{}); else run(std::false_type{}, R0{}); }
#ifdef CVM_STAMPS
  if (lane == 0 && blockIdx.x < 1024) {
    unsigned long long *o = g_stamps + ((size_t)blockIdx.x * 8 + wave) * 4;
    o[0] = t_a; o[1] = t_b; o[2] = t_c; o[3] = (unsigned long long)nstages;
  }
#endif

  // ---- store partials -------------------------------------------------------------------
  // combine the four row classes (lanes lc, lc+16, lc+32, lc+48) in class order
  auto comb = [&](double v) -> double {
    const double v1 = __shfl(v, lc + 16), v2 = __shfl(v, lc + 32), v3 = __shfl(v, lc + 48);
    return ((v + v1) + v2) + v3;   // meaningful in lanes 0..15
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
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            hp[(size_t)(a_col + 16 * m + MF<T>::drow(lane, r)) * g.Mp + 16 * n + lc] = acc[m * 2 + n][r];
    }
  } else if (do_g) {
    T *tp = unit_tiles<T>(a.ws, g, u) + (size_t)it * TILE * TILE;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          tp[(a_col + 16 * m + MF<T>::drow(lane, r)) * TILE + b_col + 16 * n + lc] = acc[m * 2 + n][r];
  }
}

// ----------------------------------------------------------------------------------
// wgram4_kernel: the fast path (float64, 16-byte aligned rows, even M).
//
// Same work decomposition, LDS stage image and partial layout as wgram_kernel, but the
// eight waves of a workgroup (one workgroup per CU) are specialised:
//   waves 0-3  COMPUTE, one per SIMD.  Wave (wr,wc) owns the 64x64 block (wr,wc) of the
//              128x128 tile: 4x4 MFMA tiles, 16 accumulators (128 VGPRs), 16 MFMAs per 8
//              LDS fragment reads.  They never touch global memory inside the loop, so no
//              vector-memory instruction ever blocks their issue (a 1 KiB load costs its
//              wave 200-450 cycles of issue on a busy CU: tools/dma_issue.hip).
//              On a diagonal tile the strictly-lower block (1,0) is not computed; its wave
//              (the "H wave") computes panel_i^T W Y[:, 32c..32c+32) (8x2 MFMA tiles).
//              Waves 0 and 3 also sum the X columns of their A fragments, wave 1 (panel 0)
//              the Y columns, sw and nz -- on the VALU, in the shadow of their MFMAs.
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
constexpr int NBUF4 = 4;        // LDS stage buffers
constexpr size_t LDS4_BYTES = (size_t)NBUF4 * BUF_ELEMS * 8;

// The body is instantiated once per wave role and kept out of line: inlined together, the
// register allocator has to give all roles one common assignment of the 128 accumulator
// registers and spills hundreds of values; as separate functions every role fits.
//   ROLER 0/1/2: compute wave without sums / with X column sums / with Y column sums
//   ROLER 3: loader wave
__host__ __device__ inline size_t fstat_len(int K, int M);
__device__ __forceinline__ void fused_finish_block(double (*Ts)[65], const double *rs, bool diagb, int a0,
                                                   int b0, int K, const double *Gt, double *out,
                                                   double swt, bool cX, bool sX, int lane);
constexpr int WAVE_LDS_DOUBLES = 64 * 65 + 256;   // a wave's 64x64 block + row/column means and stds

template <bool WEIGHTED, bool GATHER, bool HWR, bool MFMR, int ROLER, bool FUSEDR = false>
__device__ __noinline__ void wgram4_body(const WgramArgs<double> &a_ref) {
  typedef double T;
  typedef MF<double>::acc_t acc_t;
#ifdef CVM_STAMPS
  const unsigned long long c_entry = __builtin_amdgcn_s_memtime();
#endif
  const WgramArgs<double> a = scalarize(a_ref);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  T *smem = reinterpret_cast<T *>(smem_raw);
  const Geom &g = a.g;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wave = wave_all & 3;

  const long b = blockIdx.x;
  const long item = (b & 7) * a.items_per_xcd + (b >> 3);
  if ((b >> 3) >= a.items_per_xcd || item >= a.n_items) return;
  const long u = item / g.nT;
  const int it = (int)(item - u * g.nT);
  const int seg = (int)(u / a.splits);
  const int sp = (int)(u - (long)seg * a.splits);
  int ti, tj, yc;
  if (g.diag_only) { ti = tj = it / g.Yc; yc = it - ti * g.Yc; }
  else if (it < g.nTiles) { decode_tile(it, g.P, ti, tj); yc = 0; }
  else { int e = it - g.nTiles; ti = tj = e / (g.Yc - 1); yc = 1 + e - ti * (g.Yc - 1); }
  const bool diag = (ti == tj);
  const int wr = wave >> 1, wc = wave & 1;
  const bool h_wave = diag && wave == 2;
  const bool do_g = !g.diag_only && yc == 0;

  int64_t seg_begin, seg_rows;
  if (a.offs) { seg_begin = a.offs[a.seg0 + seg]; seg_rows = a.offs[a.seg0 + seg + 1] - seg_begin; }
  else { seg_begin = 0; seg_rows = a.N; }
  int64_t r0, r1;
  split_range(seg_rows, a.splits, sp, r0, r1);
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
    const char *one_src = reinterpret_cast<const char *>(unip(g_one_line));
    int oa = 2 * lane, ob = 2 * lane, oy = 2 * (lane & 15);
    if (colA0 + oa > g.K - 2) oa = g.K - 2 - colA0;
    if (colB0 + ob > g.K - 2) ob = g.K - 2 - colB0;
    if (oa < 0) oa = 0;
    if (ob < 0) ob = 0;
    const int ycol0 = yc * YT;
    if (g.M > 0) { if (ycol0 + oy > g.M - 2) oy = g.M - 2 - ycol0; if (oy < 0) oy = 0; } else oy = 0;
    const unsigned va = 8u * (unsigned)oa, vb = 8u * (unsigned)ob, vy = 8u * (unsigned)oy, vw = 4u * (unsigned)lane;
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
      const unsigned bufb = lds0 + (unsigned)((t % NBUF4) * BUF_ELEMS) * 8u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int lrow = d + 4 * j;
        const bool valid = ok[j];
        const char *xrow = reinterpret_cast<const char *>(a.X + rn[j] * (int64_t)g.K);
        dma16_all(valid ? xrow + 8 * (int64_t)colA0 : zero_src, va, bufb + (unsigned)(lrow * PITCH) * 8u);
        if (!diag) {
          dma16_all(valid ? xrow + 8 * (int64_t)colB0 : zero_src, vb,
                    bufb + (unsigned)(PANEL_ELEMS + lrow * PITCH) * 8u);
        } else {
          const char *yrow = (valid && g.M > 0)
              ? reinterpret_cast<const char *>(a.Y + rn[j] * (int64_t)g.M + ycol0) : zero_src;
          dma16_lo16(yrow, vy, bufb + (unsigned)(PANEL_ELEMS + lrow * YPITCH) * 8u);
        }
        const char *wsrc = valid ? (WEIGHTED ? reinterpret_cast<const char *>(a.w + rn[j]) : one_src) : zero_src;
        dma4_lo2(wsrc, vw, bufb + (unsigned)(2 * PANEL_ELEMS + lrow) * 8u);
      }
    };
    // 12 LDS-DMA instructions per stage; ONE stage may stay in flight across a barrier, so
    // that at barrier B_s stage s+2 is in LDS: the compute waves may then read the first
    // fragments of stage s+1 before they reach B_s
    auto wait_one_stage_in_flight = [&]() { asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); };
    int64_t rn[4];
    bool ok[4];
    __builtin_amdgcn_s_setprio(3);
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
    if (FUSEDR) __builtin_amdgcn_s_barrier();       // the compute waves reuse the ring in their epilogue
#ifdef CVM_STAMPS
    if (lane == 0 && blockIdx.x < 1024) {
      unsigned long long *o = g_stamps + ((size_t)blockIdx.x * 8 + wave_all) * 4;
      o[0] = t_a; o[1] = t_b; o[2] = t_c; o[3] = (unsigned long long)nstages;
    }
#endif
    return;
  }

  // ---- compute waves ----------------------------------------------------------------------
  const int stat_role = ROLER;
  acc_t acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = (acc_t){0, 0, 0, 0};
  double st_s[4] = {0, 0, 0, 0}, st_q[4] = {0, 0, 0, 0};

  const int lk = lane >> 4, lc = lane & 15;
  const int a_col = h_wave ? 0 : 64 * wr;
  const int b_col = h_wave ? 0 : 64 * wc;
  const int a_off = a_col + lc;
  const int b_off = h_wave ? PANEL_ELEMS + lc : (diag ? 0 : PANEL_ELEMS) + b_col + lc;

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
  // wave waits for the others at the stage barrier; but the kernel is power-limited and the
  // clock rises, about 1 % measured.  Dropping the padded second column tile of the H wave the
  // same way made the gathered variant 1.7 % slower -- code placement -- and was not kept.)
  constexpr bool TRI = (ROLE == 1) && !HW;
  constexpr int NA = HW ? 8 : 4, NB = HW ? 2 : 4;
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
#pragma unroll 1
  for (int s = 0; s < nstages; ++s) {
#ifdef CVM_STAMPS
    STAMP(t0);
    STAMP(t1);
    if (a.dbg & 2) { __syncthreads(); continue; }   // diagnostic: loaders alone
#endif
    const T *buf = smem + (s % NBUF4) * BUF_ELEMS;
    const T *nbuf = smem + ((s + 1) % NBUF4) * BUF_ELEMS;
    if (MFM || ROLE != 0) {
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
#ifdef CVM_STAMPS
  STAMP(c_loop1);
  if (lane == 0 && blockIdx.x < 1024) {
    unsigned long long *o = g_stamps + ((size_t)blockIdx.x * 8 + wave) * 4;
    o[0] = t_a; o[1] = t_b; o[2] = t_c; o[3] = (unsigned long long)nstages;
  }
#endif

  if (FUSEDR) {
    // ---- fused single-split epilogue: no partials, no apply kernel --------------------------
    // The fold's statistics are already in a.fstats (colstats_kernel + fold_stats_kernel ran
    // first); every wave finishes its own block: total - update, rank-1 centring, outer-std
    // scaling (cvmatrix.py:1001-1010), mirrored store through the wave's slice of the ring.
    __syncthreads();   // all loaders have drained their LDS-DMA
    const int K = g.K, M = g.M;
    const double *fs = a.fstats + (size_t)seg * fstat_len(K, M);
    const double swt = fs[2 * K + 2 * M];
    const bool cX = a.flags & CVM_CENTER_X, cY = a.flags & CVM_CENTER_Y;
    const bool sX = a.flags & CVM_SCALE_X, sY = a.flags & CVM_SCALE_Y;
    const size_t fo = (size_t)(a.seg0 + seg);
    if (h_wave) {
      if (a.out_XTY && M > 0) {
        double *out = (double *)a.out_XTY + fo * (size_t)K * M;
        const double *Ht = (const double *)a.H;
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int row = ti * TILE + 16 * m + MF<T>::drow(lane, r), col = yc * YT + 16 * n + lc;
              if (row < K && col < M) {
                double v = Ht[(size_t)row * M + col] - acc[m * 2 + n][r];
                if (cX || cY) v -= swt * (fs[row] * fs[2 * K + col]);
                if (sX && sY) v = v / (fs[K + row] * fs[2 * K + M + col]);
                else if (sX) v = v / fs[K + row];
                else if (sY) v = v / fs[2 * K + M + col];
                out[(size_t)row * M + col] = v;
              }
            }
      }
    } else if (do_g && MFM && a.out_XTX) {
      const int a0 = ti * TILE + 64 * wr, b0 = tj * TILE + 64 * wc;
      if (a0 < K && b0 < K) {
        double *slice = smem + (size_t)wave * WAVE_LDS_DOUBLES;
        double (*Ts)[65] = reinterpret_cast<double (*)[65]>(slice);
        double *rs = slice + 64 * 65;
        rs[lane] = (cX && a0 + lane < K) ? fs[a0 + lane] : 0.0;
        rs[64 + lane] = (sX && a0 + lane < K) ? fs[K + a0 + lane] : 1.0;
        rs[128 + lane] = (cX && b0 + lane < K) ? fs[b0 + lane] : 0.0;
        rs[192 + lane] = (sX && b0 + lane < K) ? fs[K + b0 + lane] : 1.0;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              Ts[16 * m + MF<T>::drow(lane, r)][16 * n + lc] = acc[m * 4 + n][r];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        fused_finish_block(Ts, rs, diag && wr == wc, a0, b0, K, (const double *)a.G,
                           (double *)a.out_XTX + fo * (size_t)K * K, swt, cX, sX, lane);
      }
    }
    return;
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
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          tp[(a_col + 16 * m + MF<T>::drow(lane, r)) * TILE + b_col + 16 * n + lc] = acc[m * 4 + n][r];
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
}

template <bool WEIGHTED, bool GATHER, bool FUSED = false>
__global__ __launch_bounds__(NT4, 2) void wgram4_kernel(const WgramArgs<double> a) {
  // role of this wave (same decode as in the body)
  const Geom &g = a.g;
  const int wave_all = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wave = wave_all & 3;
  const long b = blockIdx.x;
  const long item = (b & 7) * a.items_per_xcd + (b >> 3);
  if ((b >> 3) >= a.items_per_xcd || item >= a.n_items) return;
#ifdef CVM_STAMPS
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime();
  auto fin = [&]() {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0 && blockIdx.x < 1024) {
      unsigned long long *o = g_stamps2 + ((size_t)blockIdx.x * 8 + wave_all) * 4;
      o[0] = c1 - c0; o[1] = q1 - q0; o[2] = q0; o[3] = q1;
    }
  };
  if (wave_all >= 4) { wgram4_body<WEIGHTED, GATHER, false, false, 3, FUSED>(a); fin(); return; }
#else
  if (wave_all >= 4) { wgram4_body<WEIGHTED, GATHER, false, false, 3, FUSED>(a); return; }
#endif
  const int it = (int)(item % g.nT);
  int ti, tj, yc;
  if (g.diag_only) { ti = tj = it / g.Yc; yc = it - ti * g.Yc; }
  else if (it < g.nTiles) { decode_tile(it, g.P, ti, tj); yc = 0; }
  else { int e = it - g.nTiles; ti = tj = e / (g.Yc - 1); yc = 1 + e - ti * (g.Yc - 1); }
  const bool diag = (ti == tj);
  const bool do_g = !g.diag_only && yc == 0;
  if (FUSED) {   // statistics come from colstats_kernel: no summing roles
    if (diag && wave == 2) wgram4_body<WEIGHTED, GATHER, true, true, 0, true>(a);
    else if (do_g) wgram4_body<WEIGHTED, GATHER, false, true, 0, true>(a);
    else wgram4_body<WEIGHTED, GATHER, false, false, 0, true>(a);
#ifdef CVM_STAMPS
    fin();
#endif
    return;
  }
  const int role = !diag ? 0 : ((yc == 0 && (wave == 0 || wave == 3)) ? 1 : ((ti == 0 && wave == 1) ? 2 : 0));
  if (diag && wave == 2) wgram4_body<WEIGHTED, GATHER, true, true, 0>(a);
  else if (role == 1) { if (do_g) wgram4_body<WEIGHTED, GATHER, false, true, 1>(a); else wgram4_body<WEIGHTED, GATHER, false, false, 1>(a); }
  else if (role == 2) { if (do_g) wgram4_body<WEIGHTED, GATHER, false, true, 2>(a); else wgram4_body<WEIGHTED, GATHER, false, false, 2>(a); }
  else { if (do_g) wgram4_body<WEIGHTED, GATHER, false, true, 0>(a); else wgram4_body<WEIGHTED, GATHER, false, false, 0>(a); }
#ifdef CVM_STAMPS
  fin();
#endif
}

// ----------------------------------------------------------------------------------
// finalize kernels
// ----------------------------------------------------------------------------------
struct FinArgs {
  Geom g;
  int splits;
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
};
__host__ __device__ inline size_t fstat_len(int K, int M) { return 2 * (size_t)K + 2 * (size_t)M + 4; }

// column chunks (grid.y) of fold_stats_kernel: enough workgroups to fill the chip when there are
// few folds and many columns, one when there are many folds
inline int fold_stats_chunks(int K, int M, int64_t n_folds) {
  int c = (K + M + 255) / 256;
  const int64_t cap = n_folds >= 512 ? 1 : (512 + n_folds - 1) / n_folds;
  if (c > cap) c = (int)cap;
  return c < 1 ? 1 : c;
}

// fit: gstats = ordered sum of the split partials
template <typename T> __global__ void fit_stats_kernel(const FinArgs a, double *gstats) {
  const Geom &g = a.g;
  const int total = 2 * g.K + 2 * g.M + 3;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < total; c += gridDim.x * blockDim.x) {
    int src;
    if (c < g.K) src = c;
    else if (c < 2 * g.K) src = g.Kp + (c - g.K);
    else if (c < 2 * g.K + g.M) src = 2 * g.Kp + (c - 2 * g.K);
    else if (c < 2 * g.K + 2 * g.M) src = 2 * g.Kp + g.Mp + (c - 2 * g.K - g.M);
    else src = 2 * g.Kp + 2 * g.Mp + (c - 2 * g.K - 2 * g.M);
    double s = 0;
#pragma unroll 4
    for (int p = 0; p < a.splits; ++p) s += unit_stats<T>((char *)a.ws, g, p)[src];
    if (c < total - 1) gstats[c] = s;
    else if (a.neg_flag) *a.neg_flag = (s > 0) ? 1 : 0;
  }
}

// fold: training-set mean / std of every column; reference operation order
// (cvmatrix.py:612-620, 709-745, 1043, 1079, 1119-1128)
template <typename T> __global__ void fold_stats_kernel(const FinArgs a) {
  const Geom &g = a.g;
  const int f = blockIdx.x;
  const int K = g.K, M = g.M;
  const bool weighted = a.w != nullptr;
  const long u0 = (long)f * a.splits;
  double swv = 0, nzv = 0;
  if (weighted) {
    for (int p = 0; p < a.splits; ++p) {
      const double *st = unit_stats<T>((char *)a.ws, g, u0 + p);
      swv += st[2 * g.Kp + 2 * g.Mp + 0];
      nzv += st[2 * g.Kp + 2 * g.Mp + 1];
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
    double sv = 0, qv = 0;
#pragma unroll 4
    for (int p = 0; p < a.splits; ++p) {
      const double *st = unit_stats<T>((char *)a.ws, g, u0 + p);
      sv += st[s_src]; qv += st[q_src];
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
    fs[isX ? K + cc : 2 * K + M + cc] = sd;
    T *omu = (T *)(isX ? a.out_muX : a.out_muY), *osd = (T *)(isX ? a.out_sdX : a.out_sdY);
    const size_t o = (size_t)(a.seg0 + f) * (isX ? K : M) + cc;
    if (omu) omu[o] = (T)mu;
    if (osd && (isX ? want_sdX : want_sdY)) osd[o] = (T)sd;
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
template <typename T, bool FOLD>
__device__ __forceinline__ void finish_store_tile(double (*Ts)[ST + 1], bool diag, int a0, int b0, int K,
                                                  const T *Gt, T *out, const double *fs, double swt,
                                                  bool cX, bool sX, int tid, int nthreads) {
  constexpr int VW = 16 / sizeof(T);            // elements per 16-byte access
  constexpr int LPR = ST / VW;                  // lanes per row segment
  typedef T vst_t __attribute__((ext_vector_type(VW)));
  const bool vec_ok = ((size_t)K * sizeof(T)) % 16 == 0 && ((uintptr_t)out % 16 == 0) &&
                      (!FOLD || (uintptr_t)Gt % 16 == 0);
  for (int pass = 0; pass < (diag ? 1 : 2); ++pass) {
    const int r0g = pass ? b0 : a0, c0g = pass ? a0 : b0;
    for (int q = tid; q < ST * LPR; q += nthreads) {
      const int lr = q / LPR, lc = (q - lr * LPR) * VW;
      const int gr = r0g + lr, gc = c0g + lc;
      if (gr >= K || gc >= K) continue;
      const bool full = vec_ok && gc + VW <= K;
      T vals[VW];
      if (pass == 0) {
        T gvv[VW];
        if (FOLD) {
          if (full) {
            const vst_t t = *reinterpret_cast<const vst_t *>(Gt + (size_t)gr * K + gc);
#pragma unroll
            for (int e = 0; e < VW; ++e) gvv[e] = t[e];
          } else {
#pragma unroll
            for (int e = 0; e < VW; ++e) gvv[e] = (gc + e < K) ? Gt[(size_t)gr * K + gc + e] : (T)0;
          }
        }
        const double mur = (FOLD && cX) ? fs[gr] : 0.0, sdr = (FOLD && sX) ? fs[K + gr] : 1.0;
#pragma unroll
        for (int e = 0; e < VW; ++e) {
          const int cc = lc + e, gce = gc + e;
          double v = 0;
          if (gce < K) {
            const double upd = (diag && lr > cc) ? Ts[cc][lr] : Ts[lr][cc];
            if (FOLD) {
              v = (double)gvv[e] - upd;
              if (cX) v -= swt * (mur * fs[gce]);
              if (sX) v = v / (sdr * fs[K + gce]);
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
        *reinterpret_cast<vst_t *>(dst) = vv;
      } else {
#pragma unroll
        for (int e = 0; e < VW; ++e) if (gc + e < K) dst[e] = vals[e];
      }
      if (pass == 0 && !diag) {
        // park the finished values in place (off the diagonal every raw element is read by
        // this thread only) for the mirrored pass
#pragma unroll
        for (int e = 0; e < VW; ++e) Ts[lr][lc + e] = (double)vals[e];
      }
    }
    __syncthreads();
  }
}

// The same finishing step for ONE WAVE inside wgram4_kernel<.., FUSED> (float64, K even): the raw
// update of a 64x64 block is in Ts, the row/column means and stds in rs[0..255].  A wave has no
// other wave to hide its latency behind, so the G loads go out eight rows at a time.
__device__ __forceinline__ void fused_finish_block(double (*Ts)[65], const double *rs, bool diagb, int a0,
                                                   int b0, int K, const double *Gt, double *out,
                                                   double swt, bool cX, bool sX, int lane) {
  typedef double v2 __attribute__((ext_vector_type(2)));
  const int half = lane >> 5, lc = 2 * (lane & 31);
  const int gc = b0 + lc;
  const bool col_ok = gc < K;                      // K is even: gc + 1 < K too
  const double muc0 = rs[128 + lc], muc1 = rs[128 + lc + 1];
  const double sdc0 = rs[192 + lc], sdc1 = rs[192 + lc + 1];
#pragma unroll 1
  for (int it0 = 0; it0 < 32; it0 += 8) {
    v2 gv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int lr = 2 * (it0 + j) + half, gr = a0 + lr;
      gv[j] = (col_ok && gr < K) ? *reinterpret_cast<const v2 *>(Gt + (size_t)gr * K + gc) : (v2){0, 0};
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int lr = 2 * (it0 + j) + half, gr = a0 + lr;
      if (!(col_ok && gr < K)) continue;
      const double mur = rs[lr], sdr = rs[64 + lr];
      const double u0 = (diagb && lr > lc) ? Ts[lc][lr] : Ts[lr][lc];
      const double u1 = (diagb && lr > lc + 1) ? Ts[lc + 1][lr] : Ts[lr][lc + 1];
      double v0 = gv[j][0] - u0, v1 = gv[j][1] - u1;
      if (cX) { v0 -= swt * (mur * muc0); v1 -= swt * (mur * muc1); }
      if (sX) { v0 = v0 / (sdr * sdc0); v1 = v1 / (sdr * sdc1); }
      *reinterpret_cast<v2 *>(out + (size_t)gr * K + gc) = (v2){v0, v1};
      if (!diagb) { Ts[lr][lc] = v0; Ts[lr][lc + 1] = v1; }   // parked for the mirrored store
    }
  }
  if (diagb) return;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // mirrored block: rows b0.., columns a0..; out[b0 + r][a0 + c] = finished[c][r]
  const int gc2 = a0 + lc;
  if (gc2 >= K) return;
#pragma unroll 4
  for (int it = 0; it < 32; ++it) {
    const int lr = 2 * it + half, gr = b0 + lr;
    if (gr >= K) continue;
    *reinterpret_cast<v2 *>(out + (size_t)gr * K + gc2) = (v2){Ts[lc][lr], Ts[lc + 1][lr]};
  }
}

// One 64x64 sub-tile of a 128x128 upper tile (or one 128 x M panel of H) of one segment:
// ordered sum of the split partials (16-byte loads, four in flight, added in split order)
// staged in LDS, then finish_store_tile: total - partial, rank-1 centring, outer-std scaling
// and the mirrored store.  HBM-bound.
constexpr int APPLY_THREADS = 256;
constexpr int APPLY_SUB = 4;   // 64x64 sub-tiles per 128x128 tile
template <typename T, bool FOLD> __global__ __launch_bounds__(APPLY_THREADS) void apply_kernel(const FinArgs a) {
  const Geom &g = a.g;
  const int f = blockIdx.y;
  const int x = blockIdx.x;
  const int K = g.K, M = g.M;
  const long u0 = (long)f * a.splits;
  const double *fs = FOLD ? a.fstats + (size_t)f * fstat_len(K, M) : nullptr;
  const double swt = FOLD ? fs[2 * K + 2 * M] : 0.0;
  const bool cX = a.flags & CVM_CENTER_X, cY = a.flags & CVM_CENTER_Y;
  const bool sX = a.flags & CVM_SCALE_X, sY = a.flags & CVM_SCALE_Y;
  const size_t fo = (size_t)(a.seg0 + f);
  const char *ws0 = a.ws + (size_t)u0 * g.unit_bytes;
  if (x < g.nTiles * APPLY_SUB) {
    if (!a.out_XTX) return;
    const int t = x / APPLY_SUB, sub = x - t * APPLY_SUB;
    int ti, tj;
    decode_tile(t, g.P, ti, tj);
    const int si = sub >> 1, sj = sub & 1;
    if (ti == tj && si > sj) return;                 // strictly lower: mirror of sub-tile (0,1)
    const int a0 = ti * TILE + si * ST, b0 = tj * TILE + sj * ST;
    if (a0 >= K || b0 >= K) return;
    __shared__ __attribute__((aligned(16))) double sm[ST * (ST + 1)];
    double (*Ts)[ST + 1] = reinterpret_cast<double (*)[ST + 1]>(sm);
    constexpr int VW = 16 / sizeof(T);
    constexpr int LPR = ST / VW;
    typedef T vld_t __attribute__((ext_vector_type(VW)));
    const int tid = threadIdx.x;
    for (int q = tid; q < ST * LPR; q += APPLY_THREADS) {
      const int lr = q / LPR, lc = (q - lr * LPR) * VW;
      const size_t off = (size_t)t * TILE * TILE + (size_t)(si * ST + lr) * TILE + sj * ST + lc;
      const char *pp = ws0 + off * sizeof(T);
      double v[VW];
#pragma unroll
      for (int e = 0; e < VW; ++e) v[e] = 0;
#pragma unroll 4
      for (int p = 0; p < a.splits; ++p) {
        const vld_t qv = *reinterpret_cast<const vld_t *>(pp + (size_t)p * g.unit_bytes);
#pragma unroll
        for (int e = 0; e < VW; ++e) v[e] += (double)qv[e];
      }
#pragma unroll
      for (int e = 0; e < VW; ++e) Ts[lr][lc + e] = v[e];
    }
    __syncthreads();
    T *out = (T *)a.out_XTX + (FOLD ? fo * (size_t)K * K : 0);
    finish_store_tile<T, FOLD>(Ts, ti == tj && si == sj, a0, b0, K, (const T *)a.G, out, fs, swt, cX, sX,
                               tid, APPLY_THREADS);
  } else {
    if (!a.out_XTY || M == 0) return;
    const int ti = x - g.nTiles * APPLY_SUB;
    T *out = (T *)a.out_XTY + (FOLD ? fo * (size_t)K * M : 0);
    const T *Ht = (const T *)a.H;
    const size_t hoff = (g.tile_elems * sizeof(T) + 255) / 256 * 256;
    for (int e = threadIdx.x; e < TILE * M; e += APPLY_THREADS) {
      const int ra = e / M, m = e - ra * M;
      const int ga = ti * TILE + ra;
      if (ga >= K) continue;
      double v = 0;
      const char *pp = ws0 + hoff + ((size_t)ga * g.Mp + m) * sizeof(T);
#pragma unroll 4
      for (int p = 0; p < a.splits; ++p) v += (double)*reinterpret_cast<const T *>(pp + (size_t)p * g.unit_bytes);
      if (FOLD) {
        v = (double)Ht[(size_t)ga * M + m] - v;
        if (cX || cY) v -= swt * (fs[ga] * fs[2 * K + m]);
        if (sX && sY) v = v / (fs[K + ga] * fs[2 * K + M + m]);
        else if (sX) v = v / fs[K + ga];
        else if (sY) v = v / fs[2 * K + M + m];
      }
      out[(size_t)ga * M + m] = (T)v;
    }
  }
}

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
      for (; r + COL_UNROLL <= r1; r += COL_UNROLL) {
        int64_t rows[COL_UNROLL];
        T wr[COL_UNROLL];
        vec_t x[COL_UNROLL];
#pragma unroll
        for (int j = 0; j < COL_UNROLL; ++j) rows[j] = idx[r + j];
#pragma unroll
        for (int j = 0; j < COL_UNROLL; ++j) { x[j] = load(rows[j]); wr[j] = WEIGHTED ? wp[rows[j]] : (T)1; }
#pragma unroll
        for (int j = 0; j < COL_UNROLL; ++j) acc1(x[j], wr[j]);
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
constexpr int SMALL_ROWS = 32;
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
};

template <typename T, bool WEIGHTED> __global__ __launch_bounds__(256) void small_stats_kernel(const SmallArgs a) {
  const int f = blockIdx.x;
  const int K = a.K, M = a.M;
  const int64_t o0 = a.offs[a.seg0 + f];
  const int n = (int)(a.offs[a.seg0 + f + 1] - o0);
  const T *X = (const T *)a.X, *Y = (const T *)a.Y, *W = (const T *)a.w;
  __shared__ int64_t rows[SMALL_ROWS];
  __shared__ double wl[SMALL_ROWS];
  if (threadIdx.x < n) {
    const int64_t r = a.idx[o0 + threadIdx.x];
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
    for (int r = 0; r < n; ++r) {
      const T xv = isX ? X[rows[r] * (int64_t)K + cc] : Y[rows[r] * (int64_t)M + cc];
      if (sizeof(T) == 8) {
        const T pv = WEIGHTED ? (T)((T)wl[r] * xv) : xv;
        sv += (double)pv; qv += (double)(pv * xv);
      } else {
        const double pv = wl[r] * (double)xv;
        sv += pv; qv += pv * (double)xv;
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
    fs[isX ? K + cc : 2 * K + M + cc] = sd;
    T *omu = (T *)(isX ? a.out_muX : a.out_muY), *osd = (T *)(isX ? a.out_sdX : a.out_sdY);
    const size_t o = (size_t)(a.seg0 + f) * (isX ? K : M) + cc;
    if (omu) omu[o] = (T)mu;
    if (osd && (isX ? want_sdX : want_sdY)) osd[o] = (T)sd;
  }
}

template <typename T, bool WEIGHTED> __global__ __launch_bounds__(256) void small_apply_kernel(const SmallArgs a) {
  const int f = blockIdx.y;
  const int x = blockIdx.x;
  const int K = a.K, M = a.M;
  const int tid = threadIdx.x;
  const int64_t o0 = a.offs[a.seg0 + f];
  const int n = (int)(a.offs[a.seg0 + f + 1] - o0);
  const T *X = (const T *)a.X, *Y = (const T *)a.Y, *W = (const T *)a.w;
  const double *fs = a.fstats + (size_t)f * fstat_len(K, M);
  const double swt = fs[2 * K + 2 * M];
  const bool cX = a.flags & CVM_CENTER_X, cY = a.flags & CVM_CENTER_Y;
  const bool sX = a.flags & CVM_SCALE_X, sY = a.flags & CVM_SCALE_Y;
  const size_t fo = (size_t)(a.seg0 + f);
  // As: w * x, columns of the tile's rows; Bs: x (or y), columns of the tile's columns;
  // Ts: the finished tile for the transposed store -- it reuses the As/Bs space (33 KB per
  // workgroup instead of 66: four workgroups per CU keep more loads and stores in flight)
  __shared__ __attribute__((aligned(16))) double sm[ST * (ST + 1)];
  double (*As)[ST] = reinterpret_cast<double (*)[ST]>(sm);
  double (*Bs)[ST] = reinterpret_cast<double (*)[ST]>(sm + SMALL_ROWS * ST);
  double (*Ts)[ST + 1] = reinterpret_cast<double (*)[ST + 1]>(sm);
  __shared__ int64_t rows[SMALL_ROWS];
  __shared__ double wl[SMALL_ROWS];
  if (tid < n) {
    const int64_t r = a.idx[o0 + tid];
    rows[tid] = r;
    wl[tid] = WEIGHTED ? (double)W[r] : 1.0;
  }
  __syncthreads();
  if (x < a.nT64) {
    if (!a.out_XTX) return;
    int ti, tj;
    decode_tile(x, a.P64, ti, tj);
    const int a0 = ti * ST, b0 = tj * ST;
    const int ty = tid >> 4, tx = tid & 15;      // rows 4ty.., columns 4tx..
    for (int e = tid; e < n * ST; e += 256) {
      const int r = e / ST, c = e - r * ST;
      const T xa = (a0 + c < K) ? X[rows[r] * (int64_t)K + a0 + c] : (T)0;
      const T xb = (b0 + c < K) ? X[rows[r] * (int64_t)K + b0 + c] : (T)0;
      As[r][c] = (sizeof(T) == 8) ? (double)(WEIGHTED ? (T)((T)wl[r] * xa) : xa) : wl[r] * (double)xa;
      Bs[r][c] = (double)xb;
    }
    __syncthreads();
    double acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[i][j] = 0;
    for (int r = 0; r < n; ++r) {
      double av[4], bv[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) { av[i] = As[r][4 * ty + i]; bv[i] = Bs[r][4 * tx + i]; }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] += av[i] * bv[j];
    }
    __syncthreads();   // every thread is done with As/Bs: Ts may overwrite them
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) Ts[4 * ty + i][4 * tx + j] = acc[i][j];
    __syncthreads();
    T *out = (T *)a.out_XTX + fo * (size_t)K * K;
    finish_store_tile<T, true>(Ts, ti == tj, a0, b0, K, (const T *)a.G, out, fs, swt, cX, sX, tid, 256);
  } else {
    if (!a.out_XTY || M == 0) return;
    const int ti = x - a.nT64;
    const int a0 = ti * ST;
    const T *Ht = (const T *)a.H;
    T *out = (T *)a.out_XTY + fo * (size_t)K * M;
    for (int e = tid; e < n * ST; e += 256) {
      const int r = e / ST, c = e - r * ST;
      const T xa = (a0 + c < K) ? X[rows[r] * (int64_t)K + a0 + c] : (T)0;
      As[r][c] = (sizeof(T) == 8) ? (double)(WEIGHTED ? (T)((T)wl[r] * xa) : xa) : wl[r] * (double)xa;
    }
    for (int m0 = 0; m0 < M; m0 += ST) {
      __syncthreads();
      for (int e = tid; e < n * ST; e += 256) {
        const int r = e / ST, c = e - r * ST;
        Bs[r][c] = (m0 + c < M) ? (double)Y[rows[r] * (int64_t)M + m0 + c] : 0.0;
      }
      __syncthreads();
      const int mw = (M - m0 < ST) ? M - m0 : ST;
      for (int e = tid; e < ST * mw; e += 256) {
        const int la = e / mw, lm = e - la * mw;
        const int ga = a0 + la, gm = m0 + lm;
        if (ga >= K) continue;
        double acc = 0;
        for (int r = 0; r < n; ++r) acc += As[r][la] * Bs[r][lm];
        double v = (double)Ht[(size_t)ga * M + gm] - acc;
        if (cX || cY) v -= swt * (fs[ga] * fs[2 * K + gm]);
        if (sX && sY) v = v / (fs[K + ga] * fs[2 * K + M + gm]);
        else if (sX) v = v / fs[K + ga];
        else if (sY) v = v / fs[2 * K + M + gm];
        out[(size_t)ga * M + gm] = (T)v;
      }
    }
  }
}

// ----------------------------------------------------------------------------------
// host side
// ----------------------------------------------------------------------------------
thread_local char g_err[512] = "";
int fail(int code, const char *fmt, const char *detail = "") {
  snprintf(g_err, sizeof(g_err), fmt, detail);
  return code;
}
#define HIP_OK(expr)                                                             \
  do {                                                                           \
    hipError_t e_ = (expr);                                                      \
    if (e_ != hipSuccess) return fail(CVM_ELAUNCH, #expr ": %s", hipGetErrorString(e_)); \
  } while (0)

// ---- optional per-launch timing of the Gram kernel (bench.py's roofline figure) --------
struct TimedLaunch { hipEvent_t a, b; int kind; };
bool g_timing = false;
TimedLaunch g_timed[8192];
int g_ntimed = 0;
int g_timing_kind = 0;   // 0: fit stage, 1: fold stage

struct Plan {
  Geom g;
  int splits;
  int64_t folds_per_batch;
  size_t fstat_bytes_per_fold;
};

// Row splits per segment.  Model: TARGET_WG workgroups are resident at a time and take
// equal time, so W = items*splits workgroups cost ceil(W/TARGET_WG) rounds; pick the split
// count with the best fill, lightly preferring fewer splits (less partial traffic).
int choose_splits(int64_t n_seg, int64_t max_rows, const Geom &g, int TARGET_WG) {
  const int64_t items = (n_seg > 0 ? n_seg : 1) * g.nT;
  int64_t cap = max_rows / 64;                       // >= 64 rows per split
  const int64_t mem_cap = (int64_t)(((size_t)3 << 30) / ((size_t)(n_seg > 0 ? n_seg : 1) * g.unit_bytes));
  if (cap > mem_cap) cap = mem_cap;
  if (cap > 64) cap = 64;
  if (cap < 1) cap = 1;
  // estimated launch time in 16-row stages: rounds of workgroups x (stages of one split + a
  // fixed per-workgroup cost: prologue, epilogue, partial store ~ 6 stages); fewest splits on
  // near-ties (less partial traffic; one split per fold also lets the float64 kernel finish
  // folds in its epilogue)
  int best = 1;
  double best_score = 1e300;
  for (int64_t s = 1; s <= cap; ++s) {
    const int64_t W = items * s;
    const int64_t rounds = (W + TARGET_WG - 1) / TARGET_WG;
    const int64_t stages = ((max_rows + s - 1) / s + STAGE_ROWS - 1) / STAGE_ROWS;
    const double score = (double)rounds * (double)(stages + 6) + 0.5 * (double)s;
    if (score < best_score) { best_score = score; best = (int)s; }
  }
  return best;
}

// which Gram kernel variant a problem gets (pointers from torch are 256-byte aligned; a
// misaligned X falls back to the register path at launch, only the split heuristic differs)
int target_wg(int K, int M, int esize) {
  return (esize == 8 && ((size_t)K * esize) % 16 == 0 && M % 2 == 0) ? TARGET_WG_2 : TARGET_WG_1;
}

int make_plan(int64_t n_folds, int64_t max_rows, int K, int M, int dtype, unsigned flags,
              size_t ws_bytes, bool fold_mode, Plan &p) {
  const int esize = dtype == CVM_F64 ? 8 : 4;
  const int diag_only = fold_mode && !(flags & CVM_RET_XTX);
  p.g = make_geom(K, M, esize, diag_only);
  p.splits = choose_splits(n_folds, max_rows, p.g, target_wg(K, M, esize));
  p.fstat_bytes_per_fold = fold_mode ? align_up(fstat_len(K, M) * 8, 256) : 0;
  for (;;) {
    const size_t per_fold = (size_t)p.splits * p.g.unit_bytes + p.fstat_bytes_per_fold;
    int64_t nb = (int64_t)(ws_bytes / per_fold);
    if (nb >= 1) { p.folds_per_batch = nb < n_folds ? nb : n_folds; return CVM_OK; }
    if (p.splits == 1) return CVM_EWORKSPACE;
    p.splits = (p.splits + 1) / 2;
  }
}

// can this problem take the float64 LDS-DMA kernel (wgram4_kernel)?
template <typename T> bool wgram4_ok(const WgramArgs<T> &a, bool aligned) {
  static const bool force_fallback = getenv("CVM_FORCE_FALLBACK") && atoi(getenv("CVM_FORCE_FALLBACK")) != 0;
  return sizeof(T) == 8 && aligned && (a.g.M % 2 == 0) && ((uintptr_t)a.Y % 16 == 0) &&
         ((uintptr_t)a.w % 8 == 0) && !force_fallback;
}

template <typename T>
int launch_wgram(const WgramArgs<T> &a, bool weighted, bool gather, bool aligned, hipStream_t st,
                 bool fused = false) {
  const long per_xcd = (a.n_items + 7) / 8;
  WgramArgs<T> args = a;
  args.items_per_xcd = per_xcd;
  // CVM_FORCE_FALLBACK=1 sends float64 problems through the general (register-staged)
  // kernel too -- used by the tests to cover both kernels.  The ablation switches of
  // CVM_DEBUG (wrong results by design) exist in the -DCVM_STAMPS diagnostic build only.
#ifdef CVM_STAMPS
  static const int dbg_env = getenv("CVM_DEBUG") ? atoi(getenv("CVM_DEBUG")) : 0;
#else
  static const int dbg_env = 0;
#endif
  args.dbg = dbg_env;
  const dim3 grid((unsigned)(per_xcd * 8)), block(NTHREADS);
  const size_t lds = 2 * BUF_ELEMS * sizeof(T) + 3 * STAGE_ROWS * sizeof(int64_t);
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
#define CVM_LAUNCH(W, GA, AL)                                                                 \
  do {                                                                                     \
    static unsigned long long attr_done = 0;   /* one bit per device */                   \
    if (!((attr_done >> (dev & 63)) & 1ull)) {                                             \
      HIP_OK(hipFuncSetAttribute((const void *)wgram_kernel<T, W, GA, AL>,                 \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   \
      attr_done |= 1ull << (dev & 63);                                                     \
    }                                                                                      \
    hipLaunchKernelGGL((wgram_kernel<T, W, GA, AL>), grid, block, lds, st, args);          \
  } while (0)
  TimedLaunch *tl = nullptr;
  if (g_timing && g_ntimed < 8192) {
    tl = &g_timed[g_ntimed];
    if (!tl->a) { HIP_OK(hipEventCreate(&tl->a)); HIP_OK(hipEventCreate(&tl->b)); }
    tl->kind = g_timing_kind;
    HIP_OK(hipEventRecord(tl->a, st));
  }
  constexpr bool CAN_DMA = sizeof(T) == 8;
  const bool fast = wgram4_ok<T>(a, aligned) && !(dbg_env & 16);
  if (fused && !(fast && gather)) return fail(CVM_EINVAL, "launch_wgram: fused epilogue needs the float64 fast path%s");
  if (fast) {
    if constexpr (CAN_DMA) {
      const dim3 block4(NT4);
#define CVM_LAUNCH4(W, GA)                                                                  \
  do {                                                                                      \
    static unsigned long long attr_done = 0;                                                \
    if (!((attr_done >> (dev & 63)) & 1ull)) {                                              \
      HIP_OK(hipFuncSetAttribute((const void *)wgram4_kernel<W, GA>,                        \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS4_BYTES)); \
      attr_done |= 1ull << (dev & 63);                                                      \
    }                                                                                       \
    hipLaunchKernelGGL((wgram4_kernel<W, GA>), grid, block4, LDS4_BYTES, st, args);         \
  } while (0)
      if (fused) {
#define CVM_LAUNCH4F(W)                                                                     \
  do {                                                                                      \
    static unsigned long long attr_done = 0;                                                \
    if (!((attr_done >> (dev & 63)) & 1ull)) {                                              \
      HIP_OK(hipFuncSetAttribute((const void *)wgram4_kernel<W, true, true>,                \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS4_BYTES)); \
      attr_done |= 1ull << (dev & 63);                                                      \
    }                                                                                       \
    hipLaunchKernelGGL((wgram4_kernel<W, true, true>), grid, block4, LDS4_BYTES, st, args); \
  } while (0)
        if (weighted) CVM_LAUNCH4F(true); else CVM_LAUNCH4F(false);
#undef CVM_LAUNCH4F
      } else if (weighted) { if (gather) CVM_LAUNCH4(true, true); else CVM_LAUNCH4(true, false); }
      else { if (gather) CVM_LAUNCH4(false, true); else CVM_LAUNCH4(false, false); }
#undef CVM_LAUNCH4
    }
  } else if (weighted) {
    if (gather) { if (aligned) CVM_LAUNCH(true, true, true); else CVM_LAUNCH(true, true, false); }
    else { if (aligned) CVM_LAUNCH(true, false, true); else CVM_LAUNCH(true, false, false); }
  } else {
    if (gather) { if (aligned) CVM_LAUNCH(false, true, true); else CVM_LAUNCH(false, true, false); }
    else { if (aligned) CVM_LAUNCH(false, false, true); else CVM_LAUNCH(false, false, false); }
  }
#undef CVM_LAUNCH
  if (tl) { HIP_OK(hipEventRecord(tl->b, st)); ++g_ntimed; }
  HIP_OK(hipGetLastError());
  return CVM_OK;
}

bool rows_aligned(const void *X, int K, int esize) {
  return ((uintptr_t)X % 16 == 0) && (((size_t)K * esize) % 16 == 0);
}

template <typename T>
int gram_fit_impl(const void *X, const void *Y, const void *w, int64_t N, int K, int M, int dtype,
                  void *G, void *H, double *gstats, int32_t *neg_flag, void *ws, size_t ws_bytes,
                  hipStream_t st) {
  Plan p;
  int rc = make_plan(1, N, K, M, dtype, CVM_RET_XTX | CVM_RET_XTY, ws_bytes, false, p);
  if (rc != CVM_OK) return fail(rc, "cvm_gram_fit: workspace too small%s");
  WgramArgs<T> a;
  memset(&a, 0, sizeof(a));
  a.X = (const T *)X; a.Y = (const T *)Y; a.w = (const T *)w;
  a.idx = nullptr; a.offs = nullptr; a.N = N; a.seg0 = 0;
  a.n_seg = 1; a.splits = p.splits; a.g = p.g;
  a.n_items = (long)p.splits * p.g.nT; a.items_per_xcd = 0;
  a.ws = (char *)ws;
  g_timing_kind = 0;
  rc = launch_wgram<T>(a, w != nullptr, false, rows_aligned(X, K, sizeof(T)), st);
  if (rc != CVM_OK) return rc;
  FinArgs f;
  memset(&f, 0, sizeof(f));
  f.g = p.g; f.splits = p.splits; f.n_seg = 1; f.seg0 = 0; f.ws = (const char *)ws;
  f.w = w; f.out_XTX = G; f.out_XTY = (Y && M > 0) ? H : nullptr; f.neg_flag = neg_flag;
  hipLaunchKernelGGL((fit_stats_kernel<T>), dim3(8), dim3(256), 0, st, f, gstats);
  hipLaunchKernelGGL((apply_kernel<T, false>), dim3(p.g.nTiles * APPLY_SUB + p.g.P, 1),
                     dim3(APPLY_THREADS), 0, st, f);
  HIP_OK(hipGetLastError());
  return CVM_OK;
}

template <typename T>
int small_fold_impl(const void *X, const void *Y, const void *w, const int64_t *idx, const int64_t *offsets,
                    int64_t n_folds, int K, int M, unsigned flags, double ddof, double resolution,
                    const void *G, const void *H, const double *gstats, void *out_XTX, void *out_XTY,
                    void *out_muX, void *out_sdX, void *out_muY, void *out_sdY, double *out_fold,
                    void *ws, size_t ws_bytes, hipStream_t st) {
  const size_t per_fold = fstat_len(K, M) * 8;
  int64_t nb_max = (int64_t)(ws_bytes / per_fold);
  if (nb_max < 1) return fail(CVM_EWORKSPACE, "cvm_fold_update: workspace cannot hold one fold%s");
  if (nb_max > 32768) nb_max = 32768;   // grid.y
  SmallArgs a;
  memset(&a, 0, sizeof(a));
  a.X = X; a.Y = Y; a.w = w; a.idx = idx; a.offs = offsets; a.K = K; a.M = M;
  a.G = G; a.H = H; a.gstats = gstats; a.fstats = (double *)ws;
  a.out_XTX = (flags & CVM_RET_XTX) ? out_XTX : nullptr;
  a.out_XTY = (flags & CVM_RET_XTY) ? out_XTY : nullptr;
  a.out_muX = out_muX; a.out_sdX = out_sdX; a.out_muY = out_muY; a.out_sdY = out_sdY;
  a.out_fold = out_fold; a.ddof = ddof; a.resolution = resolution; a.flags = flags;
  a.P64 = (K + ST - 1) / ST; a.nT64 = a.P64 * (a.P64 + 1) / 2;
  for (int64_t f0 = 0; f0 < n_folds; f0 += nb_max) {
    const int64_t nb = (n_folds - f0 < nb_max) ? n_folds - f0 : nb_max;
    a.seg0 = f0;
    const dim3 gs((unsigned)nb), ga((unsigned)(a.nT64 + a.P64), (unsigned)nb);
    if (w) {
      hipLaunchKernelGGL((small_stats_kernel<T, true>), gs, dim3(256), 0, st, a);
      if (a.out_XTX || a.out_XTY) hipLaunchKernelGGL((small_apply_kernel<T, true>), ga, dim3(256), 0, st, a);
    } else {
      hipLaunchKernelGGL((small_stats_kernel<T, false>), gs, dim3(256), 0, st, a);
      if (a.out_XTX || a.out_XTY) hipLaunchKernelGGL((small_apply_kernel<T, false>), ga, dim3(256), 0, st, a);
    }
    HIP_OK(hipGetLastError());
  }
  return CVM_OK;
}

// statistics-only fold stage: colstats_kernel + fold_stats_kernel, no Gram launch
template <typename T>
int fold_statistics_impl(const void *X, const void *Y, const void *w, const int64_t *idx,
                         const int64_t *offsets, int64_t n_folds, int64_t max_rows, int K, int M,
                         unsigned flags, double ddof, double resolution, const double *gstats,
                         void *out_muX, void *out_sdX, void *out_muY, void *out_sdY, double *out_fold,
                         void *ws, size_t ws_bytes, hipStream_t st) {
  Geom g = make_geom(K, M, sizeof(T), 1);
  g.tile_elems = 0; g.h_elems = 0;
  g.unit_bytes = align_up(g.stat_len * 8, 256);
  const size_t fst = align_up(fstat_len(K, M) * 8, 256);
  // rows per unit: short enough for >1000 workgroups in flight at the benchmark shapes, long
  // enough that the units' statistics vectors stay a few per cent of the bytes streamed
  int64_t splits = (max_rows + CVM_COL_ROWS - 1) / CVM_COL_ROWS;
  if (splits < 1) splits = 1;
  if (splits > 1024) splits = 1024;
  while (splits > 1 && (size_t)splits * g.unit_bytes + fst > ws_bytes) splits /= 2;
  const size_t per_fold = (size_t)splits * g.unit_bytes + fst;
  if (per_fold > ws_bytes) return fail(CVM_EWORKSPACE, "cvm_fold_update: workspace cannot hold one fold%s");
  int64_t per_batch = (int64_t)(ws_bytes / per_fold);
  if (per_batch > 32768) per_batch = 32768;
  const bool aligned = rows_aligned(X, K, sizeof(T));
  constexpr int VEC = 16 / (int)sizeof(T);
  const int nxb = (K + COL_THREADS * VEC - 1) / (COL_THREADS * VEC);
  for (int64_t f0 = 0; f0 < n_folds; f0 += per_batch) {
    const int64_t nb = (n_folds - f0 < per_batch) ? n_folds - f0 : per_batch;
    ColArgs c;
    c.X = X; c.Y = Y; c.w = w; c.idx = idx; c.offs = offsets; c.seg0 = f0; c.splits = (int)splits;
    c.g = g; c.ws = (char *)ws;
    const dim3 grid((unsigned)(nxb + 1), (unsigned)(nb * splits));
    if (w) {
      if (aligned) hipLaunchKernelGGL((colstats_kernel<T, true, true>), grid, dim3(COL_THREADS), 0, st, c);
      else hipLaunchKernelGGL((colstats_kernel<T, true, false>), grid, dim3(COL_THREADS), 0, st, c);
    } else {
      if (aligned) hipLaunchKernelGGL((colstats_kernel<T, false, true>), grid, dim3(COL_THREADS), 0, st, c);
      else hipLaunchKernelGGL((colstats_kernel<T, false, false>), grid, dim3(COL_THREADS), 0, st, c);
    }
    FinArgs f;
    memset(&f, 0, sizeof(f));
    f.g = g; f.splits = (int)splits; f.n_seg = (int)nb; f.seg0 = f0; f.ws = (const char *)ws;
    f.fstats = (double *)((char *)ws + (size_t)nb * splits * g.unit_bytes);
    f.offs = offsets; f.w = w; f.gstats = gstats;
    f.out_muX = out_muX; f.out_sdX = out_sdX; f.out_muY = out_muY; f.out_sdY = out_sdY;
    f.out_fold = out_fold; f.ddof = ddof; f.resolution = resolution; f.flags = flags;
    hipLaunchKernelGGL((fold_stats_kernel<T>), dim3((unsigned)nb, (unsigned)fold_stats_chunks(K, M, nb)), dim3(256), 0, st, f);
    HIP_OK(hipGetLastError());
  }
  return CVM_OK;
}

template <typename T>
int fold_update_impl(const void *X, const void *Y, const void *w, const int64_t *idx,
                     const int64_t *offsets, const int64_t *host_offsets, int64_t n_folds, int64_t N,
                     int K, int M, int dtype, unsigned flags, double ddof, double resolution,
                     const void *G, const void *H, const double *gstats, void *out_XTX,
                     void *out_XTY, void *out_muX, void *out_sdX, void *out_muY, void *out_sdY,
                     double *out_fold, void *ws, size_t ws_bytes, hipStream_t st) {
  int64_t max_rows = 0;
  for (int64_t f = 0; f < n_folds; ++f) {
    const int64_t n = host_offsets[f + 1] - host_offsets[f];
    if (n < 0) return fail(CVM_EINVAL, "cvm_fold_update: offsets must be non-decreasing%s");
    if (n > max_rows) max_rows = n;
  }
  if (max_rows <= SMALL_ROWS)
    return small_fold_impl<T>(X, Y, w, idx, offsets, n_folds, K, M, flags, ddof, resolution, G, H, gstats,
                              out_XTX, out_XTY, out_muX, out_sdX, out_muY, out_sdY, out_fold, ws, ws_bytes, st);
  const bool want_xtx = (flags & CVM_RET_XTX) && out_XTX, want_xty = (flags & CVM_RET_XTY) && out_XTY;
  if (!want_xtx && !want_xty)   // statistics only: stream the rows once, no Gram launch
    return fold_statistics_impl<T>(X, Y, w, idx, offsets, n_folds, max_rows, K, M, flags, ddof, resolution,
                                   gstats, out_muX, out_sdX, out_muY, out_sdY, out_fold, ws, ws_bytes, st);
  Plan p;
  // (planned against an unlimited workspace first: the fused route below needs far less than
  //  the partials the general route plans for)
  int rc = make_plan(n_folds, max_rows, K, M, dtype, flags, (size_t)1 << 60, true, p);
  const bool aligned = rows_aligned(X, K, sizeof(T));
  {
    // Folds too small to be split over workgroups (one unit per fold): finish in the Gram
    // kernel's epilogue instead of writing partials for apply_kernel to read back.  The fold
    // statistics the epilogue needs come from the streaming kernel first.
    static const bool no_fused = getenv("CVM_NO_FUSED") && atoi(getenv("CVM_NO_FUSED")) != 0;
    WgramArgs<T> probe;
    memset(&probe, 0, sizeof(probe));
    probe.Y = (const T *)Y; probe.w = (const T *)w; probe.g = p.g;
    if (p.splits == 1 && want_xtx && !no_fused && wgram4_ok<T>(probe, aligned)) {
      Geom gs = make_geom(K, M, sizeof(T), 1);
      gs.tile_elems = 0; gs.h_elems = 0;
      gs.unit_bytes = align_up(gs.stat_len * 8, 256);
      const size_t fst = align_up(fstat_len(K, M) * 8, 256);
      int64_t csplits = (max_rows + CVM_COL_ROWS - 1) / CVM_COL_ROWS;
      if (csplits < 1) csplits = 1;
      while (csplits > 1 && (size_t)csplits * gs.unit_bytes + fst > ws_bytes) csplits /= 2;
      const size_t per_fold = (size_t)csplits * gs.unit_bytes + fst;
      if (per_fold > ws_bytes) return fail(CVM_EWORKSPACE, "cvm_fold_update: workspace cannot hold one fold%s");
      int64_t per_batch = (int64_t)(ws_bytes / per_fold);
      if (per_batch > 16384) per_batch = 16384;
      constexpr int VEC = 16 / (int)sizeof(T);
      const int nxb = (K + COL_THREADS * VEC - 1) / (COL_THREADS * VEC);
      for (int64_t f0 = 0; f0 < n_folds; f0 += per_batch) {
        const int64_t nb = (n_folds - f0 < per_batch) ? n_folds - f0 : per_batch;
        ColArgs c;
        c.X = X; c.Y = Y; c.w = w; c.idx = idx; c.offs = offsets; c.seg0 = f0; c.splits = (int)csplits;
        c.g = gs; c.ws = (char *)ws;
        const dim3 cgrid((unsigned)(nxb + 1), (unsigned)(nb * csplits));
        if (w) hipLaunchKernelGGL((colstats_kernel<T, true, true>), cgrid, dim3(COL_THREADS), 0, st, c);
        else hipLaunchKernelGGL((colstats_kernel<T, false, true>), cgrid, dim3(COL_THREADS), 0, st, c);
        FinArgs f;
        memset(&f, 0, sizeof(f));
        f.g = gs; f.splits = (int)csplits; f.n_seg = (int)nb; f.seg0 = f0; f.ws = (const char *)ws;
        f.fstats = (double *)((char *)ws + (size_t)nb * csplits * gs.unit_bytes);
        f.offs = offsets; f.w = w; f.gstats = gstats;
        f.out_muX = out_muX; f.out_sdX = out_sdX; f.out_muY = out_muY; f.out_sdY = out_sdY;
        f.out_fold = out_fold; f.ddof = ddof; f.resolution = resolution; f.flags = flags;
        hipLaunchKernelGGL((fold_stats_kernel<T>), dim3((unsigned)nb, (unsigned)fold_stats_chunks(K, M, nb)),
                           dim3(256), 0, st, f);
        WgramArgs<T> a;
        memset(&a, 0, sizeof(a));
        a.X = (const T *)X; a.Y = (const T *)Y; a.w = (const T *)w;
        a.idx = idx; a.offs = offsets; a.N = N; a.seg0 = f0;
        a.n_seg = (int)nb; a.splits = 1; a.g = p.g;
        a.n_items = (long)nb * p.g.nT; a.items_per_xcd = 0;
        a.ws = nullptr;
        a.fstats = f.fstats; a.G = G; a.H = H;
        a.out_XTX = out_XTX; a.out_XTY = want_xty ? out_XTY : nullptr; a.flags = flags;
        g_timing_kind = 1;
        rc = launch_wgram<T>(a, w != nullptr, true, aligned, st, true);
        if (rc != CVM_OK) return rc;
      }
      return CVM_OK;
    }
  }
  rc = make_plan(n_folds, max_rows, K, M, dtype, flags, ws_bytes, true, p);
  if (rc != CVM_OK) return fail(rc, "cvm_fold_update: workspace cannot hold one fold%s");
  for (int64_t f0 = 0; f0 < n_folds; f0 += p.folds_per_batch) {
    const int64_t nb = (n_folds - f0 < p.folds_per_batch) ? n_folds - f0 : p.folds_per_batch;
    char *units = (char *)ws;
    double *fstats = (double *)((char *)ws + (size_t)nb * p.splits * p.g.unit_bytes);
    WgramArgs<T> a;
    memset(&a, 0, sizeof(a));
    a.X = (const T *)X; a.Y = (const T *)Y; a.w = (const T *)w;
    a.idx = idx; a.offs = offsets; a.N = N; a.seg0 = f0;
    a.n_seg = (int)nb; a.splits = p.splits; a.g = p.g;
    a.n_items = (long)nb * p.splits * p.g.nT; a.items_per_xcd = 0;
    a.ws = units;
    g_timing_kind = 1;
    rc = launch_wgram<T>(a, w != nullptr, true, aligned, st);
    if (rc != CVM_OK) return rc;
    FinArgs f;
    memset(&f, 0, sizeof(f));
    f.g = p.g; f.splits = p.splits; f.n_seg = (int)nb; f.seg0 = f0; f.ws = units;
    f.fstats = (double *)((char *)fstats);
    f.offs = offsets; f.w = w; f.G = G; f.H = H; f.gstats = gstats;
    f.out_XTX = (flags & CVM_RET_XTX) ? out_XTX : nullptr;
    f.out_XTY = (flags & CVM_RET_XTY) ? out_XTY : nullptr;
    f.out_muX = out_muX; f.out_sdX = out_sdX; f.out_muY = out_muY; f.out_sdY = out_sdY;
    f.out_fold = out_fold; f.ddof = ddof; f.resolution = resolution; f.flags = flags;
    // fstats rows are fstat_len doubles apart inside the 256-byte aligned slots? keep dense
    hipLaunchKernelGGL((fold_stats_kernel<T>), dim3((unsigned)nb, (unsigned)fold_stats_chunks(K, M, nb)), dim3(256), 0, st, f);
    if (f.out_XTX || f.out_XTY) {
      hipLaunchKernelGGL((apply_kernel<T, true>), dim3(p.g.nTiles * APPLY_SUB + p.g.P, (unsigned)nb),
                         dim3(APPLY_THREADS), 0, st, f);
    }
    HIP_OK(hipGetLastError());
  }
  return CVM_OK;
}

// One-sweep cross-validation: when the folds partition the rows, the
// full-data matrices are the ordered sum of the folds' validation matrices, G = sum_f G_Vf.
// sweep_fit runs the Gram kernel ONCE over all folds (gathered), sums every unit's partials
// into G, H, gstats and leaves the partials in the workspace; sweep_folds then only runs
// the finalize kernels on them.  Half the flops of fit + fold update.
template <typename T>
int sweep_fit_impl(const void *X, const void *Y, const void *w, const int64_t *idx,
                   const int64_t *offsets, const int64_t *host_offsets, int64_t n_folds, int64_t N,
                   int K, int M, int dtype, void *G, void *H, double *gstats, int32_t *neg_flag,
                   void *ws, size_t ws_bytes, hipStream_t st, int64_t *splits_out) {
  int64_t max_rows = 0;
  for (int64_t f = 0; f < n_folds; ++f) {
    const int64_t n = host_offsets[f + 1] - host_offsets[f];
    if (n < 0) return fail(CVM_EINVAL, "cvm_sweep_fit: offsets must be non-decreasing%s");
    if (n > max_rows) max_rows = n;
  }
  if (host_offsets[n_folds] - host_offsets[0] != N)
    return fail(CVM_EINVAL, "cvm_sweep_fit: the folds must cover each of the N rows exactly once%s");
  Plan p;
  const unsigned flags = CVM_RET_XTX | CVM_RET_XTY;
  int rc = make_plan(n_folds, max_rows, K, M, dtype, flags, ws_bytes, true, p);
  if (rc != CVM_OK || p.folds_per_batch < n_folds)
    return fail(CVM_EWORKSPACE, "cvm_sweep_fit: the workspace must hold the partials of all folds%s");
  WgramArgs<T> a;
  memset(&a, 0, sizeof(a));
  a.X = (const T *)X; a.Y = (const T *)Y; a.w = (const T *)w;
  a.idx = idx; a.offs = offsets; a.N = N; a.seg0 = 0;
  a.n_seg = (int)n_folds; a.splits = p.splits; a.g = p.g;
  a.n_items = (long)n_folds * p.splits * p.g.nT; a.items_per_xcd = 0;
  a.ws = (char *)ws;
  g_timing_kind = 1;
  rc = launch_wgram<T>(a, w != nullptr, true, rows_aligned(X, K, sizeof(T)), st);
  if (rc != CVM_OK) return rc;
  FinArgs f;
  memset(&f, 0, sizeof(f));
  f.g = p.g; f.splits = (int)(n_folds * p.splits);   // every unit of every fold, fold-major
  f.n_seg = 1; f.seg0 = 0; f.ws = (const char *)ws;
  f.w = w; f.out_XTX = G; f.out_XTY = (Y && M > 0) ? H : nullptr; f.neg_flag = neg_flag;
  hipLaunchKernelGGL((fit_stats_kernel<T>), dim3(8), dim3(256), 0, st, f, gstats);
  hipLaunchKernelGGL((apply_kernel<T, false>), dim3(p.g.nTiles * APPLY_SUB + p.g.P, 1),
                     dim3(APPLY_THREADS), 0, st, f);
  HIP_OK(hipGetLastError());
  if (splits_out) *splits_out = p.splits;
  return CVM_OK;
}

template <typename T>
int sweep_folds_impl(const int64_t *offsets, int64_t n_folds, int K, int M, int dtype, unsigned flags,
                     double ddof, double resolution, int weighted, const void *G, const void *H,
                     const double *gstats, void *out_XTX, void *out_XTY, void *out_muX, void *out_sdX,
                     void *out_muY, void *out_sdY, double *out_fold, void *ws, size_t ws_bytes,
                     int64_t splits, hipStream_t st) {
  const Geom g = make_geom(K, M, sizeof(T), 0);
  const size_t units = (size_t)n_folds * (size_t)splits * g.unit_bytes;
  if (units + (size_t)n_folds * fstat_len(K, M) * 8 > ws_bytes)
    return fail(CVM_EWORKSPACE, "cvm_sweep_folds: workspace smaller than the one cvm_sweep_fit filled%s");
  FinArgs f;
  memset(&f, 0, sizeof(f));
  f.g = g; f.splits = (int)splits; f.n_seg = (int)n_folds; f.seg0 = 0; f.ws = (const char *)ws;
  f.fstats = (double *)((char *)ws + units);
  f.offs = offsets; f.w = weighted ? (const void *)G : nullptr;   // non-null = weighted
  f.G = G; f.H = H; f.gstats = gstats;
  f.out_XTX = (flags & CVM_RET_XTX) ? out_XTX : nullptr;
  f.out_XTY = (flags & CVM_RET_XTY) ? out_XTY : nullptr;
  f.out_muX = out_muX; f.out_sdX = out_sdX; f.out_muY = out_muY; f.out_sdY = out_sdY;
  f.out_fold = out_fold; f.ddof = ddof; f.resolution = resolution; f.flags = flags;
  hipLaunchKernelGGL((fold_stats_kernel<T>), dim3((unsigned)n_folds, (unsigned)fold_stats_chunks(K, M, n_folds)), dim3(256), 0, st, f);
  if (f.out_XTX || f.out_XTY)
    hipLaunchKernelGGL((apply_kernel<T, true>), dim3(g.nTiles * APPLY_SUB + g.P, (unsigned)n_folds),
                       dim3(APPLY_THREADS), 0, st, f);
  HIP_OK(hipGetLastError());
  return CVM_OK;
}

}  // namespace

// ----------------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------------
extern "C" {

const char *cvm_version(void) { return "cvmhip 0.1.0 (gfx950) built " __DATE__ " " __TIME__; }
const char *cvm_last_error(void) { return g_err; }

size_t cvm_gstats_len(int K, int M) { return 2 * (size_t)K + 2 * (size_t)M + 2; }

size_t cvm_fit_workspace_bytes(int64_t N, int K, int M, int dtype) {
  const Geom g = make_geom(K, M, dtype == CVM_F64 ? 8 : 4, 0);
  return (size_t)choose_splits(1, N, g, target_wg(K, M, dtype == CVM_F64 ? 8 : 4)) * g.unit_bytes;
}

int cvm_gram_fit(const void *X, const void *Y, const void *w, int64_t N, int K, int M, int dtype,
                 void *G, void *H, double *gstats, int32_t *neg_flag, void *ws, size_t ws_bytes,
                 void *stream) {
  if (!X || !G || !gstats || !ws) return fail(CVM_EINVAL, "cvm_gram_fit: null pointer%s");
  if (N < 0 || K <= 0 || M < 0 || (M > 0 && (!Y || !H)) || (M == 0 && Y))
    return fail(CVM_EINVAL, "cvm_gram_fit: bad shape%s");
  if (dtype == CVM_F64)
    return gram_fit_impl<double>(X, Y, w, N, K, M, dtype, G, H, gstats, neg_flag, ws, ws_bytes,
                                 (hipStream_t)stream);
  if (dtype == CVM_F32)
    return gram_fit_impl<float>(X, Y, w, N, K, M, dtype, G, H, gstats, neg_flag, ws, ws_bytes,
                                (hipStream_t)stream);
  return fail(CVM_EINVAL, "cvm_gram_fit: dtype must be CVM_F32 or CVM_F64%s");
}

size_t cvm_fold_workspace_bytes(int64_t n_folds, int64_t n_idx, int64_t max_fold_rows, int K, int M,
                                int dtype, unsigned flags) {
  (void)n_idx;
  if (max_fold_rows <= SMALL_ROWS) {   // direct path: only the per-fold statistics live in ws
    const int64_t nb = n_folds < 32768 ? (n_folds > 0 ? n_folds : 1) : 32768;
    return (size_t)nb * fstat_len(K, M) * 8;
  }
  const Geom g = make_geom(K, M, dtype == CVM_F64 ? 8 : 4, !(flags & CVM_RET_XTX));
  const int splits = choose_splits(n_folds, max_fold_rows, g, target_wg(K, M, dtype == CVM_F64 ? 8 : 4));
  const size_t per_fold = (size_t)splits * g.unit_bytes + align_up(fstat_len(K, M) * 8, 256);
  size_t want = per_fold * (size_t)(n_folds > 0 ? n_folds : 1);
  const size_t cap = (size_t)8 << 30;   // beyond 8 GiB walk the folds in batches
  if (want > cap) want = (cap / per_fold > 0 ? cap / per_fold : 1) * per_fold;
  return want;
}

int cvm_fold_update(const void *X, const void *Y, const void *w, const int64_t *idx,
                    const int64_t *offsets, const int64_t *host_offsets, int64_t n_folds, int64_t N,
                    int K, int M, int dtype, unsigned flags, double ddof, double resolution,
                    const void *G, const void *H, const double *gstats, void *out_XTX,
                    void *out_XTY, void *out_muX, void *out_sdX, void *out_muY, void *out_sdY,
                    double *out_fold, void *ws, size_t ws_bytes, void *stream) {
  if (!X || !offsets || !host_offsets || !G || !gstats || !ws)
    return fail(CVM_EINVAL, "cvm_fold_update: null pointer%s");
  if (n_folds < 0 || N < 0 || K <= 0 || M < 0 || (M > 0 && !Y))
    return fail(CVM_EINVAL, "cvm_fold_update: bad shape%s");
  if (!idx && host_offsets[n_folds] > 0) return fail(CVM_EINVAL, "cvm_fold_update: idx is null%s");
  if ((flags & CVM_RET_XTY) && (M == 0 || !H))
    return fail(CVM_EINVAL, "cvm_fold_update: CVM_RET_XTY needs Y and H%s");
  if (n_folds == 0) return CVM_OK;
  if (dtype == CVM_F64)
    return fold_update_impl<double>(X, Y, w, idx, offsets, host_offsets, n_folds, N, K, M, dtype,
                                    flags, ddof, resolution, G, H, gstats, out_XTX, out_XTY, out_muX,
                                    out_sdX, out_muY, out_sdY, out_fold, ws, ws_bytes,
                                    (hipStream_t)stream);
  if (dtype == CVM_F32)
    return fold_update_impl<float>(X, Y, w, idx, offsets, host_offsets, n_folds, N, K, M, dtype,
                                   flags, ddof, resolution, G, H, gstats, out_XTX, out_XTY, out_muX,
                                   out_sdX, out_muY, out_sdY, out_fold, ws, ws_bytes,
                                   (hipStream_t)stream);
  return fail(CVM_EINVAL, "cvm_fold_update: dtype must be CVM_F32 or CVM_F64%s");
}

#ifdef CVM_STAMPS
int cvm_debug_stamps(unsigned long long *host_out) {
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 1024 * 8 * 4));
  return CVM_OK;
}
int cvm_debug_stamps3(unsigned long long *host_out) {
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps3), sizeof(unsigned long long) * 1024 * 8 * 2));
  return CVM_OK;
}
int cvm_debug_stamps2(unsigned long long *host_out) {
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps2), sizeof(unsigned long long) * 1024 * 8 * 4));
  return CVM_OK;
}
#endif

size_t cvm_sweep_workspace_bytes(int64_t n_folds, int64_t max_fold_rows, int K, int M, int dtype) {
  const Geom g = make_geom(K, M, dtype == CVM_F64 ? 8 : 4, 0);
  const int splits = choose_splits(n_folds, max_fold_rows, g, target_wg(K, M, dtype == CVM_F64 ? 8 : 4));
  return ((size_t)splits * g.unit_bytes + align_up(fstat_len(K, M) * 8, 256)) * (size_t)(n_folds > 0 ? n_folds : 1);
}

int cvm_sweep_fit(const void *X, const void *Y, const void *w, const int64_t *idx,
                  const int64_t *offsets, const int64_t *host_offsets, int64_t n_folds, int64_t N,
                  int K, int M, int dtype, void *G, void *H, double *gstats, int32_t *neg_flag,
                  void *ws, size_t ws_bytes, void *stream, int64_t *splits_out) {
  if (!X || !idx || !offsets || !host_offsets || !G || !gstats || !ws)
    return fail(CVM_EINVAL, "cvm_sweep_fit: null pointer%s");
  if (n_folds <= 0 || N <= 0 || K <= 0 || M < 0 || (M > 0 && (!Y || !H)) || (M == 0 && Y))
    return fail(CVM_EINVAL, "cvm_sweep_fit: bad shape%s");
  if (dtype == CVM_F64)
    return sweep_fit_impl<double>(X, Y, w, idx, offsets, host_offsets, n_folds, N, K, M, dtype, G, H,
                                  gstats, neg_flag, ws, ws_bytes, (hipStream_t)stream, splits_out);
  if (dtype == CVM_F32)
    return sweep_fit_impl<float>(X, Y, w, idx, offsets, host_offsets, n_folds, N, K, M, dtype, G, H,
                                 gstats, neg_flag, ws, ws_bytes, (hipStream_t)stream, splits_out);
  return fail(CVM_EINVAL, "cvm_sweep_fit: dtype must be CVM_F32 or CVM_F64%s");
}

int cvm_sweep_folds(const int64_t *offsets, int64_t n_folds, int K, int M, int dtype, unsigned flags,
                    double ddof, double resolution, int weighted, const void *G, const void *H,
                    const double *gstats, void *out_XTX, void *out_XTY, void *out_muX, void *out_sdX,
                    void *out_muY, void *out_sdY, double *out_fold, void *ws, size_t ws_bytes,
                    int64_t splits, void *stream) {
  if (!offsets || !G || !gstats || !ws) return fail(CVM_EINVAL, "cvm_sweep_folds: null pointer%s");
  if (n_folds <= 0 || K <= 0 || M < 0 || splits <= 0) return fail(CVM_EINVAL, "cvm_sweep_folds: bad shape%s");
  if ((flags & CVM_RET_XTY) && (M == 0 || !H))
    return fail(CVM_EINVAL, "cvm_sweep_folds: CVM_RET_XTY needs Y and H%s");
  if (dtype == CVM_F64)
    return sweep_folds_impl<double>(offsets, n_folds, K, M, dtype, flags, ddof, resolution, weighted, G, H,
                                    gstats, out_XTX, out_XTY, out_muX, out_sdX, out_muY, out_sdY, out_fold,
                                    ws, ws_bytes, splits, (hipStream_t)stream);
  if (dtype == CVM_F32)
    return sweep_folds_impl<float>(offsets, n_folds, K, M, dtype, flags, ddof, resolution, weighted, G, H,
                                   gstats, out_XTX, out_XTY, out_muX, out_sdX, out_muY, out_sdY, out_fold,
                                   ws, ws_bytes, splits, (hipStream_t)stream);
  return fail(CVM_EINVAL, "cvm_sweep_folds: dtype must be CVM_F32 or CVM_F64%s");
}

int cvm_timing_enable(int on) {
  g_timing = on != 0;
  g_ntimed = 0;
  return CVM_OK;
}

int cvm_timing_read(double *ms_fit, int64_t *n_fit, double *ms_fold, int64_t *n_fold) {
  double ms[2] = {0, 0};
  int64_t n[2] = {0, 0};
  for (int i = 0; i < g_ntimed; ++i) {
    HIP_OK(hipEventSynchronize(g_timed[i].b));
    float t = 0;
    HIP_OK(hipEventElapsedTime(&t, g_timed[i].a, g_timed[i].b));
    ms[g_timed[i].kind] += t;
    ++n[g_timed[i].kind];
  }
  g_ntimed = 0;
  if (ms_fit) *ms_fit = ms[0];
  if (n_fit) *n_fit = n[0];
  if (ms_fold) *ms_fold = ms[1];
  if (n_fold) *n_fold = n[1];
  return CVM_OK;
}

int cvm_plan_fold(int64_t n_folds, int64_t max_fold_rows, int K, int M, int dtype, unsigned flags,
                  size_t ws_bytes, int64_t *info) {
  if (!info || K <= 0 || M < 0) return fail(CVM_EINVAL, "cvm_plan_fold: bad argument%s");
  Plan p;
  const bool fold_mode = (flags & 0x80000000u) == 0;   // bit 31 set: plan the fit stage
  int rc = make_plan(n_folds, max_fold_rows, K, M, dtype, flags & 0x7fffffffu, ws_bytes, fold_mode, p);
  if (rc != CVM_OK) return fail(rc, "cvm_plan_fold: workspace too small%s");
  info[0] = p.splits;
  info[1] = (int64_t)p.folds_per_batch * p.splits * p.g.nT;
  info[2] = p.g.P;
  info[3] = p.g.nT;
  info[4] = p.folds_per_batch;
  // MFMA instructions issued per 4 rows of one unit (executed work, incl. padding)
  int64_t per4 = 0;
  // a diagonal tile runs 3 blocks of 16; the float64 LDS-DMA kernel skips the strictly-lower
  // tiles of its two diagonal blocks (16 + 10 + 10) unless it finishes folds in its epilogue
  const bool tri = dtype == CVM_F64 && ((size_t)K * 8) % 16 == 0 && M % 2 == 0 && !(fold_mode && p.splits == 1);
  if (!p.g.diag_only) per4 += (int64_t)(p.g.nTiles - p.g.P) * 64 + (int64_t)p.g.P * (tri ? 36 : 48);
  if (M > 0) per4 += (int64_t)p.g.P * p.g.Yc * 16;   // two H waves x 8
  info[5] = per4;
  return CVM_OK;
}

}  // extern "C"
