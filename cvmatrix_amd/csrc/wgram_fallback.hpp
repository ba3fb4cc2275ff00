// wgram_fallback.hpp -- part of libcvmhip.so (included by cvmhip.hip inside its anonymous namespace).
// wgram_kernel: the general Gram kernel (float32, odd K or M, unaligned rows).
#pragma once

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
__device__ float g_one_line_f[4] = {1.0f, 1.0f, 1.0f, 1.0f};
// 1 KiB of ones per element type (small_tile_kernel: the "reciprocal standard deviations" of a call that scales nothing)
#define CVM_R8(x) x, x, x, x, x, x, x, x
__device__ double g_ones_line_d[128] = {CVM_R8(CVM_R8(1.0)), CVM_R8(CVM_R8(1.0))};
__device__ float g_ones_line_f[256] = {CVM_R8(CVM_R8(1.0f)), CVM_R8(CVM_R8(1.0f)), CVM_R8(CVM_R8(1.0f)), CVM_R8(CVM_R8(1.0f))};
#undef CVM_R8
#ifdef CVM_STAMPS
// diagnostic build only: per (workgroup, wave) cycle sums of the three phases of a stage
__device__ unsigned long long g_stamps[1024 * 8 * 4];
__device__ unsigned long long g_stamps2[1024 * 8 * 4];
__device__ unsigned long long g_stamps4[1024 * 8];   // fused off-diagonal epilogue, wave 0: 8 time points
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
  Item wi;
  if (!decode_item(a, (long)blockIdx.x, wi)) return;
  const long u = wi.u;
  const int it = wi.it, seg = wi.seg, sp = wi.sp, ti = wi.ti, tj = wi.tj, yc = wi.yc;
  const bool diag = (ti == tj);
  const int wr = wave >> 2, wc = wave & 3;
  const bool h_wave = diag && wr == 1 && wc < 2;
  const bool do_g = !g.diag_only && yc == 0;       // the G tile of this item is wanted

  int64_t seg_begin, seg_rows;
  if (a.offs) { seg_begin = a.offs[a.seg0 + seg]; seg_rows = a.offs[a.seg0 + seg + 1] - seg_begin; }
  else { seg_begin = 0; seg_rows = a.N; }
  int64_t r0, r1;
  split_range(seg_rows, wi.nsp, sp, r0, r1);
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
  constexpr bool TWO_LEVEL = CVM_TWO_LEVEL && sizeof(T) == 4;      // float32: two-level sums, see wgram4_body (wgram4.hpp)
  acc_t acc2[TWO_LEVEL ? 8 : 1];
  if (TWO_LEVEL) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc2[i] = (acc_t){0, 0, 0, 0};
  }
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
  // (Round 1 gave the second-dispatched half of the workgroup a static priority of 1 against the
  //  arbitration advantage of the older half, MI355X_MICROARCH "Two waves per SIMD" item 4; measured
  //  again in round 2 next to the LDS-DMA kernel's priorities: 1-2 % faster without it,
  //  tools/bench_odd_k.py K=511 0.808 -> 0.797 ms, K=510 / M=15 0.719 -> 0.704 ms.)

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
    if constexpr (TWO_LEVEL) {
      if (MFM && (s & (FOLD_STAGES - 1)) == FOLD_STAGES - 1) {
#pragma unroll
        for (int i = 0; i < 8; ++i) { acc2[i] += acc[i]; acc[i] = (acc_t){0, 0, 0, 0}; }
      }
    }
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
  if constexpr (TWO_LEVEL) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = acc2[i] + acc[i];
  }
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
