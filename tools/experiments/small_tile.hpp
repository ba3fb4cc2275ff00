// small_tile.hpp -- part of libcvmhip.so (included by cvmhip.hip inside its anonymous namespace).
// small_tile_kernel (round 4): the XTX update of folds of at most SMALL_ROWS rows as a pure store stream.
//
// What the round-3 tile kernel (small_apply_kernel) spent its time on, per fold and 64 x 64 tile:
// four workgroup barriers, the accumulators written to LDS and read back twice (once for the
// finish, once transposed for the mirrored store), 16 scalar global loads per thread for the rows
// whose wait also waited for the previous fold's stores (one in-order memory counter), and a
// float64 finish of ~9 instructions per element.  It moved 3.6 (float32) / 4.6 (float64) TB/s where a
// plain fill moves 6.2.  This kernel is organised so that a fold costs one barrier and no LDS
// round trip of results:
//
//   * EVERY 64 x 64 tile of the K x K output is a work item (not only the upper triangle) and is
//     computed in the orientation in which the accumulator registers ARE row segments: the fold's
//     rows at the tile's ROWS go in as the MFMA's B operand, the rows at the tile's COLUMNS as its A
//     operand, so lane (j, q) of wave w ends up with output row 16 w + j and, per register quad,
//     16 contiguous bytes of it; the four lanes q = 0..3 of a row make 64 contiguous bytes.  Results
//     go from the accumulators straight to HBM (nontemporal 16-byte stores); nothing is mirrored.
//   * Exact symmetry without a transpose: element (a, b) is sum_i (w_i x_ia) * x_ib in the upper
//     triangle (the reference's WX^T X, cvmatrix.py:1001) and sum_i x_ia * (w_i x_ib) below it -- the
//     same two numbers multiplied (a product commutes) and summed in the same order.  The weight
//     multiplies the operand of whichever side has the smaller tile index; a diagonal tile computes
//     both products (two accumulator sets) and picks per element.
//   * The centring term is one more row of the rank-n update: row n of a fold is the vector of
//     training means with "weight" sw_T, so that the accumulator holds U + sw_T mu mu^T and the
//     finish is  (G - acc) * (isd_a * isd_b)  -- two or three instructions per element, in T.
//   * Operands travel global -> LDS by LDS-DMA (global_load_lds_dwordx4, per-lane source addresses,
//     no registers) in the lane order of the MFMA's A operand, one fold AHEAD: the DMAs of fold f+1
//     are issued at the top of fold f, behind the wait for fold f's own operands.  That wait is
//     `s_waitcnt vmcnt(<stores of one fold>)`: loads and stores share one in-order counter on this
//     hardware, so the only memory operations allowed to be outstanding are the previous fold's
//     stores -- which are never waited for before they are a whole fold old.  There is no
//     compiler-visible vector load inside the fold loop (G lives in registers, everything per fold
//     comes through LDS), so the compiler inserts no wait of its own there.
//   * The row-side (B) operand is read out of the same kind of LDS image as the column-side one (a
//     permuted 4- or 8-byte read), so a tile stages two 64-column slabs per fold (one on the
//     diagonal) and nothing else; the reciprocal standard deviations ride along as one more "row".
//
// small_stats_kernel leaves, per fold, a record with the row numbers, the weights and T-typed copies
// of the means and reciprocal standard deviations (small_rec_layout), so that the prologue of a
// workgroup has one level of dependent loads instead of three (offsets -> indices -> weights).
#pragma once

struct SmallRecLayout { unsigned mu, isd, rows, w, stride; };
__host__ __device__ inline SmallRecLayout small_rec_layout(int K, int es) {
  SmallRecLayout L;
  const unsigned kb = (unsigned)(((size_t)K * es + 15) / 16 * 16);
  L.mu = 16;                               // [0, 4): rows of the fold; [8, 16): sw_T (double)
  L.isd = L.mu + kb;
  L.rows = L.isd + kb;                     // SMALL_ROWS row numbers (int64; -1 past the fold's end)
  L.w = L.rows + SMALL_ROWS * 8;           // SMALL_ROWS weights (T)
  L.stride = (L.w + SMALL_ROWS * 8 + 255) / 256 * 256;
  return L;
}

template <typename T> struct TileCfg;
// V: elements of a 16-byte piece; GRP: groups of V MFMA column tiles (4 column tiles = 64 columns);
// WPE: waves per SIMD the register budget is cut for
template <> struct TileCfg<float> { static constexpr int V = 4, GRP = 1, WPE = 4; };
template <> struct TileCfg<double> { static constexpr int V = 2, GRP = 2, WPE = 3; };
constexpr int TL_FPB = 16;                 // folds per workgroup at most
constexpr int TL_MAXSTEPS = 10;            // k-steps a buffer holds at most: 32 rows + the means -> 9, + the isd step

// all but the n youngest vector-memory operations of this wave are done (n wave-uniform, 0..16)
__device__ __forceinline__ void wait_vmcnt_le(int n) {
#define CVM_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
  switch (n) {
    CVM_W(0) CVM_W(1) CVM_W(2) CVM_W(3) CVM_W(4) CVM_W(5) CVM_W(6) CVM_W(7) CVM_W(8)
    CVM_W(9) CVM_W(10) CVM_W(11) CVM_W(12) CVM_W(13) CVM_W(14) CVM_W(15)
    default: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
  }
#undef CVM_W
}

// one LDS-DMA instruction: lane l copies 16 bytes from its own `src` to LDS byte address lds_addr + 16 l
__device__ __forceinline__ void dma16_lanes(const void *src, unsigned lds_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
               "global_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(src), "s"(lds_addr) : "memory");
}

// The accumulators come out of the MFMA with lane (j, q) holding row j and, in register quad r of
// group g, the 16-byte piece 16 g + 4 r + q of the row: four lanes per row, 64 contiguous bytes per
// store instruction -- and half-written 128-byte lines cost the memory system twice (measured: 3.0 TB/s
// with such stores where the 256-byte segments of the round-3 kernel reach 4.8).  So the 4 x 4 block of
// (lane group u = j / 4, register quad r) is transposed inside every 16-lane row of the wave by two
// rounds of data-parallel-primitive moves (no LDS): afterwards lane (j = 4 u + v, q) holds, in quad k,
// row 4 k + v, piece 4 u + q -- sixteen lanes per row, 256 contiguous bytes, four rows per store.
typedef int tl_i4 __attribute__((ext_vector_type(4)));
// exchange across lanes l and l ^ D of a 16-lane row (D = 8: banks {0,1} <-> {2,3}; D = 4: {0,2} <-> {1,3}):
// the low lane keeps A and gets the high lane's A as its B; the high lane keeps B and gets the low lane's B as its A
template <int D> __device__ __forceinline__ void tl_xchg(tl_i4 &A, tl_i4 &B) {
  constexpr int LO = D == 8 ? 0x3 : 0x5, HI = D == 8 ? 0xC : 0xA;
  constexpr int ROR_UP = 0x120 + D;          // row_ror:D      lane l reads lane l - D  (for the high lanes)
  constexpr int ROR_DN = 0x120 + (16 - D);   // row_ror:16-D   lane l reads lane l + D  (for the low lanes)
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int t = __builtin_amdgcn_update_dpp(0, B[e], ROR_UP, 0xF, HI, false);      // high lanes: the low lane's B
    B[e] = __builtin_amdgcn_update_dpp(B[e], A[e], ROR_DN, 0xF, LO, false);          // low lanes: B = the high lane's A
    A[e] = __builtin_amdgcn_update_dpp(A[e], t, 0xE4, 0xF, HI, false);               // high lanes: A = t
  }
}
template <typename VT> __device__ __forceinline__ void tl_transpose4(VT (&pc)[4]) {
  tl_i4 x[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) x[r] = __builtin_bit_cast(tl_i4, pc[r]);
  tl_xchg<8>(x[0], x[2]); tl_xchg<8>(x[1], x[3]);
  tl_xchg<4>(x[0], x[1]); tl_xchg<4>(x[2], x[3]);
#pragma unroll
  for (int r = 0; r < 4; ++r) pc[r] = __builtin_bit_cast(VT, x[r]);
}

// what a lane knows about its place in the tile AFTER that transpose, and its pieces of G in that layout:
// quad (g, k) = row 16 wave + 4 k + (j & 3), columns 16 V g + V (4 (j >> 2) + q) .. + V - 1 of the tile
template <typename T> struct TileLane {
  static constexpr int V = TileCfg<T>::V, GRP = TileCfg<T>::GRP;
  typedef T vec_t __attribute__((ext_vector_type(V)));
  unsigned okmask, pvmask;      // quads (g, k) this lane / any lane of this wave stores
  vec_t gt[GRP][4];             // the tile of G
};

template <typename T> __device__ __forceinline__ void small_tile_preload(const SmallArgs &a, int a0, int b0, TileLane<T> &tl) {
  constexpr int V = TileCfg<T>::V, GRP = TileCfg<T>::GRP;
  typedef T vec_t __attribute__((ext_vector_type(V)));
  const int K = a.K;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, j = lane & 15, q = lane >> 4;
  const int row0 = a0 + 16 * wave + (j & 3), col0 = b0 + V * (4 * (j >> 2) + q);
  unsigned okmask = 0;
#pragma unroll
  for (int g = 0; g < GRP; ++g)
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (row0 + 4 * k < K && col0 + 16 * V * g < K) okmask |= 1u << (g * 4 + k);
  unsigned pvmask = 0;
#pragma unroll
  for (int p = 0; p < 4 * GRP; ++p)
    if (__ballot((okmask >> p) & 1) != 0ull) pvmask |= 1u << p;
  tl.okmask = okmask;
  tl.pvmask = (unsigned)uni((int)pvmask);
  const T *Gt = (const T *)a.G;
#pragma unroll
  for (int g = 0; g < GRP; ++g)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
      for (int e = 0; e < V; ++e) tl.gt[g][k][e] = (T)0;
      if ((okmask >> (g * 4 + k)) & 1)
        tl.gt[g][k] = *reinterpret_cast<const vec_t *>(Gt + (size_t)(row0 + 4 * k) * K + col0 + 16 * V * g);
    }
}

// MODE 0: tile above the diagonal (the row side carries the weight), 1: below it (the column side does),
// 2: on it (both products, picked per element).  FULL: every piece of every lane lies inside the matrix.
template <typename T, int MODE, bool FULL>
__device__ __forceinline__ void small_tile_folds(const SmallArgs &a, TileLane<T> &tl, char *img,
                                                 const unsigned long long *ptab, const T *wtab, const int *n_all,
                                                 int nf, int f_first, int a0, int b0, int TM, unsigned lds0) {
  constexpr int V = TileCfg<T>::V, GRP = TileCfg<T>::GRP, ES = (int)sizeof(T);
  constexpr bool DIAG = MODE == 2;
  typedef T vec_t __attribute__((ext_vector_type(V)));
  typedef typename MF<T>::acc_t acc_t;
  const int K = a.K;
  const int tid = threadIdx.x, wave = uni(tid >> 6), lane = tid & 63, j = lane & 15, q = lane >> 4;
  const bool cX = a.flags & CVM_CENTER_X, sX = a.flags & CVM_SCALE_X;
  const int NS = 4 * TM;
  // ---- lane constants ------------------------------------------------------------------------
  // column P(i) of the slab that lane (i, kq) of an operand image holds (16 bytes from there on): the
  // permutation that makes the accumulator registers of a lane contiguous in the output row
  const int li = lane & 15;
  const int Pi = ES == 4 ? 4 * (li >> 2) + 16 * (li & 3) : 2 * li;
  // where this lane's ROW-side operand sits in an image of the row slab: column c = 16 wave + j
  const int c = 16 * wave + j;
  const int gq = c / (16 * V), cq = c - gq * 16 * V, m = cq / V, e_ = cq - m * V;
  const int ii = ES == 4 ? 4 * (m & 3) + (m >> 2) : m;
  const unsigned rs_off = (unsigned)(gq * 1024 + (q * 16 + ii) * 16 + e_ * ES);   // k-slot q of a step
  const int row_l = 16 * wave + j;                 // this lane's output row inside the tile, as the MFMA leaves it
  // after the transpose: quad k is row 16 wave + 4 k + (j & 3); where that row's isd sits in the isd step's
  // image of the row slab (k-slot 0), and where this lane's column piece sits in the column slab's
  unsigned isd_r[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c2 = 16 * wave + 4 * k + (j & 3);
    const int g2 = c2 / (16 * V), q2 = c2 - g2 * 16 * V, m2 = q2 / V, e2 = q2 - m2 * V;
    const int i2 = ES == 4 ? 4 * (m2 & 3) + (m2 >> 2) : m2;
    isd_r[k] = (unsigned)(g2 * 1024 + i2 * 16 + e2 * ES);
  }
  const int pi = 4 * (j >> 2) + q;                 // this lane's 16-byte piece of a 16-piece row segment
  const unsigned isd_c = (unsigned)((ES == 4 ? 4 * (pi & 3) + (pi >> 2) : pi) * 16);
  const unsigned okmask = tl.okmask, pvmask = tl.pvmask;
  const int nst = FULL ? 4 * GRP : __builtin_popcount(pvmask);      // store instructions of this wave per fold
  vec_t (&gt)[GRP][4] = tl.gt;
  const char *zero = reinterpret_cast<const char *>(g_zero_line);
  const char *ones = ES == 8 ? reinterpret_cast<const char *>(g_ones_line_d) : reinterpret_cast<const char *>(g_ones_line_f);
  // DMAs of fold fn into buffer bb: instruction d = (step, slab, group) goes to wave d mod 4.  The step
  // after the fold's last k-step holds the reciprocal standard deviations (ones when nothing is scaled).
  auto issue = [&](int fn, int bb) {
    const int n = uni(n_all[fn]);
    const int S = (n + (cX ? 1 : 0) + 3) >> 2;
    constexpr int nsl = DIAG ? 1 : 2;
    const int D = (S + 1) * nsl * GRP;
    for (int d = wave; d < ((a.dbg & 2) ? 0 : D); d += 4) {
      const int g = d % GRP, t2 = d / GRP, sl = t2 % nsl, s = t2 / nsl;
      const unsigned long long p = ptab[fn * NS + 4 * s + q];
      const int col = (sl ? b0 : a0) + 16 * V * g + Pi;
      const bool ok = !(p & 3ull) && col < K;
      // (bit 0: a row of zeros, bit 1: a row of ones -- both without a column offset)
      const char *src = ok ? reinterpret_cast<const char *>(p) + (size_t)col * ES : ((p & 2ull) ? ones : zero) + Pi * ES;
      dma16_lanes(src, lds0 + (unsigned)((((bb * 2 + sl) * TM + s) * GRP + g) << 10));      // (bb: buffer of fold fn)
    }
  };
  // NB operand buffers: fold f's operands are issued NB - 1 folds ahead (two buffers: one fold -- at K = 4096 the
  // DMA's latency under a saturated memory system was just longer than a fold's own work; three: two folds)
  const int NB = a.nbuf;
  for (int f = 0; f < NB - 1 && f < nf; ++f) issue(f, f);
  // every load so far (the tile of G, the tables) has landed before the loop starts: nothing the
  // compiler knows of is in flight there, so it places no vector-memory wait inside the loop
#pragma unroll
  for (int g = 0; g < GRP; ++g)
#pragma unroll
    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(gt[g][r]));
  const size_t obase = (size_t)(a0 + 16 * wave + (j & 3)) * K + b0 + V * pi;
  for (int ff = 0; ff < nf; ++ff) {
    const int b = ff % NB;
    if (!(a.dbg & 8)) {
      // this wave's DMAs of fold ff have landed: they are older than the stores of the last NB - 1 folds (and
      // than the DMAs issued in between, which may be forced to land with them: never more than a fold's)
      // (the folds issued before the loop all landed with the first wait)
      if (ff == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (ff >= NB - 1) wait_vmcnt_le((NB - 1) * nst);
    }
    if (!(a.dbg & 16)) lds_barrier();              // ... and everybody else's; the buffer of fold ff - 1 is free
    if (ff + NB - 1 < nf) issue(ff + NB - 1, (ff + NB - 1) % NB);
    const int n = uni(n_all[ff]);
    const int S = (n + (cX ? 1 : 0) + 3) >> 2;
    const char *imgR = img + (size_t)((b * 2 + 0) * TM) * GRP * 1024;      // (b: this fold's buffer)
    const char *imgC = DIAG ? imgR : img + (size_t)((b * 2 + 1) * TM) * GRP * 1024;
    acc_t acc[4], acc2[DIAG ? 4 : 1];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = (acc_t){0, 0, 0, 0};
    if (DIAG) {
#pragma unroll
      for (int t = 0; t < 4; ++t) acc2[t] = (acc_t){0, 0, 0, 0};
    }
    const T *wt = wtab + ff * NS + q;
    // the operands of step s + 1 are requested before the MFMAs of step s (step S is the isd step: a
    // valid image, read by the last k-step for nothing)
    T wv = wt[0];
    T xb = *reinterpret_cast<const T *>(imgR + rs_off);
    vec_t xa[GRP];
#pragma unroll
    for (int g = 0; g < GRP; ++g) xa[g] = *reinterpret_cast<const vec_t *>(imgC + (size_t)g * 1024 + lane * 16);
    for (int s = 0; s < ((a.dbg & 1) ? 0 : S); ++s) {
      const T wv1 = wt[4 * (s + 1)];
      const T xb1 = *reinterpret_cast<const T *>(imgR + (size_t)(s + 1) * GRP * 1024 + rs_off);
      vec_t xa1[GRP];
#pragma unroll
      for (int g = 0; g < GRP; ++g)
        xa1[g] = *reinterpret_cast<const vec_t *>(imgC + (size_t)((s + 1) * GRP + g) * 1024 + lane * 16);
      if (MODE == 2) {
        const T xbw = wv * xb;
#pragma unroll
        for (int g = 0; g < GRP; ++g)
#pragma unroll
          for (int e = 0; e < V; ++e) {
            acc[g * V + e] = MF<T>::mfma(xa[g][e], xbw, acc[g * V + e]);
            acc2[g * V + e] = MF<T>::mfma(wv * xa[g][e], xb, acc2[g * V + e]);
          }
      } else if (MODE == 0) {
        const T xbw = wv * xb;
#pragma unroll
        for (int g = 0; g < GRP; ++g)
#pragma unroll
          for (int e = 0; e < V; ++e) acc[g * V + e] = MF<T>::mfma(xa[g][e], xbw, acc[g * V + e]);
      } else {
#pragma unroll
        for (int g = 0; g < GRP; ++g)
#pragma unroll
          for (int e = 0; e < V; ++e) acc[g * V + e] = MF<T>::mfma(wv * xa[g][e], xb, acc[g * V + e]);
      }
      wv = wv1; xb = xb1;
#pragma unroll
      for (int g = 0; g < GRP; ++g) xa[g] = xa1[g];
    }
    // ---- finish and store: transpose, (G - acc) * (isd_row * isd_col), 4 rows x 256 bytes per store ----
    T *out = (T *)a.out_XTX + (size_t)(a.seg0 + f_first + ff) * (size_t)K * K + obase;
    T sr[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) sr[k] = *reinterpret_cast<const T *>(imgR + (size_t)S * GRP * 1024 + isd_r[k]);
#pragma unroll
    for (int g = 0; g < GRP; ++g) {
      const vec_t sc = *reinterpret_cast<const vec_t *>(imgC + (size_t)(S * GRP + g) * 1024 + isd_c);
      vec_t pc[4];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < V; ++e) {
          T av = acc[g * V + e][r];
          if (DIAG) av = (row_l <= 16 * V * g + 4 * V * r + V * q + e) ? av : acc2[g * V + e][r];
          pc[r][e] = av;
        }
      if (!(a.dbg & 4)) tl_transpose4(pc);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        vec_t o;
#pragma unroll
        for (int e = 0; e < V; ++e) o[e] = (gt[g][k][e] - pc[k][e]) * (sr[k] * sc[e]);
        const int p = g * 4 + k;
        if (FULL) out_store(reinterpret_cast<vec_t *>(out + (size_t)4 * k * K + 16 * V * g), o);
        else if ((pvmask >> p) & 1) {
          if ((okmask >> p) & 1) out_store(reinterpret_cast<vec_t *>(out + (size_t)4 * k * K + 16 * V * g), o);
        }
      }
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256, TileCfg<T>::WPE) void small_tile_kernel(const SmallArgs a) {
  constexpr int GRP = TileCfg<T>::GRP;
  extern __shared__ __attribute__((aligned(16))) char tl_smem[];
  // XCD-contiguous ranges of (fold group, tile), as in small_apply_kernel
  const unsigned lin = blockIdx.x, tot = (unsigned)a.gx * (unsigned)a.gy;
  const unsigned per = (tot + 7) / 8;
  const unsigned item = (lin & 7) * per + (lin >> 3);
  if (item >= tot) return;
  const int x = (int)(item % (unsigned)a.gx), by = (int)(item / (unsigned)a.gx);
  const int ti = x / a.P64, tj = x - ti * a.P64;
  const int a0 = ti * ST, b0 = tj * ST;
  const int K = a.K;
  const int tid = threadIdx.x;
  const bool cX = a.flags & CVM_CENTER_X, sX = a.flags & CVM_SCALE_X;
  const int TM = a.tsteps, NS = 4 * TM;
  char *img = tl_smem;                                                  // [nbuf buffers][2 slabs][TM][GRP][1 KiB]
  unsigned long long *ptab = reinterpret_cast<unsigned long long *>(tl_smem + (size_t)2 * a.nbuf * TM * GRP * 1024);
  T *wtab = reinterpret_cast<T *>(ptab + (size_t)a.fpb * NS);
  int *n_all = reinterpret_cast<int *>(wtab + (size_t)a.fpb * NS);
  const int f_first = by * a.fpb;
  const int nf = (a.nb - f_first < a.fpb) ? a.nb - f_first : a.fpb;
  TileLane<T> tl;
  small_tile_preload<T>(a, a0, b0, tl);          // (in flight while the table is filled)
  // ---- the row table of this group's folds: where each k-slot's row starts, and its weight ----
  for (int e = tid; e < nf * NS; e += 256) {
    const int ff = e / NS, slot = e - ff * NS;
    const char *rec = a.rec + (size_t)(f_first + ff) * a.rec_stride;
    const int n = *reinterpret_cast<const int *>(rec);
    const int sl = slot < SMALL_ROWS ? slot : SMALL_ROWS - 1;
    const int64_t r = reinterpret_cast<const int64_t *>(rec + a.rec_rows)[sl];
    const T wr = reinterpret_cast<const T *>(rec + a.rec_w)[sl];
    const double swt = *reinterpret_cast<const double *>(rec + 8);
    const int S = (n + (cX ? 1 : 0) + 3) >> 2;
    // (all four loads above are issued whatever the slot turns out to be: one round trip, not two)
    asm volatile("" :: "v"(n), "v"(r), "v"(wr), "v"(swt));
    unsigned long long p = 1ull;                                        // bit 0: a row of zeros
    T wv = (T)0;
    if (slot < n) { p = (unsigned long long)(uintptr_t)((const T *)a.X + r * (int64_t)K); wv = wr; }
    else if (cX && slot == n) { p = (unsigned long long)(uintptr_t)(rec + a.rec_mu); wv = (T)swt; }
    else if (slot == 4 * S) p = sX ? (unsigned long long)(uintptr_t)(rec + a.rec_isd) : 2ull;   // bit 1: a row of ones
    ptab[e] = p;
    wtab[e] = wv;
    if (slot == 0) n_all[ff] = n;
  }
  lds_barrier();
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)tl_smem);
  const bool full = a0 + ST <= K && b0 + ST <= K;
#define CVM_TILE_FOLDS(MODE, FULL) small_tile_folds<T, MODE, FULL>(a, tl, img, ptab, wtab, n_all, nf, f_first, a0, b0, TM, lds0)
  if (full) {
    if (ti < tj) CVM_TILE_FOLDS(0, true);
    else if (ti > tj) CVM_TILE_FOLDS(1, true);
    else CVM_TILE_FOLDS(2, true);
  } else {
    if (ti < tj) CVM_TILE_FOLDS(0, false);
    else if (ti > tj) CVM_TILE_FOLDS(1, false);
    else CVM_TILE_FOLDS(2, false);
  }
#undef CVM_TILE_FOLDS
}

// bytes of dynamic LDS of a small_tile_kernel launch
template <typename T> inline size_t small_tile_lds(int tsteps, int fpb, int nbuf) {
  return (size_t)2 * nbuf * tsteps * TileCfg<T>::GRP * 1024 + (size_t)fpb * 4 * tsteps * (8 + sizeof(T)) + (size_t)fpb * 4 + 16;
}
