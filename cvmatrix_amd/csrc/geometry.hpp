// geometry.hpp -- part of libcvmhip.so (included by cvmhip.hip inside its anonymous namespace).
// Tile geometry, partial-workspace layout and launch arguments shared by the Gram kernels and
// the finalize kernels; wave-uniform helpers.
#pragma once

// ----------------------------------------------------------------------------------
// geometry
// ----------------------------------------------------------------------------------
constexpr int TILE = 128;     // columns per panel; a workgroup owns a TILE x TILE output tile
constexpr int STAGE_ROWS = 16;  // rows staged in LDS per pipeline stage (4 MFMA k-steps)
#ifndef CVM_TWO_LEVEL
#define CVM_TWO_LEVEL 1      // 0: the single-chain float32 accumulation of rounds 1-3 (comparisons)
#endif
#ifndef CVM_FOLD_STAGES
#define CVM_FOLD_STAGES 64
#endif
constexpr int FOLD_STAGES = CVM_FOLD_STAGES;   // float32 Gram kernels: accumulator chains of at most FOLD_STAGES * 16 rows (a power of two)
constexpr int PITCH = 144;    // LDS row pitch of a panel, in elements (see bank note below)
constexpr int YT = 32;        // Y columns handled per diagonal work item (2 MFMA col tiles)
constexpr int YPITCH = 48;    // LDS row pitch of the Y tile, in elements
constexpr int NTHREADS = 512; // 8 waves, two per SIMD
constexpr int PANEL_ELEMS = STAGE_ROWS * PITCH;             // 2304
constexpr int BUF_ELEMS = 2 * PANEL_ELEMS + STAGE_ROWS;     // A panel, B panel | Y tile, w
constexpr int TARGET_WG_1 = 256;  // resident workgroups (both Gram kernels: one 8-wave workgroup per CU)
constexpr int TARGET_WG_2 = 256;
constexpr int QUEUE_STRIDE = 32;       // unsigneds between the 8 queue heads (one 128-byte line each)
constexpr size_t QUEUE_BYTES = 8 * QUEUE_STRIDE * sizeof(unsigned);
constexpr int QUEUE_DONE = 8;          // index of the exit counter (in the first head's line)
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

// Work items of a Gram launch.  A segment (a fold, or all rows in the fit stage) is cut into row
// splits; unit u = seg * splits + sp owns one slot of the partial workspace.  The items of a unit
// come in two CLASSES with their own split counts, because they cost differently per row:
//   class 0  off-diagonal 128x128 tiles (i < j): 16 MFMAs per wave and k-step; rows cut s_off ways
//   class 1  diagonal tiles with the first Y chunk (9 + 2 MFMAs per wave and k-step in the LDS-DMA
//            kernel, they also produce XTY and the column sums) and the further Y chunks;
//            rows cut s_diag ways
// `splits` = max(s_off, s_diag) is the slot stride; class-0 partials exist for sp < s_off, class-1
// partials (diagonal tiles, XTY panels, column sums) for sp < s_diag.  The grid lists all class-0
// items first, then all class-1 items (longest first: the hardware hands workgroups to free CUs in
// order), each list spread over the 8 XCDs in contiguous ranges.
template <typename T> struct WgramArgs {
  const T *X, *Y, *w;
  const int64_t *idx;   // nullptr: rows are offs[seg]..offs[seg+1] themselves
  const int64_t *offs;  // device; nullptr: one segment [0, N)
  int64_t N;
  int64_t seg0;         // first segment of this batch
  int n_seg, splits;    // splits: slot stride = max(s_off, s_diag)
  int s_off, s_diag;
  Geom g;
  long n_items0, ipx0;  // class 0: items, items per XCD
  long n_items1, ipx1;  // class 1
  unsigned *queue;      // wgram4_kernel: 8 work-queue heads (one per XCD, QUEUE_STRIDE apart) and the exit
                        // counter, all zero at launch; the last workgroup to leave zeroes them again
  char *ws;             // unit u at ws + u*unit_bytes
  // fused single-split fold update (wgram4_kernel<.., FUSED>): finish in the epilogue
  const double *fstats; // per fold of the batch: means / stds / sw_train (fold_stats_kernel)
  const void *G, *H;    // full-data matrices
  void *out_XTX, *out_XTY;
  unsigned flags;
  // fused route with the statistics formed INSIDE the launch (round 4; stat_flags != nullptr): the diagonal
  // item of (fold, panel) sums the panel's columns while it streams the fold's rows anyway, derives the
  // training means / reciprocal stds of its 128 columns (and of Y, and the fold's totals), writes them to
  // fstats with device-coherent stores and raises stat_flags[fold * P + panel]; an off-diagonal item polls
  // the flags of its two panels before its epilogue.  The diagonal items come FIRST in every XCD's list
  // (diag_first), they wait for nobody: no deadlock.  No colstats_kernel / fold_stats_kernel pre-pass.
  int *stat_flags;
  int diag_first;       // list order of an XCD: 0 = class-0 items then class-1 items, 1 = class 1 first, 2 = FOLD-MAJOR
                        // (whole folds per XCD; per fold its class-1 items, then its class-0 items: see decode_slot)
  int fpx;              // (fold-major) folds per XCD's list
  // The wait of an off-diagonal item for its two flags is bounded (fused_wait_flag).  An item whose wait gave up
  // writes nothing and appends its list position to retry_items (count in fused_status[0]); the host then runs the
  // kernel once more over that list (retry_mode = 1: every diagonal item of the first launch has finished, all flags
  // are up).  A wait that gives up in THAT launch counts in fused_status[1] and poisons the item's outputs with NaN.
  // The last workgroup of the retry launch folds both into *status_out (1 = poisoned outputs, 2 = items recomputed).
  int *fused_status;
  unsigned long long *retry_items;
  int retry_mode;
  int test_mode;        // CVM_FUSED_TEST_TIMEOUT (tests): 1 = off-diagonal items of every third fold give up at once,
                        // 2 = a spin limit of a few polls, 3 = like 1 and the retry launch gives up too
  int32_t *status_out;
  const double *gstats;
  double ddof, resolution;
  void *out_muX, *out_sdX, *out_muY, *out_sdY;
  double *out_fold;
  int dbg;              // diagnostic ablations (env CVM_DEBUG): 1 no global loads after the
                        // first stage, 2 no MFMA, 4 no VALU column sums; results are wrong
  // clock probe of the PRODUCT kernel (cvm_clock_probe; nullptr = off, the default): workgroup b < clock_wgs stores
  // {s_memtime, s_memrealtime} once when it starts and once when it has run out of work items -- two scalar clock
  // reads per workgroup LIFETIME, nothing inside the item loop -- so that the shader clock the chip holds under
  // this very kernel is (d memtime / d memrealtime) x 100 MHz.  The buffer is the caller's and nothing else in the
  // library reads it.
  unsigned long long *clock_stamps;
  int clock_wgs;
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

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope release +
// acquire: the compiler puts s_waitcnt vmcnt(0) in front of the barrier, so every global store the
// wave has issued must be ACKNOWLEDGED by the memory system first -- with output stores streaming
// to HBM that is several microseconds per barrier (tools/exp_small_apply.py: the K = 4096 small-fold
// kernel spent ~7 of its ~9.5 us per fold and tile in two such waits).  The kernels below only
// hand LDS contents from wave to wave at these points; global results are never read back by
// another wave of the launch.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Statistics another workgroup of the SAME launch may have written (WgramArgs::stat_flags): device-coherent
// loads and stores (sc1: served by L2, never by this CU's L1), the hand-off form of MI355X_MICROARCH.md's
// table "stores all sc1, loads all sc1, one lane signals behind every storing wave's vmcnt(0) + barrier"
__device__ __forceinline__ double ldc(const double *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void stc(double *p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ long long uni64(long long v) {
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v & 0xffffffffll));
  const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)v >> 32));
  return (long long)(((unsigned long long)hi << 32) | lo);
}
template <typename P> __device__ __forceinline__ P *unip(P *p) { return (P *)uni64((long long)p); }
// The launch arguments of the Gram kernels, read where they are needed: the out-of-line wave-role
// functions receive the ADDRESS of the kernel's one argument (a WgramArgs<T> at offset 0 of the
// kernarg segment, constant address space) and load the fields with scalar loads into SGPRs.
// (Passing the struct itself to a __noinline__ function by reference made the kernel spill it to
// 224-232 bytes of private memory per lane and every role reload it with flat loads.  The kernarg
// pointer intrinsic itself only works inside the kernel function: a callee gets null.)
template <typename T> using kargs_ptr = const __attribute__((address_space(4))) WgramArgs<T> *;
template <typename T> __device__ __forceinline__ kargs_ptr<T> kernarg_address() {
  return (kargs_ptr<T>)__builtin_amdgcn_kernarg_segment_ptr();
}
template <typename T> __device__ __forceinline__ WgramArgs<T> kernel_args(kargs_ptr<T> rv) {
  // the pointer arrives in vector registers (function argument): make it scalar again
  kargs_ptr<T> r = (kargs_ptr<T>)(unsigned long long)uni64((long long)(unsigned long long)rv);
  WgramArgs<T> a;
  a.X = r->X; a.Y = r->Y; a.w = r->w; a.idx = r->idx; a.offs = r->offs;
  a.N = r->N; a.seg0 = r->seg0; a.n_seg = r->n_seg; a.splits = r->splits;
  a.s_off = r->s_off; a.s_diag = r->s_diag;
  a.g.K = r->g.K; a.g.M = r->g.M; a.g.P = r->g.P; a.g.Kp = r->g.Kp;
  a.g.Yc = r->g.Yc; a.g.Mp = r->g.Mp; a.g.nTiles = r->g.nTiles; a.g.nT = r->g.nT;
  a.g.diag_only = r->g.diag_only;
  a.g.tile_elems = r->g.tile_elems; a.g.h_elems = r->g.h_elems;
  a.g.stat_len = r->g.stat_len; a.g.unit_bytes = r->g.unit_bytes;
  a.n_items0 = r->n_items0; a.ipx0 = r->ipx0; a.n_items1 = r->n_items1; a.ipx1 = r->ipx1;
  a.queue = r->queue;
  a.ws = r->ws; a.dbg = r->dbg;
  a.fstats = r->fstats; a.G = r->G; a.H = r->H;
  a.out_XTX = r->out_XTX; a.out_XTY = r->out_XTY; a.flags = r->flags;
  a.stat_flags = r->stat_flags; a.diag_first = r->diag_first; a.gstats = r->gstats;
  a.fpx = r->fpx; a.fused_status = r->fused_status; a.retry_items = r->retry_items;
  a.retry_mode = r->retry_mode; a.test_mode = r->test_mode; a.status_out = r->status_out;
  a.ddof = r->ddof; a.resolution = r->resolution;
  a.out_muX = r->out_muX; a.out_sdX = r->out_sdX; a.out_muY = r->out_muY; a.out_sdY = r->out_sdY;
  a.out_fold = r->out_fold;
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

// k-th strictly-upper tile in row-major order: (0,1),(0,2)..(0,P-1),(1,2)...
__device__ __forceinline__ void decode_off_tile(int k, int P, int &ti, int &tj) {
  int i = 0, rem = k;
  while (rem >= P - 1 - i) { rem -= P - 1 - i; ++i; }
  ti = i; tj = i + 1 + rem;
}

// Which work item a workgroup owns (see WgramArgs).  Returns false for the padding workgroups of
// the grid.  `it`: the item's tile slot in the unit's partials (G tiles), `nsp`: the split count
// of its class.
struct Item { int seg, sp, nsp, it, ti, tj, yc; long u; };
// Position q of XCD `xcd`'s list: its class-0 items [0, ipx0), then its class-1 items.
template <typename T> __device__ __forceinline__ bool decode_slot(const WgramArgs<T> &a, int xcd, long q, Item &o) {
  const Geom &g = a.g;
  long item, cu;   // cu: unit number within the class (seg * nsp + sp)
  int k;
  if (a.diag_first == 2) {
    // FOLD-MAJOR lists (one unit per fold: the fused route): XCD x owns the folds [x fpx, (x + 1) fpx) whole; per
    // fold first its class-1 items (the diagonal tiles: they publish the fold's statistics and wait for nobody),
    // then its class-0 items.  Every item an off-diagonal item waits for precedes it in ITS OWN list, so it was
    // taken -- and is being computed, or done -- before the waiter was; and the fold's rows are gathered by all
    // its items while they are in that XCD's L2.
    const int per = g.nT;
    const long fl = q / per;
    k = (int)(q - fl * per);
    const long sg = (long)xcd * a.fpx + fl;
    if (fl >= a.fpx || sg >= a.n_seg) return false;
    const int n1 = g.P * g.Yc;
    if (k < n1) {
      if (g.diag_only) { o.ti = o.tj = k / g.Yc; o.yc = k - o.ti * g.Yc; }
      else if (k < g.P) { o.ti = o.tj = k; o.yc = 0; }
      else { const int e = k - g.P; o.ti = o.tj = e / (g.Yc - 1); o.yc = 1 + e - o.ti * (g.Yc - 1); }
      o.it = tile_id(o.ti, o.ti, g.P);
    } else {
      decode_off_tile(k - n1, g.P, o.ti, o.tj);
      o.yc = 0;
      o.it = tile_id(o.ti, o.tj, g.P);
    }
    o.seg = uni((int)sg); o.sp = 0; o.nsp = 1; o.it = uni(o.it);
    o.ti = uni(o.ti); o.tj = uni(o.tj); o.yc = uni(o.yc);
    o.u = (long)o.seg * a.splits;
    return true;
  }
  // (diag_first: the list of an XCD is its class-1 items, then its class-0 items)
  if (a.diag_first) q = q < a.ipx1 ? q + a.ipx0 : q - a.ipx1;
  if (q < a.ipx0) {
    item = (long)xcd * a.ipx0 + q;
    if (item >= a.n_items0) return false;
    const int per = g.nTiles - g.P;
    cu = item / per; k = (int)(item - cu * per);
    o.nsp = a.s_off;
    decode_off_tile(k, g.P, o.ti, o.tj);
    o.yc = 0;
    o.it = tile_id(o.ti, o.tj, g.P);
  } else {
    const long q1 = q - a.ipx0;
    if (q1 >= a.ipx1) return false;
    item = (long)xcd * a.ipx1 + q1;
    if (item >= a.n_items1) return false;
    const int per = g.P * g.Yc;                 // P first-chunk items, then P * (Yc - 1) further chunks
    cu = item / per; k = (int)(item - cu * per);
    o.nsp = a.s_diag;
    if (g.diag_only) { o.ti = o.tj = k / g.Yc; o.yc = k - o.ti * g.Yc; }
    else if (k < g.P) { o.ti = o.tj = k; o.yc = 0; }
    else { const int e = k - g.P; o.ti = o.tj = e / (g.Yc - 1); o.yc = 1 + e - o.ti * (g.Yc - 1); }
    o.it = tile_id(o.ti, o.ti, g.P);
  }
  o.seg = (int)(cu / o.nsp);
  o.sp = (int)(cu - (long)o.seg * o.nsp);
  // wave-uniform by construction (the 64-bit divisions run on the VALU): say so, the loader waves
  // want every address in scalar registers
  o.seg = uni(o.seg); o.sp = uni(o.sp); o.nsp = uni(o.nsp); o.it = uni(o.it);
  o.ti = uni(o.ti); o.tj = uni(o.tj); o.yc = uni(o.yc);
  o.u = (long)o.seg * a.splits + o.sp;
  return true;
}
// Static launch (wgram_kernel, one workgroup per item): the grid lists the class-0 positions of
// all XCDs, then the class-1 positions; the hardware deals workgroups to the XCDs round-robin.
template <typename T> __device__ __forceinline__ bool decode_item(const WgramArgs<T> &a, long b, Item &o) {
  const long b0 = 8 * a.ipx0;
  if (b < b0) return decode_slot(a, (int)(b & 7), b >> 3, o);
  const long bb = b - b0;
  return decode_slot(a, (int)(bb & 7), a.ipx0 + (bb >> 3), o);
}

// rows of segment `seg` handled by split `sp`
__device__ __forceinline__ void split_range(int64_t n, int splits, int sp, int64_t &r0, int64_t &r1) {
  int64_t per = (n + splits - 1) / splits;
  per = (per + STAGE_ROWS - 1) / STAGE_ROWS * STAGE_ROWS;
  r0 = (int64_t)sp * per; if (r0 > n) r0 = n;
  r1 = r0 + per; if (r1 > n) r1 = n;
}

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
