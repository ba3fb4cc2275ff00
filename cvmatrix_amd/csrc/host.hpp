// host.hpp -- part of libcvmhip.so (included by cvmhip.hip inside its anonymous namespace).
// Host side: planning (row splits, batches), launches, the implementation of every C entry point.
#pragma once

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
// One process-wide recorder (cvm_timing_enable / cvm_timing_read): a pair of events around every
// Gram launch, recorded on that launch's own stream.  Calls from several threads may record
// concurrently: slots are claimed under a mutex and the launch kind travels with the call, not in
// a global.
struct TimedLaunch { hipEvent_t a, b; int kind; };
constexpr int MAX_TIMED = 8192;
std::atomic<bool> g_timing{false};
std::mutex g_timing_mu;
TimedLaunch g_timed[MAX_TIMED];
int g_ntimed = 0;        // slots handed out (guarded by g_timing_mu)
// kinds: 0 Gram launch of the fit stage, 1 of the fold stage, 2 small_stats_kernel, 3 the small-fold update kernels
inline TimedLaunch *timed_begin(int kind, hipStream_t st) {
  if (!g_timing.load(std::memory_order_relaxed)) return nullptr;
  TimedLaunch *tl = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_timing_mu);
    if (g_ntimed >= MAX_TIMED) return nullptr;
    tl = &g_timed[g_ntimed++];
    if (!tl->a) {
      if (hipEventCreate(&tl->a) != hipSuccess || hipEventCreate(&tl->b) != hipSuccess) { --g_ntimed; return nullptr; }
    }
    tl->kind = kind;
  }
  if (hipEventRecord(tl->a, st) != hipSuccess) return nullptr;
  return tl;
}
inline void timed_end(TimedLaunch *tl, hipStream_t st) { if (tl) (void)hipEventRecord(tl->b, st); }
enum { KIND_FIT = 0, KIND_FOLD = 1 };

// ---- optional clock probe of the product Gram kernel (cvm_clock_probe) -------------------
// A caller-owned device buffer; while one is set every LDS-DMA Gram launch of the process hands it to the
// kernel (WgramArgs::clock_stamps).  Two words, read together by the launcher.
std::atomic<unsigned long long *> g_clock_buf{nullptr};
std::atomic<int> g_clock_wgs{0};

// once-per-device flags of hipFuncSetAttribute: a bit mask updated atomically (setting the
// attribute twice from two racing threads is harmless, a torn read-modify-write is not)
inline bool attr_needed(std::atomic<unsigned long long> &done, int dev) {
  return !((done.load(std::memory_order_acquire) >> (dev & 63)) & 1ull);
}
inline void attr_set(std::atomic<unsigned long long> &done, int dev) {
  done.fetch_or(1ull << (dev & 63), std::memory_order_release);
}

struct Plan {
  Geom g;
  int splits;            // slot stride of the partial workspace = max(s_off, s_diag)
  int s_off, s_diag;     // row splits of the off-diagonal tiles / of the diagonal-class items
  int64_t folds_per_batch;
  size_t fstat_bytes_per_fold;
};

// which Gram kernel variant a problem gets (pointers from torch are 256-byte aligned; a
// misaligned X falls back to the register path at launch: same plan, it only balances less well)
bool fast_shape(int K, int M, int esize) {
  return ((size_t)K * esize) % 16 == 0 && (esize == 4 || M % 2 == 0);
}

// ---- row splits ---------------------------------------------------------------------------
// The launch is a list of work items (geometry.hpp: all off-diagonal-tile items, then the
// diagonal-class items) that TARGET_WG persistent workgroups (one 8-wave workgroup per CU) take in
// order, each as soon as it is free (wgram4_kernel's work queues).  An item costs its 16-row
// stages times the per-stage cost of its kind plus a fixed prologue/epilogue of about one stage.
// The constants are fitted to measured launch times of 23 split pairs at the C3 shape (11 of the
// 10-fold sweep, 12 of the fit stage; tools/exp_splits.sh, tools/exp_fit_splits.py): the simulation
// below reproduces them to 1.9 % rms with 2.05 us per off-diagonal stage (1.93 us since the loader
// waves stopped outranking the compute waves and the k-step's reads follow its MFMAs one by one,
// wgram4.hpp) and a diagonal tile costing 0.78 of it -- more than its 11/16 share of the MFMAs: it reads almost twice the LDS fragments per
// MFMA, and those reads are what the loop pays for beyond the matrix instructions.  Per-stage costs, in units of an off-diagonal tile's stage (16 MFMAs per wave and
// k-step): a diagonal tile in the LDS-DMA kernel issues 9 + 2*NBY (the upper triangle of its 8x8
// grid of MFMA tiles shared out evenly + XTY), a further-Y-chunk item keeps one wave busy with 16;
// in the general kernel every wave of every item runs the same loop.
// Because the two classes cost differently they get their own split counts: at C3 (10 folds x
// (6 off-diagonal + 4 diagonal tiles)) 4 and 7 give 240 long + 280 short items that pack 256 CUs
// to 94 % (measured: 0.467 ms against 0.507 ms with 5 and 5) -- any common count leaves a second
// round of workgroups a quarter empty.
// The estimate is a simulation of that in-order hand-out, evaluated for a few hundred
// (s_off, s_diag) candidates and cached per shape.
struct SplitKey {
  int64_t n_seg, max_rows; int K, M, esize, diag_only, target; int64_t cap;
  bool operator<(const SplitKey &o) const {
    return std::tie(n_seg, max_rows, K, M, esize, diag_only, target, cap) <
           std::tie(o.n_seg, o.max_rows, o.K, o.M, o.esize, o.diag_only, o.target, o.cap);
  }
};
struct SplitChoice { int s_off, s_diag; };

// `lists` = 8: the kernel's own hand-out (wgram4_kernel): 8 lists (one per XCD: its share of the
// class-0 items, then of the class-1 items), target/8 workgroups per list, each takes the next
// item of its list when it is free and helps the following lists once its own is empty.
// `lists` = 1: one XCD's list and workgroups alone (an eighth of the work: the coarse pass).
double simulate_launch(int64_t n_seg, int64_t max_rows, const Geom &g, int s_off, int s_diag, double c_diag,
                       int target, int lists) {
  const int nOff = g.diag_only ? 0 : g.nTiles - g.P;
  const int nFirst = g.diag_only ? 0 : g.P;                 // diagonal tiles with the first Y chunk
  const int nY = g.diag_only ? g.P * g.Yc : g.P * (g.Yc - 1); // XTY-only items (one busy wave, 16 MFMAs)
  auto stages_of = [&](int s) {
    int64_t per = (max_rows + s - 1) / s;
    return (double)((per + STAGE_ROWS - 1) / STAGE_ROWS);
  };
  const double fixed = 1.0;
  const double c_off = stages_of(s_off) + fixed;
  const double c_first = stages_of(s_diag) * c_diag + fixed, c_y = stages_of(s_diag) + fixed;
  const int64_t n0 = n_seg * s_off * nOff, per1 = nFirst + nY, n1 = n_seg * s_diag * per1;
  const int64_t ipx0 = (n0 + 7) / 8, ipx1 = (n1 + 7) / 8;
  int64_t pos[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  auto next_cost = [&](int x, double &cost) -> bool {
    if (pos[x] < ipx0 && (int64_t)x * ipx0 + pos[x] >= n0) pos[x] = ipx0;   // (a shorter last list)
    const int64_t q = pos[x];
    if (q < ipx0) {
      cost = c_off;
    } else {
      const int64_t item = (int64_t)x * ipx1 + (q - ipx0);
      if (q - ipx0 >= ipx1 || item >= n1) return false;
      cost = (item % per1) < nFirst ? c_first : c_y;
    }
    ++pos[x];
    return true;
  };
  const int workers = lists == 8 ? target : (target + 7) / 8;
  // a small binary heap of the workgroups' free times (workgroup number in the low bits)
  std::vector<std::pair<double, int>> heap;
  heap.reserve((size_t)workers);
  for (int i = 0; i < workers; ++i) heap.emplace_back(0.0, i);
  auto cmp = [](const std::pair<double, int> &a, const std::pair<double, int> &b) { return a > b; };
  double t_end = 0;
  while (!heap.empty()) {
    std::pop_heap(heap.begin(), heap.end(), cmp);
    std::pair<double, int> e = heap.back();
    const int home = lists == 8 ? (e.second & 7) : 0;
    double cost = 0;
    bool got = false;
    for (int d = 0; d < lists && !got; ++d) got = next_cost((home + d) & 7, cost);
    if (!got) { if (e.first > t_end) t_end = e.first; heap.pop_back(); continue; }
    heap.back().first = e.first + cost;
    std::push_heap(heap.begin(), heap.end(), cmp);
  }
  return t_end;
}

// A forced row-split plan (experiments, tests): one atomic word (s_off << 16 | s_diag, 0 = the planner decides),
// set by cvm_debug_force_splits; the environment variable CVM_FORCE_SPLITS="s_off,s_diag" is read ONCE, as its
// initial value (getenv beside another thread's setenv is a data race in glibc: nothing here calls it per launch).
std::atomic<unsigned> g_force_splits{0xffffffffu};       // 0xffffffff: environment not consulted yet
inline unsigned forced_splits() {
  unsigned v = g_force_splits.load(std::memory_order_relaxed);
  if (v != 0xffffffffu) return v;
  unsigned init = 0;
  if (const char *force = getenv("CVM_FORCE_SPLITS")) {
    int so = 0, sd = 0;
    if (sscanf(force, "%d,%d", &so, &sd) == 2 && so >= 1 && sd >= 1 && so < 65536 && sd < 65536) init = ((unsigned)so << 16) | (unsigned)sd;
  }
  unsigned expect = 0xffffffffu;
  g_force_splits.compare_exchange_strong(expect, init, std::memory_order_relaxed);
  return g_force_splits.load(std::memory_order_relaxed);
}

SplitChoice choose_splits2(int64_t n_seg, int64_t max_rows, const Geom &g, int esize, int target) {
  if (n_seg < 1) n_seg = 1;
  // plan for a rounded-up row count (1/16 steps of the leading power of two): ragged folds asked
  // for one call at a time would otherwise each pay for a plan of their own
  {
    int64_t step = 1;
    while (step * 32 <= max_rows) step *= 2;
    max_rows = (max_rows + step - 1) / step * step;
  }
  int64_t cap = max_rows / 64;                       // >= 64 rows per split
  const int64_t mem_cap = (int64_t)(((size_t)3 << 30) / ((size_t)n_seg * g.unit_bytes));
  if (cap > mem_cap) cap = mem_cap;
  if (cap > 128) cap = 128;
  if (cap < 1) cap = 1;
  // a forced plan (cvm_debug_force_splits / CVM_FORCE_SPLITS: experiments and tests), clamped to the caps
  const unsigned force = forced_splits();
  if (force && !g.diag_only && g.nTiles > g.P) {
    const int so = (int)(force >> 16), sd = (int)(force & 0xffffu);
    return SplitChoice{(int)(so > cap ? cap : so), (int)(sd > cap ? cap : sd)};
  }
  static std::mutex mu;
  static std::map<SplitKey, SplitChoice> cache;
  const SplitKey key{n_seg, max_rows, g.K, g.M, esize, g.diag_only, target, cap};
  {
    std::lock_guard<std::mutex> lk(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
  }
  const bool fast = fast_shape(g.K, g.M, esize);
  // (9 + 2 NBY MFMAs of 16 per k-step, times the measured 0.78 / 0.6875 for their LDS reads)
  double c_diag = fast ? (9.0 + 2.0 * (g.M > 16 ? 2 : 1)) / 16.0 * (0.78 / 0.6875) : 1.0;
  if (c_diag > 1.0) c_diag = 1.0;
  const int64_t items1 = n_seg * (int64_t)g.nT;
  const int nOff = g.diag_only ? 0 : g.nTiles - g.P;
  // what extra splits cost after the launch: their partial tiles are read back by the finalize
  // kernels (about 1.5 times on average: the sweep reads them twice) at ~3.5 TB/s; in stages of
  // 2.05 us, shared by all CUs
  const double tile_bytes = (double)TILE * TILE * esize;
  auto partial_cost = [&](int so, int sd) {
    const double bytes = (double)n_seg * ((double)so * nOff * tile_bytes +
                                          (double)sd * ((double)g.P * tile_bytes + (double)g.h_elems * esize));
    return 1.5 * bytes / 3.5e12 / 2.05e-6;
  };
  struct Cand { double score; int so, sd; };
  std::vector<Cand> cands;
  auto consider = [&](int so, int sd) {
    if (so < 1 || sd < 1 || so > cap || sd > cap) return;
    const double t = simulate_launch(n_seg, max_rows, g, so, sd, c_diag, target, 1);
    cands.push_back(Cand{t + partial_cost(so, sd), so, sd});
  };
  if (items1 >= 8 * (int64_t)target || g.diag_only || g.nTiles == g.P) {
    // many rounds of workgroups (or a single class): the tail is a small part of the launch
    for (int sx = 1; sx <= cap && sx <= 8; ++sx) consider(sx, sx);
  } else {
    static const double ratios[] = {0.6, 0.7, 0.75, 0.875, 1.0, 1.25, 1.5, 1.75, 2.0};
    for (int so = 1; so <= cap; ++so) {
      if ((int64_t)so * n_seg * nOff > 16 * (int64_t)target && so > 1) break;   // more rounds than pay
      int last = 0;
      for (double r : ratios) {
        int sd = (int)(so * r + 0.5);
        if (sd < 1) sd = 1;
        if (sd == last) continue;
        last = sd;
        consider(so, sd);
      }
    }
  }
  // the coarse pass looked at one XCD alone; the best few get the whole machine, with stealing
  std::sort(cands.begin(), cands.end(), [](const Cand &a, const Cand &b) {
    return a.score != b.score ? a.score < b.score : (a.so + a.sd) < (b.so + b.sd);
  });
  SplitChoice best{1, 1};
  double best_score = 1e300;
  for (size_t i = 0; i < cands.size() && i < 6; ++i) {
    const Cand &c = cands[i];
    const double score = simulate_launch(n_seg, max_rows, g, c.so, c.sd, c_diag, target, 8) + partial_cost(c.so, c.sd);
    // fewest splits on near-ties: one split per fold lets the float64 kernel finish folds in its epilogue
    if (score < best_score * (1.0 - 1e-3)) { best_score = score; best = SplitChoice{c.so, c.sd}; }
  }
  std::lock_guard<std::mutex> lk(mu);
  if (cache.size() > 4096) cache.clear();
  cache[key] = best;
  return best;
}

int target_wg(int K, int M, int esize) { return fast_shape(K, M, esize) ? TARGET_WG_2 : TARGET_WG_1; }

// partial slots per segment the plan of a problem needs (workspace sizing)
int plan_stride(int64_t n_seg, int64_t max_rows, const Geom &g, int esize) {
  const SplitChoice c = choose_splits2(n_seg, max_rows, g, esize, target_wg(g.K, g.M, esize));
  return c.s_off > c.s_diag ? c.s_off : c.s_diag;
}

void set_plan_splits(Plan &p, int s_off, int s_diag) {
  p.s_off = s_off; p.s_diag = s_diag;
  p.splits = s_off > s_diag ? s_off : s_diag;
}

int make_plan(int64_t n_folds, int64_t max_rows, int K, int M, int dtype, unsigned flags,
              size_t ws_bytes, bool fold_mode, Plan &p) {
  const int esize = dtype == CVM_F64 ? 8 : 4;
  const int diag_only = fold_mode && !(flags & CVM_RET_XTX);
  p.g = make_geom(K, M, esize, diag_only);
  const SplitChoice c = choose_splits2(n_folds, max_rows, p.g, esize, target_wg(K, M, esize));
  set_plan_splits(p, c.s_off, c.s_diag);
  p.fstat_bytes_per_fold = fold_mode ? align_up(fstat_len(K, M) * 8, 256) : 0;
  for (;;) {
    const size_t per_fold = (size_t)p.splits * p.g.unit_bytes + p.fstat_bytes_per_fold;
    int64_t nb = (int64_t)(ws_bytes / per_fold);
    if (nb >= 1) { p.folds_per_batch = nb < n_folds ? nb : n_folds; return CVM_OK; }
    if (p.splits == 1) return CVM_EWORKSPACE;
    set_plan_splits(p, (p.s_off + 1) / 2, (p.s_diag + 1) / 2);
  }
}

// item counts and grid of a launch over n_seg segments
template <typename T> void set_items(WgramArgs<T> &a, const Plan &p, int64_t n_seg) {
  const Geom &g = p.g;
  a.n_seg = (int)n_seg; a.splits = p.splits; a.s_off = p.s_off; a.s_diag = p.s_diag; a.g = g;
  const int per0 = g.diag_only ? 0 : g.nTiles - g.P, per1 = g.P * g.Yc;
  a.n_items0 = (long)n_seg * p.s_off * per0;
  a.n_items1 = (long)n_seg * p.s_diag * per1;
  a.ipx0 = (a.n_items0 + 7) / 8;
  a.ipx1 = (a.n_items1 + 7) / 8;
}
void set_fin_splits(FinArgs &f, const Plan &p, int n_sum) {
  f.splits = p.splits; f.s_off = p.s_off; f.s_diag = p.s_diag; f.n_sum = n_sum;
}

// can this problem take the float64 LDS-DMA kernel (wgram4_kernel)?
template <typename T> bool wgram4_ok(const WgramArgs<T> &a, bool aligned) {
  static const bool force_fallback = getenv("CVM_FORCE_FALLBACK") && atoi(getenv("CVM_FORCE_FALLBACK")) != 0;
  if (force_fallback || !aligned) return false;   // aligned: X and its rows on 16-byte boundaries
  if (sizeof(T) == 8)   // the Y tile rows go by 16-byte pieces too: M even, Y 16-byte aligned
    return (a.g.M % 2 == 0) && ((uintptr_t)a.Y % 16 == 0) && ((uintptr_t)a.w % 8 == 0);
  return ((uintptr_t)a.Y % 4 == 0) && ((uintptr_t)a.w % 4 == 0);   // float32: Y and w by dwords
}

// (Earlier versions kept the queue heads in the last QUEUE_BYTES of the caller's workspace: the
// reserve is still carved out so that workspace sizes do not change.)  They were zeroed on the
// stream before every launch.
struct WsCarve { size_t usable; unsigned *queue; };
WsCarve carve_queue(void *ws, size_t ws_bytes) {
  if (!ws || ws_bytes < QUEUE_BYTES + 256) return WsCarve{0, nullptr};
  const size_t off = (ws_bytes - QUEUE_BYTES) & ~(size_t)127;
  return WsCarve{off, (unsigned *)((char *)ws + off)};
}
constexpr size_t QUEUE_RESERVE = QUEUE_BYTES + 256;   // what the workspace-size functions add for it

// Work-queue blocks of the persistent Gram kernel (8 heads + an exit counter, QUEUE_BYTES each):
// a pool per device, allocated and zeroed once.  Every stream gets its own block (launches on one
// stream run one after the other, and the last workgroup of a launch leaves the block zeroed;
// launches on different streams may overlap and never share a block).  (The heads used to live in
// the caller's workspace and were zeroed on the stream before every launch: a 5 us kernel in front
// of every Gram launch.)
// The pool is BOUNDED and recycled: a process may create and destroy any number of streams.  When all
// QUEUE_POOL blocks have owners, the block whose stream has gone longest without a Gram launch and has
// nothing in flight (hipStreamQuery: idle, or no longer a stream at all) changes hands -- a block is all
// zero whenever no launch of its stream is running.  Only QUEUE_POOL streams with Gram launches IN FLIGHT
// at the same moment exhaust it.
constexpr unsigned QUEUE_POOL = 1024;       // streams per device that can have Gram launches in flight at once
struct QueuePool {
  unsigned *mem = nullptr;
  std::map<hipStream_t, unsigned> block_of;
  std::vector<hipStream_t> owner;           // by block
  std::vector<unsigned long long> last;     // by block: tick of its last hand-out
  std::vector<hipEvent_t> ev;               // by block: recorded behind the block's last Gram launch (the library's own)
  std::vector<char> state;                  // by block: 0 never launched (all zero), 1 ev says when its last launch is done,
                                            //           2 handed out, launch not enqueued yet, 3 never recycled (captured),
                                            //           4 launched while the pool did not track completions (never recycled)
  unsigned long long tick = 0;
};
// Completion tracking (an event of the library's own behind every Gram launch: ~4 us of command-processor time per
// launch, measured on the emulated 8-GPU step, 0.131 -> 0.136 ms) starts only when a process has used more than
// this many streams on a device -- a service that creates and destroys streams; the blocks handed out before that
// stay with their streams for good (at most QUEUE_TRACK_FROM of the 1024).
constexpr size_t QUEUE_TRACK_FROM = 64;
std::mutex g_queue_mu;
QueuePool g_queue_pools[64];
// The block of stream `st` (handed out in state 2: the caller enqueues its launch and then calls queue_launched).
// When all blocks have owners, the least recently used block whose LAST launch has completed -- by the library's
// own event behind that launch, never by a query of somebody else's (possibly destroyed, possibly capturing)
// stream handle -- changes hands; a block handed out whose launch is not enqueued yet (state 2) and a block whose
// launch was captured into a graph (state 3: it may replay at any time) are never taken.
unsigned *acquire_queue(int dev, hipStream_t st, unsigned *block_out) {
  QueuePool &p = g_queue_pools[dev & 63];
  std::lock_guard<std::mutex> lock(g_queue_mu);
  if (!p.mem) {
    void *mem = nullptr;
    if (hipMalloc(&mem, (size_t)QUEUE_POOL * QUEUE_BYTES) != hipSuccess) return nullptr;
    if (hipMemset(mem, 0, (size_t)QUEUE_POOL * QUEUE_BYTES) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
      (void)hipFree(mem);
      return nullptr;
    }
    p.mem = (unsigned *)mem;
  }
  ++p.tick;
  auto it = p.block_of.find(st);
  if (it == p.block_of.end()) {
    unsigned b;
    if (p.owner.size() < QUEUE_POOL) {
      b = (unsigned)p.owner.size();
      p.owner.push_back(st);
      p.last.push_back(0);
      p.ev.push_back(nullptr);
      p.state.push_back(0);
    } else {
      // least recently used first; take the first one whose last launch is known to be over
      std::vector<unsigned> order(QUEUE_POOL);
      for (unsigned i = 0; i < QUEUE_POOL; ++i) order[i] = i;
      std::sort(order.begin(), order.end(), [&](unsigned a, unsigned c) { return p.last[a] < p.last[c]; });
      b = QUEUE_POOL;
      for (unsigned cand : order) {
        if (p.state[cand] >= 2) continue;
        if (p.state[cand] == 1) {
          const hipError_t q = hipEventQuery(p.ev[cand]);
          if (q != hipSuccess) { if (q != hipErrorNotReady) (void)hipGetLastError(); continue; }
        }
        b = cand;
        break;
      }
      if (b == QUEUE_POOL) return nullptr;
      p.block_of.erase(p.owner[b]);
      p.owner[b] = st;
    }
    it = p.block_of.emplace(st, b).first;
  }
  const unsigned b = it->second;
  p.last[b] = p.tick;
  const bool track = p.owner.size() > QUEUE_TRACK_FROM;
  if (p.state[b] != 3) p.state[b] = track ? 2 : 4;
  // (bit 31 of *block_out: the caller reports the enqueued launch with queue_launched)
  if (block_out) *block_out = b | (track && p.state[b] == 2 ? 0x80000000u : 0u);
  return p.mem + (size_t)b * (QUEUE_BYTES / sizeof(unsigned));
}
// the launch that uses block `b` has been enqueued on `st` (or has failed: then the block is as it was)
void queue_launched(int dev, unsigned b, hipStream_t st) {
  QueuePool &p = g_queue_pools[dev & 63];
  std::lock_guard<std::mutex> lock(g_queue_mu);
  if (b >= p.state.size() || p.state[b] == 3) return;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
  if (cs != hipStreamCaptureStatusNone) { p.state[b] = 3; return; }       // a graph may replay the launch whenever it likes
  if (!p.ev[b] && hipEventCreateWithFlags(&p.ev[b], hipEventDisableTiming) != hipSuccess) { p.ev[b] = nullptr; p.state[b] = 3; return; }
  if (hipEventRecord(p.ev[b], st) != hipSuccess) { (void)hipGetLastError(); p.state[b] = 3; return; }
  p.state[b] = 1;
}

int device_cu_count(int dev) {
  static std::atomic<int> cus[64];
  int c = cus[dev & 63].load(std::memory_order_relaxed);
  if (c > 0) return c;
  if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) c = 256;
  cus[dev & 63].store(c, std::memory_order_relaxed);
  return c;
}

int current_cu_count() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  return device_cu_count(dev);
}

// kind: KIND_FIT / KIND_FOLD for the timing recorder, -1 = not recorded; grid_cap: at most so many persistent
// workgroups (0: one per CU -- the retry launch of the fused route takes a few)
template <typename T>
int launch_wgram(const WgramArgs<T> &a, bool weighted, bool gather, bool aligned, hipStream_t st,
                 int kind, unsigned *queue, bool fused = false, int grid_cap = 0) {
  WgramArgs<T> args = a;
  args.queue = queue;
  // CVM_FORCE_FALLBACK=1 sends float64 problems through the general (register-staged)
  // kernel too -- used by the tests to cover both kernels.  The ablation switches of
  // CVM_DEBUG (wrong results by design) exist in the -DCVM_STAMPS diagnostic build only.
#ifdef CVM_STAMPS
  static const int dbg_env = getenv("CVM_DEBUG") ? atoi(getenv("CVM_DEBUG")) : 0;
#else
  static const int dbg_env = 0;
#endif
  args.dbg = dbg_env;
  const dim3 grid((unsigned)((a.ipx0 + a.ipx1) * 8)), block(NTHREADS);
  const size_t lds = 2 * BUF_ELEMS * sizeof(T) + 3 * STAGE_ROWS * sizeof(int64_t);
  int dev = 0;
  HIP_OK(hipGetDevice(&dev));
#define CVM_LAUNCH(W, GA, AL)                                                                 \
  do {                                                                                     \
    static std::atomic<unsigned long long> attr_done{0};   /* one bit per device */       \
    if (attr_needed(attr_done, dev)) {                                                     \
      HIP_OK(hipFuncSetAttribute((const void *)wgram_kernel<T, W, GA, AL>,                 \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));   \
      attr_set(attr_done, dev);                                                            \
    }                                                                                      \
    hipLaunchKernelGGL((wgram_kernel<T, W, GA, AL>), grid, block, lds, st, args);          \
  } while (0)
  TimedLaunch *tl = nullptr;
  if (kind >= 0 && g_timing.load(std::memory_order_relaxed)) {
    std::lock_guard<std::mutex> lk(g_timing_mu);
    if (g_ntimed < MAX_TIMED) {
      tl = &g_timed[g_ntimed++];
      if (!tl->a) { HIP_OK(hipEventCreate(&tl->a)); HIP_OK(hipEventCreate(&tl->b)); }
      tl->kind = kind;
    }
  }
  if (tl) HIP_OK(hipEventRecord(tl->a, st));
  const bool fast = wgram4_ok<T>(a, aligned) && !(dbg_env & 16);
  if (fused && !(fast && gather))
    return fail(CVM_EINVAL, "launch_wgram: fused epilogue needs the LDS-DMA kernel%s");
  if (fast) {
    (void)queue;
    args.clock_stamps = g_clock_buf.load(std::memory_order_acquire);
    args.clock_wgs = args.clock_stamps ? g_clock_wgs.load(std::memory_order_relaxed) : 0;
    unsigned qblock = 0;
    args.queue = acquire_queue(dev, st, &qblock);
    if (!args.queue) return fail(CVM_ELAUNCH, "launch_wgram: no work-queue block (allocation failed, or 1024 streams with Gram launches in flight at once)%s");
    // persistent workgroups: as many as the device keeps resident at once -- one per CU by the kernel's LDS,
    // asked of the occupancy API once per kernel and device -- and no more than the lists are long.  (Results
    // never depend on the number: the workgroups PULL items; what does is that no workgroup of the grid waits
    // in the dispatcher while resident ones spin on a flag.)
    long wgs = 8 * (a.ipx0 + a.ipx1);
    const long cus = device_cu_count(dev);
    const dim3 block4(NT4);
    // (the fused epilogue reuses the stage ring for its tiles: a little more than the ring in float32)
    const size_t lds4 = fused ? fused_launch_lds_bytes<T>() : lds4_bytes<T>();
#define CVM_LAUNCH4(W, GA, FU)                                                              \
  do {                                                                                      \
    static std::atomic<unsigned long long> attr_done{0};                                    \
    static std::atomic<int> per_cu[64];                                                     \
    if (attr_needed(attr_done, dev)) {                                                      \
      HIP_OK(hipFuncSetAttribute((const void *)wgram4_kernel<T, W, GA, FU>,                 \
                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));   \
      int occ = 0;                                                                          \
      if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)wgram4_kernel<T, W, GA, FU>, NT4, lds4) != hipSuccess || occ < 1) \
        return fail(CVM_ELAUNCH, "launch_wgram: the Gram kernel does not fit a compute unit%s"); \
      per_cu[dev & 63].store(occ > 1 ? 1 : occ, std::memory_order_relaxed);   /* (LDS: one) */ \
      attr_set(attr_done, dev);                                                             \
    }                                                                                       \
    const long resident = cus * per_cu[dev & 63].load(std::memory_order_relaxed);           \
    if (wgs > resident) wgs = resident;                                                     \
    if (grid_cap > 0 && wgs > grid_cap) wgs = grid_cap;                                     \
    hipLaunchKernelGGL((wgram4_kernel<T, W, GA, FU>), dim3((unsigned)wgs), block4, lds4, st, args); \
  } while (0)
    if (fused) {
      if (weighted) CVM_LAUNCH4(true, true, true); else CVM_LAUNCH4(false, true, true);
    } else if (weighted) { if (gather) CVM_LAUNCH4(true, true, false); else CVM_LAUNCH4(true, false, false); }
    else { if (gather) CVM_LAUNCH4(false, true, false); else CVM_LAUNCH4(false, false, false); }
#undef CVM_LAUNCH4
    if (qblock & 0x80000000u) queue_launched(dev, qblock & 0x7fffffffu, st);
  } else if (weighted) {
    if (gather) { if (aligned) CVM_LAUNCH(true, true, true); else CVM_LAUNCH(true, true, false); }
    else { if (aligned) CVM_LAUNCH(true, false, true); else CVM_LAUNCH(true, false, false); }
  } else {
    if (gather) { if (aligned) CVM_LAUNCH(false, true, true); else CVM_LAUNCH(false, true, false); }
    else { if (aligned) CVM_LAUNCH(false, false, true); else CVM_LAUNCH(false, false, false); }
  }
#undef CVM_LAUNCH
  if (tl) HIP_OK(hipEventRecord(tl->b, st));
  HIP_OK(hipGetLastError());
  return CVM_OK;
}

bool rows_aligned(const void *X, int K, int esize) {
  return ((uintptr_t)X % 16 == 0) && (((size_t)K * esize) % 16 == 0);
}

// full-data matrices and column statistics from the partials: one launch of the many-workgroup
// kernel when rows are 16-byte aligned, else the statistics kernel and the general apply kernel
// (returns whether the kernel that can compact the folds' partials ran: FinArgs::compact)
template <typename T> bool launch_fit_apply(FinArgs f, const Geom &g, double *gstats, hipStream_t st) {
  const bool aligned = ((size_t)g.K * sizeof(T)) % 16 == 0 && ((uintptr_t)f.out_XTX % 16 == 0);
  if (!aligned) f.compact = 0;
  if (aligned) {
    f.gstats = gstats;
    hipLaunchKernelGGL((fit_apply_kernel<T>), dim3(g.nTiles * APPLY_SUB * fit_rc<T>() + g.P * FIT_PCH + FIT_STAT_WGS),
                       dim3(FIT_THREADS), 0, st, f);
  } else {
    hipLaunchKernelGGL((fit_stats_kernel<T>), dim3(32), dim3(64), 0, st, f, gstats);
    hipLaunchKernelGGL((apply_kernel<T, false>), dim3(g.nTiles * APPLY_SUB + g.P, 1), dim3(APPLY_THREADS_FIT), 0, st, f);
  }
  return f.compact != 0;
}

template <typename T>
int gram_fit_impl(const void *X, const void *Y, const void *w, int64_t N, int K, int M, int dtype,
                  void *G, void *H, double *gstats, int32_t *neg_flag, void *ws, size_t ws_bytes,
                  hipStream_t st) {
  const WsCarve wq = carve_queue(ws, ws_bytes);
  ws_bytes = wq.usable;
  Plan p;
  int rc = make_plan(1, N, K, M, dtype, CVM_RET_XTX | CVM_RET_XTY, ws_bytes, false, p);
  if (rc != CVM_OK) return fail(rc, "cvm_gram_fit: workspace too small%s");
  WgramArgs<T> a;
  memset(&a, 0, sizeof(a));
  a.X = (const T *)X; a.Y = (const T *)Y; a.w = (const T *)w;
  a.idx = nullptr; a.offs = nullptr; a.N = N; a.seg0 = 0;
  set_items(a, p, 1);
  a.ws = (char *)ws;
  rc = launch_wgram<T>(a, w != nullptr, false, rows_aligned(X, K, sizeof(T)), st, KIND_FIT, wq.queue);
  if (rc != CVM_OK) return rc;
  FinArgs f;
  memset(&f, 0, sizeof(f));
  f.g = p.g; set_fin_splits(f, p, 1); f.n_seg = 1; f.seg0 = 0; f.ws = (const char *)ws;
  f.w = w; f.out_XTX = G; f.out_XTY = (Y && M > 0) ? H : nullptr; f.neg_flag = neg_flag;
  launch_fit_apply<T>(f, p.g, gstats, st);
  HIP_OK(hipGetLastError());
  return CVM_OK;
}

// Folds of at most so many rows take the direct kernels (small_folds.hpp).  Beyond one 32-row chunk
// the direct kernels pay about 0.2 of a fold's store time per 16 rows, the fused Gram route
// (one unit per fold) starts higher and grows slower; the crossover was measured per shape on
// one box (tools/exp_small_limit.py, profiles/r3/exp_small_limit.txt: outputs in TB/s, direct /
// fused): float64 K=512 33 rows 2.36 / 2.35, 48 rows 2.07 / 2.29; K=1024 48 rows 2.81 / 2.65, 64
// rows 2.34 / 2.46; K=2048 48 rows 3.09 / 2.88, 64 rows 2.51 / 2.58; K=4096 (G no longer fits the
// caches of the fused epilogue) 33 rows 3.53 / 1.65, 128 rows 1.65 / 1.24; float32 K=512 48 rows
// 1.91 / 1.87, 64 rows 1.59 / 1.73; K=1024 ... 4096 64 rows 2.13 / 2.02 ... 2.38 / 2.35, 80 rows
// 1.77 / 1.84 ... 2.04 / 2.24.  CVM_SMALL_MAXN (32 .. 128) overrides the table for measurements
// and tests.
// Round 6, the resident route (resident.hpp: float32 XTX of folds of at most 32 rows with G in the register files of the
// whole chip): K a multiple of the 1024-column block, at least 4 folds per workgroup set.
// Where it is the route (cvm_debug_resident / CVM_RESIDENT: 2 = this rule, the default; 1 = wherever the shape allows: tests and
// measurements; 0 = never): K = 2048 or a multiple of 4096 (the blocks fill whole sets of 512 workgroups: K = 3072 has 288 blocks and
// loses 20 %) and at least 16 folds per workgroup set (K >= 4096: 16 folds per batch; K = 2048, four sets: 64).  Same-box alternations
// with the eight-wave kernel (profiles/r6/hbm_regime/resident_route.txt section 5), ms per call against the tile kernel / mid_tile_kernel:
// K = 4096, 48 folds of 16 / 8 / 1 rows 0.611 / 0.583 / 0.580 against 0.673 / 0.653 / 0.642 (-9 ... -11 %), 160 folds 1.91 against
// 2.20 (-13 %), 24 folds 0.337 against 0.352, 12 folds 0.190 against 0.189; K = 8192, 40 folds 1.95 against 2.25 (8 rows: 1.85 / 1.98);
// K = 2048, 400 folds 1.25-1.29 against 1.42-1.45, 80 folds 0.290 against 0.31; K = 1024, 1000 folds of 16 rows 0.85 against 0.95
// but of 4 rows 0.80 against 0.775: not in the rule.  Folds of 17 to 32 rows (operand blocks of 36 rows, one tile per step): K = 4096,
// 48 folds of 32 / 24 rows 0.689 / 0.670 against 0.821 / 0.748 (-16 / -10 %).
constexpr int RES_NP = 32;                               // rows per fold at most (operand blocks of 8, 16 or 32 rows)
constexpr int RES_AUTO_MINK = 4096;
std::atomic<int> g_resident{-1};                        // -1: environment not consulted yet
inline int resident_mode() {
  int v = g_resident.load(std::memory_order_relaxed);
  if (v >= 0) return v;
  const char *e = getenv("CVM_RESIDENT");
  int init = e ? atoi(e) : 2, expect = -1;
  if (init < 0 || init > 2) init = 2;
  g_resident.compare_exchange_strong(expect, init, std::memory_order_relaxed);
  return g_resident.load(std::memory_order_relaxed);
}
// the shape alone (workspace sizing); the number of folds is looked at when the batch is launched
inline bool res_shape_ok(int K, int esize, int64_t max_rows) {
  const int mode = resident_mode();
  return mode != 0 && esize == 4 && max_rows <= RES_NP && K >= RES_BC && K % RES_BC == 0 && (mode == 1 || K == 2048 || K % RES_AUTO_MINK == 0);
}
inline bool res_folds_ok(int64_t nb, int K) {
  const int nblk = (K / 32) * (K / RES_BC), sets = nblk >= RES_WG ? 1 : RES_WG / (nblk > 0 ? nblk : 1);
  return resident_mode() == 1 || nb >= (int64_t)16 * sets;
}
inline size_t res_pack_bytes(int K, int64_t max_rows) { return (size_t)2 * ((max_rows <= 16 ? 16 : 32) + 4) * K * 4; }
// workspace of the direct small-fold route per fold: the statistics vector (+ the resident route's operand block)
inline size_t small_ws_per_fold(int K, int M, int esize, int64_t max_rows) {
  return align_up(fstat_len(K, M) * 8, 256) + (res_shape_ok(K, esize, max_rows) ? res_pack_bytes(K, max_rows) : 0);
}
int small_route_limit(int K, int esize) {
  static const int forced = [] {
    const char *e = getenv("CVM_SMALL_MAXN");
    int v = e ? atoi(e) : 0;
    if (e && v < SMALL_ROWS) v = SMALL_ROWS;
    if (v > SMALL_MAXN) v = SMALL_MAXN;
    return v;
  }();
  if (forced) return forced;
  if (esize == 4) return K < 768 ? 48 : 64;
  return K < 768 ? SMALL_ROWS : (K < 4096 ? 48 : SMALL_MAXN);
}

template <typename T>
int small_fold_impl(const void *X, const void *Y, const void *w, const int64_t *idx, const int64_t *offsets,
                    int64_t n_folds, int64_t max_rows, int K, int M, unsigned flags, double ddof, double resolution,
                    const void *G, const void *H, const double *gstats, void *out_XTX, void *out_XTY,
                    void *out_muX, void *out_sdX, void *out_muY, void *out_sdY, double *out_fold,
                    void *ws, size_t ws_bytes, hipStream_t st) {
  // per fold: the float64 statistics vector
  const size_t per_fold = small_ws_per_fold(K, M, (int)sizeof(T), max_rows);
  int64_t nb_max = (int64_t)((ws_bytes > 256 ? ws_bytes - 256 : 0) / per_fold);
  if (nb_max < 1) return fail(CVM_EWORKSPACE, "cvm_fold_update: workspace cannot hold one fold%s");
  if (nb_max > 32768) nb_max = 32768;   // grid.y
  if (nb_max > n_folds) nb_max = n_folds;
  static const bool no_direct = getenv("CVM_NO_DIRECT") != nullptr;   // tests: force the transposing kernel
  SmallArgs a;
  memset(&a, 0, sizeof(a));
  a.X = X; a.Y = Y; a.w = w; a.idx = idx; a.offs = offsets; a.K = K; a.M = M;
  a.G = G; a.H = H; a.gstats = gstats; a.fstats = (double *)ws;
  // (round 4's accumulator-direct tile kernel -- small_tile_kernel, CVM_SMALL_TILE -- was measured slower than the
  //  kernels below and left the product in round 6: tools/experiments/pruned_r6_routes.patch)
  a.out_XTX = (flags & CVM_RET_XTX) ? out_XTX : nullptr;
  a.out_XTY = (flags & CVM_RET_XTY) ? out_XTY : nullptr;
  a.out_muX = out_muX; a.out_sdX = out_sdX; a.out_muY = out_muY; a.out_sdY = out_sdY;
  a.out_fold = out_fold; a.ddof = ddof; a.resolution = resolution; a.flags = flags;
  a.P64 = (K + ST - 1) / ST; a.nT64 = a.P64 * (a.P64 + 1) / 2;
  static const int noremap_env = getenv("CVM_SMALL_NOREMAP") ? atoi(getenv("CVM_SMALL_NOREMAP")) : 0;   // (measurements)
  a.noremap = noremap_env;
  a.inl_n = -1;
  if (flags & CVM_IDX_HOST) {            // one fold, indices on the host: into the kernel arguments
    a.inl_n = (int)(offsets[1] - offsets[0]);
    for (int i = 0; i < a.inl_n; ++i) a.inl[i] = idx[offsets[0] + i];
    a.idx = nullptr; a.offs = nullptr;
  }
  for (int64_t f0 = 0; f0 < n_folds; f0 += nb_max) {
    const int64_t nb = (n_folds - f0 < nb_max) ? n_folds - f0 : nb_max;
    a.seg0 = f0;
    // folds per workgroup of the apply kernel: as many as leave >= 16 workgroups per CU in the launch
    const int64_t wg1 = (int64_t)(a.nT64 + a.P64) * nb;
    int fpb = (int)(wg1 / (16 * 256));
    if (fpb < 1) fpb = 1;
    // (the workgroup keeps the row numbers of its folds in 256 slots: 8 folds of up to 32 rows, 4 of 64, 2 of 128)
    a.rshift = max_rows <= 32 ? 5 : (max_rows <= 64 ? 6 : 7);
    static const int fpb_env = getenv("CVM_SMALL_FPB") ? atoi(getenv("CVM_SMALL_FPB")) : 0;   // (measurements)
    if (fpb_env > 0) fpb = fpb_env;
    if (fpb > (256 >> a.rshift)) fpb = 256 >> a.rshift;
    a.nb = (int)nb; a.fpb = fpb;
    // (statistics: one workgroup per fold and block of 256 columns -- a thread that walks 16 columns of 16 rows
    //  one load at a time took 44 us at K = 4096, a seventh of the whole call)
    int sy = (K + M + 255) / 256;
    while ((int64_t)sy * nb > 65535 * 4 && sy > 1) sy = (sy + 1) / 2;
    const dim3 gs((unsigned)nb, (unsigned)sy), ga((unsigned)(a.nT64 + a.P64), (unsigned)((nb + fpb - 1) / fpb));
    // one- and two-row folds (leave-one-out) of a matrix whose rows are not whole 128-byte lines and
    // fit one column chunk: whole rows of the full output, nothing transposed (small_rows_kernel;
    // measured +23 % at the reference's published leave-one-out shape K = 500 in float64 and +25 %
    // in float32; it reads G for both triangles, so only while G stays in an XCD's L2 -- 2 MB:
    // float32 K = 900 -8 %, K = 1000 -18 %)
    const int vw = 16 / (int)sizeof(T);
    const int lpr = K <= 64 * vw ? 64 : (K <= 128 * vw ? 128 : 256);     // pieces per row of a workgroup
    const int tc = lpr * vw;
    const bool direct = !no_direct && max_rows <= 2 && nb >= 8 && K <= tc && 2 * K > tc &&
                        (size_t)K * K * sizeof(T) <= ((size_t)2 << 20) + (64 << 10) &&
                        ((size_t)K * sizeof(T)) % 16 == 0 && ((size_t)K * sizeof(T)) % 128 != 0 &&
                        ((uintptr_t)G % 16 == 0) && ((uintptr_t)X % 16 == 0) &&
                        (!a.out_XTX || (uintptr_t)a.out_XTX % 16 == 0);
    TimedLaunch *t_st = timed_begin(2, st);
    if (w) hipLaunchKernelGGL((small_stats_kernel<T, true>), gs, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((small_stats_kernel<T, false>), gs, dim3(256), 0, st, a);
    timed_end(t_st, st);
    TimedLaunch *t_up = (a.out_XTX || a.out_XTY) ? timed_begin(3, st) : nullptr;
    if (a.out_XTX || a.out_XTY) {
      if (direct) {
        const int panels = ((K + sr_rows<T>(lpr) - 1) / sr_rows<T>(lpr)) * ((K + tc - 1) / tc);
        int fpr = (int)((int64_t)panels * nb / (16 * 256));
        if (fpr < 1) fpr = 1;
        if (fpr > 8) fpr = 8;   // (measured flat from 8 to 16, worse at 32: too few workgroups)
        static const int fpr_env = getenv("CVM_SMALL_FPR") ? atoi(getenv("CVM_SMALL_FPR")) : 0;   // (measurements)
        if (fpr_env > 0) fpr = fpr_env > SA_FPB ? SA_FPB : fpr_env;
        a.fpb = fpr;
        a.gx = panels; a.gy = (int)((nb + fpr - 1) / fpr);
        const dim3 gd((unsigned)(8 * (((size_t)a.gx * a.gy + 7) / 8)));
#define CVM_ROWS(L)                                                                          \
  do {                                                                                       \
    if (w) hipLaunchKernelGGL((small_rows_kernel<T, true, L>), gd, dim3(256), 0, st, a);     \
    else hipLaunchKernelGGL((small_rows_kernel<T, false, L>), gd, dim3(256), 0, st, a);      \
  } while (0)
        if (lpr == 64) CVM_ROWS(64);
        else if (lpr == 128) CVM_ROWS(128);
        else CVM_ROWS(256);
#undef CVM_ROWS
      } else {
        // XTX by the resident route where the shape allows it (the XTY panels stay with small_apply_kernel)
        bool resident = false, xty_packed = false;
        if constexpr (sizeof(T) == 4) {
          const bool shape = res_shape_ok(K, 4, max_rows);            // (K >= 1024 then)
          const int nblk_all = shape ? (K / 32) * (K / RES_BC) : 1;
          int groups = nblk_all >= RES_WG ? 1 : RES_WG / nblk_all;
          if (groups > nb / 4) groups = (int)(nb / 4);
          resident = shape && res_folds_ok(nb, K) && a.out_XTX && !(flags & CVM_IDX_HOST) && groups >= 1 &&
                     (uintptr_t)G % 4 == 0 && (uintptr_t)a.out_XTX % 4 == 0;
          if (resident) {
            float *pk = (float *)((char *)ws + align_up((size_t)nb_max * align_up(fstat_len(K, M) * 8, 256), 256));
            xty_packed = a.out_XTY && M > 0 && M <= RES_XTY_M;
            ResArgs r;
            memset(&r, 0, sizeof(r));
            r.G = G; r.out = a.out_XTX; r.pk = pk; r.K = K; r.nb = (int)nb; r.seg0 = f0; r.nbc = K / RES_BC; r.groups = groups;
            int dev = 0;
            HIP_OK(hipGetDevice(&dev));
            // (operand blocks of 8 rows for batches of folds of at most 8 rows: five k-pairs per tile instead of nine, two
            //  LDS-DMA instructions per operand instead of three)
            auto run = [&](auto np_tag) -> int {
              constexpr int NPR = decltype(np_tag)::value;
              // eight waves per workgroup (four per SIMD) unless CVM_RES_WAVES=4 asks for the first kernel (comparisons)
              static const bool four_env = getenv("CVM_RES_WAVES") && atoi(getenv("CVM_RES_WAVES")) == 4;
              constexpr bool has4 = NPR <= 16;                       // (the four-wave kernel's LDS holds blocks of at most 16 rows)
              const bool four = four_env && has4;
              constexpr int lds4 = 4 * 7 * (NPR + 4) * 128, lds8 = res8_lds<NPR>();
              static std::atomic<unsigned long long> attr_done{0};   // one bit per device (one per instantiation)
              if (attr_needed(attr_done, dev)) {
                if constexpr (has4)
                  HIP_OK(hipFuncSetAttribute((const void *)res_apply_kernel<NPR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds4));
                HIP_OK(hipFuncSetAttribute((const void *)res8_apply_kernel<NPR>, hipFuncAttributeMaxDynamicSharedMemorySize, lds8));
                attr_set(attr_done, dev);
              }
              const dim3 gp((unsigned)nb, (unsigned)(K / 256));        // one column per thread
              if (w) hipLaunchKernelGGL((res_pack_kernel<float, NPR, true>), gp, dim3(256), 0, st, a, pk, (int)xty_packed);
              else hipLaunchKernelGGL((res_pack_kernel<float, NPR, false>), gp, dim3(256), 0, st, a, pk, (int)xty_packed);
              for (int b0 = 0; b0 < nblk_all; b0 += RES_WG) {
                r.blk0 = b0; r.nblk = nblk_all - b0 < RES_WG ? nblk_all - b0 : RES_WG;
                const unsigned wgs = (unsigned)(8 * (((size_t)r.nblk * groups + 7) / 8));
                if constexpr (has4) {
                  if (four) { hipLaunchKernelGGL((res_apply_kernel<NPR>), dim3(wgs), dim3(256), lds4, st, r); continue; }
                }
                hipLaunchKernelGGL((res8_apply_kernel<NPR>), dim3(wgs), dim3(512), lds8, st, r);
              }
              return CVM_OK;
            };
            const int rc = max_rows <= 8 ? run(std::integral_constant<int, 8>{})
                                          : (max_rows <= 16 ? run(std::integral_constant<int, 16>{}) : run(std::integral_constant<int, 32>{}));
            if (rc != CVM_OK) return rc;
          }
        }
        if (resident) { a.x0 = a.nT64; a.gx = a.P64; a.gy = (int)ga.y; }
        else { a.x0 = 0; a.gx = (int)ga.x; a.gy = (int)ga.y; }
        if (!resident || (a.out_XTY && M > 0 && !xty_packed)) {
          const dim3 g1((unsigned)(8 * (((size_t)a.gx * a.gy + 7) / 8)));
          if (w) hipLaunchKernelGGL((small_apply_kernel<T, true>), g1, dim3(256), 0, st, a);
          else hipLaunchKernelGGL((small_apply_kernel<T, false>), g1, dim3(256), 0, st, a);
        }
        a.x0 = 0;
      }
    }
    timed_end(t_up, st);
    HIP_OK(hipGetLastError());
  }
  return CVM_OK;
}

// mid_tile_kernel over the folds of one batch (their statistics are in m.fstats)
// Where mid_tile_kernel is the route (tools/exp_mid_small.sh; profiles/r4/mid_tile/exp_mid_small*.txt), rows per fold:
//   float64  K = 512: from 8 (1.32 against 1.46 ms for 2774 folds; 32 rows 1.93 / 2.40) to 256 (0.79 / 0.81; 320: 0.79 / 0.75)
//            K = 1024, 2048: from 16 (1.33 / 1.40 ms, 1.32 / 1.32) to 200 (2.76 / 2.84, 3.63 / 3.71)
//            K = 4096: never (3.0 against 1.1 ms at 8 rows: 2080 tiles per fold, G's tile rows come from HBM one by one)
//   float32  K = 512, 1024: from 8 (1.60 / 1.87 ms for 5548 folds; 100 rows 0.55 / 0.81) to 320 (0.44 / 0.48; 1.51 / 1.51)
//            K = 2048: from 16 (1.43 / 1.51) to 256 (4.70 / 5.06)
// below: the direct small-fold kernels; above: the fused epilogue of the Gram kernel.
inline int mid_default_minn(int K, int esize) {
  if (K > 2048) return 1 << 30;
  if (esize == 4) return K <= 1024 ? 8 : 16;
  return K < 768 ? 8 : 16;
}
inline int mid_default_maxn(int K, int esize) {
  if (K > 2048) return 0;
  if (esize == 4) return K <= 1024 ? 320 : 256;
  return K < 768 ? 256 : 200;
}
// what the kernel needs of the operands: rows of X, Y in whole 16-byte pieces, 32-bit row numbers
template <typename T> bool mid_operands_ok(const void *X, const void *Y, const void *w, int64_t N, int K, int M) {
  constexpr int EPL = 16 / (int)sizeof(T);
  return N <= 0x7fffffffLL && K >= EPL && rows_aligned(X, K, sizeof(T)) && M % EPL == 0 && (uintptr_t)Y % 16 == 0 &&
         (uintptr_t)w % sizeof(T) == 0;
}
template <typename T> int launch_mid(MidArgs m, bool weighted, int64_t nb, int64_t max_rows, hipStream_t st) {
  m.nt = (m.K + 63) / 64;
  m.n_xtx = m.nt * (m.nt + 1) / 2;
  m.yextra = (m.out_XTY && m.M > 16) ? (m.M - 16 + 63) / 64 : 0;
  m.ipf = m.n_xtx + m.nt * m.yextra;
  m.n_items = (long long)nb * m.ipf;
  m.per_xcd = (m.n_items + 7) / 8;
  m.nb = (int)nb;
  m.maxn = (int)((max_rows + 15) / 16 * 16);
  if (m.maxn < 16) m.maxn = 16;
  const size_t lds = mid_lds_bytes<T>(m.maxn);
  if (m.per_xcd * 8 > 0x7fffffffLL) return fail(CVM_EINVAL, "launch_mid: too many work items%s");
  const dim3 grid((unsigned)(m.per_xcd * 8));
  if (lds > 64 * 1024) return fail(CVM_EINVAL, "launch_mid: folds too long for the LDS lists%s");
  TimedLaunch *tl = timed_begin(KIND_FOLD, st);
  if (weighted) hipLaunchKernelGGL((mid_tile_kernel<T, true>), grid, dim3(MID_THREADS), lds, st, m);
  else hipLaunchKernelGGL((mid_tile_kernel<T, false>), grid, dim3(MID_THREADS), lds, st, m);
  timed_end(tl, st);
  HIP_OK(hipGetLastError());
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
  int64_t splits = colstats_splits(max_rows, n_folds, K, sizeof(T), current_cu_count());
  while (splits > 1 && (size_t)splits * g.unit_bytes + fst > ws_bytes) splits /= 2;
  const size_t per_fold = (size_t)splits * g.unit_bytes + fst;
  if (per_fold > ws_bytes) return fail(CVM_EWORKSPACE, "cvm_fold_update: workspace cannot hold one fold%s");
  int64_t per_batch = (int64_t)(ws_bytes / per_fold);
  if (per_batch > 32768) per_batch = 32768;
  const bool aligned = rows_aligned(X, K, sizeof(T));
  for (int64_t f0 = 0; f0 < n_folds; f0 += per_batch) {
    const int64_t nb = (n_folds - f0 < per_batch) ? n_folds - f0 : per_batch;
    ColArgs c;
    c.X = X; c.Y = Y; c.w = w; c.idx = idx; c.offs = offsets; c.seg0 = f0;
    c.g = g; c.ws = (char *)ws;
    colstats_shape<T>(c, K, M, nb, (int)splits);
    const dim3 grid = colstats_grid(c, nb);
    if (w) {
      if (aligned) hipLaunchKernelGGL((colstats_kernel<T, true, true>), grid, dim3(COL_THREADS), 0, st, c);
      else hipLaunchKernelGGL((colstats_kernel<T, true, false>), grid, dim3(COL_THREADS), 0, st, c);
    } else {
      if (aligned) hipLaunchKernelGGL((colstats_kernel<T, false, true>), grid, dim3(COL_THREADS), 0, st, c);
      else hipLaunchKernelGGL((colstats_kernel<T, false, false>), grid, dim3(COL_THREADS), 0, st, c);
    }
    FinArgs f;
    memset(&f, 0, sizeof(f));
    f.g = g; f.splits = f.s_off = f.s_diag = (int)splits; f.n_sum = 1;
    f.n_seg = (int)nb; f.seg0 = f0; f.ws = (const char *)ws;
    f.fstats = (double *)((char *)ws + (size_t)nb * splits * g.unit_bytes);
    f.offs = offsets; f.w = w; f.gstats = gstats;
    f.out_muX = out_muX; f.out_sdX = out_sdX; f.out_muY = out_muY; f.out_sdY = out_sdY;
    f.out_fold = out_fold; f.ddof = ddof; f.resolution = resolution; f.flags = flags;
    hipLaunchKernelGGL((fold_stats_kernel<T>), dim3((unsigned)nb, (unsigned)fold_stats_chunks(K, M, nb)), dim3(256), 0, st, f);
    HIP_OK(hipGetLastError());
  }
  return CVM_OK;
}

// Route switches of the fold stage: read from the environment ONCE, in one place (tests, tools/route_matrix.sh
// and the experiment scripts set them; none is needed in production).
struct FoldSwitches {
  int mid_minn, mid_maxn;      // CVM_MID_MINN / CVM_MID_MAXN: row limits of mid_tile_kernel (0: the measured table)
  bool mid_off;                // CVM_MID_TILE=0: never mid_tile_kernel
  bool force_fallback;         // CVM_FORCE_FALLBACK=1: the general Gram kernel
  bool no_fused;               // CVM_NO_FUSED=1: partials + apply_kernel for one-unit folds too
  bool prepass;                // CVM_FUSED_PREPASS=1: statistics by colstats_kernel + fold_stats_kernel, not in the launch
  int fused_order;             // CVM_FUSED_ORDER: 2 = fold-major lists (default), 1 = every list's diagonal items first
  int fused_test;              // CVM_FUSED_TEST_TIMEOUT: WgramArgs::test_mode
};
const FoldSwitches &fold_switches() {
  static const FoldSwitches sw = [] {
    auto num = [](const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; };
    FoldSwitches f;
    f.mid_minn = num("CVM_MID_MINN", 0); f.mid_maxn = num("CVM_MID_MAXN", 0);
    f.mid_off = num("CVM_MID_TILE", 1) == 0;
    f.force_fallback = num("CVM_FORCE_FALLBACK", 0) != 0;
    f.no_fused = num("CVM_NO_FUSED", 0) != 0;
    f.prepass = num("CVM_FUSED_PREPASS", 0) != 0;
    f.fused_order = num("CVM_FUSED_ORDER", 2) == 1 ? 1 : 2;
    f.fused_test = num("CVM_FUSED_TEST_TIMEOUT", 0);
    return f;
  }();
  return sw;
}

// what every kernel of one cvm_fold_update call is handed (the call's own arguments, typed once)
struct FoldCall {
  const void *X, *Y, *w;
  const int64_t *idx, *offsets;
  int64_t N;
  int K, M;
  unsigned flags;
  double ddof, resolution;
  const void *G, *H;
  const double *gstats;
  void *out_XTX, *out_XTY, *out_muX, *out_sdX, *out_muY, *out_sdY;
  double *out_fold;
  bool want_xty;
  int32_t *status;
};

// the statistics pre-pass over the folds [f0, f0 + nb): colstats_kernel + fold_stats_kernel; returns where the
// folds' statistics vectors lie in ws
template <typename T>
double *launch_prepass(const FoldCall &c, const Geom &gs, int64_t csplits, int64_t f0, int64_t nb, void *ws, hipStream_t st) {
  ColArgs ca;
  ca.X = c.X; ca.Y = c.Y; ca.w = c.w; ca.idx = c.idx; ca.offs = c.offsets; ca.seg0 = f0;
  ca.g = gs; ca.ws = (char *)ws;
  colstats_shape<T>(ca, c.K, c.M, nb, (int)csplits);
  const dim3 cgrid = colstats_grid(ca, nb);
  if (c.w) hipLaunchKernelGGL((colstats_kernel<T, true, true>), cgrid, dim3(COL_THREADS), 0, st, ca);
  else hipLaunchKernelGGL((colstats_kernel<T, false, true>), cgrid, dim3(COL_THREADS), 0, st, ca);
  FinArgs f;
  memset(&f, 0, sizeof(f));
  f.g = gs; f.splits = f.s_off = f.s_diag = (int)csplits; f.n_sum = 1;
  f.n_seg = (int)nb; f.seg0 = f0; f.ws = (const char *)ws;
  f.fstats = (double *)((char *)ws + (size_t)nb * csplits * gs.unit_bytes);
  f.offs = c.offsets; f.w = c.w; f.gstats = c.gstats;
  f.out_muX = c.out_muX; f.out_sdX = c.out_sdX; f.out_muY = c.out_muY; f.out_sdY = c.out_sdY;
  f.out_fold = c.out_fold; f.ddof = c.ddof; f.resolution = c.resolution; f.flags = c.flags;
  hipLaunchKernelGGL((fold_stats_kernel<T>), dim3((unsigned)nb, (unsigned)fold_stats_chunks(c.K, c.M, nb)),
                     dim3(256), 0, st, f);
  return f.fstats;
}

// arguments of mid_tile_kernel over the folds [f0, ...) (fstats: from the pre-pass)
inline MidArgs mid_args(const FoldCall &c, int64_t f0, const double *fstats) {
  MidArgs m;
  memset(&m, 0, sizeof(m));
  m.X = c.X; m.Y = c.Y; m.w = c.w; m.idx = c.idx; m.offs = c.offsets; m.seg0 = f0;
  m.fstats = fstats;
  m.G = c.G; m.H = c.H;
  m.out_XTX = c.out_XTX; m.out_XTY = c.want_xty ? c.out_XTY : nullptr;
  m.K = c.K; m.M = c.M; m.flags = c.flags;
  return m;
}

// arguments of a fused launch (wgram4_kernel<.., FUSED>, one unit per fold) over the folds [f0, f0 + nb)
template <typename T>
WgramArgs<T> fused_args(const FoldCall &c, const Plan &p, int64_t f0, int64_t nb, const double *fstats) {
  WgramArgs<T> a;
  memset(&a, 0, sizeof(a));
  a.X = (const T *)c.X; a.Y = (const T *)c.Y; a.w = (const T *)c.w;
  a.idx = c.idx; a.offs = c.offsets; a.N = c.N; a.seg0 = f0;
  set_items(a, p, nb);               // (one unit per fold: p.s_off == p.s_diag == 1)
  a.ws = nullptr;
  a.fstats = fstats; a.G = c.G; a.H = c.H;
  a.out_XTX = c.out_XTX; a.out_XTY = c.want_xty ? c.out_XTY : nullptr; a.flags = c.flags;
  if (fold_switches().fused_order == 2) {
    // fold-major lists (decode_slot): whole folds per XCD, a fold's diagonal items in front of its others
    a.diag_first = 2;
    a.fpx = (int)((nb + 7) / 8);
    const int per0 = p.g.diag_only ? 0 : p.g.nTiles - p.g.P, per1 = p.g.P * p.g.Yc;
    a.ipx0 = (long)a.fpx * per0; a.ipx1 = (long)a.fpx * per1;
  }
  return a;
}

template <typename T>
int fold_update_impl(const void *X, const void *Y, const void *w, const int64_t *idx,
                     const int64_t *offsets, const int64_t *host_offsets, int64_t n_folds, int64_t N,
                     int K, int M, int dtype, unsigned flags, double ddof, double resolution,
                     const void *G, const void *H, const double *gstats, void *out_XTX,
                     void *out_XTY, void *out_muX, void *out_sdX, void *out_muY, void *out_sdY,
                     double *out_fold, void *ws, size_t ws_bytes, hipStream_t st, int32_t *status) {
  int64_t max_rows = 0;
  for (int64_t f = 0; f < n_folds; ++f) {
    const int64_t n = host_offsets[f + 1] - host_offsets[f];
    if (n < 0) return fail(CVM_EINVAL, "cvm_fold_update: offsets must be non-decreasing%s");
    if (n > max_rows) max_rows = n;
  }
  const WsCarve wq = carve_queue(ws, ws_bytes);
  ws_bytes = wq.usable;
  const FoldSwitches &sw = fold_switches();
  const int esize = (int)sizeof(T);
  // where mid_tile_kernel is the route: rows per fold between the measured limits (CVM_MID_MINN / _MAXN override)
  const int mid_minn = sw.mid_off ? (1 << 30) : (sw.mid_minn > 0 ? sw.mid_minn : mid_default_minn(K, esize));
  const int mid_maxn = sw.mid_maxn > 0 ? sw.mid_maxn : mid_default_maxn(K, esize);
  // (... only where that kernel can run -- the conditions of the fused route below; a fold whose indices come
  //  inside the call -- CVM_IDX_HOST, one fold of at most 32 rows -- is the small route's)
  const bool skip_small = !(flags & CVM_IDX_HOST) && max_rows >= mid_minn && ((flags & CVM_RET_XTX) && out_XTX) &&
                          max_rows <= mid_maxn && mid_operands_ok<T>(X, Y, w, N, K, M) && !sw.force_fallback && !sw.no_fused;
  // (... and only if the fold stage is planned with one unit per fold -- always, unless a test forces a split plan)
  bool skip_small_ok = skip_small;
  // (round 6: float32 batches the resident route takes -- K a multiple of 1024, folds of at most 16 rows -- stay small folds)
  if (skip_small_ok && res_shape_ok(K, esize, max_rows) && res_folds_ok(n_folds, K) && n_folds >= 8 && K >= 2048) skip_small_ok = false;
  if (skip_small_ok) {
    Plan pp;
    if (make_plan(n_folds, max_rows, K, M, dtype, flags, (size_t)1 << 60, true, pp) != CVM_OK || pp.splits != 1) skip_small_ok = false;
  }
  if (max_rows <= small_route_limit(K, esize) && !skip_small_ok)
    return small_fold_impl<T>(X, Y, w, idx, offsets, n_folds, max_rows, K, M, flags, ddof, resolution, G, H, gstats,
                              out_XTX, out_XTY, out_muX, out_sdX, out_muY, out_sdY, out_fold, ws, ws_bytes, st);
  const bool want_xtx = (flags & CVM_RET_XTX) && out_XTX, want_xty = (flags & CVM_RET_XTY) && out_XTY;
  if (!want_xtx && !want_xty)   // statistics only: stream the rows once, no Gram launch
    return fold_statistics_impl<T>(X, Y, w, idx, offsets, n_folds, max_rows, K, M, flags, ddof, resolution,
                                   gstats, out_muX, out_sdX, out_muY, out_sdY, out_fold, ws, ws_bytes, st);
  const FoldCall c{X, Y, w, idx, offsets, N, K, M, flags, ddof, resolution, G, H, gstats,
                   out_XTX, out_XTY, out_muX, out_sdX, out_muY, out_sdY, out_fold, want_xty, status};
  Plan p;
  // (planned against an unlimited workspace first: the fused route below needs far less than
  //  the partials the general route plans for)
  int rc = make_plan(n_folds, max_rows, K, M, dtype, flags, (size_t)1 << 60, true, p);
  const bool aligned = rows_aligned(X, K, sizeof(T));
  {
    // Folds too small to be split over workgroups (one unit per fold): finish in the Gram
    // kernel's epilogue instead of writing partials for apply_kernel to read back.
    WgramArgs<T> probe;
    memset(&probe, 0, sizeof(probe));
    probe.Y = (const T *)Y; probe.w = (const T *)w; probe.g = p.g;
    if (p.splits == 1 && want_xtx && !sw.no_fused && wgram4_ok<T>(probe, aligned)) {
      Geom gs = make_geom(K, M, sizeof(T), 1);
      gs.tile_elems = 0; gs.h_elems = 0;
      gs.unit_bytes = align_up(gs.stat_len * 8, 256);
      const size_t fst = align_up(fstat_len(K, M) * 8, 256);
      int64_t csplits = colstats_splits(max_rows, n_folds, K, sizeof(T), current_cu_count());
      while (csplits > 1 && (size_t)csplits * gs.unit_bytes + fst > ws_bytes) csplits /= 2;
      const size_t per_fold = (size_t)csplits * gs.unit_bytes + fst;
      if (per_fold > ws_bytes) return fail(CVM_EWORKSPACE, "cvm_fold_update: workspace cannot hold one fold%s");
      int64_t per_batch = (int64_t)(ws_bytes / per_fold);
      if (per_batch > 16384) per_batch = 16384;
      // Folds of up to a few hundred rows: mid_tile_kernel (mid_tile.hpp) -- small work items, four
      // workgroups per CU, so that one item's stores overlap another's MFMAs -- behind the statistics pre-pass.
      const bool mid = !sw.mid_off && max_rows <= mid_maxn && mid_operands_ok<T>(X, Y, w, N, K, M);
      // Statistics formed INSIDE the Gram launch (round 4): the diagonal item of (fold, panel) sums the panel's
      // columns while it streams the fold's rows anyway and publishes the panel's training means / stds, the
      // off-diagonal items wait for the two flags they need (wgram4.hpp).  No colstats_kernel + fold_stats_kernel
      // pre-pass (70-90 us in front of a 0.5-1.2 ms launch at the C3 rows cut into 100 / 1000 folds).  One Y chunk
      // (M <= 32); CVM_FUSED_PREPASS=1: the pre-pass route (tests, comparisons).
      // Workspace of a batch: [fstats nb x fst | flags nb x P | status 4 ints | retry list nb x off-diagonal tiles]
      const int per0 = p.g.nTiles - p.g.P;
      const bool ink = !mid && !sw.prepass && p.g.Yc == 1 &&
                       fst + (size_t)p.g.P * 4 + (size_t)per0 * 8 + 512 <= per_fold;
      for (int64_t f0 = 0; f0 < n_folds; f0 += per_batch) {
        const int64_t nb = (n_folds - f0 < per_batch) ? n_folds - f0 : per_batch;
        if (ink) {
          double *fstats = (double *)ws;
          int *sflags = (int *)((char *)ws + align_up((size_t)nb * fst, 256));
          const size_t flag_bytes = align_up((size_t)nb * p.g.P * sizeof(int), 16);
          int *fstatus = (int *)((char *)sflags + flag_bytes);
          HIP_OK(hipMemsetAsync(sflags, 0, flag_bytes + 16, st));      // the flags and the two status words
          WgramArgs<T> a = fused_args<T>(c, p, f0, nb, fstats);
          a.stat_flags = sflags; if (!a.diag_first) a.diag_first = 1;
          a.gstats = gstats; a.ddof = ddof; a.resolution = resolution;
          a.out_muX = out_muX; a.out_sdX = out_sdX; a.out_muY = out_muY; a.out_sdY = out_sdY; a.out_fold = out_fold;
          a.fused_status = fstatus;
          a.retry_items = (unsigned long long *)((char *)fstatus + 16);
          a.test_mode = sw.fused_test;
          // (both launches between ONE pair of the timing recorder's events, like launch_mid's: the route's
          //  kernel time includes its -- normally empty, ~2 us -- retry launch)
          TimedLaunch *tl = timed_begin(KIND_FOLD, st);
          rc = launch_wgram<T>(a, w != nullptr, true, aligned, st, -1, wq.queue, true);
          if (rc != CVM_OK) return rc;
          // the items whose wait for a flag gave up (none, unless the device is shared in a way that breaks the
          // kernel's progress argument): once more, behind the launch that has raised every flag
          a.retry_mode = 1; a.status_out = status;
          rc = launch_wgram<T>(a, w != nullptr, true, aligned, st, -1, wq.queue, true, 64);
          timed_end(tl, st);
          if (rc != CVM_OK) return rc;
          continue;
        }
        const double *fstats = launch_prepass<T>(c, gs, csplits, f0, nb, ws, st);
        if (mid) rc = launch_mid<T>(mid_args(c, f0, fstats), w != nullptr, nb, max_rows, st);
        else rc = launch_wgram<T>(fused_args<T>(c, p, f0, nb, fstats), w != nullptr, true, aligned, st, KIND_FOLD, wq.queue, true);
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
    set_items(a, p, nb);
    a.ws = units;
    rc = launch_wgram<T>(a, w != nullptr, true, aligned, st, KIND_FOLD, wq.queue);
    if (rc != CVM_OK) return rc;
    FinArgs f;
    memset(&f, 0, sizeof(f));
    f.g = p.g; set_fin_splits(f, p, 1); f.n_seg = (int)nb; f.seg0 = f0; f.ws = units;
    f.fstats = (double *)((char *)fstats);
    f.offs = offsets; f.w = w; f.G = G; f.H = H; f.gstats = gstats;
    f.out_XTX = (flags & CVM_RET_XTX) ? out_XTX : nullptr;
    f.out_XTY = (flags & CVM_RET_XTY) ? out_XTY : nullptr;
    f.out_muX = out_muX; f.out_sdX = out_sdX; f.out_muY = out_muY; f.out_sdY = out_sdY;
    f.out_fold = out_fold; f.ddof = ddof; f.resolution = resolution; f.flags = flags;
    hipLaunchKernelGGL((fold_stats_kernel<T>), dim3((unsigned)nb, (unsigned)fold_stats_chunks(K, M, nb)), dim3(256), 0, st, f);
    if (f.out_XTX || f.out_XTY) {
      f.gx = p.g.nTiles * APPLY_SUB + p.g.P; f.gy = (int)nb;
      hipLaunchKernelGGL((apply_kernel<T, true>), dim3((unsigned)(8 * (((size_t)f.gx * f.gy + 7) / 8))),
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
  const WsCarve wq = carve_queue(ws, ws_bytes);
  ws_bytes = wq.usable;
  Plan p;
  const unsigned flags = CVM_RET_XTX | CVM_RET_XTY;
  int rc = make_plan(n_folds, max_rows, K, M, dtype, flags, ws_bytes, true, p);
  if (rc != CVM_OK || p.folds_per_batch < n_folds)
    return fail(CVM_EWORKSPACE, "cvm_sweep_fit: the workspace must hold the partials of all folds%s");
  WgramArgs<T> a;
  memset(&a, 0, sizeof(a));
  a.X = (const T *)X; a.Y = (const T *)Y; a.w = (const T *)w;
  a.idx = idx; a.offs = offsets; a.N = N; a.seg0 = 0;
  set_items(a, p, n_folds);
  a.ws = (char *)ws;
  rc = launch_wgram<T>(a, w != nullptr, true, rows_aligned(X, K, sizeof(T)), st, KIND_FOLD, wq.queue);
  if (rc != CVM_OK) return rc;
  FinArgs f;
  memset(&f, 0, sizeof(f));
  f.g = p.g; set_fin_splits(f, p, (int)n_folds);      // every unit of every fold, fold-major
  f.n_seg = 1; f.seg0 = 0; f.ws = (const char *)ws;
  f.w = w; f.out_XTX = G; f.out_XTY = (Y && M > 0) ? H : nullptr; f.neg_flag = neg_flag;
  // float64 with more than one row split: the fit finalize leaves every fold's raw update in the
  // fold's slot 0 (FinArgs::compact), and cvm_sweep_folds / cvm_sweep_fold_range then read one
  // partial per fold and tile (the multi-GPU step: this call, the exchange, that call -- with 1-2
  // folds per GPU the fold stage read 11-28 partials per tile: 21 us -> 6 us at C3 on 8 GPUs)
  static const bool no_compact = getenv("CVM_NO_COMPACT") != nullptr;       // tests: the uncompacted route
  f.compact = (sizeof(T) == 8 && !no_compact && (p.s_off > 1 || p.s_diag > 1)) ? 1 : 0;
  const bool compacted = launch_fit_apply<T>(f, p.g, gstats, st);
  HIP_OK(hipGetLastError());
  // plan token for cvm_sweep_folds: s_off | s_diag << 20 | compacted << 40
  if (splits_out) *splits_out = (int64_t)p.s_off | ((int64_t)p.s_diag << 20) | ((int64_t)(compacted ? 1 : 0) << 40);
  return CVM_OK;
}

template <typename T>
int sweep_folds_impl(const int64_t *offsets, int64_t n_total, int64_t fold0, int64_t n_folds, int K, int M, int dtype,
                     unsigned flags, double ddof, double resolution, int weighted, const void *G, const void *H,
                     const double *gstats, void *out_XTX, void *out_XTY, void *out_muX, void *out_sdX,
                     void *out_muY, void *out_sdY, double *out_fold, void *ws, size_t ws_bytes,
                     int64_t splits, hipStream_t st) {
  // folds [fold0, fold0 + n_folds) of the n_total folds whose partials cvm_sweep_fit left in ws;
  // outputs are written from index 0
  const Geom g = make_geom(K, M, sizeof(T), 0);
  Plan p;
  p.g = g;
  set_plan_splits(p, (int)(splits & 0xfffff), (int)((splits >> 20) & 0xfffff));
  const bool compacted = (splits >> 40) & 1;      // every fold's slot 0 holds the sum of its partials
  if (p.s_off < 1 || p.s_diag < 1) return fail(CVM_EINVAL, "cvm_sweep_folds: not a plan token of cvm_sweep_fit%s");
  const size_t units = (size_t)n_total * (size_t)p.splits * g.unit_bytes;
  if (units + (size_t)n_total * fstat_len(K, M) * 8 > ws_bytes)
    return fail(CVM_EWORKSPACE, "cvm_sweep_folds: workspace smaller than the one cvm_sweep_fit filled%s");
  FinArgs f;
  memset(&f, 0, sizeof(f));
  f.g = g; set_fin_splits(f, p, 1); f.n_seg = (int)n_folds; f.seg0 = 0;
  if (compacted) f.s_off = f.s_diag = 1;          // (f.splits stays the slot stride)
  f.ws = (const char *)ws + (size_t)fold0 * (size_t)p.splits * g.unit_bytes;
  f.fstats = (double *)((char *)ws + units) + (size_t)fold0 * fstat_len(K, M);
  f.offs = offsets + fold0; f.w = weighted ? (const void *)G : nullptr;   // non-null = weighted
  f.G = G; f.H = H; f.gstats = gstats;
  f.out_XTX = (flags & CVM_RET_XTX) ? out_XTX : nullptr;
  f.out_XTY = (flags & CVM_RET_XTY) ? out_XTY : nullptr;
  f.out_muX = out_muX; f.out_sdX = out_sdX; f.out_muY = out_muY; f.out_sdY = out_sdY;
  f.out_fold = out_fold; f.ddof = ddof; f.resolution = resolution; f.flags = flags;
  // Matrices wanted and few partials per fold (compacted: one): every workgroup of apply_kernel derives
  // the fold statistics of its tile itself and the statistics kernel is not launched at all -- one
  // launch and one launch gap less per call (the per-rank step of the multi-GPU path, the reference's
  // per-fold loop).  CVM_NO_INLINE_STATS: the two-launch route (tests).
  static const bool no_inline = getenv("CVM_NO_INLINE_STATS") != nullptr;
  const bool mats = f.out_XTX || f.out_XTY;
  f.inline_stats = (mats && !no_inline && f.s_diag <= 4 && M <= APPLY_INLINE_MAXM) ? 1 : 0;
  if (!f.inline_stats)
    hipLaunchKernelGGL((fold_stats_kernel<T>), dim3((unsigned)n_folds, (unsigned)fold_stats_chunks(K, M, n_folds)), dim3(256), 0, st, f);
  if (mats) {
    f.gx = g.nTiles * APPLY_SUB + g.P; f.gy = (int)n_folds;
    hipLaunchKernelGGL((apply_kernel<T, true>), dim3((unsigned)(8 * (((size_t)f.gx * f.gy + 7) / 8))),
                       dim3(APPLY_THREADS), 0, st, f);
  }
  HIP_OK(hipGetLastError());
  return CVM_OK;
}

// cvm_sweep_all: cvm_sweep_fit and cvm_sweep_folds in one call.  With few folds (<= SWF_MAX) and
// 16-byte aligned rows the finalize half is two launches that read every partial once
// (sweep_stats_kernel, sweep_finish_kernel); otherwise the two calls' own kernels.  Same bits.
template <typename T>
int sweep_all_impl(const void *X, const void *Y, const void *w, const int64_t *idx, const int64_t *offsets,
                   const int64_t *host_offsets, int64_t n_folds, int64_t N, int K, int M, int dtype, unsigned flags,
                   double ddof, double resolution, void *G, void *H, double *gstats, int32_t *neg_flag,
                   void *out_XTX, void *out_XTY, void *out_muX, void *out_sdX, void *out_muY, void *out_sdY,
                   double *out_fold, void *ws, size_t ws_bytes, hipStream_t st, int64_t *splits_out) {
  static const bool no_merge = getenv("CVM_NO_SWEEP_MERGE") != nullptr;     // tests: the separate kernels
  const bool want_xtx = flags & CVM_RET_XTX, want_xty = flags & CVM_RET_XTY;
  const bool merged = !no_merge && n_folds <= SWF_MAX && ((size_t)K * sizeof(T)) % 16 == 0 &&
                      ((uintptr_t)G % 16 == 0) && (!want_xtx || (uintptr_t)out_XTX % 16 == 0);
  if (!merged) {
    int64_t token = 0;
    int rc = sweep_fit_impl<T>(X, Y, w, idx, offsets, host_offsets, n_folds, N, K, M, dtype, G, H, gstats, neg_flag,
                               ws, ws_bytes, st, &token);
    if (rc != CVM_OK) return rc;
    if (splits_out) *splits_out = token;
    return sweep_folds_impl<T>(offsets, n_folds, 0, n_folds, K, M, dtype, flags, ddof, resolution, w != nullptr, G, H,
                               gstats, out_XTX, out_XTY, out_muX, out_sdX, out_muY, out_sdY, out_fold, ws,
                               carve_queue(ws, ws_bytes).usable, token, st);
  }
  int64_t max_rows = 0;
  for (int64_t f = 0; f < n_folds; ++f) {
    const int64_t n = host_offsets[f + 1] - host_offsets[f];
    if (n < 0) return fail(CVM_EINVAL, "cvm_sweep_all: offsets must be non-decreasing%s");
    if (n > max_rows) max_rows = n;
  }
  if (host_offsets[n_folds] - host_offsets[0] != N)
    return fail(CVM_EINVAL, "cvm_sweep_all: the folds must cover each of the N rows exactly once%s");
  const WsCarve wq = carve_queue(ws, ws_bytes);
  ws_bytes = wq.usable;
  Plan p;
  int rc = make_plan(n_folds, max_rows, K, M, dtype, CVM_RET_XTX | CVM_RET_XTY, ws_bytes, true, p);
  if (rc != CVM_OK || p.folds_per_batch < n_folds)
    return fail(CVM_EWORKSPACE, "cvm_sweep_all: the workspace must hold the partials of all folds%s");
  WgramArgs<T> a;
  memset(&a, 0, sizeof(a));
  a.X = (const T *)X; a.Y = (const T *)Y; a.w = (const T *)w;
  a.idx = idx; a.offs = offsets; a.N = N; a.seg0 = 0;
  set_items(a, p, n_folds);
  a.ws = (char *)ws;
  rc = launch_wgram<T>(a, w != nullptr, true, rows_aligned(X, K, sizeof(T)), st, KIND_FOLD, wq.queue);
  if (rc != CVM_OK) return rc;
  const size_t units = (size_t)n_folds * (size_t)p.splits * p.g.unit_bytes;
  FinArgs f;
  memset(&f, 0, sizeof(f));
  f.g = p.g; set_fin_splits(f, p, 1); f.n_seg = (int)n_folds; f.seg0 = 0; f.ws = (const char *)ws;
  f.fstats = (double *)((char *)ws + units);
  f.offs = offsets; f.w = w; f.G = G; f.H = H; f.gstats = gstats; f.neg_flag = neg_flag;
  f.out_XTX = want_xtx ? out_XTX : nullptr;
  f.out_XTY = want_xty ? out_XTY : nullptr;
  f.out_muX = out_muX; f.out_sdX = out_sdX; f.out_muY = out_muY; f.out_sdY = out_sdY;
  f.out_fold = out_fold; f.ddof = ddof; f.resolution = resolution; f.flags = flags;
  hipLaunchKernelGGL((sweep_stats_kernel<T>), dim3((unsigned)((K + M + 15) / 16)), dim3(256), 0, st, f, gstats);
  constexpr int C = 16 * (16 / (int)sizeof(T));
  const unsigned xb = (unsigned)(((K + SWF_R - 1) / SWF_R) * ((K + C - 1) / C));
  // (round 6) the folds' partial loads dealt over four groups of threads -- 2-3 dependent round trips per thread instead
  // of ten, the same sums -- when its LDS (the folds' updates of a block, float64) fits; CVM_SWF4=0: the 256-thread kernel
  static const bool swf4 = !(getenv("CVM_SWF4") && atoi(getenv("CVM_SWF4")) == 0);
  constexpr int PPR = CVM_SWF4_PPR;
  const size_t lds4 = swf4_lds_bytes<T, PPR>((int)n_folds);
  if (swf4 && lds4 <= (size_t)150 * 1024) {
    int dev = 0;
    HIP_OK(hipGetDevice(&dev));
    static std::atomic<unsigned long long> attr_done{0};   // one bit per device
    if (attr_needed(attr_done, dev)) {
      HIP_OK(hipFuncSetAttribute((const void *)sweep_finish4_kernel<T, PPR>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
      attr_set(attr_done, dev);
    }
    constexpr int C4 = PPR * (16 / (int)sizeof(T));
    const unsigned xb4 = (unsigned)(((K + SWF_R - 1) / SWF_R) * ((K + C4 - 1) / C4));
    const unsigned hb4 = (Y && M > 0) ? (unsigned)(((size_t)K * M + swf4_epw<PPR>() - 1) / swf4_epw<PPR>()) : 0u;
    hipLaunchKernelGGL((sweep_finish4_kernel<T, PPR>), dim3(xb4 + hb4), dim3(swf4_threads<PPR>()), lds4, st, f, (T *)G,
                       (T *)((Y && M > 0) ? H : nullptr));
    HIP_OK(hipGetLastError());
    if (splits_out) *splits_out = (int64_t)p.s_off | ((int64_t)p.s_diag << 20);
    return CVM_OK;
  }
  const unsigned hb = (Y && M > 0) ? (unsigned)(((size_t)K * M + SWF_EPW - 1) / SWF_EPW) : 0u;
  hipLaunchKernelGGL((sweep_finish_kernel<T>), dim3(xb + hb), dim3(SWF_T), 0, st, f, (T *)G, (T *)((Y && M > 0) ? H : nullptr));
  HIP_OK(hipGetLastError());
  if (splits_out) *splits_out = (int64_t)p.s_off | ((int64_t)p.s_diag << 20);
  return CVM_OK;
}

