// pls.hpp -- part of libcvmhip.so (included by cvmhip.hip inside its anonymous namespace).
// The step after the hot path: Improved Kernel PLS, algorithm #2 of
// Dayal & MacGregor (J. Chemometrics 11 (1997) 73-85) -- what the out-of-tree consumer `ikpls`
// (reference README.md:23, cvmatrix/partitioner.py:27-31) runs on every fold's (XTX, XTY) --
// on the training matrices where cvm_fold_update left them, in HBM, for a batch of folds.
//
// Per component the only work that grows like K^2 is u = XTX r; everything else is O(K (M + A)).
// A fold is cut into S row slices, one workgroup each: a workgroup owns rows [k0, k1) of XTX,
// of the deflated XTY, of W, P, R and B, keeps its slice of XTX in LDS when it fits (XTX is then
// read from HBM once, not once per component), and trades only K + O(S (M^2 + A)) numbers with
// the other slices of its fold per component, through L2/HBM, behind a per-fold barrier (a
// monotonic counter; slices of one launch are co-resident by construction: folds x S <= CUs).
// (Two sliced launches running at once on one device can each hold half the CUs and starve the
// other's slices: the spin limit then reports status 1 instead of hanging; run one at a time.)
// With S == 1 (many folds) the barrier is a __syncthreads().  All arithmetic in float64; fixed
// summation orders, no float atomics: results do not depend on scheduling.
#pragma once

constexpr int PLS_THREADS = 512;
constexpr int PLS_NW = PLS_THREADS / 64;
constexpr int PLS_MAXM = 64;        // responses (the M x M eigenproblem lives in LDS)
constexpr int PLS_MAXA = 512;       // components
constexpr size_t PLS_LDS_BUDGET = 150 * 1024;

struct PlsArgs {
  const void *XTX, *XTY;            // [F][K][K], [F][K][M]
  int K, M, A, S, rows;             // slices per fold, rows per slice
  int y_in_lds, pr_in_lds;
  double *Yw, *Bw;                  // [F][K][M] deflated XTY (unless in LDS), running B
  double *Pw, *Rw;                  // [F][A][K] (unless in LDS)
  double *xch;                      // [F][xch_len] (S > 1)
  unsigned *cnt;                    // [F] barrier counters (zeroed before the launch)
  int *status;                      // [1]  set to 1 if a barrier timed out
  void *B, *W, *P, *Q, *R;          // outputs ([F][A][K][M]; [F][K][A] x3 and [F][M][A], optional)
  int *n_fit;                       // [F]
  double eps;
  // pls_rep_kernel (replicated small state): folds of this launch, folds per XCD, each slice's private
  // copies of P^T and R^T ([fold][slice][2][A][K]), the folds' u exchange buffers ([fold][2][K])
  int nf, per_x;
  double *prw, *xu;
  // the safe re-run behind a launch whose workgroups wait for each other (round 4): a kernel given
  // run_if returns at once unless *run_if == 1 (a barrier of the launch before it timed out)
  const int *run_if;
  long spin_limit;                  // barrier spins before a slice gives up (1 << 21: seconds)
  int skip_block;                   // tests only (CVM_PLS_TEST_TIMEOUT): this workgroup returns at once, -1: none
};

#ifdef CVM_STAMPS
__device__ unsigned long long g_pls_stamps[16];
#define PLS_COUNT_SQUARING() do { if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && threadIdx.x < 64) g_pls_stamps[15] += 1; } while (0)
#else
#define PLS_COUNT_SQUARING()
#endif
__host__ __device__ inline size_t pls_xch_len(int K, int M, int A, int S) {
  return (size_t)S * M * M + (size_t)S * (1 + A) + (size_t)K + (size_t)S * (1 + M);
}

// Sum over the 64 lanes, the same value in every lane.  Data-parallel-primitive moves inside the
// VALU (row shifts inside 16-lane rows, then the row broadcasts 15 and 31 of the GFX9 family) instead
// of ds_bpermute round trips through the LDS pipeline: about a third of the latency, and the PLS
// kernel's small phases are chains of such reductions.  Lane 63 ends up with the total.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_add(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, BANK_MASK, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, BANK_MASK, false);
  return v + __hiloint2double(hi, lo);      // lanes without a source add +0.0
}
__device__ __forceinline__ double wave_sum(double v) {
  v = dpp_add<0x111, 0xF, 0xF>(v);          // row_shr:1
  v = dpp_add<0x112, 0xF, 0xF>(v);          // row_shr:2
  v = dpp_add<0x114, 0xF, 0xE>(v);          // row_shr:4, banks 1-3
  v = dpp_add<0x118, 0xF, 0xC>(v);          // row_shr:8, banks 2-3: lane 15 of every row = row sum
  v = dpp_add<0x142, 0xA, 0xF>(v);          // row_bcast:15 into rows 1 and 3
  v = dpp_add<0x143, 0xC, 0xF>(v);          // row_bcast:31 into rows 2 and 3: lane 63 = total
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

// Largest of 64 non-negative values, the same in every lane (lanes without a source compare with +0.0).
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ double dpp_max(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, ROW_MASK, BANK_MASK, false);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, ROW_MASK, BANK_MASK, false);
  return fmax(v, __hiloint2double(hi, lo));
}
__device__ __forceinline__ double wave_max_nonneg(double v) {
  v = dpp_max<0x111, 0xF, 0xF>(v);
  v = dpp_max<0x112, 0xF, 0xF>(v);
  v = dpp_max<0x114, 0xF, 0xE>(v);
  v = dpp_max<0x118, 0xF, 0xC>(v);
  v = dpp_max<0x142, 0xA, 0xF>(v);
  v = dpp_max<0x143, 0xC, 0xF>(v);
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), 63);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), 63);
  return __hiloint2double(hi, lo);
}

// sum over the workgroup, same value (bitwise) in every thread
__device__ __forceinline__ double block_sum(double v, double *red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < PLS_NW; ++w) t += red[w];
  return t;
}

// Numbers traded between the slices of a fold.  SLICED: device-coherent accesses (sc1: they go
// past the non-coherent per-XCD L2, so no cache has to be written back or invalidated around the
// barrier); one slice per fold: the same buffers live in LDS.
template <bool SLICED> __device__ __forceinline__ void xput(double *p, double v) {
  if constexpr (SLICED) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else *p = v;
}
template <bool SLICED> __device__ __forceinline__ double xget(const double *p) {
  if constexpr (SLICED) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else return *p;
}

// barrier over the S workgroups of one fold; `target` counts the arrivals expected so far.
// Every thread first waits for its own stores to be acknowledged, so the numbers are in place
// before the arrival is counted.
template <bool SLICED>
__device__ __forceinline__ bool fold_barrier(unsigned *cnt, unsigned &target, int S, int *status, int *lflag, long spin_limit) {
  if constexpr (!SLICED) { __syncthreads(); return true; }
  target += (unsigned)S;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    // Ordering is done at the instruction level, not with release/acquire fences: every number the
    // slices trade is written and read by device-coherent accesses (xput / xget: sc1, they do not
    // live in the non-coherent per-XCD L2), every thread has waited for its stores to be
    // acknowledged (vmcnt(0) above) before the workgroup barrier that precedes this arrival, and
    // the readers issue their loads after the barrier that follows the wait.  Agent-scope fences
    // here would write back and invalidate whole caches four times per component: measured
    // 47 -> 67 us per component at the C3 shape (release/acquire on the counter: 70 us).
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    int ok = 1;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > spin_limit) { ok = 0; break; }      // (1 << 21: seconds) the slices were not co-resident
    }
    if (!ok) *status = 1;
    *lflag = ok;
  }
  asm volatile("" ::: "memory");
  __syncthreads();
  return *lflag != 0;
}

#ifdef CVM_STAMPS
#define PLS_STAMP(i)                                                          \
  do {                                                                        \
    if (blockIdx.x == 0 && threadIdx.x == 0) {                                \
      const unsigned long long now_ = __builtin_readcyclecounter();           \
      g_pls_stamps[i] += now_ - stamp_;                                       \
      stamp_ = now_;                                                          \
    }                                                                         \
  } while (0)
#else
#define PLS_STAMP(i)
#endif

// u[i] = sum_k Xrow_i[k] r[k] for ROWS rows at a time per wave: 16-byte loads, ROWS x 4 of them in
// flight per lane (the stream from HBM/L2 needs the parallelism; from LDS it does no harm)
template <typename T, int ROWS>
__device__ __forceinline__ void matvec_rows(const T *__restrict__ x0, size_t ld, int K, const double *__restrict__ rl,
                                            int lane, double (&acc)[ROWS]) {
  constexpr int VW = 16 / (int)sizeof(T);
  typedef T vec_t __attribute__((ext_vector_type(VW)));
  const int nch = K / VW;                                      // K % VW == 0 here
  int ch = lane;
  for (; ch + 3 * 64 < nch; ch += 4 * 64) {
    vec_t v[ROWS][4];
#pragma unroll
    for (int q = 0; q < ROWS; ++q)
#pragma unroll
      for (int u = 0; u < 4; ++u) v[q][u] = *reinterpret_cast<const vec_t *>(x0 + q * ld + (size_t)(ch + u * 64) * VW);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      double rv[VW];
#pragma unroll
      for (int e = 0; e < VW; ++e) rv[e] = rl[(ch + u * 64) * VW + e];
#pragma unroll
      for (int q = 0; q < ROWS; ++q)
#pragma unroll
        for (int e = 0; e < VW; ++e) acc[q] += (double)v[q][u][e] * rv[e];
    }
  }
  for (; ch < nch; ch += 64) {
    vec_t v[ROWS];
#pragma unroll
    for (int q = 0; q < ROWS; ++q) v[q] = *reinterpret_cast<const vec_t *>(x0 + q * ld + (size_t)ch * VW);
    double rv[VW];
#pragma unroll
    for (int e = 0; e < VW; ++e) rv[e] = rl[ch * VW + e];
#pragma unroll
    for (int q = 0; q < ROWS; ++q)
#pragma unroll
      for (int e = 0; e < VW; ++e) acc[q] += (double)v[q][e] * rv[e];
  }
}

typedef double pls_v4d __attribute__((ext_vector_type(4)));

// v_mfma_f64_16x16x4_f64 register maps: A operand lane l = A[l & 15][l >> 4], B operand lane l =
// B[l >> 4][l & 15], C/D register q of lane l = C[(l >> 4) + 4 q][l & 15].  So register q of a
// block X held in the C/D map is, as it stands, the B operand of k-step q for  . X  and the A
// operand of k-step q for  X^T . : an M x M matrix cut in NB x NB blocks of 16 is multiplied by
// itself without moving a number.

// slice part of XTY^T XTY: s[bi][bj] += Y[:, bi]^T Y[:, bj], four rows per MFMA
template <int NB>
__device__ __forceinline__ void pls_gram_mfma(const double *Y, int yst, int n, int M, int wave, int lane,
                                              pls_v4d (&s)[NB][NB]) {
  const int col = lane & 15, sub = lane >> 4;
  for (int r0 = 4 * wave; r0 < n; r0 += 4 * PLS_NW) {
    const int r = r0 + sub;
    double y[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) y[b] = (16 * b + col < M && r < n) ? Y[(size_t)r * yst + 16 * b + col] : 0.0;
#pragma unroll
    for (int bi = 0; bi < NB; ++bi)
#pragma unroll
      for (int bj = 0; bj < NB; ++bj) s[bi][bj] = __builtin_amdgcn_mfma_f64_16x16x4f64(y[bi], y[bj], s[bi][bj], 0, 0, 0);
  }
}

// one wave: m <- m^2 / trace(m^2) until m is numerically rank one (m symmetric, trace 1).
// trace(m^2) = sum of the squares of m's entries (m symmetric): it is reduced over the lanes from
// the INPUT of a squaring while the matrix cores form the product, so the divisor is there when
// the product is.
template <int NB>
__device__ __forceinline__ void pls_power_mfma(pls_v4d (&m)[NB][NB], int lane) {
  for (int it = 0; it < 64; ++it) {
    pls_v4d sq[NB][NB];
#pragma unroll
    for (int bi = 0; bi < NB; ++bi)
#pragma unroll
      for (int bj = 0; bj < NB; ++bj) {
        pls_v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int bk = 0; bk < NB; ++bk)                        // (m m)[bi][bj] += m[bi][bk] m[bk][bj]; A operand: m[bi][bk]^T = m[bk][bi]
#pragma unroll
          for (int q = 0; q < 4; ++q) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(m[bk][bi][q], m[bk][bj][q], acc, 0, 0, 0);
        sq[bi][bj] = acc;
      }
    double d = 0.0;
#pragma unroll
    for (int bi = 0; bi < NB; ++bi)
#pragma unroll
      for (int bj = 0; bj < NB; ++bj)
#pragma unroll
        for (int q = 0; q < 4; ++q) d = fma(m[bi][bj][q], m[bi][bj][q], d);
    PLS_COUNT_SQUARING();
    const double t2 = wave_sum(d);                             // sum lambda^2 with sum lambda = 1
    const double inv = __builtin_amdgcn_rcp(t2);               // a scaling only: the test below reads t2 itself
#pragma unroll
    for (int bi = 0; bi < NB; ++bi)
#pragma unroll
      for (int bj = 0; bj < NB; ++bj)
#pragma unroll
        for (int q = 0; q < 4; ++q) m[bi][bj][q] = sq[bi][bj][q] * inv;
    if (1.0 - t2 < 1e-13) break;                               // rank one: lambda2/lambda1 < 5e-14
  }
}

// phase 1 of a component: the slice's part of XTY^T XTY (M <= 16 NB), summed over the waves in a
// fixed order and handed to the other slices of the fold
template <int NB, bool SLICED>
__device__ __forceinline__ void pls_gram_phase(const double *Y, int yst, int n, int M, double *ms, double *x_S_slice,
                                               int tid) {
  const int lane = tid & 63, wave = tid >> 6, col = lane & 15, sub = lane >> 4;
  pls_v4d s[NB][NB];
#pragma unroll
  for (int bi = 0; bi < NB; ++bi)
#pragma unroll
    for (int bj = 0; bj < NB; ++bj) s[bi][bj] = pls_v4d{0.0, 0.0, 0.0, 0.0};
  pls_gram_mfma<NB>(Y, yst, n, M, wave, lane, s);
#pragma unroll
  for (int bi = 0; bi < NB; ++bi)
#pragma unroll
    for (int bj = 0; bj < NB; ++bj) {
      if (bi + bj) __syncthreads();                            // ms is reused block after block
#pragma unroll
      for (int q = 0; q < 4; ++q) ms[wave * 256 + (sub + 4 * q) * 16 + col] = s[bi][bj][q];
      __syncthreads();
      if (tid < 256) {
        double msum = 0.0;
#pragma unroll
        for (int w = 0; w < PLS_NW; ++w) msum += ms[w * 256 + tid];
        const int i = 16 * bi + (tid >> 4), j = 16 * bj + (tid & 15);
        if (i < M && j < M) xput<SLICED>(&x_S_slice[i * M + j], msum);
      }
    }
}

template <int NB>
__device__ __forceinline__ void pls_eig_phase(const double *S0, double tr, int M, double *fin, int lane) {
  const int col = lane & 15, sub = lane >> 4;
  const double itr = 1.0 / tr;
  pls_v4d m[NB][NB];
#pragma unroll
  for (int bi = 0; bi < NB; ++bi)
#pragma unroll
    for (int bj = 0; bj < NB; ++bj)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = 16 * bi + sub + 4 * q, c = 16 * bj + col;
        m[bi][bj][q] = (row < M && c < M) ? S0[(size_t)row * M + c] * itr : 0.0;
      }
  pls_power_mfma<NB>(m, lane);
#pragma unroll
  for (int bi = 0; bi < NB; ++bi)
#pragma unroll
    for (int bj = 0; bj < NB; ++bj)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = 16 * bi + sub + 4 * q, c = 16 * bj + col;
        if (row < M && c < M) fin[(size_t)row * M + c] = m[bi][bj][q];
      }
}

// 32 < M <= 16 NB: only the blocks on and above the diagonal are accumulated (NB (NB + 1) / 2
// accumulators instead of NB^2) and each is handed over at (i, j) and (j, i) -- the same sum, so
// the matrix is symmetric to the bit.
template <int NB, bool SLICED>
__device__ __forceinline__ void pls_gram_phase_tri(const double *Y, int yst, int n, int M, double *ms,
                                                   double *x_S_slice, int tid) {
  const int lane = tid & 63, wave = tid >> 6, col = lane & 15, sub = lane >> 4;
  constexpr int NT = NB * (NB + 1) / 2;
  pls_v4d s[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) s[t] = pls_v4d{0.0, 0.0, 0.0, 0.0};
  for (int r0 = 4 * wave; r0 < n; r0 += 4 * PLS_NW) {
    const int r = r0 + sub;
    double y[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) y[b] = (16 * b + col < M && r < n) ? Y[(size_t)r * yst + 16 * b + col] : 0.0;
    int t = 0;
#pragma unroll
    for (int bi = 0; bi < NB; ++bi)
#pragma unroll
      for (int bj = bi; bj < NB; ++bj, ++t) s[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(y[bi], y[bj], s[t], 0, 0, 0);
  }
  int t = 0;
#pragma unroll
  for (int bi = 0; bi < NB; ++bi)
#pragma unroll
    for (int bj = bi; bj < NB; ++bj, ++t) {
      if (t) __syncthreads();
#pragma unroll
      for (int q = 0; q < 4; ++q) ms[wave * 256 + (sub + 4 * q) * 16 + col] = s[t][q];
      __syncthreads();
      if (tid < 256) {
        double msum = 0.0;
#pragma unroll
        for (int w = 0; w < PLS_NW; ++w) msum += ms[w * 256 + tid];
        const int i = 16 * bi + (tid >> 4), j = 16 * bj + (tid & 15);
        if (i < M && j < M) {
          xput<SLICED>(&x_S_slice[i * M + j], msum);
          if (bi != bj) xput<SLICED>(&x_S_slice[j * M + i], msum);
        }
      }
    }
}

// 32 < M <= 16 NB: the squarings with the matrix in LDS and every wave at work -- block (bi, bj) of
// the product by wave (bi NB + bj) mod waves, its operands read from LDS straight into the register
// maps above (A[i][k] = cur[k][i]: the matrix is symmetric).  cur, scaled by `scale`, has trace 1;
// the product is stored with that scaling and its trace (sum lambda^2) is the next divisor.  Two
// LDS buffers in turn, one barrier per squaring.  Returns the buffer of the last product (not
// divided by its trace: the caller takes a column's direction).  Called by all threads.
template <int NB>
__device__ __forceinline__ double *pls_eig_lds(const double *S0, double tr, int M, double *Ba, double *Bb,
                                               double *trp, int tid) {
  const int lane = tid & 63, wave = tid >> 6, col = lane & 15, sub = lane >> 4;
  const double *cur = S0;
  double *nxt = Ba;
  double scale = 1.0 / tr;
  for (int it = 0; it < 64; ++it) {
    const double s2 = scale * scale;
    double d = 0.0;
    for (int b = wave; b < NB * NB; b += PLS_NW) {
      const int bi = b / NB, bj = b - bi * NB;
      const int ca = 16 * bi + col, cb = 16 * bj + col;
      pls_v4d acc = {0.0, 0.0, 0.0, 0.0};
      double av[4 * NB], bv[4 * NB];                           // all operands in flight before the first MFMA
#pragma unroll
      for (int t = 0; t < 4 * NB; ++t) {
        const int k = 4 * t + sub;
        av[t] = (k < M && ca < M) ? cur[(size_t)k * M + ca] : 0.0;
        bv[t] = (k < M && cb < M) ? cur[(size_t)k * M + cb] : 0.0;
      }
#pragma unroll
      for (int t = 0; t < 4 * NB; ++t) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[t], bv[t], acc, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int row = 16 * bi + sub + 4 * q;
        const double v = acc[q] * s2;
        if (row < M && cb < M) nxt[(size_t)row * M + cb] = v;
        if (bi == bj && sub + 4 * q == col) d += v;
      }
    }
    PLS_COUNT_SQUARING();
    d = wave_sum(d);
    if (lane == 0) trp[(it & 1) * PLS_NW + wave] = d;
    __syncthreads();
    double t2 = 0.0;
#pragma unroll
    for (int w = 0; w < PLS_NW; ++w) t2 += trp[(it & 1) * PLS_NW + w];
    scale = __builtin_amdgcn_rcp(t2);
    cur = nxt;
    nxt = (nxt == Ba) ? Bb : Ba;
    if (1.0 - t2 < 1e-13) break;                               // the same test as pls_power_mfma
  }
  return const_cast<double *>(cur);
}

// q = the dominant eigenvector of the symmetric M x M matrix S0 (LDS), left in qv[0, M): repeated
// squaring B_0 = S / trace, B_{t+1} = B_t^2 / trace(B_t^2) -- trace(B_t^2) = sum lambda^2 (trace 1
// before) reaches 1 when B_t is numerically rank one -- then the column with the largest diagonal
// entry, polished by two power steps with S0 itself.  Called by all threads after a barrier that
// completed S0; Ba, Bb: two more M x M buffers, trp: 2 x waves numbers.  The small reductions
// (trace, largest diagonal, norms) are one lane per entry and a wave reduction, the same in every wave.
#ifdef CVM_STAMPS
#define PLS_QSTAMP(i) do { if (stamp_p && blockIdx.x == 0 && threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); g_pls_stamps[i] += now_ - *stamp_p; } } while (0)
#else
#define PLS_QSTAMP(i)
#endif
__device__ __forceinline__ void pls_dominant_q(const double *S0, double *Ba, double *Bb, double *trp, double *qv,
                                               int M, int tid, unsigned long long *stamp_p = nullptr) {
  const int lane = tid & 63, wave = tid >> 6;
  const double tr = wave_sum(lane < M ? S0[(size_t)lane * M + lane] : 0.0);
  if (!(tr > 0.0)) {
    if (tid < M) qv[tid] = 0.0;
    __syncthreads();
    return;
  }
  double *fin = Ba;                                            // the (nearly) rank-one power of S
  if (M <= 32) {
    // wave 0, the matrix in registers (see the register maps above)
    if (wave == 0) {
      if (M <= 16) pls_eig_phase<1>(S0, tr, M, fin, lane);
      else pls_eig_phase<2>(S0, tr, M, fin, lane);
    }
    __syncthreads();
  } else {
    fin = M <= 48 ? pls_eig_lds<3>(S0, tr, M, Ba, Bb, trp, tid) : pls_eig_lds<4>(S0, tr, M, Ba, Bb, trp, tid);
  }
  PLS_QSTAMP(12);                                              // (diagnostic builds: cycles since phase 2a began)
  double *scr = fin == Bb ? Ba : Bb;
  const double dg = lane < M ? fmax(fin[(size_t)lane * M + lane], 0.0) : 0.0;
  const double mx = wave_max_nonneg(dg);
  int best = __ffsll((unsigned long long)__ballot(lane < M && dg == mx)) - 1;   // the lowest index among equals
  if (best < 0) best = 0;
  if (tid < M) qv[tid] = fin[(size_t)tid * M + best];
  __syncthreads();
  PLS_QSTAMP(13);
  for (int polish = 0; polish < 2; ++polish) {
    double v = 0.0;
    if (tid < M) {
      for (int k = 0; k < M; ++k) v += S0[(size_t)k * M + tid] * qv[k];        // S0 is symmetric to the bit
      scr[tid] = v;
    }
    __syncthreads();
    const double sv = lane < M ? scr[lane] : 0.0;
    const double nn = sqrt(wave_sum(sv * sv));
    if (tid < M) qv[tid] = nn > 0.0 ? v / nn : 0.0;
    __syncthreads();
  }
}

// dst[p] = sum over the S slices of x[t * len + p], p in [0, len): a thread per element, the
// loads of 16 slices in flight at a time, summed in slice order
template <bool SLICED>
__device__ __forceinline__ void xreduce(const double *x, int len, int S, double *dst, int tid) {
  if constexpr (!SLICED) {
    for (int p = tid; p < len; p += PLS_THREADS) dst[p] = x[p];
  } else {
    for (int p = tid; p < len; p += PLS_THREADS) {
      double acc = 0.0;
      for (int t0 = 0; t0 < S; t0 += 16) {
        double v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = (t0 + u < S) ? xget<true>(&x[(size_t)(t0 + u) * len + p]) : 0.0;
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += v[u];
      }
      dst[p] = acc;
    }
  }
}

template <typename T, bool XRES, bool SLICED>
__global__ __launch_bounds__(PLS_THREADS) void pls_kernel(const PlsArgs a) {
#ifdef CVM_STAMPS
  unsigned long long stamp_ = __builtin_readcyclecounter();
#endif
  extern __shared__ __attribute__((aligned(16))) unsigned char pls_smem[];
  if (a.run_if && *a.run_if != 1) return;
  if ((int)blockIdx.x == a.skip_block) return;
  const int K = a.K, M = a.M, A = a.A, S = SLICED ? a.S : 1;
  const int f = blockIdx.x / S, s = blockIdx.x - f * S;
  const int rows = a.rows;
  const int k0 = s * rows;
  const int n = (k0 + rows <= K) ? rows : (K - k0 > 0 ? K - k0 : 0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int MM = M * M;
  const int rp = (rows + 1) & ~1;
  const int yst = a.y_in_lds ? (M | 1) : M;                    // odd pitch in LDS: rows on distinct banks

  double *rl = reinterpret_cast<double *>(pls_smem);           // r, all K rows
  double *wl = rl + ((K + 1) & ~1);                            // w, r, u of the slice
  double *rs = wl + rp;
  double *us = rs + rp;
  double *S0 = us + rp;                                        // M x M: XTY^T XTY and two squarings
  double *Ba = S0 + MM;
  double *Bb = Ba + MM;
  double *ms = Bb + MM;                                        // waves x 256: per-wave MFMA partials
  double *qx = ms + PLS_NW * 256;                                   // 1 + M: tTt, XTY^T r / q
  double *cx = qx + PLS_MAXM + 2;                              // 1 + A: |w|^2, P^T w
  double *red = cx + ((A + 2) & ~1);                           // one per wave
  int *lflag = reinterpret_cast<int *>(red + PLS_NW);
  double *xl = red + PLS_NW + 2;                                        // !SLICED: the exchange buffers
  double *ys = xl + (SLICED ? 0 : (size_t)MM + (1 + A) + (1 + M) + ((MM + A + M) & 1));
  double *pl = ys + (a.y_in_lds ? (((size_t)rows * yst + 1) & ~(size_t)1) : 0);   // [A][rp] x 2 (optional)
  T *xs = reinterpret_cast<T *>(pl + (a.pr_in_lds ? 2 * (size_t)A * rp : 0));   // rows x K (XRES)
  double *qv = qx + 1, *cj = cx + 1;

  const T *XTX = (const T *)a.XTX + (size_t)f * K * K;
  const T *XTY = (const T *)a.XTY + (size_t)f * K * M;
  double *Y = a.y_in_lds ? ys : a.Yw + (size_t)f * K * M + (size_t)k0 * M;   // the slice's deflated XTY
  double *Bw = a.Bw + (size_t)f * K * M + (size_t)k0 * M;
  // the slice's columns of P^T and R^T: component j, row k at [j * pst + k]
  double *Pp = a.pr_in_lds ? pl : a.Pw + (size_t)f * K * A + k0;
  double *Rp = a.pr_in_lds ? pl + (size_t)A * rp : a.Rw + (size_t)f * K * A + k0;
  const size_t pst = a.pr_in_lds ? (size_t)rp : (size_t)K;
  double *xch = SLICED ? a.xch + (size_t)f * pls_xch_len(K, M, A, S) : xl;
  double *x_S = xch;                                           // [S][MM]
  double *x_2 = x_S + (size_t)S * MM;                          // [S][1 + c] (room for 1 + A)
  double *x_4 = x_2 + (size_t)S * (1 + A);                     // [S][1 + M]
  double *x_r = SLICED ? x_4 + (size_t)S * (1 + M) : rl;       // [K]
  unsigned *cnt = a.cnt + f;
  unsigned target = 0;
  const bool vec_ok = (K % (16 / (int)sizeof(T))) == 0;

  // ---- prologue: working copies of the slice -------------------------------------------
  for (int e = tid; e < n * M; e += PLS_THREADS) {
    const int k = e / M, j = e - k * M;
    Y[(size_t)k * yst + j] = (double)XTY[(size_t)k0 * M + e];
    Bw[e] = 0.0;
  }
  if (XRES) {
    const T *src = XTX + (size_t)k0 * K;
    for (size_t e = tid; e < (size_t)n * K; e += PLS_THREADS) xs[e] = src[e];
  }
  __syncthreads();
  PLS_STAMP(0);

  int fit = 0;
  for (int c = 0; c < A; ++c) {
    if (M > 1) {
      // ---- 1: partial XTY^T XTY of the slice's rows ----------------------------------------
      if (M <= 16) pls_gram_phase<1, SLICED>(Y, yst, n, M, ms, x_S + (size_t)s * MM, tid);
      else if (M <= 32) pls_gram_phase<2, SLICED>(Y, yst, n, M, ms, x_S + (size_t)s * MM, tid);
      else if (M <= 48) pls_gram_phase_tri<3, SLICED>(Y, yst, n, M, ms, x_S + (size_t)s * MM, tid);
      else pls_gram_phase_tri<4, SLICED>(Y, yst, n, M, ms, x_S + (size_t)s * MM, tid);
      PLS_STAMP(1);
      if (!fold_barrier<SLICED>(cnt, target, S, a.status, lflag, a.spin_limit)) return;
      PLS_STAMP(2);
      // ---- 2a: dominant eigenvector q of the M x M sum, by repeated squaring -----------------
      // B_0 = S / trace; B_{t+1} = B_t^2 / trace(B_t^2): trace(B_t^2) = sum lambda^2 (trace 1
      // before) reaches 1 when B_t is numerically rank one.
      xreduce<SLICED>(x_S, MM, S, S0, tid);
      __syncthreads();
#ifdef CVM_STAMPS
      if (blockIdx.x == 0 && tid == 0) g_pls_stamps[14] += __builtin_readcyclecounter() - stamp_;
      pls_dominant_q(S0, Ba, Bb, ms, qv, M, tid, &stamp_);
#else
      pls_dominant_q(S0, Ba, Bb, ms, qv, M, tid);
#endif
    }
    PLS_STAMP(3);
    // ---- 2b: w of the slice (not normalised yet), its partial norm and partial P^T w -------
    double nrm2 = 0.0;
    for (int k = tid; k < n; k += PLS_THREADS) {
      double v;
      if (M == 1) v = Y[k];
      else {
        v = 0.0;
        for (int j = 0; j < M; ++j) v += Y[(size_t)k * yst + j] * qv[j];
      }
      wl[k] = v;
      nrm2 += v * v;
    }
    nrm2 = block_sum(nrm2, red);
    if (tid == 0) xput<SLICED>(&x_2[(size_t)s * (1 + c)], nrm2);
    for (int j0 = wave * 4; j0 < c; j0 += 4 * PLS_NW) {
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      for (int k = lane; k < n; k += 256) {      // sixteen loads of P in flight per lane
        double pv[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            pv[u][q] = (k + 64 * u < n && j0 + q < c) ? Pp[(size_t)(j0 + q) * pst + k + 64 * u] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double wv = (k + 64 * u < n) ? wl[k + 64 * u] : 0.0;
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] += pv[u][q] * wv;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double v = wave_sum(acc[q]);
        if (lane == 0 && j0 + q < c) xput<SLICED>(&x_2[(size_t)s * (1 + c) + 1 + j0 + q], v);
      }
    }
    PLS_STAMP(4);
    if (!fold_barrier<SLICED>(cnt, target, S, a.status, lflag, a.spin_limit)) return;
    PLS_STAMP(5);
    // ---- 3: normalise, r = w - R (P^T w) -----------------------------------------------------
    xreduce<SLICED>(x_2, 1 + c, S, cx, tid);
    __syncthreads();
    const double nrm = sqrt(cx[0]);
    if (!(nrm > a.eps)) break;                                 // nothing left to extract (uniform)
    for (int j = tid; j < c; j += PLS_THREADS) cj[j] = cj[j] / nrm;
    __syncthreads();
    for (int k = tid; k < n; k += PLS_THREADS) {
      const double w = wl[k] / nrm;
      double corr = 0.0;
      for (int j0 = 0; j0 < c; j0 += 8) {
        double rv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) rv[u] = (j0 + u < c) ? Rp[(size_t)(j0 + u) * pst + k] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) if (j0 + u < c) corr += rv[u] * cj[j0 + u];
      }
      const double r = w - corr;
      wl[k] = w;
      rs[k] = r;
      Rp[(size_t)c * pst + k] = r;
      xput<SLICED>(&x_r[k0 + k], r);
      if (a.W) ((T *)a.W)[((size_t)f * K + k0 + k) * A + c] = (T)w;
      if (a.R) ((T *)a.R)[((size_t)f * K + k0 + k) * A + c] = (T)r;
    }
    PLS_STAMP(6);
    if (!fold_barrier<SLICED>(cnt, target, S, a.status, lflag, a.spin_limit)) return;
    PLS_STAMP(7);
    // ---- 4: u = XTX[slice, :] r, partial r^T u and partial XTY^T r ---------------------------
    if constexpr (SLICED) {
      for (int k = tid; k < K; k += PLS_THREADS) rl[k] = xget<true>(&x_r[k]);
      __syncthreads();
    }
    {
      const T *xb = XRES ? xs : XTX + (size_t)k0 * K;
      constexpr int NW = PLS_THREADS / 64;
      int done = 0;
      if (vec_ok) {
        done = n & ~3;
        for (int i0 = wave * 4; i0 < done; i0 += 4 * NW) {
          double acc[4] = {0.0, 0.0, 0.0, 0.0};
          matvec_rows<T, 4>(xb + (size_t)i0 * K, (size_t)K, K, rl, lane, acc);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const double v = wave_sum(acc[q]);
            if (lane == 0) us[i0 + q] = v;
          }
        }
      }
      // ragged rows (and all rows of an unaligned K): one at a time, scalar loads
      for (int i = done + wave; i < n; i += NW) {
        double acc = 0.0;
        for (int k = lane; k < K; k += 64) acc += (double)xb[(size_t)i * K + k] * rl[k];
        acc = wave_sum(acc);
        if (lane == 0) us[i] = acc;
      }
    }
    __syncthreads();
    PLS_STAMP(8);
    double tt = 0.0;
    for (int k = tid; k < n; k += PLS_THREADS) tt += rs[k] * us[k];
    tt = block_sum(tt, red);
    if (tid == 0) xput<SLICED>(&x_4[(size_t)s * (1 + M)], tt);
    for (int j = wave; j < M; j += PLS_THREADS / 64) {
      double acc = 0.0;
      for (int k = lane; k < n; k += 64) acc += Y[(size_t)k * yst + j] * rs[k];
      acc = wave_sum(acc);
      if (lane == 0) xput<SLICED>(&x_4[(size_t)s * (1 + M) + 1 + j], acc);
    }
    PLS_STAMP(9);
    if (!fold_barrier<SLICED>(cnt, target, S, a.status, lflag, a.spin_limit)) return;
    PLS_STAMP(10);
    // ---- 5: p, q, deflation of the slice, B ---------------------------------------------------
    xreduce<SLICED>(x_4, 1 + M, S, qx, tid);
    __syncthreads();
    const double tTt = qx[0];
    const double qmine = tid < M ? qv[tid] / tTt : 0.0;
    if (tid < M) {
      qv[tid] = qmine;
      if (a.Q && s == 0) ((T *)a.Q)[((size_t)f * M + tid) * A + c] = (T)qmine;
    }
    for (int k = tid; k < n; k += PLS_THREADS) {
      const double p = us[k] / tTt;
      us[k] = p;
      Pp[(size_t)c * pst + k] = p;
      if (a.P) ((T *)a.P)[((size_t)f * K + k0 + k) * A + c] = (T)p;
    }
    __syncthreads();
    T *Bout = (T *)a.B + (((size_t)f * A + c) * K + k0) * M;
    for (int e0 = tid; e0 < n * M; e0 += 8 * PLS_THREADS) {
      double bw[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * PLS_THREADS;
        bw[u] = e < n * M ? Bw[e] : 0.0;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int e = e0 + u * PLS_THREADS;
        if (e < n * M) {
          const int k = e / M, j = e - k * M;
          Y[(size_t)k * yst + j] = Y[(size_t)k * yst + j] - (us[k] * qv[j]) * tTt;
          const double b = bw[u] + rs[k] * qv[j];
          Bw[e] = b;
          Bout[e] = (T)b;
        }
      }
    }
    __syncthreads();
    PLS_STAMP(11);
    fit = c + 1;
  }
  // components that could not be extracted stay zero
  for (int c = fit; c < A; ++c) {
    T *Bout = (T *)a.B + (((size_t)f * A + c) * K + k0) * M;
    for (int e = tid; e < n * M; e += PLS_THREADS) Bout[e] = (T)0;
    for (int k = tid; k < n; k += PLS_THREADS) {
      if (a.W) ((T *)a.W)[((size_t)f * K + k0 + k) * A + c] = (T)0;
      if (a.R) ((T *)a.R)[((size_t)f * K + k0 + k) * A + c] = (T)0;
      if (a.P) ((T *)a.P)[((size_t)f * K + k0 + k) * A + c] = (T)0;
    }
    if (a.Q && s == 0 && tid < M) ((T *)a.Q)[((size_t)f * M + tid) * A + c] = (T)0;
  }
  if (s == 0 && tid == 0) a.n_fit[f] = fit;
}

// ----------------------------------------------------------------------------------
// pls_rep_kernel (round 3): ONE per-fold barrier per component instead of four.
// Everything of a component except u = XTX r is O(K (M + A)) work on small data -- the deflated
// XTY (K x M), P, R, w, r -- and pls_kernel spends most of its 47 us per component (C3) waiting
// for the other slices of its fold four times so that each may own only its rows of them.  Here
// every slice of a fold keeps the WHOLE deflated XTY in LDS and whole P^T, R^T of its own in a
// private global buffer (L2), and computes w, P^T w, r, q and the deflation redundantly -- the same
// instructions on the same numbers, so all slices agree bit for bit -- while XTX alone stays cut
// in row slices: a slice forms its rows of u, publishes them, and the one barrier of the component
// is the all-gather of u.  The slices of a fold are placed on ONE XCD (block b runs on XCD b % 8:
// tools/dispatch_probe.hip), where a device-coherent round trip costs 0.5-0.75 us instead of
// 1.15-1.2 us across XCDs (tools/xcd_pingpong.hip); placement is a matter of speed only, the
// accesses stay device-coherent.
// ----------------------------------------------------------------------------------
size_t pls_rep_lds_bytes(int K, int M, int A, int rows) {
  const size_t yst = (size_t)(M | 1);
  size_t d = 3 * (size_t)((K + 1) & ~1) + 3 * (size_t)M * M + PLS_NW * 256 + (PLS_MAXM + 2) + ((A + 2) & ~1) +
             PLS_NW + 2 + (((size_t)K * yst + 1) & ~(size_t)1) + (((size_t)rows * M + 1) & ~(size_t)1);
  return d * 8;
}

template <typename T>
__global__ __launch_bounds__(PLS_THREADS) void pls_rep_kernel(const PlsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pls_smem[];
  if ((int)blockIdx.x == a.skip_block) return;
  const int K = a.K, M = a.M, A = a.A, S = a.S;
  const int xcd = blockIdx.x & 7, bi = blockIdx.x >> 3;
  const int f = (bi / S) * 8 + xcd, s = bi - (bi / S) * S;      // fold g of the launch lives on XCD g % 8
  if (f >= a.nf) return;                                       // padding blocks of the XCD-affine grid
  const int rows = a.rows;
  const int k0 = s * rows;
  const int n = (k0 + rows <= K) ? rows : (K - k0 > 0 ? K - k0 : 0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int MM = M * M;
  const int Kp = (K + 1) & ~1;
  const int yst = M | 1;

  double *rl = reinterpret_cast<double *>(pls_smem);           // r, w, u: all K rows
  double *wl = rl + Kp;
  double *ul = wl + Kp;
  double *S0 = ul + Kp;
  double *Ba = S0 + MM;
  double *Bb = Ba + MM;
  double *ms = Bb + MM;
  double *qx = ms + PLS_NW * 256;
  double *cx = qx + PLS_MAXM + 2;
  double *red = cx + ((A + 2) & ~1);
  int *lflag = reinterpret_cast<int *>(red + PLS_NW);
  double *Yl = red + PLS_NW + 2;                               // the whole deflated XTY, pitch yst
  double *Bl = Yl + (((size_t)K * yst + 1) & ~(size_t)1);      // running B of the slice's rows
  double *qv = qx + 1, *cj = cx + 1;

  const T *XTX = (const T *)a.XTX + (size_t)f * K * K;
  const T *XTY = (const T *)a.XTY + (size_t)f * K * M;
  double *Pp = a.prw + ((size_t)f * S + s) * 2 * (size_t)A * K;   // [A][K], this slice's own copy
  double *Rp = Pp + (size_t)A * K;
  double *xu = a.xu + (size_t)f * 2 * K;
  unsigned *cnt = a.cnt + f;
  unsigned target = 0;
  const bool vec_ok = (K % (16 / (int)sizeof(T))) == 0;

  for (int e = tid; e < K * M; e += PLS_THREADS) {
    const int k = e / M, j = e - k * M;
    Yl[(size_t)k * yst + j] = (double)XTY[e];
  }
  for (int e = tid; e < n * M; e += PLS_THREADS) Bl[e] = 0.0;
  __syncthreads();

  int fit = 0;
  for (int c = 0; c < A; ++c) {
    if (M > 1) {
      // ---- XTY^T XTY over all rows, its dominant eigenvector q (as in pls_kernel, nothing traded)
      if (M <= 16) pls_gram_phase<1, false>(Yl, yst, K, M, ms, S0, tid);
      else if (M <= 32) pls_gram_phase<2, false>(Yl, yst, K, M, ms, S0, tid);
      else if (M <= 48) pls_gram_phase_tri<3, false>(Yl, yst, K, M, ms, S0, tid);
      else pls_gram_phase_tri<4, false>(Yl, yst, K, M, ms, S0, tid);
      __syncthreads();
      pls_dominant_q(S0, Ba, Bb, ms, qv, M, tid);
    }
    // ---- w (all rows), its norm, P^T w ---------------------------------------------------------
    double nrm2 = 0.0;
    for (int k = tid; k < K; k += PLS_THREADS) {
      double v;
      if (M == 1) v = Yl[k];
      else {
        v = 0.0;
        for (int j = 0; j < M; ++j) v += Yl[(size_t)k * yst + j] * qv[j];
      }
      wl[k] = v;
      nrm2 += v * v;
    }
    nrm2 = block_sum(nrm2, red);                               // (ends behind a barrier: wl is visible)
    const double nrm = sqrt(nrm2);
    if (!(nrm > a.eps)) break;                                 // every slice sees the same number
    for (int j0 = wave * 4; j0 < c; j0 += 4 * PLS_NW) {
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      for (int k = lane; k < K; k += 256) {
        double pv[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            pv[u][q] = (k + 64 * u < K && j0 + q < c) ? Pp[(size_t)(j0 + q) * K + k + 64 * u] : 0.0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double wv = (k + 64 * u < K) ? wl[k + 64 * u] : 0.0;
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q] += pv[u][q] * wv;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double v = wave_sum(acc[q]);
        if (lane == 0 && j0 + q < c) cj[j0 + q] = v / nrm;
      }
    }
    __syncthreads();
    // ---- r = w / |w| - R (P^T w) -----------------------------------------------------------------
    for (int k = tid; k < K; k += PLS_THREADS) {
      const double w = wl[k] / nrm;
      double corr = 0.0;
      for (int j0 = 0; j0 < c; j0 += 8) {
        double rv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) rv[u] = (j0 + u < c) ? Rp[(size_t)(j0 + u) * K + k] : 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) if (j0 + u < c) corr += rv[u] * cj[j0 + u];
      }
      const double r = w - corr;
      rl[k] = r;
      Rp[(size_t)c * K + k] = r;
      if (k >= k0 && k < k0 + n) {
        if (a.W) ((T *)a.W)[((size_t)f * K + k) * A + c] = (T)w;
        if (a.R) ((T *)a.R)[((size_t)f * K + k) * A + c] = (T)r;
      }
    }
    __syncthreads();
    // ---- the slice's rows of u = XTX r, published ----------------------------------------------
    double *xuc = xu + (size_t)(c & 1) * K;
    {
      const T *xb = XTX + (size_t)k0 * K;
      int done = 0;
      if (vec_ok) {
        done = n & ~3;
        for (int i0 = wave * 4; i0 < done; i0 += 4 * PLS_NW) {
          double acc[4] = {0.0, 0.0, 0.0, 0.0};
          matvec_rows<T, 4>(xb + (size_t)i0 * K, (size_t)K, K, rl, lane, acc);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const double v = wave_sum(acc[q]);
            if (lane == 0) xput<true>(&xuc[k0 + i0 + q], v);
          }
        }
      }
      for (int i = done + wave; i < n; i += PLS_NW) {
        double acc = 0.0;
        for (int k = lane; k < K; k += 64) acc += (double)xb[(size_t)i * K + k] * rl[k];
        acc = wave_sum(acc);
        if (lane == 0) xput<true>(&xuc[k0 + i], acc);
      }
    }
    if (!fold_barrier<true>(cnt, target, S, a.status, lflag, a.spin_limit)) return;
    for (int k = tid; k < K; k += PLS_THREADS) ul[k] = xget<true>(&xuc[k]);
    __syncthreads();
    // ---- t^T t, q, p, deflation (all rows), B (the slice's rows) -------------------------------
    double tt = 0.0;
    for (int k = tid; k < K; k += PLS_THREADS) tt += rl[k] * ul[k];
    tt = block_sum(tt, red);
    for (int j = wave; j < M; j += PLS_NW) {
      double acc = 0.0;
      for (int k = lane; k < K; k += 64) acc += Yl[(size_t)k * yst + j] * rl[k];
      acc = wave_sum(acc);
      if (lane == 0) {
        const double qm = acc / tt;
        qv[j] = qm;
        if (a.Q && s == 0) ((T *)a.Q)[((size_t)f * M + j) * A + c] = (T)qm;
      }
    }
    for (int k = tid; k < K; k += PLS_THREADS) {
      const double p = ul[k] / tt;
      ul[k] = p;
      Pp[(size_t)c * K + k] = p;
      if (a.P && k >= k0 && k < k0 + n) ((T *)a.P)[((size_t)f * K + k) * A + c] = (T)p;
    }
    __syncthreads();
    for (int e = tid; e < K * M; e += PLS_THREADS) {
      const int k = e / M, j = e - k * M;
      Yl[(size_t)k * yst + j] = Yl[(size_t)k * yst + j] - (ul[k] * qv[j]) * tt;
    }
    T *Bout = (T *)a.B + (((size_t)f * A + c) * K + k0) * M;
    for (int e = tid; e < n * M; e += PLS_THREADS) {
      const int k = e / M, j = e - k * M;
      const double b = Bl[e] + rl[k0 + k] * qv[j];
      Bl[e] = b;
      Bout[e] = (T)b;
    }
    __syncthreads();
    fit = c + 1;
  }
  for (int c = fit; c < A; ++c) {
    T *Bout = (T *)a.B + (((size_t)f * A + c) * K + k0) * M;
    for (int e = tid; e < n * M; e += PLS_THREADS) Bout[e] = (T)0;
    for (int k = tid; k < n; k += PLS_THREADS) {
      if (a.W) ((T *)a.W)[((size_t)f * K + k0 + k) * A + c] = (T)0;
      if (a.R) ((T *)a.R)[((size_t)f * K + k0 + k) * A + c] = (T)0;
      if (a.P) ((T *)a.P)[((size_t)f * K + k0 + k) * A + c] = (T)0;
    }
    if (a.Q && s == 0 && tid < M) ((T *)a.Q)[((size_t)f * M + tid) * A + c] = (T)0;
  }
  if (s == 0 && tid == 0) a.n_fit[f] = fit;
}

// Before the launches of a call: the folds' barrier counters and the status word
__global__ void pls_zero_kernel(unsigned *cnt, int64_t n, int32_t *status) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) cnt[i] = 0u;
  if (i == 0) *status = 0;
}

// After the launches of a call: if any barrier timed out (status != 0: the slices of a fold were
// not co-resident), nothing of the call can be trusted -- every coefficient becomes NaN and every
// n_fit -1, so that a caller who does not read the status word cannot consume half-written models.
template <typename T>
__global__ void pls_poison_kernel(const int *status, T *B, size_t nB, T *W, T *P, T *R, size_t nW, T *Q, size_t nQ,
                                  int *n_fit, int64_t F) {
  if (*status == 0) return;
  const T nan = (T)__builtin_nan("");
  const size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x, step = (size_t)gridDim.x * blockDim.x;
  for (size_t i = i0; i < nB; i += step) B[i] = nan;
  for (size_t i = i0; i < nW; i += step) { if (W) W[i] = nan; if (P) P[i] = nan; if (R) R[i] = nan; }
  if (Q) for (size_t i = i0; i < nQ; i += step) Q[i] = nan;
  for (size_t i = i0; i < (size_t)F; i += step) n_fit[i] = -1;
}

// ---- host ---------------------------------------------------------------------------------
struct PlsPlan {
  int S, rows, folds_per_launch, y_in_lds, pr_in_lds, xres;
  size_t lds;
};

size_t pls_lds_bytes(int K, int M, int A, int rows, int sliced, int y_in_lds, int pr_in_lds, int xres, int esize) {
  const size_t rp = (rows + 1) & ~1;
  size_t d = ((K + 1) & ~1) + 3 * rp + 3 * (size_t)M * M + PLS_NW * 256 + (PLS_MAXM + 2) + ((A + 2) & ~1) + PLS_NW + 2;
  if (!sliced) d += (size_t)M * M + (1 + A) + (1 + M) + ((M * M + A + M) & 1);
  if (y_in_lds) d += ((size_t)rows * (M | 1) + 1) & ~(size_t)1;
  if (pr_in_lds) d += 2 * (size_t)A * rp;
  size_t b = d * 8;
  if (xres) b += (size_t)rows * K * esize;
  return b;
}

// Per component a slice streams rows*K*esize bytes of XTX at what one CU draws from L2/HBM, and,
// when a fold has several slices, passes the per-fold barrier (3 times for M == 1, else 4).
constexpr double PLS_CU_BYTES_PER_US = 50e3;    // one CU streaming from L2 / HBM (measured 27-50)
constexpr double PLS_LDS_BYTES_PER_US = 100e3;  // the slice resident in LDS (short rows: latency)
constexpr double PLS_BARRIER_US = 6.0;          // measured: ~14k cycles

bool make_pls_plan(int64_t F, int K, int M, int A, int esize, int cus, PlsPlan &p) {
  const int64_t Fe = F > 0 ? F : 1;
  int s_max = (Fe >= cus) ? 1 : (int)(cus / Fe);               // folds x S <= CUs: co-resident
  if (s_max > (K + 7) / 8) s_max = (K + 7) / 8;                // >= 8 rows per slice
  if (s_max < 1) s_max = 1;
  const int s_lim = K < cus ? K : cus;
  int best = 0;
  double best_t = 1e300;
  for (int S = 1; S <= s_lim; ++S) {
    if (S > s_max && best) break;                              // more slices than s_max only if the LDS demands it
    const int rows = (K + S - 1) / S;
    if ((K + rows - 1) / rows != S) continue;                  // the same cut as a smaller S
    if (pls_lds_bytes(K, M, A, rows, S > 1, 0, 0, 0, esize) > PLS_LDS_BUDGET) continue;
    const bool res = pls_lds_bytes(K, M, A, rows, S > 1, 0, 0, 1, esize) <= PLS_LDS_BUDGET;
    const double bytes = (double)rows * K * esize;
    const double t = bytes / (res ? PLS_LDS_BYTES_PER_US : PLS_CU_BYTES_PER_US) +
                     (S > 1 ? (M > 1 ? 4 : 3) * PLS_BARRIER_US : 0.0);
    if (t < best_t) { best_t = t; best = S; }
  }
  if (!best) return false;
  p.S = best;
  p.rows = (K + best - 1) / best;
  p.folds_per_launch = best == 1 ? (int)(Fe < (1 << 20) ? Fe : (1 << 20)) : cus / best;
  if (p.folds_per_launch < 1) return false;
  // what else stays in LDS, most valuable first: the slice of XTX, of the deflated XTY, of P and R
  p.xres = pls_lds_bytes(K, M, A, p.rows, best > 1, 0, 0, 1, esize) <= PLS_LDS_BUDGET;
  p.y_in_lds = pls_lds_bytes(K, M, A, p.rows, best > 1, 1, 0, p.xres, esize) <= PLS_LDS_BUDGET;
  p.pr_in_lds = pls_lds_bytes(K, M, A, p.rows, best > 1, p.y_in_lds, 1, p.xres, esize) <= PLS_LDS_BUDGET;
  p.lds = pls_lds_bytes(K, M, A, p.rows, best > 1, p.y_in_lds, p.pr_in_lds, p.xres, esize);
  return true;
}

int pls_cu_count() {
  static int cus = 0;
  if (!cus) {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess &&
        hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      cus = v;
    else
      cus = 256;
  }
  return cus;
}

// After the safe re-run: status 1 (a barrier timed out) becomes 2 (timed out, every fold recomputed by one
// workgroup each: the outputs are valid)
__global__ void pls_recovered_kernel(int32_t *status) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && *status == 1) *status = 2;
}

// The plan of the kernel that cannot wait for anybody: one workgroup per fold (S = 1).  False when a whole
// fold's state does not fit the LDS budget (then a timed-out launch can only be poisoned).
bool make_pls_plan_s1(int64_t F, int K, int M, int A, int esize, PlsPlan &p) {
  if (pls_lds_bytes(K, M, A, K, 0, 0, 0, 0, esize) > PLS_LDS_BUDGET) return false;
  const int64_t Fe = F > 0 ? F : 1;
  p.S = 1; p.rows = K;
  p.folds_per_launch = (int)(Fe < (1 << 20) ? Fe : (1 << 20));
  p.xres = pls_lds_bytes(K, M, A, K, 0, 0, 0, 1, esize) <= PLS_LDS_BUDGET;
  p.y_in_lds = pls_lds_bytes(K, M, A, K, 0, 1, 0, p.xres, esize) <= PLS_LDS_BUDGET;
  p.pr_in_lds = pls_lds_bytes(K, M, A, K, 0, p.y_in_lds, 1, p.xres, esize) <= PLS_LDS_BUDGET;
  p.lds = pls_lds_bytes(K, M, A, K, 0, p.y_in_lds, p.pr_in_lds, p.xres, esize);
  return true;
}

// pls_rep_kernel's cut: few folds (at most 8 per XCD with at least 4 slices each), the whole deflated
// XTY in LDS.  The slices of a fold sit on one XCD: per_x folds per XCD, S <= CUs-per-XCD / per_x.
struct PlsRepPlan { int S, rows, per_x, folds_per_launch; size_t lds; };
bool make_pls_rep_plan(int64_t F, int K, int M, int A, int esize, int cus, PlsRepPlan &p) {
  static const bool off = getenv("CVM_PLS_NO_REP") != nullptr;      // tests / comparisons: the four-barrier kernel
  const int per_xcd = cus / 8;
  // for the problems whose components are bound by the barriers, not by streaming XTX: at most 64
  // folds (one launch) and an XTX of at most 8 MB per fold (it stays in the L2 of the fold's XCD)
  if (off || F < 1 || F > 64 || per_xcd < 8 || (size_t)K * K * esize > ((size_t)8 << 20)) return false;
  const int64_t Fl = F;                                             // folds per launch
  const int per_x = (int)((Fl + 7) / 8);
  // (one CU per XCD is left out of the count: the slices spin on each other, and a CU that is busy with
  //  something else -- another stream, a masked CU -- must not leave a slice without a place to run)
  int S = (per_xcd - 1) / per_x;
  if (S > (K + 7) / 8) S = (K + 7) / 8;                             // >= 8 rows per slice
  if (S < 4) return false;                                          // many folds: one slice per fold is the better cut
  const int rows = (K + S - 1) / S;
  S = (K + rows - 1) / rows;
  const size_t lds = pls_rep_lds_bytes(K, M, A, rows);
  if (lds > PLS_LDS_BUDGET) return false;
  p.S = S; p.rows = rows; p.per_x = per_x; p.folds_per_launch = 8 * per_x; p.lds = lds;
  return true;
}

size_t pls_workspace_bytes(int64_t F, int K, int M, int A, int esize, int cus) {
  PlsPlan p;
  if (!make_pls_plan(F, K, M, A, esize, cus, p)) return 0;
  const size_t per_fold = (2 * (size_t)K * M + 2 * (size_t)K * A + pls_xch_len(K, M, A, p.S)) * 8 + 16;
  size_t need = (size_t)F * per_fold + 256;
  {   // the safe re-run's layout (S = 1) inside the same workspace
    const size_t s1 = (size_t)F * ((2 * (size_t)K * M + 2 * (size_t)K * A + pls_xch_len(K, M, A, 1)) * 8 + 16) + 256;
    if (s1 > need) need = s1;
  }
  PlsRepPlan r;
  if (make_pls_rep_plan(F, K, M, A, esize, cus, r)) {
    const size_t rep = (size_t)F * ((size_t)r.S * 2 * A * K + 2 * (size_t)K) * 8 + (size_t)F * 16 + 256;
    if (rep > need) need = rep;
  }
  return need;
}

// CVM_PLS_TEST_TIMEOUT (tests only): workgroup 0 of the sliced launches returns at once and the others give
// up after a few thousand spins, so that the safety net below is exercised
inline bool pls_test_timeout() {
  static const bool on = getenv("CVM_PLS_TEST_TIMEOUT") != nullptr;
  return on;
}

// Behind a launch whose workgroups wait for each other: if a barrier timed out (status 1: a slice was not
// resident -- another stream's kernel, a masked CU), every fold is recomputed by the kernel that waits for
// nobody (one workgroup per fold; it returns at once when status is 0) and status becomes 2; where a whole
// fold does not fit one workgroup's LDS the outputs are poisoned instead (NaN coefficients, n_fit -1).
template <typename T>
int pls_safety_net(const void *XTX, const void *XTY, int64_t F, int K, int M, int A, void *B, void *W, void *P, void *Q,
                   void *R, int32_t *n_fit, int32_t *status, void *ws, hipStream_t st) {
  PlsPlan p1;
  if (make_pls_plan_s1(F, K, M, A, (int)sizeof(T), p1)) {
    double *d = reinterpret_cast<double *>(ws);
    PlsArgs a;
    memset(&a, 0, sizeof(a));
    a.K = K; a.M = M; a.A = A; a.S = 1; a.rows = K; a.y_in_lds = p1.y_in_lds; a.pr_in_lds = p1.pr_in_lds;
    a.Yw = d; d += (size_t)F * K * M;
    a.Bw = d; d += (size_t)F * K * M;
    a.Pw = d; d += (size_t)F * K * A;
    a.Rw = d; d += (size_t)F * K * A;
    a.xch = d; d += (size_t)F * pls_xch_len(K, M, A, 1);
    a.cnt = reinterpret_cast<unsigned *>(d);
    a.status = status; a.run_if = status; a.spin_limit = 1L << 21; a.skip_block = -1;
    a.eps = sizeof(T) == 8 ? 2.220446049250313e-16 : 1.1920928955078125e-07;
    a.XTX = XTX; a.XTY = XTY; a.B = B; a.W = W; a.P = P; a.R = R; a.Q = Q; a.n_fit = n_fit;
    void (*kern)(const PlsArgs) = p1.xres ? pls_kernel<T, true, false> : pls_kernel<T, false, false>;
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p1.lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)F), dim3(PLS_THREADS), p1.lds, st, a);
    hipLaunchKernelGGL(pls_recovered_kernel, dim3(1), dim3(64), 0, st, status);
  } else {
    hipLaunchKernelGGL((pls_poison_kernel<T>), dim3(256), dim3(256), 0, st, (const int *)status, (T *)B,
                       (size_t)F * A * K * M, (T *)W, (T *)P, (T *)R, (size_t)F * K * A, (T *)Q, (size_t)F * M * A,
                       (int *)n_fit, F);
  }
  HIP_OK(hipGetLastError());
  return CVM_OK;
}

template <typename T>
int pls_fit_impl(const void *XTX, const void *XTY, int64_t F, int K, int M, int A, void *B, void *W, void *P,
                 void *Q, void *R, int32_t *n_fit, int32_t *status, void *ws, size_t ws_bytes, hipStream_t st) {
  const int cus = pls_cu_count();
  PlsPlan p;
  if (!make_pls_plan(F, K, M, A, sizeof(T), cus, p)) return fail(CVM_EINVAL, "cvm_pls_fit: K too large for the LDS plan%s");
  if (ws_bytes < pls_workspace_bytes(F, K, M, A, sizeof(T), cus)) return fail(CVM_EWORKSPACE, "cvm_pls_fit: workspace too small%s");
  if (F == 0) return CVM_OK;
  PlsRepPlan rp;
  if (make_pls_rep_plan(F, K, M, A, (int)sizeof(T), cus, rp)) {
    // few folds: replicated small state, one barrier per component (pls_rep_kernel)
    double *d = reinterpret_cast<double *>(ws);
    PlsArgs a;
    memset(&a, 0, sizeof(a));
    a.K = K; a.M = M; a.A = A; a.S = rp.S; a.rows = rp.rows; a.per_x = rp.per_x;
    a.prw = d; d += (size_t)F * rp.S * 2 * A * K;
    a.xu = d; d += (size_t)F * 2 * K;
    a.cnt = reinterpret_cast<unsigned *>(d);
    a.status = status;
    a.spin_limit = pls_test_timeout() ? (1L << 12) : (1L << 21);
    a.skip_block = pls_test_timeout() ? 0 : -1;
    a.eps = sizeof(T) == 8 ? 2.220446049250313e-16 : 1.1920928955078125e-07;
    hipLaunchKernelGGL(pls_zero_kernel, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, st, a.cnt, F, status);
    void (*kern)(const PlsArgs) = pls_rep_kernel<T>;
    HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)rp.lds));
    int per_cu = 0;
    HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), PLS_THREADS, rp.lds));
    if (per_cu < 1) return fail(CVM_ELAUNCH, "cvm_pls_fit: the replicated-state launch would not be resident on this device%s");
    for (int64_t f0 = 0; f0 < F; f0 += rp.folds_per_launch) {
      const int64_t nf = F - f0 < rp.folds_per_launch ? F - f0 : rp.folds_per_launch;
      PlsArgs b = a;
      b.nf = (int)nf;
      b.XTX = (const T *)XTX + (size_t)f0 * K * K;
      b.XTY = (const T *)XTY + (size_t)f0 * K * M;
      b.prw = a.prw + (size_t)f0 * rp.S * 2 * A * K;
      b.xu = a.xu + (size_t)f0 * 2 * K;
      b.cnt = a.cnt + f0;
      b.B = (T *)B + (size_t)f0 * A * K * M;
      b.W = W ? (T *)W + (size_t)f0 * K * A : nullptr;
      b.P = P ? (T *)P + (size_t)f0 * K * A : nullptr;
      b.R = R ? (T *)R + (size_t)f0 * K * A : nullptr;
      b.Q = Q ? (T *)Q + (size_t)f0 * M * A : nullptr;
      b.n_fit = n_fit + f0;
      // (fold g of the launch -> XCD g % 8)
      hipLaunchKernelGGL(kern, dim3((unsigned)(8 * rp.per_x * rp.S)), dim3(PLS_THREADS), rp.lds, st, b);
      HIP_OK(hipGetLastError());
    }
    return pls_safety_net<T>(XTX, XTY, F, K, M, A, B, W, P, Q, R, n_fit, status, ws, st);
  }
  double *d = reinterpret_cast<double *>(ws);
  PlsArgs a;
  memset(&a, 0, sizeof(a));
  a.K = K; a.M = M; a.A = A; a.S = p.S; a.rows = p.rows; a.y_in_lds = p.y_in_lds; a.pr_in_lds = p.pr_in_lds;
  a.Yw = d; d += (size_t)F * K * M;
  a.Bw = d; d += (size_t)F * K * M;
  a.Pw = d; d += (size_t)F * K * A;
  a.Rw = d; d += (size_t)F * K * A;
  a.xch = d; d += (size_t)F * pls_xch_len(K, M, A, p.S);
  a.cnt = reinterpret_cast<unsigned *>(d);
  a.status = status;
  a.spin_limit = pls_test_timeout() ? (1L << 12) : (1L << 21);
  a.skip_block = (p.S > 1 && pls_test_timeout()) ? 0 : -1;
  a.eps = sizeof(T) == 8 ? 2.220446049250313e-16 : 1.1920928955078125e-07;
  // the barrier counters and the status word, zeroed by one small kernel (two runtime fills cost
  // two 5 us kernels in front of a 0.9 ms fit)
  hipLaunchKernelGGL(pls_zero_kernel, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, st, a.cnt, F, status);
  void (*kern)(const PlsArgs) = p.S > 1 ? (p.xres ? pls_kernel<T, true, true> : pls_kernel<T, false, true>)
                                        : (p.xres ? pls_kernel<T, true, false> : pls_kernel<T, false, false>);
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
  if (p.S > 1) {
    // the slices of a fold wait for each other: a launch must fit on the device at once (the plan
    // sizes it for one workgroup per CU; make sure the runtime agrees before relying on it)
    int per_cu = 0;
    HIP_OK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(kern), PLS_THREADS, p.lds));
    const int64_t resident = (int64_t)per_cu * cus;
    const int64_t biggest = (F < p.folds_per_launch ? F : p.folds_per_launch) * p.S;
    if (per_cu < 1 || biggest > resident)
      return fail(CVM_ELAUNCH, "cvm_pls_fit: a sliced launch would not be co-resident on this device%s");
  }
  for (int64_t f0 = 0; f0 < F; f0 += p.folds_per_launch) {
    const int64_t nf = F - f0 < p.folds_per_launch ? F - f0 : p.folds_per_launch;
    PlsArgs b = a;
    b.XTX = (const T *)XTX + (size_t)f0 * K * K;
    b.XTY = (const T *)XTY + (size_t)f0 * K * M;
    b.Yw = a.Yw + (size_t)f0 * K * M;
    b.Bw = a.Bw + (size_t)f0 * K * M;
    b.Pw = a.Pw + (size_t)f0 * K * A;
    b.Rw = a.Rw + (size_t)f0 * K * A;
    b.xch = a.xch + (size_t)f0 * pls_xch_len(K, M, A, p.S);
    b.cnt = a.cnt + f0;
    b.B = (T *)B + (size_t)f0 * A * K * M;
    b.W = W ? (T *)W + (size_t)f0 * K * A : nullptr;
    b.P = P ? (T *)P + (size_t)f0 * K * A : nullptr;
    b.R = R ? (T *)R + (size_t)f0 * K * A : nullptr;
    b.Q = Q ? (T *)Q + (size_t)f0 * M * A : nullptr;
    b.n_fit = n_fit + f0;
    hipLaunchKernelGGL(kern, dim3((unsigned)(nf * p.S)), dim3(PLS_THREADS), p.lds, st, b);
    HIP_OK(hipGetLastError());
  }
  if (p.S > 1) return pls_safety_net<T>(XTX, XTY, F, K, M, A, B, W, P, Q, R, n_fit, status, ws, st);
  return CVM_OK;
}

// ----------------------------------------------------------------------------------
// Validation errors of the folds' PLS models (round 3): the last step of a cross-validation on the
// device.  For fold f, model with a + 1 components, response m:
//   sse[f][a][m] = sum over the fold's validation rows i of
//                  w_i ( ((x_i - muX_f) / sdX_f) . B[f][a][:, m] * sdY_f[m] + muY_f[m] - y_im )^2
//   wsum[f]      = sum w_i
// (centre / scale vectors are the fold stage's statistics outputs; a NULL one means 0 / 1).
// One workgroup = 64 validation rows x 64 of the A M columns (a, m): the standardised rows and the
// coefficient columns go through LDS 16 k at a time, wave w owns row tile w and four MFMA column
// tiles; the squared errors are summed over the rows in a fixed order (registers -> lanes ->
// waves -> a second small kernel over the row chunks: no float atomics).
// ----------------------------------------------------------------------------------
constexpr int SSE_ROWS = 64, SSE_KS = 16, SSE_ZP = 80, SSE_MAXNT = 6;   // (Z pitch 80: conflict-free MFMA operand reads)
#ifdef CVM_STAMPS
__device__ unsigned long long g_sse_stamps[8];
#define SSE_STAMP(i) do { if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) { const unsigned long long now_ = __builtin_readcyclecounter(); g_sse_stamps[i] += now_ - sstamp_; sstamp_ = now_; } } while (0)
#else
#define SSE_STAMP(i)
#endif
struct SseArgs {
  const void *X, *Y, *w, *muX, *sdX, *muY, *sdY, *B;
  const int64_t *idx, *offs;
  int K, M, A, n_chunks;        // n_chunks: row chunks of the longest fold
  int st_in_lds;                // the fold's K means and reciprocal standard deviations fit in LDS next to the stages
  double *part;                 // [F][n_chunks][A M] partial sums, [F][n_chunks] weight sums behind them
  double *sse, *wsum;
  int64_t F;
};

// Workgroup = 64 validation rows x W = 64 NT of the A M columns (a, m); wave w owns the 16 NT columns
// from 16 NT w on, for all 64 rows: 4 x NT MFMA tiles, 4 A and NT B fragment reads per 4 NT MFMAs.
// One workgroup per CU (the two LDS stages of B take most of it), one wave per SIMD: the matrix
// pipe of a SIMD is never shared with another wave's vector arithmetic (a VALU instruction behind a
// stream of float64 MFMAs of ANOTHER wave waits hundreds of cycles, tools/dma_vs_mfma.hip -- the
// round-3 kernel, 64 x 64 tiles at several workgroups per CU, stood at 24 TFLOP/s), the next
// stage's rows and coefficients are requested before the stage's MFMAs and written to the other
// LDS buffer after them, one barrier per 16 k.  Rows are standardised on the way to LDS with the
// reciprocal standard deviation (v_rcp + one Newton step: within 1 ulp of the division).
// V: elements per global load (16 bytes' worth when K and M are multiples of that and the arrays are
// aligned, else 1).  A vector-memory instruction costs the issuing wave hundreds of cycles on a busy CU
// (tools/dma_issue.hip) and these waves also feed the matrix cores: 32 eight-byte loads per thread and
// stage held the kernel at 25 TFLOP/s; with 16-byte loads and the statistics in LDS a stage takes 12.
template <typename T, int NT, int V>
__global__ __launch_bounds__(256) void pls_sse_kernel(const SseArgs a) {
  constexpr int W = 64 * NT, BP = W + 16;                     // LDS pitch of B: conflict-free fragment reads
#ifdef CVM_STAMPS
  unsigned long long sstamp_ = __builtin_readcyclecounter();
#endif
  const int K = a.K, M = a.M, C = a.A * M;
  const int chunk = blockIdx.x, cg = blockIdx.y, f = blockIdx.z;
  const int64_t o0 = a.offs[f];
  const int n = (int)(a.offs[f + 1] - o0);
  const int r0 = chunk * SSE_ROWS;
  double *part = a.part + ((size_t)f * a.n_chunks + chunk) * C;
  double *wpart = a.part + (size_t)a.F * a.n_chunks * C + (size_t)f * a.n_chunks + chunk;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lk = lane >> 4, lc = lane & 15;
  const int c0 = cg * W;
  const int Cg = C - c0 < W ? C - c0 : W;                     // columns of this group
  if (r0 >= n) {                                              // past this fold's rows: zeros
    for (int c = tid; c < Cg; c += 256) part[c0 + c] = 0.0;
    if (cg == 0 && tid == 0) *wpart = 0.0;
    return;
  }
  extern __shared__ __attribute__((aligned(16))) unsigned char sse_smem[];
  T *Zs = reinterpret_cast<T *>(sse_smem);                    // [2][KS][ZP], [k][row]: the A operand's lanes run over rows
  T *Bs = Zs + 2 * SSE_KS * SSE_ZP;                           // [2][KS][BP], [k][column]
  __shared__ int64_t rows[SSE_ROWS];
  __shared__ double wl[SSE_ROWS];
  const T *X = (const T *)a.X, *Y = (const T *)a.Y, *Wt = (const T *)a.w;
  const T *muX = a.muX ? (const T *)a.muX + (size_t)f * K : nullptr;
  const T *sdX = a.sdX ? (const T *)a.sdX + (size_t)f * K : nullptr;
  const T *Bf = (const T *)a.B + (size_t)f * a.A * K * M;
  if (tid < SSE_ROWS) {
    const bool ok = r0 + tid < n;
    const int64_t r = ok ? a.idx[o0 + r0 + tid] : 0;
    rows[tid] = r;
    wl[tid] = ok ? (Wt ? (double)Wt[r] : 1.0) : 0.0;
  }
  __syncthreads();
  typedef typename MF<T>::acc_t acc_t;
  acc_t acc[4][NT];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[rt][t] = (acc_t){0, 0, 0, 0};
  // staging maps.  Z: thread -> (row zr = tid / 4, k quad zq = tid % 4): four CONSECUTIVE k of one
  // row (a wave instruction touches 16 rows' lines once), 4 / V loads.  B: piece e = tid + 256 i of
  // the stage's 16 x (W / V) pieces of V columns: k = e / (W / V), columns (e % (W / V)) V ...
  // All loads are branch-free -- rows past the fold's end re-read row 0, k past K re-reads the
  // last piece, columns past the group's end re-read column 0, and the values are zeroed in
  // `store`: a per-thread branch around a load makes the compiler wait for everything in flight.
  typedef T vec_t __attribute__((ext_vector_type(V)));
  constexpr int ZL = 4 / V < 1 ? 1 : 4 / V;                   // Z loads per thread and stage
  constexpr int ZV = 4 / ZL;                                  // elements per Z load (V, or 4 when V > 4 never happens)
  static_assert(V == 1 || V == 2 || V == 4, "V is 1 or 16 bytes' worth");
  constexpr int WP = W / V;                                   // pieces per k row
  constexpr int NB = SSE_KS * WP / 256;                       // B loads per thread and stage
  const int zr = tid >> 2, zq = tid & 3;
  const bool zok = r0 + zr < n;
  const T *xrow = X + rows[zr] * (int64_t)K;
  const bool has_mu = muX != nullptr, has_sd = sdX != nullptr;
  // statistics: (mean, 1 / sd) of all K columns in LDS when they fit, else read per stage
  double *stl = reinterpret_cast<double *>(Bs + 2 * SSE_KS * BP);
  const bool st_lds = a.st_in_lds != 0;
  if (st_lds) {
    for (int k = tid; k < K; k += 256) {
      stl[2 * k] = has_mu ? (double)muX[k] : 0.0;
      stl[2 * k + 1] = has_sd ? 1.0 / (double)sdX[k] : 1.0;
    }
  }
  const T *mup = has_mu ? muX : xrow, *sdp = has_sd ? sdX : xrow;   // (absent statistics: a harmless second read of the row)
  size_t boff[NB];
  int blds[NB];
  bool bok[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int e = tid + 256 * i;
    const int kk = e / WP, col = (e - kk * WP) * V;
    bok[i] = c0 + col < C;                                    // (C is a multiple of V in the vector build)
    const int gcol = bok[i] ? c0 + col : 0;
    const int ba = gcol / M, bm = gcol - ba * M;
    boff[i] = ((size_t)ba * K + kk) * M + bm;
    blds[i] = kk * BP + col;
  }
  vec_t zv[ZL], zm[ZL], zs[ZL], bv[NB];
  auto loadZ = [&](int j, int k0) {
    const int k = k0 + 4 * zq + ZV * j;
    const int kc = k < K ? k : K - ZV;
    zv[j] = *reinterpret_cast<const vec_t *>(xrow + kc);
    if (!st_lds) {
      zm[j] = *reinterpret_cast<const vec_t *>(mup + kc);
      zs[j] = *reinterpret_cast<const vec_t *>(sdp + kc);
    }
  };
  auto loadB = [&](int i, int k0) {
    // (k0 + kk < K except in the last stage: there the row K - 1 is re-read and zeroed in storeB)
    const int kk = (tid + 256 * i) / WP;
    const size_t ko = (size_t)(k0 + kk < K ? k0 : K - 1 - kk) * M;
    bv[i] = *reinterpret_cast<const vec_t *>(Bf + boff[i] + ko);
  };
  auto storeZ = [&](int j, int k0, int buf) {
    T *Zb = Zs + buf * SSE_KS * SSE_ZP;
#pragma unroll
    for (int e = 0; e < ZV; ++e) {
      const int kq = 4 * zq + ZV * j + e, k = k0 + kq;
      double mu, isd;
      if (st_lds) {
        const int kc = k < K ? k : K - 1;
        mu = stl[2 * kc]; isd = stl[2 * kc + 1];
      } else {
        mu = has_mu ? (double)zm[j][e] : 0.0;
        const double sd = has_sd ? (double)zs[j][e] : 1.0;
        isd = __builtin_amdgcn_rcp(sd);
        isd = fma(fma(-sd, isd, 1.0), isd, isd);
      }
      const T z = (T)(((double)zv[j][e] - mu) * isd);
      Zb[kq * SSE_ZP + zr] = (zok && k < K) ? z : (T)0;
    }
  };
  auto storeB = [&](int i, int k0, int buf) {
    T *Bb = Bs + buf * SSE_KS * BP;
    const int kk = (tid + 256 * i) / WP;
    const bool ok = bok[i] && k0 + kk < K;
    vec_t v = bv[i];
#pragma unroll
    for (int e = 0; e < V; ++e) v[e] = ok ? v[e] : (T)0;
    *reinterpret_cast<vec_t *>(Bb + blds[i]) = v;
  };
  // The next stage's requests and LDS writes are dealt out over the stage's 4 NT groups of four MFMAs
  // ("steps"), one or two per step and pinned there: a vector-memory instruction takes its wave
  // 150-450 cycles to issue on a busy CU and the matrix pipe runs dry behind a clump of them
  // (tools/sse_stamps.py: 12 loads 2.0 k, arithmetic and LDS writes 1.9 k cycles per stage next to
  // 6.8 k of MFMAs, one after the other).  Requests in the first half of the steps (rows first),
  // writes in the second half (coefficients first, rows -- the longest latency -- last).
  constexpr int STEPS = 4 * NT, NL = ZL + NB, NLS = (STEPS + 1) / 2, LPS = (NL + NLS - 1) / NLS;
  constexpr int NSS = STEPS - NLS, SPS = (NL + NSS - 1) / NSS;
  auto aux = [&](int step, int k1, int buf) {
    if (step < NLS) {
#pragma unroll
      for (int q = 0; q < LPS; ++q) {
        const int idx = step * LPS + q;
        if (idx < ZL) loadZ(idx, k1);
        else if (idx < NL) loadB(idx - ZL, k1);
      }
    } else {
#pragma unroll
      for (int q = 0; q < SPS; ++q) {
        const int idx = (step - NLS) * SPS + q;
        if (idx < NB) storeB(idx, k1, buf);
        else if (idx < NL) storeZ(idx - NB, k1, buf);
      }
    }
  };
  const int nst = (K + SSE_KS - 1) / SSE_KS;
#pragma unroll
  for (int j = 0; j < ZL; ++j) loadZ(j, 0);
#pragma unroll
  for (int i = 0; i < NB; ++i) loadB(i, 0);
  if (st_lds) __syncthreads();                                // (stl)
#pragma unroll
  for (int i = 0; i < NB; ++i) storeB(i, 0, 0);
#pragma unroll
  for (int j = 0; j < ZL; ++j) storeZ(j, 0, 0);
  __syncthreads();
  SSE_STAMP(0);
  // (narrow groups, NT < 3: several workgroups share a CU and hide each other's latencies; the
  //  requests go first, the writes last, unpinned -- C = 20 at K = 4096 in float32: 0.95 ms against
  //  1.6 ms with the pinned interleave)
  constexpr bool PIN = NT >= 3;
  for (int s = 0; s < nst; ++s) {
    const bool more = s + 1 < nst;
    const int k1 = (s + 1) * SSE_KS, nbuf = (s + 1) & 1;      // (that buffer was last read before the previous barrier)
    const T *Zb = Zs + (s & 1) * SSE_KS * SSE_ZP, *Bb = Bs + (s & 1) * SSE_KS * BP;
    if (!PIN && more) {
#pragma unroll
      for (int j = 0; j < ZL; ++j) loadZ(j, k1);
#pragma unroll
      for (int i = 0; i < NB; ++i) loadB(i, k1);
    }
#pragma unroll
    for (int ks = 0; ks < SSE_KS; ks += 4) {
      T af[4];
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) af[rt] = Zb[(ks + lk) * SSE_ZP + 16 * rt + lc];
      // (tiles past the group's last column multiply zeros: the four waves run side by side on
      //  their SIMDs, skipping a tile in one of them would not shorten the stage)
      T bf[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) bf[t] = Bb[(ks + lk) * BP + 16 * (wave * NT + t) + lc];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) acc[rt][t] = MF<T>::mfma(af[rt], bf[t], acc[rt][t]);
        if (PIN) {
          if (more) aux((ks / 4) * NT + t, k1, nbuf);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (!PIN && more) {
#pragma unroll
      for (int i = 0; i < NB; ++i) storeB(i, k1, nbuf);
#pragma unroll
      for (int j = 0; j < ZL; ++j) storeZ(j, k1, nbuf);
    }
    SSE_STAMP(2);
    __syncthreads();
    SSE_STAMP(4);
  }
  // squared errors: register r of tile (rt, t) is (row 16 rt + drow(lane, r), column 16 (wave NT + t) + lc);
  // summed over registers, row tiles, then the four lane groups -- a fixed order
  const T *muY = a.muY ? (const T *)a.muY + (size_t)f * M : nullptr;
  const T *sdY = a.sdY ? (const T *)a.sdY + (size_t)f * M : nullptr;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int lcol = 16 * (wave * NT + t) + lc;
    const bool valid = lcol < Cg;
    const int col = c0 + lcol;
    const int m = valid ? col % M : 0;
    const double sy = sdY ? (double)sdY[m] : 1.0, my = muY ? (double)muY[m] : 0.0;
    double sacc = 0.0;
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int lr = 16 * rt + MF<T>::drow(lane, r);
        if (valid && r0 + lr < n) {
          const double e = (double)acc[rt][t][r] * sy + my - (double)Y[rows[lr] * (int64_t)M + m];
          sacc += wl[lr] * (e * e);
        }
      }
    const double s1 = __shfl(sacc, lc + 16), s2 = __shfl(sacc, lc + 32), s3 = __shfl(sacc, lc + 48);
    const double sw = ((__shfl(sacc, lc) + s1) + s2) + s3;
    if (lk == 0 && valid) part[col] = sw;
  }
  if (cg == 0 && tid == 0) {
    double t = 0.0;
    for (int i = 0; i < SSE_ROWS; ++i) t += wl[i];
    *wpart = t;
  }
  SSE_STAMP(5);
}

// sums over a fold's row chunks, in chunk order
__global__ void pls_sse_reduce_kernel(const SseArgs a) {
  const int C = a.A * a.M;
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < a.F * C) {
    const int64_t f = e / C;
    const int c = (int)(e - f * C);
    const int64_t n = a.offs[f + 1] - a.offs[f];
    const int used = (int)((n + SSE_ROWS - 1) / SSE_ROWS);
    double s = 0.0;
    for (int ch = 0; ch < used; ++ch) s += a.part[((size_t)f * a.n_chunks + ch) * C + c];
    a.sse[e] = s;
  }
  if (e < a.F) {
    const int64_t n = a.offs[e + 1] - a.offs[e];
    const int used = (int)((n + SSE_ROWS - 1) / SSE_ROWS);
    double s = 0.0;
    for (int ch = 0; ch < used; ++ch) s += a.part[(size_t)a.F * a.n_chunks * C + (size_t)e * a.n_chunks + ch];
    a.wsum[e] = s;
  }
}

size_t pls_sse_workspace_bytes(int64_t F, int64_t max_rows, int M, int A) {
  const int64_t chunks = (max_rows + SSE_ROWS - 1) / SSE_ROWS;
  return (size_t)F * (chunks > 0 ? chunks : 1) * ((size_t)A * M + 1) * 8 + 256;
}

template <typename T>
int pls_sse_impl(const void *X, const void *Y, const void *w, const int64_t *idx, const int64_t *offsets,
                 int64_t F, int64_t max_rows, int K, int M, int A, const void *muX, const void *sdX, const void *muY,
                 const void *sdY, const void *B, double *sse, double *wsum, void *ws, size_t ws_bytes, hipStream_t st) {
  if (ws_bytes < pls_sse_workspace_bytes(F, max_rows, M, A)) return fail(CVM_EWORKSPACE, "cvm_pls_validation_sse: workspace too small%s");
  if (F == 0) return CVM_OK;
  SseArgs a;
  memset(&a, 0, sizeof(a));
  a.X = X; a.Y = Y; a.w = w; a.muX = muX; a.sdX = sdX; a.muY = muY; a.sdY = sdY; a.B = B;
  a.idx = idx; a.offs = offsets; a.K = K; a.M = M; a.A = A; a.F = F;
  int64_t chunks = (max_rows + SSE_ROWS - 1) / SSE_ROWS;
  if (chunks < 1) chunks = 1;
  a.n_chunks = (int)chunks;
  a.part = reinterpret_cast<double *>(ws);
  a.sse = sse; a.wsum = wsum;
  const int C = A * M;
  if (F > 65535) return fail(CVM_EINVAL, "cvm_pls_validation_sse: at most 65535 folds per call%s");
  // 16 NT columns per wave: as few column groups as possible (every group stages the rows again),
  // then as little padding as possible
  constexpr int VW = 16 / (int)sizeof(T);
  const bool vec = K % VW == 0 && M % VW == 0 && K >= 4 && ((uintptr_t)X % 16 == 0) && ((uintptr_t)B % 16 == 0) &&
                   (!muX || (uintptr_t)muX % 16 == 0) && (!sdX || (uintptr_t)sdX % 16 == 0);
  // (float64 takes five column tiles per wave at most, four without 16-byte pieces: the wider variants of
  //  those combinations spill -- 33 / 90 registers at six tiles, 3 at five scalar ones)
  const int max_nt = sizeof(T) == 8 ? (vec ? 5 : 4) : SSE_MAXNT;
  int nt = 1, best = 1 << 30;
  for (int c = 1; c <= max_nt; ++c) {
    const int groups = (C + 64 * c - 1) / (64 * c);
    const int cost = groups * (4 * c + 1);
    if (cost < best) { best = cost; nt = c; }
  }
  const unsigned groups = (unsigned)((C + 64 * nt - 1) / (64 * nt));
  size_t lds = (size_t)2 * SSE_KS * (SSE_ZP + 64 * nt + 16) * sizeof(T);
  // (statistics in LDS only where they do not cost residency: one workgroup per CU anyway, or a short K)
  a.st_in_lds = (lds + (size_t)K * 16 + 2048 <= PLS_LDS_BUDGET && (lds > 80 * 1024 || K <= 512)) ? 1 : 0;
  if (a.st_in_lds) lds += (size_t)K * 16;
  void (*kern)(const SseArgs) = nullptr;
#define CVM_SSE_PICK(N) kern = vec ? pls_sse_kernel<T, N, VW> : pls_sse_kernel<T, N, 1>
  switch (nt) {
    case 1: CVM_SSE_PICK(1); break;
    case 2: CVM_SSE_PICK(2); break;
    case 3: CVM_SSE_PICK(3); break;
    case 4: CVM_SSE_PICK(4); break;
    case 5:
      if constexpr (sizeof(T) == 8) kern = pls_sse_kernel<T, 5, VW>;      // (vec only: see max_nt)
      else CVM_SSE_PICK(5);
      break;
    default:
      if constexpr (sizeof(T) == 4) CVM_SSE_PICK(6);
      break;
  }
  if (!kern) return fail(CVM_EINVAL, "cvm_pls_validation_sse: no kernel for this shape%s");
#undef CVM_SSE_PICK
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kern, dim3((unsigned)chunks, groups, (unsigned)F), dim3(256), lds, st, a);
  const int64_t ne = F * C > F ? F * C : F;
  hipLaunchKernelGGL(pls_sse_reduce_kernel, dim3((unsigned)((ne + 255) / 256)), dim3(256), 0, st, a);
  HIP_OK(hipGetLastError());
  return CVM_OK;
}
