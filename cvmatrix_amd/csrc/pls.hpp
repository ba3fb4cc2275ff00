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
// With S == 1 (many folds) the barrier is a __syncthreads().  All arithmetic in float64; fixed
// summation orders, no float atomics: results do not depend on scheduling.
#pragma once

constexpr int PLS_THREADS = 256;
constexpr int PLS_MAXM = 32;        // responses (the M x M eigenproblem lives in LDS)
constexpr int PLS_MAXA = 512;       // components
constexpr size_t PLS_LDS_BUDGET = 150 * 1024;

struct PlsArgs {
  const void *XTX, *XTY;            // [F][K][K], [F][K][M]
  int K, M, A, S, rows;             // slices per fold, rows per slice
  int y_in_lds;
  double *Yw, *Bw;                  // [F][K][M] deflated XTY, running B
  double *Pw, *Rw;                  // [F][K][A]
  double *xch;                      // [F][xch_len]
  unsigned *cnt;                    // [F] barrier counters (zeroed before the launch)
  int *status;                      // [1]  set to 1 if a barrier timed out
  void *B, *W, *P, *Q, *R;          // outputs ([F][A][K][M]; [F][K][A] x3 and [F][M][A], optional)
  int *n_fit;                       // [F]
  double eps;
};

__host__ __device__ inline size_t pls_xch_len(int K, int M, int A, int S) {
  return (size_t)S * M * M + (size_t)S * (1 + A) + (size_t)K + (size_t)S * (1 + M);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over the workgroup, same value (bitwise) in every thread
__device__ __forceinline__ double block_sum(double v, double *red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

// barrier over the S workgroups of one fold; `target` counts arrivals so far
__device__ __forceinline__ bool fold_barrier(unsigned *cnt, unsigned &target, int S, int *status, int *lflag) {
  if (S == 1) { __syncthreads(); return true; }
  target += (unsigned)S;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    long spins = 0;
    int ok = 1;
    while (__hip_atomic_load(cnt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1L << 24)) { ok = 0; break; }      // seconds: the slices were not co-resident
    }
    if (!ok) *status = 1;
    *lflag = ok;
  }
  __syncthreads();
  __threadfence();
  return *lflag != 0;
}

template <typename T, bool XRES>
__global__ __launch_bounds__(PLS_THREADS) void pls_kernel(const PlsArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pls_smem[];
  const int K = a.K, M = a.M, A = a.A, S = a.S;
  const int f = blockIdx.x / S, s = blockIdx.x - f * S;
  const int k0 = s * a.rows;
  const int n = (k0 + a.rows <= K) ? a.rows : (K - k0 > 0 ? K - k0 : 0);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int MM = M * M;

  double *rl = reinterpret_cast<double *>(pls_smem);           // r, all K rows
  double *wl = rl + ((K + 1) & ~1);                            // w, r, u of the slice
  double *rs = wl + ((a.rows + 1) & ~1);
  double *us = rs + ((a.rows + 1) & ~1);
  double *S0 = us + ((a.rows + 1) & ~1);                       // M x M: XTY^T XTY and two squarings
  double *Ba = S0 + MM;
  double *Bb = Ba + MM;
  double *qv = Bb + MM;                                        // M
  double *cj = qv + PLS_MAXM;                                  // A
  double *red = cj + ((A + 1) & ~1);                           // 4
  int *lflag = reinterpret_cast<int *>(red + 4);
  double *ys = red + 6;                                        // rows x M (optional)
  T *xs = reinterpret_cast<T *>(ys + (a.y_in_lds ? (size_t)a.rows * M : 0));   // rows x K (XRES)

  const T *XTX = (const T *)a.XTX + (size_t)f * K * K;
  const T *XTY = (const T *)a.XTY + (size_t)f * K * M;
  double *Yg = a.Yw + (size_t)f * K * M + (size_t)k0 * M;
  double *Y = a.y_in_lds ? ys : Yg;                            // the slice's deflated XTY
  double *Bw = a.Bw + (size_t)f * K * M + (size_t)k0 * M;
  double *Pw = a.Pw + (size_t)f * K * A + (size_t)k0 * A;
  double *Rw = a.Rw + (size_t)f * K * A + (size_t)k0 * A;
  double *xch = a.xch + (size_t)f * pls_xch_len(K, M, A, S);
  double *x_S = xch;                                           // [S][MM]
  double *x_2 = x_S + (size_t)S * MM;                          // [S][1 + A]
  double *x_r = x_2 + (size_t)S * (1 + A);                     // [K]
  double *x_4 = x_r + K;                                       // [S][1 + M]
  unsigned *cnt = a.cnt + f;
  unsigned target = 0;

  // ---- prologue: working copies of the slice -------------------------------------------
  for (int e = tid; e < n * M; e += PLS_THREADS) {
    Y[e] = (double)XTY[(size_t)k0 * M + e];
    Bw[e] = 0.0;
  }
  if (XRES) {
    const T *src = XTX + (size_t)k0 * K;
    for (size_t e = tid; e < (size_t)n * K; e += PLS_THREADS) xs[e] = src[e];
  }
  __syncthreads();

  int fit = 0;
  for (int c = 0; c < A; ++c) {
    // ---- 1: partial XTY^T XTY of the slice's rows (upper triangle, mirrored) ---------------
    if (M > 1) {
      for (int p = tid; p < MM; p += PLS_THREADS) {
        const int i = p / M, j = p - i * M;
        if (i > j) continue;
        double acc = 0.0;
        for (int k = 0; k < n; ++k) acc += Y[(size_t)k * M + i] * Y[(size_t)k * M + j];
        x_S[(size_t)s * MM + p] = acc;
        x_S[(size_t)s * MM + (size_t)j * M + i] = acc;
      }
      if (!fold_barrier(cnt, target, S, a.status, lflag)) return;
      // ---- 2a: dominant eigenvector q of the M x M sum, by repeated squaring -----------------
      for (int p = tid; p < MM; p += PLS_THREADS) {
        double acc = 0.0;
        for (int t = 0; t < S; ++t) acc += x_S[(size_t)t * MM + p];
        S0[p] = acc;
      }
      __syncthreads();
      double tr = 0.0;
      for (int i = 0; i < M; ++i) tr += S0[(size_t)i * M + i];
      if (tr > 0.0) {
        for (int p = tid; p < MM; p += PLS_THREADS) Ba[p] = S0[p] / tr;
        __syncthreads();
        double *src = Ba, *dst = Bb;
        for (int it = 0; it < 64; ++it) {
          for (int p = tid; p < MM; p += PLS_THREADS) {
            const int i = p / M, j = p - i * M;
            double acc = 0.0;
            for (int k = 0; k < M; ++k) acc += src[(size_t)i * M + k] * src[(size_t)k * M + j];
            dst[p] = acc;
          }
          __syncthreads();
          double t2 = 0.0;                                   // trace of the square: sum lambda^2, trace 1 before
          for (int i = 0; i < M; ++i) t2 += dst[(size_t)i * M + i];
          __syncthreads();
          for (int p = tid; p < MM; p += PLS_THREADS) dst[p] = dst[p] / t2;
          __syncthreads();
          double *tmp = src; src = dst; dst = tmp;
          if (1.0 - t2 < 1e-15) break;                         // numerically rank one
        }
        // the column with the largest diagonal entry, then two power steps with the sum itself
        int best = 0;
        for (int i = 1; i < M; ++i) if (src[(size_t)i * M + i] > src[(size_t)best * M + best]) best = i;
        if (tid < M) qv[tid] = src[(size_t)tid * M + best];
        __syncthreads();
        for (int polish = 0; polish < 2; ++polish) {
          double v = 0.0;
          if (tid < M) for (int k = 0; k < M; ++k) v += S0[(size_t)tid * M + k] * qv[k];
          __syncthreads();
          if (tid < M) dst[tid] = v;
          __syncthreads();
          double nn = 0.0;
          for (int k = 0; k < M; ++k) nn += dst[k] * dst[k];
          nn = sqrt(nn);
          if (tid < M) qv[tid] = nn > 0.0 ? dst[tid] / nn : 0.0;
          __syncthreads();
        }
      } else {
        if (tid < M) qv[tid] = 0.0;
        __syncthreads();
      }
    }
    // ---- 2b: w of the slice (not normalised yet), its partial norm and partial P^T w -------
    double nrm2 = 0.0;
    for (int k = tid; k < n; k += PLS_THREADS) {
      double v;
      if (M == 1) v = Y[k];
      else {
        v = 0.0;
        for (int j = 0; j < M; ++j) v += Y[(size_t)k * M + j] * qv[j];
      }
      wl[k] = v;
      nrm2 += v * v;
    }
    nrm2 = block_sum(nrm2, red);
    if (tid == 0) x_2[(size_t)s * (1 + A)] = nrm2;
    for (int j = wave; j < c; j += PLS_THREADS / 64) {
      double acc = 0.0;
      for (int k = lane; k < n; k += 64) acc += Pw[(size_t)k * A + j] * wl[k];
      acc = wave_sum(acc);
      if (lane == 0) x_2[(size_t)s * (1 + A) + 1 + j] = acc;
    }
    if (!fold_barrier(cnt, target, S, a.status, lflag)) return;
    // ---- 3: normalise, r = w - R (P^T w) -----------------------------------------------------
    double nrm = 0.0;
    for (int t = 0; t < S; ++t) nrm += x_2[(size_t)t * (1 + A)];
    nrm = sqrt(nrm);
    if (!(nrm > a.eps)) break;                                 // nothing left to extract (uniform)
    for (int j = tid; j < c; j += PLS_THREADS) {
      double acc = 0.0;
      for (int t = 0; t < S; ++t) acc += x_2[(size_t)t * (1 + A) + 1 + j];
      cj[j] = acc / nrm;
    }
    __syncthreads();
    for (int k = tid; k < n; k += PLS_THREADS) {
      const double w = wl[k] / nrm;
      double corr = 0.0;
      for (int j = 0; j < c; ++j) corr += Rw[(size_t)k * A + j] * cj[j];
      const double r = w - corr;
      wl[k] = w;
      rs[k] = r;
      Rw[(size_t)k * A + c] = r;
      x_r[k0 + k] = r;
      if (a.W) ((T *)a.W)[((size_t)f * K + k0 + k) * A + c] = (T)w;
      if (a.R) ((T *)a.R)[((size_t)f * K + k0 + k) * A + c] = (T)r;
    }
    if (!fold_barrier(cnt, target, S, a.status, lflag)) return;
    // ---- 4: u = XTX[slice, :] r, partial r^T u and partial XTY^T r ---------------------------
    for (int k = tid; k < K; k += PLS_THREADS) rl[k] = x_r[k];
    __syncthreads();
    for (int i0 = wave * 4; i0 < n; i0 += 4 * (PLS_THREADS / 64)) {
      double acc[4] = {0.0, 0.0, 0.0, 0.0};
      const int ni = n - i0 < 4 ? n - i0 : 4;
      if (ni == 4) {
        for (int k = lane; k < K; k += 64) {
          const double rv = rl[k];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const T xv = XRES ? xs[(size_t)(i0 + q) * K + k] : XTX[(size_t)(k0 + i0 + q) * K + k];
            acc[q] += (double)xv * rv;
          }
        }
      } else {
        for (int k = lane; k < K; k += 64) {
          const double rv = rl[k];
          for (int q = 0; q < ni; ++q) {
            const T xv = XRES ? xs[(size_t)(i0 + q) * K + k] : XTX[(size_t)(k0 + i0 + q) * K + k];
            acc[q] += (double)xv * rv;
          }
        }
      }
      for (int q = 0; q < ni; ++q) {
        const double v = wave_sum(acc[q]);
        if (lane == 0) us[i0 + q] = v;
      }
    }
    __syncthreads();
    double tt = 0.0;
    for (int k = tid; k < n; k += PLS_THREADS) tt += rs[k] * us[k];
    tt = block_sum(tt, red);
    if (tid == 0) x_4[(size_t)s * (1 + M)] = tt;
    for (int j = wave; j < M; j += PLS_THREADS / 64) {
      double acc = 0.0;
      for (int k = lane; k < n; k += 64) acc += Y[(size_t)k * M + j] * rs[k];
      acc = wave_sum(acc);
      if (lane == 0) x_4[(size_t)s * (1 + M) + 1 + j] = acc;
    }
    if (!fold_barrier(cnt, target, S, a.status, lflag)) return;
    // ---- 5: p, q, deflation of the slice, B ---------------------------------------------------
    double tTt = 0.0;
    for (int t = 0; t < S; ++t) tTt += x_4[(size_t)t * (1 + M)];
    if (tid < M) {
      double acc = 0.0;
      for (int t = 0; t < S; ++t) acc += x_4[(size_t)t * (1 + M) + 1 + tid];
      const double q = acc / tTt;
      qv[tid] = q;
      if (a.Q && s == 0) ((T *)a.Q)[((size_t)f * M + tid) * A + c] = (T)q;
    }
    __syncthreads();
    for (int k = tid; k < n; k += PLS_THREADS) {
      const double p = us[k] / tTt;
      us[k] = p;
      Pw[(size_t)k * A + c] = p;
      if (a.P) ((T *)a.P)[((size_t)f * K + k0 + k) * A + c] = (T)p;
    }
    __syncthreads();
    T *Bout = (T *)a.B + (((size_t)f * A + c) * K + k0) * M;
    for (int e = tid; e < n * M; e += PLS_THREADS) {
      const int k = e / M, j = e - k * M;
      Y[e] = Y[e] - (us[k] * qv[j]) * tTt;
      const double b = Bw[e] + rs[k] * qv[j];
      Bw[e] = b;
      Bout[e] = (T)b;
    }
    __syncthreads();
    fit = c + 1;
  }
  if (s == 0 && tid == 0) a.n_fit[f] = fit;
}

// ---- host ---------------------------------------------------------------------------------
struct PlsPlan {
  int S, rows, folds_per_launch, y_in_lds, xres;
  size_t lds;
};

size_t pls_lds_bytes(int K, int M, int A, int rows, int y_in_lds, int xres, int esize) {
  size_t d = ((K + 1) & ~1) + 3 * (size_t)((rows + 1) & ~1) + 3 * (size_t)M * M + PLS_MAXM + ((A + 1) & ~1) + 6;
  if (y_in_lds) d += (size_t)rows * M;
  size_t b = d * 8;
  if (xres) b += (size_t)rows * K * esize;
  return b;
}

bool make_pls_plan(int64_t F, int K, int M, int A, int esize, int cus, PlsPlan &p) {
  // slices: as many as keep every workgroup of a launch resident (folds x S <= CUs); at least as
  // many as the LDS needs for the per-slice vectors
  int S = (F >= cus) ? 1 : (int)(cus / (F > 0 ? F : 1));
  if (S > (K + 7) / 8) S = (K + 7) / 8;                        // >= 8 rows per slice
  if (S < 1) S = 1;
  for (;; ++S) {
    const int rows = (K + S - 1) / S;
    if (pls_lds_bytes(K, M, A, rows, 0, 0, esize) <= PLS_LDS_BUDGET) break;
    if (S >= K || S >= cus) return false;
  }
  p.rows = (K + S - 1) / S;
  p.S = (K + p.rows - 1) / p.rows;
  p.folds_per_launch = p.S == 1 ? (int)(F < (1 << 20) ? (F > 0 ? F : 1) : (1 << 20)) : cus / p.S;
  if (p.folds_per_launch < 1) return false;
  p.y_in_lds = pls_lds_bytes(K, M, A, p.rows, 1, 0, esize) <= PLS_LDS_BUDGET && (size_t)p.rows * M * 8 <= 48 * 1024;
  p.xres = pls_lds_bytes(K, M, A, p.rows, p.y_in_lds, 1, esize) <= PLS_LDS_BUDGET;
  p.lds = pls_lds_bytes(K, M, A, p.rows, p.y_in_lds, p.xres, esize);
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

size_t pls_workspace_bytes(int64_t F, int K, int M, int A, int esize, int cus) {
  PlsPlan p;
  if (!make_pls_plan(F, K, M, A, esize, cus, p)) return 0;
  const size_t per_fold = (2 * (size_t)K * M + 2 * (size_t)K * A + pls_xch_len(K, M, A, p.S)) * 8 + 16;
  return (size_t)F * per_fold + 256;
}

template <typename T>
int pls_fit_impl(const void *XTX, const void *XTY, int64_t F, int K, int M, int A, void *B, void *W, void *P,
                 void *Q, void *R, int32_t *n_fit, int32_t *status, void *ws, size_t ws_bytes, hipStream_t st) {
  const int cus = pls_cu_count();
  PlsPlan p;
  if (!make_pls_plan(F, K, M, A, sizeof(T), cus, p)) return fail(CVM_EINVAL, "cvm_pls_fit: K too large for the LDS plan%s");
  if (ws_bytes < pls_workspace_bytes(F, K, M, A, sizeof(T), cus)) return fail(CVM_EWORKSPACE, "cvm_pls_fit: workspace too small%s");
  if (F == 0) return CVM_OK;
  double *d = reinterpret_cast<double *>(ws);
  PlsArgs a;
  a.K = K; a.M = M; a.A = A; a.S = p.S; a.rows = p.rows; a.y_in_lds = p.y_in_lds;
  a.Yw = d; d += (size_t)F * K * M;
  a.Bw = d; d += (size_t)F * K * M;
  a.Pw = d; d += (size_t)F * K * A;
  a.Rw = d; d += (size_t)F * K * A;
  a.xch = d; d += (size_t)F * pls_xch_len(K, M, A, p.S);
  a.cnt = reinterpret_cast<unsigned *>(d);
  a.status = status;
  a.eps = sizeof(T) == 8 ? 2.220446049250313e-16 : 1.1920928955078125e-07;
  HIP_OK(hipMemsetAsync(a.cnt, 0, (size_t)F * sizeof(unsigned), st));
  HIP_OK(hipMemsetAsync(status, 0, sizeof(int32_t), st));
  const size_t es = sizeof(T);
  // components that are not extracted (stopping rule) stay zero
  HIP_OK(hipMemsetAsync(B, 0, (size_t)F * A * K * M * es, st));
  if (W) HIP_OK(hipMemsetAsync(W, 0, (size_t)F * K * A * es, st));
  if (P) HIP_OK(hipMemsetAsync(P, 0, (size_t)F * K * A * es, st));
  if (R) HIP_OK(hipMemsetAsync(R, 0, (size_t)F * K * A * es, st));
  if (Q) HIP_OK(hipMemsetAsync(Q, 0, (size_t)F * M * A * es, st));
  auto kern = p.xres ? pls_kernel<T, true> : pls_kernel<T, false>;
  HIP_OK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds));
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
  return CVM_OK;
}
