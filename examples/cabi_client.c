/*
 * cabi_client.c -- the C ABI of libcvmhip.so used from plain C: no Python, no torch.
 *
 * Allocates X, Y, w on the device with the HIP runtime, runs the fit stage (cvm_gram_fit) and the
 * fold stage (cvm_fold_update) for three folds, and checks the first fold's training matrices
 * against a direct float64 computation from the training rows (the reference's NaiveCVMatrix
 * definition, tests/naive_cvmatrix.py:171-277, centring + scaling, ddof = 1).
 *
 * Build (tests/test_gpu_parity.py::test_c_abi_from_plain_c does this), plain C11:
 *   gcc -std=c11 -O2 examples/cabi_client.c -Iinclude -I/opt/rocm/include -Lcvmatrix_amd -lcvmhip \
 *       -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/cvmatrix_amd -Wl,-rpath,/opt/rocm/lib -lm -o cabi_client
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "cvmhip.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
  fprintf(stderr, "HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } } while (0)
#define CHECK_CVM(x) do { int r_ = (x); if (r_ != CVM_OK) { \
  fprintf(stderr, "cvm error %d (%s) at %s:%d\n", r_, cvm_last_error(), __FILE__, __LINE__); return 3; } } while (0)

static double urand(uint64_t *s) {   /* xorshift: reproducible inputs */
  *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17;
  return (double)(*s >> 11) / 9007199254740992.0;
}

int main(void) {
  const int64_t N = 6000;
  const int K = 192, M = 4, P = 3;
  const unsigned flags = CVM_RET_XTX | CVM_RET_XTY | CVM_CENTER_X | CVM_CENTER_Y | CVM_SCALE_X | CVM_SCALE_Y;
  uint64_t seed = 88172645463325252ull;
  double *X = malloc(sizeof(double) * N * K), *Y = malloc(sizeof(double) * N * M), *w = malloc(sizeof(double) * N);
  for (int64_t i = 0; i < N * K; ++i) X[i] = urand(&seed) - 0.3;
  for (int64_t i = 0; i < N * M; ++i) Y[i] = urand(&seed);
  for (int64_t i = 0; i < N; ++i) w[i] = (i % 17 == 0) ? 0.0 : urand(&seed);
  /* folds: row i belongs to fold i % P; CSR of the validation rows */
  int64_t *idx = malloc(sizeof(int64_t) * N), offs[4] = {0, 0, 0, 0};
  int64_t n = 0;
  for (int f = 0; f < P; ++f) {
    for (int64_t i = f; i < N; i += P) idx[n++] = i;
    offs[f + 1] = n;
  }

  printf("%s\n", cvm_version());
  double *dX, *dY, *dw, *dG, *dH, *dgs, *dXTX, *dXTY, *dmuX, *dsdX, *dmuY, *dsdY, *dfold;
  int64_t *didx, *doffs;
  int32_t *dneg;
  void *dws;
  const size_t ngs = cvm_gstats_len(K, M);
  size_t ws_bytes = cvm_fit_workspace_bytes(N, K, M, CVM_F64);
  const size_t ws2 = cvm_fold_workspace_bytes(P, N, N / P + 1, K, M, CVM_F64, flags);
  if (ws2 > ws_bytes) ws_bytes = ws2;
  CHECK_HIP(hipMalloc((void **)&dX, sizeof(double) * N * K));
  CHECK_HIP(hipMalloc((void **)&dY, sizeof(double) * N * M));
  CHECK_HIP(hipMalloc((void **)&dw, sizeof(double) * N));
  CHECK_HIP(hipMalloc((void **)&didx, sizeof(int64_t) * N));
  CHECK_HIP(hipMalloc((void **)&doffs, sizeof(int64_t) * (P + 1)));
  CHECK_HIP(hipMalloc((void **)&dG, sizeof(double) * K * K));
  CHECK_HIP(hipMalloc((void **)&dH, sizeof(double) * K * M));
  CHECK_HIP(hipMalloc((void **)&dgs, sizeof(double) * ngs));
  CHECK_HIP(hipMalloc((void **)&dneg, sizeof(int32_t)));
  CHECK_HIP(hipMalloc((void **)&dXTX, sizeof(double) * P * K * K));
  CHECK_HIP(hipMalloc((void **)&dXTY, sizeof(double) * P * K * M));
  CHECK_HIP(hipMalloc((void **)&dmuX, sizeof(double) * P * K));
  CHECK_HIP(hipMalloc((void **)&dsdX, sizeof(double) * P * K));
  CHECK_HIP(hipMalloc((void **)&dmuY, sizeof(double) * P * M));
  CHECK_HIP(hipMalloc((void **)&dsdY, sizeof(double) * P * M));
  CHECK_HIP(hipMalloc((void **)&dfold, sizeof(double) * P * 4));
  CHECK_HIP(hipMalloc(&dws, ws_bytes));
  CHECK_HIP(hipMemcpy(dX, X, sizeof(double) * N * K, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(dY, Y, sizeof(double) * N * M, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(dw, w, sizeof(double) * N, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(didx, idx, sizeof(int64_t) * N, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(doffs, offs, sizeof(int64_t) * (P + 1), hipMemcpyHostToDevice));

  hipStream_t st;
  CHECK_HIP(hipStreamCreate(&st));
  CHECK_CVM(cvm_gram_fit(dX, dY, dw, N, K, M, CVM_F64, dG, dH, dgs, dneg, dws, ws_bytes, st));
  CHECK_CVM(cvm_fold_update(dX, dY, dw, didx, doffs, offs, P, N, K, M, CVM_F64, flags, 1.0, 1e-14, dG, dH, dgs,
                            dXTX, dXTY, dmuX, dsdX, dmuY, dsdY, dfold, dws, ws_bytes, st));
  CHECK_HIP(hipStreamSynchronize(st));

  double *XTX = malloc(sizeof(double) * K * K), *XTY = malloc(sizeof(double) * K * M), fold[4];
  CHECK_HIP(hipMemcpy(XTX, dXTX, sizeof(double) * K * K, hipMemcpyDeviceToHost));   /* fold 0 */
  CHECK_HIP(hipMemcpy(XTY, dXTY, sizeof(double) * K * M, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(fold, dfold, sizeof(fold), hipMemcpyDeviceToHost));

  /* direct computation for fold 0 from its TRAINING rows (i % P != 0) */
  double sw = 0, nz = 0;
  double *mX = calloc(K, sizeof(double)), *mY = calloc(M, sizeof(double));
  double *vX = calloc(K, sizeof(double)), *vY = calloc(M, sizeof(double));
  for (int64_t i = 0; i < N; ++i) {
    if (i % P == 0) continue;
    sw += w[i]; nz += (w[i] != 0.0);
    for (int k = 0; k < K; ++k) mX[k] += w[i] * X[i * K + k];
    for (int m = 0; m < M; ++m) mY[m] += w[i] * Y[i * M + m];
  }
  for (int k = 0; k < K; ++k) mX[k] /= sw;
  for (int m = 0; m < M; ++m) mY[m] /= sw;
  for (int64_t i = 0; i < N; ++i) {
    if (i % P == 0) continue;
    for (int k = 0; k < K; ++k) { const double d = X[i * K + k] - mX[k]; vX[k] += w[i] * d * d; }
    for (int m = 0; m < M; ++m) { const double d = Y[i * M + m] - mY[m]; vY[m] += w[i] * d * d; }
  }
  const double div = (nz - 1.0) * sw / nz;
  for (int k = 0; k < K; ++k) vX[k] = sqrt(vX[k] / div);
  for (int m = 0; m < M; ++m) vY[m] = sqrt(vY[m] / div);
  double *RX = calloc((size_t)K * K, sizeof(double)), *RY = calloc((size_t)K * M, sizeof(double));
  double *xs = malloc(sizeof(double) * K), *ys = malloc(sizeof(double) * M);
  for (int64_t i = 0; i < N; ++i) {
    if (i % P == 0 || w[i] == 0.0) continue;
    for (int k = 0; k < K; ++k) xs[k] = (X[i * K + k] - mX[k]) / vX[k];
    for (int m = 0; m < M; ++m) ys[m] = (Y[i * M + m] - mY[m]) / vY[m];
    for (int a = 0; a < K; ++a) {
      const double wa = w[i] * xs[a];
      for (int b = 0; b < K; ++b) RX[(size_t)a * K + b] += wa * xs[b];
      for (int m = 0; m < M; ++m) RY[(size_t)a * M + m] += wa * ys[m];
    }
  }
  double ex = 0, mx = 0, ey = 0, my = 0;
  for (size_t i = 0; i < (size_t)K * K; ++i) { ex = fmax(ex, fabs(XTX[i] - RX[i])); mx = fmax(mx, fabs(RX[i])); }
  for (size_t i = 0; i < (size_t)K * M; ++i) { ey = fmax(ey, fabs(XTY[i] - RY[i])); my = fmax(my, fabs(RY[i])); }
  printf("fold 0: sw_train %.6f (direct %.6f)  nz_train %.0f (direct %.0f)\n", fold[0], sw, fold[1], nz);
  printf("max|XTX - direct| / max|direct| = %.3e\nmax|XTY - direct| / max|direct| = %.3e\n", ex / mx, ey / my);
  const int ok = ex <= 1e-10 * mx && ey <= 1e-10 * my && fabs(fold[0] - sw) <= 1e-9 * sw && fold[1] == nz;
  printf(ok ? "OK\n" : "MISMATCH\n");
  return ok ? 0 : 1;
}
