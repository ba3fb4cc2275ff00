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
#include <algorithm>
#include <atomic>
#include <functional>
#include <map>
#include <mutex>
#include <queue>
#include <tuple>
#include <type_traits>
#include <vector>

#include "../../include/cvmhip.h"

namespace {

#include "geometry.hpp"
#include "wgram_fallback.hpp"
#include "wgram4.hpp"
#include "finalize.hpp"
#include "colstats.hpp"
#include "small_folds.hpp"
#include "resident.hpp"
#include "mid_tile.hpp"
#include "host.hpp"
#include "partition.hpp"
#include "pls.hpp"

}  // namespace

// ----------------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------------
extern "C" {

// CVM_SRC_SHA: sha256 (first 16 hex digits) over the sources the library was built from
// (cvmatrix_amd/build.py computes it; the Python loader compares it with the sources it finds
// next to the library, so a stale prebuilt .so cannot be used by accident)
#ifndef CVM_SRC_SHA
#define CVM_SRC_SHA "unknown"
#endif
const char *cvm_version(void) { return "cvmhip 0.3.0 (gfx950) src " CVM_SRC_SHA; }
const char *cvm_source_hash(void) { return CVM_SRC_SHA; }
const char *cvm_last_error(void) { return g_err; }

size_t cvm_gstats_len(int K, int M) { return 2 * (size_t)K + 2 * (size_t)M + 2; }

size_t cvm_fit_workspace_bytes(int64_t N, int K, int M, int dtype) {
  const Geom g = make_geom(K, M, dtype == CVM_F64 ? 8 : 4, 0);
  return (size_t)plan_stride(1, N, g, dtype == CVM_F64 ? 8 : 4) * g.unit_bytes + QUEUE_RESERVE;
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
    const size_t small = (size_t)nb * small_ws_per_fold(K, M, dtype == CVM_F64 ? 8 : 4, max_fold_rows) + 512;
    // (batches of folds of 8 rows or more may take mid_tile_kernel behind the statistics pre-pass: one
    //  statistics unit and one statistics vector per fold)
    const Geom gs = make_geom(K, M, dtype == CVM_F64 ? 8 : 4, 1);
    const size_t pre = (size_t)nb * (align_up(gs.stat_len * 8, 256) + align_up(fstat_len(K, M) * 8, 256));
    return (small > pre ? small : pre) + QUEUE_RESERVE;
  }
  const Geom g = make_geom(K, M, dtype == CVM_F64 ? 8 : 4, !(flags & CVM_RET_XTX));
  const int splits = plan_stride(n_folds, max_fold_rows, g, dtype == CVM_F64 ? 8 : 4);
  const size_t per_fold = (size_t)splits * g.unit_bytes + align_up(fstat_len(K, M) * 8, 256);
  size_t want = per_fold * (size_t)(n_folds > 0 ? n_folds : 1);
  const size_t cap = (size_t)8 << 30;   // beyond 8 GiB walk the folds in batches
  if (want > cap) want = (cap / per_fold > 0 ? cap / per_fold : 1) * per_fold;
  return want + QUEUE_RESERVE;
}

int cvm_fold_update_ex(const void *X, const void *Y, const void *w, const int64_t *idx,
                       const int64_t *offsets, const int64_t *host_offsets, int64_t n_folds, int64_t N,
                       int K, int M, int dtype, unsigned flags, double ddof, double resolution,
                       const void *G, const void *H, const double *gstats, void *out_XTX,
                       void *out_XTY, void *out_muX, void *out_sdX, void *out_muY, void *out_sdY,
                       double *out_fold, void *ws, size_t ws_bytes, void *stream, int32_t *status) {
  if (!X || !offsets || !host_offsets || !G || !gstats || !ws)
    return fail(CVM_EINVAL, "cvm_fold_update: null pointer%s");
  if (n_folds < 0 || N < 0 || K <= 0 || M < 0 || (M > 0 && !Y))
    return fail(CVM_EINVAL, "cvm_fold_update: bad shape%s");
  if (!idx && host_offsets[n_folds] > 0) return fail(CVM_EINVAL, "cvm_fold_update: idx is null%s");
  if ((flags & CVM_RET_XTY) && (M == 0 || !H))
    return fail(CVM_EINVAL, "cvm_fold_update: CVM_RET_XTY needs Y and H%s");
  if (n_folds == 0) return CVM_OK;
  if ((flags & CVM_IDX_HOST) && (n_folds != 1 || host_offsets[1] - host_offsets[0] > SMALL_ROWS ||
                                host_offsets[1] < host_offsets[0]))
    return fail(CVM_EINVAL, "cvm_fold_update: CVM_IDX_HOST takes one fold of at most 32 rows%s");
  if (dtype == CVM_F64)
    return fold_update_impl<double>(X, Y, w, idx, offsets, host_offsets, n_folds, N, K, M, dtype,
                                    flags, ddof, resolution, G, H, gstats, out_XTX, out_XTY, out_muX,
                                    out_sdX, out_muY, out_sdY, out_fold, ws, ws_bytes,
                                    (hipStream_t)stream, status);
  if (dtype == CVM_F32)
    return fold_update_impl<float>(X, Y, w, idx, offsets, host_offsets, n_folds, N, K, M, dtype,
                                   flags, ddof, resolution, G, H, gstats, out_XTX, out_XTY, out_muX,
                                   out_sdX, out_muY, out_sdY, out_fold, ws, ws_bytes,
                                   (hipStream_t)stream, status);
  return fail(CVM_EINVAL, "cvm_fold_update: dtype must be CVM_F32 or CVM_F64%s");
}

int cvm_fold_update(const void *X, const void *Y, const void *w, const int64_t *idx,
                    const int64_t *offsets, const int64_t *host_offsets, int64_t n_folds, int64_t N,
                    int K, int M, int dtype, unsigned flags, double ddof, double resolution,
                    const void *G, const void *H, const double *gstats, void *out_XTX,
                    void *out_XTY, void *out_muX, void *out_sdX, void *out_muY, void *out_sdY,
                    double *out_fold, void *ws, size_t ws_bytes, void *stream) {
  return cvm_fold_update_ex(X, Y, w, idx, offsets, host_offsets, n_folds, N, K, M, dtype, flags, ddof, resolution,
                            G, H, gstats, out_XTX, out_XTY, out_muX, out_sdX, out_muY, out_sdY, out_fold, ws,
                            ws_bytes, stream, nullptr);
}

#ifdef CVM_STAMPS
int cvm_debug_stamps(unsigned long long *host_out) {
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 1024 * 8 * 4));
  return CVM_OK;
}
int cvm_debug_pls_stamps(unsigned long long *host_out, int reset) {
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_pls_stamps), sizeof(unsigned long long) * 16));
  if (reset) {
    unsigned long long z[16] = {0};
    HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_pls_stamps), z, sizeof(z)));
  }
  return CVM_OK;
}
int cvm_debug_sse_stamps(unsigned long long *host_out, int reset) {
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_sse_stamps), sizeof(unsigned long long) * 8));
  if (reset) {
    unsigned long long z[8] = {0};
    HIP_OK(hipMemcpyToSymbol(HIP_SYMBOL(g_sse_stamps), z, sizeof(z)));
  }
  return CVM_OK;
}
int cvm_debug_stamps4(unsigned long long *host_out) {
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps4), sizeof(unsigned long long) * 1024 * 8));
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
  const int splits = plan_stride(n_folds, max_fold_rows, g, dtype == CVM_F64 ? 8 : 4);
  return ((size_t)splits * g.unit_bytes + align_up(fstat_len(K, M) * 8, 256)) * (size_t)(n_folds > 0 ? n_folds : 1) +
         QUEUE_RESERVE;
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

int cvm_sweep_all(const void *X, const void *Y, const void *w, const int64_t *idx, const int64_t *offsets,
                  const int64_t *host_offsets, int64_t n_folds, int64_t N, int K, int M, int dtype, unsigned flags,
                  double ddof, double resolution, void *G, void *H, double *gstats, int32_t *neg_flag,
                  void *out_XTX, void *out_XTY, void *out_muX, void *out_sdX, void *out_muY, void *out_sdY,
                  double *out_fold, void *ws, size_t ws_bytes, void *stream, int64_t *splits_out) {
  if (!X || !idx || !offsets || !host_offsets || !G || !gstats || !ws)
    return fail(CVM_EINVAL, "cvm_sweep_all: null pointer%s");
  if (n_folds <= 0 || N <= 0 || K <= 0 || M < 0 || (M > 0 && (!Y || !H)) || (M == 0 && Y))
    return fail(CVM_EINVAL, "cvm_sweep_all: bad shape%s");
  if ((flags & CVM_RET_XTY) && M == 0) return fail(CVM_EINVAL, "cvm_sweep_all: CVM_RET_XTY needs Y and H%s");
  if (dtype == CVM_F64)
    return sweep_all_impl<double>(X, Y, w, idx, offsets, host_offsets, n_folds, N, K, M, dtype, flags, ddof, resolution,
                                  G, H, gstats, neg_flag, out_XTX, out_XTY, out_muX, out_sdX, out_muY, out_sdY,
                                  out_fold, ws, ws_bytes, (hipStream_t)stream, splits_out);
  if (dtype == CVM_F32)
    return sweep_all_impl<float>(X, Y, w, idx, offsets, host_offsets, n_folds, N, K, M, dtype, flags, ddof, resolution,
                                 G, H, gstats, neg_flag, out_XTX, out_XTY, out_muX, out_sdX, out_muY, out_sdY,
                                 out_fold, ws, ws_bytes, (hipStream_t)stream, splits_out);
  return fail(CVM_EINVAL, "cvm_sweep_all: dtype must be CVM_F32 or CVM_F64%s");
}

int cvm_sweep_fold_range(const int64_t *offsets, int64_t n_total, int64_t fold0, int64_t n_folds, int K, int M,
                         int dtype, unsigned flags, double ddof, double resolution, int weighted, const void *G,
                         const void *H, const double *gstats, void *out_XTX, void *out_XTY, void *out_muX,
                         void *out_sdX, void *out_muY, void *out_sdY, double *out_fold, void *ws, size_t ws_bytes,
                         int64_t splits, void *stream) {
  if (!offsets || !G || !gstats || !ws) return fail(CVM_EINVAL, "cvm_sweep_folds: null pointer%s");
  if (n_total <= 0 || fold0 < 0 || n_folds <= 0 || fold0 + n_folds > n_total || K <= 0 || M < 0 || splits <= 0)
    return fail(CVM_EINVAL, "cvm_sweep_folds: bad shape%s");
  if ((flags & CVM_RET_XTY) && (M == 0 || !H))
    return fail(CVM_EINVAL, "cvm_sweep_folds: CVM_RET_XTY needs Y and H%s");
  if (dtype == CVM_F64)
    return sweep_folds_impl<double>(offsets, n_total, fold0, n_folds, K, M, dtype, flags, ddof, resolution, weighted,
                                    G, H, gstats, out_XTX, out_XTY, out_muX, out_sdX, out_muY, out_sdY, out_fold,
                                    ws, ws_bytes, splits, (hipStream_t)stream);
  if (dtype == CVM_F32)
    return sweep_folds_impl<float>(offsets, n_total, fold0, n_folds, K, M, dtype, flags, ddof, resolution, weighted,
                                   G, H, gstats, out_XTX, out_XTY, out_muX, out_sdX, out_muY, out_sdY, out_fold,
                                   ws, ws_bytes, splits, (hipStream_t)stream);
  return fail(CVM_EINVAL, "cvm_sweep_folds: dtype must be CVM_F32 or CVM_F64%s");
}

int cvm_sweep_folds(const int64_t *offsets, int64_t n_folds, int K, int M, int dtype, unsigned flags,
                    double ddof, double resolution, int weighted, const void *G, const void *H,
                    const double *gstats, void *out_XTX, void *out_XTY, void *out_muX, void *out_sdX,
                    void *out_muY, void *out_sdY, double *out_fold, void *ws, size_t ws_bytes,
                    int64_t splits, void *stream) {
  return cvm_sweep_fold_range(offsets, n_folds, 0, n_folds, K, M, dtype, flags, ddof, resolution, weighted, G, H,
                              gstats, out_XTX, out_XTY, out_muX, out_sdX, out_muY, out_sdY, out_fold, ws, ws_bytes,
                              splits, stream);
}

size_t cvm_partition_workspace_bytes(int64_t N, int64_t n_labels) { return partition_workspace_bytes(N, n_labels); }

int cvm_partition_labels(const int64_t *labels, int64_t N, int64_t n_labels, int64_t *idx_out,
                         int64_t *offsets_out, int64_t *first_out, int32_t *err_flag, void *ws,
                         size_t ws_bytes, void *stream) {
  if (!labels || !idx_out || !offsets_out || !first_out || !err_flag || !ws || N < 0)
    return fail(CVM_EINVAL, "cvm_partition_labels: bad argument%s");
  return partition_impl(labels, N, n_labels, idx_out, offsets_out, first_out, err_flag, ws, ws_bytes,
                        (hipStream_t)stream);
}

int cvm_partition_periodic(const int64_t *labels, int64_t N, int64_t n_labels, const void *w, int dtype,
                           int64_t *idx_out, int64_t *offsets_out, int64_t *nz_out, int32_t *not_periodic, void *stream) {
  if (!labels || !idx_out || !offsets_out || !not_periodic || N < 1)
    return fail(CVM_EINVAL, "cvm_partition_periodic: bad argument%s");
  if (dtype != CVM_F64 && dtype != CVM_F32) return fail(CVM_EINVAL, "cvm_partition_periodic: dtype must be CVM_F32 or CVM_F64%s");
  return partition_periodic_impl(labels, N, n_labels, w, dtype, idx_out, offsets_out, nz_out, not_periodic,
                                 (hipStream_t)stream);
}

int cvm_weights_check(const void *w, int64_t N, int dtype, int64_t *out2, void *stream) {
  if (!out2 || N < 0 || (N > 0 && !w)) return fail(CVM_EINVAL, "cvm_weights_check: bad argument%s");
  if (dtype != CVM_F64 && dtype != CVM_F32) return fail(CVM_EINVAL, "cvm_weights_check: dtype must be CVM_F32 or CVM_F64%s");
  return weights_check_impl(w, N, dtype, out2, (hipStream_t)stream);
}

size_t cvm_pls_workspace_bytes(int64_t n_folds, int K, int M, int A, int dtype) {
  if (n_folds < 0 || K <= 0 || M <= 0 || M > PLS_MAXM || A <= 0 || A > PLS_MAXA) return 0;
  return pls_workspace_bytes(n_folds, K, M, A, dtype == CVM_F64 ? 8 : 4, pls_cu_count());
}

int cvm_pls_fit(const void *XTX, const void *XTY, int64_t n_folds, int K, int M, int A, int dtype,
                void *B, void *W, void *P, void *Q, void *R, int32_t *n_fit, int32_t *status,
                void *ws, size_t ws_bytes, void *stream) {
  if (!XTX || !XTY || !B || !n_fit || !status || !ws) return fail(CVM_EINVAL, "cvm_pls_fit: null pointer%s");
  if (n_folds < 0 || K <= 0 || M <= 0 || A <= 0) return fail(CVM_EINVAL, "cvm_pls_fit: bad shape%s");
  if (M > PLS_MAXM) return fail(CVM_EINVAL, "cvm_pls_fit: at most 64 responses%s");
  if (A > PLS_MAXA) return fail(CVM_EINVAL, "cvm_pls_fit: at most 512 components%s");
  if (dtype == CVM_F64)
    return pls_fit_impl<double>(XTX, XTY, n_folds, K, M, A, B, W, P, Q, R, n_fit, status, ws, ws_bytes, (hipStream_t)stream);
  if (dtype == CVM_F32)
    return pls_fit_impl<float>(XTX, XTY, n_folds, K, M, A, B, W, P, Q, R, n_fit, status, ws, ws_bytes, (hipStream_t)stream);
  return fail(CVM_EINVAL, "cvm_pls_fit: dtype must be CVM_F32 or CVM_F64%s");
}

size_t cvm_pls_sse_workspace_bytes(int64_t n_folds, int64_t max_fold_rows, int M, int A) {
  if (n_folds < 0 || max_fold_rows < 0 || M <= 0 || A <= 0) return 0;
  return pls_sse_workspace_bytes(n_folds, max_fold_rows, M, A);
}

int cvm_pls_validation_sse(const void *X, const void *Y, const void *w, const int64_t *idx, const int64_t *offsets,
                           int64_t n_folds, int64_t max_fold_rows, int K, int M, int A, int dtype, const void *muX,
                           const void *sdX, const void *muY, const void *sdY, const void *B, double *sse,
                           double *wsum, void *ws, size_t ws_bytes, void *stream) {
  if (!X || !Y || !offsets || !B || !sse || !wsum || !ws) return fail(CVM_EINVAL, "cvm_pls_validation_sse: null pointer%s");
  if (n_folds < 0 || max_fold_rows < 0 || K <= 0 || M <= 0 || A <= 0 || (!idx && max_fold_rows > 0))
    return fail(CVM_EINVAL, "cvm_pls_validation_sse: bad shape%s");
  if (dtype == CVM_F64)
    return pls_sse_impl<double>(X, Y, w, idx, offsets, n_folds, max_fold_rows, K, M, A, muX, sdX, muY, sdY, B, sse, wsum, ws,
                                ws_bytes, (hipStream_t)stream);
  if (dtype == CVM_F32)
    return pls_sse_impl<float>(X, Y, w, idx, offsets, n_folds, max_fold_rows, K, M, A, muX, sdX, muY, sdY, B, sse, wsum, ws,
                               ws_bytes, (hipStream_t)stream);
  return fail(CVM_EINVAL, "cvm_pls_validation_sse: dtype must be CVM_F32 or CVM_F64%s");
}

int cvm_pls_plan(int64_t n_folds, int K, int M, int A, int dtype, int64_t *info) {
  if (!info || n_folds < 0 || K <= 0 || M <= 0 || M > PLS_MAXM || A <= 0 || A > PLS_MAXA)
    return fail(CVM_EINVAL, "cvm_pls_plan: bad argument%s");
  PlsPlan p;
  if (!make_pls_plan(n_folds, K, M, A, dtype == CVM_F64 ? 8 : 4, pls_cu_count(), p))
    return fail(CVM_EINVAL, "cvm_pls_plan: K too large for the LDS plan%s");
  info[0] = p.S; info[1] = p.rows; info[2] = p.folds_per_launch; info[3] = p.xres; info[4] = (int64_t)p.lds;
  PlsRepPlan r;
  if (make_pls_rep_plan(n_folds, K, M, A, dtype == CVM_F64 ? 8 : 4, pls_cu_count(), r)) {      // few folds: the one-barrier kernel
    info[0] = r.S; info[1] = r.rows; info[2] = r.folds_per_launch; info[3] = 2; info[4] = (int64_t)r.lds;
  }
  return CVM_OK;
}

int cvm_timing_enable(int on) {
  std::lock_guard<std::mutex> lk(g_timing_mu);
  g_timing.store(on != 0);
  g_ntimed = 0;
  return CVM_OK;
}

static int timing_collect(double *ms, int64_t *n) {      // ms[4], n[4] by kind; resets the list
  std::lock_guard<std::mutex> lk(g_timing_mu);
  for (int k = 0; k < 4; ++k) { ms[k] = 0; n[k] = 0; }
  for (int i = 0; i < g_ntimed; ++i) {
    HIP_OK(hipEventSynchronize(g_timed[i].b));
    float t = 0;
    HIP_OK(hipEventElapsedTime(&t, g_timed[i].a, g_timed[i].b));
    const int k = g_timed[i].kind;
    if (k >= 0 && k < 4) { ms[k] += t; ++n[k]; }
  }
  g_ntimed = 0;
  return CVM_OK;
}

int cvm_timing_read(double *ms_fit, int64_t *n_fit, double *ms_fold, int64_t *n_fold) {
  double ms[4];
  int64_t n[4];
  const int rc = timing_collect(ms, n);
  if (rc != CVM_OK) return rc;
  if (ms_fit) *ms_fit = ms[0];
  if (n_fit) *n_fit = n[0];
  if (ms_fold) *ms_fold = ms[1];
  if (n_fold) *n_fold = n[1];
  return CVM_OK;
}

int cvm_timing_read_kinds(double *ms4, int64_t *n4) {
  if (!ms4 || !n4) return fail(CVM_EINVAL, "cvm_timing_read_kinds: null pointer%s");
  return timing_collect(ms4, n4);
}

int cvm_clock_probe(void *device_buf, size_t bytes) {
  if (device_buf && ((uintptr_t)device_buf % 8 || bytes < 32)) return fail(CVM_EINVAL, "cvm_clock_probe: an 8-byte aligned buffer of at least 32 bytes%s");
  const size_t wgs = device_buf ? bytes / 32 : 0;
  g_clock_buf.store(nullptr, std::memory_order_release);
  g_clock_wgs.store((int)(wgs > 65536 ? 65536 : wgs), std::memory_order_relaxed);
  g_clock_buf.store((unsigned long long *)device_buf, std::memory_order_release);
  return CVM_OK;
}

int cvm_fill_probe(void *buf, size_t bytes, void *stream) {
  if (!buf || bytes % 16 || (uintptr_t)buf % 16) return fail(CVM_EINVAL, "cvm_fill_probe: 16-byte pieces%s");
  const size_t pieces = bytes / 16;
  if (!pieces) return CVM_OK;
  hipLaunchKernelGGL(fill_probe_kernel, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (float *)buf, pieces);
  HIP_OK(hipGetLastError());
  return CVM_OK;
}

int cvm_debug_force_splits(int s_off, int s_diag) {
  if (s_off == 0 && s_diag == 0) { g_force_splits.store(0, std::memory_order_relaxed); return CVM_OK; }
  if (s_off < 1 || s_diag < 1 || s_off > 65535 || s_diag > 65535) return fail(CVM_EINVAL, "cvm_debug_force_splits: 1 <= splits <= 65535, or 0, 0%s");
  g_force_splits.store(((unsigned)s_off << 16) | (unsigned)s_diag, std::memory_order_relaxed);
  return CVM_OK;
}

int cvm_debug_resident(int mode) {
  if (mode < 0 || mode > 2) return fail(CVM_EINVAL, "cvm_debug_resident: 0 (never), 1 (wherever the shape allows) or 2 (the default rule)%s");
  g_resident.store(mode, std::memory_order_relaxed);
  return CVM_OK;
}

int cvm_plan_fold(int64_t n_folds, int64_t max_fold_rows, int K, int M, int dtype, unsigned flags,
                  size_t ws_bytes, int64_t *info) {
  if (!info || K <= 0 || M < 0) return fail(CVM_EINVAL, "cvm_plan_fold: bad argument%s");
  Plan p;
  const bool fold_mode = (flags & 0x80000000u) == 0;   // bit 31 set: plan the fit stage
  int rc = make_plan(n_folds, max_fold_rows, K, M, dtype, flags & 0x7fffffffu, ws_bytes, fold_mode, p);
  if (rc != CVM_OK) return fail(rc, "cvm_plan_fold: workspace too small%s");
  info[0] = p.s_off;
  {
    const int per0 = p.g.diag_only ? 0 : p.g.nTiles - p.g.P, per1 = p.g.P * p.g.Yc;
    info[1] = (int64_t)p.folds_per_batch * ((int64_t)p.s_off * per0 + (int64_t)p.s_diag * per1);
  }
  info[6] = p.s_diag;
  info[7] = p.splits;
  info[2] = p.g.P;
  info[3] = p.g.nT;
  info[4] = p.folds_per_batch;
  // MFMA instructions issued per 4 rows of one unit (executed work, incl. padding)
  int64_t per4 = 0;
  // MFMAs per k-step: an off-diagonal tile 4 waves x 16; a diagonal tile in the float64 LDS-DMA
  // kernel 36 (the upper triangle of its 8 x 8 grid) + 8 per 16 live columns of the first Y
  // chunk; in the general kernel, or when folds are finished in the epilogue, 3 blocks of 16 +
  // 16 for XTY
  const int es_ = dtype == CVM_F64 ? 8 : 4;
  const bool tri = ((size_t)K * es_) % 16 == 0 && (es_ == 4 || M % 2 == 0) &&
                   !(es_ == 8 && fold_mode && p.splits == 1);
  if (!p.g.diag_only) per4 += (int64_t)(p.g.nTiles - p.g.P) * 64 + (int64_t)p.g.P * (tri ? 36 : 48);
  if (M > 0) {
    if (tri && !p.g.diag_only) per4 += (int64_t)p.g.P * (M > 16 ? 16 : 8) + (int64_t)p.g.P * (p.g.Yc - 1) * 16;
    else per4 += (int64_t)p.g.P * p.g.Yc * 16;
  }
  info[5] = per4;
  return CVM_OK;
}

}  // extern "C"
