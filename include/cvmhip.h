/*
 * cvmhip.h -- C ABI of libcvmhip.so: the MI355X (gfx950) implementation of the cvmatrix
 * per-fold training-matrix hot path.
 *
 * The reference (sm00thix/cvmatrix v3.2.1) has no FFI of its own: its seam is the array
 * namespace chosen by `_resolve_backend` (cvmatrix/cvmatrix.py:58-96) and used through
 * `self.xp` by the private methods of `CVMatrix`.  The two entry points below replace,
 * for a backend literal "hip", exactly the two fused stages behind that seam:
 *
 *   cvm_gram_fit     <- CVMatrix.fit(): _init_weighted_mats, _init_matrix_products,
 *                       _init_stats                      (cvmatrix.py:1193-1243)
 *   cvm_fold_update  <- CVMatrix._training_matrices / training_statistics for a BATCH of
 *                       folds: _get_val_matrices, _compute_training_stats,
 *                       _training_kernel_matrix          (cvmatrix.py:589-1129)
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'ed / torch-ROCm storage) unless it is
 *     named host_*; row-major, contiguous, no strides;
 *   - the library never allocates or frees device memory and keeps no pointer after a
 *     call returns; scratch comes from the caller's workspace `ws`;
 *   - calls enqueue work on `stream` (a hipStream_t passed as void*, NULL = default
 *     stream) and return without synchronising;
 *   - return value: CVM_OK or a CVM_E* code; cvm_last_error() gives the text for the
 *     calling thread;
 *   - dtype: CVM_F64 or CVM_F32 (the reference accepts any NumPy float; f16/f128 are not
 *     offered on the device).  Column statistics are always kept in float64.
 *   - results are deterministic: no floating-point atomics, fixed-order reductions;
 *   - threads: every entry point may be called concurrently from several host threads (one
 *     per stream / device is the intended use).  The library's only process-wide state is the
 *     optional launch-timing recorder below (mutex-guarded), once-per-device kernel
 *     attributes (atomic flags) and one 1 MiB device allocation per device, made on first use
 *     and kept (1024 work-queue blocks of the persistent Gram kernel, one per stream that uses
 *     it, handed out under a mutex and RECYCLED: when all are taken, the least recently used
 *     block whose last Gram launch is over -- by an event the library itself recorded behind
 *     that launch, once the process has used more than 64 streams on the device; never by a
 *     query of a foreign stream handle -- changes hands, so a service may create and destroy
 *     streams without bound); the optional clock-probe buffer and forced split plan are single
 *     atomic words; planning depends on the arguments alone.
 */
#ifndef CVMHIP_H
#define CVMHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CVM_OK 0
#define CVM_EINVAL 1   /* bad argument */
#define CVM_EWORKSPACE 2 /* workspace too small */
#define CVM_ELAUNCH 3  /* HIP launch / runtime error */

#define CVM_F32 0
#define CVM_F64 1

/* flags for cvm_fold_update (cvmatrix.py:157-163 constructor flags + what to return) */
#define CVM_RET_XTX 0x01u
#define CVM_RET_XTY 0x02u
#define CVM_CENTER_X 0x04u
#define CVM_CENTER_Y 0x08u
#define CVM_SCALE_X 0x10u
#define CVM_SCALE_Y 0x20u
/* cvm_fold_update only: `idx` and `offsets` are HOST pointers.  For one fold of at most 32 rows
 * (the reference's call pattern in leave-one-out: one training_XTX_XTY(validation_indices) per
 * sample, cvmatrix.py:451-517): the indices travel inside the kernel arguments and no device
 * copy of the index array is made. */
#define CVM_IDX_HOST 0x40u

const char *cvm_version(void);
/* sha256 (first 16 hex digits) over the source files the library was built from, as computed by
 * cvmatrix_amd/build.py; the Python loader refuses (or rebuilds) a library whose hash differs from
 * the sources lying next to it. */
const char *cvm_source_hash(void);
const char *cvm_last_error(void);

/* Number of float64 entries of the global statistics vector written by cvm_gram_fit and
 * read by cvm_fold_update:  [ sX(K) | qX(K) | sY(M) | qY(M) | sw | nz ]
 *   sX = sum_i w_i x_i   (cvmatrix.py:1231)      qX = sum_i w_i x_i^2   (1235-1236)
 *   sY = sum_i w_i y_i   (1233)                  qY = sum_i w_i y_i^2   (1240-1241)
 *   sw = sum_i w_i       (1225 / 1228)           nz = #{w_i != 0}       (1226 / 1229)
 * With w == NULL: w_i = 1, sw = nz = N. */
size_t cvm_gstats_len(int K, int M);

/* Workspace (bytes) that lets cvm_gram_fit run N rows in one launch sequence. */
size_t cvm_fit_workspace_bytes(int64_t N, int K, int M, int dtype);

/* Full-data Gram and column statistics (replaces cvmatrix.py:1193-1243).
 *   X [N,K], Y [N,M] or NULL (then M must be 0), w [N] or NULL (unweighted)
 *   G [K,K]  = X^T diag(w) X   (exactly symmetric)
 *   H [K,M]  = X^T diag(w) Y   (untouched if Y == NULL)
 *   gstats   float64[cvm_gstats_len(K,M)] as laid out above
 *   neg_flag int32[1], set to 1 if any w_i < 0 (cvmatrix.py:1188), else 0; may be NULL */
int cvm_gram_fit(const void *X, const void *Y, const void *w, int64_t N, int K, int M,
                 int dtype, void *G, void *H, double *gstats, int32_t *neg_flag,
                 void *ws, size_t ws_bytes, void *stream);

/* Workspace (bytes) recommended for a cvm_fold_update call: with it all folds are
 * processed in one batch.  A smaller workspace is legal as long as it holds one fold;
 * the call then walks the folds in several batches. */
size_t cvm_fold_workspace_bytes(int64_t n_folds, int64_t n_idx, int64_t max_fold_rows,
                                int K, int M, int dtype, unsigned flags);

/* Training-set matrices for a batch of folds (replaces, per fold, cvmatrix.py:754-896).
 *   idx      int64[n_idx]      validation row numbers of all folds, concatenated
 *                              (each in [0,N); duplicates are subtracted twice, as
 *                              NumPy fancy indexing would)
 *   offsets  int64[n_folds+1]  DEVICE: fold f owns idx[offsets[f] .. offsets[f+1])
 *   host_offsets               the same array in HOST memory (sizes the launch)
 *   G,H,gstats                 outputs of cvm_gram_fit (replicated on every GPU)
 *   ddof, resolution           cvmatrix.py:172, 187
 * outputs (any may be NULL if not wanted; *_XTX needs CVM_RET_XTX etc.):
 *   out_XTX [n_folds,K,K], out_XTY [n_folds,K,M]           dtype
 *   out_muX,out_sdX [n_folds,K], out_muY,out_sdY [n_folds,M]  dtype
 *   out_fold float64[n_folds,4] = { sw_train, nz_train, sw_val, nz_val }
 * The kernels never raise: the caller turns nz_train == 0 / nz_train <= ddof into the
 * reference's ValueErrors (cvmatrix.py:625-629, 1074-1078).  Statistics are computed
 * whenever a centre/scale flag asks for them (same conditions as cvmatrix.py:828-831);
 * un-requested statistic outputs are left untouched. */
int cvm_fold_update(const void *X, const void *Y, const void *w, const int64_t *idx,
                    const int64_t *offsets, const int64_t *host_offsets,
                    int64_t n_folds, int64_t N, int K, int M, int dtype, unsigned flags,
                    double ddof, double resolution, const void *G, const void *H,
                    const double *gstats, void *out_XTX, void *out_XTY, void *out_muX,
                    void *out_sdX, void *out_muY, void *out_sdY, double *out_fold,
                    void *ws, size_t ws_bytes, void *stream);

/* cvm_fold_update with a status word (the reference's contract is "raise or return correct numbers",
 * cvmatrix.py:625-629, 1074-1078: this is how a caller proves the second half).
 *   status   DEVICE int32[1], zeroed by the caller before the call, or NULL.
 * One route of the fold stage -- mid-size folds that form their statistics inside the Gram launch -- lets
 * work items wait for flags that other items of the same launch raise.  The wait is bounded; an item whose
 * wait gives up writes nothing and is recomputed by a second launch of the same call once every flag is up.
 * Behind the call (on `stream`):  *status == 0  nothing gave up;  2  some items were recomputed, every output
 * is valid;  1  an item gave up in the second launch too (not possible by the kernel's own logic: a fault) and
 * its outputs hold NaN.  Every other route leaves *status untouched. */
int cvm_fold_update_ex(const void *X, const void *Y, const void *w, const int64_t *idx,
                       const int64_t *offsets, const int64_t *host_offsets,
                       int64_t n_folds, int64_t N, int K, int M, int dtype, unsigned flags,
                       double ddof, double resolution, const void *G, const void *H,
                       const double *gstats, void *out_XTX, void *out_XTY, void *out_muX,
                       void *out_sdX, void *out_muY, void *out_sdY, double *out_fold,
                       void *ws, size_t ws_bytes, void *stream, int32_t *status);

/* One-sweep cross-validation (no counterpart in the reference).  When the
 * folds PARTITION the rows -- every row of X in exactly one fold, as the reference's own
 * benchmark and README build them (benchmarks/benchmark.py:232, README.md:120-141) -- the
 * full-data matrices of cvm_gram_fit are the sum of the folds' validation matrices.
 *   cvm_sweep_fit    runs the Gram kernel once over all folds (rows gathered by idx), writes
 *                    G, H, gstats, neg_flag like cvm_gram_fit (as the sum over the folds of each
 *                    fold's own sum over its row splits) and leaves the per-fold partials in ws; *splits_out receives an
 *                    opaque token (the row-split plan) to be handed back to cvm_sweep_folds
 *   cvm_sweep_folds  = the finalize half of cvm_fold_update on those partials (same outputs),
 *                    valid while ws is untouched; `weighted` = 1 if cvm_sweep_fit got w != NULL
 * Together they do half the arithmetic of cvm_gram_fit + cvm_fold_update.  The caller
 * checks the partition property; ws must hold all folds (cvm_sweep_workspace_bytes). */
size_t cvm_sweep_workspace_bytes(int64_t n_folds, int64_t max_fold_rows, int K, int M, int dtype);
int cvm_sweep_fit(const void *X, const void *Y, const void *w, const int64_t *idx,
                  const int64_t *offsets, const int64_t *host_offsets, int64_t n_folds, int64_t N,
                  int K, int M, int dtype, void *G, void *H, double *gstats, int32_t *neg_flag,
                  void *ws, size_t ws_bytes, void *stream, int64_t *splits_out);
int cvm_sweep_folds(const int64_t *offsets, int64_t n_folds, int K, int M, int dtype, unsigned flags,
                    double ddof, double resolution, int weighted, const void *G, const void *H,
                    const double *gstats, void *out_XTX, void *out_XTY, void *out_muX, void *out_sdX,
                    void *out_muY, void *out_sdY, double *out_fold, void *ws, size_t ws_bytes,
                    int64_t splits, void *stream);

/* cvm_sweep_fit followed by cvm_sweep_folds over all folds, as one call (same arguments, same
 * results to the bit, the partials stay in ws for cvm_sweep_fold_range).  With at most 16 folds
 * and 16-byte aligned rows the finalize half runs as two launches that read every partial once:
 * a fold's update stays in registers while the full-data matrix is formed as the sum of the
 * folds' updates. */
int cvm_sweep_all(const void *X, const void *Y, const void *w, const int64_t *idx, const int64_t *offsets,
                  const int64_t *host_offsets, int64_t n_folds, int64_t N, int K, int M, int dtype, unsigned flags,
                  double ddof, double resolution, void *G, void *H, double *gstats, int32_t *neg_flag,
                  void *out_XTX, void *out_XTY, void *out_muX, void *out_sdX, void *out_muY, void *out_sdY,
                  double *out_fold, void *ws, size_t ws_bytes, void *stream, int64_t *splits_out);

/* The same for folds [fold0, fold0 + n_folds) of the n_total folds of the sweep; outputs are written
 * from index 0.  Serves the reference's one-call-per-fold loop (README.md:120-141) from the sweep's
 * partials: training_XTX_XTY(p.get_validation_indices(fold)) then costs two small finalize kernels. */
int cvm_sweep_fold_range(const int64_t *offsets, int64_t n_total, int64_t fold0, int64_t n_folds, int K, int M,
                         int dtype, unsigned flags, double ddof, double resolution, int weighted, const void *G,
                         const void *H, const double *gstats, void *out_XTX, void *out_XTY, void *out_muX,
                         void *out_sdX, void *out_muY, void *out_sdY, double *out_fold, void *ws, size_t ws_bytes,
                         int64_t splits, void *stream);

/* Device-side Partitioner (replaces cvmatrix/partitioner.py:89-107 for integer labels):
 *   labels      int64[N], one fold label per row, each in [0, n_labels) (up to 4096 labels: one
 *               stable counting sort; more, e.g. leave-one-out with a label per row: the same
 *               sort over 12-bit digits of the label, least significant first)
 *   idx_out     int64[N]           row numbers grouped by label, ascending inside a group
 *   offsets_out int64[n_labels+1]  label l owns idx_out[offsets_out[l] .. offsets_out[l+1])
 *   first_out   int64[n_labels]    first row of each label (N if the label does not occur): the
 *                                  reference orders its folds by first appearance
 *   err_flag    int32[1]           set to 1 if some label is outside [0, n_labels), else 0
 * The output is exactly the idx / offsets pair cvm_fold_update takes. */
size_t cvm_partition_workspace_bytes(int64_t N, int64_t n_labels);
int cvm_partition_labels(const int64_t *labels, int64_t N, int64_t n_labels, int64_t *idx_out,
                         int64_t *offsets_out, int64_t *first_out, int32_t *err_flag, void *ws,
                         size_t ws_bytes, void *stream);

/* The fit stage's weight validation for weights that live on the device -- cvmatrix/cvmatrix.py:1188-1189
 * (`any(weights < 0)` -> ValueError("Weights must be non-negative.")) and :1226 (`count_nonzero(weights)`):
 * out2 (device, int64[2]) receives [#(w < 0), #(w != 0)] as exact integer counts, by one small launch on
 * `stream`.  The host class copies the two words to pinned memory asynchronously and reads them when a
 * result of that fit is first handed out, so fit() neither reads the weights back nor waits for the device
 * (the reference's own JAX backend defers its data-dependent raises the same way, cvmatrix.py:621-625,
 * 1071-1074). */
int cvm_weights_check(const void *w, int64_t N, int dtype, int64_t *out2, void *stream);

/* Labels that are arange(N) % n_labels (the reference benchmark's folds, benchmarks/benchmark.py:232;
 * n_labels == N: leave-one-out) need no sort.  One launch: checks the labels (not_periodic[0] = 1 if
 * they are anything else -- the outputs are then garbage and the caller runs cvm_partition_labels),
 * writes idx_out int64[N] (fold-major, rows ascending inside a fold), offsets_out int64[n_labels + 1]
 * and, if nz_out is not NULL, every fold's number of rows with a non-zero weight (w NULL: its row
 * count) -- what cvm_fold_update's host-side checks need (cvmatrix.py:612-630).  Replaces
 * Partitioner._init_folds_dict (partitioner.py:89-107) for such labels. */
int cvm_partition_periodic(const int64_t *labels, int64_t N, int64_t n_labels, const void *w, int dtype,
                           int64_t *idx_out, int64_t *offsets_out, int64_t *nz_out, int32_t *not_periodic,
                           void *stream);

/* The step after the path: Improved Kernel PLS, algorithm #2 of Dayal &
 * MacGregor (1997), on the training matrices of a batch of folds where cvm_fold_update left them.
 * It is what the out-of-tree consumer named by the reference runs per fold (reference
 * README.md:23, cvmatrix/partitioner.py:27-31: `ikpls`, fast cross-validation); the reference
 * itself holds no PLS code, so there is no reference line to cite beyond those.
 *   XTX [n_folds][K][K], XTY [n_folds][K][M]   the out_XTX / out_XTY of cvm_fold_update (not modified)
 *   A                    components, 1 <= A <= 512;  M <= 64
 *   B   [n_folds][A][K][M]   B[f][a] = regression coefficients with a+1 components (required)
 *   W, P, R [n_folds][K][A], Q [n_folds][M][A]      weights / loadings / rotations (each may be NULL)
 *   n_fit  int32[n_folds]    components extracted (< A only if XTY deflated to zero: the rest stay 0)
 *   status int32[1]          0; the routes that cut a fold into slices need an otherwise idle device (the
 *                            slices wait for each other): if a slice found no place to run, every fold is
 *                            recomputed in the same call by one workgroup per fold and status is 2 (outputs
 *                            valid); 1 only where a fold does not fit one workgroup's LDS -- the outputs are
 *                            then NaN and n_fit -1, never half-written
 * Component signs are those of the dominant eigenvector found by repeated squaring of XTY^T XTY;
 * B does not depend on them.  Arithmetic in float64 for both dtypes. */
size_t cvm_pls_workspace_bytes(int64_t n_folds, int K, int M, int A, int dtype);
int cvm_pls_fit(const void *XTX, const void *XTY, int64_t n_folds, int K, int M, int A, int dtype,
                void *B, void *W, void *P, void *Q, void *R, int32_t *n_fit, int32_t *status,
                void *ws, size_t ws_bytes, void *stream);
/* Validation errors of the folds' PLS models -- the last step of the cross-validation the reference's
 * README describes (README.md:23), which its consumer ikpls leaves to the caller: for fold f, the model
 * with a+1 components and response m
 *   sse[f][a][m] = sum over the fold's validation rows i (idx / offsets as in cvm_fold_update) of
 *                  w_i * ( ((x_i - muX[f]) / sdX[f]) . B[f][a][:, m] * sdY[f][m] + muY[f][m] - y_im )^2
 *   wsum[f]      = sum w_i    (w NULL: the row count)
 * muX, sdX [n_folds][K], muY, sdY [n_folds][M] in `dtype`: the statistics outputs of cvm_fold_update
 * (NULL: no centring / no scaling).  B as cvm_pls_fit wrote it.  sse float64[n_folds][A][M], wsum
 * float64[n_folds].  MFMA in `dtype`, errors accumulated in float64 in a fixed order (no atomics).
 * RMSE of the cross-validation with a+1 components: sqrt(sum_f sse[f][a][m] / sum_f wsum[f]). */
size_t cvm_pls_sse_workspace_bytes(int64_t n_folds, int64_t max_fold_rows, int M, int A);
int cvm_pls_validation_sse(const void *X, const void *Y, const void *w, const int64_t *idx, const int64_t *offsets,
                           int64_t n_folds, int64_t max_fold_rows, int K, int M, int A, int dtype, const void *muX,
                           const void *sdX, const void *muY, const void *sdY, const void *B, double *sse,
                           double *wsum, void *ws, size_t ws_bytes, void *stream);
/* info[0]=row slices per fold, [1]=rows per slice, [2]=folds per launch, [3]=1 if the slice of XTX
 * stays in LDS, 0 if it is streamed, 2: few folds -- the kernel that keeps the small state of a fold
 * (deflated XTY, P, R) whole in every slice and passes ONE per-fold barrier per component, XTX
 * streamed from L2, the slices of a fold on one XCD; [4]=LDS bytes per workgroup */
int cvm_pls_plan(int64_t n_folds, int K, int M, int A, int dtype, int64_t *info);

/* Benchmark support: when enabled, a hipEvent pair is recorded on the launch stream around
 * every launch of the Gram kernel (at most 8192 pairs between reads).  cvm_timing_read
 * waits for the recorded events, returns the summed kernel milliseconds and launch counts
 * of the fit stage and of the fold stage since the last read, and resets the list.
 * One recorder per process (all streams, all threads; slots are handed out under a mutex):
 * meant for bench.py. */
int cvm_timing_enable(int on);
int cvm_timing_read(double *ms_fit, int64_t *n_fit, double *ms_fold, int64_t *n_fold);
/* The same list by kind: ms4 / n4 [0] Gram launches of the fit stage, [1] of the fold stage, [2] the
 * statistics kernel of the small-fold route (small_stats_kernel), [3] its update kernels (everything
 * a call launches after the statistics: tile / whole-rows kernel and the XTY panels). */
int cvm_timing_read_kinds(double *ms4, int64_t *n4);

/* Benchmark support: the write ceiling of this device, measured with the product's own kind of store --
 * one launch that fills `bytes` (a multiple of 16) from `buf` (16-byte aligned) with zeros by nontemporal
 * 16-byte stores in linear order.  bench.py times it on the output buffer of the HBM-regime shapes, in the
 * same run, so that a measured rate can be read against what THIS box's memory system takes
 * (`frac_of_fill` next to `frac`).  No reference counterpart: the reference has no device. */
int cvm_fill_probe(void *buf, size_t bytes, void *stream);

/* Benchmark support: the shader clock the chip holds UNDER the product's LDS-DMA Gram kernel (the kernel of
 * cvm_gram_fit / cvm_sweep_* / cvm_fold_update), read inside the shipped kernel itself.  While a caller-owned
 * device buffer is set (8-byte aligned; NULL switches the probe off, the default), every such launch of the
 * process makes workgroup b < bytes / 32 store four 64-bit words at buf + 32 b:
 *   [0] s_memtime, [1] s_memrealtime when the workgroup starts;  [2], [3] the same pair when it has run out of
 *   work items
 * -- two scalar clock reads per workgroup lifetime, nothing inside the item loop; results are unchanged.
 * ([2] - [0]) / ([3] - [1]) x 100 MHz is the average shader clock over that workgroup's life (s_memrealtime
 * ticks at 100 MHz whatever the chip does; MI355X_MICROARCH.md "DVFS give-back" item 6), ([3] - [1]) its
 * duration in 10 ns units.  bench.py reports the median over the workgroups of the last timed launch as
 * roofline.effective_clock_mhz.  The caller synchronises before it reads the buffer, and clears the probe
 * before it frees it.  No reference counterpart: the reference has no device. */
int cvm_clock_probe(void *device_buf, size_t bytes);

/* Experiments and tests: pin the row-split plan of every later call of this process to (s_off, s_diag) row
 * splits of the off-diagonal / diagonal tile items (clamped to what the problem allows); (0, 0) hands the
 * decision back to the planner.  Stored in one atomic word; the environment variable CVM_FORCE_SPLITS="a,b" is
 * read once, as its initial value.  Results stay within the parity bar under any plan (float32: one more
 * rounding per extra partial, cvmatrix_amd/fp32_gate.py).  No reference counterpart. */
int cvm_debug_force_splits(int s_off, int s_diag);

/* Experiments and tests: where float32 XTX batches of folds of at most 32 rows, K a multiple of 1024, take the round-6
 * "resident" kernel (csrc/resident.hpp: G in the register files of the whole chip, every tile computed directly, no
 * mirrored store).  mode 2 (the default): K = 2048 or a multiple of 4096 and at least 16 folds per workgroup set (16 folds per
 * batch from K = 4096 on, 64 at K = 2048), where it measures 6-13 % faster than the tile kernel; 1: wherever the shape allows (slower below K = 4096 and for few folds); 0: never.  The environment
 * variable CVM_RESIDENT=0|1|2 is read once, as the initial value.  Results stay within the float32 parity bar and exactly
 * symmetric on either route; size the workspace with cvm_fold_workspace_bytes AFTER changing the mode (the route keeps an
 * operand block per fold there; with a smaller workspace the folds are walked in smaller batches or take the tile
 * kernel).  No reference counterpart. */
int cvm_debug_resident(int mode);

/* Introspection for benchmarks/profiles: geometry chosen for a problem (info: int64[8]).
 * info[0]=row splits per fold of the off-diagonal 128x128 tiles, [6]=row splits of the diagonal
 * tiles (which also produce XTY and the column sums and cost less per row: the two kinds are cut
 * differently so that all workgroups of a launch take about the same time), [7]=partial slots per
 * fold in the workspace (the larger of the two), [1]=workgroups of the Gram kernel per batch,
 * [2]=column panels, [3]=work items per (fold, split) if both kinds were cut alike, [4]=folds per
 * batch, [5]=MFMA instructions the kernel issues per 4 rows of one fold (executed work, 2048 flop
 * each).  Bit 31 of `flags` set: plan the fit stage instead.  Same decision procedure as the real
 * calls. */
int cvm_plan_fold(int64_t n_folds, int64_t max_fold_rows, int K, int M, int dtype,
                  unsigned flags, size_t ws_bytes, int64_t *info);

#ifdef __cplusplus
}
#endif
#endif /* CVMHIP_H */
