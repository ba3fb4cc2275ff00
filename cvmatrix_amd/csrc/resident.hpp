// resident.hpp -- part of libcvmhip.so (included by cvmhip.hip inside its anonymous namespace).
// Round 6: the HBM regime with G RESIDENT IN THE REGISTER FILES OF THE WHOLE CHIP (float32 XTX of folds of at most 32 rows).
// (Three versions, kept in this order below: the operand kernel, the four-wave kernel of the first two versions -- CVM_RES_WAVES=4 --
//  and, at the end of the file, the eight-wave kernel that is the product.)
//
// small_apply_kernel keeps a 64 x 64 tile of G for eight folds, computes the upper triangle and writes it twice (direct +
// mirrored); its store pattern alone tops out at 5.1-5.4 TB/s (tools/xcd_stack_probe.hip) and every group of eight folds
// fetches G again.  Here a launch is 512 PERSISTENT workgroups (two per CU) that each own a 32-row x 1024-column block of G
// in registers -- 64 MiB over the chip: all of a K = 4096 float32 G; 64 registers per lane in the eight-wave kernel, 128 in
// the four-wave one -- for ALL folds of the call:
//   * G crosses the memory system once per launch, not once per fold group;
//   * every tile of the output is computed directly (both triangles: a 16-row fold keeps the matrix cores busy a third of
//     the time a fold's stores take), so there is no transposed copy through LDS and no mirrored store: a store
//     instruction writes two whole 128-byte lines straight from the accumulators of v_mfma_f32_32x32x2_f32 (32 lanes on a
//     row), eight of them side by side per row and wave;
//   * all workgroups are inside the SAME output matrix at any time (equal work per fold, no communication): the stores in
//     flight span one or two matrices instead of the eight a fold group spreads them over;
//   * bit-exact symmetry without mirroring: a tile above the diagonal multiplies (-w x)[row] * x[column], its mirror image
//     below the diagonal x[row] * (-w x)[column] -- the same products in the same k order; the diagonal tile computes
//     both and selects by element.  Centring is one more k-step ((sqrt(sw) mu)[row] * (sqrt(sw) mu)[column]), scaling ONE
//     multiply per element by sd^-1[row] * sd^-1[column], itself an MFMA outer product -- the accumulator chain starts
//     from the tile of G (C operand), so the finish costs one vector instruction per element.
// Operands: res_pack_kernel writes, per fold and 32-column tile, RB = NP + 4 rows x 32 columns  {x_0 .. x_NP-1 (zero rows
// beyond the fold), sqrt(sw) mu, 0, sd^-1, 0}  ("P") and then the same with {-w x, -sqrt(sw) mu, 0, sd^-1, 0} ("Q"); an MFMA
// operand of a tile is 2560 contiguous bytes (NP = 16) and arrives by three LDS-DMA instructions into the wave's own buffers; the
// row-side operands of a block are the same kind of operand (the block's 32 rows as columns of P and Q), fetched once per fold.
// Every step (two tiles) ends in a full drain of its wave's memory operations: hand-counted waits that never wait for a store were
// built first and measure SLOWER (CVM_RES_SAFE below) -- the drain is what keeps the workgroups of the chip in step.
// tools/resident_probe.hip is the measurement this is built on (profiles/r6/hbm_regime/resident_probe.txt).
#pragma once

constexpr int RES_NT = 8;                // 32-column tiles per wave (256 columns; a block is 4 waves = 1024 columns)
constexpr int RES_BC = 4 * RES_NT * 32;
constexpr int RES_WG = 512;              // resident workgroups: 2 per CU, 128 registers of G per lane
struct ResArgs {
  const void *G;
  void *out;
  const void *pk;                        // [fold of batch][2 RB][K]
  int K, nb;
  int64_t seg0;
  int nbc;                               // blocks per 32-row band (K / 1024)
  int blk0, nblk;                        // blocks of this launch
  int groups;                            // fold groups (workgroup sets that each hold the launch's blocks)
};

#ifndef CVM_RES_PROBE                    // (tools/res_kernel_probe.hip includes this file for res_apply_kernel alone)
// (with_xty: the launch also writes the fold's XTY -- M <= RES_XTY_M responses -- in small_apply_kernel's arithmetic: the sum over
//  the fold's rows of T(w x) * y in row order, then total - update, centring and scaling in float64)
constexpr int RES_XTY_M = 16;
template <typename T, int NP, bool WEIGHTED> __global__ __launch_bounds__(256) void res_pack_kernel(const SmallArgs a, T *pk, int with_xty) {
  constexpr int RB = NP + 4;
  const int f = blockIdx.x, K = a.K, M = a.M, tid = threadIdx.x;
  const int64_t o0 = a.offs[a.seg0 + f];
  const int n = (int)(a.offs[a.seg0 + f + 1] - o0);
  __shared__ int64_t rows[NP];
  __shared__ T wl[NP];
  __shared__ T yl[NP][RES_XTY_M];
  if (tid < NP) {
    const int64_t r = tid < n ? a.idx[o0 + tid] : 0;
    rows[tid] = r;
    wl[tid] = tid < n ? (WEIGHTED ? ((const T *)a.w)[r] : (T)1) : (T)0;
  }
  __syncthreads();
  if (with_xty)
    for (int e = tid; e < NP * RES_XTY_M; e += 256) {
      const int r = e / RES_XTY_M, m = e - r * RES_XTY_M;
      yl[r][m] = (r < n && m < M) ? ((const T *)a.Y)[rows[r] * (int64_t)M + m] : (T)0;
    }
  if (with_xty) __syncthreads();
  const double *fs = a.fstats + (size_t)f * fstat_len(K, M);
  const double swt = fs[2 * K + 2 * M];
  const bool cX = a.flags & CVM_CENTER_X, sX = a.flags & CVM_SCALE_X, cY = a.flags & CVM_CENTER_Y, sY = a.flags & CVM_SCALE_Y;
  const double rsw = sqrt(swt);
  // tile-major: column tile j = c / 32 holds its P operand (RB rows x 32 columns) and then its Q operand
  T *blk = pk + (size_t)f * 2 * RB * K;
  const T *X = (const T *)a.X;
  for (int c = blockIdx.y * 256 + tid; c < K; c += gridDim.y * 256) {
    T *P = blk + (size_t)(c >> 5) * (2 * RB * 32) + (c & 31), *Q = P + RB * 32;
    T wx[NP];
#pragma unroll
    for (int r = 0; r < NP; ++r) {
      const T x = r < n ? X[rows[r] * (int64_t)K + c] : (T)0;
      wx[r] = WEIGHTED ? (T)(wl[r] * x) : x;
      P[r * 32] = x;
      Q[r * 32] = r < n ? -wx[r] : (T)0;
    }
    const T cm = cX ? (T)(rsw * fs[c]) : (T)0, sd = sX ? (T)fs[K + c] : (T)1;
    P[NP * 32] = cm; Q[NP * 32] = -cm;
    P[(NP + 1) * 32] = (T)0; Q[(NP + 1) * 32] = (T)0;
    P[(NP + 2) * 32] = sd; Q[(NP + 2) * 32] = sd;
    P[(NP + 3) * 32] = (T)0; Q[(NP + 3) * 32] = (T)0;
    if (with_xty) {
      const T *Ht = (const T *)a.H;
      T *out = (T *)a.out_XTY + (size_t)(a.seg0 + f) * (size_t)K * M;
      for (int m = 0; m < M; ++m) {
        T s_ = 0;
#pragma unroll
        for (int r = 0; r < NP; ++r)
          if (r < n) s_ += wx[r] * yl[r][m];
        double v = (double)Ht[(size_t)c * M + m] - (double)s_;
        if (cX || cY) v -= swt * (fs[c] * fs[2 * K + m]);
        if (sX && sY) v = v * (fs[K + c] * fs[2 * K + M + m]);
        else if (sX) v = v * fs[K + c];
        else if (sY) v = v * fs[2 * K + M + m];
        out[(size_t)c * M + m] = (T)v;
      }
    }
  }
}
#endif

// all but the n youngest vector-memory operations of this wave are done (n wave-uniform, 0..63)
__device__ __forceinline__ void res_wait_vmcnt(int n) {
#define CVM_W(k) case k: asm volatile("s_waitcnt vmcnt(" #k ")" ::: "memory"); break;
#define CVM_W8(k) CVM_W(k) CVM_W(k + 1) CVM_W(k + 2) CVM_W(k + 3) CVM_W(k + 4) CVM_W(k + 5) CVM_W(k + 6) CVM_W(k + 7)
  switch (n) {
    CVM_W(0) CVM_W(1) CVM_W(2) CVM_W(3) CVM_W(4) CVM_W(5) CVM_W(6) CVM_W(7)
    CVM_W(8) CVM_W(9) CVM_W(10) CVM_W(11) CVM_W(12) CVM_W(13) CVM_W(14) CVM_W(15)
    CVM_W(16) CVM_W(17) CVM_W(18) CVM_W(19) CVM_W(20) CVM_W(21) CVM_W(22) CVM_W(23)
    CVM_W(24) CVM_W(25) CVM_W(26) CVM_W(27) CVM_W(28) CVM_W(29) CVM_W(30) CVM_W(31)
    CVM_W(32) CVM_W(33) CVM_W(34) CVM_W(35) CVM_W(36) CVM_W(37) CVM_W(38) CVM_W(39)
    CVM_W(40) CVM_W(41) CVM_W(42) CVM_W(43) CVM_W(44) CVM_W(45) CVM_W(46) CVM_W(47)
    CVM_W(48) CVM_W(49) CVM_W(50) CVM_W(51) CVM_W(52) CVM_W(53) CVM_W(54) CVM_W(55)
    CVM_W(56) CVM_W(57) CVM_W(58) CVM_W(59) CVM_W(60) CVM_W(61) CVM_W(62)
    default: asm volatile("s_waitcnt vmcnt(63)" ::: "memory"); break;
  }
#undef CVM_W8
#undef CVM_W
}

// How a step waits for its operands (same-box runs of tools/res_kernel_probe.hip, K = 4096, TB/s of outputs, 48 / 160 folds):
//   0  counted: vmcnt(32 stores of the previous step + 6 DMAs of the next) -- loads and stores retire in order, so the wait never
//      waits for a store                                                                              5.15-5.21 / 5.18-5.20
//   1  request the next step's operands, then vmcnt(0): every step drains its wave's stores AND waits out the round trip of the
//      operands it has just requested                                                                 5.22-5.34 / 5.65-5.67
//   2  vmcnt(0) first, then request the next step's operands (drained, but the operands a step ahead) 5.14 / 5.19-5.52
// The wave that stalls longest is the fastest: fewer stores in flight per wave keep the workgroups of the chip inside the same
// one or two output matrices (counted waits let them drift apart over a long batch).  1 is the product.
#ifndef CVM_RES_SAFE
#define CVM_RES_SAFE 1
#endif
#ifndef CVM_RES_DRAIN
#define CVM_RES_DRAIN 0                  // measurements: 1 = s_waitcnt vmcnt(0) behind every tile's stores, 2 = vmcnt(16) there
#endif
#ifndef CVM_RES_ABLATE
#define CVM_RES_ABLATE 0                 // probe builds only (wrong results by design): 1 no MFMA, 2 no DMA, 4 no stores, 8 no scaling multiply
#endif

template <int NP> __global__ __launch_bounds__(256, 2) void res_apply_kernel(const ResArgs a) {
  typedef float f16v __attribute__((ext_vector_type(16)));
  constexpr int RB = NP + 4, KK = NP / 2 + 1;       // rows of an operand block; k-pairs of the chain (rows and the centring term)
  constexpr int OPB = RB * 128;                     // bytes of a tile operand in LDS: RB rows x 32 floats
  constexpr int NI = (RB + 7) / 8;                  // LDS-DMA instructions per operand (8 rows x 128 B each)
#ifndef CVM_RES_TS
#define CVM_RES_TS 2                     // tiles per step: 2 = operands double-buffered, requested one step ahead; 4 = single-buffered
#endif                                   //   steps of four tiles (the request, the drain and the round trip twice per fold instead of four times):
                                         //   measured the same within the scatter (1.89-2.01 ms for 160 folds either way); 2 is the product
  constexpr int TS = CVM_RES_TS, SB = TS == 4;      // (SB: the step's own operands are requested at its top)
  constexpr int STEPS = RES_NT / TS, ST_STORES = 16 * TS;
  constexpr int WAVE_LDS = OPB * 7;                 // B[2][2] | the diagonal tile's second operand | A[2]
  extern __shared__ __attribute__((aligned(16))) char res_lds[];
  const int K = a.K, tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l32 = lane & 31, lh = lane >> 5;
  // workgroup -> XCD blockIdx % 8: contiguous ranges of (group, block) per XCD
  const unsigned lin = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int group = (int)(lin / (unsigned)a.nblk), bl = (int)(lin % (unsigned)a.nblk);
  if (group >= a.groups) return;
  const int blk = a.blk0 + bl, band = blk / a.nbc, ch = blk - band * a.nbc;
  const int r0 = band * 32, cw = ch * RES_BC + wave * (RES_NT * 32), jt0 = cw >> 5;
  const int nfm = (a.nb - group + a.groups - 1) / a.groups;        // folds of this workgroup: group, group + groups, ...
  if (nfm <= 0) return;
  const unsigned lds0 = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)res_lds) + (unsigned)(wave * WAVE_LDS);
  const float *ldsf = reinterpret_cast<const float *>(res_lds + wave * WAVE_LDS);
  // ---- the block of G: four (two) 32 x 32 tiles at a time by LDS-DMA into the (still unused) operand buffers -- a DMA instruction moves
  //      8 rows x 128 bytes (lane l: row l / 8, 16-byte piece l % 8), 16 instructions per group of four tiles instead of 64 two-line
  //      loads -- then into the MFMA's C layout by LDS reads (launch prologue 50 -> ~20 us: it is what a call of few folds pays)
  f16v g[RES_NT];
  {
    const unsigned vg = (unsigned)(((lane >> 3) * K) * 4 + (lane & 7) * 16);
    const char *Gb = (const char *)a.G + ((size_t)r0 * K + cw) * 4;
    constexpr int TGL = WAVE_LDS >= 16384 ? 4 : 2;          // tiles staged at a time (4 KB each)
#pragma unroll
    for (int t0 = 0; t0 < RES_NT; t0 += TGL) {
#pragma unroll
      for (int t = 0; t < TGL; ++t)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const char *sb = Gb + (size_t)(8 * i) * K * 4 + 128 * (t0 + t);
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                       "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(vg), "s"(sb), "s"(lds0 + (unsigned)(4096 * t + 1024 * i)) : "memory");
        }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int t = 0; t < TGL; ++t)
#pragma unroll
        for (int v = 0; v < 16; ++v) g[t0 + t][v] = ldsf[1024 * t + ((v & 3) + 8 * (v >> 2) + 4 * lh) * 32 + l32];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  const unsigned vdma = (unsigned)(lane * 16);                                 // lane l of a DMA: 16-byte piece l of 1 KiB
  const unsigned vout = (unsigned)((4 * lh * K + l32) * 4);
  const size_t pkf = (size_t)2 * RB * K * 4;                                   // bytes of a fold's pack block
  const size_t K4 = (size_t)K * 4, K20 = (size_t)K * 20;
  // `bytes` contiguous bytes of the pack block (a multiple of 512: an operand is 2560, the two of a tile 5120) -> LDS
  auto dma_run = [&](const char *src, int bytes, unsigned lds_addr) {
    if (CVM_RES_ABLATE & 2) return;
#pragma unroll
    for (int o = 0; o < bytes; o += 1024) {
      unsigned keep;
      if (bytes - o >= 1024) {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(vdma), "s"(src + o), "s"(lds_addr + (unsigned)o) : "memory");
      } else {
        unsigned long long ex;
        const unsigned long long mask = (1ull << ((bytes - o) / 16)) - 1;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b64 exec, %5\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(ex) : "v"(vdma), "s"(src + o), "s"(lds_addr + (unsigned)o), "s"(mask) : "memory");
      }
    }
  };
  // the pack block of a fold is tile-major: column tile j holds its P operand (RB x 32 floats) and then its Q operand
  // the operands of step s of the fold at `fb` -> B buffers of parity `par`
  auto issue_step = [&](const char *fb, int s, int par) {
#pragma unroll
    for (int tt = 0; tt < TS; ++tt) {
      const int t = TS * s + tt, j = jt0 + t;
      const char *tb = fb + (size_t)j * (2 * OPB);
      // above the diagonal (and on it): B = x (P), A = -w x; below: B = -w x (Q), A = x
      dma_run(tb + (j >= band ? 0 : OPB), OPB, lds0 + (unsigned)(((SB ? 0 : 2 * par) + tt) * OPB));
      if (j == band) dma_run(tb + OPB, OPB, lds0 + 4u * OPB);
    }
  };
  // the A side: the block's 32 rows as column tile `band` of P and Q (one run of 2 OPB bytes: LDS slots 5 = P, 6 = Q)
  auto issue_A = [&](const char *fb) { dma_run(fb + (size_t)band * (2 * OPB), 2 * OPB, lds0 + 5u * OPB); };
  const char *pk0 = (const char *)a.pk + (size_t)group * pkf;
  const size_t pk_step = (size_t)a.groups * pkf;
  char *out0 = (char *)a.out + ((size_t)(a.seg0 + group) * K * K + (size_t)r0 * K + cw) * 4;
  const size_t out_step = (size_t)a.groups * K * K * 4;
  issue_A(pk0);
  if (!SB) issue_step(pk0, 0, 0);
  float aQ[KK], aP[KK], aSd = 0.f;
  // Waits.  In front of step g the wave's queue holds, oldest first: the operands of step g (issued one step ago), at
  // most one A-side run, the 32 stores of step g - 1, the operands of step g + 1 (2 NI instructions; 3 NI for the one
  // step of a fold that holds the diagonal tile: the constant below then also waits for 3 of the stores -- once per fold
  // in one wave of one workgroup per band).
  constexpr int NST = (CVM_RES_ABLATE & 4) ? 0 : ST_STORES, NDMA = (CVM_RES_ABLATE & 2) ? 0 : 2 * NI;
  for (int fi = 0; fi < nfm; ++fi) {
    const char *fb = pk0 + (size_t)fi * pk_step;
    char *ob = out0 + (size_t)fi * out_step;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      const int par = s & 1;                       // STEPS is even: the parity of a step does not depend on the fold
      // ---- this step's operands have arrived; next step's are requested ----
#if CVM_RES_TS == 4
      issue_step(fb, s, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#elif CVM_RES_SAFE == 2
      // drain FIRST (this step's operands were requested a step ago, in front of that step's stores), then request the next
      // step's: the operands still travel a whole step ahead, and no store of an earlier step is in flight beside this step's
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (s + 1 < STEPS) issue_step(fb, s + 1, par ^ 1);
      else if (fi + 1 < nfm) issue_step(fb + pk_step, 0, par ^ 1);
#else
      if (s + 1 < STEPS) issue_step(fb, s + 1, par ^ 1);
      else if (fi + 1 < nfm) issue_step(fb + pk_step, 0, par ^ 1);
#if CVM_RES_SAFE
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
      if (s == 0 && fi == 0) res_wait_vmcnt(NDMA);
      else if (s + 1 == STEPS && fi + 1 == nfm) res_wait_vmcnt(NST);
      else res_wait_vmcnt(NST + NDMA);
#endif
#endif
      if (s == 0) {
        const float *A0 = ldsf + 6 * (OPB / 4), *A1 = ldsf + 5 * (OPB / 4);
#pragma unroll
        for (int kk = 0; kk < KK; ++kk) { aQ[kk] = A0[(2 * kk + lh) * 32 + l32]; aP[kk] = A1[(2 * kk + lh) * 32 + l32]; }
        aSd = A0[(NP + 2 + lh) * 32 + l32];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (fi + 1 < nfm) issue_A(fb + pk_step);   // (older than this step's stores: the next step's wait covers it)
      }
#pragma unroll
      for (int tt = 0; tt < TS; ++tt) {
        const int t = TS * s + tt, j = jt0 + t;
        const float *B = ldsf + ((SB ? 0 : 2 * par) + tt) * (OPB / 4);
        const bool up = j >= band;
        f16v acc = g[t];
#pragma unroll
        for (int kk = 0; kk < ((CVM_RES_ABLATE & 1) ? 1 : KK); ++kk)
          acc = __builtin_amdgcn_mfma_f32_32x32x2f32(up ? aQ[kk] : aP[kk], B[(2 * kk + lh) * 32 + l32], acc, 0, 0, 0);
        f16v ps;
#pragma unroll
        for (int v = 0; v < 16; ++v) ps[v] = 0.f;
        ps = __builtin_amdgcn_mfma_f32_32x32x2f32(aSd, B[(NP + 2 + lh) * 32 + l32], ps, 0, 0, 0);
        if (j == band) {
          // the diagonal tile: elements below the diagonal from the mirrored product x[row] * (-w x)[column]
          const float *B2 = ldsf + 4 * (OPB / 4);
          f16v acc2 = g[t];
#pragma unroll
          for (int kk = 0; kk < KK; ++kk)
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(aP[kk], B2[(2 * kk + lh) * 32 + l32], acc2, 0, 0, 0);
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int row = (v & 3) + 8 * (v >> 2) + 4 * lh;
            acc[v] = row > l32 ? acc2[v] : acc[v];
          }
        }
        // finish: one (packed) multiply per element; a store writes rows r, r + 4 of the tile (two whole 128-byte lines);
        // the row pointer advances by scalar adds (rows 0-3, 8-11, 16-19, 24-27 of the lane half)
        typedef float f2v __attribute__((ext_vector_type(2)));
        const char *obv = ob + 128 * t;
#pragma unroll
        for (int v = 0; v < 16; v += 2) {
          f2v val = {acc[v], acc[v + 1]};
          if (!(CVM_RES_ABLATE & 8)) val = val * (f2v){ps[v], ps[v + 1]};
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            if (!(CVM_RES_ABLATE & 4) || val[e] == 12345.678f)
              asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(vout), "v"(val[e]), "s"(obv) : "memory");
            obv += ((v + e) & 3) == 3 ? K20 : K4;
            asm volatile("" : "+s"(obv));
          }
        }
        if (CVM_RES_DRAIN == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (CVM_RES_DRAIN == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------------------------------------
// Third version (round 6): the same 32 x 1024 block per workgroup, EIGHT waves of four tiles each -- 64 registers of G per lane,
// 512 threads, two workgroups per CU = four waves per SIMD instead of two: while one wave is held at the issue of its MFMA
// chain (640 cycles per tile of a 16-row fold) three others can issue stores.  What had to give: the row-side operands are
// shared by the workgroup (one LDS copy per fold, requested by wave 0 a fold ahead, handed over behind ONE LDS-only barrier
// per fold) and read from LDS per tile instead of living in registers; a step is two tiles whose operands are requested at its
// top (single-buffered: csrc comment on CVM_RES_TS -- the same rate as requesting a step ahead once every step drains anyway).
// LDS per workgroup: A[2 folds][P | Q] | the diagonal tile's second operand | B[8 waves][2 tiles] = 21 operands (53.8 KB).
// Folds of 17 to 32 rows take operand blocks of 36 rows and one tile per step (13 operands, 59.9 KB; 17 + 1 k-pairs per tile).
constexpr int RES8_NT = 4;
// (folds of 17 to 32 rows: operand blocks of 36 rows, ONE tile per step -- 13 operands of 4.6 KB per workgroup)
template <int NP> constexpr int res8_ts() { return NP <= 16 ? 2 : 1; }
template <int NP> constexpr int res8_lds() { return (5 + 8 * res8_ts<NP>()) * (NP + 4) * 128; }
template <int NP> __global__ __launch_bounds__(512, 2) void res8_apply_kernel(const ResArgs a) {
  typedef float f16v __attribute__((ext_vector_type(16)));
  constexpr int RB = NP + 4, KK = NP / 2 + 1, OPB = RB * 128;
  constexpr int TS = res8_ts<NP>(), STEPS = RES8_NT / TS;
  extern __shared__ __attribute__((aligned(16))) char res_lds[];
  const int K = a.K, tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, l32 = lane & 31, lh = lane >> 5;
  const unsigned lin = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int group = (int)(lin / (unsigned)a.nblk), bl = (int)(lin % (unsigned)a.nblk);
  if (group >= a.groups) return;
  const int blk = a.blk0 + bl, band = blk / a.nbc, ch = blk - band * a.nbc;
  const int r0 = band * 32, cw = ch * RES_BC + wave * (RES8_NT * 32), jt0 = cw >> 5;
  const int nfm = (a.nb - group + a.groups - 1) / a.groups;
  if (nfm <= 0) return;
  const unsigned ldsA = (unsigned)(uintptr_t)((__attribute__((address_space(3))) char *)res_lds);
  const unsigned ldsD = ldsA + 4u * OPB, ldsB = ldsA + 5u * OPB + (unsigned)(wave * TS * OPB);
  const float *fA = reinterpret_cast<const float *>(res_lds), *fD = fA + 4 * (OPB / 4), *fB = fA + (5 + TS * wave) * (OPB / 4);
  // ---- the block of G, a tile (half a tile) at a time through this wave's operand buffers ----
  f16v g[RES8_NT];
  {
    const unsigned vg = (unsigned)(((lane >> 3) * K) * 4 + (lane & 7) * 16);
    const char *Gb = (const char *)a.G + ((size_t)r0 * K + cw) * 4;
    constexpr int ROWS_AT_ONCE = TS * OPB >= 4096 ? 32 : 16;         // (8-row operands: 3 KB per wave -> half a tile at a time)
    static_assert(TS * OPB >= 2048, "staging");
#pragma unroll
    for (int t = 0; t < RES8_NT; ++t)
#pragma unroll
      for (int h = 0; h < 32; h += ROWS_AT_ONCE) {
#pragma unroll
        for (int i = 0; i < ROWS_AT_ONCE / 8; ++i) {
          const char *sb = Gb + (size_t)(h + 8 * i) * K * 4 + 128 * t;
          unsigned keep;
          asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                       "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                       : "=&s"(keep) : "v"(vg), "s"(sb), "s"(ldsB + (unsigned)(1024 * i)) : "memory");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          const int row = (v & 3) + 8 * (v >> 2);                    // + 4 lh; rows h .. h + ROWS_AT_ONCE - 1 are staged
          if (ROWS_AT_ONCE == 32 || (row >= h && row < h + ROWS_AT_ONCE))
            g[t][v] = fB[(row - h + 4 * lh) * 32 + l32];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
  }
  const unsigned vdma = (unsigned)(lane * 16);
  const unsigned vout = (unsigned)((4 * lh * K + l32) * 4);
  const size_t pkf = (size_t)2 * RB * K * 4;
  const size_t K4 = (size_t)K * 4, K20 = (size_t)K * 20;
  auto dma_run = [&](const char *src, int bytes, unsigned lds_addr) {
#pragma unroll
    for (int o = 0; o < bytes; o += 1024) {
      unsigned keep;
      if (bytes - o >= 1024) {
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(vdma), "s"(src + o), "s"(lds_addr + (unsigned)o) : "memory");
      } else {
        unsigned long long ex;
        const unsigned long long mask = (1ull << ((bytes - o) / 16)) - 1;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b64 %1, exec\n\ts_mov_b64 exec, %5\n\ts_mov_b32 m0, %4\n\ts_nop 0\n\t"
                     "global_load_lds_dwordx4 %2, %3\n\ts_mov_b64 exec, %1\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep), "=&s"(ex) : "v"(vdma), "s"(src + o), "s"(lds_addr + (unsigned)o), "s"(mask) : "memory");
      }
    }
  };
  const char *pk0 = (const char *)a.pk + (size_t)group * pkf;
  const size_t pk_step = (size_t)a.groups * pkf;
  char *out0 = (char *)a.out + ((size_t)(a.seg0 + group) * K * K + (size_t)r0 * K + cw) * 4;
  const size_t out_step = (size_t)a.groups * K * K * 4;
  // the A side of fold fi: column tile `band` of the fold's P and Q (2 OPB contiguous bytes) -> A buffer fi & 1, by wave 0
  if (wave == 0) dma_run(pk0 + (size_t)band * (2 * OPB), 2 * OPB, ldsA);
  for (int fi = 0; fi < nfm; ++fi) {
    const char *fb = pk0 + (size_t)fi * pk_step;
    char *ob = out0 + (size_t)fi * out_step;
    const float *Ap = fA + (fi & 1) * (2 * (OPB / 4)), *Aq = Ap + OPB / 4;       // P | Q at the block's rows
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
      // ---- this step's operands: requested, then everything of this wave drained ----
#pragma unroll
      for (int tt = 0; tt < TS; ++tt) {
        const int t = TS * s + tt, j = jt0 + t;
        const char *tb = fb + (size_t)j * (2 * OPB);
        dma_run(tb + (j >= band ? 0 : OPB), OPB, ldsB + (unsigned)(tt * OPB));
        if (j == band) dma_run(tb + OPB, OPB, ldsD);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (s == 0) {
        // every wave has left the previous fold (its A buffer may be overwritten) and wave 0 has drained this fold's A side
        lds_barrier();
        if (wave == 0 && fi + 1 < nfm) dma_run(fb + pk_step + (size_t)band * (2 * OPB), 2 * OPB, ldsA + (unsigned)(((fi + 1) & 1) * 2 * OPB));
      }
#pragma unroll
      for (int tt = 0; tt < TS; ++tt) {
        const int t = TS * s + tt, j = jt0 + t;
        const float *B = fB + tt * (OPB / 4);
        const float *Aside = j >= band ? Aq : Ap;
        f16v acc = g[t];
        if (j == band) {
          // the diagonal tile: elements below the diagonal from the mirrored product x[row] * (-w x)[column], first
          f16v acc2 = g[t];
#pragma unroll
          for (int kk = 0; kk < KK; ++kk)
            acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(Ap[(2 * kk + lh) * 32 + l32], fD[(2 * kk + lh) * 32 + l32], acc2, 0, 0, 0);
#pragma unroll
          for (int kk = 0; kk < KK; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Aq[(2 * kk + lh) * 32 + l32], B[(2 * kk + lh) * 32 + l32], acc, 0, 0, 0);
#pragma unroll
          for (int v = 0; v < 16; ++v) {
            const int row = (v & 3) + 8 * (v >> 2) + 4 * lh;
            acc[v] = row > l32 ? acc2[v] : acc[v];
          }
        } else {
#pragma unroll
          for (int kk = 0; kk < KK; ++kk)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Aside[(2 * kk + lh) * 32 + l32], B[(2 * kk + lh) * 32 + l32], acc, 0, 0, 0);
        }
        f16v ps;
#pragma unroll
        for (int v = 0; v < 16; ++v) ps[v] = 0.f;
        ps = __builtin_amdgcn_mfma_f32_32x32x2f32(Aq[(NP + 2 + lh) * 32 + l32], B[(NP + 2 + lh) * 32 + l32], ps, 0, 0, 0);
        typedef float f2v __attribute__((ext_vector_type(2)));
        const char *obv = ob + 128 * t;
#pragma unroll
        for (int v = 0; v < 16; v += 2) {
          const f2v val = (f2v){acc[v], acc[v + 1]} * (f2v){ps[v], ps[v + 1]};
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            asm volatile("global_store_dword %0, %1, %2 nt" : : "v"(vout), "v"(val[e]), "s"(obv) : "memory");
            obv += ((v + e) & 3) == 3 ? K20 : K4;
            asm volatile("" : "+s"(obv));
          }
        }
      }
    }
  }
}
