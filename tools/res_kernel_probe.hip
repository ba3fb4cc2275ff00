// res_apply_kernel (cvmatrix_amd/csrc/resident.hpp) alone on synthetic operands, with ablations:
//   for a in 0 1 2 4 8 3 ...; do hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -DCVM_RES_ABLATE=$a -o tools/res_kernel_probe_$a tools/res_kernel_probe.hip; done
// (1 no MFMA chain, 2 no LDS-DMA, 4 no stores, 8 no scaling multiply; -DCVM_RES_SAFE=1: every wait vmcnt(0)).  K = 4096, 48 folds.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#define CVM_RES_PROBE 1
namespace {
struct SmallArgs;
#include "../cvmatrix_amd/csrc/resident.hpp"
}
int main(int argc, char **argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 4096, nb = argc > 2 ? atoi(argv[2]) : 48;
  constexpr int NP = 16, RB = NP + 4;
  float *G, *out, *pk;
  hipMalloc(&G, (size_t)K * K * 4); hipMemset(G, 0, (size_t)K * K * 4);
  hipMalloc(&out, (size_t)nb * K * K * 4);
  hipMalloc(&pk, (size_t)nb * 2 * RB * K * 4); hipMemset(pk, 0, (size_t)nb * 2 * RB * K * 4);
  ResArgs r; memset(&r, 0, sizeof(r));
  const int nblk_all = (K / 32) * (K / RES_BC);
  int groups = nblk_all >= RES_WG ? 1 : RES_WG / nblk_all;
  if (groups > nb / 4) groups = nb / 4;
  r.G = G; r.out = out; r.pk = pk; r.K = K; r.nb = nb; r.seg0 = 0; r.nbc = K / RES_BC; r.groups = groups;
  constexpr int lds = 4 * 7 * RB * 128;
  hipFuncSetAttribute((const void *)res_apply_kernel<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  auto launch = [&] {
    for (int b0 = 0; b0 < nblk_all; b0 += RES_WG) {
      r.blk0 = b0; r.nblk = nblk_all - b0 < RES_WG ? nblk_all - b0 : RES_WG;
      const unsigned wgs = (unsigned)(8 * (((size_t)r.nblk * groups + 7) / 8));
      hipLaunchKernelGGL((res_apply_kernel<NP>), dim3(wgs), dim3(256), lds, 0, r);
    }
  };
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) launch();
  hipEventRecord(a);
  const int rep = 10;
  for (int i = 0; i < rep; ++i) launch();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= rep;
  printf("ablate %2d safe %d  K = %d, %d folds, %d groups: %7.3f ms  %5.2f TB/s of outputs\n", CVM_RES_ABLATE, CVM_RES_SAFE, K, nb, groups, ms,
         (double)nb * K * K * 4 / ms / 1e9);
  return 0;
}
