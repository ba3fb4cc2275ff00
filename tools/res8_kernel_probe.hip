// res8_apply_kernel (cvmatrix_amd/csrc/resident.hpp: eight waves per workgroup) alone on synthetic operands, next to res_apply_kernel:
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++17 -o tools/res8_kernel_probe tools/res8_kernel_probe.hip
//   tools/res8_kernel_probe [K] [folds] [NP: 16 | 8 | 32]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#define CVM_RES_PROBE 1
namespace {
struct SmallArgs;
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
#include "../cvmatrix_amd/csrc/resident.hpp"
}
template <int NP> void run(int K, int nb) {
  constexpr int RB = NP + 4;
  float *G, *out, *pk;
  hipMalloc(&G, (size_t)K * K * 4); hipMemset(G, 0, (size_t)K * K * 4);
  hipMalloc(&out, (size_t)nb * K * K * 4);
  hipMalloc(&pk, (size_t)nb * 2 * RB * K * 4); hipMemset(pk, 0, (size_t)nb * 2 * RB * K * 4);
  ResArgs r; memset(&r, 0, sizeof(r));
  const int nblk_all = (K / 32) * (K / RES_BC);
  int groups = nblk_all >= RES_WG ? 1 : RES_WG / nblk_all;
  if (groups > nb / 4) groups = nb / 4;
  r.G = G; r.out = out; r.pk = pk; r.K = K; r.nb = nb; r.seg0 = 0; r.nbc = K / RES_BC; r.groups = groups;
  constexpr int lds4 = 4 * 7 * RB * 128, lds8 = res8_lds<NP>();
  if (NP <= 16) hipFuncSetAttribute((const void *)res_apply_kernel<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds4);
  hipFuncSetAttribute((const void *)res8_apply_kernel<NP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds8);
  for (int which = (NP > 16 ? 1 : 0); which < 2; ++which) {
    auto launch = [&] {
      for (int b0 = 0; b0 < nblk_all; b0 += RES_WG) {
        r.blk0 = b0; r.nblk = nblk_all - b0 < RES_WG ? nblk_all - b0 : RES_WG;
        const unsigned wgs = (unsigned)(8 * (((size_t)r.nblk * groups + 7) / 8));
        if (which == 0) { if constexpr (NP <= 16) hipLaunchKernelGGL((res_apply_kernel<NP>), dim3(wgs), dim3(256), lds4, 0, r); }
        else hipLaunchKernelGGL((res8_apply_kernel<NP>), dim3(wgs), dim3(512), lds8, 0, r);
      }
    };
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) launch();
    for (int rep3 = 0; rep3 < 2; ++rep3) {
      hipEventRecord(a);
      const int rep = 10;
      for (int i = 0; i < rep; ++i) launch();
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); ms /= rep;
      printf("%s NP %2d  K = %d, %d folds: %7.3f ms  %5.2f TB/s of outputs\n", which ? "eight waves" : "four waves ", NP, K, nb, ms,
             (double)nb * K * K * 4 / ms / 1e9);
    }
  }
  hipFree(G); hipFree(out); hipFree(pk);
}
int main(int argc, char **argv) {
  const int K = argc > 1 ? atoi(argv[1]) : 4096, nb = argc > 2 ? atoi(argv[2]) : 48, np = argc > 3 ? atoi(argv[3]) : 16;
  if (np == 8) run<8>(K, nb); else if (np == 32) run<32>(K, nb); else run<16>(K, nb);
  return 0;
}
