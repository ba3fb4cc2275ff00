// Microbenchmark: in what order, and where, does the MI355X hand workgroups of ONE launch to CUs
// when a workgroup fills a CU (148 KB of LDS, 512 threads) and workgroups differ in duration?
// Every workgroup records its start / end time (100 MHz realtime counter) and where it ran
// (XCC_ID, SE, CU from the hardware id registers), then idles for its programmed time.
//   hipcc --offload-arch=gfx950 -O2 -o tools/dispatch_probe tools/dispatch_probe.hip
//   tools/dispatch_probe <pattern>   pattern: "N0:T0,N1:T1,..." = N0 workgroups of T0 us, then N1 of T1 us ...
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Rec { unsigned long long t0, t1; unsigned hw, xcc; };

__global__ __launch_bounds__(512, 2) void probe(const unsigned *dur_ticks, Rec *out) {
  extern __shared__ char smem[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  const unsigned long long until = t0 + dur_ticks[blockIdx.x];
  while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0) {
    smem[0] = 1;
    Rec r; r.t0 = t0; r.t1 = __builtin_amdgcn_s_memrealtime(); r.hw = hw; r.xcc = xcc;
    out[blockIdx.x] = r;
  }
}

int main(int argc, char **argv) {
  const char *pat = argc > 1 ? argv[1] : "240:280,280:120";
  std::vector<unsigned> dur;
  {
    char *s = strdup(pat);
    for (char *tok = strtok(s, ","); tok; tok = strtok(nullptr, ",")) {
      int n = 0; double us = 0;
      if (sscanf(tok, "%d:%lf", &n, &us) != 2) { printf("bad pattern\n"); return 1; }
      for (int i = 0; i < n; ++i) dur.push_back((unsigned)(us * 100.0));
    }
    free(s);
  }
  const int nb = (int)dur.size();
  unsigned *d_dur; Rec *d_out;
  CK(hipMalloc(&d_dur, nb * 4)); CK(hipMalloc(&d_out, nb * sizeof(Rec)));
  CK(hipMemcpy(d_dur, dur.data(), nb * 4, hipMemcpyHostToDevice));
  const size_t lds = 148 * 1024;
  CK(hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  std::vector<Rec> rec(nb);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe, dim3(nb), dim3(512), lds, 0, d_dur, d_out);
    CK(hipDeviceSynchronize());
  }
  CK(hipMemcpy(rec.data(), d_out, nb * sizeof(Rec), hipMemcpyDeviceToHost));
  unsigned long long tmin = ~0ull, tmax = 0;
  for (auto &r : rec) { tmin = std::min(tmin, r.t0); tmax = std::max(tmax, r.t1); }
  printf("# pattern %s: %d workgroups, makespan %.1f us\n", pat, nb, (tmax - tmin) / 100.0);
  printf("# block start_us end_us xcc se sh cu\n");
  for (int b = 0; b < nb; ++b) {
    const Rec &r = rec[b];
    // gfx9 HW_ID: CU_ID 11:8, SH_ID 12, SE_ID 15:13
    printf("%d %.2f %.2f %u %u %u %u\n", b, (r.t0 - tmin) / 100.0, (r.t1 - tmin) / 100.0, r.xcc & 0xf,
           (r.hw >> 13) & 7, (r.hw >> 12) & 1, (r.hw >> 8) & 0xf);
  }
  return 0;
}
