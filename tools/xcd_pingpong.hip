// Microbenchmark: round-trip latency of a flag ping-pong between two workgroups through global memory,
// on the SAME XCD vs on DIFFERENT XCDs, with the access flavours the PLS kernel's per-fold barrier could
// use: agent-scope relaxed atomics (sc1: what it uses now), workgroup-scope atomics (sc0) and agent-scope
// read-modify-write atomics (performed in L2).  16 workgroups of 64 threads; block b runs on XCD b % 8
// (tools/dispatch_probe.hip), so (b, b + 8) share an XCD and (b, b + 1) do not.
//   hipcc --offload-arch=gfx950 -O3 -o xcd_pingpong tools/xcd_pingpong.hip && ./xcd_pingpong
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE> __device__ __forceinline__ void put(unsigned *p, unsigned v) {
  if (MODE == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if (MODE == 1) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <int MODE> __device__ __forceinline__ unsigned get(unsigned *p) {
  if (MODE == 0) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  else if (MODE == 1) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  else return __hip_atomic_fetch_add(p, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// pair p: blocks a and b; flags[2 p] written by a, flags[2 p + 1] written by b (128 bytes apart)
template <int MODE>
__global__ void k(unsigned *flags, const int *pa, const int *pb, int npairs, int rounds, unsigned long long *out, unsigned *xcc) {
  if (threadIdx.x != 0) return;
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  xcc[blockIdx.x] = x & 7u;
  for (int p = 0; p < npairs; ++p) {
    const bool first = (int)blockIdx.x == pa[p], second = (int)blockIdx.x == pb[p];
    if (!first && !second) continue;
    unsigned *mine = flags + 64 * (2 * p + (second ? 1 : 0)), *theirs = flags + 64 * (2 * p + (second ? 0 : 1));
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool dead = false;                 // the partner's stores never became visible (this flavour is not coherent here)
    for (int r = 1; r <= rounds && !dead; ++r) {
      if (first) put<MODE>(mine, (unsigned)r);
      long spins = 0;
      while (get<MODE>(theirs) < (unsigned)r) { if (++spins > 50000) { dead = true; break; } }
      if (!first) put<MODE>(mine, dead ? 0xffffffffu : (unsigned)r);
    }
    if (dead) put<MODE>(mine, 0xffffffffu);          // releases the partner
    const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
    if (first) out[p] = dead ? 0ull : t1 - t0;      // 100 MHz ticks; 0: not coherent
  }
}

template <int MODE> int run(const char *what, unsigned *flags, int *pa, int *pb, unsigned long long *out, unsigned *xcc) {
  const int rounds = 2000;
  // pairs: same XCD (b, b + 8) for b = 0..3 -- run one at a time is not needed: disjoint blocks
  int ha[8] = {0, 1, 2, 3, 4, 6, 12, 14}, hb[8] = {8, 9, 10, 11, 5, 7, 13, 15};
  CK(hipMemcpy(pa, ha, sizeof(ha), hipMemcpyHostToDevice));
  CK(hipMemcpy(pb, hb, sizeof(hb), hipMemcpyHostToDevice));
  CK(hipMemset(flags, 0, 64 * 16 * sizeof(unsigned)));
  hipLaunchKernelGGL((k<MODE>), dim3(16), dim3(64), 0, 0, flags, pa, pb, 8, rounds, out, xcc);
  CK(hipDeviceSynchronize());
  unsigned long long h[8]; unsigned hx[16];
  CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
  CK(hipMemcpy(hx, xcc, sizeof(hx), hipMemcpyDeviceToHost));
  printf("%s\n", what); fflush(stdout);
  for (int p = 0; p < 8; ++p)
    printf("  blocks %2d (XCD %u) <-> %2d (XCD %u): %7.1f ns per round trip\n", ha[p], hx[ha[p]], hb[p], hx[hb[p]],
           (double)h[p] * 10.0 / rounds);
  fflush(stdout);
  return 0;
}

int main() {
  unsigned *flags, *xcc; int *pa, *pb; unsigned long long *out;
  CK(hipMalloc(&flags, 64 * 16 * sizeof(unsigned))); CK(hipMalloc(&xcc, 16 * sizeof(unsigned)));
  CK(hipMalloc(&pa, 8 * sizeof(int))); CK(hipMalloc(&pb, 8 * sizeof(int))); CK(hipMalloc(&out, 8 * sizeof(unsigned long long)));
  if (run<0>("agent-scope relaxed atomic store / load (sc1)", flags, pa, pb, out, xcc)) return 1;
  if (run<1>("workgroup-scope relaxed atomic store / load (sc0)", flags, pa, pb, out, xcc)) return 1;
  if (run<2>("agent-scope atomic exchange / fetch_add 0 (read-modify-write in L2)", flags, pa, pb, out, xcc)) return 1;
  return 0;
}
