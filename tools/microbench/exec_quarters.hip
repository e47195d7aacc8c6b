// Does a VALU instruction of a LONE wavefront cost less when only some 16-lane quarters of EXEC are on?  One wavefront on an idle chip walks a
// dependent chain (v_mad_u64_u32, v_and_b32, v_add_u32 in turn) with 64 / 32 / 16 / 1 lanes active; cycles by s_memtime around the chain.
// hipcc --offload-arch=gfx950 -O3 -o exec_quarters exec_quarters.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int KIND>
__global__ void __launch_bounds__(64) k(unsigned long long* o, unsigned lanes, int n, unsigned seed) {
  unsigned long long acc = seed + threadIdx.x;
  unsigned x = seed | 1u;
  unsigned long long acc2 = seed * 3u + threadIdx.x;
  unsigned long long t0 = 0, t1 = 0;
  if (threadIdx.x < lanes) {
    t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int i = 0; i < n; ++i) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (KIND == 0) { acc = (unsigned long long)(unsigned)acc * x + acc; }
        else if (KIND == 1) { unsigned a = (unsigned)acc; a = (a & 0x3ffffffu) + x; a ^= a >> 3; acc = a; }
        else if (KIND == 3) { acc = (unsigned long long)(unsigned)acc * x + acc; acc2 = (unsigned long long)(unsigned)acc2 * x + acc2; }
        else if (KIND == 4) { asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %1, %0\n" : "+v"(acc) : "v"(x) : "s10", "s11"); }
        else if (KIND == 5) { asm volatile("v_mad_u64_u32 %0, s[10:11], %2, %2, %0\nv_mad_u64_u32 %1, s[10:11], %2, %2, %1\n" : "+v"(acc), "+v"(acc2) : "v"(x) : "s10", "s11"); }
        else if (KIND == 6) { asm volatile("v_mad_u64_u32 %0, s[10:11], %2, %2, %0\nv_and_b32 %3, %2, %3\nv_mad_u64_u32 %1, s[10:11], %2, %2, %1\nv_and_b32 %3, %2, %3\n" : "+v"(acc), "+v"(acc2), "+v"(x) : "v"(seed) : "s10", "s11"); }
        else { asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0\n" : "+v"(acc) : "v"((unsigned)acc), "v"(x) : "s10", "s11"); unsigned a = (unsigned)acc & 0x3ffffffu; acc = a + x; }
      }
    }
    t1 = __builtin_readcyclecounter();
  }
  if (threadIdx.x == 0) { o[0] = t1 - t0; o[1] = acc + acc2; }
}
template <int KIND>
static void run(const char* what, int per_iter) {
  unsigned long long* d; unsigned long long h[2];
  hipMalloc(&d, 16);
  const int n = 4000;
  const unsigned lanes[] = {64, 48, 32, 16, 1};
  for (unsigned l : lanes) {
    unsigned long long best = ~0ull;
    for (int rep = 0; rep < 5; ++rep) {
      hipLaunchKernelGGL(k<KIND>, dim3(1), dim3(64), 0, 0, d, l, n, 12345u);
      hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
      if (h[0] < best) best = h[0];
    }
    printf("%-42s lanes %2u: %8.2f shader-clock ticks per instruction\n", what, l, (double)best / ((double)n * 16 * per_iter));
  }
  hipFree(d);
}
int main() {
  run<0>("dependent v_mad_u64_u32", 1);
  run<1>("dependent v_and / v_add / v_lshr / v_xor", 4);
  run<2>("v_mad_u64_u32, v_and, v_add (dependent)", 3);
  run<3>("two independent chains of v_mad_u64_u32 (multiplicand = own accumulator)", 2);
  run<4>("v_mad_u64_u32 accumulating only (acc += x * x)", 1);
  run<5>("two accumulate-only chains", 2);
  run<6>("two accumulate-only chains with an independent v_and between the mads", 4);
  return 0;
}
