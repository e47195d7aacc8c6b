// Does an LDS instruction cost VALU issue time on gfx950?  (profiles/r05/issue_breakdown.md: in k_mul_base64 the shares of SIMD cycles with a VALU
// instruction and with an LDS instruction active add up to 100 % — 93.7 + 6.1 — while the table-free ladder sits at 99 % VALU.)
// One workgroup of 1,024 threads per CU (4 wavefronts per SIMD, the fixed-base kernel's shape); every wavefront runs a loop whose body is
//   V  v_mad_u64_u32 on 8 independent accumulators  and  L  LDS instructions (ds_bpermute_b32, or ds_read_b128 at lane-linear addresses)
// whose results are only consumed after the loop.  Shader cycles per loop iteration (s_memtime), slowest wavefront of the grid:
//   V only, L only, and the mix.  additive: mix = V + L  (an LDS instruction occupies the SIMD's issue like a VALU one);  overlapped: mix = max(V, L).
// build: hipcc --offload-arch=gfx950 -O3 -o lds_coissue lds_coissue.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

__device__ __forceinline__ unsigned long long memtime() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}

template <int V, int L, int KIND>      // KIND 0: ds_bpermute_b32, 1: ds_read_b128
__global__ void __launch_bounds__(1024) k_mix(int iters, unsigned seed, unsigned long long* cyc, unsigned* sink) {
  __shared__ uint4 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 1024) lds[i] = uint4{(unsigned)i, seed, 3u, 4u};
  __syncthreads();
  unsigned long long acc[8];
  unsigned a = threadIdx.x * 2654435761u + seed, b = threadIdx.x * 40503u + 977u;
  for (int c = 0; c < 8; ++c) acc[c] = a + c * 7919u;
  unsigned got[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  uint4 rd[2] = {uint4{0, 0, 0, 0}, uint4{0, 0, 0, 0}};
  const int src = (int)(((threadIdx.x * 7u + seed) & 63u) << 2);
  const uint4* row = lds + (threadIdx.x & 1023u);
  const unsigned long long t0 = memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      // an eighth of the body: V/8 multiply-adds, L/8 LDS instructions between them
#pragma unroll
      for (int v = 0; v < V / 8; ++v) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[v & 7]) : "v"(a), "v"(b) : "vcc");
#pragma unroll
      for (int l = 0; l < L / 8; ++l) {
        if (KIND == 0) asm volatile("ds_bpermute_b32 %0, %1, %2" : "=v"(got[l & 7]) : "v"(src), "v"(a));
        else asm volatile("ds_read_b128 %0, %1" : "=v"(rd[l & 1]) : "v"((unsigned)(uintptr_t)row));
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const unsigned long long t1 = memtime();
  unsigned long long s = 0;
  for (int c = 0; c < 8; ++c) s += acc[c] + got[c];
  s += rd[0].x + rd[1].y;
  if (s == 0x1234567ull) sink[0] = (unsigned)s;
  if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int V, int L, int KIND>
double run(int cus, unsigned long long* d_cyc, unsigned* d_sink, int iters) {
  const int waves = cus * 16;
  k_mix<V, L, KIND><<<cus, 1024>>>(iters, 7u, d_cyc, d_sink);      // warm
  k_mix<V, L, KIND><<<cus, 1024>>>(iters, 9u, d_cyc, d_sink);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(waves);
  CK(hipMemcpy(h.data(), d_cyc, sizeof(unsigned long long) * waves, hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  return (double)h[waves / 2] / iters;         // shader-clock ticks of s_memtime are 100 MHz-independent core cycles on gfx9: reported as is
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  const int cus = p.multiProcessorCount;
  unsigned long long* d_cyc; unsigned* d_sink;
  CK(hipMalloc(&d_cyc, sizeof(unsigned long long) * cus * 16)); CK(hipMalloc(&d_sink, 64));
  const int iters = 2000;
  // the fixed-base window: ~1,056 VALU instructions (700 of them multiply-adds), 30 ds_bpermute + 9 reads.  Scaled to a body of 1,024 V:
  printf("{\"gpu\": \"%s\", \"cus\": %d, \"waves_per_simd\": 4, \"unit\": \"s_memtime ticks per loop iteration, median wavefront\",\n", p.name, cus);
#define ROW(V_, L_, K_, NAME) printf(" \"%s\": %.1f,\n", NAME, run<V_, L_, K_>(cus, d_cyc, d_sink, iters))
  ROW(1024, 0, 0, "V=1024 mad only");
  ROW(0, 32, 0, "L=32 bpermute only");
  ROW(1024, 32, 0, "V=1024 + L=32 bpermute");
  ROW(0, 64, 0, "L=64 bpermute only");
  ROW(1024, 64, 0, "V=1024 + L=64 bpermute");
  ROW(0, 128, 0, "L=128 bpermute only");
  ROW(1024, 128, 0, "V=1024 + L=128 bpermute");
  ROW(0, 16, 1, "L=16 read_b128 only");
  ROW(1024, 16, 1, "V=1024 + L=16 read_b128");
  ROW(0, 64, 1, "L=64 read_b128 only");
  ROW(1024, 64, 1, "V=1024 + L=64 read_b128");
  printf(" \"note\": \"additive = LDS instructions take issue time from the VALU of their SIMD; max = they overlap\"}\n");
  return 0;
}
