// Which compute units does a stream created with hipExtStreamCreateWithCUMask really get on this 8-XCD part?  Every wavefront of a large grid
// records (XCC_ID, SE_ID, CU_ID) from the hardware registers; the host counts the distinct compute units per mask.
// build: hipcc --offload-arch=gfx950 -O3 -o cu_mask_map cu_mask_map.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <set>
#include <map>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 1; } } while (0)
__global__ void k_where(uint32_t* out, int spin) {
  uint32_t hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  uint32_t acc = threadIdx.x;
  for (int i = 0; i < spin; ++i) acc = acc * 1664525u + 1013904223u;      // keep the wavefront resident long enough for the grid to spread
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = (xcc & 0xfu) | (acc == 7u ? 0x80000000u : 0u); }
}
int main() {
  const int blocks = 8192;
  uint32_t* d; CK(hipMalloc(&d, 8 * blocks));
  struct M { int words; uint32_t w[8]; const char* name; } masks[] = {
    {8, {~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u, ~0u}, "all 256 bits"},
    {2, {~0u, ~0u}, "bits 0..63"},
    {1, {~0u}, "bits 0..31"},
    {1, {0xffu}, "bits 0..7"},
    {1, {0x1u}, "bit 0"},
    {8, {0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u, 0x11111111u}, "every 4th bit"},
    {8, {0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u, 0x01010101u}, "every 8th bit"},
    {8, {0xffu, 0xffu, 0xffu, 0xffu, 0xffu, 0xffu, 0xffu, 0xffu}, "low byte of every word"},
    {8, {~0u, 0, 0, 0, 0, 0, 0, 0}, "word 0 of 8"},
    {8, {0, 0, 0, 0, 0, 0, 0, ~0u}, "word 7 of 8"}};
  for (auto& m : masks) {
    hipStream_t st;
    CK(hipExtStreamCreateWithCUMask(&st, (uint32_t)m.words, m.w));
    CK(hipMemsetAsync(d, 0, 8 * blocks, st));
    hipLaunchKernelGGL(k_where, dim3(blocks), dim3(64), 0, st, d, 20000);
    CK(hipStreamSynchronize(st));
    std::vector<uint32_t> h(2 * blocks);
    CK(hipMemcpy(h.data(), d, 8 * blocks, hipMemcpyDeviceToHost));
    std::set<uint32_t> cus; std::map<uint32_t, std::set<uint32_t>> per_xcc;
    for (int b = 0; b < blocks; ++b) {
      const uint32_t hw = h[2 * b], xcc = h[2 * b + 1] & 0xf;
      const uint32_t cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;      // HW_ID: [11:8] CU_ID, [12] SH_ID, [15:13] SE_ID
      const uint32_t id = (xcc << 12) | (se << 8) | (sh << 4) | cu;
      cus.insert(id); per_xcc[xcc].insert(id);
    }
    printf("%-24s -> %3zu compute units on %zu XCDs (", m.name, cus.size(), per_xcc.size());
    for (auto& kv : per_xcc) printf("%u:%zu ", kv.first, kv.second.size());
    printf(")\n");
    CK(hipStreamDestroy(st));
  }
  return 0;
}
