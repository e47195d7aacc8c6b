// What v_permlane16_swap_b32 / v_permlane32_swap_b32 (gfx950) do to a wavefront: every lane starts with its lane number in A and 100 + lane in B;
// prints the row (16 lanes) each row of the four results came from.  hipcc --offload-arch=gfx950 -O3 -o permlane_probe permlane_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
  const unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  auto s = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[threadIdx.x] = r[0]; o[64 + threadIdx.x] = r[1]; o[128 + threadIdx.x] = s[0]; o[192 + threadIdx.x] = s[1];
}
int main() {
  unsigned* d; unsigned h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[4] = {"permlane16_swap result[0]", "permlane16_swap result[1]", "permlane32_swap result[0]", "permlane32_swap result[1]"};
  for (int q = 0; q < 4; ++q) {
    printf("%s:", names[q]);
    for (int row = 0; row < 4; ++row) {
      const unsigned v = h[64 * q + 16 * row];
      bool uniform = true;
      for (int l = 0; l < 16; ++l) uniform &= h[64 * q + 16 * row + l] == v + l;
      printf("  row%d <- %s.row%u%s", row, v >= 100 ? "B" : "A", (v % 100) / 16, uniform ? "" : " (lanes permuted!)");
    }
    printf("\n");
  }
  return 0;
}
