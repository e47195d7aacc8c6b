#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int CTRL, bool BC> __device__ __forceinline__ uint32_t dpp(uint32_t old, uint32_t v) {
  return (uint32_t)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, 0xf, 0xf, BC);
}
__global__ void k(uint32_t* out) {
  const uint32_t lane = threadIdx.x;
  const uint32_t v = 1000 + lane;
  out[0 * 64 + lane] = dpp<0x150 + 3, true>(0, v);          // row_newbcast:3
  out[1 * 64 + lane] = dpp<0x110 + 1, false>(0xdead, v);    // row_shr:1, keep old on invalid
  out[2 * 64 + lane] = dpp<0x100 + 9, true>(0, v);          // row_shl:9, zero on invalid
  out[3 * 64 + lane] = dpp<0x110 + 3, false>(dpp<0x100 + 7, true>(0, v), v);   // rotate by 3 mod 10
  out[4 * 64 + lane] = lane + dpp<0x110 + 1, true>(0, v);   // should fold to v_add_u32_dpp
}
int main() {
  uint32_t* d; hipMalloc(&d, 5 * 64 * 4);
  k<<<1, 64>>>(d);
  uint32_t h[5 * 64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int r = 0; r < 5; ++r) { for (int l = 0; l < 32; ++l) printf("%5u ", h[r * 64 + l]); printf("\n"); }
  return 0;
}
