// Can one 1024-thread workgroup own (nearly) the whole 160 KiB LDS of a gfx950 CU?  (fixed-base radix-64 table: 163,200 B)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int BYTES>
__global__ void __launch_bounds__(1024, 4) k(uint32_t* out) {
  __shared__ uint4 t[BYTES / 16];
  for (int i = threadIdx.x; i < BYTES / 16; i += 1024) t[i] = make_uint4(i, i + 1, i + 2, i + 3);
  __syncthreads();
  uint32_t acc = 0;
  for (int i = threadIdx.x; i < BYTES / 16; i += 97) acc += t[(i * 31) % (BYTES / 16)].y;
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
}
int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  printf("sharedMemPerBlock %zu maxSharedMemoryPerMultiProcessor %zu regsPerBlock %d\n", p.sharedMemPerBlock, p.maxSharedMemoryPerMultiProcessor, p.regsPerBlock);
  uint32_t* d;
  hipMalloc(&d, 256 * 1024 * 4);
  hipLaunchKernelGGL((k<163200>), dim3(256), dim3(1024), 0, 0, d);
  hipError_t e = hipDeviceSynchronize();
  printf("163200 B: launch %s / %s\n", hipGetErrorString(hipGetLastError()), hipGetErrorString(e));
  hipLaunchKernelGGL((k<163840>), dim3(256), dim3(1024), 0, 0, d);
  e = hipDeviceSynchronize();
  printf("163840 B: launch %s / %s\n", hipGetErrorString(hipGetLastError()), hipGetErrorString(e));
  return 0;
}
