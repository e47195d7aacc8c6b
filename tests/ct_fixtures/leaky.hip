// Fixture for tests/test_ct_check.py: four kernels over a "secret" array (kernel argument 0), three of which leak it in one of the ways
// tools/ct_check.py is supposed to catch, and one that does the same work in constant time.  Never built into the library.
#include <hip/hip_runtime.h>
#include <stdint.h>

// (1) a branch on a secret bit: square-and-multiply the naive way
extern "C" __global__ void leak_branch(const uint32_t* __restrict__ secret, uint32_t* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t k = secret[i];
  uint32_t acc = 1, base = 3;
  for (int b = 0; b < 32; ++b) {
    if ((k >> b) & 1u) acc *= base;          // data-dependent branch (or, at best, a data-dependent EXEC mask around a block)
    base *= base;
    asm volatile("" : "+v"(acc), "+v"(base));
    if (((k >> b) & 1u) && (acc & 1u) == 0u) out[n + (b & 3)] = acc;      // ... with a store inside
  }
  out[i] = acc;
}

// (2) a table lookup at a secret index
extern "C" __global__ void leak_address(const uint32_t* __restrict__ secret, const uint32_t* __restrict__ table, uint32_t* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = table[secret[i] & 255u];
}

// (3) a secret-indexed LDS read
extern "C" __global__ void leak_lds(const uint32_t* __restrict__ secret, const uint32_t* __restrict__ table, uint32_t* __restrict__ out, int n) {
  __shared__ uint32_t t[256];
  t[threadIdx.x & 255] = table[threadIdx.x & 255];
  __syncthreads();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  out[i] = t[secret[i] & 255u];
}

// (4) the same lookup in constant time: a scan over public addresses with mask selects, and the selection moved by ds_bpermute
extern "C" __global__ void clean_scan(const uint32_t* __restrict__ secret, const uint32_t* __restrict__ table, uint32_t* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t k = secret[i] & 63u;
  uint32_t v = 0;
  for (uint32_t j = 0; j < 64; ++j) {
    const uint32_t m = 0u - (uint32_t)(j == k);
    v |= table[j] & m;
  }
  const uint32_t mine = table[64 + (threadIdx.x & 63)];
  v ^= (uint32_t)__builtin_amdgcn_ds_bpermute((int)(k << 2), (int)mine);      // lane select by a secret: the exempt primitive
  out[i] = v;
}
