// Nine limbs or ten?  (DESIGN.md "Why the step stays near 1,200".)
//
// The field arithmetic of this engine is ref10's radix 2^25.5: ten limbs, 100 v_mad_u64_u32 per product, 55 per square, the x19 of the
// wrap-around premultiplied into a 32-bit operand.  Nine limbs of 29 / 28 / 28 bits (positions ceil(85 i / 3): 0, 29, 57, 85, 114, 142, 170,
// 199, 227) need only 81 / 45 multiply-adds — but 19 x 2^29 no longer fits 32 bits, so the upper eight columns are carried down to limbs
// first, folded into the lower nine with eight more multiply-adds, and carried again; and a column holds up to nine terms of 2^58 or
// 2^59 (the factor 2 of the positions that do not add up: (i mod 3, j mod 3) in {(1,1), (1,2), (2,1)}), i.e. operands must be TIGHT.
// This program runs both as dependent chains (a = a * b, b = b * a: the shape of the ladder's data flow) at 3 wavefronts per SIMD,
// checks the nine-limb results against the ten-limb ones through the 32-byte encoding, and prints the time per product / square.
// A third variant splits each ten-term column of the ten-limb product into two five-term chains joined by one 64-bit addition.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I kyber-rs_amd/csrc -o tools/microbench/fe9 tools/microbench/fe9.hip && tools/microbench/fe9
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include "fe25519.h"
using namespace kyb;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ---- nine limbs --------------------------------------------------------------------------------------------------------------------
struct fe9 { uint32_t v[9]; };
__device__ constexpr int pos9(int i) { return (85 * i + 2) / 3; }
__device__ constexpr int bits9(int i) { return pos9(i + 1) - pos9(i); }                      // 29, 28, 28, ...
__device__ constexpr int delta9(int i, int j) { return pos9(i) + pos9(j) - pos9((i + j) % 9) - ((i + j) >= 9 ? 255 : 0); }

__device__ __forceinline__ uint64_t mad(uint32_t a, uint32_t b, uint64_t c) {
  asm("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b) : "vcc");
  return c;
}
// 8 words (little-endian, < 2^255) -> nine tight limbs and back (the value may stay >= p: weakly reduced)
__device__ void fe9_from_words(fe9& h, const uint32_t w[8]) {
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int p = pos9(i), wd = p >> 5, sh = p & 31, nb = bits9(i);
    uint64_t x = (uint64_t)w[wd] | (wd + 1 < 8 ? (uint64_t)w[wd + 1] << 32 : 0ull);
    h.v[i] = (uint32_t)(x >> sh) & ((1u << nb) - 1u);
  }
}
__device__ void fe9_to_words(uint32_t w[8], const fe9& f) {          // limbs may exceed their width by a bit: added, not OR-ed
  uint64_t acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0;
#pragma unroll
  for (int i = 0; i < 9; ++i) {
    const int p = pos9(i), wd = p >> 5, sh = p & 31;
    const uint64_t x = (uint64_t)f.v[i] << sh;
    acc[wd] += (uint32_t)x;
    acc[wd + 1] += x >> 32;
  }
  uint64_t c = 0;
#pragma unroll
  for (int k = 0; k < 8; ++k) { c += acc[k]; w[k] = (uint32_t)c; c >>= 32; }
  // anything at or above 2^255 (never more than a few units) folds back times 19
  uint64_t top = ((c + acc[8]) << 1) | (w[7] >> 31);
  w[7] &= 0x7fffffffu;
  c = top * 19u;
#pragma unroll
  for (int k = 0; k < 8; ++k) { c += w[k]; w[k] = (uint32_t)c; c >>= 32; }
}
// columns 0..16 of f * g (SQ: g = f, symmetric terms once), the 2^delta factors on doubled copies of f
template <bool SQ>
__device__ __forceinline__ void fe9_mul_t(fe9& h, const fe9& f, const fe9& g) {
  uint32_t f2[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) f2[i] = f.v[i] + f.v[i];
  uint32_t r[17];
  uint64_t acc = 0;
#pragma unroll
  for (int k = 0; k < 17; ++k) {
#pragma unroll
    for (int i = 0; i < 9; ++i) {
      const int j = k - i;
      if (j < 0 || j > 8) continue;
      if (SQ && i > j) continue;
      // factor: 2^delta of the positions, and 2 for a cross term of the square
      const int d = pos9(i) + pos9(j) - (k < 9 ? pos9(k) : pos9(k - 9) + 255);
      const int fac = (d ? 2 : 1) * ((SQ && i != j) ? 2 : 1);                               // 1, 2 or 4
      const uint32_t a = fac == 1 ? f.v[i] : f2[i];
      const uint32_t b = fac == 4 ? (SQ ? f2[j] : g.v[j]) : g.v[j];
      acc = mad(a, b, acc);
    }
    const int nb = bits9(k % 9);
    r[k] = (uint32_t)acc & ((1u << nb) - 1u);
    acc >>= nb;
  }
  // columns 9..16 (and the carry out of column 16, as a 10th high limb) times 19 into columns 0..8, then the second carry chain
  uint64_t c = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    c += r[k];
    if (k < 8) c = mad(r[9 + k], 19u, c);
    else c += (uint64_t)((uint32_t)acc) * 19u + (((acc >> 32) * 19u) << 32);               // the carry out of column 16 belongs to position 255 + pos9(8)
    const int nb = bits9(k);
    h.v[k] = (uint32_t)c & ((1u << nb) - 1u);
    c >>= nb;
  }
  // carry out of the top limb: x 19 back into limb 0 (and one more step: limb 0 may exceed 29 bits by one)
  uint32_t t0 = h.v[0] + (uint32_t)c * 19u;
  h.v[0] = t0 & ((1u << 29) - 1u);
  h.v[1] += t0 >> 29;
}

// ---- ten limbs, every column as two chains of five joined by one 64-bit addition ----------------------------------------------------
__device__ __forceinline__ void fe_mul_split(fe& h, const fe& f, const fe& g) {
  uint32_t g19[10], f2[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) { g19[i] = g.v[i] * 19u; f2[i] = f.v[i] + f.v[i]; }
  uint64_t acc = 0;
  uint32_t r[10];
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    uint64_t lo = acc, hi = 0;
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const int j = (k - i + 10) % 10;
      const uint32_t a = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
      const uint32_t b = (i > k) ? g19[j] : g.v[j];
      if (i < 5) lo = mad(a, b, lo); else hi = mad(a, b, hi);
    }
    acc = lo + hi;
    r[k] = (uint32_t)acc & KYB_MASK(k);
    acc >>= KYB_BITS(k);
  }
  fe_fold<true>(h, r, acc);
}

// ---- kernels: dependent chains, 2 operations per iteration -----------------------------------------------------------------------------
enum { V_MUL10, V_SQ10, V_MUL9, V_SQ9, V_MUL10_SPLIT };
template <int V>
__global__ void __launch_bounds__(256, 3) k_chain(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int iters) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t wa[8], wb[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { wa[k] = in[16 * i + k]; wb[k] = in[16 * i + 8 + k]; }
  wa[7] &= 0x7fffffffu; wb[7] &= 0x7fffffffu;
  if (V == V_MUL9 || V == V_SQ9) {
    fe9 a, b;
    fe9_from_words(a, wa); fe9_from_words(b, wb);
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
      if (V == V_MUL9) { fe9_mul_t<false>(a, a, b); fe9_mul_t<false>(b, b, a); }
      else { fe9_mul_t<true>(a, a, a); fe9_mul_t<true>(b, b, b); }
    }
    fe9_to_words(wa, a); fe9_to_words(wb, b);
    // canonical bytes through the ten-limb code (from_words accepts values up to 2^255 - 1, to_words reduces)
    fe ta, tb;
    fe_from_words(ta, wa); fe_from_words(tb, wb);
    fe_to_words(wa, ta); fe_to_words(wb, tb);
  } else {
    fe a, b;
    fe_from_words(a, wa); fe_from_words(b, wb);
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
      if (V == V_MUL10) { fe_mul_b6(a, a, b); fe_mul_b6(b, b, a); }
      else if (V == V_MUL10_SPLIT) { fe_mul_split(a, a, b); fe_mul_split(b, b, a); }
      else { fe_sq_b2(a, a); fe_sq_b2(b, b); }
    }
    fe_to_words(wa, a); fe_to_words(wb, b);
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) { out[16 * i + k] = wa[k]; out[16 * i + 8 + k] = wb[k]; }
}

template <int V>
static double run(const uint32_t* d_in, uint32_t* d_out, int blocks, int iters, std::vector<uint32_t>* host = nullptr) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_chain<V>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (rep && ms < best) best = ms;
  }
  if (host) CK(hipMemcpy(host->data(), d_out, host->size() * 4, hipMemcpyDeviceToHost));
  return best;
}

int main() {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, blocks = cus * 3;                               // 3 workgroups of 4 wavefronts per CU = 3 per SIMD
  const size_t lanes = (size_t)blocks * 256;
  std::vector<uint32_t> in(lanes * 16), o10(lanes * 16), o9(lanes * 16), os(lanes * 16);
  srand(7);
  for (auto& w : in) w = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
  uint32_t *d_in, *d_out;
  CK(hipMalloc(&d_in, in.size() * 4)); CK(hipMalloc(&d_out, in.size() * 4));
  CK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice));
  // correctness first, on short chains: nine limbs == ten limbs == split chains
  for (int iters : {1, 7}) {
    run<V_MUL10>(d_in, d_out, blocks, iters, &o10); run<V_MUL9>(d_in, d_out, blocks, iters, &o9); run<V_MUL10_SPLIT>(d_in, d_out, blocks, iters, &os);
    size_t bad9 = 0, bads = 0;
    for (size_t i = 0; i < o10.size(); ++i) { bad9 += o10[i] != o9[i]; bads += o10[i] != os[i]; }
    printf("check, %d iterations of (a = a b, b = b a): nine-limb words differing from ten-limb: %zu of %zu; split-chain: %zu\n", iters, bad9, o10.size(), bads);
    run<V_SQ10>(d_in, d_out, blocks, iters, &o10); run<V_SQ9>(d_in, d_out, blocks, iters, &o9);
    bad9 = 0;
    for (size_t i = 0; i < o10.size(); ++i) bad9 += o10[i] != o9[i];
    printf("check, %d iterations of (a = a^2, b = b^2): nine-limb words differing from ten-limb: %zu of %zu\n", iters, bad9, o10.size());
  }
  const int iters = 2000;
  const double ops = 2.0 * iters;
  const double m10 = run<V_MUL10>(d_in, d_out, blocks, iters), s10 = run<V_SQ10>(d_in, d_out, blocks, iters);
  const double m9 = run<V_MUL9>(d_in, d_out, blocks, iters), s9 = run<V_SQ9>(d_in, d_out, blocks, iters);
  const double ms = run<V_MUL10_SPLIT>(d_in, d_out, blocks, iters);
  printf("%s, %d CUs, 3 wavefronts per SIMD, %d dependent operations per lane; ns per operation of a lane (kernel time / operations):\n", prop.name, cus, (int)ops);
  printf("  product   ten limbs (fe_mul_b6, the library's)   %8.2f\n", m10 * 1e6 / ops);
  printf("  product   ten limbs, columns as 2 x 5 chains       %8.2f   (x%.3f)\n", ms * 1e6 / ops, ms / m10);
  printf("  product   nine limbs 29/28/28                      %8.2f   (x%.3f)\n", m9 * 1e6 / ops, m9 / m10);
  printf("  square    ten limbs (fe_sq_b2, the library's)     %8.2f\n", s10 * 1e6 / ops);
  printf("  square    nine limbs 29/28/28                      %8.2f   (x%.3f)\n", s9 * 1e6 / ops, s9 / s10);
  printf("  ladder step = 5 products + 4 squares: ten limbs %.2f, nine limbs %.2f (x%.3f) — before the extra carry passes nine tight-only limbs need after every addition\n",
         (5 * m10 + 4 * s10) * 1e6 / ops, (5 * m9 + 4 * s9) * 1e6 / ops, (5 * m9 + 4 * s9) / (5 * m10 + 4 * s10));
  return 0;
}
