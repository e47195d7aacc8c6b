// Linear combinations over SHARED points with PUBLIC scalars (one of the translation units mapped in launch.h):
//   out[g] = sum_{j < t} s[g][j] * P[j],   g < m                                     recover_pub_poly, share/poly.rs:607-634
// The reference performs the m * t multiplications one by one with its 64-window routine; kyb_lincomb_batch runs them as m * t
// constant-time ladders.  When the caller DECLARES the scalars public (kyb_lincomb_public_batch: Lagrange basis coefficients of
// public share indices) the t points are worth a table each, shared by all m outputs:
//   k_msm_bases_coop  (kernels_coop.hip)  64^w P_j, w = 0 .. 42: one wavefront per point, 252 cooperative doublings
//   k_msm_tables      T[j][w][e] = (e + 1) * 64^w * P_j, e = 0 .. 31, as cached elements (Y+X, Y-X, Z, 2dT): one lane per (j, w), 31 additions
//   k_msm_accumulate  one lane per (output g, chunk of points): signed radix-64 digits d in -32 .. 31 of the integer the reference
//                     multiplies by (sc_effective: the scalar as stored, top-digit quirk included), one table addition per non-zero
//                     digit — 43 additions per product where the ladder has 255 steps of the same size; the chunk's partial sum
//                     goes to the projective staging buffer, k_pair_sum and k_finish (the constant-time path's own tail) do the rest.
//   k_lagrange_at_zero the Lagrange coefficients at 0 of recover_commit (poly.rs:580-594) for m share sets, one lane per coefficient
// Exact on every curve point (the complete addition law; small-order and mixed-order points included).  NOT constant time: table
// addresses and the skipped zero digits depend on the scalars — which is why the entry point exists only in a "public" form.
#include <hip/hip_runtime.h>
#include "launch.h"
#include "ge_scalarmult.h"
#include "sc25519.h"
using namespace kyb;
#include "device_tables.h"

constexpr int MSM_WINDOWS = 43, MSM_ENTRIES = 32;

__device__ __forceinline__ void msm_load40(uint32_t f[40], const uint32_t* __restrict__ p) {
  const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
  for (int i = 0; i < 10; ++i) { const uint4 v = q[i]; f[4 * i] = v.x; f[4 * i + 1] = v.y; f[4 * i + 2] = v.z; f[4 * i + 3] = v.w; }
}
__device__ __forceinline__ void msm_store_cached(uint32_t* __restrict__ p, const ge_cached& c) {
  uint32_t f[40];
#pragma unroll
  for (int k = 0; k < 10; ++k) { f[k] = c.YpX.v[k]; f[10 + k] = c.YmX.v[k]; f[20 + k] = c.Z.v[k]; f[30 + k] = c.T2d.v[k]; }
  uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
  for (int i = 0; i < 10; ++i) q[i] = make_uint4(f[4 * i], f[4 * i + 1], f[4 * i + 2], f[4 * i + 3]);
}

// lane (w, j), w-major: the lanes of a wavefront share their window
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_msm_tables(const uint32_t* __restrict__ bases, size_t t, uint32_t* __restrict__ tab) {
  const size_t id = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (id >= t * MSM_WINDOWS) return;
  const size_t w = id / t, j = id - w * t;
  uint32_t f[40];
  msm_load40(f, bases + (j * MSM_WINDOWS + w) * 40);
  ge_p3 E;
#pragma unroll
  for (int k = 0; k < 10; ++k) { E.X.v[k] = f[k]; E.Y.v[k] = f[10 + k]; E.Z.v[k] = f[20 + k]; E.T.v[k] = f[30 + k]; }
  ge_cached B, c;
  ge_p3_to_cached(B, E);
  fe_reduce_weak(B.YpX, B.YpX); fe_reduce_weak(B.YmX, B.YmX); fe_reduce_weak(B.T2d, B.T2d);      // tight: inside every bound ge_add states
  uint32_t* out = tab + ((j * MSM_WINDOWS + w) * MSM_ENTRIES) * 40;
  msm_store_cached(out, B);
#pragma unroll 1
  for (int e = 1; e < MSM_ENTRIES; ++e) {
    ge_p1p1 r;
    ge_add(r, E, B);
    ge_p1p1_to_p3_after_add(E, r);
    ge_p3_to_cached(c, E);
    fe_reduce_weak(c.YpX, c.YpX); fe_reduce_weak(c.YmX, c.YmX); fe_reduce_weak(c.T2d, c.T2d);
    msm_store_cached(out + 40 * e, c);
  }
}

// lane (chunk, g), chunk-major: the lanes of a wavefront walk the same points' tables
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_msm_accumulate(const uint8_t* __restrict__ scalars, const uint32_t* __restrict__ tab, size_t m, size_t t, int chunk, size_t nchunks,
                 uint4* __restrict__ proj, size_t stride) {
  const size_t id = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (id >= m * nchunks) return;
  const size_t c = id / m, g = id - c * m;
  const size_t j_lo = c * (size_t)chunk, j_hi = (j_lo + (size_t)chunk < t) ? j_lo + (size_t)chunk : t;
  ge_p3 acc;
  ge_p3_0(acc);
#pragma unroll 1
  for (size_t j = j_lo; j < j_hi; ++j) {
    uint32_t a[8], neg, mag[8];
    load_words8(a, scalars, g * t + j);
    sc_effective(neg, mag, a);                       // the integer the reference multiplies by (|.| < 9 * 2^252), and its sign
    const uint32_t* tj = tab + (j * MSM_WINDOWS) * MSM_ENTRIES * 40;
    uint32_t carry = 0;
#pragma unroll 1
    for (int w = 0; w < MSM_WINDOWS; ++w) {
      const uint32_t raw = (mag[0] & 63u) + carry;   // 0 .. 64
      KYB_UNROLL for (int i = 0; i < 7; ++i) mag[i] = (mag[i] >> 6) | (mag[i + 1] << 26);
      mag[7] >>= 6;
      carry = raw >= 32u ? 1u : 0u;
      const int d = (int)raw - (int)(carry << 6);    // -32 .. 31
      if (d == 0) continue;                          // public scalars: nothing to hide
      const uint32_t dneg = (uint32_t)(d < 0) ^ neg;
      const uint32_t idx = (uint32_t)(d < 0 ? -d : d) - 1u;
      uint32_t f[40];
      msm_load40(f, tj + ((size_t)w * MSM_ENTRIES + idx) * 40);
      ge_cached q;
#pragma unroll
      for (int k = 0; k < 10; ++k) { q.YpX.v[k] = f[k]; q.YmX.v[k] = f[10 + k]; q.Z.v[k] = f[20 + k]; q.T2d.v[k] = f[30 + k]; }
      ge_cached_cneg(q, dneg);
      ge_p1p1 r;
      ge_add(r, acc, q);
      ge_p1p1_to_p3_after_add(acc, r);
    }
  }
  store_proj(proj, stride, g * nchunks + c, acc.X, acc.Y, acc.Z);
}

// Lagrange coefficients at 0 of recover_commit (poly.rs:580-594): for share set g with indices idx[g][0 .. t), x_i = idx_i + 1 as Scalars,
//     lambda[g][i] = prod_{j != i} x_j * ( prod_{j != i} (x_j - x_i) )^(L - 2)   mod L
// — the reference's num / den with its Scalar::div = multiplication by den^(L-2) (scalar.rs:185-215; the inverse of 0 is 0).  One lane per
// (g, i), where the reference walks the t^2 products on one core.  Share indices are public, so nothing here needs to be constant time, and
// a multiplication mod L (sc_muladd: a 512-bit product and its reduction, ~1 us of a lone lane) is what the time goes to:
//   * the factors are small — x_j <= 2^32, |x_j - x_i| < 2^32, ten bits for the share indices of a real DKG — so F = floor(63 / bits of the
//     group's largest x) of them are multiplied in a 64-bit word first and only every F-th step is a multiplication mod L (the sign of the
//     denominator is counted and applied once); F is the same for every lane of a group, the schedule is uniform;
//   * den^(L - 2) walks the FIXED exponent with a sliding window over the odd powers den^1 .. den^15: 253 squarings, 8 + 28 products instead of
//     253 + 253.
// 1,870 -> 560 multiplications mod L for one set of 683 ten-bit indices.
__device__ __forceinline__ void sc_from_u64(uint32_t r[8], uint64_t v) {
  r[0] = (uint32_t)v; r[1] = (uint32_t)(v >> 32);
  KYB_UNROLL for (int i = 2; i < 8; ++i) r[i] = 0u;
}
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_lagrange_at_zero(const uint32_t* __restrict__ idx, size_t m, size_t t, uint8_t* __restrict__ out) {
  const size_t id = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (id >= m * t) return;
  const size_t g = id / t, i = id - g * t;
  const uint32_t* xs = idx + g * t;
  const uint32_t Lw[8] = KYB_W_L;
  const uint32_t zero[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  const uint64_t xi = (uint64_t)xs[i] + 1u;
  uint32_t top = 0;
#pragma unroll 1
  for (size_t j = 0; j < t; ++j) top = xs[j] > top ? xs[j] : top;
  const int bits = 64 - __clzll((unsigned long long)top + 1ull);      // of the largest x (and of every difference): 1 .. 33
  const int F = 63 / bits;                                             // factors per 64-bit word
  uint32_t num[8], den[8];
  sc_from_u64(num, 1u);
  sc_from_u64(den, 1u);
  uint64_t pn = 1, pd = 1;
  uint32_t neg = 0;
  int cnt = 0;
#pragma unroll 1
  for (size_t j = 0; j < t; ++j) {
    const uint64_t xj = (uint64_t)xs[j] + 1u;
    const bool skip = j == i, below = xj < xi;
    pn *= skip ? 1ull : xj;
    pd *= skip ? 1ull : (below ? xi - xj : xj - xi);                   // |x_j - x_i|; 0 for a repeated index: the coefficient becomes 0
    neg ^= (uint32_t)(below && !skip);
    if (++cnt == F || j + 1 == t) {                                    // uniform in the group: F and t are
      uint32_t a[8], r[8];
      sc_from_u64(a, pn);
      sc_muladd(r, num, a, zero);
      KYB_UNROLL for (int q = 0; q < 8; ++q) num[q] = r[q];
      sc_from_u64(a, pd);
      sc_muladd(r, den, a, zero);
      KYB_UNROLL for (int q = 0; q < 8; ++q) den[q] = r[q];
      pn = pd = 1; cnt = 0;
    }
  }
  {                                                                    // an odd number of negative factors: den = L - den (0 stays 0)
    uint32_t nd[8], nz = 0;
    mw_sub<8>(nd, Lw, den);
    KYB_UNROLL for (int q = 0; q < 8; ++q) nz |= den[q];
    const bool flip = neg != 0u && nz != 0u;
    KYB_UNROLL for (int q = 0; q < 8; ++q) den[q] = flip ? nd[q] : den[q];
  }
  // den^(L - 2), top bit first, sliding window of 4 bits over the odd powers (the exponent is a constant: every branch below is uniform)
  uint32_t e[8];
  { const uint32_t two[8] = {2u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}; mw_sub<8>(e, Lw, two); }
  uint32_t p1[8], p3[8], p5[8], p7[8], p9[8], p11[8], p13[8], p15[8], d2[8];      // (named, not an array: nothing may index them at run time)
  KYB_UNROLL for (int q = 0; q < 8; ++q) p1[q] = den[q];
  sc_muladd(d2, den, den, zero);
  sc_muladd(p3, p1, d2, zero); sc_muladd(p5, p3, d2, zero); sc_muladd(p7, p5, d2, zero); sc_muladd(p9, p7, d2, zero);
  sc_muladd(p11, p9, d2, zero); sc_muladd(p13, p11, d2, zero); sc_muladd(p15, p13, d2, zero);
  uint32_t inv[8];
  sc_from_u64(inv, 1u);
  auto ebit = [&](int b) -> uint32_t {
    uint32_t w = 0;
    KYB_UNROLL for (int q = 0; q < 8; ++q) w = (b >> 5) == q ? e[q] : w;
    return (w >> (b & 31)) & 1u;
  };
  int b = 252;
#pragma unroll 1
  while (b >= 0) {
    uint32_t r[8];
    if (ebit(b) == 0u) {
      sc_muladd(r, inv, inv, zero);
      KYB_UNROLL for (int q = 0; q < 8; ++q) inv[q] = r[q];
      --b;
      continue;
    }
    int lo = b - 3 < 0 ? 0 : b - 3;                                   // the window [b .. lo] ends on a set bit
    while (ebit(lo) == 0u) ++lo;
    uint32_t v = 0;
    for (int c = b; c >= lo; --c) v = (v << 1) | ebit(c);
#pragma unroll 1
    for (int c = b; c >= lo; --c) {
      sc_muladd(r, inv, inv, zero);
      KYB_UNROLL for (int q = 0; q < 8; ++q) inv[q] = r[q];
    }
    uint32_t f[8];
    const uint32_t h = v >> 1;
    KYB_UNROLL for (int q = 0; q < 8; ++q)
      f[q] = h == 0u ? p1[q] : h == 1u ? p3[q] : h == 2u ? p5[q] : h == 3u ? p7[q] : h == 4u ? p9[q] : h == 5u ? p11[q] : h == 6u ? p13[q] : p15[q];
    sc_muladd(r, inv, f, zero);
    KYB_UNROLL for (int q = 0; q < 8; ++q) inv[q] = r[q];
    b = lo - 1;
  }
  uint32_t lam[8];
  sc_muladd(lam, num, inv, zero);
  store_words8(out, id, lam);
}

namespace kyb { namespace launch {
hipError_t lagrange_at_zero(hipStream_t st, const uint32_t* idx, size_t m, size_t t, uint8_t* out) {
  hipLaunchKernelGGL(k_lagrange_at_zero, dim3((unsigned)((m * t + KYB_BLOCK - 1) / KYB_BLOCK)), dim3(KYB_BLOCK), 0, st, idx, m, t, out);
  return hipGetLastError();
}
static inline unsigned blocks_for(size_t n) { return (unsigned)((n + KYB_BLOCK - 1) / KYB_BLOCK); }
hipError_t msm_tables(hipStream_t st, const uint32_t* bases, size_t t, uint32_t* tab) {
  hipLaunchKernelGGL(k_msm_tables, dim3(blocks_for(t * MSM_WINDOWS)), dim3(KYB_BLOCK), 0, st, bases, t, tab);
  return hipGetLastError();
}
hipError_t msm_accumulate(hipStream_t st, const uint8_t* scalars, const uint32_t* tab, size_t m, size_t t, int chunk, size_t nchunks, uint4* proj, size_t stride) {
  hipLaunchKernelGGL(k_msm_accumulate, dim3(blocks_for(m * nchunks)), dim3(KYB_BLOCK), 0, st, scalars, tab, m, t, chunk, nchunks, proj, stride);
  return hipGetLastError();
}
}}  // namespace kyb::launch
