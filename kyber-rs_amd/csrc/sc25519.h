// Scalar arithmetic modulo the group order L = 2^252 + 27742317777372353535851937790883648493.
//
// Functional counterpart of /root/reference src/group/edwards25519/scalar.rs: sc_mul_add
// (scalar.rs:279-744), sc_add (:759), sc_sub (:1187), sc_mul (:1596) and of Scalar::set_bytes
// (scalar.rs:175-177 -> integer.rs:386-397, little-endian bytes mod L).  The reference uses ref10's
// 12x21-bit signed limbs; its outputs are the canonical residues in [0,L) for ANY 256-bit inputs
// (clamped EdDSA keys are passed unreduced, eddsa_sig.rs:142), so any exact mod-L arithmetic is
// bit-identical.  Here: 32-bit words, schoolbook product, and folding with 2^252 = -c (mod L),
// c = L - 2^252 (125 bits).  Shared by the device sign kernel and the host-side Scalar class.
#pragma once
#include <stdint.h>
#include "fe25519.h"  // KYB_HD, consts

namespace kyb {

// r[0..NA+NB) = a * b
template <int NA, int NB>
KYB_HD void mw_mul(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  KYB_UNROLL for (int i = 0; i < NA + NB; ++i) r[i] = 0;
  KYB_UNROLL for (int i = 0; i < NA; ++i) {
    uint64_t c = 0;
    KYB_UNROLL for (int j = 0; j < NB; ++j) {
      c += (uint64_t)a[i] * b[j] + r[i + j];
      r[i + j] = (uint32_t)c;
      c >>= 32;
    }
    r[i + NB] = (uint32_t)c;
  }
}
// r = a + b (N words), returns carry
template <int N>
KYB_HD uint32_t mw_add(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  uint64_t c = 0;
  KYB_UNROLL for (int i = 0; i < N; ++i) {
    c += (uint64_t)a[i] + b[i];
    r[i] = (uint32_t)c;
    c >>= 32;
  }
  return (uint32_t)c;
}
// r = a - b (N words), returns borrow (1 if a < b)
template <int N>
KYB_HD uint32_t mw_sub(uint32_t* r, const uint32_t* a, const uint32_t* b) {
  int64_t c = 0;
  KYB_UNROLL for (int i = 0; i < N; ++i) {
    c += (int64_t)a[i] - (int64_t)b[i];
    r[i] = (uint32_t)c;
    c >>= 32;  // arithmetic: 0 or -1
  }
  return (uint32_t)(c & 1);
}
// split x (N words) at bit 252: lo[8] = x mod 2^252, hi[N-7] = x >> 252
template <int N>
KYB_HD void mw_split252(uint32_t lo[8], uint32_t* hi, const uint32_t* x) {
  KYB_UNROLL for (int i = 0; i < 7; ++i) lo[i] = x[i];
  lo[7] = x[7] & 0x0fffffffu;
  KYB_UNROLL for (int i = 0; i < N - 7; ++i) {
    uint32_t a = x[7 + i] >> 28;
    uint32_t b = (8 + i < N) ? (x[8 + i] << 4) : 0u;
    hi[i] = a | b;
  }
}

// out[8] = x mod L for a 17-word x (< 2^544; callers pass < 2^513)
KYB_HD void sc_reduce544(uint32_t out[8], const uint32_t x[17]) {
  const uint32_t c4[4] = KYB_W_LC;
  const uint32_t Lw[8] = KYB_W_L;
  uint32_t lo0[8], hi0[10];
  mw_split252<17>(lo0, hi0, x);              // hi0 < 2^292 (10 words)
  uint32_t y1[14];
  mw_mul<10, 4>(y1, hi0, c4);                // c*hi0 < 2^417
  uint32_t lo1[8], hi1[7];
  mw_split252<14>(lo1, hi1, y1);             // hi1 < 2^165 (6 words used)
  uint32_t y2[11];
  mw_mul<7, 4>(y2, hi1, c4);                 // < 2^290
  uint32_t lo2[8], hi2[4];
  mw_split252<11>(lo2, hi2, y2);             // hi2 < 2^38 (2 words used)
  uint32_t y3[8];
  mw_mul<4, 4>(y3, hi2, c4);                 // < 2^163, below 2^252
  // x = lo0 - lo1 + lo2 - y3 (mod L), |value| < 2^253.  Work in 9 words with a 2L bias.
  uint32_t acc[9], t[9], L2[9];
  KYB_UNROLL for (int i = 0; i < 8; ++i) { acc[i] = lo0[i]; L2[i] = (Lw[i] << 1) | (i ? (Lw[i - 1] >> 31) : 0u); }
  acc[8] = 0; L2[8] = Lw[7] >> 31;
  mw_add<9>(acc, acc, L2);                   // + 2L
  KYB_UNROLL for (int i = 0; i < 8; ++i) t[i] = lo2[i];
  t[8] = 0;
  mw_add<9>(acc, acc, t);
  KYB_UNROLL for (int i = 0; i < 8; ++i) t[i] = lo1[i];
  mw_sub<9>(acc, acc, t);
  KYB_UNROLL for (int i = 0; i < 8; ++i) t[i] = y3[i];
  mw_sub<9>(acc, acc, t);
  // 0 < acc < 2L + 2^253 < 4L : subtract L up to three times, branch-free
  uint32_t L1[9];
  KYB_UNROLL for (int i = 0; i < 8; ++i) L1[i] = Lw[i];
  L1[8] = 0;
  KYB_UNROLL for (int k = 0; k < 3; ++k) {
    uint32_t borrow = mw_sub<9>(t, acc, L1);
    KYB_UNROLL for (int i = 0; i < 9; ++i) acc[i] = borrow ? acc[i] : t[i];
  }
  KYB_UNROLL for (int i = 0; i < 8; ++i) out[i] = acc[i];
}

// out = (a*b + c) mod L, any 256-bit a, b, c   (sc_mul_add, scalar.rs:279-744)
KYB_HD void sc_muladd(uint32_t out[8], const uint32_t a[8], const uint32_t b[8], const uint32_t c[8]) {
  uint32_t x[17];
  mw_mul<8, 8>(x, a, b);
  x[16] = 0;
  uint64_t k = 0;
  KYB_UNROLL for (int i = 0; i < 17; ++i) {
    k += (uint64_t)x[i] + (i < 8 ? c[i] : 0u);
    x[i] = (uint32_t)k;
    k >>= 32;
  }
  sc_reduce544(out, x);
}
// x^e as a multiplier of curve points: a point's order divides 8L (cofactor 8), so the integer x^e may be replaced by any
// representative of x^e mod 8L — mod L alone would change the result on points with a small-order component, which the reference's
// exact Horner evaluation (poly.rs:457-469) treats like any other.  Returns the representative of smallest magnitude, |v| < 4L
// < 2^255 (below the top-digit quirk of the multiplication routines), as (mag, neg).  x^e mod L by square-and-multiply on
// sc_muladd, x^e mod 8 in machine arithmetic, recombined with L^-1 = 5 (mod 8).
KYB_HD void sc_pow_mod8L_signed(uint32_t mag[8], uint32_t& neg, uint32_t x, uint32_t e) {
  const uint32_t Lw[8] = {0x5cf5d3edu, 0x5812631au, 0xa2f79cd6u, 0x14def9deu, 0u, 0u, 0u, 0x10000000u};
  uint32_t a[8] = {1u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, xs[8] = {x, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  const uint32_t zero[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  uint32_t b = 1u;
  int top = 31;
  while (top > 0 && !((e >> top) & 1u)) --top;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int i = top; i >= 0; --i) {
    uint32_t t[8];
    sc_muladd(t, a, a, zero);
    KYB_UNROLL for (int j = 0; j < 8; ++j) a[j] = t[j];
    b = (b * b) & 7u;
    if ((e >> i) & 1u) {
      sc_muladd(t, a, xs, zero);
      KYB_UNROLL for (int j = 0; j < 8; ++j) a[j] = t[j];
      b = (b * x) & 7u;
    }
  }
  // v = a + k L with v = b (mod 8):  k = 5 (b - a) mod 8
  const uint32_t k = (5u * ((b - a[0]) & 7u)) & 7u;
  uint32_t v[8], kl[8];
  {
    uint64_t c = 0;
    KYB_UNROLL for (int j = 0; j < 8; ++j) { c += (uint64_t)Lw[j] * k; kl[j] = (uint32_t)c; c >>= 32; }      // k L < 8L < 2^256
  }
  mw_add<8>(v, a, kl);
  uint32_t l4[8], l8[8], d[8];
  KYB_UNROLL for (int j = 7; j >= 0; --j) {
    l4[j] = (Lw[j] << 2) | (j ? Lw[j - 1] >> 30 : 0u);
    l8[j] = (Lw[j] << 3) | (j ? Lw[j - 1] >> 29 : 0u);
  }
  const uint32_t below = mw_sub<8>(d, v, l4);          // borrow: v < 4L
  mw_sub<8>(d, l8, v);                                  // 8L - v
  neg = 1u - below;
  KYB_UNROLL for (int j = 0; j < 8; ++j) mag[j] = below ? v[j] : d[j];
}
// out = x mod L for a 512-bit little-endian x (Scalar::set_bytes on a SHA-512 digest)
KYB_HD void sc_reduce512(uint32_t out[8], const uint32_t x16[16]) {
  uint32_t x[17];
  KYB_UNROLL for (int i = 0; i < 16; ++i) x[i] = x16[i];
  x[16] = 0;
  sc_reduce544(out, x);
}
// out = x mod L for a 256-bit x (Scalar::marshal_binary, scalar.rs:91-100)
KYB_HD void sc_reduce256(uint32_t out[8], const uint32_t x8[8]) {
  uint32_t x[17];
  KYB_UNROLL for (int i = 0; i < 17; ++i) x[i] = i < 8 ? x8[i] : 0u;
  sc_reduce544(out, x);
}

}  // namespace kyb
