// Group operations on edwards25519 for the MI355X engine (shared host-test / device source).
//
// Functional counterpart of /root/reference src/group/edwards25519/ge.rs:
//   representations  ge.rs:20-26,78-83,203-208,301-305,330-335  (P2 / P3 / P1P1 / precomp / cached)
//   doubling         ge.rs:35-49        addition ge.rs:217-233   mixed addition ge.rs:274-290
//   conversions      ge.rs:211-215,292-297,93-98
//   encode / decode  ge.rs:112-122 / ge.rs:124-179
//   recoding + scalar mult   ge.rs:442-486 (fixed base), ge.rs:508-568 (variable base)
// Formulas are the a=-1 twisted-Edwards ones (any complete formula set gives the same group
// element; parity with the reference is defined on the 32-byte encoding).
// Limb bounds (see fe25519.h) are tracked in the comments: every p1p1 produced here satisfies
//   X <= 5T, Y <= 2T, Z <= 3T, T tight
// which is what p1p1_to_p2 / p1p1_to_p3 require.
#pragma once
#include "fe25519.h"
#include "fe_invert_gcd.h"

namespace kyb {

struct ge_p2 { fe X, Y, Z; };
struct ge_p3 { fe X, Y, Z, T; };
struct ge_p1p1 { fe X, Y, Z, T; };
struct ge_precomp { fe ypx, ymx, xy2d; };   // affine (y+x, y-x, 2dxy), canonical limbs
struct ge_cached { fe YpX, YmX, Z, T2d; };

KYB_HD void ge_p3_0(ge_p3& h) { fe_zero(h.X); fe_one(h.Y); fe_one(h.Z); fe_zero(h.T); }
// neutral element in completed form: x = X/Z = 0, y = Y/T = 1
KYB_HD void ge_p1p1_0(ge_p1p1& h) { fe_zero(h.X); fe_one(h.Y); fe_one(h.Z); fe_one(h.T); }

// ge.rs:211-215  (3M)
KYB_HD void ge_p1p1_to_p2(ge_p2& r, const ge_p1p1& p) {
  fe_mul_b6(r.X, p.X, p.T);     // 5T x tight
  fe_mul_b6(r.Y, p.Z, p.Y);     // 3T x 2T
  fe_mul_b6(r.Z, p.Z, p.T);     // 3T x tight
}
// ge.rs:292-297  (4M)
KYB_HD void ge_p1p1_to_p3(ge_p3& r, const ge_p1p1& p) {
  fe_mul_b6(r.X, p.X, p.T);     // 5T x tight
  fe_mul_b6(r.Y, p.Z, p.Y);     // 3T x 2T
  fe_mul_b6(r.Z, p.Z, p.T);     // 3T x tight
  fe_mul(r.T, p.X, p.Y);        // 5T x 2T after a doubling: general fold
}
// the same for a p1p1 that came out of ge_add / ge_madd (X <= 3T): every product is within the short fold's bound
KYB_HD void ge_p1p1_to_p3_after_add(ge_p3& r, const ge_p1p1& p) {
  fe_mul_b6(r.X, p.X, p.T);
  fe_mul_b6(r.Y, p.Z, p.Y);
  fe_mul_b6(r.Z, p.Z, p.T);
  fe_mul_b6(r.T, p.X, p.Y);     // 3T x 2T
}

// ge.rs:35-49  (4S).  p tight.
KYB_HD void ge_p2_dbl(ge_p1p1& r, const fe& X, const fe& Y, const fe& Z) {
  fe xx, yy, zz, a, aa;
  fe_sq_b2(xx, X);
  fe_sq_b2(yy, Y);
  fe_sq_b2(zz, Z);
  fe_addw(a, X, Y);              // 2T
  fe_sq_b2(aa, a);
  fe_addw(r.Y, yy, xx);          // 2T
  fe_subw(r.Z, yy, xx);          // 3T
  fe_subw(r.X, aa, yy);          // 3T
  fe_subw(r.X, r.X, xx);         // 5T   = (X+Y)^2 - Y^2 - X^2
  fe t;
  fe_addw(t, zz, zz);            // 2T
  fe_addw(t, t, xx);             // 3T
  fe_subw(t, t, yy);             // 5T   = 2Z^2 - (Y^2 - X^2)
  fe_reduce_weak(r.T, t);       // tight
}

// ge.rs:217-233  (4M).  p tight; q: YpX <= 2T, YmX <= 3T, Z tight, T2d <= 2T
KYB_HD void ge_add(ge_p1p1& r, const ge_p3& p, const ge_cached& q) {
  fe a, b, A, B, C, D, t;
  fe_addw(a, p.Y, p.X);          // 2T
  fe_subw(b, p.Y, p.X);          // 3T
  fe_mul_b6(A, a, q.YpX);       // 2T x 2T
  fe_mul(B, b, q.YmX);          // 3T x 3T: general fold
  fe_mul_b6(C, q.T2d, p.T);     // 2T x tight
  fe_mul_b6(D, p.Z, q.Z);       // tight x tight
  fe_addw(D, D, D);              // 2T
  fe_subw(r.X, A, B);            // 3T
  fe_addw(r.Y, A, B);            // 2T
  fe_addw(r.Z, D, C);            // 3T
  fe_subw(t, D, C);              // 4T
  fe_reduce_weak(r.T, t);
}

// ge.rs:274-290  (3M).  p tight; q: ypx, ymx tight, xy2d <= 2T
KYB_HD void ge_madd(ge_p1p1& r, const ge_p3& p, const ge_precomp& q) {
  fe a, b, A, B, C, D, t;
  fe_addw(a, p.Y, p.X);
  fe_subw(b, p.Y, p.X);
  fe_mul_b6(A, a, q.ypx);       // 2T x tight
  fe_mul_b6(B, b, q.ymx);       // 3T x tight
  fe_mul_b6(C, q.xy2d, p.T);    // 2T x tight
  fe_addw(D, p.Z, p.Z);
  fe_subw(r.X, A, B);
  fe_addw(r.Y, A, B);
  fe_addw(r.Z, D, C);
  fe_subw(t, D, C);
  fe_reduce_weak(r.T, t);
}

// ge.rs:93-98  (1M)
KYB_HD void ge_p3_to_cached(ge_cached& r, const ge_p3& p) {
  const fe d2 = {KYB_FE_D2};
  fe_add(r.YpX, p.Y, p.X);
  fe_sub(r.YmX, p.Y, p.X);
  fe_copy(r.Z, p.Z);
  fe_mul(r.T2d, p.T, d2);
}

// (X:Y:Z) -> extended (XZ : YZ : Z^2 : XY), 3M + 1S; and the sum of two projective points through
// the unified (complete) addition above -- the accumulation step of recover_commit (share/poly.rs:596-597)
// when the operands come out of a scalar multiplication in projective form.
KYB_HD void ge_p2_to_p3(ge_p3& r, const ge_p2& p) {
  fe_mul(r.X, p.X, p.Z);
  fe_mul(r.Y, p.Y, p.Z);
  fe_sq(r.Z, p.Z);
  fe_mul(r.T, p.X, p.Y);
}
KYB_HD void ge_p2_add(ge_p2& r, const ge_p2& a, const ge_p2& b) {
  ge_p3 A, B;
  ge_p2_to_p3(A, a);
  ge_p2_to_p3(B, b);
  ge_cached c;
  ge_p3_to_cached(c, B);
  ge_p1p1 t;
  ge_add(t, A, c);
  ge_p1p1_to_p2(r, t);
}

// The fixed-base loop's pair: the mixed addition leaves T = D - C unreduced (<= 4T) and the conversion takes it as the
// FIRST operand of its two products with T (first operands may be <= 6T): one 30-instruction carry pass less per window
// for two general folds (6 instructions) more.
KYB_HD void ge_madd_lazy_t(ge_p1p1& r, const ge_p3& p, const ge_precomp& q) {
  fe a, b, A, B, C, D;
  fe_addw(a, p.Y, p.X);
  fe_subw(b, p.Y, p.X);
  fe_mul_b6(A, a, q.ypx);       // 2T x tight
  fe_mul_b6(B, b, q.ymx);       // 3T x tight
  fe_mul_b6(C, q.xy2d, p.T);    // 2T x tight
  fe_addw(D, p.Z, p.Z);          // 2T
  fe_subw(r.X, A, B);            // 3T
  fe_addw(r.Y, A, B);            // 2T
  fe_addw(r.Z, D, C);            // 3T
  fe_subw(r.T, D, C);            // 4T
}
// the same mixed addition in two halves (A, B first; then C and the sums) for callers that start the next window's table reads in between
KYB_HD void ge_madd_lazy_t_ab(fe& A, fe& B, const ge_p3& p, const ge_precomp& q) {
  fe a, b;
  fe_addw(a, p.Y, p.X);
  fe_subw(b, p.Y, p.X);
  fe_mul_b6(A, a, q.ypx);
  fe_mul_b6(B, b, q.ymx);
}
KYB_HD void ge_madd_lazy_t_c(ge_p1p1& r, const fe& A, const fe& B, const ge_p3& p, const ge_precomp& q) {
  fe C, D;
  fe_mul_b6(C, q.xy2d, p.T);
  fe_addw(D, p.Z, p.Z);
  fe_subw(r.X, A, B);
  fe_addw(r.Y, A, B);
  fe_addw(r.Z, D, C);
  fe_subw(r.T, D, C);
}
KYB_HD void ge_p1p1_to_p3_lazy_t(ge_p3& r, const ge_p1p1& p) {
  fe_mul(r.X, p.T, p.X);        // 4T x 3T: general fold
  fe_mul_b6(r.Y, p.Z, p.Y);     // 3T x 2T
  fe_mul(r.Z, p.T, p.Z);        // 4T x 3T: general fold
  fe_mul_b6(r.T, p.X, p.Y);     // 3T x 2T
}

// conditional negation of a cached / precomputed entry (ge.rs:314-318, 354-359), neg in {0,1}
KYB_HD void ge_cached_cneg(ge_cached& c, uint32_t neg) {
  fe nt;
  fe_cswap(c.YpX, c.YmX, neg);
  fe_neg(nt, c.T2d);
  fe_cmov(c.T2d, nt, neg);
}
KYB_HD void ge_precomp_cneg(ge_precomp& c, uint32_t neg) {
  fe nt;
  fe_cswap(c.ypx, c.ymx, neg);
  fe_neg(nt, c.xy2d);
  fe_cmov(c.xy2d, nt, neg);
}

// 32-byte encoding as 8 LE words (ge.rs:112-122): y with the sign of x in bit 255
KYB_HD void ge_encode(uint32_t w[8], const fe& X, const fe& Y, const fe& Z) {
  fe recip, x, y;
  fe_inv(recip, Z);
  fe_mul(x, X, recip);
  fe_mul(y, Y, recip);
  fe_to_words(w, y);
  w[7] ^= fe_is_negative(x) << 31;
}
// same, given 1/Z (batched inversion path)
KYB_HD void ge_encode_with_recip(uint32_t w[8], const fe& X, const fe& Y, const fe& recip) {
  fe x, y;
  fe_mul(x, X, recip);
  fe_mul(y, Y, recip);
  fe_to_words(w, y);
  w[7] ^= fe_is_negative(x) << 31;
}

// Decode (ge.rs:124-179).  Returns 1 on success.  Quirks kept: bit 255 of y ignored by the field
// load, y >= p accepted, x = 0 with sign bit set accepted (x stays 0), Z = 1, T = x*y.
// `pow22523(out, z)` computes z^((p-5)/8) for a tight z: fe_pow22523 in the batch kernels, the lane-cooperative chain in the
// small-batch kernels (kernels_coop.hip) — 252 of the decode's ~270 dependent multiplications.
template <class Pow>
KYB_HD uint32_t ge_decode_with(ge_p3& h, const uint32_t w[8], Pow&& pow22523) {
  const fe d = {KYB_FE_D};
  const fe sqrtm1 = {KYB_FE_SQRTM1};
  fe u, v, v3, vxx, chk, x, xm;
  fe_from_words(h.Y, w);
  fe_one(h.Z);
  fe_sq(u, h.Y);
  fe_mul(v, u, d);
  fe_sub(u, u, h.Z);            // u = y^2 - 1   (3T)
  fe_add(v, v, h.Z);            // v = d y^2 + 1 (2T)
  fe_sq(v3, v);
  fe_mul(v3, v3, v);            // v^3
  fe_sq(x, v3);
  fe_mul(x, x, v);
  fe_mul(x, x, u);              // u v^7
  pow22523(x, x);
  fe_mul(x, x, v3);
  fe_mul(x, x, u);              // u v^3 (u v^7)^((p-5)/8)
  fe_sq(vxx, x);
  fe_mul(vxx, vxx, v);
  fe_sub4(chk, vxx, u);         // v x^2 - u      (u is 3T -> 4p bias)
  uint32_t nz1 = fe_is_nonzero(chk);
  fe_add(chk, vxx, u);          // v x^2 + u
  uint32_t nz2 = fe_is_nonzero(chk);
  fe_mul(xm, x, sqrtm1);
  fe_cmov(x, xm, nz1);          // x *= sqrt(-1) when v x^2 != u
  uint32_t ok = 1u - (nz1 & nz2);
  uint32_t sign = w[7] >> 31;
  uint32_t flip = fe_is_negative(x) ^ sign;
  fe xn;
  fe_neg(xn, x);
  fe_reduce_weak(xn, xn);
  fe_cmov(x, xn, flip);
  fe_copy(h.X, x);
  fe_mul(h.T, h.X, h.Y);
  return ok;
}
struct fe_pow22523_fn { KYB_HD void operator()(fe& o, const fe& z) const { fe_pow22523(o, z); } };
KYB_HD uint32_t ge_decode(ge_p3& h, const uint32_t w[8]) { return ge_decode_with(h, w, fe_pow22523_fn()); }

// ---------------------------------------------------------------------------------------------
// Scalar recoding (ge.rs:443-459 / 519-534).  The reference turns the 64 nibbles of the scalar into
// signed digits e[i] in [-8,8) (i < 63) by a carry sweep; that sweep is exactly the 256-bit addition
//     b = a + 0x0888...8   (an 8 in every nibble 0..62)
// after which nibble i of b is e[i]+8 for i < 63 and the top nibble plus the carry out is e[63]
// (0..16, NOT recentred — ge.rs:459).  A top digit of 9..16 (scalar >= 2^255, precondition
// ge.rs:440-441 violated) matches no table entry and is silently dropped by the reference's select
// (ge.rs:423-434 / 488-500); top_digit() reproduces that by mapping 9..16 to 0.
struct sc_digits {
  uint32_t w[8];   // recoded nibbles, digit i (i<63) = nibble(i) - 8
  uint32_t top;    // e[63] as the reference's select sees it: 0..8
};
KYB_HD void sc_recode(sc_digits& d, const uint32_t a[8]) {
  uint64_t c = 0;
  KYB_UNROLL for (int i = 0; i < 8; ++i) {
    c += (uint64_t)a[i] + (i == 7 ? 0x08888888u : 0x88888888u);
    d.w[i] = (uint32_t)c;
    c >>= 32;
  }
  uint32_t e63 = (d.w[7] >> 28) + ((uint32_t)c << 4);   // 0..16
  d.top = (e63 <= 8u) ? e63 : 0u;
}
// digit i in 0..62 as (magnitude 0..8, negative flag); i must be wave-uniform on the device
KYB_HD void sc_digit(uint32_t& mag, uint32_t& neg, const sc_digits& d, int i) {
  uint32_t word = 0;
  KYB_UNROLL for (int k = 0; k < 8; ++k) word = ((i >> 3) == k) ? d.w[k] : word;
  int v = (int)((word >> ((i & 7) * 4)) & 15u) - 8;
  neg = v < 0;
  mag = neg ? (uint32_t)(-v) : (uint32_t)v;
}

// The integer a' the reference's routines multiply by (see sc_recode): a itself unless the top radix-16
// digit is 9..16, in which case that digit is dropped: a' = (a mod 2^252) - c63 * 2^252 with c63 the
// carry into digit 63.  Returned as sign (1 = negative) and 256-bit magnitude.
KYB_HD void sc_effective(uint32_t& neg, uint32_t mag[8], const uint32_t a[8]) {
  uint64_t c = 0;
  uint32_t b7 = 0;
  KYB_UNROLL for (int i = 0; i < 8; ++i) {
    c += (uint64_t)a[i] + (i == 7 ? 0x08888888u : 0x88888888u);
    if (i == 7) b7 = (uint32_t)c;
    c >>= 32;
  }
  const uint32_t e63 = (b7 >> 28) + ((uint32_t)c << 4);          // 0..16
  const uint32_t dropped = e63 > 8u;
  const uint32_t c63 = e63 - (a[7] >> 28);                       // carry into digit 63: 0 or 1
  // candidate magnitudes: a (not dropped), low252(a) (dropped, c63 = 0), 2^252 - low252(a) (dropped, c63 = 1)
  uint32_t low[8], negm[8];
  KYB_UNROLL for (int i = 0; i < 8; ++i) low[i] = a[i];
  low[7] &= 0x0fffffffu;
  int64_t bw = 0;
  KYB_UNROLL for (int i = 0; i < 8; ++i) {
    bw += (int64_t)(i == 7 ? 0x10000000u : 0u) - (int64_t)low[i];
    negm[i] = (uint32_t)bw;
    bw >>= 32;
  }
  neg = dropped & c63;
  KYB_UNROLL for (int i = 0; i < 8; ++i) mag[i] = dropped ? (c63 ? negm[i] : low[i]) : a[i];
}

}  // namespace kyb
