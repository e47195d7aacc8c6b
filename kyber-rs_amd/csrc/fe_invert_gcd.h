// Field inversion without the 254-squaring chain: Bernstein-Yang "safegcd" divsteps (Fast constant-time gcd computation and modular
// inversion, TCHES 2019) in the 32-bit formulation — 20 batches of 30 divsteps on the low words of (f, g), each batch summarised by a
// 2x2 matrix scaled by 2^30 and applied to f, g (exact division by 2^30) and to d, e (division mod p), nine signed 30-bit limbs each.
//
// Same function as fe_invert (fe.rs:857-944: z^(p-2), which maps 0 to 0): the inverse is unique mod p, so the canonical result is the same
// bit for bit; 0 -> 0 falls out of the algorithm (g = 0 keeps d = 0).  ~15,600 instructions, 1,800 of them multiply-adds, against
// 26,200 / 15,070 for the exponentiation — the batched finish / Montgomery-image kernels are latency-bound on exactly this chain
// (two wavefronts per SIMD at 2^20 items, one at DKG-sized batches).  Constant time: 600 divsteps whatever the input, selections by
// masks, no table, no branch (tools/ct_check.py covers k_finish, whose operand — Z of a shared secret — is the sensitive one).
// tools/safegcd_proto.py is the limb-exact prototype (every intermediate in the width used here, checked against x^(p-2)).
#pragma once
#include "fe25519.h"

namespace kyb {

struct gcd_mat { int32_t u, v, q, r; };

// 30 divsteps on the low words; returns the new zeta.  (delta = 1/2 variant: zeta = -(delta + 1/2), start -1.)
KYB_HD int32_t gcd_divsteps_30(int32_t zeta, uint32_t f0, uint32_t g0, gcd_mat& t) {
  uint32_t u = 1, v = 0, q = 0, r = 1, f = f0, g = g0;
  KYB_UNROLL for (int i = 0; i < 30; ++i) {
    uint32_t c1 = (uint32_t)(zeta >> 31);           // all ones iff zeta < 0
    const uint32_t c2 = 0u - (g & 1u);              // all ones iff g odd
    const uint32_t x = (f ^ c1) - c1, y = (u ^ c1) - c1, z = (v ^ c1) - c1;      // (f, u, v) negated when zeta < 0
    g += x & c2; q += y & c2; r += z & c2;
    c1 &= c2;
    zeta = (int32_t)((uint32_t)zeta ^ c1) - 1;      // zeta < 0 and g odd: zeta = -zeta - 2, else zeta - 1
    f += g & c1; u += q & c1; v += r & c1;
    g >>= 1; u <<= 1; v <<= 1;
  }
  t.u = (int32_t)u; t.v = (int32_t)v; t.q = (int32_t)q; t.r = (int32_t)r;
  return zeta;
}

#define KYB_GCD_M30 0x3fffffff
#define KYB_GCD_MOD {0x3fffffed, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x7fff}
#define KYB_GCD_MODINV30 0x179435e5u                /* p^-1 mod 2^30 */

// (d, e) <- (u d + v e, q d + r e) / 2^30 mod p; d, e in (-2p, p)
KYB_HD void gcd_update_de(int32_t d[9], int32_t e[9], const gcd_mat& t) {
  const int32_t mod[9] = KYB_GCD_MOD;
  const int32_t u = t.u, v = t.v, q = t.q, r = t.r;
  const int32_t sd = d[8] >> 31, se = e[8] >> 31;
  int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);       // start from a non-negative representative
  int64_t cd = (int64_t)u * d[0] + (int64_t)v * e[0];
  int64_t ce = (int64_t)q * d[0] + (int64_t)r * e[0];
  md -= (int32_t)((KYB_GCD_MODINV30 * (uint32_t)cd + (uint32_t)md) & KYB_GCD_M30);      // the multiple of p that clears the low 30 bits
  me -= (int32_t)((KYB_GCD_MODINV30 * (uint32_t)ce + (uint32_t)me) & KYB_GCD_M30);
  cd += (int64_t)mod[0] * md;
  ce += (int64_t)mod[0] * me;
  cd >>= 30; ce >>= 30;
  KYB_UNROLL for (int i = 1; i < 9; ++i) {
    cd += (int64_t)u * d[i] + (int64_t)v * e[i] + (int64_t)mod[i] * md;
    ce += (int64_t)q * d[i] + (int64_t)r * e[i] + (int64_t)mod[i] * me;
    d[i - 1] = (int32_t)cd & KYB_GCD_M30; cd >>= 30;
    e[i - 1] = (int32_t)ce & KYB_GCD_M30; ce >>= 30;
  }
  d[8] = (int32_t)cd; e[8] = (int32_t)ce;
}
// (f, g) <- (u f + v g, q f + r g) / 2^30 (exact)
KYB_HD void gcd_update_fg(int32_t f[9], int32_t g[9], const gcd_mat& t) {
  const int32_t u = t.u, v = t.v, q = t.q, r = t.r;
  int64_t cf = (int64_t)u * f[0] + (int64_t)v * g[0];
  int64_t cg = (int64_t)q * f[0] + (int64_t)r * g[0];
  cf >>= 30; cg >>= 30;
  KYB_UNROLL for (int i = 1; i < 9; ++i) {
    cf += (int64_t)u * f[i] + (int64_t)v * g[i];
    cg += (int64_t)q * f[i] + (int64_t)r * g[i];
    f[i - 1] = (int32_t)cf & KYB_GCD_M30; cf >>= 30;
    g[i - 1] = (int32_t)cg & KYB_GCD_M30; cg >>= 30;
  }
  f[8] = (int32_t)cf; g[8] = (int32_t)cg;
}
// r in (-2p, p) -> [0, p), negated first when sign < 0
KYB_HD void gcd_normalize(int32_t r[9], int32_t sign) {
  const int32_t mod[9] = KYB_GCD_MOD;
  int32_t add = r[8] >> 31;
  KYB_UNROLL for (int i = 0; i < 9; ++i) r[i] += mod[i] & add;
  const int32_t neg = sign >> 31;
  KYB_UNROLL for (int i = 0; i < 9; ++i) r[i] = (r[i] ^ neg) - neg;
  KYB_UNROLL for (int i = 0; i < 8; ++i) { r[i + 1] += r[i] >> 30; r[i] &= KYB_GCD_M30; }
  add = r[8] >> 31;
  KYB_UNROLL for (int i = 0; i < 9; ++i) r[i] += mod[i] & add;
  KYB_UNROLL for (int i = 0; i < 8; ++i) { r[i + 1] += r[i] >> 30; r[i] &= KYB_GCD_M30; }
}

// h = z^-1 (0 for z = 0); z: any limbs fe_to_words accepts (<= 2^31); h tight
KYB_HD void fe_invert_gcd(fe& h, const fe& z) {
  uint32_t w[8];
  fe_to_words(w, z);                                                // canonical value, 8 x 32 bits
  int32_t d[9], e[9], f[9] = KYB_GCD_MOD, g[9];
  g[0] = (int32_t)(w[0] & KYB_GCD_M30);
  KYB_UNROLL for (int i = 1; i < 8; ++i) g[i] = (int32_t)(((w[i - 1] >> (32 - 2 * i)) | (w[i] << (2 * i))) & KYB_GCD_M30);      // bits 30 i .. 30 i + 29
  g[8] = (int32_t)(w[7] >> 16);
  KYB_UNROLL for (int i = 0; i < 9; ++i) { d[i] = 0; e[i] = 0; }
  e[0] = 1;
  int32_t zeta = -1;
#if defined(__HIPCC__)
#pragma unroll 1
#endif
  for (int it = 0; it < 20; ++it) {
    gcd_mat t;
    zeta = gcd_divsteps_30(zeta, (uint32_t)f[0], (uint32_t)g[0], t);
    gcd_update_de(d, e, t);
    gcd_update_fg(f, g, t);
  }
  gcd_normalize(d, f[8]);                                            // g = 0 and f = +-1 now (f = +-p for z = 0, where d stayed 0)
  uint32_t o[8];
  KYB_UNROLL for (int i = 0; i < 8; ++i) o[i] = ((uint32_t)d[i] >> (2 * i)) | ((uint32_t)d[i + 1] << (30 - 2 * i));
  fe_from_words(h, o);
}

// the inversion the kernels use (the exponentiation fe_invert gives the same results: A/B in profiles/r03/ab_invert_gcd.log)
KYB_HD void fe_inv(fe& h, const fe& z) { fe_invert_gcd(h, z); }

}  // namespace kyb
