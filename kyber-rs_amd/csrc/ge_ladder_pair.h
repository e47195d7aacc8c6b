// The Montgomery ladder of ge_ladder.h with TWO lanes per item, for batches that do not fill the chip.
//
// Between the one-item-per-wavefront kernels (up to ~6,000 items) and ~10^5 items a variable-base batch sits on the latency of ONE
// lane walking 255 ladder steps of 739 multiply-adds each (0.8 ms whatever the size): there are more SIMDs than wavefronts.  A step's
// products come in independent pairs, so lanes 2i and 2i+1 of a wavefront share item i:
//
//     even lane (holds x2, x3)                     odd lane (holds z2, z3)
//     a = x2 + z2, c = x3 + z3                     b = x2 - z2, d = x3 - z3            operands exchanged by DPP quad_perm [1,0,3,2]
//     aa = (swap ? c : a)^2                        bb = (swap ? d : b)^2
//     da = a * d                                   cb = b * c
//     x3' = (da + cb)^2                            tt  = (da - cb)^2
//     x2' = aa * bb                                z2' = e * (aa + a24 e),  e = aa - bb
//     x3' * W1                                     z3' = tt * U1,  u(P) = U1 / W1
//
// i.e. S, M, S, M, M per lane: 410 multiply-adds of dependent chain per step instead of 739.  The even lane's last product is what lets
// the base point stay PROJECTIVE for free (x3' * W1, mont_ladder_proj): there is no k_mont_prep launch and no field inversion in front
// of this kernel, it reads the extended points themselves.  The code is uniform: what differs between the two lanes sits in per-lane
// REGISTERS — the sign of the additions (fe_addsub_lane), a24 or 0, U1 or W1 — never in control flow, and nothing depends on the
// scalar except the masked selection of the doubling's operand, exactly as in mont_ladder.  Same group element as mont_ladder (tools/ladder_proto.py, mul_via_ladder_proj, is the big-integer model of
// the projective form); the encodings and the affine limbs k_finish makes of it are identical.  Device code only.
#pragma once
#include "ge_ladder.h"

namespace kyb {
template <int CTRL>
__device__ __forceinline__ uint32_t quad_dpp(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }
constexpr int KYB_QP_PARTNER = 0xB1;   // quad_perm [1,0,3,2]
constexpr int KYB_QP_EVEN = 0xA0;      // quad_perm [0,0,2,2]
constexpr int KYB_QP_ODD = 0xF5;       // quad_perm [1,1,3,3]

template <int CTRL>
__device__ __forceinline__ void fe_quad(fe& h, const fe& f) {
  KYB_UNROLL for (int i = 0; i < 10; ++i) h.v[i] = quad_dpp<CTRL>(f.v[i]);
}

// per-lane constants of the pair: even lane adds, odd lane subtracts
struct pair_lane {
  uint32_t odd;       // 0 / 1
  uint32_t mask;      // 0 / 0xffffffff
  uint32_t k[10];     // 0 / 2p_i + 1:   (g ^ mask) + k = g on the even lane, 2p_i - g on the odd lane
  uint32_t a24;       // 0 / 121665
};
__device__ __forceinline__ void pair_lane_init(pair_lane& s, uint32_t odd) {
  const uint32_t p2[10] = KYB_FE_2P;
  s.odd = odd;
  s.mask = 0u - odd;
  KYB_UNROLL for (int i = 0; i < 10; ++i) s.k[i] = (p2[i] + 1u) & s.mask;
  s.a24 = 121665u & s.mask;
}
// h = f + g (even lane), f - g = f + 2p - g (odd lane); g <= 1.99T; bound(h) = bound(f) + 2T
__device__ __forceinline__ void fe_addsub_lane(fe& h, const fe& f, const fe& g, const pair_lane& s) {
  KYB_UNROLL for (int i = 0; i < 10; ++i) h.v[i] = f.v[i] + ((g.v[i] ^ s.mask) + s.k[i]);
}

// q1 = (x2 | z2), q2 = (x3 | z3) on the (even | odd) lane after the ladder over the 256-bit mag.  The base point's u is U1 / W1 (both lanes hold
// both; W1 = 1 for an affine image): x3' = W1 (da + cb)^2 on the even lane, z3' = U1 (da - cb)^2 on the odd one — the projective form of
// mont_ladder_proj costs nothing here, the even lane's last product would otherwise multiply by one.
__device__ __forceinline__ void mont_ladder_pair(fe& q1, fe& q2, const fe& U1, const fe& W1, const uint32_t mag[8], int skip, const pair_lane& s) {
  fe one, zero, U;
  fe_one(one); fe_zero(zero);
  fe_select(q1, one, zero, s.odd);       // x2 = 1 | z2 = 0
  fe_select(q2, U1, W1, s.odd);          // x3 = U1 | z3 = W1
  fe_select(U, W1, U1, s.odd);           // the last product's second operand: W1 | U1
  uint32_t U19[10];
  fe_x19(U19, U);
  uint32_t swap = 0;
#pragma unroll 1
  for (int w = 7; w >= 0; --w) {
    uint32_t word = 0;
    KYB_UNROLL for (int k = 0; k < 8; ++k) word = (w == k) ? mag[k] : word;
    const int first = (w == 7) ? skip : 0;
    word <<= first;
#pragma unroll 1
    for (int j = first; j < 32; ++j) {
      const uint32_t bit = word >> 31;
      word <<= 1;
      swap ^= bit;
      fe p, ab, cd, sel, r1, r2, p1, p2, t, r3, e, g, a;
      fe_quad<KYB_QP_PARTNER>(p, q1);
      fe_addsub_lane(ab, p, q1, s);        // a = z2 + x2 (2T) | b = x2 - z2 (3T)
      fe_quad<KYB_QP_PARTNER>(p, q2);
      fe_addsub_lane(cd, p, q2, s);        // c = z3 + x3 (2T) | d = x3 - z3 (3T)
      fe_select(sel, ab, cd, swap);        // the swap only matters to the doubling (mont_ladder)
      swap = bit;
      fe_sq(r1, sel);                      // aa | bb
      fe_quad<KYB_QP_PARTNER>(p, cd);      // d | c
      fe_mul_b6(r2, ab, p);                // da = a d | cb = b c          (2T x 3T | 3T x 2T)
      fe_quad<KYB_QP_PARTNER>(p1, r1);     // bb | aa
      fe_quad<KYB_QP_PARTNER>(p2, r2);     // cb | da
      fe_addsub_lane(t, p2, r2, s);        // cb + da (2T) | da - cb (3T)
      fe_sq(r3, t);                        // (da + cb)^2 | (da - cb)^2
      fe_subw(e, p1, r1);                  // (-e) | e = aa - bb (3T)
      fe_mul_small(g, e, s.a24);           // 0 | a24 e
      fe_addw(g, g, p1);                   // bb | aa + a24 e (2T)
      fe_select(a, r1, e, s.odd);          // aa | e
      fe_mul_b6(q1, a, g);                 // x2' = aa bb | z2' = e (aa + a24 e)      (1T x 1T | 3T x 2T)
      fe_mul_g19<true>(q2, r3, U, U19);    // x3' = W1 (da + cb)^2 | z3' = U1 (da - cb)^2
    }
  }
  fe_cswap(q1, q2, swap);
}

// whole multiplication of one item on a pair of lanes, straight from the extended point: projective image (no inversion, mont_prep_proj),
// ladder, recovery; both lanes return the result
__device__ __forceinline__ void ge_scalarmult_ladder_pair(ge_p2& out, const uint32_t a[8], const ge_p3& P, int skip, uint32_t odd) {
  uint32_t neg, mag[8];
  sc_effective(neg, mag, a);
  pair_lane s;
  pair_lane_init(s, odd);
  mont_point_proj m;
  mont_prep_proj(m, P);
  fe q1, q2, x2, z2, x3, z3;
  mont_ladder_pair(q1, q2, m.U, m.W, mag, skip, s);
  fe_quad<KYB_QP_EVEN>(x2, q1); fe_quad<KYB_QP_ODD>(z2, q1);
  fe_quad<KYB_QP_EVEN>(x3, q2); fe_quad<KYB_QP_ODD>(z3, q2);
  mont_recover_to_edwards_proj(out, m, x2, z2, x3, z3, mag[0] & 1u, neg);
}

// ---- the same from the WIRE encoding, the square root taken out of the way (round 4) ---------------------------------------------------------
// A multiplication from 32 bytes (Diffie-Hellman: dh_impl.rs:74-80 on a key as it arrived) starts with unmarshal_binary, whose square root
// z^((p-5)/8) is a chain of 252 squarings — a fifth of a DKG-sized call.  The ladder does not need it: u(P) = (1 + y) / (1 - y) comes from the
// y of the encoding alone, and the x-only state the ladder leaves does not depend on a common factor of (U1, W1) — (x2 : z2) is built by the
// doubling alone, (x3 : z3) is scaled by a power of that factor as a whole, and the recovery formulas are homogeneous in (x3, z3).  So the
// ladder runs on (1 + y : 1 - y) while ANOTHER kernel on a side stream decodes the point; a third, short kernel builds the full projective
// image (mont_prep_proj on the decoded point) and recovers y(kP) from the stored state.  The point is the same, hence its bytes.
// Encodings that decode to no point, and the points with x = 0 (u = 0 or infinity), ride the ladder with whatever their y gives; the recovery
// replaces the result by what the flags of the decoded point say, exactly as the one-kernel form does.
__device__ __forceinline__ void mont_ladder_pair_from_y(fe& x2, fe& z2, fe& x3, fe& z3, const uint32_t a[8], const uint32_t enc_words[8], int skip, uint32_t odd) {
  uint32_t neg, mag[8];
  sc_effective(neg, mag, a);
  pair_lane s;
  pair_lane_init(s, odd);
  fe y, one, U1, W1, q1, q2;
  fe_from_words(y, enc_words);            // bit 255 ignored, y >= p accepted (fe_from_bytes)
  fe_one(one);
  fe_add(U1, one, y);                     // 1 + y
  fe_sub(W1, one, y);                     // 1 - y  (3T)
  fe_reduce_weak(U1, U1);
  fe_reduce_weak(W1, W1);                 // tight: operands of the ladder's last product and its first additions
  mont_ladder_pair(q1, q2, U1, W1, mag, skip, s);
  fe_quad<KYB_QP_EVEN>(x2, q1); fe_quad<KYB_QP_ODD>(z2, q1);
  fe_quad<KYB_QP_EVEN>(x3, q2); fe_quad<KYB_QP_ODD>(z3, q2);
}
// ... and the end of it, one lane per item: the decoded point (any representation), the state the ladder left
__device__ __forceinline__ void ge_recover_from_state(ge_p2& out, const uint32_t a[8], const ge_p3& P, const fe& x2, const fe& z2, const fe& x3, const fe& z3) {
  uint32_t neg, mag[8];
  sc_effective(neg, mag, a);
  mont_point_proj m;
  mont_prep_proj(m, P);
  mont_recover_to_edwards_proj(out, m, x2, z2, x3, z3, mag[0] & 1u, neg);
}

}  // namespace kyb
