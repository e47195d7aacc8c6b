// The Montgomery ladder of ge_ladder.h with FOUR lanes per item, for batches between the one-item-per-wavefront kernels and ~2^14 items.
//
// ge_ladder_pair.h gives an item two lanes and walks S, M, S, M, M per step: 410 dependent multiply-adds, 0.40 ms whatever the size, with a quarter
// of the SIMDs busy at 8,192 items.  A step's products come in independent FOURS, so here lanes 4i .. 4i+3 of a wavefront share item i and a
// step is three products deep:
//
//     lane 0 (holds x3)          lane 1 (holds z3)          lane 2 (holds x2)          lane 3 (holds z2)
//     c = x3 + z3                d = x3 - z3                a = x2 + z2                b = x2 - z2                 partner by quad_perm [1,0,3,2]
//     (swap ? c : a)^2 = aa      (swap ? d : b)^2 = bb      da = a d                   cb = b c                    operands by quad_perm [2,3,1,0]
//     (da + cb)^2                (da - cb)^2                x2' = aa bb                z2' = e (aa + a24 e)         operands by quad_perm [2,2,0,0] / [3,3,1,1]
//     x3' = W1 (da + cb)^2       z3' = U1 (da - cb)^2       (x2' * 1)                  (z2' * 1)                   e = aa - bb,  u(P) = U1 / W1
//
// i.e. M, M, M per lane (a square is a product of equal operands here: the four lanes run ONE instruction stream): 310 multiply-adds of dependent
// chain per step instead of 410.  As in the two-lane form the base point's image stays projective (no k_mont_prep launch, no inversion in front),
// the code is uniform — what differs between the lanes sits in per-lane REGISTERS (signs, a24 or 0, the third product's constant operand), never
// in control flow — and nothing depends on the scalar except the masked selection of the doubling's operand.  Same group element as mont_ladder,
// hence the same bytes.  Device code only.
#pragma once
#include "ge_ladder_pair.h"

namespace kyb {
constexpr int KYB_QP_CROSS = 0x1E;     // quad_perm [2,3,1,0]
constexpr int KYB_QP_FIRST = 0x0A;     // quad_perm [2,2,0,0]
constexpr int KYB_QP_SECOND = 0x5F;    // quad_perm [3,3,1,1]
constexpr int KYB_QP_OTHER_PAIR = 0x4E;  // quad_perm [2,3,0,1]
constexpr int KYB_QP_L0 = 0x00, KYB_QP_L1 = 0x55, KYB_QP_L2 = 0xAA, KYB_QP_L3 = 0xFF;

// q1 = (x3 | z3 | x2 | z2) on lanes (0 | 1 | 2 | 3) of the item's quad after the ladder over the 256-bit mag
__device__ __forceinline__ void mont_ladder_quad(fe& st, const fe& U1, const fe& W1, const uint32_t mag[8], int skip, uint32_t q) {
  pair_lane s;
  pair_lane_init(s, q & 1u);                           // odd lanes subtract
  const uint32_t hi = q >> 1;                          // the pair that holds (x2, z2)
  const uint32_t is2 = (uint32_t)(q == 2u), is3 = (uint32_t)(q == 3u);
  const uint32_t a24 = is3 ? 121665u : 0u;
  fe one, zero, U, t0;
  fe_one(one); fe_zero(zero);
  fe_select(t0, U1, W1, q & 1u);                       // x3 = U1 | z3 = W1
  fe_select(st, one, zero, q & 1u);                    // x2 = 1 | z2 = 0
  fe_select(st, t0, st, hi);
  fe_select(U, W1, U1, q & 1u);                        // the third product's constant operand: W1 | U1 | 1 | 1
  fe_select(U, U, one, hi);
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
      fe p, sd, X, F, G, r1, Y1, Y2, T, g, r2;
      fe_quad<KYB_QP_PARTNER>(p, st);
      fe_addsub_lane(sd, p, st, s);                    // c = z3 + x3 (2T) | d = x3 - z3 (3T) | a = z2 + x2 (2T) | b = x2 - z2 (3T)
      fe_quad<KYB_QP_CROSS>(X, sd);                    // a | b | d | c
      fe_select(F, X, sd, hi | swap);                  // swap ? c : a | swap ? d : b | a | b          (the swap only matters to the doubling: mont_ladder)
      fe_select(G, F, X, hi);                          // the same again (a square) | d | c
      swap = bit;
      fe_mul(r1, F, G);                                // aa | bb | da | cb                            (<= 3T x 3T)
      fe_quad<KYB_QP_FIRST>(Y1, r1);                   // da | da | aa | aa
      fe_quad<KYB_QP_SECOND>(Y2, r1);                  // cb | cb | bb | bb
      fe_addsub_lane(T, Y1, Y2, s);                    // da + cb (2T) | da - cb (3T) | aa + bb (unused) | e = aa - bb (3T)
      fe_mul_small(g, T, a24);                         // 0 | 0 | 0 | a24 e
      fe_addw(g, g, Y1);                               // . | . | . | aa + a24 e (2T)
      fe_select(F, T, Y1, is2);                        // da + cb | da - cb | aa | e
      fe_select(G, T, g, is3);
      fe_select(G, G, Y2, is2);                        // da + cb | da - cb | bb | aa + a24 e
      fe_mul(r2, F, G);                                // (da + cb)^2 | (da - cb)^2 | x2' = aa bb | z2' = e (aa + a24 e)
      fe_mul_g19<true>(st, r2, U, U19);                // x3' = W1 (da + cb)^2 | z3' = U1 (da - cb)^2 | x2' | z2'
    }
  }
  fe o;
  fe_quad<KYB_QP_OTHER_PAIR>(o, st);
  fe_select(st, st, o, swap);                          // the closing swap exchanges the two pairs
}

// whole multiplication of one item on four lanes, straight from the extended point: projective image (no inversion, mont_prep_proj),
// ladder, recovery; all four lanes return the result
__device__ __forceinline__ void ge_scalarmult_ladder_quad(ge_p2& out, const uint32_t a[8], const ge_p3& P, int skip, uint32_t q) {
  uint32_t neg, mag[8];
  sc_effective(neg, mag, a);
  mont_point_proj m;
  mont_prep_proj(m, P);
  fe st, x2, z2, x3, z3;
  mont_ladder_quad(st, m.U, m.W, mag, skip, q);
  fe_quad<KYB_QP_L0>(x3, st); fe_quad<KYB_QP_L1>(z3, st);
  fe_quad<KYB_QP_L2>(x2, st); fe_quad<KYB_QP_L3>(z2, st);
  mont_recover_to_edwards_proj(out, m, x2, z2, x3, z3, mag[0] & 1u, neg);
}

// ... and from the WIRE encoding (ge_ladder_pair.h, "from the WIRE encoding"): the ladder on (1 + y : 1 - y); all four lanes return the x-only state
__device__ __forceinline__ void mont_ladder_quad_from_y(fe& x2, fe& z2, fe& x3, fe& z3, const uint32_t a[8], const uint32_t enc_words[8], int skip, uint32_t q) {
  uint32_t neg, mag[8];
  sc_effective(neg, mag, a);
  fe y, one, U1, W1, st;
  fe_from_words(y, enc_words);            // bit 255 ignored, y >= p accepted (fe_from_bytes)
  fe_one(one);
  fe_add(U1, one, y);                     // 1 + y
  fe_sub(W1, one, y);                     // 1 - y  (3T)
  fe_reduce_weak(U1, U1);
  fe_reduce_weak(W1, W1);                 // tight: operands of the ladder's last product and its first additions
  mont_ladder_quad(st, U1, W1, mag, skip, q);
  fe_quad<KYB_QP_L0>(x3, st); fe_quad<KYB_QP_L1>(z3, st);
  fe_quad<KYB_QP_L2>(x2, st); fe_quad<KYB_QP_L3>(z2, st);
}

// LG = 1: two lanes per item, LG = 2: four — one spelling for the kernels that exist in both forms (sub = the lane's number within its item)
template <int LG>
__device__ __forceinline__ void ge_scalarmult_ladder_lanes(ge_p2& out, const uint32_t a[8], const ge_p3& P, int skip, uint32_t sub) {
  if constexpr (LG == 1) ge_scalarmult_ladder_pair(out, a, P, skip, sub); else ge_scalarmult_ladder_quad(out, a, P, skip, sub);
}
template <int LG>
__device__ __forceinline__ void mont_ladder_lanes_from_y(fe& x2, fe& z2, fe& x3, fe& z3, const uint32_t a[8], const uint32_t enc_words[8], int skip, uint32_t sub) {
  if constexpr (LG == 1) mont_ladder_pair_from_y(x2, z2, x3, z3, a, enc_words, skip, sub); else mont_ladder_quad_from_y(x2, z2, x3, z3, a, enc_words, skip, sub);
}

}  // namespace kyb
