// Table-free variable-base scalar multiplication: Montgomery ladder + y-recovery.
//
// Same function as ge_scalarmult (ge_scalarmult.h) — Point::mul(s, Some(P)), /root/reference
// src/group/edwards25519/ge.rs:508-568, top-digit quirk included — but computed without the per-lane
// table 1P..8P, whose constant-time scan costs 82 KB of L2/Infinity-Cache traffic per multiplication
// (profiles/r01/mul_pmc_summary.json) and 256 VGPRs.  Data flow per item:
//
//   prep   (u, v) = Montgomery image of P:  u = (Z+Y)/(Z-Y),  v = c*u*Z/X,  c = sqrt(-486664)
//          (one field inversion per 8 items, Montgomery's trick — k_mont_prep)
//   ladder 256 steps of the x-only differential addition/doubling (RFC 7748 §5 formulas, 5M + 4S + one
//          multiplication by a24 = 121665 per bit) on |a'|, a' = the integer the reference's
//          recoding actually multiplies by (sc_effective)
//   recover  Okeya-Sakurai y-recovery from x(kP), x((k+1)P) and the affine (u, v) of P; map back to
//          projective Edwards (X:Y:Z); negate when a' < 0
//   finish batched inversion + encode (k_finish)
//
// Exceptional cases, all resolved by selects at the end (no data-dependent branch): P = neutral
// element or the point of order 2 (the only points whose u is 0 or infinity), kP = infinity, kP = -P
// ((k+1)P = infinity), kP = the 2-torsion point.  tools/ladder_proto.py is the big-integer prototype
// of exactly this flow; tests pin it on the quirk vectors (small-order / mixed-order points, scalars
// 0, L, 8L, >= 2^255).  Constant time: the step count is fixed (256) and the swap is a masked select.
#pragma once
#include "ge25519.h"

namespace kyb {

// h = f * k for a small constant k < 2^21; f <= 6T.  10 mads.  Output tight.
KYB_HD void fe_mul_small(fe& h, const fe& f, uint32_t k) {
  uint64_t acc = 0;
  uint32_t r[10];
  KYB_UNROLL for (int i = 0; i < 10; ++i) {
    acc = kyb_mad(f.v[i], k, acc);
    r[i] = (uint32_t)acc & KYB_MASK(i);
    acc >>= KYB_BITS(i);
  }
  uint64_t t = (uint64_t)r[0] + acc * 19u;
  h.v[0] = (uint32_t)t & KYB_MASK(0);
  h.v[1] = r[1] + (uint32_t)(t >> 26);
  KYB_UNROLL for (int i = 2; i < 10; ++i) h.v[i] = r[i];
}

struct mont_point {
  fe u, v;           // affine Montgomery coordinates of P (garbage when flags != 0)
  uint32_t flags;    // bit 0: P is the neutral element; bit 1: P is the point of order 2 (0, -1)
};

// numerators / denominator for the map of one point: d = (Z-Y)*X must be inverted
//   u = (Z+Y) * X * (1/d),   v = c * u * Z * (Z-Y) * (1/d)
// Exceptional inputs (d == 0): X == 0 -> neutral element (Y == Z) or order-2 point (Y == -Z).
KYB_HD void mont_prep_den(fe& d, uint32_t& flags, const ge_p3& P) {
  fe zmy, zpy;
  fe_sub(zmy, P.Z, P.Y);                 // 3T
  fe_add(zpy, P.Z, P.Y);                 // 2T
  fe_mul(d, zmy, P.X);
  const uint32_t x0 = 1u - fe_is_nonzero(P.X);
  const uint32_t id = x0 & (1u - fe_is_nonzero(zmy));
  const uint32_t o2 = x0 & (1u - fe_is_nonzero(zpy));
  // a point with Z - Y == 0 but X != 0 is not on the curve; treat it like the neutral element
  const uint32_t degenerate = 1u - fe_is_nonzero(d);
  flags = (id | (degenerate & (1u - o2))) | (o2 << 1);
  fe one;
  fe_one(one);
  fe_cmov(d, one, degenerate);
}
KYB_HD void mont_prep_finish(mont_point& m, const ge_p3& P, const fe& dinv, uint32_t flags) {
  const fe c = {KYB_FE_MONT_C};
  fe zmy, zpy, t;
  fe_sub(zmy, P.Z, P.Y);
  fe_add(zpy, P.Z, P.Y);
  fe_mul(t, P.X, dinv);
  fe_mul(m.u, t, zpy);                   // g = zpy 2T
  fe_mul(t, zmy, dinv);                  // f = zmy 3T
  fe_mul(t, t, P.Z);
  fe_mul(t, t, m.u);
  fe_mul(m.v, t, c);
  m.flags = flags;
}

// x-only ladder: (x2:z2) = k*P, (x3:z3) = (k+1)*P for the 256-bit k = mag, u1 = u(P) affine.
// skip: leading bits of the 256-bit register known to be zero for EVERY item of the launch on public grounds
// (3 for a scalar reduced mod L, as the challenge h of a verification): the ladder starts below them.
KYB_HD void mont_ladder(fe& x2, fe& z2, fe& x3, fe& z3, const fe& u1, const uint32_t mag[8], int skip = 0) {
  fe_one(x2); fe_zero(z2); fe_copy(x3, u1); fe_one(z3);
  uint32_t swap = 0;
  uint32_t u1_19[10];                    // 19 * u(P): the same nine premultiplies in every one of the 256 steps
  fe_x19(u1_19, u1);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int w = 7; w >= 0; --w) {
    uint32_t word = 0;
    KYB_UNROLL for (int k = 0; k < 8; ++k) word = (w == k) ? mag[k] : word;
    const int first = (w == 7) ? skip : 0;
    word <<= first;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
    for (int j = first; j < 32; ++j) {
      const uint32_t bit = word >> 31;
      word <<= 1;
      swap ^= bit;
      // RFC 7748 swaps (x2,z2) <-> (x3,z3) here.  The swap only exchanges (a,b) with (c,d): da + cb is
      // symmetric under it and da - cb merely changes sign before being squared, so the differential
      // addition does not see it; only the two operands of the doubling need the selection.
      // Bounds: every product below is (<= 3T) x (<= 2T) or tighter, every square of a sum is of <= 2T: the short fold
      // (fe_mul_b6 / fe_sq_b2) applies; the two squares of differences (3T) take the general one.
      fe a, aa, b, bb, e, c, d, da, cb, t, sa, sb;
      fe_addw(a, x2, z2);                 // 2T
      fe_subw(b, x2, z2);                 // 3T
      fe_addw(c, x3, z3);                 // 2T
      fe_subw(d, x3, z3);                 // 3T
      fe_select(sa, a, c, swap);
      fe_select(sb, b, d, swap);
      swap = bit;
      fe_sq_b2(aa, sa);
      fe_sq(bb, sb);
      fe_mul_b6(da, d, a);               // f 3T, g 2T
      fe_mul_b6(cb, b, c);               // f 3T, g 2T
      fe_subw(e, aa, bb);                 // 3T
      fe_addw(t, da, cb);                 // 2T
      fe_sq_b2(x3, t);
      fe_subw(t, da, cb);                 // 3T
      fe_sq(t, t);
      fe_mul_g19<true>(z3, t, u1, u1_19);
      fe_mul_b6(x2, aa, bb);
      fe_mul_small(t, e, 121665u);       // a24 * E
      fe_addw(t, t, aa);                  // 2T
      fe_mul_b6(z2, e, t);               // f 3T, g 2T
    }
  }
  fe_cswap(x2, x3, swap);
  fe_cswap(z2, z3, swap);
}

// Okeya-Sakurai recovery + map back to projective Edwards, with the exceptional cases selected in.
// (Costello & Smith, "Montgomery curves and their arithmetic", Alg. 5, B = 1, A = 486662.)
KYB_HD void mont_recover_to_edwards(ge_p2& out, const mont_point& m, const fe& x2, const fe& z2, const fe& x3, const fe& z3,
                                    uint32_t k_is_odd, uint32_t negate) {
  const fe cc = {KYB_FE_MONT_C};
  fe t1, t2, t3, t4, Yp, U, V, W;
  fe_mul(t1, m.u, z2);
  fe_add(t2, x2, t1);                    // 2T
  fe_sub(t3, x2, t1);                    // 3T
  fe_sq(t3, t3);
  fe_mul(t3, t3, x3);
  fe_mul_small(t1, z2, 2u * 486662u);    // 2A * Z_Q
  fe_add(t2, t2, t1);                    // 3T
  fe_mul(t4, m.u, x2);
  fe_add(t4, t4, z2);                    // 2T
  fe_mul(t2, t2, t4);                    // f 3T, g 2T
  fe_mul(t1, t1, z2);
  fe_sub(t2, t2, t1);                    // 3T
  fe_mul(t2, t2, z3);
  fe_sub(Yp, t2, t3);                    // 3T
  fe_add(t1, m.v, m.v);                  // 2 * v(P)   (2T)
  fe_mul(t1, t1, z2);
  fe_mul(t1, t1, z3);
  fe_mul(U, t1, x2);
  fe_reduce_weak(V, Yp);
  fe_mul(W, t1, z2);
  // Edwards: x = c*u/v, y = (u-1)/(u+1)  ->  X = c*U*(U+W), Y = (U-W)*V, Z = V*(U+W)
  fe upw, umw, X, Y, Z;
  fe_add(upw, U, W);                     // 2T
  fe_sub(umw, U, W);                     // 3T
  fe_mul(t1, U, cc);
  fe_mul(X, t1, upw);
  fe_mul(Y, umw, V);
  fe_mul(Z, V, upw);
  // exceptional results
  const uint32_t z2_zero = 1u - fe_is_nonzero(z2);
  const uint32_t z3_zero = 1u - fe_is_nonzero(z3);
  const uint32_t x2_zero = 1u - fe_is_nonzero(x2);
  const uint32_t res_inf = z2_zero;
  const uint32_t res_negp = z3_zero & (1u - z2_zero);
  const uint32_t res_o2 = x2_zero & (1u - z2_zero);
  const uint32_t p_id = m.flags & 1u, p_o2 = (m.flags >> 1) & 1u;
  // -P in Edwards coordinates from (u, v): x = c*u/v, y = (u-1)/(u+1) -> (X:Y:Z) = (-c*u*(u+1) : (u-1)*v : v*(u+1))
  fe one, zero, mone, up1, um1, nX, nY, nZ;
  fe_one(one); fe_zero(zero); fe_neg(mone, one);
  fe_add(up1, m.u, one);                 // 2T
  fe_sub(um1, m.u, one);                 // 3T
  fe_mul(t1, m.u, cc);
  fe_mul(nX, t1, up1);
  fe_neg(nX, nX);                        // 2T
  fe_mul(nY, um1, m.v);
  fe_mul(nZ, m.v, up1);
  fe_cmov(X, nX, res_negp); fe_cmov(Y, nY, res_negp); fe_cmov(Z, nZ, res_negp);
  fe_cmov(X, zero, res_o2); fe_cmov(Y, mone, res_o2); fe_cmov(Z, one, res_o2);
  fe_cmov(X, zero, res_inf); fe_cmov(Y, one, res_inf); fe_cmov(Z, one, res_inf);
  fe_cmov(X, zero, p_id); fe_cmov(Y, one, p_id); fe_cmov(Z, one, p_id);
  // order-2 input: k*(0,-1) = (0,-1) for odd k, neutral for even k
  fe yo2;
  fe_select(yo2, one, mone, k_is_odd);
  fe_cmov(X, zero, p_o2); fe_cmov(Y, yo2, p_o2); fe_cmov(Z, one, p_o2);
  fe nx;
  fe_reduce_weak(X, X);
  fe_neg(nx, X);
  fe_cmov(X, nx, negate);
  fe_reduce_weak(out.X, X);
  fe_reduce_weak(out.Y, Y);
  fe_reduce_weak(out.Z, Z);
}

// whole multiplication for one item given its prepared Montgomery image
KYB_HD void ge_scalarmult_ladder(ge_p2& out, const uint32_t a[8], const mont_point& m, int skip = 0) {
  uint32_t neg, mag[8];
  sc_effective(neg, mag, a);
  fe x2, z2, x3, z3;
  mont_ladder(x2, z2, x3, z3, m.u, mag, skip);
  mont_recover_to_edwards(out, m, x2, z2, x3, z3, mag[0] & 1u, neg);
}

// ---- projective base point: no inversion before the ladder (small-batch kernel, kernels_coop.hip) --------------------------
// The batch kernel normalises the image (u, v) with one shared inversion per eight points and saves a multiplication in each
// of the 256 steps.  The one-item-per-wavefront kernel has idle rows in its third multiplication level, so there the extra
// product is free and the inversion (a 265-multiplication dependent chain) is what costs: it keeps the image projective,
//     u = U / W,  v = V / W   with   U = (Z+Y) X,  V = c (Z+Y) Z,  W = (Z-Y) X,
// runs the ladder with x3' = W (da+cb)^2, z3' = U (da-cb)^2, and recovers y with every term of the Okeya-Sakurai formulas
// scaled by W^2 (the result is projective anyway).  tools/ladder_proto.py (mul_via_ladder_proj) is the big-integer prototype.
struct mont_point_proj {
  fe U, V, W;        // (1, 1, 1) when flags != 0
  uint32_t flags;    // as mont_point
};
KYB_HD void mont_prep_proj(mont_point_proj& m, const ge_p3& P) {
  const fe c = {KYB_FE_MONT_C};
  fe zmy, zpy, t, one;
  fe_sub(zmy, P.Z, P.Y);                 // 3T
  fe_add(zpy, P.Z, P.Y);                 // 2T
  fe_mul(m.W, zmy, P.X);
  fe_mul(m.U, zpy, P.X);
  fe_mul(t, zpy, P.Z);
  fe_mul(m.V, t, c);
  const uint32_t x0 = 1u - fe_is_nonzero(P.X);
  const uint32_t id = x0 & (1u - fe_is_nonzero(zmy));
  const uint32_t o2 = x0 & (1u - fe_is_nonzero(zpy));
  const uint32_t degenerate = 1u - fe_is_nonzero(m.W);          // X == 0, or Z == Y with X != 0 (not on the curve): neutral element
  m.flags = (id | (degenerate & (1u - o2))) | (o2 << 1);
  fe_one(one);
  fe_cmov(m.U, one, degenerate); fe_cmov(m.V, one, degenerate); fe_cmov(m.W, one, degenerate);
}
// the same ladder with the base point's u as U1 / W1 (one-lane form: reference for the cooperative kernel and the host-test build)
KYB_HD void mont_ladder_proj(fe& x2, fe& z2, fe& x3, fe& z3, const fe& U1, const fe& W1, const uint32_t mag[8], int skip = 0) {
  fe_one(x2); fe_zero(z2); fe_copy(x3, U1); fe_copy(z3, W1);
  uint32_t swap = 0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int i = 255 - skip; i >= 0; --i) {
    uint32_t word = 0;
    KYB_UNROLL for (int k = 0; k < 8; ++k) word = ((i >> 5) == k) ? mag[k] : word;
    const uint32_t bit = (word >> (i & 31)) & 1u;
    swap ^= bit;
    fe a, aa, b, bb, e, c, d, da, cb, t, sa, sb;
    fe_addw(a, x2, z2); fe_subw(b, x2, z2); fe_addw(c, x3, z3); fe_subw(d, x3, z3);
    fe_select(sa, a, c, swap);
    fe_select(sb, b, d, swap);
    swap = bit;
    fe_sq(aa, sa); fe_sq(bb, sb);
    fe_mul(da, d, a); fe_mul(cb, b, c);
    fe_subw(e, aa, bb);
    fe_addw(t, da, cb); fe_sq(t, t); fe_mul(x3, t, W1);
    fe_subw(t, da, cb); fe_sq(t, t); fe_mul(z3, t, U1);
    fe_mul(x2, aa, bb);
    fe_mul_small(t, e, 121665u); fe_addw(t, t, aa); fe_mul(z2, e, t);
  }
  fe_cswap(x2, x3, swap);
  fe_cswap(z2, z3, swap);
}
KYB_HD void mont_recover_to_edwards_proj(ge_p2& out, const mont_point_proj& m, const fe& x2, const fe& z2, const fe& x3, const fe& z3,
                                         uint32_t k_is_odd, uint32_t negate) {
  const fe cc = {KYB_FE_MONT_C};
  fe T1, Wx2, T2, T3, a_, T4, b_, W2, t, YP, TT, U, V, W;
  fe_mul(T1, m.U, z2);
  fe_mul(Wx2, m.W, x2);
  fe_add(T2, Wx2, T1);                   // 2T   W t2
  fe_sub(T3, Wx2, T1);                   // 3T
  fe_sq(T3, T3);
  fe_mul(T3, T3, x3);                    // W^2 t3
  fe_mul_small(a_, z2, 2u * 486662u);    // 2A Z_Q
  fe_mul(t, m.W, a_);
  fe_add(T2, T2, t);                     // 3T   W (t2 + 2A z2)
  fe_mul(T4, m.U, x2);
  fe_mul(t, m.W, z2);
  fe_add(T4, T4, t);                     // 2T   W t4
  fe_mul(b_, a_, z2);
  fe_sq(W2, m.W);
  fe_mul(T2, T2, T4);                    // f 3T, g 2T
  fe_mul(t, W2, b_);
  fe_sub(T2, T2, t);                     // 3T
  fe_mul(T2, T2, z3);
  fe_sub(YP, T2, T3);                    // 3T   W^2 Yp
  fe_add(TT, m.V, m.V);                  // 2T
  fe_mul(TT, TT, z2);
  fe_mul(TT, TT, z3);                    // W t1
  fe_mul(t, m.W, TT);
  fe_mul(U, t, x2);
  fe_reduce_weak(V, YP);
  fe_mul(W, t, z2);
  // Edwards: x = c*u/v, y = (u-1)/(u+1)  ->  X = c*U*(U+W), Y = (U-W)*V, Z = V*(U+W)
  fe upw, umw, X, Y, Z, t1;
  fe_add(upw, U, W);                     // 2T
  fe_sub(umw, U, W);                     // 3T
  fe_mul(t1, U, cc);
  fe_mul(X, t1, upw);
  fe_mul(Y, umw, V);
  fe_mul(Z, V, upw);
  const uint32_t z2_zero = 1u - fe_is_nonzero(z2);
  const uint32_t z3_zero = 1u - fe_is_nonzero(z3);
  const uint32_t x2_zero = 1u - fe_is_nonzero(x2);
  const uint32_t res_inf = z2_zero;
  const uint32_t res_negp = z3_zero & (1u - z2_zero);
  const uint32_t res_o2 = x2_zero & (1u - z2_zero);
  const uint32_t p_id = m.flags & 1u, p_o2 = (m.flags >> 1) & 1u;
  // -P from the projective image: (X:Y:Z) = (-c*U1*(U1+W1) : (U1-W1)*V1 : V1*(U1+W1))
  fe one, zero, mone, up1, um1, nX, nY, nZ;
  fe_one(one); fe_zero(zero); fe_neg(mone, one);
  fe_add(up1, m.U, m.W);                 // 2T
  fe_sub(um1, m.U, m.W);                 // 3T
  fe_mul(t1, m.U, cc);
  fe_mul(nX, t1, up1);
  fe_neg(nX, nX);                        // 2T
  fe_mul(nY, um1, m.V);
  fe_mul(nZ, m.V, up1);
  fe_cmov(X, nX, res_negp); fe_cmov(Y, nY, res_negp); fe_cmov(Z, nZ, res_negp);
  fe_cmov(X, zero, res_o2); fe_cmov(Y, mone, res_o2); fe_cmov(Z, one, res_o2);
  fe_cmov(X, zero, res_inf); fe_cmov(Y, one, res_inf); fe_cmov(Z, one, res_inf);
  fe_cmov(X, zero, p_id); fe_cmov(Y, one, p_id); fe_cmov(Z, one, p_id);
  fe yo2;
  fe_select(yo2, one, mone, k_is_odd);
  fe_cmov(X, zero, p_o2); fe_cmov(Y, yo2, p_o2); fe_cmov(Z, one, p_o2);
  fe nx;
  fe_reduce_weak(X, X);
  fe_neg(nx, X);
  fe_cmov(X, nx, negate);
  fe_reduce_weak(out.X, X);
  fe_reduce_weak(out.Y, Y);
  fe_reduce_weak(out.Z, Z);
}
KYB_HD void ge_scalarmult_ladder_proj(ge_p2& out, const uint32_t a[8], const ge_p3& P, int skip = 0) {
  uint32_t neg, mag[8];
  sc_effective(neg, mag, a);
  mont_point_proj m;
  mont_prep_proj(m, P);
  fe x2, z2, x3, z3;
  mont_ladder_proj(x2, z2, x3, z3, m.U, m.W, mag, skip);
  mont_recover_to_edwards_proj(out, m, x2, z2, x3, z3, mag[0] & 1u, neg);
}


}  // namespace kyb
