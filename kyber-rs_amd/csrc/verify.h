// Signature verification with the reference's checks.
//
//   eddsa::verify_with_checks     /root/reference src/sign/eddsa/eddsa_sig.rs:159-212   (flavor 0)
//   schnorr::verify_with_checks   src/sign/schnorr/schnorr_sig.rs:53-110               (flavor 1)
// Both evaluate  s*B == R + h*A  with h = SHA-512(enc R || enc A || msg) mod L after the canonical /
// small-order checks of point.rs:286-337 and scalar.rs:54-75; they differ only in the ORDER of the
// checks, i.e. in which error is reported when several apply.  Status codes (0 = valid):
//   1 InvalidSignatureLength   2 SignatureNotCanonical   3 RNotCanonical   4 R does not decode
//   5 RSmallOrder   6 PublicKeyNotCanonical   7 public key does not decode   8 PublicKeySmallOrder
//   9 InvalidSignature (equation fails)
// The batch is cut into device stages (kernels.hip): A half (checks, decode, hash) -> variable-base h*A;
// R half (checks, decode) and fixed-base s*B, independent of the former; final (status of the checks, one
// addition and a projective comparison: no field inversion anywhere, where the reference's `eq` pays two,
// point.rs:227-241).
#pragma once
#include "ge_scalarmult.h"
#include "sc25519.h"
#include "sha512.h"


namespace kyb {

// scalar.rs:54-75: s < L
KYB_HD uint32_t sc_is_canonical_w(const uint32_t s[8]) {
  const uint32_t lw[8] = KYB_W_L;
  uint32_t t[8];
  return mw_sub<8>(t, s, lw);          // borrow <=> s < L
}
// point.rs:315-337 AS THE REFERENCE EVALUATES IT (the sign bit is ignored).  The reference cites libsodium's
// (0xED - 1 - b0) >> 8 but computes (0xED - (1 - b0)) >> 8 in wrapping u16 arithmetic (overflow-checks = false,
// Cargo.toml:9-10): with bytes 1..30 = 0xff and b31 & 0x7f = 0x7f it answers "not canonical" for every b0 >= 0x14,
// i.e. for y in [2^255 - 236, 2^255 - 1] — the 19 values >= p and also the 217 canonical values p-217 .. p-1.
// A drop-in reproduces that answer (KATs at b0 = 0x13, 0x14, 0xec, 0xed in tests/golden/kats.json).
KYB_HD uint32_t pt_is_canonical_w(const uint32_t w[8]) {
  uint32_t all = w[1] & w[2] & w[3] & w[4] & w[5] & w[6];
  uint32_t rejected = (all == 0xffffffffu) & ((w[7] & 0x7fffffffu) == 0x7fffffffu) & (w[0] >= 0xffffff14u);
  return 1u - rejected;
}
// point.rs:286-313: the canonical re-encoding, sign bit masked, equals one of the five WEAK_KEYS
// (constants.rs:3744-3775) <=> canonical y in {0, 1, p-1, y8a, y8b}
KYB_HD uint32_t pt_has_small_order(const fe& Y) {
  const uint32_t y8a[8] = KYB_W_ORDER8_Y0, y8b[8] = KYB_W_ORDER8_Y1, pw[8] = KYB_W_P;
  uint32_t w[8];
  fe_to_words(w, Y);
  uint32_t d0 = 0, d1 = 0, dm = 0, da = 0, db = 0;
  KYB_UNROLL for (int i = 0; i < 8; ++i) {
    d0 |= w[i];
    d1 |= w[i] ^ (i == 0 ? 1u : 0u);
    dm |= w[i] ^ (i == 0 ? pw[0] - 1u : pw[i]);
    da |= w[i] ^ y8a[i];
    db |= w[i] ^ y8b[i];
  }
  return (d0 == 0) | (d1 == 0) | (dm == 0) | (da == 0) | (db == 0);
}

// Stage 1 is cut in two independent halves (the kernels run them on different streams when the batch is small):
//   A half  s < L, checks and decode of the public key, the challenge h          -> flags_a, h, A
//   R half  checks and decode of R                                               -> flags_r, R
// A failed decode is replaced by the neutral element so that the later stages stay well defined.
//   flags_a: bit 0 s canonical, 1 A canonical, 2 A decodes, 3 A has small order
//   flags_r: bit 0 R canonical, 1 R decodes, 2 R has small order
// (`decode(P, words)`: ge_decode, or the small-batch kernels' cooperative form of it)
template <class Dec>
KYB_HD uint32_t verify_prep_a_with(uint32_t h[8], ge_p3& A, const uint32_t pub[8], const uint32_t sig[16], const uint8_t* msg, uint32_t msg_len, Dec&& decode) {
  const uint32_t s_ok = sc_is_canonical_w(sig + 8);
  const uint32_t a_can = pt_is_canonical_w(pub);
  const uint32_t a_dec = decode(A, pub);
  const uint32_t a_small = pt_has_small_order(A.Y);
  ge_p3 id;
  ge_p3_0(id);
  fe_cmov(A.X, id.X, 1u - a_dec); fe_cmov(A.Y, id.Y, 1u - a_dec); fe_cmov(A.Z, id.Z, 1u - a_dec); fe_cmov(A.T, id.T, 1u - a_dec);
  uint32_t ra[16];
  for (int i = 0; i < 8; ++i) { ra[i] = sig[i]; ra[8 + i] = pub[i]; }
  sha512_ctx c;
  sha512_init(c);
  sha512_words64(c, ra);
  sha512_bytes(c, msg, msg_len);
  uint32_t dig[16];
  sha512_final(dig, c);
  sc_reduce512(h, dig);
  return s_ok | (a_can << 1) | (a_dec << 2) | (a_small << 3);
}
template <class Dec>
KYB_HD uint32_t verify_prep_r_with(ge_p3& R, const uint32_t sig[16], Dec&& decode) {
  const uint32_t r_can = pt_is_canonical_w(sig);
  const uint32_t r_dec = decode(R, sig);
  const uint32_t r_small = pt_has_small_order(R.Y);
  ge_p3 id;
  ge_p3_0(id);
  fe_cmov(R.X, id.X, 1u - r_dec); fe_cmov(R.Y, id.Y, 1u - r_dec); fe_cmov(R.Z, id.Z, 1u - r_dec); fe_cmov(R.T, id.T, 1u - r_dec);
  return r_can | (r_dec << 1) | (r_small << 2);
}
// ---- the public key given as a POINT -----------------------------------------------------------------------------------
// schnorr::verify and eddsa::verify (schnorr_sig.rs:114-127, eddsa_sig.rs:216-...: what every caller in dkg / vss / dss uses) take the
// public key as a Point, marshal it, and hand the bytes to verify_with_checks, which unmarshals them again — a second square root for a point
// the caller already holds.  Given the point and its encoding, the decode is only needed when the limbs are NOT a point of the curve:
// a set of limbs with Z != 0, X Y = Z T and -X^2 + Y^2 = Z^2 + d T^2 is the group element its own encoding decodes to, so the rest of the
// verification (which depends on the group element only) may use it as it is.  Anything else goes through the bytes, as the reference would.
KYB_HD uint32_t ge_on_curve(const ge_p3& P) {
  const fe d = {KYB_FE_D};
  fe xx, yy, zz, tt, lhs, rhs, xy, zt, e1, e2;
  fe_sq(xx, P.X); fe_sq(yy, P.Y); fe_sq(zz, P.Z); fe_sq(tt, P.T);
  fe_sub(lhs, yy, xx);                   // 3T
  fe_mul(rhs, tt, d);
  fe_add(rhs, rhs, zz);                  // 2T
  fe_sub4(e1, lhs, rhs);                 // rhs 2T: 4p bias
  fe_mul(xy, P.X, P.Y);
  fe_mul(zt, P.Z, P.T);
  fe_sub(e2, xy, zt);
  return (1u - fe_is_nonzero(e1)) & (1u - fe_is_nonzero(e2)) & fe_is_nonzero(P.Z);
}
// as verify_prep_a_with; P = the caller's point (tight limbs), pub = its marshal_binary (computed by the engine, ge.rs:112-122)
template <class Dec>
KYB_HD uint32_t verify_prep_a_point_with(uint32_t h[8], ge_p3& A, const ge_p3& P, const uint32_t pub[8], const uint32_t sig[16], const uint8_t* msg, uint32_t msg_len,
                                         Dec&& decode) {
  const uint32_t s_ok = sc_is_canonical_w(sig + 8);
  const uint32_t a_can = pt_is_canonical_w(pub);
  uint32_t a_dec = 1u;
  if (ge_on_curve(P)) {                  // public data: the branch reveals nothing
    A = P;
  } else {
    a_dec = decode(A, pub);
  }
  fe y;
  fe_from_words(y, pub);                 // the y the decode works on (bit 255 ignored)
  const uint32_t a_small = pt_has_small_order(y);
  ge_p3 id;
  ge_p3_0(id);
  fe_cmov(A.X, id.X, 1u - a_dec); fe_cmov(A.Y, id.Y, 1u - a_dec); fe_cmov(A.Z, id.Z, 1u - a_dec); fe_cmov(A.T, id.T, 1u - a_dec);
  uint32_t ra[16];
  for (int i = 0; i < 8; ++i) { ra[i] = sig[i]; ra[8 + i] = pub[i]; }
  sha512_ctx c;
  sha512_init(c);
  sha512_words64(c, ra);
  sha512_bytes(c, msg, msg_len);
  uint32_t dig[16];
  sha512_final(dig, c);
  sc_reduce512(h, dig);
  return s_ok | (a_can << 1) | (a_dec << 2) | (a_small << 3);
}
struct ge_decode_fn { KYB_HD uint32_t operator()(ge_p3& P, const uint32_t w[8]) const { return ge_decode(P, w); } };
KYB_HD uint32_t verify_prep_a(uint32_t h[8], ge_p3& A, const uint32_t pub[8], const uint32_t sig[16], const uint8_t* msg, uint32_t msg_len) {
  return verify_prep_a_with(h, A, pub, sig, msg, msg_len, ge_decode_fn());
}
KYB_HD uint32_t verify_prep_r(ge_p3& R, const uint32_t sig[16]) { return verify_prep_r_with(R, sig, ge_decode_fn()); }
// status of the pre-equation checks: the FIRST failing one in the order of the flavour (evaluated last-to-first)
KYB_HD uint32_t verify_status(uint32_t flags_a, uint32_t flags_r, int flavor) {
  const uint32_t s_ok = flags_a & 1u, a_can = (flags_a >> 1) & 1u, a_dec = (flags_a >> 2) & 1u, a_small = (flags_a >> 3) & 1u;
  const uint32_t r_can = flags_r & 1u, r_dec = (flags_r >> 1) & 1u, r_small = (flags_r >> 2) & 1u;
  uint32_t st = 0;
  if (flavor == 0) {
    st = a_small ? 8u : st;  st = !a_dec ? 7u : st;  st = !a_can ? 6u : st;
    st = (r_dec && r_small) ? 5u : st;  st = !r_dec ? 4u : st;  st = !r_can ? 3u : st;  st = !s_ok ? 2u : st;
  } else {
    st = (a_dec && a_small) ? 8u : st;  st = !a_can ? 6u : st;  st = !a_dec ? 7u : st;
    st = !s_ok ? 2u : st;  st = (r_dec && r_small) ? 5u : st;  st = !r_can ? 3u : st;  st = !r_dec ? 4u : st;
  }
  return st;
}
// both halves and the status in one go (host-compiled check build)
KYB_HD uint32_t verify_prep(uint32_t h[8], ge_p3& R, ge_p3& A, const uint32_t pub[8], const uint32_t sig[16],
                            const uint8_t* msg, uint32_t msg_len, int flavor) {
  const uint32_t fa = verify_prep_a(h, A, pub, sig, msg, msg_len);
  const uint32_t fr = verify_prep_r(R, sig);
  return verify_status(fa, fr, flavor);
}

// Stage 4.  R affine (X, Y, Z = 1), hA and sB projective (X:Y:Z).  Returns 1 iff R + hA == sB.
KYB_HD uint32_t verify_final(const fe& RX, const fe& RY, const ge_p2& hA, const ge_p2& sB) {
  ge_p3 R, H;
  fe_copy(R.X, RX); fe_copy(R.Y, RY); fe_one(R.Z); fe_mul(R.T, RX, RY);
  ge_p2_to_p3(H, hA);
  ge_cached c;
  ge_p3_to_cached(c, H);
  ge_p1p1 t;
  ge_add(t, R, c);
  ge_p2 sum;
  ge_p1p1_to_p2(sum, t);
  fe l, r, d;
  fe_mul(l, sum.X, sB.Z); fe_mul(r, sB.X, sum.Z); fe_sub(d, l, r);
  const uint32_t xne = fe_is_nonzero(d);
  fe_mul(l, sum.Y, sB.Z); fe_mul(r, sB.Y, sum.Z); fe_sub(d, l, r);
  const uint32_t yne = fe_is_nonzero(d);
  return 1u - (xne | yne);
}

}  // namespace kyb
