// Schnorr / EdDSA signing equation for one (x, k, msg) triple (BASELINE config 4).
//
// /root/reference src/sign/schnorr/schnorr_sig.rs:25-47 with the nonce k supplied by the caller
// (the reference draws it from OS entropy, schnorr_sig.rs:31, so its output is not reproducible;
// EdDSA::sign, eddsa_sig.rs:120-152, is the same computation with k = SHA-512(prefix||msg) mod L):
//     R = k*B            (fixed-base mult #1)
//     A = x*B            (fixed-base mult #2; recomputed on every call, schnorr_sig.rs:35)
//     h = SHA-512(enc(R) || enc(A) || msg) as a little-endian integer mod L   (schnorr_sig.rs:128-141)
//     s = k + x*h mod L  (scalar.rs sc_mul + sc_add; x and k may be any 256-bit strings)
//     sig = enc(R) || s
// The two encodings share one field inversion (Montgomery's trick on Z_R * Z_A).
#pragma once
#include "ge_scalarmult.h"
#include "sc25519.h"
#include "sha512.h"

namespace kyb {

template <class Tbl>
KYB_HD void schnorr_sign(uint32_t sig[16], const uint32_t x[8], const uint32_t k[8], const uint8_t* msg,
                         uint32_t msg_len, Tbl& tbl) {
  ge_p3 R, A;
  ge_scalarmult_base(R, k, tbl);
  ge_scalarmult_base(A, x, tbl);
  fe zz, zi, ziR, ziA;
  fe_mul(zz, R.Z, A.Z);
  fe_inv(zi, zz);
  fe_mul(ziR, zi, A.Z);     // 1/Z_R
  fe_mul(ziA, zi, R.Z);     // 1/Z_A
  uint32_t renc[8], aenc[8];
  ge_encode_with_recip(renc, R.X, R.Y, ziR);
  ge_encode_with_recip(aenc, A.X, A.Y, ziA);
  sha512_ctx c;
  sha512_init(c);
  uint32_t ra[16];
  for (int i = 0; i < 8; ++i) { ra[i] = renc[i]; ra[8 + i] = aenc[i]; }
  sha512_words64(c, ra);
  sha512_bytes(c, msg, msg_len);
  uint32_t dig[16], h[8], s[8];
  sha512_final(dig, c);
  sc_reduce512(h, dig);
  sc_muladd(s, x, h, k);
  for (int i = 0; i < 8; ++i) { sig[i] = renc[i]; sig[8 + i] = s[i]; }
}

// EdDSA key expansion and deterministic nonce (curve.rs:74-87, eddsa_sig.rs:120-131):
//   d = SHA-512(seed); x = clamp(d[0..32)) (NOT reduced mod L); prefix = d[32..64)
//   r = SHA-512(prefix || msg) as a little-endian integer mod L
// EdDSA::sign is then schnorr_sign(x, r, msg).
KYB_HD void eddsa_expand_and_nonce(uint32_t x[8], uint32_t r[8], const uint32_t seed[8], const uint8_t* msg, uint32_t msg_len) {
  sha512_ctx c;
  uint32_t d[16];
  sha512_init(c);
  sha512_words32_at0(c, seed);
  sha512_final(d, c);
  for (int i = 0; i < 8; ++i) x[i] = d[i];
  x[0] &= 0xfffffff8u;
  x[7] &= 0x7fffffffu;
  x[7] |= 0x40000000u;
  sha512_init(c);
  sha512_words32_at0(c, d + 8);
  sha512_bytes(c, msg, msg_len);
  uint32_t dig[16];
  sha512_final(dig, c);
  sc_reduce512(r, dig);
}

}  // namespace kyb
