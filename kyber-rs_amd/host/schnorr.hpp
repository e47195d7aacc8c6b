// C++ mirror of kyber-rs's Schnorr signing / verification over the engine.
//   sign                /root/reference src/sign/schnorr/schnorr_sig.rs:25-47
//   verify_with_checks  src/sign/schnorr/schnorr_sig.rs:53-110, verify :114-126
//   eddsa::verify_with_checks  src/sign/eddsa/eddsa_sig.rs:159-212
// Error texts are the reference's (src/sign/error.rs as surfaced by the tests: "signature is not
// canonical", "R is not canonical", "R has small order", "public key is not canonical", "public key has
// small order", "reconstructed S is not equal to signature").
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "edwards25519.hpp"

namespace kyber {
namespace sign {

struct SignatureError : std::runtime_error {
  int code;
  SignatureError(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

namespace detail {
inline void throw_status(int st, size_t sig_len, bool eddsa) {
  switch (st) {
    case 0: return;
    case 1: throw SignatureError(1, eddsa ? "expect 64 got " + std::to_string(sig_len)
                                          : "schnorr: signature of invalid length " + std::to_string(sig_len) + " instead of 64");
    case 2: throw SignatureError(2, "signature is not canonical");
    case 3: throw SignatureError(3, "R is not canonical");
    case 4: case 7: throw SignatureError(st, "invalid Ed25519 curve point");
    case 5: throw SignatureError(5, "R has small order");
    case 6: throw SignatureError(6, "public key is not canonical");
    case 8: throw SignatureError(8, "public key has small order");
    default: throw SignatureError(9, "reconstructed S is not equal to signature");
  }
}
inline int verify_one(const uint8_t pub[32], const uint8_t* msg, size_t n, const uint8_t* sig, size_t sig_len, int flavor) {
  if (sig_len != 64) return 1;
  uint32_t off[2] = {0, (uint32_t)n};
  uint8_t st = 0, dummy = 0;
  group::edwards25519::detail::engine_must(kyb_verify_batch(pub, n ? msg : &dummy, off, sig, 1, flavor, &st), "verify");
  return st;
}
}  // namespace detail

namespace schnorr {
using group::edwards25519::Point;
using group::edwards25519::Scalar;

// schnorr_sig.rs:25-47: k = pick(random), R = k*B, h = H(R || A || msg), s = k + x*h, out = R || s
inline std::vector<uint8_t> sign(Stream& random, const Scalar& priv, const uint8_t* msg, size_t n) {
  Scalar k = Scalar().pick(random);
  uint32_t off[2] = {0, (uint32_t)n};
  uint8_t dummy = 0;
  std::vector<uint8_t> sig(64);
  group::edwards25519::detail::engine_must(kyb_schnorr_sign_batch(priv.v.data(), k.v.data(), n ? msg : &dummy, off, 1, sig.data()), "schnorr::sign");
  return sig;
}
// schnorr_sig.rs:53-110
inline void verify_with_checks(const uint8_t* pub, size_t pub_len, const uint8_t* msg, size_t n, const uint8_t* sig, size_t sig_len) {
  if (pub_len != 32) throw SignatureError(7, "invalid Ed25519 curve point");
  detail::throw_status(detail::verify_one(pub, msg, n, sig, sig_len, 1), sig_len, false);
}
// schnorr_sig.rs:114-126
inline void verify(const Point& pub, const uint8_t* msg, size_t n, const uint8_t* sig, size_t sig_len) {
  std::vector<uint8_t> pb = pub.marshal_binary();
  verify_with_checks(pb.data(), pb.size(), msg, n, sig, sig_len);
}
}  // namespace schnorr

namespace eddsa {
// eddsa_sig.rs:159-212
inline void verify_with_checks(const uint8_t* pub, size_t pub_len, const uint8_t* msg, size_t n, const uint8_t* sig, size_t sig_len) {
  if (pub_len != 32) throw SignatureError(7, "invalid Ed25519 curve point");
  detail::throw_status(detail::verify_one(pub, msg, n, sig, sig_len, 0), sig_len, true);
}
}  // namespace eddsa

}  // namespace sign
}  // namespace kyber
