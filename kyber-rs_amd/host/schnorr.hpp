// C++ mirror of kyber-rs's Schnorr signing / verification over the engine.
//   sign                /root/reference src/sign/schnorr/schnorr_sig.rs:25-47
//   verify_with_checks  src/sign/schnorr/schnorr_sig.rs:53-110, verify :114-126
//   eddsa::verify_with_checks  src/sign/eddsa/eddsa_sig.rs:159-212
//   eddsa::EdDSA (new / from seed / marshal / unmarshal / sign)   src/sign/eddsa/eddsa_sig.rs:17-152, curve.rs:74-99
// Error texts are the reference's (src/sign/error.rs as surfaced by the tests: "signature is not
// canonical", "R is not canonical", "R has small order", "public key is not canonical", "public key has
// small order", "reconstructed S is not equal to signature").
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "edwards25519.hpp"
#include "../csrc/sha512.h"

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
// the public key as the Point the caller holds (schnorr::verify / eddsa::verify): kyb_verify_points_batch marshals it on the GPU and skips
// the square root of unmarshalling it again; same status as verify_one on marshal_binary(pub)
inline int verify_point_one(const group::edwards25519::Point& pub, const uint8_t* msg, size_t n, const uint8_t* sig, size_t sig_len, int flavor) {
  if (sig_len != 64) return 1;
  uint32_t off[2] = {0, (uint32_t)n};
  uint8_t st = 0, dummy = 0;
  group::edwards25519::detail::engine_must(kyb_verify_points_batch(pub.limbs(), n ? msg : &dummy, off, sig, 1, flavor, &st), "verify");
  return st;
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
// schnorr_sig.rs:114-126 (marshal_binary of the key, then verify_with_checks: both on the GPU here)
inline void verify(const Point& pub, const uint8_t* msg, size_t n, const uint8_t* sig, size_t sig_len) {
  detail::throw_status(detail::verify_point_one(pub, msg, n, sig, sig_len, 1), sig_len, false);
}
// many (key, message, signature) triples in one call: status per item, 0 = valid (the codes of throw_status)
inline std::vector<uint8_t> verify_batch(const std::vector<Point>& pubs, const std::vector<std::vector<uint8_t>>& msgs, const std::vector<std::vector<uint8_t>>& sigs) {
  const size_t n = pubs.size();
  std::vector<int32_t> px(40 * n);
  std::vector<uint8_t> blob(1), sg(64 * n), st(n, 1);
  std::vector<uint32_t> off(n + 1, 0);
  for (size_t i = 0; i < n; ++i) {
    std::memcpy(&px[40 * i], pubs[i].limbs(), 160);
    blob.insert(blob.end() - 1, msgs[i].begin(), msgs[i].end());
    off[i + 1] = (uint32_t)(blob.size() - 1);
    if (sigs[i].size() == 64) std::memcpy(&sg[64 * i], sigs[i].data(), 64);
  }
  if (n) group::edwards25519::detail::engine_must(kyb_verify_points_batch(px.data(), blob.data(), off.data(), sg.data(), n, 1, st.data()), "schnorr::verify_batch");
  for (size_t i = 0; i < n; ++i) if (sigs[i].size() != 64) st[i] = 1;
  return st;
}
}  // namespace schnorr

namespace eddsa {
using group::edwards25519::Point;
using group::edwards25519::Scalar;

// eddsa_sig.rs:17-152.  The key object keeps the public key, so sign() costs one fixed-base multiplication
// (kyb_schnorr_sign_keyed_batch); hashing the prefix into the nonce is host work as in the reference.
class EdDSA {
 public:
  Scalar secret;                 // hashed + bit-tweaked, NOT reduced mod L (curve.rs:79-84)
  Point public_key;
  std::vector<uint8_t> seed, prefix;

  EdDSA() { public_key.null(); }                                           // Default (eddsa_sig.rs:45-54)
  // EdDSA::new (eddsa_sig.rs:31-43) with Curve::new_key_and_seed (curve.rs:91-99): 32 bytes from the stream
  explicit EdDSA(Stream& stream) {
    uint8_t buf[32], z[32] = {0};
    stream.xor_key_stream(buf, z, 32);
    from_seed(buf);
  }
  static EdDSA from_seed_bytes(const uint8_t s[32]) { EdDSA e; e.from_seed(s); return e; }
  // From<Pair> (eddsa_sig.rs:107-117): no seed, empty prefix
  static EdDSA from_pair(const Scalar& priv, const Point& pub) { EdDSA e; e.secret = priv; e.public_key = pub; return e; }

  // "seed || Public" (eddsa_sig.rs:94-105)
  std::vector<uint8_t> marshal_binary() const {
    std::vector<uint8_t> out(64, 0), pb = public_key.marshal_binary();
    std::memcpy(out.data(), seed.data(), seed.size() < 32 ? seed.size() : 32);
    std::memcpy(out.data() + 32, pb.data(), 32);
    return out;
  }
  // eddsa_sig.rs:75-91
  void unmarshal_binary(const uint8_t* buff, size_t n) {
    if (n != 64) throw MarshallingError("wrong length for decoding EdDSA private");
    from_seed(buff);
  }
  bool operator==(const EdDSA& o) const { return seed == o.seed && prefix == o.prefix && secret == o.secret && public_key == o.public_key; }

  // eddsa_sig.rs:120-152
  std::vector<uint8_t> sign(const uint8_t* msg, size_t n) const {
    kyb::sha512_ctx c;
    kyb::sha512_init(c);
    if (!prefix.empty()) kyb::sha512_bytes(c, prefix.data(), (uint32_t)prefix.size());
    if (n) kyb::sha512_bytes(c, msg, (uint32_t)n);
    uint32_t dig[16], r[8];
    kyb::sha512_final(dig, c);
    kyb::sc_reduce512(r, dig);
    uint8_t rb[32];
    std::memcpy(rb, r, 32);
    std::vector<uint8_t> pb = public_key.marshal_binary(), sig(64);
    uint32_t off[2] = {0, (uint32_t)n};
    uint8_t dummy = 0;
    group::edwards25519::detail::engine_must(
        kyb_schnorr_sign_keyed_batch(secret.v.data(), pb.data(), rb, n ? msg : &dummy, off, 1, sig.data()), "EdDSA::sign");
    return sig;
  }
  void verify(const uint8_t* msg, size_t n, const uint8_t* sig, size_t sig_len) const;   // eddsa_sig.rs:214-222, below

 private:
  void from_seed(const uint8_t s[32]) {
    kyb::sha512_ctx c;
    kyb::sha512_init(c);
    kyb::sha512_bytes(c, s, 32);
    uint32_t dig[16];
    kyb::sha512_final(dig, c);
    uint8_t d[64];
    std::memcpy(d, dig, 64);
    uint8_t pre[32];
    secret = group::edwards25519::Curve::clamp_digest(d, pre);
    seed.assign(s, s + 32);
    prefix.assign(pre, pre + 32);
    public_key = Point().mul(secret, nullptr);
  }
};

// eddsa_sig.rs:159-212
inline void verify_with_checks(const uint8_t* pub, size_t pub_len, const uint8_t* msg, size_t n, const uint8_t* sig, size_t sig_len) {
  if (pub_len != 32) throw SignatureError(7, "invalid Ed25519 curve point");
  detail::throw_status(detail::verify_one(pub, msg, n, sig, sig_len, 0), sig_len, true);
}
// eddsa_sig.rs:214-222
inline void verify(const Point& pub, const uint8_t* msg, size_t n, const uint8_t* sig, size_t sig_len) {
  std::vector<uint8_t> pb = pub.marshal_binary();
  verify_with_checks(pb.data(), pb.size(), msg, n, sig, sig_len);
}
inline void EdDSA::verify(const uint8_t* msg, size_t n, const uint8_t* sig, size_t sig_len) const {
  std::vector<uint8_t> pb = public_key.marshal_binary();
  verify_with_checks(pb.data(), pb.size(), msg, n, sig, sig_len);
}
}  // namespace eddsa

}  // namespace sign
}  // namespace kyber
