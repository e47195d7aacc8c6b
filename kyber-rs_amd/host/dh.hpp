// C++ mirror of the Diffie-Hellman step of kyber-rs that reaches the engine.
//   Dh::dh_exchange   /root/reference src/dh/dh_impl.rs:74-80        suite.point().mul(own_private, Some(remote_public))
// Callers (vss/pedersen/vss.rs:371-375, 651-658; encrypt/ecies/ecies_impl.rs:23-26, 69) marshal the shared
// point and feed the 32 bytes to HKDF + AES-GCM; those stay on the host, in the Rust layer.  A dealer that
// encrypts n deals performs n exchanges with the same private key: dh_exchange_batch is that loop as one
// variable-base batch that takes the remote keys as they arrive on the wire (32-byte encodings, decoded on the
// GPU) and returns the marshalled shared points, i.e. the HKDF inputs.
#pragma once
#include <vector>

#include "edwards25519.hpp"

namespace kyber {
namespace dh {

using group::edwards25519::Point;
using group::edwards25519::Scalar;

inline Point dh_exchange(const Scalar& own_private, const Point& remote_public) {
  return Point().mul(own_private, &remote_public);
}

// pre-shared keys of one private key with n remote public keys (encodings); throws like
// Point::unmarshal_binary when one of them is not a curve point
inline std::vector<std::vector<uint8_t>> dh_exchange_batch(const std::vector<Scalar>& own_private,
                                                           const std::vector<std::vector<uint8_t>>& remote_public_enc) {
  const size_t n = remote_public_enc.size();
  if (own_private.size() != n && own_private.size() != 1) throw std::invalid_argument("dh_exchange_batch: size mismatch");
  std::vector<uint8_t> sc(32 * n), pe(32 * n), out(32 * n), ok(n);
  for (size_t i = 0; i < n; ++i) {
    if (remote_public_enc[i].size() != 32) throw MarshallingError("invalid Ed25519 curve point");
    std::memcpy(&sc[32 * i], own_private[own_private.size() == 1 ? 0 : i].v.data(), 32);
    std::memcpy(&pe[32 * i], remote_public_enc[i].data(), 32);
  }
  group::edwards25519::detail::engine_must(kyb_mul_batch(sc.data(), pe.data(), nullptr, n, out.data(), nullptr, ok.data()), "dh_exchange_batch");
  std::vector<std::vector<uint8_t>> r(n);
  for (size_t i = 0; i < n; ++i) {
    if (!ok[i]) throw MarshallingError("invalid Ed25519 curve point");
    r[i].assign(out.begin() + 32 * i, out.begin() + 32 * (i + 1));
  }
  return r;
}

}  // namespace dh
}  // namespace kyber
