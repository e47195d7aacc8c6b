// C++ mirror of kyber-rs's Ed25519 `Scalar` / `Point` / `Curve` surface over the C ABI.
//
// The reference is compiled Rust (no toolchain for it in this image), so the host side above the C ABI
// is written in C++ with the reference's names, argument meaning and error behaviour:
//   trait Scalar  /root/reference src/group.rs:22-69   impl: src/group/edwards25519/scalar.rs:144-222
//   trait Point   src/group.rs:85-140                  impl: src/group/edwards25519/point.rs:75-225
//   Marshaling    src/encoding/encodings.rs:12-27      point.rs:35-51, scalar.rs:91-112
//   canonical / small-order checks  point.rs:286-337, scalar.rs:54-75
//   Curve::new_key_and_seed_with_input  curve.rs:74-87
// The trait is per-element and synchronous and a batch-of-1 engine call costs more than
// the CPU needs for the operation, so BY DEFAULT mul / add / sub / neg RECORD their operation in the
// engine's arena (kyb_defer_*, include/kyber_ed25519.h "deferred points") and a Point holds a handle until
// somebody needs its bytes or limbs — marshal_binary, ==, data, hex, the batch helpers — which is
// when the engine evaluates the recorded graph in batches (one call for the t multiplications of
// PriPoly::commit, one for the whole Horner chain of PubPoly::eval, ...).  Handles stay valid for as long
// as protocol state holds them (the arena keeps the values of evaluated nodes; "Lifetime" in the header).
// set_deferred(false) per thread / KYBER_HIP_EAGER=1 in the environment: every Point operation that does
// curve arithmetic is a batch-of-1 call into the GPU engine instead.  Protocol code written against the
// trait, call by call, runs unchanged in either mode and yields the same bytes; throughput callers use
// the *_batch statics.
// Scalar arithmetic stays on the host (microseconds) and shares sc25519.h with
// the device sign kernel.  `mul`/`add`/... are infallible in the reference: an engine failure aborts.
#pragma once
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/kyber_ed25519.h"
#include "../csrc/sc25519.h"

namespace kyber {

// src/cipher/stream.rs:6-24
struct Stream {
  virtual ~Stream() = default;
  // dst = src XOR keystream (src and dst have equal length)
  virtual void xor_key_stream(uint8_t* dst, const uint8_t* src, size_t n) = 0;
};

// src/encoding/encodings.rs MarshallingError::InvalidInput
struct MarshallingError : std::runtime_error {
  using std::runtime_error::runtime_error;
};
// src/group.rs PointError::EmbedDataLength
struct PointError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

namespace group {
namespace edwards25519 {

namespace detail {
// (on unless KYBER_HIP_EAGER is set in the environment: the mode a maintainer gets by default is the fast one)
inline bool& deferred_flag() { static thread_local bool on = std::getenv("KYBER_HIP_EAGER") == nullptr; return on; }
inline void engine_must(int rc, const char* what) {
  if (rc != KYB_OK) {  // the trait has no error channel (group.rs:139): abort like a Rust panic
    std::fprintf(stderr, "kyber-ed25519-hip: %s failed (%d): %s\n", what, rc, kyb_last_error());
    std::abort();
  }
}
inline void words(uint32_t w[8], const uint8_t b[32]) { std::memcpy(w, b, 32); }
inline void bytes(uint8_t b[32], const uint32_t w[8]) { std::memcpy(b, w, 32); }
static const uint8_t L_BYTES[32] = {0xed, 0xd3, 0xf5, 0x5c, 0x1a, 0x63, 0x12, 0x58, 0xd6, 0x9c, 0xf7, 0xa2, 0xde, 0xf9, 0xde, 0x14,
                                    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0x10};
}  // namespace detail

// deferred evaluation of this thread's Point operations — ON by default; set_deferred(false) / KYBER_HIP_EAGER=1: every trait call is its own
// engine call (three times SLOWER than a CPU core on protocol code: bench.py protocol_phases).  See the header of this file.
inline void set_deferred(bool on) { detail::deferred_flag() = on; }
inline bool deferred() { return detail::deferred_flag(); }

// ------------------------------------------------------------------------------------------ Scalar
class Scalar {
 public:
  std::array<uint8_t, 32> v{};  // little-endian, scalar.rs:23-26

  Scalar set(const Scalar& a) { v = a.v; return *this; }                       // scalar.rs:146-149
  Scalar zero() { v.fill(0); return *this; }
  Scalar one() { v.fill(0); v[0] = 1; return *this; }
  // scalar.rs:152-154: Int::new_int64(v, L) -> value mod L in [0, L)
  Scalar set_int64(int64_t x) {
    uint32_t a[8] = {0}, r[8];
    uint64_t mag = x < 0 ? (uint64_t)(-(x + 1)) + 1 : (uint64_t)x;
    a[0] = (uint32_t)mag; a[1] = (uint32_t)(mag >> 32);
    if (x < 0) { uint32_t z[8] = {0}, lw[8] = KYB_W_L; kyb::mw_sub<8>(r, lw, a); (void)z; kyb::sc_reduce256(r, r); }
    else kyb::sc_reduce256(r, a);
    detail::bytes(v.data(), r);
    return *this;
  }
  // scalar.rs:175-177: little-endian bytes (any length up to 64) mod L
  Scalar set_bytes(const uint8_t* b, size_t n) {
    if (n > 64) throw std::invalid_argument("Scalar::set_bytes: more than 64 bytes are not supported by this mirror");
    uint8_t buf[64] = {0};
    std::memcpy(buf, b, n);
    uint32_t x[16], r[8];
    std::memcpy(x, buf, 64);
    kyb::sc_reduce512(r, x);
    detail::bytes(v.data(), r);
    return *this;
  }
  Scalar set_bytes(const std::vector<uint8_t>& b) { return set_bytes(b.data(), b.size()); }
  // scalar.rs:167-173 -> util/random random_int(L): rejection sampling on 253-bit strings
  Scalar pick(Stream& rand) {
    for (;;) {
      uint8_t b[32], z[32] = {0};
      rand.xor_key_stream(b, z, 32);
      // random_stream.rs:36-46 draws bit_len(L) = 253 bits big-endian and retries while >= L
      uint8_t le[32];
      for (int i = 0; i < 32; ++i) le[i] = b[31 - i];
      le[31] &= 0x1f;
      uint32_t w[8], lw[8] = KYB_W_L, t[8];
      detail::words(w, le);
      bool nonzero = false;
      for (int i = 0; i < 8; ++i) nonzero |= w[i] != 0;
      if (nonzero && kyb::mw_sub<8>(t, w, lw) == 1) { std::memcpy(v.data(), le, 32); return *this; }
    }
  }
  Scalar sub(const Scalar& a, const Scalar& b) {                              // sc_sub, scalar.rs:1187
    uint32_t wb[8], nb[8], lw[8] = KYB_W_L, one[8] = {1, 0, 0, 0, 0, 0, 0, 0}, wa[8], r[8];
    detail::words(wb, b.v.data()); detail::words(wa, a.v.data());
    kyb::sc_reduce256(nb, wb);
    kyb::mw_sub<8>(nb, lw, nb);                                               // L - (b mod L) in (0, L]
    kyb::sc_muladd(r, nb, one, wa);
    detail::bytes(v.data(), r);
    return *this;
  }
  Scalar neg(const Scalar& a) { Scalar z; return sub(z, a); }                 // scalar.rs:216-221
  Scalar inv(const Scalar& a) {                                               // scalar.rs:191-214: a^(L-2)
    uint32_t e[8] = KYB_W_L, acc[8] = {1, 0, 0, 0, 0, 0, 0, 0}, base[8], zero[8] = {0};
    e[0] -= 2;
    detail::words(base, a.v.data());
    for (int i = 255; i >= 0; --i) {
      kyb::sc_muladd(acc, acc, acc, zero);
      if ((e[i >> 5] >> (i & 31)) & 1) kyb::sc_muladd(acc, acc, base, zero);
    }
    detail::bytes(v.data(), acc);
    return *this;
  }
  Scalar div(const Scalar& a, const Scalar& b) { Scalar i; i.inv(b); *this = a * i; return *this; }  // scalar.rs:183-189
  friend Scalar operator*(const Scalar& a, const Scalar& b) {                  // sc_mul, scalar.rs:132-136
    uint32_t wa[8], wb[8], zero[8] = {0}, r[8];
    detail::words(wa, a.v.data()); detail::words(wb, b.v.data());
    kyb::sc_muladd(r, wa, wb, zero);
    Scalar s; detail::bytes(s.v.data(), r); return s;
  }
  friend Scalar operator+(const Scalar& a, const Scalar& b) {                  // sc_add, scalar.rs:138-142
    uint32_t wa[8], wb[8], one[8] = {1, 0, 0, 0, 0, 0, 0, 0}, r[8];
    detail::words(wa, a.v.data()); detail::words(wb, b.v.data());
    kyb::sc_muladd(r, wa, one, wb);
    Scalar s; detail::bytes(s.v.data(), r); return s;
  }
  bool operator==(const Scalar& o) const { return v == o.v; }                  // byte equality, scalar.rs:78-83
  bool operator!=(const Scalar& o) const { return !(*this == o); }
  // Marshaling: marshal_binary reduces mod L (scalar.rs:91-100); unmarshal is a raw copy, length checked only
  std::vector<uint8_t> marshal_binary() const {
    uint32_t w[8], r[8];
    detail::words(w, v.data());
    kyb::sc_reduce256(r, w);
    std::vector<uint8_t> out(32);
    detail::bytes(out.data(), r);
    return out;
  }
  void unmarshal_binary(const uint8_t* data, size_t n) {
    if (n != 32) throw MarshallingError("wrong size buffer");
    std::memcpy(v.data(), data, 32);
  }
  size_t marshal_size() const { return 32; }
  // scalar.rs:54-75
  bool is_canonical(const uint8_t* sb, size_t n) const {
    if (n != 32) return false;
    if ((sb[31] & 0xf0) == 0) return true;
    uint8_t c = 0, m = 1;
    for (int i = 31; i >= 0; --i) {
      c |= (uint8_t)((((uint16_t)sb[i] - (uint16_t)detail::L_BYTES[i]) >> 8) & m);
      m &= (uint8_t)(((((uint16_t)sb[i] ^ (uint16_t)detail::L_BYTES[i]) - 1) >> 8));
    }
    return c != 0;
  }
  std::string hex() const {
    static const char* d = "0123456789abcdef";
    std::string s;
    for (uint8_t b : v) { s.push_back(d[b >> 4]); s.push_back(d[b & 15]); }
    return s;
  }
};

// ------------------------------------------------------------------------------------------- Point
class Point {
 public:
  // X Y Z T, reference limb layout (ExtendedGroupElement, ge.rs:78-83).  Valid when have_ge; a point recorded in the engine's arena
  // and not asked for yet has have_ge == false and only its handle.  Read the limbs through limbs(), which fetches them when needed.
  mutable int32_t ge[40];
  bool var_time = false; // point.rs:26 (never set; kept for layout parity)
  mutable uint64_t pend = 0;     // handle in the engine's arena (kyb_defer_*), 0 = none
  mutable bool have_ge = true;
  // The 32 bytes marshal_binary() returns for this value, once they are known: a point that was unmarshalled from its canonical encoding, or
  // has been marshalled before.  Protocol code marshals what it has just unmarshalled (every hash over received commitments or keys), asks
  // has_small_order() of it (schnorr_sig.rs:79-95) and compares it (point.rs:227-241 compares ENCODINGS): with the bytes at hand none of these
  // reaches the engine.  Part of the value: every operation that changes the point drops it.
  mutable bool have_enc = false;
  mutable uint8_t enc[32];

  Point() { std::memset(ge, 0, sizeof(ge)); }

  // this object now holds the limbs `l` / is the recorded node `h`
  void hold(const int32_t* l) { std::memcpy(ge, l, sizeof(ge)); have_ge = true; pend = 0; have_enc = false; }
  void recorded(uint64_t h) { pend = h; have_ge = false; have_enc = false; }

  // the limbs, evaluated now if the point is still only recorded
  const int32_t* limbs() const {
    if (!have_ge) { detail::engine_must(kyb_defer_get(pend, ge, nullptr), "Point: evaluation of a deferred point"); have_ge = true; }
    return ge;
  }
  // this point as an operand of a recorded operation: its handle, or a leaf made of its limbs.  A point that HOLDS its limbs keeps the
  // handle only as a cache (the marshal_binary that follows is then a hit in the arena): when the arena has dropped the node meanwhile
  // (a leaf leaves no value behind when the arena's window moves on, and kyb_defer_floor drops everything older than its mark -> KYB_E_STALE) the
  // limbs are registered again instead of aborting — record() / marshal / == below.
  uint64_t handle() const {
    if (pend == 0) detail::engine_must(kyb_defer_input_enc(ge, have_enc ? enc : nullptr, &pend), "Point: kyb_defer_input_enc");      // (with its bytes, when it has them)
    return pend;
  }
  bool forget_stale_handle() const { if (have_ge && pend != 0) { pend = 0; return true; } return false; }
  // one recorded operation on up to two operands; a stale CACHED handle of an operand that still holds its limbs is renewed once
  template <class F>
  static uint64_t record(const char* what, const Point* a, const Point* b, F call) {
    uint64_t h = 0;
    int rc = call(&h);
    if (rc == KYB_E_STALE) {
      bool renewed = a != nullptr && a->forget_stale_handle();
      if (b != nullptr && b != a) renewed = b->forget_stale_handle() || renewed;
      if (renewed) rc = call(&h);
    }
    detail::engine_must(rc, what);
    return h;
  }

  Point null() { std::memset(ge, 0, sizeof(ge)); ge[10] = 1; ge[20] = 1; have_ge = true; pend = 0; have_enc = false; return *this; }        // point.rs:79-82
  // 1 * B as the engine hands it out, asked for once per process (the reference copies the literal BASEEXT, constants.rs:70-87: the same point)
  static const int32_t* base_ext() {
    static const std::array<int32_t, 40> b = [] {
      std::array<int32_t, 40> l{};
      uint8_t one[32] = {1};
      detail::engine_must(kyb_mul_base_batch(one, 1, nullptr, l.data()), "Point::base");
      return l;
    }();
    return b.data();
  }
  Point base() { hold(base_ext()); return *this; }   // point.rs:85-88
  Point set(const Point& p) { *this = p; return *this; }                // point.rs:94-97
  size_t embed_len() const { return (255 - 8 - 8) / 8; }                                        // point.rs:99-104
  Point pick(Stream& rand) { return embed(nullptr, 0, rand); }                                  // point.rs:90-92

  // point.rs:106-167
  Point embed(const uint8_t* data, size_t data_len, Stream& rand) {
    size_t dl = embed_len();
    if (dl > data_len) dl = data_len;
    for (;;) {
      uint8_t b[32], z[32] = {0};
      rand.xor_key_stream(b, z, 32);
      if (data != nullptr) { b[0] = (uint8_t)dl; std::memcpy(b + 1, data, dl); }
      uint8_t ok = 0;
      have_ge = true; pend = 0; have_enc = false;              // the rejection loop needs every answer at once: eager calls in either mode
      detail::engine_must(kyb_decode_batch(b, 1, ge, &ok), "Point::embed decode");
      if (!ok) continue;
      if (data == nullptr) {
        uint8_t eight[32] = {8};                              // COFACTOR_SCALAR (constants.rs:47-49)
        uint8_t enc[32];
        int32_t out[40];
        detail::engine_must(kyb_mul_batch(eight, nullptr, ge, 1, enc, out, nullptr), "Point::embed cofactor mul");
        std::memcpy(ge, out, sizeof(ge));
        if (is_identity_encoding(enc)) continue;
        return *this;
      }
      uint8_t enc[32];
      detail::engine_must(kyb_mul_batch(detail::L_BYTES, nullptr, ge, 1, enc, nullptr, nullptr), "Point::embed order check");  // PRIME_ORDER_SCALAR
      if (is_identity_encoding(enc)) return *this;
    }
  }
  // point.rs:169-177
  std::vector<uint8_t> data() const {
    std::vector<uint8_t> b = marshal_binary();
    size_t dl = b[0];
    if (dl > embed_len()) throw PointError("invalid embedded data length");
    return std::vector<uint8_t>(b.begin() + 1, b.begin() + 1 + dl);
  }
  Point add(const Point& a, const Point& b) { return add_sub(a, b, 0); }                       // point.rs:179-188
  Point sub(const Point& a, const Point& b) { return add_sub(a, b, 1); }                       // point.rs:190-199
  Point neg(const Point& a) {                                                                  // point.rs:201-204, ge.rs:86-91... neg X and T
    if (deferred()) { const uint64_t h = record("Point::neg", &a, nullptr, [&](uint64_t* o) { return kyb_defer_neg(a.handle(), o); }); recorded(h); return *this; }
    const int32_t* l = a.limbs();
    int32_t out[40];
    for (int i = 0; i < 10; ++i) { out[i] = -l[i]; out[10 + i] = l[10 + i]; out[20 + i] = l[20 + i]; out[30 + i] = -l[30 + i]; }
    hold(out);
    return *this;
  }
  // point.rs:207-224: p == nullptr -> fixed base (ge_scalar_mult_base), else variable base.  Every in-tree caller passes the generator as
  // Some(base) (PriPoly::commit, poly.rs:195-206; vss): when the operand is, limb for limb, what base() hands out and the scalar is below
  // 2^255 (no top-digit quirk in either routine, ge.rs:440-441) the fixed-base kernel gives the same point in a sixth of the time.
  Point mul(const Scalar& s, const Point* p) {
    if (p != nullptr && p->have_ge && (s.v[31] & 0x80) == 0 && std::memcmp(p->ge, base_ext(), sizeof(ge)) == 0) p = nullptr;
    if (deferred()) {
      const uint64_t h = p == nullptr ? record("Point::mul (base)", nullptr, nullptr, [&](uint64_t* o) { return kyb_defer_mul_base(s.v.data(), o); })
                                      : record("Point::mul", p, nullptr, [&](uint64_t* o) { return kyb_defer_mul(s.v.data(), p->handle(), o); });
      recorded(h);
      return *this;
    }
    int32_t out[40];
    if (p == nullptr) detail::engine_must(kyb_mul_base_batch(s.v.data(), 1, nullptr, out), "Point::mul (base)");
    else detail::engine_must(kyb_mul_batch(s.v.data(), nullptr, p->limbs(), 1, nullptr, out, nullptr), "Point::mul");
    hold(out);
    return *this;
  }
  // Marshaling, point.rs:35-60
  std::vector<uint8_t> marshal_binary() const {
    if (have_enc) return std::vector<uint8_t>(enc, enc + 32);
    std::vector<uint8_t> b(32);
    // (deferred mode: a point that holds limbs without bytes is marshalled THROUGH the arena, as the leaf its limbs are — the bytes stay with the
    //  leaf, so marshalling the same value again costs no engine call even for a binding whose point cannot remember them: the Rust type is Copy)
    if (pend == 0 && deferred()) (void)handle();
    if (pend != 0) {                       // recorded (or registered as an operand): the arena evaluates what it depends on and caches the bytes
      const int rc = kyb_defer_get(pend, have_ge ? nullptr : ge, b.data());
      if (!(rc == KYB_E_STALE && forget_stale_handle())) {      // (a dropped node of a point that holds its limbs: marshal the limbs, below)
        detail::engine_must(rc, "Point::marshal_binary");
        have_ge = true;
        std::memcpy(enc, b.data(), 32); have_enc = true;
        return b;
      }
    }
    detail::engine_must(kyb_encode_batch(ge, 1, b.data()), "Point::marshal_binary");
    std::memcpy(enc, b.data(), 32); have_enc = true;
    return b;
  }
  void unmarshal_binary(const uint8_t* data, size_t n) {
    uint8_t ok = 0;
    int32_t out[40];
    if (n == 32) detail::engine_must(kyb_decode_batch(data, 1, out, &ok), "Point::unmarshal_binary");
    if (n != 32 || !ok) throw MarshallingError("invalid Ed25519 curve point");
    hold(out);
    if (is_the_canonical_encoding(data)) { std::memcpy(enc, data, 32); have_enc = true; }
  }
  // are these 32 bytes, which decode, exactly what marshal_binary() yields for the point they decode to?  Not when y >= p (fe_from_bytes accepts
  // it, fe_to_bytes reduces it: fe.rs:52-122) and not when x = 0 carries a sign bit (ge.rs:124-179 accepts it, the encoder writes sign 0).
  static bool is_the_canonical_encoding(const uint8_t b[32]) {
    uint8_t ff = 0xff, zero = 0;
    for (int i = 1; i <= 30; ++i) { ff &= b[i]; zero |= b[i]; }
    const uint8_t top = b[31] & 0x7f;
    const bool y_ge_p = ff == 0xff && top == 0x7f && b[0] >= 0xed;
    const bool y_is_one = zero == 0 && top == 0 && b[0] == 1, y_is_minus_one = ff == 0xff && top == 0x7f && b[0] == 0xec;
    return !y_ge_p && !((b[31] & 0x80) != 0 && (y_is_one || y_is_minus_one));
  }
  size_t marshal_size() const { return 32; }
  // point.rs:227-241 compares the two encodings (two field inversions); the engine compares projectively, same answer
  bool operator==(const Point& o) const {
    uint8_t eq = 0;
    if (have_enc && o.have_enc) return std::memcmp(enc, o.enc, 32) == 0;      // what the reference compares
    if (!have_ge || !o.have_ge) {
      int rc = kyb_defer_equal(handle(), o.handle(), &eq);
      if (rc == KYB_E_STALE && (forget_stale_handle() | o.forget_stale_handle())) rc = kyb_defer_equal(handle(), o.handle(), &eq);
      detail::engine_must(rc, "Point::eq");
      return eq != 0;
    }
    detail::engine_must(kyb_equal_batch(ge, o.ge, 1, &eq), "Point::eq");
    return eq != 0;
  }
  bool operator!=(const Point& o) const { return !(*this == o); }
  std::string hex() const {
    static const char* d = "0123456789abcdef";
    std::string s;
    for (uint8_t b : marshal_binary()) { s.push_back(d[b >> 4]); s.push_back(d[b & 15]); }
    return s;
  }
  // point.rs:286-313: the encoding against the five WEAK_KEYS below p (constants.rs:3744-3775) — on the engine, which marshals the limbs
  // and compares (kyb_point_checks_batch, bit 1): total on any limbs, as the reference's is
  bool has_small_order() const {
    if (have_enc) return small_order_encoding(enc);
    uint8_t flags = 0;
    detail::engine_must(kyb_point_checks_batch(nullptr, limbs(), 1, &flags), "Point::has_small_order");
    return (flags & 2) != 0;
  }
  // the comparison of point.rs:286-313 itself, on bytes that ARE the point's encoding: sign bit masked, against the encodings of y = 0, 1, p - 1 and
  // of the two order-8 classes (WEAK_KEYS below p, constants.rs:3744-3775; csrc/verify.h pt_has_small_order is the engine's form of it)
  static bool small_order_encoding(const uint8_t e[32]) {
    const uint32_t y8a[8] = KYB_W_ORDER8_Y0, y8b[8] = KYB_W_ORDER8_Y1, pw[8] = KYB_W_P;
    uint32_t w[8];
    std::memcpy(w, e, 32);
    w[7] &= 0x7fffffffu;
    uint32_t d0 = 0, d1 = 0, dm = 0, da = 0, db = 0;
    for (int i = 0; i < 8; ++i) {
      d0 |= w[i];
      d1 |= w[i] ^ (i == 0 ? 1u : 0u);
      dm |= w[i] ^ (i == 0 ? pw[0] - 1u : pw[i]);
      da |= w[i] ^ y8a[i];
      db |= w[i] ^ y8b[i];
    }
    return d0 == 0 || d1 == 0 || dm == 0 || da == 0 || db == 0;
  }
  // point.rs:315-337
  bool is_canonical(const uint8_t* b, size_t n) const {
    if (n != 32) return false;
    uint8_t c = (b[31] & 0x7f) ^ 0x7f;
    for (int i = 30; i >= 1; --i) c |= b[i] ^ 0xff;
    c = (uint8_t)((((uint16_t)c) - 1) >> 8);
    // the reference's own expression, 0xED - (1 - b0) in wrapping u16 (not libsodium's 0xED - 1 - b0): see csrc/verify.h
    const uint16_t inner = (uint16_t)(1u - (uint16_t)b[0]);
    uint8_t d = (uint8_t)((uint16_t)(0xEDu - inner) >> 8);
    return 1 - (c & d & 1) == 1;
  }

  // both checks for many RECEIVED encodings in one engine call (kyb_point_checks_batch: bytes only, no curve arithmetic):
  // first = is_canonical(bytes), second = has_small_order() of the point the bytes unmarshal to (false when they do not)
  static std::vector<std::pair<bool, bool>> checks_batch(const std::vector<std::array<uint8_t, 32>>& encs) {
    std::vector<uint8_t> flags(encs.size());
    if (!encs.empty()) detail::engine_must(kyb_point_checks_batch(encs[0].data(), nullptr, encs.size(), flags.data()), "Point::checks_batch");
    std::vector<std::pair<bool, bool>> r;
    for (uint8_t f : flags) r.emplace_back((f & 1) != 0, (f & 2) != 0);
    return r;
  }

  // ---- batch entry points for throughput callers (PriPoly::commit, poly.rs:195-206) ----
  static std::vector<Point> mul_batch(const std::vector<Scalar>& s, const std::vector<Point>* pts) {
    const size_t n = s.size();
    std::vector<uint8_t> sc(32 * n);
    std::vector<int32_t> in(40 * n), out(40 * n);
    for (size_t i = 0; i < n; ++i) std::memcpy(&sc[32 * i], s[i].v.data(), 32);
    if (pts) {
      if (pts->size() != n) throw std::invalid_argument("mul_batch: size mismatch");
      for (size_t i = 0; i < n; ++i) std::memcpy(&in[40 * i], (*pts)[i].limbs(), 160);
      detail::engine_must(kyb_mul_batch(sc.data(), nullptr, in.data(), n, nullptr, out.data(), nullptr), "Point::mul_batch");
    } else {
      detail::engine_must(kyb_mul_base_batch(sc.data(), n, nullptr, out.data()), "Point::mul_batch (base)");
    }
    std::vector<Point> r(n);
    for (size_t i = 0; i < n; ++i) std::memcpy(r[i].ge, &out[40 * i], 160);
    return r;
  }

 private:
  Point add_sub(const Point& a, const Point& b, int subtract) {
    if (deferred()) { const uint64_t h = record("Point::add", &a, &b, [&](uint64_t* o) { return kyb_defer_add(a.handle(), b.handle(), subtract, o); }); recorded(h); return *this; }
    int32_t out[40];
    detail::engine_must(kyb_add_batch(a.limbs(), b.limbs(), 1, out, subtract), subtract ? "Point::sub" : "Point::add");
    hold(out);
    return *this;
  }
  static bool is_identity_encoding(const uint8_t e[32]) {
    if (e[0] != 1) return false;
    for (int i = 1; i < 32; ++i) if (e[i]) return false;
    return true;
  }
};

// ------------------------------------------------------------------------------------------- Curve
// curve.rs:21-87: Group impl (constructs default Scalar / Point) + Ed25519 key clamping.
class Curve {
 public:
  size_t scalar_len() const { return 32; }
  size_t point_len() const { return 32; }
  Scalar scalar() const { return Scalar(); }
  Point point() const { return Point(); }
  // new_key_and_seed_with_input (curve.rs:74-87) needs SHA-512; the caller passes the 64-byte digest
  // of the seed (the hash itself stays in the Rust/sha2 layer): secret = clamp(digest[0..32]) unreduced.
  static Scalar clamp_digest(const uint8_t digest[64], uint8_t prefix[32]) {
    Scalar s;
    std::memcpy(s.v.data(), digest, 32);
    s.v[0] &= 0xf8; s.v[31] &= 0x7f; s.v[31] |= 0x40;
    if (prefix) std::memcpy(prefix, digest + 32, 32);
    return s;
  }
};

}  // namespace edwards25519
}  // namespace group
}  // namespace kyber
