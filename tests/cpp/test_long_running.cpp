// A node that runs for a long time on UNMODIFIED protocol code in the bindings' default (deferred) mode — round-5 review item 2.
//
// The reference's Point is Copy (point.rs:23) and protocol state holds points for the life of a node (dkg.rs:41,170; dss_sig.rs:44).  The Rust
// binding's point of a recorded operation is nothing but an arena handle (rust/edwards25519_hip/point.rs): `&self` methods cannot cache what they
// fetch, nobody calls materialize() or defer_floor() in unmodified code, and in round 5 such a handle went KYB_E_STALE — the process aborted — once
// defer.max_nodes (2^18) younger nodes existed: after about 43 dealer rounds at n = 64, t = 43.
//
// This program is that client, twice, for `rounds` dealer rounds (vss.rs:287-337, 904-909 and poly.rs:195-206, 457-469 call by call):
//   HandleOnly  the Rust binding's shape through the raw C ABI: a Copy struct of {limbs or handle}; limbs() and encoding() ask the engine EVERY time,
//               nothing is ever stored back, no floor, no materialize;
//   Mirror      the C++ mirror (host/edwards25519.hpp) in its default mode, with the long-lived points kept as copies made BEFORE they were first
//               evaluated (so they too hold only the handle) and every use made through a fresh copy.
// Long-lived state, as a DKG / DSS node has it: a distributed key (a recorded sum of commitments of round 0), marshalled and multiplied in every round,
// and one commitment of every round kept to the end (a DistKeyShare's commits).  Every answer is checked: against bytes taken when the value was new,
// against scalar arithmetic on the host (s * key == (s * k) B), and — the verifiers' checks — g^share == PubPoly::eval(i).
// Prints one line "SOAK {json}" with the arena's statistics; exit code 0 iff nothing went wrong.  Built against the engine (tests/test_gpu_long_running.py)
// and against tests/cpp/cpu_port_abi.cpp -DKYB_CPU_PORT_DEFER — the product's csrc/defer.inc on the CPU, the curve answered by the oracle
// (tests/test_long_running_cpu_port.py), there with a small window so that it is crossed hundreds of times.
//   test_long_running n t rounds [window] [keep_mib]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../kyber-rs_amd/host/edwards25519.hpp"

using namespace kyber::group::edwards25519;
extern "C" int kyb_set_option(const char* key, int value);

static long g_bad = 0;
#define CHECK(c, what) do { if (!(c)) { if (g_bad < 10) std::printf("FAIL line %d: %s (%s)\n", __LINE__, what, kyb_last_error()); ++g_bad; } } while (0)
static void must(int rc, const char* what) {      // what the Rust binding's must() does: the trait has no error channel
  if (rc != KYB_OK) { std::printf("ABORT: %s failed (%d): %s\n", what, rc, kyb_last_error()); std::fflush(stdout); std::abort(); }
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct Rng : kyber::Stream {
  uint64_t s[2] = {0x9E3779B97F4A7C15ULL, 0xD1B54A32D192ED03ULL};
  uint64_t next() { uint64_t a = s[0], b = s[1]; s[0] = b; a ^= a << 23; a ^= a >> 17; a ^= b ^ (b >> 26); s[1] = a; return a + b; }
  void xor_key_stream(uint8_t* dst, const uint8_t* src, size_t n) override { for (size_t i = 0; i < n; ++i) dst[i] = src[i] ^ (uint8_t)(next() >> 32); }
};
typedef std::array<uint8_t, 32> Bytes;

// ---- the Rust binding's point, in C++: Copy, {limbs | handle}, nothing cached behind a const method ----------------------------------------
struct HandleOnly {
  int32_t ge[40];
  uint64_t pend = 0;
  static const char* name() { return "handle_only"; }
  uint64_t handle() const { if (pend) return pend; uint64_t h = 0; must(kyb_defer_input(ge, &h), "defer_input"); return h; }
  static HandleOnly recorded(uint64_t h) { HandleOnly p; std::memset(p.ge, 0, sizeof(p.ge)); p.pend = h; return p; }
  static HandleOnly null() { HandleOnly p; std::memset(p.ge, 0, sizeof(p.ge)); p.ge[10] = 1; p.ge[20] = 1; return p; }
  static HandleOnly mul_base(const Scalar& s) { uint64_t h = 0; must(kyb_defer_mul_base(s.v.data(), &h), "defer_mul_base"); return recorded(h); }
  static HandleOnly mul(const Scalar& s, const HandleOnly& p) { uint64_t h = 0; must(kyb_defer_mul(s.v.data(), p.handle(), &h), "defer_mul"); return recorded(h); }
  static HandleOnly add(const HandleOnly& a, const HandleOnly& b) { uint64_t h = 0; must(kyb_defer_add(a.handle(), b.handle(), 0, &h), "defer_add"); return recorded(h); }
  Bytes encoding() const {
    Bytes b;
    if (pend) must(kyb_defer_get(pend, nullptr, b.data()), "defer_get"); else must(kyb_encode_batch(ge, 1, b.data()), "encode");
    return b;
  }
  bool eq(const HandleOnly& o) const { uint8_t e = 0; must(kyb_defer_equal(handle(), o.handle(), &e), "defer_equal"); return e != 0; }
  HandleOnly fresh_copy() const { return *this; }      // Copy
};

// ---- the C++ mirror in its default mode; long-lived points are copies taken before the first evaluation (handle only) ------------------------
struct Mirror {
  Point p;
  static const char* name() { return "cpp_mirror"; }
  static Mirror null() { Mirror m; m.p.null(); return m; }
  static Mirror mul_base(const Scalar& s) { Mirror m; m.p.mul(s, nullptr); return m; }
  static Mirror mul(const Scalar& s, const Mirror& q) { Mirror m; m.p.mul(s, &q.p); return m; }
  static Mirror add(const Mirror& a, const Mirror& b) { Mirror m; m.p.add(a.p, b.p); return m; }
  Bytes encoding() const { Point c = p; const std::vector<uint8_t> v = c.marshal_binary(); Bytes b; std::memcpy(b.data(), v.data(), 32); return b; }      // (of a copy: `p` itself learns nothing)
  bool eq(const Mirror& o) const { Point a = p, b = o.p; return a == b; }
  Mirror fresh_copy() const { return *this; }
};

static Scalar small_scalar(uint32_t x) { Scalar s; s.v.fill(0); std::memcpy(s.v.data(), &x, 4); return s; }

template <class P>
static bool run(size_t n, size_t t, size_t rounds, std::string& json) {
  Rng rand;
  uint64_t st0[12], st1[12];
  kyb_defer_stats(st0, 12);
  P key = P::null();                    // the distributed key: handle only from the moment it is recorded
  Bytes key_bytes{};
  Scalar key_scalar; key_scalar.v.fill(0);
  std::vector<P> held;                  // one commitment of every round ...
  std::vector<Bytes> held_bytes;        // ... and the bytes it had when it was new
  const long bad0 = g_bad;
  const double t0 = now_ms();
  for (size_t r = 0; r < rounds; ++r) {
    // new_dealer: the secret polynomial and its commitments (PriPoly::commit: t trait calls), marshalled for the session id
    std::vector<Scalar> coeffs(t);
    for (size_t j = 0; j < t; ++j) coeffs[j] = Scalar().pick(rand);
    std::vector<P> commits;
    for (size_t j = 0; j < t; ++j) commits.push_back(P::mul_base(coeffs[j]));
    std::vector<Bytes> commit_bytes;
    for (size_t j = 0; j < t; ++j) commit_bytes.push_back(commits[j].encoding());
    if (r == 0) {
      const size_t terms = t < 3 ? t : 3;
      for (size_t j = 0; j < terms; ++j) { key = P::add(key, commits[j]); key_scalar = key_scalar + coeffs[j]; }
      key_bytes = P(key).encoding();
      CHECK(key_bytes == P::mul_base(key_scalar).encoding(), "the key is the commitment of the sum of its coefficients");
    }
    held.push_back(commits[t - 1]);
    held_bytes.push_back(commit_bytes[t - 1]);
    // every verifier: verify_deal — g^share against the public polynomial's evaluation at its index (Horner, 2 t trait calls)
    for (size_t i = 0; i < n; ++i) {
      const Scalar xi = small_scalar(1 + (uint32_t)i);
      Scalar share; share.v.fill(0);
      for (size_t j = t; j-- > 0;) share = share * xi + coeffs[j];                     // PriPoly::eval, host scalars
      P v = P::null();
      for (size_t j = t; j-- > 0;) { v = P::mul(xi, v); v = P::add(v, commits[j]); }    // PubPoly::eval, poly.rs:457-469
      const P fig = P::mul_base(share);
      CHECK(fig.eq(v), "verify_deal");
      if (i == 1 && r % 16 == 0) CHECK(!P::mul_base(share + small_scalar(1)).eq(v), "a wrong share is refused");
    }
    // the long-lived key: marshalled (a hash over it), and an operand (its use in a signature check), through fresh copies of the handle
    CHECK(key.fresh_copy().encoding() == key_bytes, "the distributed key, marshalled in a later round");
    const Scalar s = Scalar().pick(rand);
    CHECK(P::mul(s, key.fresh_copy()).eq(P::mul_base(s * key_scalar)), "s * key == (s k) B");
    if (r % 16 == 15 || r + 1 == rounds) {
      // state that is looked at rarely: a commitment of the first round and one from the middle
      CHECK(held[0].fresh_copy().encoding() == held_bytes[0], "a commitment of round 0");
      CHECK(held[r / 2].fresh_copy().encoding() == held_bytes[r / 2], "a commitment from the middle");
    }
  }
  for (size_t r = 0; r < rounds; ++r) CHECK(held[r].fresh_copy().encoding() == held_bytes[r], "every kept commitment at the end");
  const double ms = now_ms() - t0;
  kyb_defer_stats(st1, 12);
  const bool ok = g_bad == bad0;
  char buf[1024];
  std::snprintf(buf, sizeof(buf),
                "\"%s\": {\"ok\": %s, \"rounds\": %zu, \"ms_per_round\": %.3f, \"nodes\": %llu, \"left_the_window\": %llu, \"in_window_now\": %llu, \"values_kept_now\": %llu, "
                "\"values_pushed_out\": %llu, \"answers_from_kept_values\": %llu, \"operands_taken_back_in\": %llu, \"flushes\": %llu, \"engine_calls\": %llu, \"horner_fused\": %llu}",
                P::name(), ok ? "true" : "false", rounds, ms / (double)rounds, (unsigned long long)(st1[0] - st0[0]), (unsigned long long)(st1[7] - st0[7]), (unsigned long long)st1[6],
                (unsigned long long)st1[8], (unsigned long long)(st1[9] - st0[9]), (unsigned long long)(st1[10] - st0[10]), (unsigned long long)(st1[11] - st0[11]),
                (unsigned long long)(st1[1] - st0[1]), (unsigned long long)(st1[2] - st0[2]), (unsigned long long)(st1[3] - st0[3]));
  json += buf;
  return ok;
}

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 64, t = argc > 2 ? (size_t)atol(argv[2]) : 43, rounds = argc > 3 ? (size_t)atol(argv[3]) : 110;
  const int window = argc > 4 ? atoi(argv[4]) : 0, keep_mib = argc > 5 ? atoi(argv[5]) : -1;
  if (kyb_init(0) != KYB_OK) { std::printf("kyb_init failed: %s\n", kyb_last_error()); return 2; }
  if (window > 0) must(kyb_set_option("defer.max_nodes", window), "defer.max_nodes");
  if (keep_mib >= 0) must(kyb_set_option("defer.keep_mib", keep_mib), "defer.keep_mib");
  if (!deferred()) { std::printf("the binding's default mode is not the deferred one (KYBER_HIP_EAGER is set?)\n"); return 3; }
  std::string json = "{\"n\": " + std::to_string(n) + ", \"t\": " + std::to_string(t) + ", \"window\": " + std::to_string(window > 0 ? window : 1 << 18) + ", ";
  const bool a = run<HandleOnly>(n, t, rounds, json);
  json += ", ";
  const bool b = run<Mirror>(n, t, rounds, json);
  json += "}";
  std::printf("SOAK %s\n%s\n", json.c_str(), a && b ? "OK" : "FAILED");
  kyb_shutdown();
  return a && b ? 0 : 1;
}
