// One Pedersen-VSS dealer round, restated CALL BY CALL against the C++ mirror of the trait surface — the curve-side calls of
//   new_dealer                 /root/reference src/share/vss/pedersen/vss.rs:287-337   (d_pubb, PriPoly::commit, session_id's marshals)
//   PriPoly::commit            src/share/poly.rs:195-206                                 (t times mul(coeff, Some(base)))
//   Dealer::encrypted_deal     vss.rs:361-386 for every verifier (:390-398)              (mul(dh_secret, None), marshal, schnorr::sign, dh_exchange, marshal)
//   Verifier::verify_deal      vss.rs:904-909 for every verifier                         (base().mul(fi.v, None), PubPoly::eval(fi.i), eq)
//   PubPoly::eval              poly.rs:457-469                                           (v = null; t times { v = mul(xi, Some(v)); v = add(v, commits[j]) })
// exactly as unmodified protocol code makes them: one trait call at a time, results looked at where the reference looks at them
// (marshal_binary for the hashes and the AEAD key, eq for the deal check).  The hashing / AEAD / message framing around these calls is
// host code that never reaches the engine and is left out; every byte string the reference would feed to it is PRINTED instead.
//
// The program runs the round twice from the same deterministic stream — first eagerly (every trait call = one batch-of-1 engine call,
// what the drop-in has done so far), then with set_deferred(true) (the calls are recorded, the engine evaluates them in batches when
// bytes are asked for: csrc/defer.inc) — prints both transcripts and the wall time of each phase.  tests/test_gpu_vss_round.py
// compares the transcripts with each other and with the oracle.
//
//   test_vss_round [n verifiers = 64] [t = 43]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../kyber-rs_amd/host/edwards25519.hpp"
#include "../../kyber-rs_amd/host/dh.hpp"
#include "../../kyber-rs_amd/host/schnorr.hpp"

using namespace kyber;
using namespace kyber::group::edwards25519;

struct XorShiftStream : Stream {
  uint64_t s[2] = {0x9E3779B97F4A7C15ULL, 0xD1B54A32D192ED03ULL};
  uint64_t next() {
    uint64_t a = s[0], b = s[1];
    s[0] = b;
    a ^= a << 23; a ^= a >> 17; a ^= b ^ (b >> 26);
    s[1] = a;
    return a + b;
  }
  void xor_key_stream(uint8_t* dst, const uint8_t* src, size_t n) override {
    for (size_t i = 0; i < n; ++i) dst[i] = src[i] ^ (uint8_t)(next() >> 32);
  }
};

static std::string hex(const std::vector<uint8_t>& v) {
  static const char* d = "0123456789abcdef";
  std::string s;
  for (uint8_t b : v) { s.push_back(d[b >> 4]); s.push_back(d[b & 15]); }
  return s;
}
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// poly.rs:133-141 (host scalar arithmetic in either mode)
static Scalar pripoly_eval(const std::vector<Scalar>& coeffs, size_t i) {
  Scalar xi = Scalar().set_int64(1 + (int64_t)i);
  Scalar v = Scalar().zero();
  for (size_t j = coeffs.size(); j-- > 0;) { v = v * xi; v = v + coeffs[j]; }
  return v;
}
// poly.rs:195-206
static std::vector<Point> pripoly_commit(const std::vector<Scalar>& coeffs, const Point* b) {
  std::vector<Point> commits;
  for (size_t i = 0; i < coeffs.size(); ++i) commits.push_back(Point().mul(coeffs[i], b));
  return commits;
}
// poly.rs:457-469
static Point pubpoly_eval(const std::vector<Point>& commits, size_t i) {
  Scalar xi = Scalar().set_int64(1 + (int64_t)i);
  Point v = Point();
  v = v.null();
  for (size_t j = commits.size(); j-- > 0;) {
    Point v_clone = v;
    v = v.mul(xi, &v_clone);
    v_clone = v;
    v = v.add(v_clone, commits[j]);
  }
  return v;
}

struct Transcript { std::vector<std::string> lines; void put(const char* tag, const std::vector<uint8_t>& b) { lines.push_back(std::string(tag) + " " + hex(b)); } };

struct Timing { double dealer_setup = 0, encrypted_deals = 0, verify_deals = 0; };

static void round_once(size_t n, size_t t, bool deferred_mode, Transcript& tr, Timing& tm, uint64_t stats[8]) {
  XorShiftStream rand;
  set_deferred(false);
  // ---- key material of the participants (outside the round: eager in both runs) ----
  Scalar longterm = Scalar().pick(rand), secret = Scalar().pick(rand);
  std::vector<Scalar> v_priv(n);
  std::vector<Point> verifiers(n);
  for (size_t i = 0; i < n; ++i) { v_priv[i] = Scalar().pick(rand); verifiers[i] = Point().mul(v_priv[i], nullptr); }
  std::vector<Scalar> coeffs(t);                                      // new_pri_poly (poly.rs:88-109): the secret, then t - 1 random coefficients
  coeffs[0] = secret;
  for (size_t j = 1; j < t; ++j) coeffs[j] = Scalar().pick(rand);
  tr.put("LONGTERM", std::vector<uint8_t>(longterm.v.begin(), longterm.v.end()));
  for (size_t j = 0; j < t; ++j) tr.put("COEFF", std::vector<uint8_t>(coeffs[j].v.begin(), coeffs[j].v.end()));
  for (size_t i = 0; i < n; ++i) tr.put("VPRIV", std::vector<uint8_t>(v_priv[i].v.begin(), v_priv[i].v.end()));
  uint64_t s0[8], s1[8];
  kyb_defer_stats(s0, 8);
  set_deferred(deferred_mode);

  // ---- new_dealer, vss.rs:287-337 ----
  double t0 = now_ms();
  Point d_pubb = Point().mul(longterm, nullptr);
  Point base = Point().base();
  std::vector<Point> secret_commits = pripoly_commit(coeffs, &base);   // f.commit(Some(&suite.point().base()))
  // session_id (vss.rs:1075-1100): hashes the dealer's key, every verifier's key and every commitment, each through marshal_binary
  tr.put("DPUB", d_pubb.marshal_binary());
  for (size_t i = 0; i < n; ++i) tr.put("VPUB", verifiers[i].marshal_binary());
  for (size_t j = 0; j < t; ++j) tr.put("COMMIT", secret_commits[j].marshal_binary());
  std::vector<Scalar> shares(n);
  for (size_t i = 0; i < n; ++i) shares[i] = pripoly_eval(coeffs, i);  // deals[i].sec_share = f.eval(i)
  tm.dealer_setup = now_ms() - t0;

  // ---- encrypted_deals, vss.rs:390-398 -> encrypted_deal(i), :361-386 ----
  t0 = now_ms();
  for (size_t i = 0; i < n; ++i) {
    Scalar dh_secret = Scalar().pick(rand);
    Point dh_public = Point().mul(dh_secret, nullptr);
    std::vector<uint8_t> dh_public_buff = dh_public.marshal_binary();
    std::vector<uint8_t> signature = sign::schnorr::sign(rand, longterm, dh_public_buff.data(), dh_public_buff.size());
    Point pre = dh::dh_exchange(dh_secret, verifiers[i]);
    std::vector<uint8_t> pre_buff = pre.marshal_binary();              // AEAD::new -> hkdf over the marshalled shared point
    tr.put("DHSECRET", std::vector<uint8_t>(dh_secret.v.begin(), dh_secret.v.end()));
    tr.put("DHKEY", dh_public_buff);
    tr.put("SIG", signature);
    tr.put("PRE", pre_buff);
  }
  tm.encrypted_deals = now_ms() - t0;

  // ---- every verifier: verify_deal, vss.rs:904-909 (the deal's commitments are the points the verifier decoded from it) ----
  t0 = now_ms();
  for (size_t i = 0; i < n; ++i) {
    Point fig = Point().base().mul(shares[i], nullptr);
    Point pub_share = pubpoly_eval(secret_commits, i);
    const bool ok = fig == pub_share;
    tr.lines.push_back(std::string("DEALOK ") + (ok ? "1" : "0"));
    tr.put("PUBSHARE", pub_share.marshal_binary());
  }
  // a deal that does NOT verify: the share of another index against this verifier's evaluation
  {
    Point fig = Point().base().mul(shares[1], nullptr);
    Point pub_share = pubpoly_eval(secret_commits, 0);
    tr.lines.push_back(std::string("DEALBAD ") + ((fig == pub_share) ? "1" : "0"));
  }
  tm.verify_deals = now_ms() - t0;
  set_deferred(false);
  kyb_defer_stats(s1, 8);
  for (int k = 0; k < 8; ++k) stats[k] = s1[k] - s0[k];
}

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 64, t = argc > 2 ? (size_t)atol(argv[2]) : 43;
  if (kyb_init(0) != KYB_OK) { std::printf("kyb_init failed: %s\n", kyb_last_error()); return 2; }
  { Transcript warm; Timing w; uint64_t st[8]; round_once(n < 4 ? n : 4, t < 5 ? t : 5, false, warm, w, st); round_once(n < 4 ? n : 4, t < 5 ? t : 5, true, warm, w, st); }   // first-use allocations out of the way
  Transcript eager, lazy;
  Timing te, tl;
  uint64_t se[8], sl[8];
  round_once(n, t, false, eager, te, se);
  round_once(n, t, true, lazy, tl, sl);
  for (const std::string& ln : eager.lines) std::printf("E %s\n", ln.c_str());
  for (const std::string& ln : lazy.lines) std::printf("D %s\n", ln.c_str());
  const double e_all = te.dealer_setup + te.encrypted_deals + te.verify_deals, l_all = tl.dealer_setup + tl.encrypted_deals + tl.verify_deals;
  std::printf("TIMING {\"n\": %zu, \"t\": %zu, \"eager_ms\": {\"new_dealer\": %.3f, \"encrypted_deals\": %.3f, \"verify_deals\": %.3f, \"round\": %.3f}, "
              "\"deferred_ms\": {\"new_dealer\": %.3f, \"encrypted_deals\": %.3f, \"verify_deals\": %.3f, \"round\": %.3f}, \"speedup\": %.2f, "
              "\"deferred_stats\": {\"nodes\": %llu, \"flushes\": %llu, \"engine_calls\": %llu, \"horner_fused\": %llu, \"sums_fused\": %llu, \"marshal_cache_hits\": %llu}, "
              "\"eager_stats_nodes\": %llu}\n",
              n, t, te.dealer_setup, te.encrypted_deals, te.verify_deals, e_all, tl.dealer_setup, tl.encrypted_deals, tl.verify_deals, l_all, e_all / l_all,
              (unsigned long long)sl[0], (unsigned long long)sl[1], (unsigned long long)sl[2], (unsigned long long)sl[3], (unsigned long long)sl[4], (unsigned long long)sl[5],
              (unsigned long long)se[0]);
  kyb_shutdown();
  return 0;
}
