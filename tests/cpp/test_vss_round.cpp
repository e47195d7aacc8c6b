// One Pedersen-VSS dealer round, restated CALL BY CALL against the C++ mirror of the trait surface — the curve-side calls of
//   new_dealer                 /root/reference src/share/vss/pedersen/vss.rs:287-337   (d_pubb, PriPoly::commit, session_id's marshals)
//   PriPoly::commit            src/share/poly.rs:195-206                                 (t times mul(coeff, Some(base)))
//   Dealer::encrypted_deal     vss.rs:361-386 for every verifier (:390-398)              (mul(dh_secret, None), marshal, schnorr::sign — itself trait
//                                                                                         calls: schnorr_sig.rs:25-47 —, dh_exchange, marshal)
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
// A third run writes the same round the way a BATCH-AWARE caller would: the dealer's t + 1 multiplications in one
// kyb_mul_base_batch call, the n verifier keys marshalled by one kyb_encode_batch, the n ephemeral keys / signatures / Diffie-Hellman points by
// one call each, the n deal checks as one kyb_pubpoly_eval_batch + kyb_mul_base_batch + kyb_equal_batch — the same bytes, and the figure that
// says what the phases made of strictly serial one-item requests (encrypted_deals: ask, wait, ask) cost only because of the call shape.
// Built with -DKYB_CPU_PORT against tests/cpp/cpu_port_abi.cpp (the oracle behind the same ABI) and run with mode "eager", the program times
// the identical call-by-call sequence on one host core: the cpu_port_ms column.
//
//   test_vss_round [n verifiers = 64] [t = 43] [mode: all | eager]
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

// schnorr::sign as unmodified code runs it, schnorr_sig.rs:25-47 + hash :128-141, trait call by trait call: k = pick, R = mul(k, None),
// public = mul(private, None), h = SHA-512(R.marshal || public.marshal || msg) as a scalar, s = k + private * h, out = R || s.  (The mirror's
// schnorr::sign is ONE engine call, kyb_schnorr_sign_batch — what a batch-aware caller uses, and what round_batched below uses: the two must
// give the same 64 bytes.)  In deferred mode the two multiplications are recorded and evaluated together when the hash marshals R.
static std::vector<uint8_t> schnorr_sign_by_trait(Stream& rand, const Scalar& priv, const uint8_t* msg, size_t n) {
  Scalar k = Scalar().pick(rand);
  Point r = Point().mul(k, nullptr);
  Point pub = Point().mul(priv, nullptr);
  std::vector<uint8_t> rb = r.marshal_binary(), pb = pub.marshal_binary();
  kyb::sha512_ctx c;
  kyb::sha512_init(c);
  kyb::sha512_bytes(c, rb.data(), 32);
  kyb::sha512_bytes(c, pb.data(), 32);
  if (n) kyb::sha512_bytes(c, msg, (uint32_t)n);
  uint32_t dw[16];
  kyb::sha512_final(dw, c);                      // the digest as 16 little-endian words = its 64 bytes on this host
  uint8_t dig[64];
  std::memcpy(dig, dw, 64);
  Scalar h = Scalar().set_bytes(dig, 64);
  Scalar sc = k + priv * h;
  std::vector<uint8_t> sig = rb, sb = sc.marshal_binary();
  sig.insert(sig.end(), sb.begin(), sb.end());
  return sig;
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
    std::vector<uint8_t> signature = schnorr_sign_by_trait(rand, longterm, dh_public_buff.data(), dh_public_buff.size());
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

#ifndef KYB_CPU_PORT
// the same round by a caller that owns its batches: raw C-ABI batch calls, the random stream drawn in the order of the call-by-call run
static void round_batched(size_t n, size_t t, Transcript& tr, Timing& tm) {
  XorShiftStream rand;
  auto must = [](int rc, const char* what) { if (rc != KYB_OK) { std::fprintf(stderr, "%s failed (%d): %s\n", what, rc, kyb_last_error()); std::abort(); } };
  Scalar longterm = Scalar().pick(rand), secret = Scalar().pick(rand);
  std::vector<uint8_t> v_priv(32 * n), v_enc(32 * n);
  std::vector<int32_t> v_ext(40 * n);
  for (size_t i = 0; i < n; ++i) { Scalar s = Scalar().pick(rand); std::memcpy(&v_priv[32 * i], s.v.data(), 32); }
  must(kyb_mul_base_batch(v_priv.data(), n, nullptr, v_ext.data()), "verifier keys");            // outside the round, as in round_once
  std::vector<Scalar> coeffs(t);
  coeffs[0] = secret;
  for (size_t j = 1; j < t; ++j) coeffs[j] = Scalar().pick(rand);
  auto bytes32 = [](const uint8_t* p) { return std::vector<uint8_t>(p, p + 32); };
  tr.put("LONGTERM", bytes32(longterm.v.data()));
  for (size_t j = 0; j < t; ++j) tr.put("COEFF", bytes32(coeffs[j].v.data()));
  for (size_t i = 0; i < n; ++i) tr.put("VPRIV", bytes32(&v_priv[32 * i]));
  // ---- new_dealer: d_pubb and the t commitments are ONE fixed-base call; the verifier keys ONE encode ----
  double t0 = now_ms();
  std::vector<uint8_t> sc(32 * (t + 1)), enc(32 * (t + 1));
  std::vector<int32_t> commits_ext(40 * (t + 1));
  std::memcpy(&sc[0], longterm.v.data(), 32);
  for (size_t j = 0; j < t; ++j) std::memcpy(&sc[32 * (j + 1)], coeffs[j].v.data(), 32);
  must(kyb_mul_base_batch(sc.data(), t + 1, enc.data(), commits_ext.data()), "commit");
  must(kyb_encode_batch(v_ext.data(), n, v_enc.data()), "marshal of the verifier keys");
  tr.put("DPUB", bytes32(&enc[0]));
  for (size_t i = 0; i < n; ++i) tr.put("VPUB", bytes32(&v_enc[32 * i]));
  for (size_t j = 0; j < t; ++j) tr.put("COMMIT", bytes32(&enc[32 * (j + 1)]));
  std::vector<Scalar> shares(n);
  for (size_t i = 0; i < n; ++i) shares[i] = pripoly_eval(coeffs, i);
  tm.dealer_setup = now_ms() - t0;
  // ---- encrypted_deals: n ephemeral keys, n signatures over them, n Diffie-Hellman points: three calls ----
  t0 = now_ms();
  std::vector<uint8_t> dh_secret(32 * n), nonce(32 * n), xs(32 * n), dh_key(32 * n), sigs(64 * n), pre(32 * n);
  std::vector<uint32_t> off(n + 1);
  for (size_t i = 0; i < n; ++i) {
    Scalar d = Scalar().pick(rand), k = Scalar().pick(rand);                                     // the order round_once draws them in
    std::memcpy(&dh_secret[32 * i], d.v.data(), 32); std::memcpy(&nonce[32 * i], k.v.data(), 32); std::memcpy(&xs[32 * i], longterm.v.data(), 32);
    off[i] = (uint32_t)(32 * i);
  }
  off[n] = (uint32_t)(32 * n);
  must(kyb_mul_base_batch(dh_secret.data(), n, dh_key.data(), nullptr), "ephemeral keys");
  must(kyb_schnorr_sign_batch(xs.data(), nonce.data(), dh_key.data(), off.data(), n, sigs.data()), "signatures");
  must(kyb_mul_batch(dh_secret.data(), nullptr, v_ext.data(), n, pre.data(), nullptr, nullptr), "dh_exchange");
  for (size_t i = 0; i < n; ++i) {
    tr.put("DHSECRET", bytes32(&dh_secret[32 * i]));
    tr.put("DHKEY", bytes32(&dh_key[32 * i]));
    tr.put("SIG", std::vector<uint8_t>(&sigs[64 * i], &sigs[64 * i] + 64));
    tr.put("PRE", bytes32(&pre[32 * i]));
  }
  tm.encrypted_deals = now_ms() - t0;
  // ---- verify_deal for every verifier: share * B, the polynomial at every index, the comparison: three calls ----
  t0 = now_ms();
  std::vector<uint8_t> sh(32 * (n + 1)), eq(n + 1), ps_enc(32 * (n + 1));
  std::vector<int32_t> fig(40 * (n + 1)), ps_ext(40 * (n + 1));
  std::vector<uint32_t> idx(n + 1);
  for (size_t i = 0; i < n; ++i) { std::memcpy(&sh[32 * i], shares[i].v.data(), 32); idx[i] = (uint32_t)i; }
  std::memcpy(&sh[32 * n], shares[n > 1 ? 1 : 0].v.data(), 32); idx[n] = 0;                      // the deal that does NOT verify
  must(kyb_mul_base_batch(sh.data(), n + 1, nullptr, fig.data()), "share * B");
  must(kyb_pubpoly_eval_batch(commits_ext.data() + 40, t, idx.data(), n + 1, ps_enc.data(), ps_ext.data()), "PubPoly::eval");
  must(kyb_equal_batch(fig.data(), ps_ext.data(), n + 1, eq.data()), "deal check");
  for (size_t i = 0; i < n; ++i) { tr.lines.push_back(std::string("DEALOK ") + (eq[i] ? "1" : "0")); tr.put("PUBSHARE", bytes32(&ps_enc[32 * i])); }
  tr.lines.push_back(std::string("DEALBAD ") + (eq[n] ? "1" : "0"));
  tm.verify_deals = now_ms() - t0;
}
#endif

int main(int argc, char** argv) {
  const bool default_deferred = deferred();      // the mode a caller gets who never calls set_deferred: what the "default_mode" of the TIMING line reports
  const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 64, t = argc > 2 ? (size_t)atol(argv[2]) : 43;
  const bool eager_only = argc > 3 && std::string(argv[3]) == "eager";
  if (kyb_init(0) != KYB_OK) { std::printf("kyb_init failed: %s\n", kyb_last_error()); return 2; }
  // one untimed pass at the timed shape (deferred: the eager pass of a large round is seconds, a small one is enough for it): the context's staging
  // buffers are allocated on the first call of a size, once per context; the CPU port gets a pass as well (caches, page faults)
  {
    Transcript warm; Timing w; uint64_t st[8];
    if (eager_only) round_once(n, t, false, warm, w, st);
    else { round_once(n < 4 ? n : 4, t < 5 ? t : 5, false, warm, w, st); round_once(n, t, true, warm, w, st); }
  }
  if (!eager_only && kyb_defer_floor(kyb_defer_mark()) != KYB_OK) return 3;      // the timed run records into an EMPTY arena: no leaf of the warm pass is found again
  Transcript eager, lazy, batched;
  Timing te, tl, tb;
  uint64_t se[8], sl[8] = {0};
  round_once(n, t, false, eager, te, se);
  for (const std::string& ln : eager.lines) std::printf("E %s\n", ln.c_str());
  const double e_all = te.dealer_setup + te.encrypted_deals + te.verify_deals;
  if (eager_only) {
    std::printf("TIMING {\"default_mode\": \"%s\", \"n\": %zu, \"t\": %zu, \"eager_ms\": {\"new_dealer\": %.3f, \"encrypted_deals\": %.3f, \"verify_deals\": %.3f, \"round\": %.3f}}\n",
                default_deferred ? "deferred" : "eager", n, t, te.dealer_setup, te.encrypted_deals, te.verify_deals, e_all);
    kyb_shutdown();
    return 0;
  }
  round_once(n, t, true, lazy, tl, sl);
  for (const std::string& ln : lazy.lines) std::printf("D %s\n", ln.c_str());
#ifndef KYB_CPU_PORT
  { Transcript warm; Timing w; round_batched(n, t, warm, w); }
  round_batched(n, t, batched, tb);
  for (const std::string& ln : batched.lines) std::printf("B %s\n", ln.c_str());
#endif
  const double l_all = tl.dealer_setup + tl.encrypted_deals + tl.verify_deals, b_all = tb.dealer_setup + tb.encrypted_deals + tb.verify_deals;
  std::printf("TIMING {\"default_mode\": \"%s\", \"n\": %zu, \"t\": %zu, \"eager_ms\": {\"new_dealer\": %.3f, \"encrypted_deals\": %.3f, \"verify_deals\": %.3f, \"round\": %.3f}, "
              "\"deferred_ms\": {\"new_dealer\": %.3f, \"encrypted_deals\": %.3f, \"verify_deals\": %.3f, \"round\": %.3f}, "
              "\"batched_ms\": {\"new_dealer\": %.3f, \"encrypted_deals\": %.3f, \"verify_deals\": %.3f, \"round\": %.3f}, \"speedup\": %.2f, "
              "\"deferred_stats\": {\"nodes\": %llu, \"flushes\": %llu, \"engine_calls\": %llu, \"horner_fused\": %llu, \"sums_fused\": %llu, \"marshal_cache_hits\": %llu}, "
              "\"eager_stats_nodes\": %llu}\n",
              default_deferred ? "deferred" : "eager", n, t, te.dealer_setup, te.encrypted_deals, te.verify_deals, e_all, tl.dealer_setup, tl.encrypted_deals, tl.verify_deals, l_all,
              tb.dealer_setup, tb.encrypted_deals, tb.verify_deals, b_all, e_all / l_all,
              (unsigned long long)sl[0], (unsigned long long)sl[1], (unsigned long long)sl[2], (unsigned long long)sl[3], (unsigned long long)sl[4], (unsigned long long)sl[5],
              (unsigned long long)se[0]);
  kyb_shutdown();
  return 0;
}
