// GPU test program: the reference's generic group test, /root/reference
// src/util/test/group_test.rs:210-555 (`test_group`), restated against the C++ mirror of the trait
// surface (kyber-rs_amd/host/edwards25519.hpp) — i.e. every identity below runs through the C ABI and
// the HIP kernels, one trait-level call at a time (batch-of-1), exactly as unmodified protocol code
// would.  Built and run by tests/test_gpu_cpp_group.py; exit code 0 = all identities hold.
// Prints the encodings of the points it produced (the role of `compare_groups`,
// group_test.rs:567-585) so the Python side can re-check them against the oracle.
#include <cstdio>
#include <string>
#include <vector>

#include "../../kyber-rs_amd/host/edwards25519.hpp"
#include "../../kyber-rs_amd/host/dh.hpp"
#include "../../kyber-rs_amd/host/poly.hpp"
#include "../../kyber-rs_amd/host/schnorr.hpp"

using namespace kyber;
using namespace kyber::group::edwards25519;

// deterministic test stream (the role of SuiteStable's unseeded XOF, group_test.rs:20-89)
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

static int failures = 0;
#define CHECK(cond, msg) do { if (!(cond)) { std::printf("FAIL: %s (line %d)\n", msg, __LINE__); ++failures; } } while (0)

static void test_embed(Curve& g, Stream& rand, std::vector<Point>& points, const std::string& s) {  // group_test.rs:91-115
  Point p = g.point().embed((const uint8_t*)s.data(), s.size(), rand);
  std::vector<uint8_t> x = p.data();
  size_t max = g.point().embed_len();
  if (max > s.size()) max = s.size();
  CHECK(std::string(x.begin(), x.end()) == s.substr(0, max), "Point extraction/embedding corrupted the data");
  points.push_back(p);
}

int main() {
  if (kyb_init(0) != KYB_OK) { std::printf("kyb_init failed: %s\n", kyb_last_error()); return 2; }
  // everything below runs in the mode a caller gets who never chooses one: deferred, unless KYBER_HIP_EAGER is set (tests/test_gpu_cpp_group.py runs both)
  const bool default_mode = deferred();
  std::printf("MODE %s\n", default_mode ? "deferred" : "eager");
  Curve g;
  XorShiftStream rand;
  std::vector<Point> points;
  Point ptmp = g.point();
  Scalar stmp = g.scalar();
  Point pzero = g.point().null();
  Scalar szero = g.scalar().zero();
  Scalar sone = g.scalar().one();

  Scalar s1 = g.scalar().pick(rand), s2 = g.scalar().pick(rand);
  CHECK(s1 != szero && s2 != szero && s1 != s2, "unique non-zero secrets");

  Point gen = g.point().base();
  points.push_back(gen);

  // addition vs multiplication (group_test.rs:243-258)
  Point p1 = g.point().add(gen, gen);
  Point p2 = g.point().mul(Scalar().set_int64(2), nullptr);
  CHECK(p1 == p2, "multiply by two");
  p1 = Point().add(p1, p1);
  p2 = p2.mul(Scalar().set_int64(4), nullptr);
  CHECK(p1 == p2, "multiply by four");
  points.push_back(p1);

  // additive and multiplicative identities of the generator (:270-292)
  ptmp = ptmp.mul(Scalar().set_int64(-1), nullptr);
  ptmp = Point().add(ptmp, gen);
  CHECK(ptmp == pzero, "generator additive identity");
  stmp.set_int64(2);
  ptmp = ptmp.mul(stmp, nullptr);
  { Point q = ptmp; ptmp = ptmp.mul(Scalar().inv(stmp), &q); }
  CHECK(ptmp == gen, "generator multiplicative identity");

  // Diffie-Hellman (:294-322)
  p1 = p1.mul(s1, &gen);
  p2 = p2.mul(s2, &gen);
  CHECK(p1 != p2, "encryption produces unique points");
  points.push_back(p1);
  Point dh1 = g.point().mul(s2, &p1), dh2 = g.point().mul(s1, &p2);
  CHECK(dh1 == dh2, "Diffie-Hellman");
  points.push_back(dh1);

  // inverse, zero, one (:326-358)
  ptmp = ptmp.mul(Scalar().inv(s2), &dh1);
  CHECK(ptmp == p1, "scalar inverse");
  CHECK(Point().mul(szero, &dh1) == pzero, "secret = 0");
  CHECK(Point().mul(sone, &dh1) == dh1, "secret = 1");

  // additive homomorphism (:361-420)
  ptmp = ptmp.add(p1, p2);
  stmp = s1 + s2;
  Point pt2 = g.point().mul(stmp, &gen);
  CHECK(pt2 == ptmp, "additive homomorphism (+)");
  ptmp = ptmp.sub(p1, p2);
  stmp = stmp.sub(s1, s2);
  pt2 = pt2.mul(stmp, &gen);
  CHECK(pt2 == ptmp, "additive homomorphism (-)");
  Scalar st2 = g.scalar().neg(s2);
  st2 = s1 + st2;
  CHECK(stmp == st2, "Scalar.neg");
  pt2 = pt2.neg(p2);
  pt2 = Point().add(pt2, p1);
  CHECK(pt2 == ptmp, "Point.neg");

  // multiplicative homomorphism (:423-461)
  stmp = s1 * s2;
  CHECK(Point().mul(stmp, &gen) == dh1, "multiplicative homomorphism");
  st2 = st2.inv(s2);
  st2 = st2 * stmp;
  CHECK(st2 == s1, "scalar division via inverse");
  st2 = st2.div(stmp, s2);
  CHECK(st2 == s1, "scalar division");

  // randomly picked points (:465-502)
  Point last = gen;
  for (int i = 0; i < 5; ++i) {
    Point rgen = g.point().pick(rand);
    CHECK(rgen != last, "pick produces unique points");
    last = rgen;
    ptmp = Point().mul(Scalar().set_int64(-1), &rgen);
    ptmp = Point().add(ptmp, rgen);
    CHECK(ptmp == pzero, "random generator additive identity");
    stmp.set_int64(2);
    ptmp = Point().mul(stmp, &rgen);
    { Point q = ptmp; ptmp = Point().mul(Scalar().inv(stmp), &q); }
    CHECK(ptmp == rgen, "random generator multiplicative identity");
    points.push_back(rgen);
  }

  // embedding (:505-511)
  test_embed(g, rand, points, "Hi!");
  test_embed(g, rand, points, "The quick brown fox jumps over the lazy dog");

  // encoding / decoding (:516-540)
  for (int i = 0; i < 5; ++i) {
    Scalar s = g.scalar().pick(rand);
    std::vector<uint8_t> buf = s.marshal_binary();
    stmp.unmarshal_binary(buf.data(), buf.size());
    CHECK(stmp == s, "scalar decode(encode)");
    Point p = g.point().pick(rand);
    buf = p.marshal_binary();
    ptmp.unmarshal_binary(buf.data(), buf.size());
    CHECK(ptmp == p, "point decode(encode)");
    points.push_back(p);
  }
  // null point marshal round trip (:543-547)
  {
    std::vector<uint8_t> b = g.point().null().marshal_binary();
    Point q;
    bool threw = false;
    try { q.unmarshal_binary(b.data(), b.size()); } catch (const MarshallingError&) { threw = true; }
    CHECK(!threw && q == pzero, "null point round trip");
  }
  // error behaviour: invalid encoding -> MarshallingError("invalid Ed25519 curve point") (point.rs:43-50)
  {
    uint8_t bad[32] = {2};
    bool threw = false;
    try { Point().unmarshal_binary(bad, 32); } catch (const MarshallingError& e) { threw = std::string(e.what()) == "invalid Ed25519 curve point"; }
    CHECK(threw, "invalid point is rejected with the reference's message");
    threw = false;
    try { Scalar().unmarshal_binary(bad, 31); } catch (const MarshallingError& e) { threw = std::string(e.what()) == "wrong size buffer"; }
    CHECK(threw, "scalar length check");
  }
  // canonical / small-order checks (point.rs:286-337, scalar.rs:54-75; scalar_test.rs:89-105)
  {
    Point w; uint8_t wk[32] = {0}; w.unmarshal_binary(wk, 32);
    CHECK(w.has_small_order(), "weak key 0 has small order");
    CHECK(!gen.has_small_order(), "generator has large order");
    std::vector<uint8_t> ge = gen.marshal_binary();
    CHECK(gen.is_canonical(ge.data(), 32), "generator encoding canonical");
    uint8_t pm[32]; for (int i = 0; i < 32; ++i) pm[i] = 0xff; pm[0] = 0xed; pm[31] = 0x7f;   // y = p
    CHECK(!gen.is_canonical(pm, 32), "y = p is not canonical");
    uint8_t l2[32]; std::memcpy(l2, detail::L_BYTES, 32); l2[0] -= 2;
    bool exp[4] = {true, true, false, false};
    for (int i = 0; i < 4; ++i) { CHECK(Scalar().is_canonical(l2, 32) == exp[i], "scalar canonical range L-2..L+1"); l2[0] += 1; }
  }
  // scalar KATs (scalar_test.rs:27-75)
  CHECK((Scalar().set_int64(0x100) + sone).hex() == "0101000000000000000000000000000000000000000000000000000000000000", "set_int64(0x100)+1");
  CHECK(Scalar().set_int64(-1).hex() == "ecd3f55c1a631258d69cf7a2def9de1400000000000000000000000000000010", "set_int64(-1)");
  { uint8_t b[4] = {0, 1, 2, 3}; CHECK(Scalar().set_bytes(b, 4).hex() == "0001020300000000000000000000000000000000000000000000000000000000", "set_bytes LE"); }
  { Scalar two = Scalar().set_int64(2); Scalar r = Scalar().pick(rand); CHECK(two * r == r + r, "2*s == s+s"); }

  // batch entry point equals the per-call path
  {
    std::vector<Scalar> ss; std::vector<Point> pp;
    for (int i = 0; i < 70; ++i) { ss.push_back(Scalar().pick(rand)); pp.push_back(points[i % points.size()]); }
    std::vector<Point> rb = Point::mul_batch(ss, &pp), rf = Point::mul_batch(ss, nullptr);
    for (int i = 0; i < 70; i += 23) {
      CHECK(rb[i] == Point().mul(ss[i], &pp[i]), "mul_batch == mul");
      CHECK(rf[i] == Point().mul(ss[i], nullptr), "mul_batch(base) == mul(None)");
    }
  }

  // Schnorr sign / verify through the mirror (schnorr_test.rs:15-82): round trip, wrong message, tampering
  {
    using namespace kyber::sign;
    Scalar x = Scalar().pick(rand);
    Point X = Point().mul(x, nullptr);
    const uint8_t msg[] = "Hello Schnorr";
    std::vector<uint8_t> sig = schnorr::sign(rand, x, msg, sizeof(msg) - 1);
    bool ok = true;
    try { schnorr::verify(X, msg, sizeof(msg) - 1, sig.data(), sig.size()); } catch (const SignatureError&) { ok = false; }
    CHECK(ok, "schnorr sign/verify round trip");
    std::string err;
    try { schnorr::verify(X, (const uint8_t*)"wrong", 5, sig.data(), sig.size()); } catch (const SignatureError& e) { err = e.what(); }
    CHECK(err == "reconstructed S is not equal to signature", "wrong message is rejected");
    err.clear();
    try { schnorr::verify(X, msg, sizeof(msg) - 1, sig.data(), 63); } catch (const SignatureError& e) { err = e.what(); }
    CHECK(err == "schnorr: signature of invalid length 63 instead of 64", "length check");
    std::vector<uint8_t> mal = sig;                        // s + L: malleability (schnorr_test.rs:84-110)
    unsigned c = 0;
    for (int i = 0; i < 32; ++i) { c += (unsigned)mal[32 + i] + kyber::group::edwards25519::detail::L_BYTES[i]; mal[32 + i] = (uint8_t)c; c >>= 8; }
    err.clear();
    try { schnorr::verify(X, msg, sizeof(msg) - 1, mal.data(), 64); } catch (const SignatureError& e) { err = e.what(); }
    CHECK(err == "signature is not canonical", "s + L is rejected");
    // many triples in one call, keys as points: the same answers as one by one
    {
      std::vector<Point> keys;
      std::vector<std::vector<uint8_t>> ms, sgs;
      for (int i = 0; i < 9; ++i) {
        Scalar xi = Scalar().pick(rand);
        keys.push_back(Point().mul(xi, nullptr));
        ms.push_back(std::vector<uint8_t>(msg, msg + (sizeof(msg) - 1 - (size_t)i % 4)));
        sgs.push_back(schnorr::sign(rand, xi, ms.back().data(), ms.back().size()));
      }
      sgs[2][40] ^= 1; sgs[5] = mal; sgs[7].pop_back();
      std::vector<uint8_t> stv = schnorr::verify_batch(keys, ms, sgs);
      bool same = stv.size() == 9;
      for (int i = 0; i < 9 && same; ++i) {
        int one = 0;
        try { schnorr::verify(keys[i], ms[i].data(), ms[i].size(), sgs[i].data(), sgs[i].size()); } catch (const SignatureError& e) { one = e.code; }
        same = (stv[i] == 0) == (one == 0) && (i == 2 ? stv[i] == 9 : true) && (i == 7 ? stv[i] == 1 : true);
      }
      CHECK(same, "schnorr::verify_batch == verify one by one");
    }
    std::vector<uint8_t> pb = X.marshal_binary();
    ok = true;
    try { eddsa::verify_with_checks(pb.data(), 32, msg, sizeof(msg) - 1, sig.data(), 64); } catch (const SignatureError&) { ok = false; }
    CHECK(ok, "a Schnorr signature is a valid EdDSA signature (schnorr_sig.rs:22-24)");
  }
  // EdDSA key objects (eddsa_test.rs:19-46, 48-77): RFC 8032 section 7.1 tests 1 and 2, marshal round trip, fresh key
  {
    using namespace kyber::sign;
    auto unhex = [](const char* h) { std::vector<uint8_t> v; for (size_t i = 0; h[i] && h[i + 1]; i += 2) { unsigned b; std::sscanf(h + i, "%2x", &b); v.push_back((uint8_t)b); } return v; };
    auto hexs = [](const std::vector<uint8_t>& v) { static const char* d = "0123456789abcdef"; std::string s; for (uint8_t b : v) { s.push_back(d[b >> 4]); s.push_back(d[b & 15]); } return s; };
    struct { const char *seed, *pub, *msg, *sig; } vec[2] = {
        {"9d61b19deffd5a60ba844af492ec2cc44449c5697b326919703bac031cae7f60", "d75a980182b10ab7d54bfed3c964073a0ee172f3daa62325af021a68f707511a", "",
         "e5564300c360ac729086e2cc806e828a84877f1eb8e5d974d873e065224901555fb8821590a33bacc61e39701cf9b46bd25bf5f0595bbe24655141438e7a100b"},
        {"4ccd089b28ff96da9db6c346ec114e0f5b8a319f35aba624da8cf6ed4fb8a6fb", "3d4017c3e843895a92b70aa74d1b7ebc9c982ccf2ec4968cc0cd55f12af4660c", "72",
         "92a009a9f0d4cab8720e820b5f642540a2b27b5416503f8fb3762223ebdb69da085ac1e43e15996e458f3613d0f11d8c387b2eaeb4302aeeb00d291612bb0c00"}};
    for (auto& v : vec) {
      std::vector<uint8_t> seed = unhex(v.seed), msg = unhex(v.msg);
      eddsa::EdDSA ed = eddsa::EdDSA::from_seed_bytes(seed.data());
      CHECK(ed.public_key.hex() == v.pub, "RFC 8032 public key");
      std::vector<uint8_t> sig = ed.sign(msg.data(), msg.size());
      CHECK(hexs(sig) == v.sig, "RFC 8032 signature");
      bool ok = true;
      try { ed.verify(msg.data(), msg.size(), sig.data(), sig.size()); } catch (const SignatureError&) { ok = false; }
      CHECK(ok, "EdDSA::verify accepts its own signature");
      std::vector<uint8_t> m = ed.marshal_binary();
      CHECK(hexs(m) == std::string(v.seed) + v.pub, "marshal_binary = seed || public");
      eddsa::EdDSA back;
      back.unmarshal_binary(m.data(), m.size());
      CHECK(back == ed, "unmarshal_binary(marshal_binary) round trip");
    }
    eddsa::EdDSA fresh(rand);
    const uint8_t msg[] = "Hello Gophers";
    std::vector<uint8_t> sig = fresh.sign(msg, sizeof(msg) - 1);
    bool ok = true;
    try { eddsa::verify(fresh.public_key, msg, sizeof(msg) - 1, sig.data(), 64); } catch (const SignatureError&) { ok = false; }
    CHECK(ok, "fresh EdDSA key signs and verifies");
    std::string err;
    try { eddsa::EdDSA().unmarshal_binary(sig.data(), 63); } catch (const MarshallingError& e) { err = e.what(); }
    CHECK(err == "wrong length for decoding EdDSA private", "EdDSA::unmarshal_binary length check");
    // From<Pair>: empty prefix, nonce = H(msg)
    Scalar x = Scalar().pick(rand);
    eddsa::EdDSA pairkey = eddsa::EdDSA::from_pair(x, Point().mul(x, nullptr));
    sig = pairkey.sign(msg, sizeof(msg) - 1);
    ok = true;
    try { pairkey.verify(msg, sizeof(msg) - 1, sig.data(), 64); } catch (const SignatureError&) { ok = false; }
    CHECK(ok, "EdDSA from a key pair signs and verifies");
  }
  // polynomials (poly_test.rs: public shares of the commitment equal the commitments of the private shares)
  {
    using namespace kyber::share;
    PriPoly pp;
    for (int i = 0; i < 5; ++i) pp.coeffs.push_back(Scalar().pick(rand));
    PubPoly pub = pp.commit(nullptr);
    std::vector<PubShare> sh = pub.shares(9);
    for (size_t i = 0; i < 9; ++i) {
      PriShare ps = pp.eval(i);
      CHECK(sh[i].v == Point().mul(ps.v, nullptr), "PubPoly::eval(i) == commit of PriPoly::eval(i)");
      CHECK(pub.check(ps), "PubPoly::check accepts a good share");
    }
    PriShare badshare = pp.eval(3); badshare.v = badshare.v + Scalar().one();
    CHECK(!pub.check(badshare), "PubPoly::check rejects a bad share");
    Point genb = Point().base();
    PubPoly pubg = pp.commit(&genb);                     // Some(generator): fixed-base shortcut, same points as None
    CHECK(pubg.equal(pub) && pubg.b.has_value(), "commit(Some(B)) == commit(None)");
    Point gen2 = Point().add(Point().mul(Scalar().set_int64(3), nullptr), Point().neg(Point().mul(Scalar().set_int64(2), nullptr)));   // B with other limbs
    CHECK(pp.commit(&gen2).equal(pub), "commit(Some(B in another representation)) == commit(None)");
    Point h = points[5];
    PubPoly pubh = pp.commit(&h);
    CHECK(pubh.eval(2).v == Point().mul(pp.eval(2).v, &h), "commit with an explicit base");
  }

  // public recovery (poly_test.rs:140-225, 250-303): recover_commit / recover_pub_poly / add / equal
  {
    using namespace kyber::share;
    const size_t n = 10, t = n / 2 + 1;
    auto new_pri_poly = [&](size_t k) { PriPoly p; for (size_t i = 0; i < k; ++i) p.coeffs.push_back(Scalar().pick(rand)); return p; };
    auto opt = [](const std::vector<PubShare>& v) { return std::vector<std::optional<PubShare>>(v.begin(), v.end()); };
    PriPoly pri = new_pri_poly(t);
    PubPoly pub = pri.commit(nullptr);
    std::vector<std::optional<PubShare>> shares = opt(pub.shares(n));
    CHECK(recover_commit(shares, t, n) == pub.commit(), "test_public_recovery: recover_commit");
    CHECK(pub.equal(recover_pub_poly(shares, t, n)), "test_public_recovery: recover_pub_poly");
    std::vector<std::optional<PubShare>> selected(shares.begin() + (n - t), shares.end());
    CHECK(recover_commit(selected, t, t + 1) == pub.commit(), "test_public_recovery_out_index");
    std::vector<std::optional<PubShare>> holes = shares;
    holes[2].reset(); holes[5].reset(); holes[7].reset(); holes[8].reset();
    CHECK(recover_commit(holes, t, n) == pub.commit(), "test_public_recovery_delete");
    CHECK(pub.equal(recover_pub_poly(holes, t, n)), "recover_pub_poly from the surviving shares");
    holes[1].reset();
    std::string err;
    try { (void)recover_commit(holes, t, n); } catch (const PolyError& e) { err = e.what(); }
    CHECK(err == "not enough good public shares to reconstruct secret commitment", "test_public_recovery_delete_fail");
    // the Lagrange coefficients the engine computes (one GPU lane each) are the ones the reference's loop gives (poly.rs:585-594)
    {
      std::vector<uint32_t> idx = {0, 3, 4, 6, 9, 17, 250};
      std::vector<Scalar> xs;
      for (uint32_t i : idx) xs.push_back(Scalar().set_int64((int64_t)i + 1));
      std::vector<Scalar> host = kyber::share::detail::lagrange_at_zero(xs), gpu = kyber::share::detail::lagrange_at_zero_gpu(idx, 1, idx.size());
      for (size_t i = 0; i < idx.size(); ++i) CHECK(host[i] == gpu[i], "lagrange_at_zero: engine == host restatement");
    }
    // PriPoly::shares through the engine == PriPoly::eval on the host (poly.rs:133-152), short and long polynomials
    for (size_t tt : {size_t(1), size_t(5), size_t(90)}) {
      PriPoly q = new_pri_poly(tt);
      std::vector<PriShare> sh = q.shares(37);
      bool same = sh.size() == 37;
      for (size_t i = 0; i < sh.size(); ++i) same = same && sh[i].i == i && sh[i].v == q.eval(i).v;
      CHECK(same, "PriPoly::shares: engine == host Horner");
    }
    // recover_secret (poly.rs:244-280): the secret back from t of the n private shares, whichever t (the first t by index are used)
    {
      PriPoly q = new_pri_poly(t);
      std::vector<PriShare> sh = q.shares(n);
      std::vector<std::optional<PriShare>> some(sh.begin(), sh.end());
      CHECK(recover_secret(some, t, n) == q.coeffs[0], "recover_secret: all shares");
      some[0].reset(); some[3].reset();
      CHECK(recover_secret(some, t, n) == q.coeffs[0], "recover_secret: with holes");
      std::string e2;
      for (size_t i = 0; i + t - 1 < some.size(); ++i) some[i].reset();
      try { (void)recover_secret(some, t, n); } catch (const PolyError& e) { e2 = e.what(); }
      CHECK(e2 == "not enough shares to recover secret", "recover_secret: too few shares");
    }
    // batch of share sets in one launch == one by one
    {
      std::vector<std::vector<std::optional<PubShare>>> sets;
      std::vector<Point> want;
      for (int k = 0; k < 5; ++k) { PubPoly q = new_pri_poly(t).commit(nullptr); sets.push_back(opt(q.shares(n))); sets.back()[k].reset(); want.push_back(q.commit()); }
      std::vector<Point> got = recover_commit_batch(sets, t, n);
      for (int k = 0; k < 5; ++k) CHECK(got[k] == want[k], "recover_commit_batch");
    }
    // one launch for a verifier's checks of several dealers at its own index
    {
      std::vector<PubPoly> polys; std::vector<PriPoly> pris; std::vector<uint32_t> at;
      for (int k = 0; k < 6; ++k) { pris.push_back(new_pri_poly(t)); polys.push_back(pris.back().commit(nullptr)); at.push_back(3); }
      at[4] = 9;
      std::vector<PubShare> got = eval_each(polys, at);
      for (int k = 0; k < 6; ++k) CHECK(got[k].v == Point().mul(pris[k].eval(at[k]).v, nullptr), "eval_each == commit of the private share");
      // the same deals as they come off the wire: t encodings per dealer
      std::vector<uint8_t> wire;
      for (const PubPoly& q : polys) for (const Point& c : q.commits) { std::vector<uint8_t> b = c.marshal_binary(); wire.insert(wire.end(), b.begin(), b.end()); }
      std::vector<PubShare> gw = eval_each_wire(wire, t, at);
      for (int k = 0; k < 6; ++k) CHECK(gw[k].v == got[k].v && gw[k].i == got[k].i, "eval_each_wire == eval_each");
      CHECK(sum_polys_wire(wire, t, polys[0].b).equal(sum_polys(polys)), "sum_polys_wire == sum_polys");
      std::string werr;
      wire[32 * 7] = 2; for (int b = 1; b < 32; ++b) wire[32 * 7 + b] = 0;          // y = 2 is not on the curve
      try { (void)eval_each_wire(wire, t, at); } catch (const MarshallingError& e) { werr = e.what(); }
      CHECK(werr == "invalid Ed25519 curve point", "eval_each_wire: a commitment that does not decode is the reference's unmarshal error");
    }
    // test_public_add
    Point gp = points[5], h = points[6];
    PriPoly p = new_pri_poly(t), q = new_pri_poly(t);
    PubPoly pP = p.commit(&gp), qP = q.commit(&h);
    PubPoly r = pP.add(qP);
    CHECK(recover_commit(opt(r.shares(n)), t, n) == Point().add(pP.commit(), qP.commit()), "test_public_add");
    err.clear();
    try { (void)pP.add(new_pri_poly(t - 1).commit(&gp)); } catch (const PolyError& e) { err = e.what(); }
    CHECK(err == "different number of coefficients", "PubPoly::add threshold mismatch");
    // test_public_poly_equal
    PubPoly p1 = new_pri_poly(t).commit(&gp), p2 = new_pri_poly(t).commit(&gp), p3 = new_pri_poly(t).commit(&gp);
    CHECK(p1.add(p2).add(p3).equal(p1.add(p3).add(p2)), "test_public_poly_equal");
    CHECK(sum_polys({p1, p2, p3}).equal(p1.add(p2).add(p3)), "sum_polys == PubPoly::add folded over the dealers");
    CHECK(!p1.equal(p2), "different polynomials are not equal");
  }

  // Diffie-Hellman (dh_test.rs / vss.rs:371-375): both sides derive the same pre-shared key; batch == single
  {
    using namespace kyber::dh;
    Scalar a = Scalar().pick(rand);
    Point A = Point().mul(a, nullptr);
    std::vector<Scalar> bs; std::vector<std::vector<uint8_t>> Bs;
    for (int i = 0; i < 9; ++i) { bs.push_back(Scalar().pick(rand)); Bs.push_back(Point().mul(bs.back(), nullptr).marshal_binary()); }
    std::vector<std::vector<uint8_t>> pre = dh_exchange_batch({a}, Bs);
    for (int i = 0; i < 9; ++i) {
      CHECK(pre[i] == dh_exchange(bs[i], A).marshal_binary(), "dh_exchange_batch: a*B_i == b_i*A");
    }
    std::vector<std::vector<uint8_t>> badkeys = Bs;
    badkeys[4].assign(32, 0); badkeys[4][0] = 2;                      // y = 2 is not on the curve
    std::string err;
    try { (void)dh_exchange_batch({a}, badkeys); } catch (const MarshallingError& e) { err = e.what(); }
    CHECK(err == "invalid Ed25519 curve point", "dh_exchange_batch rejects an invalid remote key");
  }

  // the stand-alone checks on received encodings: one engine call == the per-point methods (point.rs:286-337)
  {
    std::vector<std::array<uint8_t, 32>> encs;
    std::vector<Point> decoded;
    for (const Point& p : points) { std::array<uint8_t, 32> e; auto b = p.marshal_binary(); std::memcpy(e.data(), b.data(), 32); encs.push_back(e); decoded.push_back(p); }
    { std::array<uint8_t, 32> e{}; e[0] = 1; encs.push_back(e); decoded.push_back(Point().null()); }                      // the neutral element: small order
    { std::array<uint8_t, 32> e; e.fill(0xff); e[0] = 0xec; e[31] = 0x7f; encs.push_back(e); Point q; q.unmarshal_binary(e.data(), 32); decoded.push_back(q); }   // y = p - 1: order 2, and "not canonical" by the reference's expression
    auto flags = Point::checks_batch(encs);
    CHECK(flags.size() == encs.size(), "checks_batch size");
    bool saw_small = false, saw_noncanonical = false;
    for (size_t i = 0; i < encs.size(); ++i) {
      CHECK(flags[i].first == Point().is_canonical(encs[i].data(), 32), "checks_batch: is_canonical");
      CHECK(flags[i].second == decoded[i].has_small_order(), "checks_batch: has_small_order");
      saw_small |= flags[i].second; saw_noncanonical |= !flags[i].first;
    }
    CHECK(saw_small && saw_noncanonical, "checks_batch: both flags exercised");
  }
  // the encoding a Point keeps once it is known (unmarshalled from its canonical bytes, or marshalled before): marshal_binary, has_small_order
  // and == then answer from the bytes — the same answers the engine gives on the limbs; encodings that are NOT what marshal_binary yields
  // (y >= p; x = 0 with a sign bit) are not kept; any operation on the point drops the bytes
  {
    std::vector<std::array<uint8_t, 32>> encs;
    for (size_t i = 0; i < 3 && i < points.size(); ++i) { std::array<uint8_t, 32> e; auto b = points[i].marshal_binary(); std::memcpy(e.data(), b.data(), 32); encs.push_back(e); }
    { std::array<uint8_t, 32> e{}; e[0] = 1; encs.push_back(e); }                                             // neutral element
    { std::array<uint8_t, 32> e{}; encs.push_back(e); }                                                       // y = 0: order 4
    { std::array<uint8_t, 32> e; e.fill(0xff); e[0] = 0xec; e[31] = 0x7f; encs.push_back(e); }              // y = p - 1: order 2
    const size_t canonical = encs.size();
    { std::array<uint8_t, 32> e{}; e[0] = 1; e[31] = 0x80; encs.push_back(e); }                              // x = 0 with a sign bit (ge.rs:124-179 accepts it)
    { std::array<uint8_t, 32> e; e.fill(0xff); e[0] = 0xec; encs.push_back(e); }                             // y = p - 1, sign bit set
    { std::array<uint8_t, 32> e; e.fill(0xff); e[0] = 0xed; e[31] = 0x7f; encs.push_back(e); }              // y = p      (= 0)
    { std::array<uint8_t, 32> e; e.fill(0xff); e[0] = 0xee; e[31] = 0x7f; encs.push_back(e); }              // y = p + 1  (= 1)
    for (size_t i = 0; i < encs.size(); ++i) {
      Point q;
      q.unmarshal_binary(encs[i].data(), 32);
      CHECK(q.have_enc == (i < canonical), "the bytes are kept exactly when they are the canonical encoding");
      uint8_t want[32], flags = 0;
      CHECK(kyb_encode_batch(q.limbs(), 1, want) == KYB_OK && kyb_point_checks_batch(nullptr, q.limbs(), 1, &flags) == KYB_OK, "engine answers");
      CHECK(q.has_small_order() == ((flags & 2) != 0), "has_small_order: bytes and engine agree");
      const std::vector<uint8_t> m = q.marshal_binary();
      CHECK(std::memcmp(m.data(), want, 32) == 0, "marshal_binary: bytes and engine agree");
      CHECK((std::memcmp(m.data(), encs[i].data(), 32) == 0) == (i < canonical), "a non-canonical encoding is not what marshal_binary returns");
      CHECK(q.have_enc && q.has_small_order() == ((flags & 2) != 0), "after a marshal the bytes are kept");
      Point fresh;                                                     // the same point without bytes: == through the engine, then through the bytes
      std::memcpy(fresh.ge, q.limbs(), sizeof(fresh.ge));
      CHECK(!fresh.have_enc && q == fresh && fresh == q, "== with bytes on one side");
      (void)fresh.marshal_binary();
      CHECK(fresh.have_enc && q == fresh, "== with bytes on both sides");
      Point moved = q;
      CHECK(moved.have_enc, "a copy keeps the bytes");
      moved = moved.add(moved, gen);
      CHECK(!moved.have_enc && !(moved == q), "an operation drops them");
      uint8_t want2[32];
      CHECK(kyb_encode_batch(moved.limbs(), 1, want2) == KYB_OK && std::memcmp(moved.marshal_binary().data(), want2, 32) == 0, "marshal after the operation");
      set_deferred(true);
      Point rec = Point().sub(moved, gen);                             // recorded: no bytes until asked; then those of q
      CHECK(!rec.have_enc && rec == q && std::memcmp(rec.marshal_binary().data(), want, 32) == 0 && rec.have_enc, "a recorded point gains its bytes when marshalled");
      set_deferred(default_mode);
    }
  }
  // a long-lived key used as a deferred operand keeps a CACHED handle; when the arena drops the node (kyb_defer_floor at the end of a round,
  // the window moving past a leaf) the limbs the point still holds are registered again — no abort, same bytes (KYB_E_STALE)
  {
    Scalar k = Scalar().pick(rand), x = Scalar().pick(rand);
    set_deferred(false);
    Point key = Point().mul(k, nullptr);                               // eager: holds its limbs
    const std::vector<uint8_t> key_bytes = Point(key).marshal_binary();      // (of a copy: `key` itself keeps no bytes, its marshal below must go to the arena)
    const Point eager = Point().add(Point().mul(x, &key), key);
    set_deferred(true);
    Point r1 = Point().add(Point().mul(x, &key), key);                 // key is registered as a leaf, its handle cached
    CHECK(key.pend != 0 && key.have_ge, "the operand caches its handle");
    CHECK(r1.marshal_binary() == eager.marshal_binary(), "deferred result before the floor");
    CHECK(kyb_defer_floor(kyb_defer_mark()) == KYB_OK, "end of the round: everything recorded so far is dropped");
    uint8_t e32[32];
    CHECK(kyb_defer_get(key.pend, nullptr, e32) == KYB_E_STALE, "the cached handle is stale now");
    CHECK(key.marshal_binary() == key_bytes, "marshal of a point with a stale cached handle: its limbs");
    CHECK(key.pend == 0, "the stale handle is forgotten");
    Point again = Point().mul(x, &key); (void)again;
    CHECK(kyb_defer_floor(kyb_defer_mark()) == KYB_OK, "floor");        // `again` is lost (it held no limbs), `key` is not
    Point r2 = Point().add(Point().mul(x, &key), key);                 // stale cached operand handle -> renewed
    CHECK(r2.marshal_binary() == eager.marshal_binary(), "deferred result after the floor");
    CHECK(kyb_defer_floor(kyb_defer_mark()) == KYB_OK, "floor");
    CHECK(key == Point().mul(k, nullptr), "== with a stale cached handle on one side");
    CHECK(Point().neg(key).marshal_binary() == Point().sub(Point().null(), key).marshal_binary(), "neg of a point with a stale handle");
    CHECK(kyb_defer_floor(0) == KYB_OK, "the mark of a thread that recorded nothing");
    set_deferred(default_mode);
  }
  for (const Point& p : points) std::printf("POINT %s\n", p.hex().c_str());
  std::printf("S1 %s\nS2 %s\n", s1.hex().c_str(), s2.hex().c_str());
  std::printf(failures ? "FAILED %d\n" : "OK\n", failures);
  kyb_shutdown();
  return failures ? 1 : 0;
}
