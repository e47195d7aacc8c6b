// The deferred-point evaluator (kyber-rs_amd/csrc/defer.inc) compiled for the CPU with the engine's batch entry points replaced by the oracle, and
// run under AddressSanitizer + UBSan: random expression graphs, Horner chains, addition chains, stale handles, several threads on one arena — every
// answer against the oracle's own eager evaluation.  A test of the HOST LOGIC of the product source (graph walking, chain recognition, batching, arena
// bookkeeping) on a machine without a GPU; the engine calls behind it are exercised on the GPU by tests/test_gpu_deferred.py and test_gpu_vss_round.py.
// Test infrastructure: built and run by tests/test_defer_host.py, nothing outside tests/ uses it.
#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/kyber_ed25519.h"

extern "C" {      // oracle/ed25519_oracle.c
void orc_init(void);
void orc_mul_base(uint8_t out_enc[32], int32_t out_ext[40], const uint8_t scalar[32]);
void orc_mul(uint8_t out_enc[32], int32_t out_ext[40], const uint8_t scalar[32], const int32_t pt_ext[40]);
int orc_decode(int32_t out_ext[40], const uint8_t enc[32]);
void orc_encode(uint8_t enc[32], const int32_t ext[40]);
void orc_add(int32_t out[40], const int32_t a[40], const int32_t b[40], int sub);
void orc_neg(int32_t out[40], const int32_t a[40]);
void orc_null(int32_t out[40]);
void orc_base(int32_t out[40]);
void orc_pubpoly_eval(uint8_t out_enc[32], const int32_t* commits_ext, size_t t, uint32_t index);
}

// ---- the little of the engine that defer.inc refers to ---------------------------------------------------------------------------------
struct DeferArena;
void defer_release(DeferArena* a);
namespace {
struct Ctx {
  bool ready = true;
  int device = 0;
  std::mutex launch_mu;
  std::atomic<DeferArena*> defer{nullptr};
  std::atomic<int> opt_defer_fuse{1};
  std::atomic<int> opt_defer_max_nodes{1 << 18};
  std::atomic<int> opt_defer_keep_mib{256};
};
Ctx g_ctx, g_ctx2;
thread_local Ctx* tl_ctx = &g_ctx;
Ctx* cur() { return tl_ctx; }
thread_local std::string g_err;
int fail(int code, const char* msg) { g_err = msg; return code; }
std::atomic<long> g_calls{0}, g_items{0};
}  // namespace
std::atomic<long> g_projective_flushes{0};
void defer_projective(bool on) { if (on) ++g_projective_flushes; }      // (engine.hip: projective limbs for the calling thread's calls)
#define ENTER() Ctx* ctx_ = cur(); Ctx& g = *ctx_
#define ENTER_HOST() ENTER()

// ---- the batch entry points the evaluator calls, answered by the oracle ----------------------------------------------------------------
extern "C" {
int kyb_mul_base_batch(const uint8_t* s, size_t n, uint8_t* enc, int32_t* ext) {
  ++g_calls; g_items += (long)n;
  for (size_t i = 0; i < n; ++i) orc_mul_base(enc ? enc + 32 * i : nullptr, ext ? ext + 40 * i : nullptr, s + 32 * i);
  return KYB_OK;
}
int kyb_mul_batch(const uint8_t* s, const uint8_t* pe, const int32_t* px, size_t n, uint8_t* enc, int32_t* ext, uint8_t* ok) {
  ++g_calls; g_items += (long)n;
  if (pe != nullptr || ok != nullptr) return fail(KYB_E_BAD_ARG, "the evaluator passes limbs");
  for (size_t i = 0; i < n; ++i) orc_mul(enc ? enc + 32 * i : nullptr, ext ? ext + 40 * i : nullptr, s + 32 * i, px + 40 * i);
  return KYB_OK;
}
int kyb_add_batch(const int32_t* a, const int32_t* b, size_t n, int32_t* out, int sub) {
  ++g_calls; g_items += (long)n;
  for (size_t i = 0; i < n; ++i) orc_add(out + 40 * i, a + 40 * i, b + 40 * i, sub);
  return KYB_OK;
}
int kyb_encode_batch(const int32_t* ext, size_t n, uint8_t* enc) {
  ++g_calls; g_items += (long)n;
  for (size_t i = 0; i < n; ++i) orc_encode(enc + 32 * i, ext + 40 * i);
  return KYB_OK;
}
int kyb_equal_batch(const int32_t* a, const int32_t* b, size_t n, uint8_t* eq) {
  ++g_calls;
  for (size_t i = 0; i < n; ++i) { uint8_t ea[32], eb[32]; orc_encode(ea, a + 40 * i); orc_encode(eb, b + 40 * i); eq[i] = memcmp(ea, eb, 32) == 0; }
  return KYB_OK;
}
int kyb_pubpoly_eval_multi_batch(const int32_t* commits, size_t t, size_t m, const uint32_t* idx, size_t k, uint8_t* enc, int32_t* ext) {
  ++g_calls; g_items += (long)(m * k);
  for (size_t gi = 0; gi < m; ++gi)
    for (size_t j = 0; j < k; ++j) {
      uint8_t e[32];
      orc_pubpoly_eval(e, commits + 40 * t * gi, t, idx[gi * k + j]);
      if (enc) memcpy(enc + 32 * (gi * k + j), e, 32);
      if (ext && !orc_decode(ext + 40 * (gi * k + j), e)) return fail(KYB_E_BAD_ARG, "oracle produced an encoding that does not decode");
    }
  return KYB_OK;
}
int kyb_sum_batch(const int32_t* pts, size_t m, size_t t, uint8_t* enc, int32_t* ext) {
  ++g_calls; g_items += (long)(m * t);
  for (size_t gi = 0; gi < m; ++gi) {
    int32_t acc[40];
    orc_null(acc);
    for (size_t j = 0; j < t; ++j) orc_add(acc, acc, pts + 40 * (gi * t + j), 0);
    if (enc) orc_encode(enc + 32 * gi, acc);
    if (ext) memcpy(ext + 40 * gi, acc, 160);
  }
  return KYB_OK;
}
}

#include "../../kyber-rs_amd/csrc/defer.inc"

// ---- the test --------------------------------------------------------------------------------------------------------------------------
static int failures = 0;
#define CHECK(c, what) do { if (!(c)) { std::printf("FAIL line %d: %s (%s)\n", __LINE__, what, g_err.c_str()); ++failures; } } while (0)
struct Val { uint64_t h; int32_t ext[40]; };
static std::string enc_of(const int32_t* ext) { uint8_t e[32]; orc_encode(e, ext); return std::string((const char*)e, 32); }
static std::string got_enc(uint64_t h) { uint8_t e[32]; if (kyb_defer_get(h, nullptr, e) != KYB_OK) return "error: " + g_err; return std::string((const char*)e, 32); }
static void scalar_small(uint8_t s[32], uint32_t x) { memset(s, 0, 32); memcpy(s, &x, 4); }

// small_window: the graphs run through a window of 48 nodes with the table of kept values on — every path of the eviction logic under random
// shapes (operands behind the window, operands about to leave it, chains cut by the window's edge, flushes forced by it).  A handle may then be
// REFUSED (KYB_E_STALE) — the test keeps handles of leaves and of inner steps of fused chains, which leave no value behind — but every answer
// that is given must be right, and a refused handle is dropped from the pool.
static long g_stale_seen = 0, g_answers = 0;
static void random_graphs(unsigned seed, int steps, bool small_window = false) {
  std::mt19937_64 rng(seed);
  std::vector<Val> nodes;
  auto rnd_scalar = [&](uint8_t s[32]) { for (int i = 0; i < 32; ++i) s[i] = (uint8_t)rng(); if (rng() % 3 == 0) { memset(s + 1, 0, 31); } if (rng() % 4) s[31] &= 0x0f; };
  // rc of a recording / asking call: KYB_OK, or (small window only) a stale operand — which is then forgotten
  auto fine = [&](int rc, std::initializer_list<uint64_t> used) {
    if (rc == KYB_OK) return true;
    if (small_window && rc == KYB_E_STALE && g_err.find("stale") != std::string::npos) {
      ++g_stale_seen;
      // forget the handles that can no longer be used: those of `used` the arena refuses when asked directly
      for (uint64_t h : used) { uint8_t e[32]; if (kyb_defer_get(h, nullptr, e) == KYB_E_STALE) nodes.erase(std::remove_if(nodes.begin(), nodes.end(), [&](const Val& v) { return v.h == h; }), nodes.end()); }
      return false;
    }
    CHECK(false, "a deferred call failed");
    return false;
  };
  auto seed_leaves = [&] {
    for (int i = 0; i < 5; ++i) { Val v; uint8_t s[32]; rnd_scalar(s); orc_mul_base(nullptr, v.ext, s); CHECK(kyb_defer_input(v.ext, &v.h) == KYB_OK, "input"); nodes.push_back(v); }
    { Val v; orc_null(v.ext); CHECK(kyb_defer_null(&v.h) == KYB_OK, "null"); nodes.push_back(v); }
    { Val v; orc_base(v.ext); CHECK(kyb_defer_base(&v.h) == KYB_OK, "base"); nodes.push_back(v); }
  };
  seed_leaves();
  for (int st = 0; st < steps; ++st) {
    if (nodes.size() < 4) seed_leaves();
    const Val a = nodes[rng() % nodes.size()], b = nodes[rng() % nodes.size()];
    uint8_t s[32]; rnd_scalar(s);
    Val v;
    switch (rng() % 7) {
      case 0: orc_mul_base(nullptr, v.ext, s); if (fine(kyb_defer_mul_base(s, &v.h), {})) nodes.push_back(v); break;
      case 1: orc_mul(nullptr, v.ext, s, a.ext); if (fine(kyb_defer_mul(s, a.h, &v.h), {a.h})) nodes.push_back(v); break;
      case 2: orc_add(v.ext, a.ext, b.ext, 0); if (fine(kyb_defer_add(a.h, b.h, 0, &v.h), {a.h, b.h})) nodes.push_back(v); break;
      case 3: orc_add(v.ext, a.ext, b.ext, 1); if (fine(kyb_defer_add(a.h, b.h, 1, &v.h), {a.h, b.h})) nodes.push_back(v); break;
      case 4: orc_neg(v.ext, a.ext); if (fine(kyb_defer_neg(a.h, &v.h), {a.h})) nodes.push_back(v); break;
      case 5: {      // a short Horner chain with a small multiplier on top of a
        uint8_t x[32]; scalar_small(x, (uint32_t)(1 + rng() % 9));
        Val cur = a;
        const int len = 1 + (int)(rng() % 5);
        bool whole = true;
        for (int k = 0; k < len && whole; ++k) {
          if (nodes.empty()) { whole = false; break; }
          const Val c = nodes[rng() % nodes.size()];
          Val m, s2;
          orc_mul(nullptr, m.ext, x, cur.ext);
          if (!fine(kyb_defer_mul(x, cur.h, &m.h), {cur.h})) { whole = false; break; }
          orc_add(s2.ext, m.ext, c.ext, 0);
          if (!fine(rng() & 1 ? kyb_defer_add(m.h, c.h, 0, &s2.h) : kyb_defer_add(c.h, m.h, 0, &s2.h), {m.h, c.h})) { whole = false; break; }
          if (rng() % 4 == 0) nodes.push_back(s2);          // sometimes an inner node is kept and asked for later
          if (rng() % 5 == 0) nodes.push_back(m);           // ... and so is an inner PRODUCT (`m = x v; s = m + c; ...; m.marshal_binary()`)
          if (rng() % 16 == 0) { uint8_t e[32]; if (fine(kyb_defer_get(m.h, nullptr, e), {m.h})) { ++g_answers; CHECK(std::string((const char*)e, 32) == enc_of(m.ext), "inner product asked for while its chain is pending"); } }
          cur = s2;
        }
        if (whole) nodes.push_back(cur);
        break;
      }
      default: {
        const int k = (int)(rng() % 3);
        if (k == 0) { uint8_t e[32]; if (fine(kyb_defer_get(a.h, nullptr, e), {a.h})) { ++g_answers; CHECK(std::string((const char*)e, 32) == enc_of(a.ext), "get"); } }
        else if (k == 1) { uint8_t eq = 2; if (fine(kyb_defer_equal(a.h, b.h, &eq), {a.h, b.h})) { ++g_answers; CHECK((eq != 0) == (enc_of(a.ext) == enc_of(b.ext)), "equal"); } }
        else CHECK(kyb_defer_flush() == KYB_OK, "flush");
      }
    }
  }
  const std::vector<Val> at_end = nodes;
  for (const Val& v : at_end) {
    int32_t ext[40]; uint8_t e[32];
    if (!fine(kyb_defer_get(v.h, ext, e), {v.h})) continue;
    ++g_answers;
    CHECK(std::string((const char*)e, 32) == enc_of(v.ext) && enc_of(ext) == enc_of(v.ext), "final value");
  }
}

int main() {
  orc_init();
  for (unsigned seed = 1; seed <= 12; ++seed) {
    g_ctx.opt_defer_fuse = seed % 4 != 0;
    random_graphs(seed, 220);
  }
  // the same random graphs through a window of 48 nodes (it moves by 12 at a time): kept values, operands taken back in, forced flushes
  g_ctx.opt_defer_max_nodes = 48;
  for (unsigned seed = 101; seed <= 124; ++seed) {
    g_ctx.opt_defer_fuse = seed % 4 != 0;
    random_graphs(seed, 400, true);
  }
  std::printf("small window: %ld answers checked, %ld calls refused as stale\n", g_answers, g_stale_seen);
  CHECK(g_answers > 3000 && g_stale_seen * 4 < g_answers, "through a small window most handles stay good, and every answer given is right");
  // ... and with a table of 40 values behind it (second-chance eviction at every step): more handles are refused, no answer is wrong
  {
    const long a0 = g_answers, s0 = g_stale_seen;
    g_ctx.opt_defer_keep_mib = -40;
    for (unsigned seed = 201; seed <= 216; ++seed) { g_ctx.opt_defer_fuse = seed % 4 != 0; random_graphs(seed, 400, true); }
    g_ctx.opt_defer_keep_mib = 256;
    uint64_t st[12]; kyb_defer_stats(st, 12);
    std::printf("small window, 40 kept values: %ld answers checked, %ld calls refused as stale, %llu values pushed out\n", g_answers - a0, g_stale_seen - s0, (unsigned long long)st[9]);
    CHECK(g_answers - a0 > 800 && st[9] > 500 && st[8] <= 40, "a tiny table: bounded, evicting, never wrong");
  }
  CHECK(kyb_defer_floor(kyb_defer_mark()) == KYB_OK, "floor");
  g_ctx.opt_defer_max_nodes = 1 << 18;
  g_ctx.opt_defer_fuse = 1;
  // PubPoly::eval: one call; six chains: still one call
  {
    const size_t t = 17;
    std::vector<int32_t> commits(40 * t);
    std::vector<uint64_t> hc(t);
    for (size_t j = 0; j < t; ++j) { uint8_t s[32]; scalar_small(s, 1000 + (uint32_t)j); orc_mul_base(nullptr, &commits[40 * j], s); CHECK(kyb_defer_input(&commits[40 * j], &hc[j]) == KYB_OK, "commit"); }
    auto record = [&](uint32_t index) {
      uint8_t x[32]; scalar_small(x, index + 1);
      uint64_t v; CHECK(kyb_defer_null(&v) == KYB_OK, "null");
      for (size_t j = t; j-- > 0;) { uint64_t m, a; CHECK(kyb_defer_mul(x, v, &m) == KYB_OK && kyb_defer_add(m, hc[j], 0, &a) == KYB_OK, "horner"); v = a; }
      return v;
    };
    uint64_t st0[8], st1[8];
    kyb_defer_stats(st0, 8);
    long c0 = g_calls;
    std::vector<uint64_t> hs;
    for (uint32_t i : {0u, 5u, 63u, 4000000000u}) hs.push_back(record(i));
    CHECK(kyb_defer_flush() == KYB_OK, "flush");
    kyb_defer_stats(st1, 8);
    CHECK(g_calls - c0 == 1 && st1[3] - st0[3] == 4, "four chains of one length: one engine call");
    size_t n = 0;
    for (uint32_t i : {0u, 5u, 63u, 4000000000u}) { uint8_t e[32]; orc_pubpoly_eval(e, commits.data(), t, i); CHECK(got_enc(hs[n++]) == std::string((const char*)e, 32), "horner value"); }
  }
  // an inner product of a pending Horner chain asked for by name, first as limbs+bytes, then through equal(): it ends the chain below it, it is
  // evaluated, and the chain above it is still right (defer.inc is_step)
  for (int how = 0; how < 3; ++how) {
    const size_t t = 5;
    uint8_t x[32]; scalar_small(x, 7);
    std::vector<Val> c(t), prod, sums;
    for (size_t j = 0; j < t; ++j) { uint8_t s[32]; scalar_small(s, 31 + (uint32_t)j + 100 * how); orc_mul_base(nullptr, c[j].ext, s); CHECK(kyb_defer_input(c[j].ext, &c[j].h) == KYB_OK, "coeff"); }
    Val v = c[t - 1];
    for (size_t j = t - 1; j-- > 0;) {
      Val m, a;
      orc_mul(nullptr, m.ext, x, v.ext); CHECK(kyb_defer_mul(x, v.h, &m.h) == KYB_OK, "chain mul");
      orc_add(a.ext, m.ext, c[j].ext, 0); CHECK(kyb_defer_add(m.h, c[j].h, 0, &a.h) == KYB_OK, "chain add");
      prod.push_back(m); sums.push_back(a); v = a;
    }
    const Val& inner = prod[1];
    if (how == 0) { int32_t ext[40]; uint8_t e[32]; CHECK(kyb_defer_get(inner.h, ext, e) == KYB_OK && enc_of(ext) == enc_of(inner.ext) && std::string((const char*)e, 32) == enc_of(inner.ext), "inner product: get"); }
    if (how == 1) { uint8_t eq = 2; CHECK(kyb_defer_equal(inner.h, prod[2].h, &eq) == KYB_OK && eq == 0 && kyb_defer_equal(inner.h, inner.h, &eq) == KYB_OK && eq == 1, "inner product: equal"); }
    if (how == 2) { CHECK(got_enc(v.h) == enc_of(v.ext), "sink first"); }          // the chain is evaluated as a whole: the inner nodes are dead, and asked for afterwards
    for (const Val& p : prod) CHECK(got_enc(p.h) == enc_of(p.ext), "every product of the chain");
    for (const Val& p : sums) CHECK(got_enc(p.h) == enc_of(p.ext), "every sum of the chain");
  }
  // DSS::process_partial_sig (dss_sig.rs:266-273): two Horner chains whose results are NOT sinks — long_poly.eval(i) goes into a multiplication by a
  // 253-bit scalar, random_poly.eval(i) into an addition with that product, and only the comparison asks.  Both chains are still one evaluation
  // call; then one multiplication, one addition (the big multiplier is no share index: no chain there), and a sum fed by a chain result keeps it a leaf
  for (int shape = 0; shape < 2; ++shape) {
    const size_t t = 11;
    std::vector<Val> lc(t), rc(t);
    for (size_t j = 0; j < t; ++j) {
      uint8_t s[32];
      scalar_small(s, 600 + (uint32_t)j + 50 * shape); orc_mul_base(nullptr, lc[j].ext, s); CHECK(kyb_defer_input(lc[j].ext, &lc[j].h) == KYB_OK, "long commit");
      scalar_small(s, 900 + (uint32_t)j + 50 * shape); orc_mul_base(nullptr, rc[j].ext, s); CHECK(kyb_defer_input(rc[j].ext, &rc[j].h) == KYB_OK, "random commit");
    }
    const uint32_t index = 37;
    auto eval = [&](const std::vector<Val>& c) {
      uint8_t x[32]; scalar_small(x, index + 1);
      Val v; orc_null(v.ext); CHECK(kyb_defer_null(&v.h) == KYB_OK, "null");
      for (size_t j = t; j-- > 0;) {
        Val m, a;
        orc_mul(nullptr, m.ext, x, v.ext); orc_add(a.ext, m.ext, c[j].ext, 0);
        CHECK(kyb_defer_mul(x, v.h, &m.h) == KYB_OK && kyb_defer_add(m.h, c[j].h, 0, &a.h) == KYB_OK, "eval");
        v = a;
      }
      return v;
    };
    uint64_t st0[8], st1[8];
    kyb_defer_stats(st0, 8);
    const long c0 = g_calls;
    Val rand_share = eval(rc), long_share = eval(lc);
    uint8_t hs[32]; for (int i = 0; i < 32; ++i) hs[i] = (uint8_t)(0x5b + 29 * i); hs[31] &= 0x0f;
    Val right, right2;
    orc_mul(nullptr, right.ext, hs, long_share.ext); CHECK(kyb_defer_mul(hs, long_share.h, &right.h) == KYB_OK, "hash * long_share");
    orc_add(right2.ext, rand_share.ext, right.ext, 0); CHECK(kyb_defer_add(rand_share.h, right.h, 0, &right2.h) == KYB_OK, "rand_share + ...");
    // partial * B (dss_sig.rs:271): nothing waits for it, so it rides in the call that multiplies by the challenge — no call of its own
    Val left; uint8_t ps[32]; for (int i = 0; i < 32; ++i) ps[i] = (uint8_t)(0xc3 + 7 * i + shape); ps[31] = shape ? 0x9e : 0x0e;      // (0x9e: above 2^255, the top-digit quirk)
    orc_mul_base(nullptr, left.ext, ps); CHECK(kyb_defer_mul_base(ps, &left.h) == KYB_OK, "partial * B");
    Val top = right2;
    if (shape == 1) {      // two more terms on top: a sum chain whose innermost term is the chain result
      for (int k = 0; k < 2; ++k) { Val nx; orc_add(nx.ext, top.ext, lc[k].ext, 0); CHECK(kyb_defer_add(top.h, lc[k].h, 0, &nx.h) == KYB_OK, "more terms"); top = nx; }
    }
    CHECK(got_enc(top.h) == enc_of(top.ext), "the value the comparison looks at");
    kyb_defer_stats(st1, 8);
    CHECK(st1[3] - st0[3] == 2, "both evaluations fused although neither is a sink");
    CHECK(g_calls - c0 == (shape == 0 ? 4 : 3), "one evaluation call for both chains, one multiplication, then an addition and its marshal / one sum");
    CHECK(got_enc(rand_share.h) == enc_of(rand_share.ext) && got_enc(long_share.h) == enc_of(long_share.ext) && got_enc(right.h) == enc_of(right.ext), "the results in between");
    CHECK(got_enc(left.h) == enc_of(left.ext) && g_calls - c0 == (shape == 0 ? 4 : 3), "the fixed-base product rode along (and everything was marshalled on the way)");
  }
  // Point::eq on a sum: no bytes are asked of that flush (its kernels hand over projective limbs, the comparison is the projective one); on two
  // kernel results: the encodings, as before
  {
    uint8_t s1[32], s2[32], s3[32];
    scalar_small(s1, 1234567); scalar_small(s2, 7654321); scalar_small(s3, 1234567 + 7654321);
    Val a, b, sum, direct;
    orc_mul_base(nullptr, a.ext, s1); orc_mul_base(nullptr, b.ext, s2); orc_mul_base(nullptr, direct.ext, s3); orc_add(sum.ext, a.ext, b.ext, 0);
    CHECK(kyb_defer_mul_base(s1, &a.h) == KYB_OK && kyb_defer_mul_base(s2, &b.h) == KYB_OK && kyb_defer_mul_base(s3, &direct.h) == KYB_OK, "products");
    CHECK(kyb_defer_add(a.h, b.h, 0, &sum.h) == KYB_OK, "sum");
    const long p0 = g_projective_flushes, c0 = g_calls;
    uint8_t eq = 2;
    CHECK(kyb_defer_equal(direct.h, sum.h, &eq) == KYB_OK && eq == 1, "s1 B + s2 B == (s1 + s2) B");
    CHECK(g_projective_flushes - p0 == 1 && g_calls - c0 == 3, "a comparison with a sum: one projective flush — products, addition, comparison");
    CHECK(kyb_defer_equal(a.h, sum.h, &eq) == KYB_OK && eq == 0, "and an unequal pair, from the values at hand");
    CHECK(got_enc(sum.h) == enc_of(sum.ext) && got_enc(a.h) == enc_of(a.ext), "bytes asked for afterwards");
    Val c, d;
    scalar_small(s1, 99); orc_mul_base(nullptr, c.ext, s1); orc_mul_base(nullptr, d.ext, s1);
    CHECK(kyb_defer_mul_base(s1, &c.h) == KYB_OK && kyb_defer_mul_base(s1, &d.h) == KYB_OK, "two kernel results");
    const long p1 = g_projective_flushes, c1 = g_calls;
    CHECK(kyb_defer_equal(c.h, d.h, &eq) == KYB_OK && eq == 1 && g_projective_flushes == p1 && g_calls - c1 == 1, "two kernel results: their encodings, one call");
  }
  // Rabin's verify_deal (vss/rabin/vss.rs: fi G + gi H == commit_poly.eval(i)): a fixed-base product riding with the variable-base one, their sum,
  // one Horner chain, one comparison whose left operand is a plain sum (so: nothing marshalled, projective comparison) — four engine calls
  {
    const size_t t = 9;
    uint8_t hs[32]; scalar_small(hs, 424242);
    Val H; orc_mul_base(nullptr, H.ext, hs); CHECK(kyb_defer_input(H.ext, &H.h) == KYB_OK, "H");
    // commitments c_j = f_j G + g_j H, share (f(i), g(i)) at index i
    const uint32_t index = 5;
    std::vector<Val> commits(t);
    uint64_t fi = 0, gi = 0, xp = 1;
    for (size_t j = 0; j < t; ++j) {
      const uint32_t fj = 1000 + 7 * (uint32_t)j, gj = 2000 + 11 * (uint32_t)j;
      uint8_t a[32], b[32]; scalar_small(a, fj); scalar_small(b, gj);
      int32_t fg[40], gh[40];
      orc_mul_base(nullptr, fg, a); orc_mul(nullptr, gh, b, H.ext); orc_add(commits[j].ext, fg, gh, 0);
      CHECK(kyb_defer_input(commits[j].ext, &commits[j].h) == KYB_OK, "commitment");
      fi += fj * xp; gi += gj * xp; xp *= (index + 1);          // (index + 1)^8 * 2100 < 2^32: no reduction needed
    }
    uint8_t fs[32] = {0}, gs[32] = {0}; memcpy(fs, &fi, 8); memcpy(gs, &gi, 8);
    const long c0 = g_calls, p0 = g_projective_flushes;
    uint64_t fig, gih, ci;
    CHECK(kyb_defer_mul_base(fs, &fig) == KYB_OK && kyb_defer_mul(gs, H.h, &gih) == KYB_OK && kyb_defer_add(fig, gih, 0, &ci) == KYB_OK, "fi G + gi H");
    uint8_t x[32]; scalar_small(x, index + 1);
    uint64_t v; CHECK(kyb_defer_null(&v) == KYB_OK, "null");
    for (size_t j = t; j-- > 0;) { uint64_t m, a; CHECK(kyb_defer_mul(x, v, &m) == KYB_OK && kyb_defer_add(m, commits[j].h, 0, &a) == KYB_OK, "eval"); v = a; }
    uint8_t eq = 2;
    CHECK(kyb_defer_equal(ci, v, &eq) == KYB_OK && eq == 1, "the deal verifies");
    CHECK(g_calls - c0 == 4 && g_projective_flushes - p0 == 1, "products in one call, the chain in one, the sum, the comparison");
  }
  // recover_commit: one batch of products, one sum
  {
    const size_t t = 9;
    uint64_t acc; CHECK(kyb_defer_null(&acc) == KYB_OK, "null");
    int32_t want[40]; orc_null(want);
    long c0 = g_calls;
    for (size_t j = 0; j < t; ++j) {
      uint8_t s[32]; for (int i = 0; i < 32; ++i) s[i] = (uint8_t)(37 * j + i); s[31] &= 0x0f;
      int32_t p[40], prod[40]; uint8_t sp[32]; scalar_small(sp, 77 + (uint32_t)j); orc_mul_base(nullptr, p, sp);
      uint64_t hp, hm, ha;
      CHECK(kyb_defer_input(p, &hp) == KYB_OK && kyb_defer_mul(s, hp, &hm) == KYB_OK && kyb_defer_add(acc, hm, 0, &ha) == KYB_OK, "lincomb");
      acc = ha;
      orc_mul(nullptr, prod, s, p); orc_add(want, want, prod, 0);
    }
    CHECK(got_enc(acc) == enc_of(want) && g_calls - c0 == 2, "sum chain: two engine calls");
  }
  // a leaf handed over WITH its bytes (a point unmarshalled from its canonical encoding): comparing it with evaluated points costs no engine call
  // — the search of the own key among the participants' keys (dss_sig.rs:180-190) — and its marshal is a cache hit
  {
    const size_t n = 12;
    std::vector<Val> keys(n);
    std::vector<std::string> key_enc(n);
    for (size_t j = 0; j < n; ++j) {
      uint8_t sc[32]; scalar_small(sc, 60000 + (uint32_t)j);
      uint8_t e[32]; orc_mul_base(e, keys[j].ext, sc); key_enc[j] = std::string((const char*)e, 32);
      CHECK(kyb_defer_input_enc(keys[j].ext, e, &keys[j].h) == KYB_OK, "leaf with bytes");
    }
    uint8_t sc[32]; scalar_small(sc, 60000 + 7);
    uint64_t own; CHECK(kyb_defer_mul_base(sc, &own) == KYB_OK, "own key");
    uint8_t eq = 0;
    CHECK(kyb_defer_equal(keys[0].h, own, &eq) == KYB_OK && eq == 0, "first comparison evaluates the own key (with its bytes)");
    const long c0 = g_calls;
    size_t found = n;
    for (size_t j = 0; j < n; ++j) { CHECK(kyb_defer_equal(keys[j].h, own, &eq) == KYB_OK, "search"); if (eq) { found = j; break; } }
    CHECK(found == 7 && g_calls == c0, "the search itself: byte comparisons, not one engine call");
    CHECK(got_enc(keys[3].h) == key_enc[3] && g_calls == c0, "the marshal of such a leaf is a hit");
    uint64_t again; CHECK(kyb_defer_input(keys[5].ext, &again) == KYB_OK && again == keys[5].h, "the same limbs without bytes: the same leaf");
    Val plain; uint8_t e2[32]; scalar_small(sc, 61000); orc_mul_base(e2, plain.ext, sc);
    CHECK(kyb_defer_input(plain.ext, &plain.h) == KYB_OK && kyb_defer_input_enc(plain.ext, e2, &again) == KYB_OK && again == plain.h, "bytes that arrive later are taken");
    CHECK(kyb_defer_equal(plain.h, own, &eq) == KYB_OK && eq == 0 && g_calls == c0, "... and used");
  }
  // a small window and NO table of kept values (defer.keep_mib = 0): old handles become stale, never wrong; floor / mark
  {
    g_ctx.opt_defer_max_nodes = 32;
    g_ctx.opt_defer_keep_mib = 0;
    uint8_t s[32]; scalar_small(s, 5);
    uint64_t first, h;
    CHECK(kyb_defer_mul_base(s, &first) == KYB_OK, "first");
    for (uint32_t i = 0; i < 100; ++i) { scalar_small(s, i); CHECK(kyb_defer_mul_base(s, &h) == KYB_OK, "fill"); }
    uint8_t e[32];
    CHECK(kyb_defer_get(first, nullptr, e) == KYB_E_STALE && g_err.find("stale") != std::string::npos, "stale handle refused");
    CHECK(kyb_defer_get(h, nullptr, e) == KYB_OK, "newest handle");
    const uint64_t mark = kyb_defer_mark();
    scalar_small(s, 7);
    CHECK(kyb_defer_mul_base(s, &h) == KYB_OK && kyb_defer_floor(mark) == KYB_OK, "floor");
    uint64_t st[8]; kyb_defer_stats(st, 8);
    uint8_t w[32]; orc_mul_base(w, nullptr, s);
    CHECK(st[6] == 1 && kyb_defer_get(h, nullptr, e) == KYB_OK && memcmp(e, w, 32) == 0, "what was recorded after the mark survives");
    CHECK(kyb_defer_get(0, nullptr, e) == KYB_E_BAD_ARG && kyb_defer_get(h + 1000, nullptr, e) == KYB_E_BAD_ARG, "null / unknown handles");
    g_ctx.opt_defer_max_nodes = 1 << 18;
    g_ctx.opt_defer_keep_mib = 256;
  }
  // The same small window WITH the table (the default): a handle-only client — what the Rust binding's Copy point is: no limbs kept, no
  // materialize, no floor — runs 300 "rounds" through a window of 256 nodes while it holds a long-lived recorded point (a distributed key: the
  // sum of the first commitments of the first round's dealers), marshals it every round and multiplies with it.  Nothing it holds goes stale.
  {
    g_ctx.opt_defer_max_nodes = 256;
    const uint64_t mark0 = kyb_defer_mark();
    uint64_t st0[12]; kyb_defer_stats(st0, 12);
    std::mt19937_64 rng(77);
    const int n = 6, t = 4, rounds = 300;
    auto rnd = [&](uint8_t sc[32]) { for (int i = 0; i < 32; ++i) sc[i] = (uint8_t)rng(); sc[31] &= 0x0f; };
    uint64_t key = 0; int32_t key_ext[40]; orc_null(key_ext);
    struct Held { uint64_t h; int32_t ext[40]; };
    std::vector<Held> old_commits;      // looked at once when made, then only held: round 0's are asked for again at the very end
    long stale = 0, wrong = 0;
    for (int r = 0; r < rounds; ++r) {
      // every dealer commits to a polynomial; a receiver evaluates each public polynomial at its index (Horner: fused) and checks g^share
      std::vector<std::vector<Held>> commits((size_t)n);
      for (int d = 0; d < n; ++d)
        for (int j = 0; j < t; ++j) { Held c; uint8_t sc[32]; rnd(sc); orc_mul_base(nullptr, c.ext, sc); if (kyb_defer_mul_base(sc, &c.h) != KYB_OK) ++stale; commits[(size_t)d].push_back(c); }
      for (int d = 0; d < n; ++d) { if (got_enc(commits[(size_t)d][0].h) != enc_of(commits[(size_t)d][0].ext)) ++wrong; }      // marshalled as they are sent
      if (r == 0) {
        CHECK(kyb_defer_null(&key) == KYB_OK, "key: null");
        for (int d = 0; d < n; ++d) { uint64_t nk; if (kyb_defer_add(key, commits[(size_t)d][0].h, 0, &nk) != KYB_OK) ++stale; key = nk; orc_add(key_ext, key_ext, commits[(size_t)d][0].ext, 0); }
        for (int d = 0; d < n; ++d) old_commits.push_back(commits[(size_t)d][1]);
        for (const Held& c : old_commits) if (got_enc(c.h) != enc_of(c.ext)) ++wrong;
      }
      for (int d = 0; d < n; ++d) {
        uint8_t x[32]; scalar_small(x, 1 + (uint32_t)(r % 5));
        uint64_t v; int32_t want[40];
        if (kyb_defer_null(&v) != KYB_OK) ++stale;
        orc_null(want);
        for (int j = t; j-- > 0;) {
          uint64_t m, a;
          if (kyb_defer_mul(x, v, &m) != KYB_OK || kyb_defer_add(m, commits[(size_t)d][(size_t)j].h, 0, &a) != KYB_OK) ++stale;
          v = a;
          int32_t prod[40]; orc_mul(nullptr, prod, x, want); orc_add(want, prod, commits[(size_t)d][(size_t)j].ext, 0);
        }
        uint8_t eq = 0; uint64_t hw;
        if (kyb_defer_input(want, &hw) != KYB_OK || kyb_defer_equal(v, hw, &eq) != KYB_OK) ++stale; else if (!eq) ++wrong;
      }
      // the long-lived point: marshalled every round, and an operand of a fresh multiplication (a DSS-like use of the distributed key)
      if (got_enc(key) != enc_of(key_ext)) ++wrong;
      uint8_t sc[32]; rnd(sc);
      uint64_t prod; int32_t want[40]; orc_mul(nullptr, want, sc, key_ext);
      if (kyb_defer_mul(sc, key, &prod) != KYB_OK) ++stale; else if (got_enc(prod) != enc_of(want)) ++wrong;
    }
    uint64_t st1[12]; kyb_defer_stats(st1, 12);
    CHECK(stale == 0 && wrong == 0, "300 rounds through a window of 256: no stale handle, no wrong answer");
    CHECK(st1[0] - st0[0] > 20000 && st1[7] - st0[7] > 19000 && st1[6] <= 256, "the window moved past (almost) everything recorded");
    CHECK(st1[8] > 1000 && st1[10] - st0[10] >= 2 * (uint64_t)rounds - 8 && st1[11] - st0[11] >= (uint64_t)rounds - 4, "the key was served from the table and taken back in as an operand");
    for (const Held& c : old_commits) CHECK(got_enc(c.h) == enc_of(c.ext), "a value last looked at 299 rounds ago");
    // a comparison between a kept value and a young node, and between two kept values
    uint8_t eq = 0; uint64_t again;
    CHECK(kyb_defer_input(key_ext, &again) == KYB_OK && kyb_defer_equal(key, again, &eq) == KYB_OK && eq == 1, "kept == young");
    CHECK(kyb_defer_equal(old_commits[0].h, old_commits[1].h, &eq) == KYB_OK && eq == 0 && kyb_defer_equal(old_commits[2].h, old_commits[2].h, &eq) == KYB_OK && eq == 1, "kept == kept");
    // an inner step of a fused chain, asked for BY NAME when its operands are already behind the window: evaluated from their kept values
    {
      std::vector<Held> cs(4);
      for (auto& c : cs) { uint8_t sc[32]; rnd(sc); orc_mul_base(nullptr, c.ext, sc); CHECK(kyb_defer_mul_base(sc, &c.h) == KYB_OK, "coefficient"); }
      for (uint32_t i = 0; i < 100; ++i) { uint8_t sc[32]; uint64_t h; scalar_small(sc, 700000 + i); CHECK(kyb_defer_mul_base(sc, &h) == KYB_OK, "filler"); }
      uint8_t x[32]; scalar_small(x, 3);
      uint64_t v, inner = 0; int32_t want[40], inner_want[40];
      CHECK(kyb_defer_null(&v) == KYB_OK, "null"); orc_null(want);
      for (int j = 4; j-- > 0;) {
        uint64_t m, a;
        CHECK(kyb_defer_mul(x, v, &m) == KYB_OK && kyb_defer_add(m, cs[(size_t)j].h, 0, &a) == KYB_OK, "horner");
        v = a;
        int32_t prod[40]; orc_mul(nullptr, prod, x, want); orc_add(want, prod, cs[(size_t)j].ext, 0);
        if (j == 2) { inner = a; memcpy(inner_want, want, 160); }
      }
      uint64_t sth[12]; kyb_defer_stats(sth, 12);
      CHECK(got_enc(v) == enc_of(want), "the chain's end");
      uint64_t sti[12]; kyb_defer_stats(sti, 12);
      CHECK(sti[3] == sth[3] + 1, "evaluated as ONE chain: its inner steps have no value");
      for (uint32_t i = 0; i < 150; ++i) { uint8_t sc[32]; uint64_t h; scalar_small(sc, 800000 + i); CHECK(kyb_defer_mul_base(sc, &h) == KYB_OK, "filler"); }
      uint64_t stj[12]; kyb_defer_stats(stj, 12);
      uint8_t e32[32];
      CHECK((cs[3].h & 0xffffffffffull) < (kyb_defer_mark() & 0xffffffffffull) - stj[6], "the coefficients are behind the window");
      CHECK((inner & 0xffffffffffull) >= (kyb_defer_mark() & 0xffffffffffull) - stj[6], "the inner step is still inside it");
      CHECK(got_enc(inner) == enc_of(inner_want), "an inner step asked for by name: its operands come from the kept values");
      (void)e32;
    }
    // the table is bounded: at 1 MiB (5,041 values) the untouched go, what is touched every so often stays
    g_ctx.opt_defer_keep_mib = 1;
    for (uint32_t i = 0; i < 12000; ++i) {
      uint8_t sc[32], e[32]; scalar_small(sc, 900000 + i);
      uint64_t h; CHECK(kyb_defer_mul_base(sc, &h) == KYB_OK, "filler");
      if (i % 64 == 63) { CHECK(kyb_defer_flush() == KYB_OK, "filler flush"); CHECK(kyb_defer_get(key, nullptr, e) == KYB_OK, "the key, touched now and then"); }
    }
    uint64_t st2[12]; kyb_defer_stats(st2, 12);
    uint8_t e32[32];
    CHECK(st2[8] <= 5041 && st2[9] > 5000, "bounded by defer.keep_mib");
    CHECK(got_enc(key) == enc_of(key_ext), "touched values stay");
    CHECK(kyb_defer_get(old_commits[0].h, nullptr, e32) == KYB_E_STALE && g_err.find("stale") != std::string::npos, "untouched values went first: refused, not answered wrongly");
    // the host's own statement ends everything older, kept values included
    CHECK(kyb_defer_floor(kyb_defer_mark()) == KYB_OK && kyb_defer_get(key, nullptr, e32) == KYB_E_STALE, "floor drops kept values too");
    uint64_t st3[12]; kyb_defer_stats(st3, 12);
    CHECK(st3[8] == 0 && st3[6] == 0, "nothing left");
    g_ctx.opt_defer_max_nodes = 1 << 18;
    g_ctx.opt_defer_keep_mib = 256;
    (void)mark0;
  }
  // the arena's storage: headers in chunks of 4,096, payloads in slabs of 256, the leaf table doubling — crossed in every direction
  {
    const uint64_t mark0 = kyb_defer_mark();
    uint64_t st0[8]; kyb_defer_stats(st0, 8);
    // 1,500 distinct leaves (the table grows 1,024 -> 2,048 -> 4,096), each registered twice: the second time is a hit
    std::vector<Val> leaves(1500);
    for (size_t i = 0; i < leaves.size(); ++i) { uint8_t sc[32]; scalar_small(sc, 50000 + (uint32_t)i); orc_mul_base(nullptr, leaves[i].ext, sc); CHECK(kyb_defer_input(leaves[i].ext, &leaves[i].h) == KYB_OK, "leaf"); }
    for (size_t i = 0; i < leaves.size(); ++i) { uint64_t again = 0; CHECK(kyb_defer_input(leaves[i].ext, &again) == KYB_OK && again == leaves[i].h, "a leaf seen before keeps its handle (table grown meanwhile)"); }
    // one chain of 9,000 additions over them (three chunks of headers, no payload for the inner sums), asked at the end and in the middle
    Val acc = leaves[0];
    std::vector<Val> kept;
    for (size_t i = 1; i <= 9000; ++i) {
      const Val& l = leaves[i % leaves.size()];
      Val nx; orc_add(nx.ext, acc.ext, l.ext, 0);
      CHECK(kyb_defer_add(acc.h, l.h, 0, &nx.h) == KYB_OK, "long chain");
      acc = nx;
      if (i % 2500 == 0) kept.push_back(acc);
    }
    uint64_t st1[8]; kyb_defer_stats(st1, 8);
    CHECK(st1[6] - st0[6] == 1500 + 9000, "nodes held");
    CHECK(got_enc(kept[1].h) == enc_of(kept[1].ext), "a sum in the middle of the chain, asked first");
    CHECK(got_enc(acc.h) == enc_of(acc.ext), "the end of the chain");
    for (const Val& k : kept) CHECK(got_enc(k.h) == enc_of(k.ext), "kept sums");
    // a floor in the middle of a chunk, then across chunks; handles on either side
    CHECK(kyb_defer_floor(kept[0].h) == KYB_OK, "floor at a node handle (a mark is a handle)");
    uint8_t e32[32];
    CHECK(kyb_defer_get(leaves[7].h, nullptr, e32) == KYB_E_STALE && kyb_defer_get(kept[0].h, nullptr, e32) == KYB_OK, "floor: below dropped, at and above kept");
    CHECK(got_enc(acc.h) == enc_of(acc.ext), "values above the floor survive it");
    CHECK(kyb_defer_floor(kyb_defer_mark()) == KYB_OK, "floor at the end");
    uint64_t st2[8]; kyb_defer_stats(st2, 8);
    CHECK(st2[6] == 0 && kyb_defer_get(acc.h, nullptr, e32) == KYB_E_STALE, "empty arena");
    // a small cap across the chunk boundary: 10,000 nodes through a window of 100
    g_ctx.opt_defer_max_nodes = 100;
    uint64_t h = 0;
    uint8_t sc[32];
    for (uint32_t i = 0; i < 10000; ++i) { scalar_small(sc, 7 + i % 3); CHECK(kyb_defer_mul_base(sc, &h) == KYB_OK, "window"); if (i % 50 == 49) CHECK(kyb_defer_flush() == KYB_OK, "window flush"); }
    uint8_t w[32]; scalar_small(sc, 7 + 9999 % 3); orc_mul_base(w, nullptr, sc);
    uint64_t st3[8]; kyb_defer_stats(st3, 8);
    CHECK(kyb_defer_get(h, nullptr, e32) == KYB_OK && memcmp(e32, w, 32) == 0 && st3[6] <= 100 && st3[6] >= 75 && st3[7] - st2[7] == 10000 - st3[6], "the newest of 10,000 through a window of 100 (which moves by quarters)");
    g_ctx.opt_defer_max_nodes = 1 << 18;
    (void)mark0;
  }
  // four threads on the one arena — with the default window, then through a window of 64 nodes (their handles are answered from the kept values)
  for (int small = 0; small < 2; ++small) {
    g_ctx.opt_defer_max_nodes = small ? 64 : 1 << 18;
    std::vector<std::thread> th;
    std::atomic<int> bad{0};
    for (int i = 0; i < 4; ++i)
      th.emplace_back([i, &bad] {
        for (int r = 0; r < 20; ++r) {
          uint64_t hs[12];
          for (uint32_t j = 0; j < 12; ++j) { uint8_t s[32]; scalar_small(s, 100 * (uint32_t)i + j + 1); if (kyb_defer_mul_base(s, &hs[j]) != KYB_OK) ++bad; }
          for (uint32_t j = 0; j < 12; ++j) { uint8_t s[32], e[32], w[32]; scalar_small(s, 100 * (uint32_t)i + j + 1); orc_mul_base(w, nullptr, s); if (kyb_defer_get(hs[j], nullptr, e) != KYB_OK || memcmp(e, w, 32)) ++bad; }
        }
      });
    for (auto& t : th) t.join();
    CHECK(bad == 0, small ? "threads through a small window" : "threads");
    g_ctx.opt_defer_max_nodes = 1 << 18;
  }
  // a handle names its arena: a point recorded through one context is read, and used as an operand, through another — also after the first context is gone
  {
    uint8_t s1[32], s2[32], e[32], w[32];
    scalar_small(s1, 4242); scalar_small(s2, 77);
    uint64_t h1, h2, h3, mine;
    CHECK(kyb_defer_mul_base(s1, &h1) == KYB_OK, "recorded in the first arena");
    tl_ctx = &g_ctx2;
    CHECK(kyb_defer_mul_base(s1, &mine) == KYB_OK && (mine >> 40) != (h1 >> 40), "handles of two arenas never collide");
    CHECK(kyb_defer_mul(s2, h1, &h2) == KYB_OK && kyb_defer_add(h2, h1, 0, &h3) == KYB_OK, "foreign operands");
    int32_t p1[40], p2[40], p3[40];
    orc_mul_base(nullptr, p1, s1); orc_mul(nullptr, p2, s2, p1); orc_add(p3, p2, p1, 0);
    CHECK(got_enc(h3) == enc_of(p3) && got_enc(h1) == enc_of(p1), "values across arenas");
    uint8_t eq = 0;
    CHECK(kyb_defer_equal(h1, mine, &eq) == KYB_OK && eq == 1 && kyb_defer_equal(h1, h3, &eq) == KYB_OK && eq == 0, "equal across arenas");
    // the first context goes away with a point still only recorded: its arena is an orphan, the point is still there
    uint64_t late;
    tl_ctx = &g_ctx;
    scalar_small(s1, 999);
    CHECK(kyb_defer_mul_base(s1, &late) == KYB_OK, "recorded, never asked for");
    defer_release(g_ctx.defer.exchange(nullptr));
    tl_ctx = &g_ctx2;
    orc_mul_base(w, nullptr, s1);
    CHECK(kyb_defer_get(late, nullptr, e) == KYB_OK && memcmp(e, w, 32) == 0, "an orphaned arena is evaluated by whoever asks");
    CHECK(kyb_defer_get(((uint64_t)0xabcdef << 40) | 5, nullptr, e) == KYB_E_BAD_ARG, "a handle of no arena");
    tl_ctx = &g_ctx;
  }
  std::printf("%s: %ld engine calls for %ld items\n", failures ? "FAILED" : "OK", g_calls.load(), g_items.load());
  return failures ? 1 : 0;
}
