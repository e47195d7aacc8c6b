// The C ABI's element-level entry points answered by the CPU oracle (oracle/ed25519_oracle.c), one item after the other on the calling thread:
// what tests/cpp/test_vss_round.cpp and test_dkg_finish.cpp link against INSTEAD of libkyber_ed25519_hip.so to time the very same call-by-call
// sequences on one host core — the "cpu_port_ms" column beside the engine's eager and deferred columns (round-4 review: the CPU figure for a
// dealer round was a sum of per-operation estimates, never run).  The C port follows the reference's algorithm and limb schedule (radix 2^25.5,
// signed radix-16 windows), so its timings stand for the reference's CPU path as far as anything in this image can.
// TEST INFRASTRUCTURE: lives under tests/, is never linked into the product, and refuses the deferred entry points (the CPU runs eagerly) —
// unless built with -DKYB_CPU_PORT_DEFER, which puts the product's own evaluator (csrc/defer.inc) on top of these entry points (end of this file).
#include <cstdint>
#include <cstdio>
#include <cstring>

#include "../../include/kyber_ed25519.h"

extern "C" {
void orc_init(void);
void orc_mul_base(uint8_t out_enc[32], int32_t out_ext[40], const uint8_t scalar[32]);
void orc_mul(uint8_t out_enc[32], int32_t out_ext[40], const uint8_t scalar[32], const int32_t pt_ext[40]);
int orc_decode(int32_t out_ext[40], const uint8_t enc[32]);
void orc_encode(uint8_t enc[32], const int32_t ext[40]);
void orc_add(int32_t out[40], const int32_t a[40], const int32_t b[40], int sub);
int orc_point_checks(const uint8_t enc[32]);
int orc_point_checks_ext(const int32_t ext[40]);
void orc_schnorr_sign(uint8_t sig[64], const uint8_t x[32], const uint8_t k[32], const uint8_t* msg, size_t n);
int orc_verify(int flavor, const uint8_t pub[32], const uint8_t* msg, size_t n, const uint8_t* sig, size_t sig_len);
}

static const char* g_msg = "";
static int refuse(const char* what) { g_msg = what; return KYB_E_BAD_ARG; }

extern "C" {
int kyb_init(int) { orc_init(); return KYB_OK; }
void kyb_shutdown(void) {}
const char* kyb_last_error(void) { return g_msg; }
int kyb_mul_base_batch(const uint8_t* s, size_t n, uint8_t* enc, int32_t* ext) {
  for (size_t i = 0; i < n; ++i) orc_mul_base(enc ? enc + 32 * i : nullptr, ext ? ext + 40 * i : nullptr, s + 32 * i);
  return KYB_OK;
}
int kyb_mul_batch(const uint8_t* s, const uint8_t* pe, const int32_t* px, size_t n, uint8_t* enc, int32_t* ext, uint8_t* ok) {
  for (size_t i = 0; i < n; ++i) {
    int32_t p[40];
    if (px) memcpy(p, px + 40 * i, 160);
    else { const int good = orc_decode(p, pe + 32 * i); if (ok) ok[i] = (uint8_t)good; if (!good) continue; }
    orc_mul(enc ? enc + 32 * i : nullptr, ext ? ext + 40 * i : nullptr, s + 32 * i, p);
  }
  return KYB_OK;
}
int kyb_add_batch(const int32_t* a, const int32_t* b, size_t n, int32_t* out, int sub) {
  for (size_t i = 0; i < n; ++i) orc_add(out + 40 * i, a + 40 * i, b + 40 * i, sub);
  return KYB_OK;
}
int kyb_encode_batch(const int32_t* ext, size_t n, uint8_t* enc) {
  for (size_t i = 0; i < n; ++i) orc_encode(enc + 32 * i, ext + 40 * i);
  return KYB_OK;
}
int kyb_decode_batch(const uint8_t* enc, size_t n, int32_t* ext, uint8_t* ok) {
  for (size_t i = 0; i < n; ++i) ok[i] = (uint8_t)orc_decode(ext + 40 * i, enc + 32 * i);
  return KYB_OK;
}
// Point::eq of the reference: both encodings, two inversions (point.rs:227-241)
int kyb_equal_batch(const int32_t* a, const int32_t* b, size_t n, uint8_t* eq) {
  for (size_t i = 0; i < n; ++i) { uint8_t ea[32], eb[32]; orc_encode(ea, a + 40 * i); orc_encode(eb, b + 40 * i); eq[i] = memcmp(ea, eb, 32) == 0; }
  return KYB_OK;
}
int kyb_point_checks_batch(const uint8_t* enc, const int32_t* ext, size_t n, uint8_t* flags) {
  for (size_t i = 0; i < n; ++i) flags[i] = (uint8_t)(enc ? orc_point_checks(enc + 32 * i) : orc_point_checks_ext(ext + 40 * i));
  return KYB_OK;
}
// schnorr::sign (schnorr_sig.rs:25-47) with the nonce as an input: two fixed-base multiplications, two encodings, SHA-512, one scalar multiply-add
int kyb_schnorr_sign_batch(const uint8_t* x, const uint8_t* k, const uint8_t* msgs, const uint32_t* off, size_t n, uint8_t* sig) {
  for (size_t i = 0; i < n; ++i) orc_schnorr_sign(sig + 64 * i, x + 32 * i, k + 32 * i, msgs + off[i], off[i + 1] - off[i]);
  return KYB_OK;
}
int kyb_verify_batch(const uint8_t* pubs, const uint8_t* msgs, const uint32_t* off, const uint8_t* sigs, size_t n, int flavor, uint8_t* status) {
  for (size_t i = 0; i < n; ++i) status[i] = (uint8_t)orc_verify(flavor, pubs + 32 * i, msgs + off[i], off[i + 1] - off[i], sigs + 64 * i, 64);
  return KYB_OK;
}
int kyb_verify_points_batch(const int32_t* pubs_ext, const uint8_t* msgs, const uint32_t* off, const uint8_t* sigs, size_t n, int flavor, uint8_t* status) {
  for (size_t i = 0; i < n; ++i) { uint8_t pub[32]; orc_encode(pub, pubs_ext + 40 * i); status[i] = (uint8_t)orc_verify(flavor, pub, msgs + off[i], off[i + 1] - off[i], sigs + 64 * i, 64); }
  return KYB_OK;
}
#ifndef KYB_CPU_PORT_DEFER
// the CPU port runs eagerly: nothing is recorded
int kyb_defer_input(const int32_t*, uint64_t*) { return refuse("cpu port: no deferred mode"); }
int kyb_defer_input_enc(const int32_t*, const uint8_t*, uint64_t*) { return refuse("cpu port: no deferred mode"); }
int kyb_defer_null(uint64_t*) { return refuse("cpu port: no deferred mode"); }
int kyb_defer_base(uint64_t*) { return refuse("cpu port: no deferred mode"); }
int kyb_defer_mul_base(const uint8_t*, uint64_t*) { return refuse("cpu port: no deferred mode"); }
int kyb_defer_mul(const uint8_t*, uint64_t, uint64_t*) { return refuse("cpu port: no deferred mode"); }
int kyb_defer_add(uint64_t, uint64_t, int, uint64_t*) { return refuse("cpu port: no deferred mode"); }
int kyb_defer_neg(uint64_t, uint64_t*) { return refuse("cpu port: no deferred mode"); }
int kyb_defer_get(uint64_t, int32_t*, uint8_t*) { return refuse("cpu port: no deferred mode"); }
int kyb_defer_equal(uint64_t, uint64_t, uint8_t*) { return refuse("cpu port: no deferred mode"); }
int kyb_defer_flush(void) { return KYB_OK; }
uint64_t kyb_defer_mark(void) { return 0; }
int kyb_defer_floor(uint64_t) { return KYB_OK; }
int kyb_defer_stats(uint64_t* out, int cap) { for (int i = 0; i < cap; ++i) out[i] = 0; return KYB_OK; }
}
#else
// -DKYB_CPU_PORT_DEFER: the PRODUCT's deferred-point evaluator (kyber-rs_amd/csrc/defer.inc: arena, window, kept values, chain recognition) compiled
// for the CPU on top of the entry points above — so that the C++ mirror's DEFAULT mode (deferred) and the long-running programs run in the CPU suite:
// host logic of the product under test, the curve arithmetic behind it answered by the oracle.  The two batch calls only the evaluator makes:
void orc_pubpoly_eval(uint8_t out_enc[32], const int32_t* commits_ext, size_t t, uint32_t index);
void orc_null(int32_t out[40]);
int kyb_pubpoly_eval_multi_batch(const int32_t* commits, size_t t, size_t m, const uint32_t* idx, size_t k, uint8_t* enc, int32_t* ext) {
  for (size_t gi = 0; gi < m; ++gi)
    for (size_t j = 0; j < k; ++j) {
      uint8_t e[32];
      orc_pubpoly_eval(e, commits + 40 * t * gi, t, idx[gi * k + j]);
      if (enc) memcpy(enc + 32 * (gi * k + j), e, 32);
      if (ext && !orc_decode(ext + 40 * (gi * k + j), e)) return refuse("oracle produced an encoding that does not decode");
    }
  return KYB_OK;
}
int kyb_sum_batch(const int32_t* pts, size_t m, size_t t, uint8_t* enc, int32_t* ext) {
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
#include <algorithm>
#include <atomic>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>
struct DeferArena;
void defer_release(DeferArena* a);
namespace {
struct Ctx {      // the little of the engine's context that defer.inc refers to
  bool ready = true;
  int device = 0;
  std::mutex launch_mu;
  std::atomic<DeferArena*> defer{nullptr};
  std::atomic<int> opt_defer_fuse{1};
  std::atomic<int> opt_defer_max_nodes{1 << 18};
  std::atomic<int> opt_defer_keep_mib{256};
};
Ctx g_ctx;
Ctx* cur() { return &g_ctx; }
thread_local std::string g_fail_text;
int fail(int code, const char* msg) { g_fail_text = msg; g_msg = g_fail_text.c_str(); return code; }
}  // namespace
void defer_projective(bool) {}      // (the engine hands projective limbs to a flush that serves a comparison; the oracle's limbs are what they are)
#define ENTER() Ctx* ctx_ = cur(); Ctx& g = *ctx_; (void)g
#define ENTER_HOST() ENTER()
#include "../../kyber-rs_amd/csrc/defer.inc"
extern "C" {
int kyb_set_option(const char* key, int value) {
  if (!strcmp(key, "defer.max_nodes")) { g_ctx.opt_defer_max_nodes = value; return KYB_OK; }
  if (!strcmp(key, "defer.keep_mib")) { g_ctx.opt_defer_keep_mib = value; return KYB_OK; }
  if (!strcmp(key, "defer.fuse")) { g_ctx.opt_defer_fuse = value; return KYB_OK; }
  return refuse("cpu port: unknown option");
}
}
#endif
