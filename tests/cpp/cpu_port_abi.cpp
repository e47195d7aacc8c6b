// The C ABI's element-level entry points answered by the CPU oracle (oracle/ed25519_oracle.c), one item after the other on the calling thread:
// what tests/cpp/test_vss_round.cpp and test_dkg_finish.cpp link against INSTEAD of libkyber_ed25519_hip.so to time the very same call-by-call
// sequences on one host core — the "cpu_port_ms" column beside the engine's eager and deferred columns (round-4 review: the CPU figure for a
// dealer round was a sum of per-operation estimates, never run).  The C port follows the reference's algorithm and limb schedule (radix 2^25.5,
// signed radix-16 windows), so its timings stand for the reference's CPU path as far as anything in this image can.
// TEST INFRASTRUCTURE: lives under tests/, is never linked into the product, and refuses the deferred entry points (the CPU runs eagerly).
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
// the CPU port runs eagerly: nothing is recorded
int kyb_defer_input(const int32_t*, uint64_t*) { return refuse("cpu port: no deferred mode"); }
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
