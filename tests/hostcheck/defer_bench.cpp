// Host-side cost of the deferred-point arena (kyber-rs_amd/csrc/defer.inc) with the engine's batch entry points replaced by stubs that only
// fill their outputs: what recording and analysing a graph costs on the calling thread, per node, without any GPU or oracle time in it.
// Shapes: dist_key_share (dkg.rs:905-953: n dealers' polynomials of t commitments folded one Point::add at a time, then t marshals) and the
// verifiers' PubPoly::eval chains (poly.rs:457-469).  Test infrastructure (tests/test_defer_host.py prints the numbers); nothing ships from here.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "../../include/kyber_ed25519.h"

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
Ctx g_ctx;
Ctx* cur() { return &g_ctx; }
thread_local std::string g_err;
int fail(int code, const char* msg) { g_err = msg; return code; }
long g_calls = 0;
void fill(int32_t* ext, size_t n) { if (ext) for (size_t i = 0; i < 40 * n; ++i) ext[i] = (int32_t)(i * 2654435761u + (uint32_t)g_calls); }
}  // namespace
#define ENTER() Ctx* ctx_ = cur(); Ctx& g = *ctx_
#define ENTER_HOST() ENTER()

extern "C" {
int kyb_mul_base_batch(const uint8_t*, size_t n, uint8_t* enc, int32_t* ext) { ++g_calls; fill(ext, n); if (enc) memset(enc, 1, 32 * n); return KYB_OK; }
int kyb_mul_batch(const uint8_t*, const uint8_t*, const int32_t*, size_t n, uint8_t* enc, int32_t* ext, uint8_t*) { ++g_calls; fill(ext, n); if (enc) memset(enc, 2, 32 * n); return KYB_OK; }
int kyb_add_batch(const int32_t*, const int32_t*, size_t n, int32_t* out, int) { ++g_calls; fill(out, n); return KYB_OK; }
int kyb_encode_batch(const int32_t*, size_t n, uint8_t* enc) { ++g_calls; memset(enc, 3, 32 * n); return KYB_OK; }
int kyb_equal_batch(const int32_t*, const int32_t*, size_t n, uint8_t* eq) { ++g_calls; memset(eq, 1, n); return KYB_OK; }
int kyb_pubpoly_eval_multi_batch(const int32_t*, size_t, size_t m, const uint32_t*, size_t k, uint8_t* enc, int32_t* ext) { ++g_calls; fill(ext, m * k); if (enc) memset(enc, 4, 32 * m * k); return KYB_OK; }
int kyb_sum_batch(const int32_t*, size_t m, size_t, uint8_t* enc, int32_t* ext) { ++g_calls; fill(ext, m); if (enc) memset(enc, 5, 32 * m); return KYB_OK; }
}

#include "../../kyber-rs_amd/csrc/defer.inc"

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const size_t n = argc > 1 ? (size_t)atol(argv[1]) : 64, t = argc > 2 ? (size_t)atol(argv[2]) : 43;
  const int reps = 20;
  std::vector<int32_t> pts(40 * n * t);
  for (size_t i = 0; i < pts.size(); ++i) pts[i] = (int32_t)(i * 2246822519u);
  double rec_best = 1e30, flush_best = 1e30, h_rec_best = 1e30, h_flush_best = 1e30;
  for (int r = 0; r < reps; ++r) {
    for (size_t i = 0; i < pts.size(); i += 40) pts[i] ^= (int32_t)(r + 1);       // fresh limbs every repetition: no leaf is a repeat
    const uint64_t mark = kyb_defer_mark();
    // dist_key_share: pubb = poly_0; for d in 1..n: pubb[j] = pubb[j] + poly_d[j]
    double a = now_us();
    std::vector<uint64_t> acc(t);
    for (size_t j = 0; j < t; ++j) kyb_defer_input(&pts[40 * j], &acc[j]);
    for (size_t d = 1; d < n; ++d)
      for (size_t j = 0; j < t; ++j) { uint64_t h, s; kyb_defer_input(&pts[40 * (d * t + j)], &h); kyb_defer_add(acc[j], h, 0, &s); acc[j] = s; }
    double b = now_us();
    uint8_t enc[32];
    for (size_t j = 0; j < t; ++j) if (kyb_defer_get(acc[j], nullptr, enc) != KYB_OK) { std::printf("FAILED: %s\n", g_err.c_str()); return 1; }
    double c = now_us();
    rec_best = std::min(rec_best, b - a); flush_best = std::min(flush_best, c - b);
    // n verifiers' PubPoly::eval over the first polynomial
    std::vector<uint64_t> hc(t);
    for (size_t j = 0; j < t; ++j) kyb_defer_input(&pts[40 * j], &hc[j]);
    a = now_us();
    std::vector<uint64_t> sinks(n);
    for (size_t i = 0; i < n; ++i) {
      uint8_t x[32] = {0}; const uint32_t xi = (uint32_t)i + 1; memcpy(x, &xi, 4);
      uint64_t v; kyb_defer_null(&v);
      for (size_t j = t; j-- > 0;) { uint64_t m, s; kyb_defer_mul(x, v, &m); kyb_defer_add(m, hc[j], 0, &s); v = s; }
      sinks[i] = v;
    }
    b = now_us();
    for (size_t i = 0; i < n; ++i) if (kyb_defer_get(sinks[i], nullptr, enc) != KYB_OK) { std::printf("FAILED: %s\n", g_err.c_str()); return 1; }
    c = now_us();
    h_rec_best = std::min(h_rec_best, b - a); h_flush_best = std::min(h_flush_best, c - b);
    kyb_defer_floor(mark);
  }
  const double adds = (double)((n - 1) * t), hn = (double)(2 * n * t);
  std::printf("{\"n\": %zu, \"t\": %zu, \"fold\": {\"nodes\": %.0f, \"record_us\": %.1f, \"flush_us\": %.1f, \"ns_per_add_recorded\": %.0f, \"ns_per_add_total\": %.0f}, "
              "\"horner\": {\"nodes\": %.0f, \"record_us\": %.1f, \"flush_us\": %.1f, \"ns_per_node_total\": %.0f}, \"engine_calls\": %ld}\n",
              n, t, adds + (double)(n * t), rec_best, flush_best, 1e3 * rec_best / adds, 1e3 * (rec_best + flush_best) / adds,
              hn, h_rec_best, h_flush_best, 1e3 * (h_rec_best + h_flush_best) / hn, g_calls);
  return 0;
}
