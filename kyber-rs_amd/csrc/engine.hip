// MI355X (gfx950) batched Ed25519 engine: the host side behind libkyber_ed25519_hip.so.
//
// One scalar(-point pair) per lane, 64 lanes per wavefront, field elements as ten 32-bit VGPRs, products on
// v_mad_u64_u32 (fe25519.h).  The path is integer-VALU bound: algorithmic HBM traffic is 64..224 B per operation
// against ~2*10^5 multiply-adds, so there is no MFMA and no LDS tiling of operands; LDS holds only the shared
// base-point table of the fixed-base kernels.  The kernels live in the kernels_*.hip units (map: launch.h); this
// unit holds
//   * contexts: one per (process, device) by default (kyb_init), any number through kyb_ctx_create — each with its own
//     streams, per-stream scratch, staging buffers, table image, options and profiling state;
//   * the launch sequences (which kernels a batch call runs, in which scratch);
//   * the host-pointer pipeline (large batches: one copy-in lane, two compute lanes, one copy-out lane over chunks that ramp up and down;
//     pageable caller memory through a page-locked bounce ring filled by copy threads);
//   * multi-device groups: one context and one host thread per GPU, shards [rN/G, (r+1)N/G), the table image moved
//     with ncclBroadcast (librccl, loaded on demand) — engine_group.inc;
//   * the extern "C" entry points of include/kyber_ed25519.h — c_abi.inc.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <unordered_map>
#include <unordered_set>
#include <algorithm>
#include <vector>

#include "../../include/kyber_ed25519.h"
#include "launch.h"
#include "scalar_scan.h"
#include "ge_scalarmult.h"      // table image geometry (KYB_BASE*_TABLE_WORDS, KYB_BT_IDX); plain C++ on the host
#include "host_copy_pool.h"

using namespace kyb;

struct DeferArena;                          // deferred points of a context (defer.inc)
void defer_release(DeferArena* a);

namespace {

thread_local std::string g_err;

// optional per-launch timing (bench.py): HIP events recorded on the launch stream around each kernel
enum KernelId { KID_MUL, KID_MUL_BASE, KID_FINISH, KID_SIGN, KID_SIGN_HASH, KID_VERIFY_PREP, KID_VERIFY_FINAL, KID_POLY_EVAL, KID_MONT_PREP,
                KID_MUL_LADDER, KID_DECODE, KID_EDDSA_PREP, KID_PAIR_SUM, KID_VERIFY_PREP_R, KID_ENCODE, KID_MUL_COOP, KID_MUL_BASE_COOP, KID_DECODE_COOP, KID_POLY_EVAL_COOP, KID_VERIFY_COOP, KID_FINISH_COOP, KID_SIGN_COOP, KID_MSM_TABLES, KID_MSM_ACCUMULATE, KID_MUL_LADDER_PAIR, KID_PRIPOLY_EVAL, KID_LADDER_RECOVER, KID_COUNT };
const char* const KERNEL_NAMES[KID_COUNT] = {"k_mul", "k_mul_base", "k_finish", "k_sign", "k_sign_hash", "k_verify_prep", "k_verify_final", "k_poly_eval",
                                             "k_mont_prep", "k_mul_ladder", "k_decode", "k_eddsa_prep", "k_pair_sum", "k_verify_prep_r", "k_encode_batched", "k_mul_coop", "k_mul_base_coop", "k_decode_coop", "k_poly_eval_coop", "k_verify_coop", "k_finish_coop", "k_sign_coop", "k_msm_tables", "k_msm_accumulate", "k_mul_ladder_pair", "k_pripoly_eval", "k_ladder_recover"};
struct ProfRec { int id; hipEvent_t a, b; };
struct Prof {
  std::mutex mu;                  // begin / read / every ProfScope: callable from any thread
  bool on = false;
  int cap = 0, used = 0;
  ProfRec* recs = nullptr;
};

// per-stream device scratch (two launches that overlap on different streams must not share it):
//   ws       variable-base table workspace of the windowed kernel (fixed size, allocated on first use)
//   proj     projective staging of the split finish (grows with the largest batch seen)
//   enc      encodings of R and A between the stages of the split signing path, verification scratch
//   aux      internal side stream (+ fork/join events) on which a small verification batch runs s*B next to the ladder
//   ev_last  recorded behind the last launch that used the slot: a slot is only recycled / released / re-bound to a
//            recycled stream handle after it (so a destroyed caller stream is never touched again)
struct StreamRes {
  hipStream_t stream = nullptr;
  uint4* ws = nullptr; uint4* proj = nullptr; size_t proj_items = 0; uint8_t* enc = nullptr; size_t enc_bytes = 0;
  uint32_t* part = nullptr; size_t part_items = 0;      // extended quads of a small linear combination's products
  uint32_t* pieces = nullptr;                           // k_mul_coop in four workgroups per item: their records and arrival counters (zero between launches)
  uint8_t* pub_enc = nullptr; size_t pub_enc_items = 0;    // kyb_verify_points_batch: marshal_binary of the callers' public-key points
  uint32_t* msm = nullptr; size_t msm_points = 0;        // kyb_lincomb_public_batch over shared points: window bases + tables of the points (227,040 B per point)
  uint32_t* top_or = nullptr; unsigned top_seq = 0;     // two alternating words behind the projective staging records (k_mont_prep / k_mul_ladder)
  hipStream_t aux = nullptr; hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_mid = nullptr, ev_last = nullptr;
  bool used = false, own = false;
  uint64_t last_use = 0;
};
constexpr size_t MAX_STREAM_SLOTS = 32;

constexpr int PIPE_CHUNKS_DEFAULT = 16;     // the first chunk of a pipelined host-pointer batch is 1/16 of it
struct Ctx {
  bool ready = false;
  int device = -1;
  int cus = 0;                    // compute units the launches are sized for: hw_cus, or option device.cus
  int hw_cus = 0;                 // what the device reports
  char name[128] = {0};
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;   // second compute lane of the pipelined host-pointer path (consecutive chunks alternate: the small kernels of one overlap the ladder of the other)
  hipStream_t stream_in = nullptr, stream_out = nullptr;      // ... its copy lanes: host -> device, device -> host (two DMA engines, full duplex), made on first use
  std::vector<hipEvent_t> pipe_ev;                            // ... and its events (input arrived / kernels done / output left, per chunk)
  hipEvent_t ev_ring[4] = {nullptr, nullptr, nullptr, nullptr};   // bounce ring of the pageable input path: a slot's last copy to the device
  uint8_t* pin_out = nullptr; size_t pin_out_bytes = 0;       // page-locked landing area for the outputs of a pageable caller
  uint32_t* table = nullptr;       // KYB_BASE_TABLE_BYTES: radix-16 image (65,536 B), radix-32 image (106,496 B), radix-64 image (163,200 B)
  uint32_t* table_coop = nullptr;  // KYB_COOP_TABLE_WORDS: entry-major copy of the radix-64 table for the one-item-per-wavefront kernels, derived from `table` on this GPU
  uint64_t* ck_dev = nullptr;      // 8 bytes: checksum of an imported table image
  std::atomic<bool> table_ready{false};
  std::vector<StreamRes*> res;
  uint64_t use_clock = 0;
  size_t ws_bytes = 0;
  std::atomic<int> grid_mul{0};
  // fault injection and accounting for tests (kyb_set_option diag.fail_alloc_after / diag.fail_launch_after, kyb_get_option diag.dev_kib / diag.host_kib):
  // counters > 0 count DOWN on every allocation attempt / kernel launch of this context; the attempt that brings one to 0 fails without being made
  std::atomic<int> diag_fail_alloc{0}, diag_fail_launch{0};
  std::atomic<long long> diag_dev_bytes{0}, diag_host_bytes{0};      // live bytes of the context's lazily grown buffers (device / page-locked)
  uint8_t* stage = nullptr;       // device staging for the host-pointer API
  size_t stage_bytes = 0;
  uint32_t* done_flag = nullptr;          // coherent page-locked word the last kernel of a small host-pointer call writes (launch.h, DoneFlag)
  uint32_t* done_counter = nullptr;       // device word: finished items of that kernel
  uint32_t done_seq = 0;                  // under mu
  uint8_t* pin[2] = {nullptr, nullptr};   // page-locked bounce buffers of the two lanes (pageable caller memory)
  hipEvent_t ev_pin[2] = {nullptr, nullptr};   // h2d(): a bounce buffer's last copy to the device
  size_t pin_bytes[2] = {0, 0};
  kyb::CopyPool copy;
  // kernel variant selection (kyb_set_option): atomics, so a set_option from one thread and launches from others do not race
  std::atomic<int> opt_copy_threads{0};       // host threads that move pageable batches through the bounce buffers (0 = auto)
  std::atomic<int> opt_mul_select{1};         // 0 cndmask, 1 and/or mask
  std::atomic<int> opt_base_select{1};        // 0 LDS broadcast scan, 1 bpermute
  std::atomic<int> opt_base_block{256};       // 256 (2 waves/SIMD) or 512 (4 waves/SIMD, 128 VGPRs)   [radix-16 kernel]
  std::atomic<int> opt_base_radix{64};        // 64 / 32: 43- / 52-window kernel for batches >= finish.min_items; 16: always the 64-window kernel
  std::atomic<int> opt_base_block64{1024};    // radix-64 kernel, full batches: 1024 (4 waves/SIMD, <= 128 VGPRs) or 512 (2 waves/SIMD)
  std::atomic<int> opt_base_small_chunks{2};  // radix-64 kernel: 256-thread workgroups up to this many chunks per CU, 1024-thread beyond
  std::atomic<int> opt_verify_overlap{1};     // small verification batches: s*B on a side stream next to the ladder
  std::atomic<int> opt_mul_algo{1};           // 0 windowed table (ge.rs structure), 1 Montgomery ladder (table-free, 1.33x faster: profiles/r01/sweep_mul_algo.log)
  std::atomic<int> opt_ladder_waves{3};       // launch bound of k_mul_ladder: waves per SIMD the register allocator must allow
  std::atomic<int> opt_finish{1};             // 0 fused inversion per item, 1 split + batched inversion (n >= finish_min)
  std::atomic<int> opt_finish_min{1};         // batches below it: fused per-item inversion in the radix-16 kernels (slower at every size; cross-check)
  std::atomic<int> opt_coop_max{0};           // variable base: batches of at most this many items take the one-item-per-wavefront kernel (0 = never; crossover measured between 6144 and 8192, profiles/r02/coop_crossover.log)
  std::atomic<int> opt_pipe_chunks{PIPE_CHUNKS_DEFAULT};      // host-pointer batches of 2^16 items or more: the chunk unit is 1/this of the batch (plan_chunks)
  std::atomic<int> opt_ext_projective{0};        // 1: small-batch multiplications asked for extended limbs ONLY return them projective (Z != 1, no inversion)
  std::atomic<int> opt_base_quarters{1};         // mid-size fixed-base launches give an item four wavefronts, a quarter of the 43 windows each (k_mul_base64_quarters); 0 = never
  std::atomic<int> opt_finish_four{2};           // k_finish with 4 instead of 8 items per shared inversion for launches of at most one wavefront per SIMD
  std::atomic<int> opt_ladder_y_only{2};         // two-lane ladder from wire encodings: ladder on y while the decode looks for x — 2: as workgroups of the same launch, 1: on a side stream (0: decode first)
  std::atomic<DeferArena*> defer{nullptr};       // recorded, not yet evaluated point operations of this context's callers (kyb_defer_*; made on first use)
  std::atomic<int> opt_defer_fuse{1};            // flushes recognise Horner chains and chains of additions (defer.inc)
  std::atomic<int> opt_defer_max_nodes{1 << 18}; // the arena's WINDOW: the youngest nodes, with their graph (40 B each + 224 B where a value is held: at most 69 MB); older nodes leave their values in ...
  std::atomic<int> opt_defer_keep_mib{256};      // ... the table of kept values (208 B per evaluated node, bounded by this many MiB; least recently touched go first; 0: values leave with their nodes)
  bool stamps_on = false;                        // this context set the device's wave-stamp slots (kyb_diag_wave_stamps): cleared again when it is released
  std::atomic<int> opt_coop_share{1};             // divide the small-batch thresholds by the number of host-pointer calls in flight in this process (coop_lim)
  std::atomic<int> opt_host_inplace{1};           // zero-copy host-pointer calls use page-locked CALLER arrays (kyb_host_alloc) where they lie instead of copying them into the context's buffer
  std::atomic<int> opt_zero_copy_kib{4096};         // host-pointer calls whose arrays fit this many KiB run their kernels on the context's page-locked buffer (no hipMemcpy); 512 in round 2, 4 MiB wins up to 16,384 items (profiles/r03/mid_size_host_calls.log)
  std::atomic<int> opt_ladder_quad_max{0};    // ... and of at most this many items FOUR lanes (k_mul_ladder_quad: one wavefront per SIMD up to here); 0 = never
  std::atomic<int> opt_ladder_pair_max{0};    // ladder launches of at most this many items give each item two lanes (k_mul_ladder_pair: one wavefront per SIMD up to here); 0 = never
  std::atomic<int> opt_coop_ladder_max{0};     // variable base from points, linear combinations: above this the two-lane batch ladder is faster than one item per wavefront (3072 until the batch path lost its in-kernel decode wait: profiles/r04/coop_vs_fused.log)
  std::atomic<int> opt_coop_ladder_enc_max{0};  // the same for calls from BYTES (multiplication from encodings, verification from key bytes): the role-split launches of ladder.y_only = 2 take over at two wavefronts per SIMD
  std::atomic<int> opt_ladder_skip_canonical{1};  // the batch ladder skips the four leading bits when no scalar of the launch has one set (canonical scalars)
  std::atomic<int> opt_mul_short_scalars{0};     // 1: EVERY host-pointer kyb_mul_batch of <= 64 items is treated like kyb_mul_public_batch (multipliers declared public:
                                                 // when all are below 2^64 the ladder skips the leading zeros).  Off by default: the ABI cannot know that a multiplier is public.
  std::atomic<int> opt_poly_batch_segments{0};   // PubPoly::eval, long polynomials at 10^3..10^4 evaluations: lanes per evaluation of the batch kernels (0 = cost model, 1 = never, 2..256)
  std::atomic<int> opt_poly_segments{0};         // PubPoly::eval, small batches: wavefronts per evaluation (0 = chosen from t and the batch size, 1 = never split, 2..32)
  std::atomic<int> opt_verify_by_enc{1};         // large verification batches: compare the encoding of s*B - h*A with R's bytes, decode R only on a mismatch
  std::atomic<int> opt_coop_verify_max{0};     // verification: up to this many signatures take the single-launch kernel (three wavefronts each)
  std::atomic<int> opt_coop_decode_max{0};    // unmarshal_binary alone: the same (crossover between 1024 and 2048)
  std::atomic<int> opt_coop_base_max{0};      // fixed base: the same (crossover at 13 wavefronts per CU since the batch path's finish shares an inversion between four items: profiles/r04/coop_vs_fused.log; 4096 before; signing counts its two multiplications per item)
  std::atomic<int> opt_encode_batched{1};     // kyb_encode_batch: 1 shared inversion per 8 points (k_encode_batched), 0 one inversion per point (k_encode)
  std::mutex mu;          // host-pointer API: staging buffers + engine streams of this context
  std::mutex launch_mu;   // every launch_* entry: per-stream scratch bookkeeping (calls from any thread, any stream)
  Prof prof;
};

// ---- context registry --------------------------------------------------------------------------------
std::mutex g_reg_mu;
std::vector<Ctx*> g_all;               // every live context (default one included)
Ctx* g_default = nullptr;              // created by kyb_init / kyb_init_no_table, destroyed by kyb_shutdown
thread_local Ctx* tl_cur = nullptr;    // kyb_ctx_set_current; nullptr = the default context
// the deferred evaluator (defer.inc) asks for projective limbs for the calls of a flush whose results only feed other operations and a projective
// comparison — the option ext.projective for the calling thread's calls only, whatever the context's option says
thread_local bool tl_defer_projective = false;
inline bool ext_projective(const Ctx& g) { return g.opt_ext_projective != 0 || tl_defer_projective; }
void defer_projective(bool on) { tl_defer_projective = on; }

int fail(int code, const char* what, hipError_t e = hipSuccess) {
  char buf[320];
  if (e != hipSuccess) snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
  else snprintf(buf, sizeof(buf), "%s", what);
  g_err = buf;
  return code;
}
#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(KYB_E_HIP, #x, e_); } while (0)
// a kernel launch of an entry point (`g` = its context in scope): the injected failure is reported without the launch being made
#define LAUNCHCK(x) do { if (diag_trip(g.diag_fail_launch)) return fail(KYB_E_HIP, "injected launch failure (diag.fail_launch_after) at " #x); \
                         hipError_t e_ = (x); if (e_ != hipSuccess) return fail(KYB_E_HIP, #x, e_); } while (0)
// (fault injection exists in the CROSS-CHECK build only — the tests' library; in the product nothing can make a launch or an allocation fail on request)
#ifdef KYB_CROSSCHECK
inline bool diag_trip(std::atomic<int>& c) {
  if (c.load(std::memory_order_relaxed) <= 0) return false;
  return c.fetch_sub(1, std::memory_order_relaxed) == 1;
}
#else
inline constexpr bool diag_trip(std::atomic<int>&) { return false; }
#endif
// every buffer a context grows on demand comes from these four: one place for the injected failure and for the byte accounting the leak tests read
inline hipError_t ctx_malloc(Ctx& g, void** p, size_t bytes) {
  if (diag_trip(g.diag_fail_alloc)) { *p = nullptr; return hipErrorOutOfMemory; }
  hipError_t e = hipMalloc(p, bytes);
  if (e == hipSuccess) g.diag_dev_bytes.fetch_add((long long)bytes, std::memory_order_relaxed);
  return e;
}
inline hipError_t ctx_host_malloc(Ctx& g, void** p, size_t bytes, unsigned flags) {
  if (diag_trip(g.diag_fail_alloc)) { *p = nullptr; return hipErrorOutOfMemory; }
  hipError_t e = hipHostMalloc(p, bytes, flags);
  if (e == hipSuccess) g.diag_host_bytes.fetch_add((long long)bytes, std::memory_order_relaxed);
  return e;
}
inline void ctx_free(Ctx& g, void* p, size_t bytes) { if (p) { (void)hipFree(p); g.diag_dev_bytes.fetch_sub((long long)bytes, std::memory_order_relaxed); } }
inline hipError_t ctx_host_free(Ctx& g, void* p, size_t bytes) { if (!p) return hipSuccess; g.diag_host_bytes.fetch_sub((long long)bytes, std::memory_order_relaxed); return hipHostFree(p); }

Ctx* cur() {
  Ctx* c = tl_cur;
  if (c == nullptr) return g_default;
  return c;
}
// Every entry point: the calling thread's context, made current on its device (HIP's current device is per thread and
// starts at 0, so a call from a fresh thread on an engine bound to another GPU would otherwise allocate and launch on GPU 0).
#define ENTER()                                                                                              \
  Ctx* ctx_ = cur();                                                                                         \
  if (ctx_ == nullptr || !ctx_->ready) return fail(KYB_E_NOT_INIT, "no initialised context: kyb_init / kyb_ctx_create has not succeeded"); \
  Ctx& g = *ctx_;                                                                                            \
  HIPCK(hipSetDevice(g.device))
// entry points that only touch host memory (recording a deferred operation, a cache hit): no device call on the way in — whatever they
// go on to run on the GPU passes through an ordinary entry point, which makes the device current
#define ENTER_HOST()                                                                                         \
  Ctx* ctx_ = cur();                                                                                         \
  if (ctx_ == nullptr || !ctx_->ready) return fail(KYB_E_NOT_INIT, "no initialised context: kyb_init / kyb_ctx_create has not succeeded"); \
  Ctx& g = *ctx_
#define REQUIRE_TABLE() do { if (!g.table_ready.load()) return fail(KYB_E_NOT_INIT, "base table not built or imported"); } while (0)

struct ProfScope {
  Ctx& g; hipStream_t st; int slot;
  ProfScope(Ctx& g_, hipStream_t s, int id) : g(g_), st(s), slot(-1) {
    std::lock_guard<std::mutex> lk(g.prof.mu);
    if (g.prof.on && g.prof.used < g.prof.cap) { slot = g.prof.used++; g.prof.recs[slot].id = id; (void)hipEventRecord(g.prof.recs[slot].a, st); }
  }
  ~ProfScope() {
    if (slot < 0) return;
    std::lock_guard<std::mutex> lk(g.prof.mu);
    if (slot < g.prof.cap) (void)hipEventRecord(g.prof.recs[slot].b, st);
  }
};

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
// NULL = the context's own (non-blocking) stream; KYB_STREAM_LEGACY = the device's null stream, which the runtime knows as the null handle
// (the name hipStreamLegacy is avoided: hipStreamWaitEvent on it crashed inside the ROCm 7.2 runtime, profiles/r06/README.md)
inline hipStream_t caller_stream(void* s) { return s == KYB_STREAM_LEGACY ? hipStream_t(nullptr) : reinterpret_cast<hipStream_t>(s); }
inline hipStream_t pick(Ctx& g, void* s) { return !s ? g.stream : caller_stream(s); }
inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }

// device buffers that held caller data (possibly private keys and nonces) are cleared before they go back to the runtime
// (callers have waited for the buffer's last user; the fill runs on the engine's own stream and is waited for there — a plain hipMemset
// would go through the device's null stream and stall every blocking stream of the process behind it)
void wipe_free_dev(Ctx& g, void* p, size_t bytes) {
  if (!p) return;
  if (g.stream && hipMemsetAsync(p, 0, bytes, g.stream) == hipSuccess) (void)hipStreamSynchronize(g.stream);
  else (void)hipMemset(p, 0, bytes);           // context being torn down before its stream existed
  ctx_free(g, p, bytes);
}

// The hand-over sizes between the kernel families are WAVEFRONTS PER COMPUTE UNIT — where a one-item-per-wavefront kernel has filled the SIMDs as
// often as it pays — not item counts: measured on 256 CUs (profiles/r02/coop_crossover.log, r04/coop_vs_fused.log, r04/coop_max_probe.log) and kept
// as per-CU figures, multiplied by the CUs the context works with: the device's, or what the host declares with option device.cus (a CU-masked
// stream, a partitioned device whose streams see fewer CUs than the property says; profiles/r05/coop_crossover_cu_mask.log).  An explicit
// kyb_set_option of one of the thresholds still sets an absolute item count.
// (round 6: the one-item-per-wavefront point operations move rows with permlane swaps instead of ds_bpermute and end in a cheaper inversion — they
//  stay ahead of the batch kernels for longer: fixed base 13 -> 18, variable base 11 -> 14 wavefronts per CU; profiles/r06/coop_crossover.log)
constexpr int COOP_MAX_PER_CU = 24, COOP_BASE_MAX_PER_CU = 18, COOP_LADDER_MAX_PER_CU = 14, COOP_LADDER_ENC_MAX_PER_CU = 6, COOP_DECODE_MAX_PER_CU = 4,
              COOP_VERIFY_MAX_PER_CU = 2, LADDER_PAIR_MAX_PER_CU = 128, LADDER_QUAD_MAX_PER_CU = 64;      // (128: two lanes per item up to one wavefront per SIMD = 4 SIMDs x 64 lanes / 2)
void apply_cu_count(Ctx& g, int cus) {
  g.cus = cus;
  g.opt_coop_max = COOP_MAX_PER_CU * cus;                       // 6,144 on 256 CUs
  g.opt_coop_base_max = COOP_BASE_MAX_PER_CU * cus;             // 4,608
  g.opt_coop_ladder_max = COOP_LADDER_MAX_PER_CU * cus;         // 3,584
  g.opt_coop_ladder_enc_max = COOP_LADDER_ENC_MAX_PER_CU * cus; // 1,536 (2,048 until the batch forms behind it got four lanes per item: 289 / 346 us at 1,536 items, 459 / 344 at 2,048)
  g.opt_coop_decode_max = COOP_DECODE_MAX_PER_CU * cus;         // 1,024
  g.opt_coop_verify_max = COOP_VERIFY_MAX_PER_CU * cus;         // 512
  g.opt_ladder_pair_max = LADDER_PAIR_MAX_PER_CU * cus;         // 32,768
  g.opt_ladder_quad_max = LADDER_QUAD_MAX_PER_CU * cus;         // 16,384
  g.grid_mul = cus * 2;
}

int ensure_stage(Ctx& g, size_t bytes) {
  if (bytes <= g.stage_bytes) return KYB_OK;
  if (g.stage) { wipe_free_dev(g, g.stage, g.stage_bytes); g.stage = nullptr; g.stage_bytes = 0; }
  size_t want = bytes + (bytes >> 2) + 4096;
  hipError_t e = ctx_malloc(g, reinterpret_cast<void**>(&g.stage), want);
  if (e != hipSuccess) return fail(KYB_E_NOMEM, "staging allocation", e);
  g.stage_bytes = want;
  return KYB_OK;
}
// The one-item-per-wavefront kernels (64 lanes per item) and the two-lane ladder buy latency with lanes: right while the rest of the chip
// would idle.  When several host threads have synchronous host-pointer calls in flight — each on its own context — they share the chip,
// and on ONE GPU a 4,096-item fixed-base call that fills it with 4,096 wavefronts makes the others wait (16 threads: 1.8e7 items/s with fixed
// thresholds, 1.5e8 with thresholds divided by the number of calls in flight; tools/concurrent_mid_calls.py).  coop.share_by_load = 0
// keeps the thresholds as set.  Results do not depend on the routing.
constexpr int MAX_COUNTED_DEVICES = 64;
std::atomic<int> g_host_calls_in_flight[MAX_COUNTED_DEVICES];      // per device: the shards of a kyb_group call run on different GPUs and do not share one
struct InflightScope {
  std::atomic<int>* slot;
  explicit InflightScope(const Ctx& g) : slot(g.device >= 0 && g.device < MAX_COUNTED_DEVICES ? &g_host_calls_in_flight[g.device] : nullptr) {
    if (slot) slot->fetch_add(1, std::memory_order_relaxed);
  }
  ~InflightScope() { if (slot) slot->fetch_sub(1, std::memory_order_relaxed); }
};
inline int host_load(const Ctx& g) {
  if (g.opt_coop_share == 0 || g.device < 0 || g.device >= MAX_COUNTED_DEVICES) return 1;
  const int load = g_host_calls_in_flight[g.device].load(std::memory_order_relaxed);
  return load < 1 ? 1 : (load > 64 ? 64 : load);
}
inline size_t coop_lim(const Ctx& g, int opt) { return opt <= 0 ? 0 : (size_t)opt / (size_t)host_load(g); }
// the closing inversion one point per wavefront (k_finish_coop) up to coop.decode_max_items; above, one point per lane and one inversion per wavefront
// (k_finish_wave) — tools/finish_crossover.py, profiles/r06/finish_crossover.log: 35.4 us flat against 33 / 35 / 42 us at 256 / 1,024 / 2,048 points
inline size_t finish_coop_lim(const Ctx& g) { return coop_lim(g, g.opt_coop_decode_max); }
// the two-lane ladder spends 2 lanes on an item, not 64: it stays worth its 12 % of extra work until the calls in flight fill the chip
// several times over (16 threads x 4,096 items: 5.7e7 items/s with it, 4.2e7 without)
// Fixed base above the one-item-per-wavefront sizes and up to 128 items per CU (two rounds of 64-item workgroups): k_mul_base64 would be ONE lane's chain of 43
// additions (105 us flat) on a mostly idle chip; four wavefronts per 64 items take a quarter of the windows each (tools/mid_size_kernels.py,
// profiles/r06/base_quarters.log)
// (a launch in this form occupies four times the compute units of the one-lane form — each workgroup owns a CU's LDS: with several host-pointer calls in flight the
//  size limit is divided by their number, like the other latency-oriented hand-overs)
inline bool base_quarters(const Ctx& g, size_t n) {
  return g.opt_base_quarters != 0 && g.opt_base_radix == 64 && n * (size_t)host_load(g) <= (size_t)128 * (size_t)g.cus;
}
// ... and with that form behind them the one-item-per-wavefront kernels hand a FIXED-BASE multiplication (and signing, two per signature) over at 5 wavefronts
// per CU instead of coop.base_max_items (tools/base_quarters_probe.py, profiles/r06/base_quarters.log: 70 against 99 us at 1,024 items, 98 / 86 at 1,536,
// 183 / 93 at 4,096); the other users of coop.base_max_items (short sums, small verifications) keep it
constexpr int COOP_BASE_TO_QUARTERS_PER_CU = 5;
inline size_t base_coop_lim(const Ctx& g) {
  const size_t lim = coop_lim(g, g.opt_coop_base_max), q = coop_lim(g, COOP_BASE_TO_QUARTERS_PER_CU * g.cus);
  return (g.opt_base_quarters != 0 && g.opt_base_radix == 64 && q < lim) ? q : lim;
}
inline bool finish_four(const Ctx& g, size_t n) { return g.opt_finish_four != 0 && n <= (size_t)64 * 4 * (size_t)g.cus; }      // launches of at most a wavefront per SIMD share an inversion between 4 items (finish.four)
// ... and with finish.four = 2 (default) launches of up to TWO wavefronts of finish lanes per SIMD close with ONE inversion per wavefront, spread over its lanes
// (k_finish_wave, kernels_coop.hip; profiles/r06/finish_crossover.log: 35 against 52 us up to 2^16 points, 48 / 64 at 2^17, 70 / 67 at 196,608)
inline bool finish_wave(const Ctx& g, size_t n) { return g.opt_finish_four == 2 && n <= (size_t)64 * 8 * (size_t)g.cus; }
inline size_t pair_lim(const Ctx& g, int opt) { const int l = host_load(g) / 4; return opt <= 0 ? 0 : (size_t)opt / (size_t)(l < 1 ? 1 : l); }
// A variable-base multiplication by FULL-SIZE scalars from points leaves the one-item-per-wavefront kernel at 9 wavefronts per CU when the four-lane ladder
// is there to take it (tools/ladder_quad_probe.py, profiles/r06/ladder_quad.log: 277 / 337 us at 2,048 items, 341 / 338 at 2,304, 450 / 334 at 3,584); the other
// users of coop.ladder_max_items (short public multipliers, linear combinations, verification: their batch forms are the two-lane ladder's) keep it
// lanes per item of a ladder launch below ladder.pair_max_items: four up to ladder.quad_max_items (a wavefront per SIMD), else two
inline int ladder_lanes(const Ctx& g, size_t n) { return (n <= pair_lim(g, g.opt_ladder_quad_max) && n <= pair_lim(g, g.opt_ladder_pair_max)) ? 4 : 2; }
constexpr int COOP_LADDER_TO_QUAD_PER_CU = 9;
inline size_t ladder_coop_lim(const Ctx& g) {
  const size_t lim = coop_lim(g, g.opt_coop_ladder_max), q = coop_lim(g, COOP_LADDER_TO_QUAD_PER_CU * g.cus);
  return (pair_lim(g, g.opt_ladder_quad_max) > q && pair_lim(g, g.opt_ladder_pair_max) > q && q < lim) ? q : lim;
}

int ensure_pin(Ctx& g, int lane, size_t bytes) {
  if (bytes <= g.pin_bytes[lane]) return KYB_OK;
  if (g.pin[lane]) { memset(g.pin[lane], 0, g.pin_bytes[lane]); HIPCK(ctx_host_free(g, g.pin[lane], g.pin_bytes[lane])); g.pin[lane] = nullptr; g.pin_bytes[lane] = 0; }
  size_t want = bytes + (bytes >> 2) + 4096;
  // coherent: the zero-copy path reads results out of it as soon as the last kernel signals, before the runtime has seen the kernel end
  hipError_t e = ctx_host_malloc(g, reinterpret_cast<void**>(&g.pin[lane]), want, hipHostMallocCoherent);
  if (e != hipSuccess) return fail(KYB_E_NOMEM, "pinned bounce buffer allocation", e);
  g.pin_bytes[lane] = want;
  return KYB_OK;
}

int copy_threads(Ctx& g) {
  if (g.opt_copy_threads > 0) return g.opt_copy_threads;
  const unsigned hw = std::thread::hardware_concurrency();
  const int t = (int)(hw / 2);
  return t < 1 ? 1 : (t > 8 ? 8 : t);
}
// page-locked (hipHostMalloc / hipHostRegister) memory is copied by the DMA engines directly
bool is_pinned(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeHost;
}

// Device-visible address of a caller array that lies in memory kyb_host_alloc handed out (page-locked, host-COHERENT, mapped for this device), or
// nullptr.  The kernels of a zero-copy call work on the context's page-locked buffer over PCIe anyway: an array that already is such memory is read and
// written where it lies, and the calling thread's memcpy into / out of the buffer (0.03 ms of a DKG-sized call) drops out.  Only kyb_host_alloc memory
// qualifies — other page-locked memory (hipHostRegister, a hipHostMalloc of the caller's) may be non-coherent, and a zero-copy call reads its results
// when the completion flag arrives, before the end of the kernel is observed; such arrays are copied like pageable ones.
constexpr size_t INPLACE_MIN_BYTES = (size_t)64 << 10;      // below this the copies cost less than looking
struct HostAllocs {
  struct Rec { size_t bytes; int device; ptrdiff_t dev_delta; };
  std::mutex mu;
  std::map<uintptr_t, Rec> recs;
};
HostAllocs& host_allocs() { static HostAllocs* h = new HostAllocs; return *h; }      // never destroyed: kyb_host_free may run during process teardown
uint8_t* pinned_dev_ptr(const Ctx& g, const void* p, size_t bytes) {
  const uintptr_t a = reinterpret_cast<uintptr_t>(p);
  if (p == nullptr || (a & 15u) != 0) return nullptr;        // the kernels load and store 16-byte words
  HostAllocs& h = host_allocs();
  std::lock_guard<std::mutex> lk(h.mu);
  auto it = h.recs.upper_bound(a);
  if (it == h.recs.begin()) return nullptr;
  --it;
  if (a - it->first > it->second.bytes || bytes > it->second.bytes - (a - it->first) || it->second.device != g.device) return nullptr;
  return reinterpret_cast<uint8_t*>(a + it->second.dev_delta);
}

// Two caller arrays of one call that are both used where they lie must not share bytes when one of them is written: staged arrays were
// copied apart (in-place updates like out_ext == pts_ext worked), the kernels' __restrict__ pointers and shared-inversion lanes do not
// allow it in place.  An overlapping INPUT is then staged as before the in-place path existed.
// a device-pointer entry point whose launch sequence failed part of the way: what it queued (on the caller's stream, on side streams of the
// slot) is waited for before the error is returned — the per-stream scratch is then free for the next call, and no kernel of the failed call
// writes into the caller's arrays after the caller has seen the error (their contents are undefined, include/kyber_ed25519.h)
inline int drained(int rc) {
  if (rc != KYB_OK) (void)hipDeviceSynchronize();
  return rc;
}
inline bool host_ranges_overlap(const void* a, size_t na, const void* b, size_t nb) {
  const uintptr_t x = reinterpret_cast<uintptr_t>(a), y = reinterpret_cast<uintptr_t>(b);
  return a != nullptr && b != nullptr && na != 0 && nb != 0 && x < y + nb && y < x + na;
}

// ---- completion flag of small host-pointer calls --------------------------------------------------------
// A zero-copy host-pointer call (run_host_batch / HostCall::run) posts a request in this thread-local; the launch sequence it runs
// on the engine stream hands the flag to its LAST kernel (take_done_flag, only where nothing is queued behind that kernel), and
// the call then spins on the flag word instead of synchronising the stream.  Sequences that do not take it are synchronised
// as before.  (Thread-local: a device-pointer call another thread makes on the same context must never pick the request up.)
struct DoneReq { bool armed = false; uint32_t seq = 0; };
thread_local DoneReq* tl_done = nullptr;
launch::DoneFlag take_done_flag(Ctx& g, hipStream_t st, size_t total) {
  if (tl_done == nullptr || tl_done->armed || st != g.stream || g.done_flag == nullptr || total == 0 || total > 0xffffffffu) return launch::DoneFlag{};
  tl_done->armed = true;
  tl_done->seq = ++g.done_seq;
  return launch::DoneFlag{g.done_counter, g.done_flag, tl_done->seq, (uint32_t)total};
}
int ensure_done_flag(Ctx& g) {
  if (g.done_flag != nullptr) return KYB_OK;
  hipError_t e = ctx_host_malloc(g, reinterpret_cast<void**>(&g.done_flag), 64, hipHostMallocCoherent);
  if (e != hipSuccess) { g.done_flag = nullptr; return fail(KYB_E_NOMEM, "completion flag allocation", e); }
  *g.done_flag = 0;
  e = ctx_malloc(g, reinterpret_cast<void**>(&g.done_counter), 64);
  if (e != hipSuccess) { (void)ctx_host_free(g, g.done_flag, 64); g.done_flag = nullptr; return fail(KYB_E_NOMEM, "completion counter allocation", e); }
  // on the stream the counting kernels run on, and finished before the first of them is queued: a plain hipMemset of device memory may
  // return before it has run and is not ordered with a non-blocking stream — a kernel that overtook it counted from whatever the recycled
  // allocation held, fired the flag early by that amount in every later call, and the caller read results the last workgroups had not
  // written yet (seen with six fresh contexts starting at once, tests/test_gpu_contexts.py)
  HIPCK(hipMemsetAsync(g.done_counter, 0, 64, g.stream));
  HIPCK(hipStreamSynchronize(g.stream));
  return KYB_OK;
}
// after the launch sequence of a zero-copy call: wait for its results
int wait_done(Ctx& g, const DoneReq& req) {
  if (req.armed) {
    volatile uint32_t* f = g.done_flag;
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned it = 1; *f != req.seq; ++it) {
#if defined(__x86_64__) || defined(__i386__)
      __builtin_ia32_pause();
#else
      std::this_thread::yield();
#endif
      if (it > 4096 && (it & 63) == 0) std::this_thread::yield();           // more waiting threads than cores: let the others run
      if ((it & 0xfff) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(50)) break;      // a long batch: let the runtime wait
    }
    if (*f == req.seq) { std::atomic_thread_fence(std::memory_order_acquire); return KYB_OK; }
  }
  HIPCK(hipStreamSynchronize(g.stream));
  return KYB_OK;
}
struct DoneScope {          // posts / withdraws the request around the launch sequence
  DoneReq req;
  DoneScope() { tl_done = &req; }
  ~DoneScope() { tl_done = nullptr; }
};

// Host-pointer batches of fixed-size records (run_host_batch): up to host.zero_copy_kib (4 MiB) the kernels work on page-locked host memory directly;
// up to 2^16 items the arrays are copied in, processed and copied out on the engine stream; beyond that the batch is pipelined
// over copy lanes and compute lanes (run_host_batch_pipelined).  Page-locked caller buffers (kyb_host_alloc) are handed to the DMA
// engines as they are.  Pageable ones would make every hipMemcpyAsync a blocking, single-threaded staging copy inside the runtime
// (~7 GB/s); they go through the context's own page-locked bounce ring instead, filled by CopyPool threads while the GPU works.
// One host-pointer call at a time per CONTEXT (g.mu); callers that want several in flight use several contexts.
struct HostArr { const void* in; void* out; size_t bytes; bool secret = false; };    // per-item size; exactly one of in/out, or neither = absent;
// secret: private keys / nonces / DH secrets (inputs) and shared secrets s*P (outputs of kyb_mul_batch).  Cleared before the call returns: the copies in
// the context's page-locked zero-copy buffer, in the device staging buffer and in the bounce ring.  NOT cleared per call (overwritten by the next call on the
// context, wiped when the context is released): the per-stream device scratch (projective staging records, decoded operands) and, for pageable callers
// of >= 2^16-item calls, the page-locked landing area of the results — clearing 200 MB of host memory would double such a call; pass kyb_host_alloc
// memory (no landing area) when that matters.  include/kyber_ed25519.h "secrets" says the same.
template <class F> struct ScopeExit { F f; ~ScopeExit() { f(); } };
template <class F> ScopeExit<F> on_scope_exit(F f) { return ScopeExit<F>{f}; }
constexpr size_t PIPE_MIN_ITEMS = (size_t)1 << 16;
inline size_t zero_copy_bytes(const Ctx& g) { return (size_t)g.opt_zero_copy_kib << 10; }      // host-pointer calls up to this size skip the copies: kernels work on page-locked host memory (host.zero_copy_kib)

// ---- large host-pointer batches: copy lanes and compute lanes ------------------------------------------------------
// The whole batch gets device staging of its own (a 2^20-item variable-base batch: 235 MB of 288 GB), so nothing waits for a
// staging buffer to come free:
//   stream_in    every chunk's operands, back to back (page-locked caller memory: DMA straight from it; pageable: CopyPool threads
//                fill a ring of four 4 MiB page-locked slots that the DMA engine drains behind them)
//   stream/2     the chunk's kernels as soon as its operands have arrived (event), consecutive chunks on alternating streams
//   stream_out   the chunk's results as soon as its kernels are done (page-locked: straight into caller memory; pageable: into a
//                page-locked landing area, moved to the caller by CopyPool threads while later chunks compute)
// Chunk sizes ramp up and down, 1 1 2 4 4 2 2 sixteenths of the batch: the first kernels start after a sixteenth of the input has
// crossed PCIe (0.25 ms), the middle launches fill the chip, the last result copy is an eighth of the output.  Round 2 cut the batch
// into 8 equal chunks on two lanes with ONE staging buffer each: a chunk's input copy could not start before the chunk two places
// ahead had left the device, and the timeline (tools/host_pipeline_trace.py) showed the GPU idle for a third of the call.
constexpr int PIPE_MAX_CHUNKS = 16;
int ensure_pipe(Ctx& g, int chunks) {
  if (!g.stream_in) HIPCK(hipStreamCreateWithFlags(&g.stream_in, hipStreamNonBlocking));
  if (!g.stream_out) HIPCK(hipStreamCreateWithFlags(&g.stream_out, hipStreamNonBlocking));
  while ((int)g.pipe_ev.size() < 3 * chunks) {
    hipEvent_t e;
    HIPCK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    g.pipe_ev.push_back(e);
  }
  for (int i = 0; i < 4; ++i) if (!g.ev_ring[i]) HIPCK(hipEventCreateWithFlags(&g.ev_ring[i], hipEventDisableTiming));
  return KYB_OK;
}
int ensure_pin_out(Ctx& g, size_t bytes) {
  if (bytes <= g.pin_out_bytes) return KYB_OK;
  if (g.pin_out) { HIPCK(ctx_host_free(g, g.pin_out, g.pin_out_bytes)); g.pin_out = nullptr; g.pin_out_bytes = 0; }
  const size_t want = bytes + (bytes >> 3) + 4096;
  hipError_t e = ctx_host_malloc(g, reinterpret_cast<void**>(&g.pin_out), want, hipHostMallocDefault);
  if (e != hipSuccess) return fail(KYB_E_NOMEM, "page-locked output landing area", e);
  g.pin_out_bytes = want;
  return KYB_OK;
}
// chunk sizes (items, multiples of 1024 except the last): sixteenths of the batch in the pattern above; `first_div` (option
// host.pipe_chunks, default 16) sets the unit
int plan_chunks(size_t n, int first_div, size_t* sizes) {
  static const int pattern[7] = {1, 1, 2, 4, 4, 2, 2};      // other shapes (1 1 2 4 8, 1 2 4 4 3 2, ...) are within 2 % (profiles/r03/host_pipeline.log)
  size_t unit = ((n + (size_t)first_div - 1) / (size_t)first_div + 1023) & ~(size_t)1023;
  if (unit < 16384) unit = 16384;
  int c = 0;
  size_t left = n;
  for (int i = 0; left > 0 && c < PIPE_MAX_CHUNKS; ++i) {
    size_t want = unit * (size_t)(i < 7 ? pattern[i] : 2);
    if (c == PIPE_MAX_CHUNKS - 1 || want > left) want = left;
    sizes[c++] = want;
    left -= want;
  }
  return c;
}
constexpr size_t RING_SLOT = (size_t)4 << 20;
template <class Fn>
int run_host_batch_pipelined(Ctx& g, size_t n, const HostArr* arrs, int na, Fn launch) {
  size_t sizes[PIPE_MAX_CHUNKS];
  const int nchunks = plan_chunks(n, g.opt_pipe_chunks, sizes);
  const size_t n_pad = (n + 1023) & ~(size_t)1023;
  size_t off[8], total = 0, out_off[8], out_total = 0;
  for (int k = 0; k < na; ++k) {
    off[k] = total; total += up256(arrs[k].bytes * n_pad);
    out_off[k] = out_total; if (arrs[k].out) out_total += up256(arrs[k].bytes * n_pad);
  }
  int rc = ensure_stage(g, total); if (rc) return rc;
  rc = ensure_pipe(g, nchunks); if (rc) return rc;
  bool pinned = true;
  for (int k = 0; k < na; ++k) {
    if (arrs[k].in) pinned = pinned && is_pinned(arrs[k].in);
    if (arrs[k].out) pinned = pinned && is_pinned(arrs[k].out);
  }
  const int threads = copy_threads(g);
  if (!pinned) {
    rc = ensure_pin(g, 0, 4 * RING_SLOT); if (rc) return rc;
    rc = ensure_pin_out(g, out_total); if (rc) return rc;
  }
  hipStream_t cmp[2] = {g.stream, g.stream2};
  auto quiesce = [&] { (void)hipStreamSynchronize(g.stream_in); (void)hipStreamSynchronize(g.stream); (void)hipStreamSynchronize(g.stream2); (void)hipStreamSynchronize(g.stream_out); };
  // every way out: nothing of this call is left in flight, and the secret operands are gone from the staging and the bounce ring
  auto wipe = on_scope_exit([&] {
    quiesce();
    bool any = false;
    for (int k = 0; k < na; ++k)
      if (arrs[k].secret && (arrs[k].in || arrs[k].out)) { (void)hipMemsetAsync(g.stage + off[k], 0, arrs[k].bytes * n_pad, g.stream); any = true; }
    if (any && !pinned && g.pin[0]) memset(g.pin[0], 0, 4 * RING_SLOT);
    if (any) (void)hipStreamSynchronize(g.stream);
  });
  int ring_next = 0;
  bool ring_used[4] = {false, false, false, false};
  auto bounce_in = [&](uint8_t* dst, const uint8_t* src, size_t bytes) -> int {
    for (size_t o = 0; o < bytes; o += RING_SLOT) {
      const int slot = ring_next & 3;
      const size_t nb = bytes - o < RING_SLOT ? bytes - o : RING_SLOT;
      if (ring_used[slot]) HIPCK(hipEventSynchronize(g.ev_ring[slot]));        // the DMA engine has drained this slot
      kyb::CopyPool::Job job{g.pin[0] + (size_t)slot * RING_SLOT, src + o, nb};
      g.copy.run(&job, 1, threads);
      HIPCK(hipMemcpyAsync(dst + o, g.pin[0] + (size_t)slot * RING_SLOT, nb, hipMemcpyHostToDevice, g.stream_in));
      HIPCK(hipEventRecord(g.ev_ring[slot], g.stream_in));
      ring_used[slot] = true;
      ++ring_next;
    }
    return KYB_OK;
  };
  // pageable caller: results of chunk c from the landing area to the caller's arrays, once they have left the device
  size_t lo_of[PIPE_MAX_CHUNKS];
  auto deliver = [&](int c) -> int {
    HIPCK(hipEventSynchronize(g.pipe_ev[3 * c + 2]));
    kyb::CopyPool::Job jobs[8];
    int nj = 0;
    for (int k = 0; k < na; ++k)
      if (arrs[k].out) jobs[nj++] = kyb::CopyPool::Job{static_cast<uint8_t*>(arrs[k].out) + arrs[k].bytes * lo_of[c], g.pin_out + out_off[k] + arrs[k].bytes * lo_of[c], arrs[k].bytes * sizes[c]};
    g.copy.run(jobs, nj, threads);
    return KYB_OK;
  };
  size_t lo = 0;
  int delivered = 0;
  for (int c = 0; c < nchunks; ++c) {
    const size_t cn = sizes[c];
    lo_of[c] = lo;
    hipEvent_t ev_in = g.pipe_ev[3 * c], ev_done = g.pipe_ev[3 * c + 1], ev_out = g.pipe_ev[3 * c + 2];
    uint8_t* dptr[8];
    for (int k = 0; k < na; ++k) {
      dptr[k] = (arrs[k].in || arrs[k].out) ? g.stage + off[k] + arrs[k].bytes * lo : nullptr;
      if (!arrs[k].in) continue;
      const uint8_t* src = static_cast<const uint8_t*>(arrs[k].in) + arrs[k].bytes * lo;
      if (pinned) HIPCK(hipMemcpyAsync(dptr[k], src, arrs[k].bytes * cn, hipMemcpyHostToDevice, g.stream_in));
      else { rc = bounce_in(dptr[k], src, arrs[k].bytes * cn); if (rc) return rc; }
    }
    HIPCK(hipEventRecord(ev_in, g.stream_in));
    hipStream_t st = cmp[c & 1];
    HIPCK(hipStreamWaitEvent(st, ev_in, 0));
    rc = launch(st, cn, dptr);
    if (rc) return rc;
    HIPCK(hipEventRecord(ev_done, st));
    HIPCK(hipStreamWaitEvent(g.stream_out, ev_done, 0));
    for (int k = 0; k < na; ++k) {
      if (!arrs[k].out) continue;
      uint8_t* dst = pinned ? static_cast<uint8_t*>(arrs[k].out) + arrs[k].bytes * lo : g.pin_out + out_off[k] + arrs[k].bytes * lo;
      HIPCK(hipMemcpyAsync(dst, dptr[k], arrs[k].bytes * cn, hipMemcpyDeviceToHost, g.stream_out));
    }
    HIPCK(hipEventRecord(ev_out, g.stream_out));
    // pageable: hand over whatever has already landed (never waits for a chunk that is still computing)
    while (!pinned && delivered < c && hipEventQuery(g.pipe_ev[3 * delivered + 2]) == hipSuccess) { rc = deliver(delivered++); if (rc) return rc; }
    lo += cn;
  }
  if (!pinned) for (; delivered < nchunks; ++delivered) { rc = deliver(delivered); if (rc) return rc; }
  HIPCK(hipStreamSynchronize(g.stream_out));
  return KYB_OK;
}

template <class Fn>
int run_host_batch(Ctx& g, size_t n, const HostArr* arrs, int na, Fn launch) {
  // counted AFTER the context's mutex is taken: threads queued on one shared context are not calls in flight — one of them runs at a
  // time, and counting the waiters routed a lone running call off the latency kernels
  std::lock_guard<std::mutex> lk(g.mu);
  InflightScope in_flight(g);
  if (n >= PIPE_MIN_ITEMS) return run_host_batch_pipelined(g, n, arrs, na, launch);
  const size_t cap = (n + 1023) & ~(size_t)1023;
  size_t off[8], total = 0;
  for (int k = 0; k < na; ++k) { off[k] = total; total += up256(arrs[k].bytes * cap); }
  if (total <= zero_copy_bytes(g)) {
    // small batch: kernels on the page-locked buffer itself (see HostCall::run)
    int rc = ensure_pin(g, 0, zero_copy_bytes(g));
    if (rc) return rc;
    uint8_t* dptr[8];
    bool staged[8];
    const bool try_inplace = g.opt_host_inplace != 0 && total >= INPLACE_MIN_BYTES;
    for (int k = 0; k < na; ++k) {
      dptr[k] = (arrs[k].in || arrs[k].out) ? g.pin[0] + off[k] : nullptr;
      staged[k] = dptr[k] != nullptr;
      if (try_inplace && staged[k] && (arrs[k].in == nullptr || arrs[k].out == nullptr || arrs[k].in == arrs[k].out)) {
        uint8_t* own = pinned_dev_ptr(g, arrs[k].in ? arrs[k].in : arrs[k].out, arrs[k].bytes * n);      // kyb_host_alloc memory: used where it lies
        if (own != nullptr) { dptr[k] = own; staged[k] = false; }
      }
    }
    for (int k = 0; k < na; ++k)               // an in-place array that shares bytes with an in-place array the kernels write: staged after all
      for (int j = 0; j < na && dptr[k] && !staged[k]; ++j) {
        if (j == k || !dptr[j] || staged[j] || (arrs[k].out == nullptr && arrs[j].out == nullptr)) continue;
        const void *pk = arrs[k].in ? arrs[k].in : arrs[k].out, *pj = arrs[j].in ? arrs[j].in : arrs[j].out;
        if (host_ranges_overlap(pk, arrs[k].bytes * n, pj, arrs[j].bytes * n) && (arrs[k].out == nullptr || j < k)) { dptr[k] = g.pin[0] + off[k]; staged[k] = true; }
      }
    for (int k = 0; k < na; ++k)
      if (arrs[k].in && staged[k]) memcpy(dptr[k], arrs[k].in, arrs[k].bytes * n);
    // whatever way the call ends, the secret operands do not stay behind in the page-locked buffer (the kernels have finished by then:
    // every return below is behind a completed wait or a stream synchronisation)
    auto wipe = on_scope_exit([&] { for (int k = 0; k < na; ++k) if (arrs[k].secret && staged[k]) memset(g.pin[0] + off[k], 0, arrs[k].bytes * n); });
    rc = ensure_done_flag(g);
    if (rc) return rc;
    {
      DoneScope ds;
      rc = launch(g.stream, n, dptr);
      tl_done = nullptr;
      if (rc) { (void)hipDeviceSynchronize(); return rc; }      // a failed launch sequence may have forked work onto a side stream: nothing of it is left in flight
      rc = wait_done(g, ds.req);
      if (rc) { (void)hipDeviceSynchronize(); return rc; }      // a failed launch sequence may have forked work onto a side stream: nothing of it is left in flight
    }
    for (int k = 0; k < na; ++k)
      if (arrs[k].out && staged[k]) memcpy(arrs[k].out, g.pin[0] + off[k], arrs[k].bytes * n);
    return KYB_OK;
  }
  // in between (up to 2^16 items): copy in, run, copy out, on the engine stream
  int rc = ensure_stage(g, total);
  if (rc) return rc;
  auto wipe = on_scope_exit([&] {
    bool any = false;
    (void)hipStreamSynchronize(g.stream);            // an error return may have left work in flight
    for (int k = 0; k < na; ++k)
      if (arrs[k].secret && (arrs[k].in || arrs[k].out)) { (void)hipMemsetAsync(g.stage + off[k], 0, arrs[k].bytes * cap, g.stream); any = true; }
    if (any) (void)hipStreamSynchronize(g.stream);
  });
  uint8_t* dptr[8];
  for (int k = 0; k < na; ++k) {
    dptr[k] = (arrs[k].in || arrs[k].out) ? g.stage + off[k] : nullptr;
    if (arrs[k].in) HIPCK(hipMemcpyAsync(dptr[k], arrs[k].in, arrs[k].bytes * n, hipMemcpyHostToDevice, g.stream));
  }
  rc = launch(g.stream, n, dptr);
  if (rc) return rc;
  for (int k = 0; k < na; ++k)
    if (arrs[k].out) HIPCK(hipMemcpyAsync(arrs[k].out, dptr[k], arrs[k].bytes * n, hipMemcpyDeviceToHost, g.stream));
  HIPCK(hipStreamSynchronize(g.stream));
  return KYB_OK;
}

static_assert(KYB_BASE_TABLE_BYTES == 4u * (KYB_BASE_TABLE_WORDS + KYB_BASE32_TABLE_WORDS + KYB_BASE64_TABLE_WORDS), "table image layout");
constexpr int KYB_CK_LO = KYB_BT_IDX(0, 0, 30);   // where the image carries its own checksum (two words: KYB_CK_LO, KYB_CK_LO + 1)
static_assert(KYB_BT_IDX(0, 0, 31) == KYB_CK_LO + 1, "the checksum words are adjacent");

// One synchronous host-pointer call of the small (non-pipelined) kind: the caller's arrays are laid out in the
// context's device staging buffer, inputs copied in, `body` queues the kernels on the engine stream, outputs copied
// back, stream synchronised.  An array whose host pointer is null takes no space and maps to a null device pointer.
// secret(): the inputs include private keys / nonces — the staging region is cleared before the call returns.
// One input array of a host-pointer call to the device, queued on the engine stream.  A large PAGEABLE array handed to
// hipMemcpyAsync is staged by the runtime on the calling thread (~5-15 GB/s, and the copy engine waits meanwhile); it is
// cut into chunks instead that CopyPool threads move into the context's two page-locked bounce buffers while the DMA engine
// drains the other one (the pipeline of run_host_batch, without kernels in between).
constexpr size_t H2D_PIPE_MIN = (size_t)8 << 20, H2D_PIPE_CHUNK = (size_t)8 << 20;
int h2d(Ctx& g, uint8_t* dst, const uint8_t* src, size_t bytes, bool* bounced) {
  if (bytes < H2D_PIPE_MIN || is_pinned(src)) {
    HIPCK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, g.stream));
    return KYB_OK;
  }
  int rc = ensure_pin(g, 0, H2D_PIPE_CHUNK); if (rc) return rc;
  rc = ensure_pin(g, 1, H2D_PIPE_CHUNK); if (rc) return rc;
  *bounced = true;                 // (a secret call's HostCall clears the bounce buffers on its way out)
  for (int l = 0; l < 2; ++l)
    if (!g.ev_pin[l]) HIPCK(hipEventCreateWithFlags(&g.ev_pin[l], hipEventDisableTiming));
  const int threads = copy_threads(g);
  int c = 0;
  for (size_t o = 0; o < bytes; o += H2D_PIPE_CHUNK, ++c) {
    const int lane = c & 1;
    const size_t nb = bytes - o < H2D_PIPE_CHUNK ? bytes - o : H2D_PIPE_CHUNK;
    if (c >= 2) HIPCK(hipEventSynchronize(g.ev_pin[lane]));          // the DMA engine has drained this bounce buffer
    kyb::CopyPool::Job job{g.pin[lane], src + o, nb};
    g.copy.run(&job, 1, threads);
    HIPCK(hipMemcpyAsync(dst + o, g.pin[lane], nb, hipMemcpyHostToDevice, g.stream));
    HIPCK(hipEventRecord(g.ev_pin[lane], g.stream));
  }
  // the bounce buffers are reused by whatever comes next under g.mu: the copies out of them must have been issued AND finished
  for (int l = 0; l < 2 && l < c; ++l) HIPCK(hipEventSynchronize(g.ev_pin[l]));
  return KYB_OK;
}

class HostCall {
 public:
  explicit HostCall(Ctx& g) : g_(g) {}
  int in(const void* p, size_t bytes, size_t pad = 0) { return add(p, nullptr, bytes, pad); }
  int out(void* p, size_t bytes) { return add(nullptr, p, bytes, 0); }
  int inout(const void* p_in, void* p_out, size_t bytes) { return add(p_in, p_out, bytes, 0); }      // one device array, filled from p_in and/or returned to p_out
  void secret() { secret_ = true; }
  template <class T = uint8_t>
  T* dev(int slot) const { return a_[slot].present ? reinterpret_cast<T*>(a_[slot].own ? a_[slot].own : base_ + a_[slot].off) : nullptr; }
  template <class Body>
  int run(Body body) {
    Ctx& g = g_;
    std::lock_guard<std::mutex> lk(g.mu);
    InflightScope in_flight(g);                    // after the mutex: see run_host_batch
    if (total_ <= zero_copy_bytes(g)) {
      // small call: the kernels read and write the context's page-locked buffer directly over PCIe — no hipMemcpy at all
      // (each costs ~10 us of runtime work, more than the transfer), one launch sequence and one stream synchronisation
      int rc = ensure_pin(g, 0, zero_copy_bytes(g));
      if (rc) return rc;
      base_ = g.pin[0];
      // secret(): cleared on EVERY way out (each return below is behind a completed wait or a stream synchronisation)
      auto wipe = on_scope_exit([&] { if (secret_ && total_) memset(base_, 0, total_); });
      if (g.opt_host_inplace != 0 && total_ >= INPLACE_MIN_BYTES)
        for (int i = 0; i < n_; ++i) {         // arrays in kyb_host_alloc memory are used where they lie (pinned_dev_ptr); not those the kernels may read past the end of
          Arr& a = a_[i];
          if (a.present && a.bytes && a.pad == 0 && (a.src == nullptr || a.dst == nullptr || a.src == a.dst)) a.own = pinned_dev_ptr(g, a.src ? a.src : a.dst, a.bytes);
        }
      for (int k = 0; k < n_; ++k)             // shared bytes between two in-place arrays, one of them written: that input is staged (host_ranges_overlap)
        for (int j = 0; j < n_ && a_[k].own; ++j) {
          if (j == k || !a_[j].own || (a_[k].dst == nullptr && a_[j].dst == nullptr)) continue;
          if (host_ranges_overlap(a_[k].src ? a_[k].src : a_[k].dst, a_[k].bytes, a_[j].src ? a_[j].src : a_[j].dst, a_[j].bytes) && (a_[k].dst == nullptr || j < k)) a_[k].own = nullptr;
        }
      for (int i = 0; i < n_; ++i)
        if (a_[i].src && a_[i].bytes && !a_[i].own) memcpy(base_ + a_[i].off, a_[i].src, a_[i].bytes);
      rc = ensure_done_flag(g);
      if (rc) return rc;
      {
        DoneScope ds;
        rc = body(g.stream);
        tl_done = nullptr;
        if (rc) { (void)hipDeviceSynchronize(); return rc; }      // a failed launch sequence may have forked work onto a side stream: nothing of it is left in flight
        rc = wait_done(g, ds.req);
        if (rc) { (void)hipDeviceSynchronize(); return rc; }      // a failed launch sequence may have forked work onto a side stream: nothing of it is left in flight
      }
      for (int i = 0; i < n_; ++i)
        if (a_[i].dst && a_[i].bytes && !a_[i].own) memcpy(a_[i].dst, base_ + a_[i].off, a_[i].bytes);
      return KYB_OK;
    }
    int rc = ensure_stage(g, total_);
    if (rc) return rc;
    base_ = g.stage;
    bool bounced = false;
    auto wipe = on_scope_exit([&] {
      if (!secret_ || !total_) return;
      (void)hipStreamSynchronize(g.stream);                  // an error return may have left copies or kernels in flight
      (void)hipMemsetAsync(g.stage, 0, total_, g.stream);
      if (bounced) for (int l = 0; l < 2; ++l) if (g.pin[l]) memset(g.pin[l], 0, g.pin_bytes[l] < H2D_PIPE_CHUNK ? g.pin_bytes[l] : H2D_PIPE_CHUNK);
      (void)hipStreamSynchronize(g.stream);
    });
    for (int i = 0; i < n_; ++i)
      if (a_[i].src && a_[i].bytes) { rc = h2d(g, g.stage + a_[i].off, static_cast<const uint8_t*>(a_[i].src), a_[i].bytes, &bounced); if (rc) return rc; }
    rc = body(g.stream);
    if (rc) return rc;
    for (int i = 0; i < n_; ++i)
      if (a_[i].dst && a_[i].bytes) HIPCK(hipMemcpyAsync(a_[i].dst, g.stage + a_[i].off, a_[i].bytes, hipMemcpyDeviceToHost, g.stream));
    HIPCK(hipStreamSynchronize(g.stream));
    return KYB_OK;
  }

 private:
  struct Arr { const void* src; void* dst; size_t bytes, off, pad; bool present; uint8_t* own; };      // own: the caller's page-locked array itself (zero-copy calls)
  int add(const void* src, void* dst, size_t bytes, size_t pad) {
    const bool present = src != nullptr || dst != nullptr;
    a_[n_] = Arr{src, dst, bytes, total_, pad, present, nullptr};
    if (present) total_ += up256(bytes + pad);
    return n_++;
  }
  Ctx& g_;
  uint8_t* base_ = nullptr;
  Arr a_[12];
  int n_ = 0;
  size_t total_ = 0;
  bool secret_ = false;
};
// message blobs: offsets must not decrease; returns the blob size through *mbytes
int check_messages(const uint8_t* msgs, const uint32_t* msg_off, size_t n, size_t* mbytes) {
  *mbytes = msg_off[n];
  if (*mbytes && !msgs) return fail(KYB_E_BAD_ARG, "null message buffer");
  for (size_t i = 0; i < n; ++i) if (msg_off[i + 1] < msg_off[i]) return fail(KYB_E_BAD_ARG, "msg_off must be non-decreasing");
  return KYB_OK;
}

// ---- per-stream scratch slots ---------------------------------------------------------------------------
constexpr size_t MSM_BASE_WORDS = 43 * 40, MSM_TAB_WORDS = 43 * 32 * 40;      // per point
constexpr size_t COOP_PIECES_ITEMS = 1024;             // most items a launch of k_mul_coop may cut into pieces (the per-CU limit times any CU count in use stays below)
constexpr size_t COOP_PIECES_BYTES = KYB_COOP_PIECES_WORDS(COOP_PIECES_ITEMS) * sizeof(uint32_t);
inline size_t proj_alloc_bytes(size_t items) { return items * 8 * sizeof(uint4) + 256; }
inline size_t msm_alloc_bytes(size_t points) { return points * (MSM_BASE_WORDS + MSM_TAB_WORDS) * sizeof(uint32_t); }
void free_slot(Ctx& g, StreamRes* r) {
  if (r->ev_last && r->used) (void)hipEventSynchronize(r->ev_last);      // everything that used the scratch has finished
  if (r->ws) ctx_free(g, r->ws, g.ws_bytes);
  if (r->proj) wipe_free_dev(g, r->proj, proj_alloc_bytes(r->proj_items));
  if (r->enc) wipe_free_dev(g, r->enc, r->enc_bytes);
  if (r->part) wipe_free_dev(g, r->part, r->part_items * 160);
  if (r->pieces) wipe_free_dev(g, r->pieces, COOP_PIECES_BYTES);
  if (r->msm) ctx_free(g, r->msm, msm_alloc_bytes(r->msm_points));
  if (r->pub_enc) ctx_free(g, r->pub_enc, 32 * r->pub_enc_items);
  if (r->aux) { (void)hipStreamSynchronize(r->aux); (void)hipStreamDestroy(r->aux); (void)hipEventDestroy(r->ev_fork); (void)hipEventDestroy(r->ev_join); if (r->ev_mid) (void)hipEventDestroy(r->ev_mid); }
  if (r->ev_last) (void)hipEventDestroy(r->ev_last);
  delete r;
}
// scratch bound to a stream (allocated on first use).  Called with g.launch_mu held.  Up to MAX_STREAM_SLOTS streams are
// registered at a time; beyond that the least recently used caller stream's slot is recycled (after its last launch).
int res_for(Ctx& g, hipStream_t st, StreamRes** out) {
  for (StreamRes* r : g.res) if (r->stream == st) { r->last_use = ++g.use_clock; *out = r; return KYB_OK; }
  if (g.res.size() >= MAX_STREAM_SLOTS) {
    size_t victim = g.res.size();
    for (size_t i = 0; i < g.res.size(); ++i)
      if (!g.res[i]->own && (victim == g.res.size() || g.res[i]->last_use < g.res[victim]->last_use)) victim = i;
    if (victim == g.res.size()) return fail(KYB_E_NOMEM, "no stream scratch slot can be recycled");
    free_slot(g, g.res[victim]);
    g.res.erase(g.res.begin() + (long)victim);
  }
  StreamRes* r = new StreamRes();
  r->stream = st;
  r->own = (st == g.stream || st == g.stream2);
  hipError_t e = hipEventCreateWithFlags(&r->ev_last, hipEventDisableTiming);
  if (e != hipSuccess) { delete r; return fail(KYB_E_HIP, "hipEventCreateWithFlags", e); }
  r->last_use = ++g.use_clock;
  g.res.push_back(r);
  *out = r;
  return KYB_OK;
}
// Brackets one launch sequence on a slot: the stream first waits for the slot's previous user (a no-op when that was the
// same stream; the protection when a destroyed stream's handle value comes back for a new stream), and the slot's
// ev_last is re-recorded behind everything the sequence queued.
struct SlotUse {
  StreamRes* r; hipStream_t st;
  // (the context's own streams live as long as their slots: nothing to protect, and two runtime calls less per launch sequence)
  SlotUse(StreamRes* r_, hipStream_t st_) : r(r_), st(st_) { if (r->used && !r->own) (void)hipStreamWaitEvent(st, r->ev_last, 0); }
  ~SlotUse() { if (!r->own) { (void)hipEventRecord(r->ev_last, st); r->used = true; } }
};
// the windowed-table kernel's per-wave table slots (160 MiB): only allocated if that kernel is used
int ensure_ws(Ctx& g, StreamRes* r) {
  if (r->ws) return KYB_OK;
  hipError_t e = ctx_malloc(g, reinterpret_cast<void**>(&r->ws), g.ws_bytes);
  if (e != hipSuccess) return fail(KYB_E_NOMEM, "table workspace allocation", e);
  return KYB_OK;
}
// grow-only; growth synchronises the stream first because earlier launches may still use the old buffer
int ensure_proj(Ctx& g, StreamRes* r, size_t items) {
  if (items <= r->proj_items) return KYB_OK;
  HIPCK(hipStreamSynchronize(r->stream));
  if (r->proj) wipe_free_dev(g, r->proj, proj_alloc_bytes(r->proj_items));
  r->proj = nullptr; r->proj_items = 0; r->top_or = nullptr;
  const size_t want = ((items + (items >> 3)) + 1023) & ~(size_t)1023;
  hipError_t e = ctx_malloc(g, reinterpret_cast<void**>(&r->proj), proj_alloc_bytes(want));
  if (e != hipSuccess) return fail(KYB_E_NOMEM, "projective staging allocation", e);
  r->proj_items = want;
  r->top_or = reinterpret_cast<uint32_t*>(r->proj + want * 8);
  HIPCK(hipMemsetAsync(r->top_or, 0, 256, r->stream));
  return KYB_OK;
}
// extended quads of k_mul_coop's products for k_sum_coop (40 words per item)
int ensure_ws_part(Ctx& g, StreamRes* r, size_t items) {
  if (items <= r->part_items) return KYB_OK;
  HIPCK(hipStreamSynchronize(r->stream));
  if (r->part) wipe_free_dev(g, r->part, r->part_items * 160);
  r->part = nullptr; r->part_items = 0;
  const size_t want = ((items + (items >> 3)) + 1023) & ~(size_t)1023;
  hipError_t e = ctx_malloc(g, reinterpret_cast<void**>(&r->part), want * 160);
  if (e != hipSuccess) return fail(KYB_E_NOMEM, "product staging allocation", e);
  r->part_items = want;
  return KYB_OK;
}
// records and arrival counters of k_mul_coop's four workgroups per item: allocated once per slot, zero from then on (the kernel clears what it used)
int ensure_pieces(Ctx& g, StreamRes* r) {
  if (r->pieces) return KYB_OK;
  hipError_t e = ctx_malloc(g, reinterpret_cast<void**>(&r->pieces), COOP_PIECES_BYTES);
  if (e != hipSuccess) { r->pieces = nullptr; return fail(KYB_E_NOMEM, "scratch of the four-piece multiplication", e); }
  e = hipMemsetAsync(r->pieces, 0, COOP_PIECES_BYTES, r->stream);
  if (e != hipSuccess) {       // never leave a registered buffer whose arrival counters are not zero: no workgroup would ever be the last to arrive
    ctx_free(g, r->pieces, COOP_PIECES_BYTES);
    r->pieces = nullptr;
    return fail(KYB_E_HIP, "hipMemsetAsync (scratch of the four-piece multiplication)", e);
  }
  return KYB_OK;
}
int ensure_msm(Ctx& g, StreamRes* r, size_t points) {
  if (points <= r->msm_points) return KYB_OK;
  HIPCK(hipStreamSynchronize(r->stream));
  if (r->msm) ctx_free(g, r->msm, msm_alloc_bytes(r->msm_points));
  r->msm = nullptr; r->msm_points = 0;
  const size_t want = points + (points >> 3) + 16;
  hipError_t e = ctx_malloc(g, reinterpret_cast<void**>(&r->msm), msm_alloc_bytes(want));
  if (e != hipSuccess) return fail(KYB_E_NOMEM, "point table allocation", e);
  r->msm_points = want;
  return KYB_OK;
}
int ensure_enc(Ctx& g, StreamRes* r, size_t bytes) {
  if (bytes <= r->enc_bytes) return KYB_OK;
  HIPCK(hipStreamSynchronize(r->stream));
  if (r->enc) wipe_free_dev(g, r->enc, r->enc_bytes);
  r->enc = nullptr; r->enc_bytes = 0;
  const size_t want = bytes + (bytes >> 3) + 4096;
  hipError_t e = ctx_malloc(g, reinterpret_cast<void**>(&r->enc), want);
  if (e != hipSuccess) return fail(KYB_E_NOMEM, "encoding staging allocation", e);
  r->enc_bytes = want;
  return KYB_OK;
}
int ensure_pub_enc(Ctx& g, StreamRes* r, size_t items) {
  if (items <= r->pub_enc_items) return KYB_OK;
  HIPCK(hipStreamSynchronize(r->stream));
  if (r->pub_enc) ctx_free(g, r->pub_enc, 32 * r->pub_enc_items);
  r->pub_enc = nullptr; r->pub_enc_items = 0;
  const size_t want = items + (items >> 3) + 1024;
  hipError_t e = ctx_malloc(g, reinterpret_cast<void**>(&r->pub_enc), 32 * want);
  if (e != hipSuccess) return fail(KYB_E_NOMEM, "public-key encoding buffer allocation", e);
  r->pub_enc_items = want;
  return KYB_OK;
}
int ensure_aux(Ctx& g, StreamRes* r) {
  (void)g;
  if (r->aux) return KYB_OK;
  HIPCK(hipStreamCreateWithFlags(&r->aux, hipStreamNonBlocking));
  HIPCK(hipEventCreateWithFlags(&r->ev_fork, hipEventDisableTiming));
  HIPCK(hipEventCreateWithFlags(&r->ev_join, hipEventDisableTiming));
  HIPCK(hipEventCreateWithFlags(&r->ev_mid, hipEventDisableTiming));
  return KYB_OK;
}
inline const uint32_t* image64(Ctx& g) { return g.table + KYB_BASE_TABLE_WORDS + KYB_BASE32_TABLE_WORDS; }
inline const uint32_t* coop_table(Ctx& g) { return g.table_coop; }
// the image in g.table is complete (built here, or imported and validated): derive what this context keeps beside it
int table_finalize(Ctx& g, hipStream_t st) {
  LAUNCHCK(launch::build_coop_table(image64(g), g.table_coop, st));
  HIPCK(hipStreamSynchronize(st));
  return KYB_OK;
}
inline bool use_split(Ctx& g, size_t n) { return g.opt_finish == 1 && n >= (size_t)g.opt_finish_min; }

// ---- context life cycle ------------------------------------------------------------------------------------
// image in g.table -> does the embedded checksum match?  (synchronous; used after an import)
int table_validate(Ctx& g, hipStream_t st) {
  uint64_t got = 0;
  uint32_t emb[2] = {0, 0};
  LAUNCHCK(launch::table_checksum(g.table, g.ck_dev, st));
  HIPCK(hipMemcpyAsync(&got, g.ck_dev, sizeof(got), hipMemcpyDeviceToHost, st));
  HIPCK(hipMemcpyAsync(emb, g.table + KYB_CK_LO, sizeof(emb), hipMemcpyDeviceToHost, st));
  HIPCK(hipStreamSynchronize(st));
  const uint64_t want = (uint64_t)emb[0] | ((uint64_t)emb[1] << 32);
  if (got != want) return fail(KYB_E_BAD_ARG, "base table image failed its checksum (truncated or corrupted transfer): table not installed");
  return table_finalize(g, st);
}

void ctx_release(Ctx* c) {
  if (c->device >= 0) (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  c->copy.stop();
  if (c->stamps_on) {      // the stamp buffer belongs to the caller and may be freed after this: no later launch on this device may write through it
    (void)hipDeviceSynchronize();
    (void)launch::diag_stamps_ladder(nullptr);
    (void)launch::diag_stamps_base(nullptr);
    c->stamps_on = false;
  }
  if (DeferArena* da = c->defer.exchange(nullptr)) defer_release(da);
  for (StreamRes* r : c->res) free_slot(*c, r);
  c->res.clear();
  wipe_free_dev(*c, c->stage, c->stage_bytes);
  for (int l = 0; l < 2; ++l) if (c->pin[l]) { memset(c->pin[l], 0, c->pin_bytes[l]); (void)ctx_host_free(*c, c->pin[l], c->pin_bytes[l]); }
  for (int l = 0; l < 2; ++l) if (c->ev_pin[l]) (void)hipEventDestroy(c->ev_pin[l]);
  for (int l = 0; l < 4; ++l) if (c->ev_ring[l]) (void)hipEventDestroy(c->ev_ring[l]);
  for (hipEvent_t e : c->pipe_ev) (void)hipEventDestroy(e);
  if (c->pin_out) { memset(c->pin_out, 0, c->pin_out_bytes); (void)ctx_host_free(*c, c->pin_out, c->pin_out_bytes); }
  if (c->stream_in) { (void)hipStreamSynchronize(c->stream_in); (void)hipStreamDestroy(c->stream_in); }
  if (c->stream_out) { (void)hipStreamSynchronize(c->stream_out); (void)hipStreamDestroy(c->stream_out); }
  if (c->done_flag) (void)ctx_host_free(*c, c->done_flag, 64);
  if (c->done_counter) ctx_free(*c, c->done_counter, 64);
  { std::lock_guard<std::mutex> lk(c->prof.mu);
    for (int i = 0; i < c->prof.cap; ++i) { (void)hipEventDestroy(c->prof.recs[i].a); (void)hipEventDestroy(c->prof.recs[i].b); }
    delete[] c->prof.recs; c->prof.recs = nullptr; c->prof.cap = c->prof.used = 0; c->prof.on = false; }
  if (c->table) (void)hipFree(c->table);
  if (c->ck_dev) (void)hipFree(c->ck_dev);
  if (c->table_coop) (void)hipFree(c->table_coop);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

// (The ROCm runtime maps all streams of a process onto GPU_MAX_HW_QUEUES hardware queues, 4 by default: concurrent one-item calls of
// many contexts want 16 — tools/microbench/concurrent_calls.cpp: 16k -> 72k variable-base calls/s with 16 threads.  The variable is read
// when the HIP runtime starts and belongs to the host program: this library never touches the environment; INTEGRATION.md says where to set it.)
int ctx_new(int device, bool build_table, Ctx** out) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) return fail(KYB_E_NO_DEVICE, "no HIP device visible (this engine has no CPU path)", e);
  if (device < 0 || device >= count) return fail(KYB_E_BAD_ARG, "device index out of range");
  HIPCK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIPCK(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    char buf[200];
    snprintf(buf, sizeof(buf), "device %d is %s; this library carries gfx950 code objects only", device, prop.gcnArchName);
    return fail(KYB_E_NO_DEVICE, buf);
  }
  Ctx* c = new Ctx();
  Ctx& g = *c;
  g.device = device;
  g.hw_cus = prop.multiProcessorCount;
  apply_cu_count(g, g.hw_cus);
  snprintf(g.name, sizeof(g.name), "%s", prop.name);
#define CTXCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { ctx_release(c); return fail(KYB_E_HIP, #x, e_); } } while (0)
  CTXCK(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
  CTXCK(hipStreamCreateWithFlags(&g.stream2, hipStreamNonBlocking));
  CTXCK(hipMalloc(&g.table, KYB_BASE_TABLE_BYTES));
  CTXCK(hipMalloc(&g.ck_dev, 16));
  CTXCK(hipMalloc(reinterpret_cast<void**>(&g.table_coop), KYB_COOP_TABLE_WORDS * sizeof(uint32_t)));
  // persistent grids of the windowed kernel: 2 blocks of 256 threads per CU = 2 waves per SIMD (needed to saturate
  // v_mad_u64_u32 issue, profiles/r01_valu_rates_mi355x.jsonl)
  g.ws_bytes = (size_t)(g.hw_cus * 2) * (KYB_BLOCK / 64) * (8 * 10 * 64) * sizeof(uint4);
  if (build_table) {
    CTXCK(launch::build_tables(g.table, g.stream));
    CTXCK(launch::build_coop_table(image64(g), g.table_coop, g.stream));
    CTXCK(hipStreamSynchronize(g.stream));
    g.table_ready = true;
  }
#undef CTXCK
  g.ready = true;
  { std::lock_guard<std::mutex> lk(g_reg_mu); g_all.push_back(c); }
  *out = c;
  return KYB_OK;
}
void ctx_delete(Ctx* c) {
  { std::lock_guard<std::mutex> lk(g_reg_mu);
    for (size_t i = 0; i < g_all.size(); ++i) if (g_all[i] == c) { g_all.erase(g_all.begin() + (long)i); break; }
    if (g_default == c) g_default = nullptr; }
  if (tl_cur == c) tl_cur = nullptr;
  c->ready = false;
  ctx_release(c);
}
bool ctx_is_live(Ctx* c) {
  std::lock_guard<std::mutex> lk(g_reg_mu);
  for (Ctx* x : g_all) if (x == c) return true;
  return false;
}

int do_init(int device, bool build_table) {
  static std::mutex init_mu;
  std::lock_guard<std::mutex> lk(init_mu);
  if (g_default != nullptr) {
    if (g_default->device != device)
      return fail(KYB_E_BAD_ARG, "the default context is bound to another device (kyb_init is one context per process; use kyb_ctx_create / kyb_group_create for more GPUs)");
    return KYB_OK;
  }
  Ctx* c = nullptr;
  int rc = ctx_new(device, build_table, &c);
  if (rc) return rc;
  { std::lock_guard<std::mutex> lk2(g_reg_mu); g_default = c; }
  return KYB_OK;
}

// ---- launch sequences ------------------------------------------------------------------------------------
// last: nothing is queued behind this launch in its call (it may carry the completion flag)
int launch_finish(Ctx& g, StreamRes* r, size_t n, uint8_t* oenc, int32_t* oext, hipStream_t st, size_t src_mul = 1, bool last = false) {
  if (n <= finish_coop_lim(g)) {          // few points: one per wavefront
    ProfScope ps(g, st, KID_FINISH_COOP);
    LAUNCHCK(launch::finish_coop(st, r->proj, r->proj_items, nullptr, n, oenc, oext, src_mul, last ? take_done_flag(g, st, n) : launch::DoneFlag{},
                                 ext_projective(g)));
    return KYB_OK;
  }
  ProfScope ps(g, st, KID_FINISH);
  // up to a wavefront per SIMD of finish lanes the kernel is one lane's chain: four items per inversion shorten it (k_finish4)
  if (finish_wave(g, n)) LAUNCHCK(launch::finish_wave(st, r->proj, r->proj_items, nullptr, n, oenc, oext, src_mul));
  else LAUNCHCK(launch::finish(st, r->proj, r->proj_items, n, oenc, oext, src_mul, finish_four(g, n)));
  return KYB_OK;
}

// leaves the results projective in r->proj[0, n): prep (batched inversion) -> 256-step ladder.
// npts == 0: item i multiplies point i.  npts > 0: the npts points are shared, item i multiplies point
// i mod npts (their Montgomery images live in records [n, n + npts)).
// skip_bits: leading zero bits every scalar of the launch has for public reasons (3 for values reduced mod L).
int launch_ladder_core(Ctx& g, const uint8_t* sc, const uint8_t* penc, const int32_t* pext, size_t n, uint8_t* ok, StreamRes* r, hipStream_t st,
                       size_t npts = 0, int skip_bits = 0) {
  const size_t np = npts ? npts : n;
  int rc = ensure_proj(g, r, n + npts); if (rc) return rc;
  if (penc != nullptr && npts == 0 && g.opt_ladder_y_only != 0 && n <= pair_lim(g, g.opt_ladder_pair_max) && g.opt_verify_overlap && host_load(g) < 4) {
    // Wire encodings, more SIMDs than wavefronts: the two-lane ladder starts on the y of the encodings while the decode looks for x on a side
    // stream (252 dependent squarings that used to sit in front of the ladder); a short kernel joins the two (ge_ladder_pair.h, round 4).
    rc = ensure_enc(g, r, 160 * n + 256); if (rc) return rc;
    rc = ensure_ws_part(g, r, n); if (rc) return rc;
    rc = ensure_aux(g, r); if (rc) return rc;
    int32_t* tmp = reinterpret_cast<int32_t*>(r->enc);
    uint4* state = reinterpret_cast<uint4*>(r->part);
    if (g.opt_ladder_y_only == 2) {      // one launch: ladder workgroups first, decoding workgroups behind them (on CUs of their own)
      { ProfScope ps(g, st, KID_MUL_LADDER_PAIR); LAUNCHCK(launch::mul_ladder_pair_y_dec(st, sc, n, penc, state, skip_bits, tmp, ok, ladder_lanes(g, n))); }
      { ProfScope ps(g, st, KID_LADDER_RECOVER); LAUNCHCK(launch::ladder_recover(st, sc, n, tmp, state, r->proj, r->proj_items)); }
      return KYB_OK;
    }
    HIPCK(hipEventRecord(r->ev_fork, st));                 // behind whatever the caller queued before this call (the encodings may be its output)
    HIPCK(hipStreamWaitEvent(r->aux, r->ev_fork, 0));
    { ProfScope ps(g, r->aux, KID_DECODE); LAUNCHCK(launch::decode_or_identity(r->aux, penc, n, tmp, ok)); }
    HIPCK(hipEventRecord(r->ev_join, r->aux));
    { ProfScope ps(g, st, KID_MUL_LADDER_PAIR); LAUNCHCK(launch::mul_ladder_pair_y(st, sc, n, penc, state, skip_bits, ladder_lanes(g, n))); }
    HIPCK(hipStreamWaitEvent(st, r->ev_join, 0));
    { ProfScope ps(g, st, KID_LADDER_RECOVER); LAUNCHCK(launch::ladder_recover(st, sc, n, tmp, state, r->proj, r->proj_items)); }
    return KYB_OK;
  }
  if (penc != nullptr) {           // unmarshal_binary of the operands first (ok flags; failed decodes become the neutral element)
    rc = ensure_enc(g, r, 160 * np + 256); if (rc) return rc;
    int32_t* tmp = reinterpret_cast<int32_t*>(r->enc);
    ProfScope ps(g, st, KID_DECODE);
    LAUNCHCK(launch::decode_or_identity(st, penc, np, tmp, ok));
    pext = tmp;
  } else if (ok != nullptr) {
    HIPCK(hipMemsetAsync(ok, 1, np, st));          // extended operands are taken as they are (k_mul does the same)
  }
  if (n <= pair_lim(g, g.opt_ladder_quad_max) && n <= pair_lim(g, g.opt_ladder_pair_max)) {      // (ladder.pair_max_items = 0 keeps every item on one lane)
    // fewer items still (a wavefront per SIMD at FOUR lanes per item): a step is three products deep instead of five (ge_ladder_quad.h)
    ProfScope ps(g, st, KID_MUL_LADDER_PAIR);
    LAUNCHCK(launch::mul_ladder_quad(st, sc, n, pext, npts, r->proj, r->proj_items, skip_bits));
    return KYB_OK;
  }
  if (n <= pair_lim(g, g.opt_ladder_pair_max)) {
    // more SIMDs than wavefronts: two lanes per item shorten the dependent chain and keep the base point projective — no k_mont_prep,
    // no inversion in front (ge_ladder_pair.h).  Without the prep there is no launch-wide canonical test: 256 - skip_bits steps.
    ProfScope ps(g, st, KID_MUL_LADDER_PAIR);
    LAUNCHCK(launch::mul_ladder_pair(st, sc, n, pext, npts, r->proj, r->proj_items, skip_bits));
    return KYB_OK;
  }
  // Are all scalars canonical (below 2^252)?  k_mont_prep ORs their top four bits into one of two alternating words on the way (one scalar
  // per point only), the ladder starts four bits lower when the word stayed 0 and clears the other word for the next call on this stream.
  uint32_t* top_or = nullptr;
  uint32_t* zero_next = nullptr;
  if (skip_bits == 0 && npts == 0 && g.opt_ladder_skip_canonical != 0) {
    top_or = r->top_or + (r->top_seq & 1u);
    zero_next = r->top_or + ((r->top_seq & 1u) ^ 1u);
    ++r->top_seq;
  }
  {
    ProfScope ps(g, st, KID_MONT_PREP);
    LAUNCHCK(launch::mont_prep(st, pext, np, r->proj + (npts ? n : 0), r->proj_items, top_or ? sc : nullptr, top_or));
  }
  {
    ProfScope ps(g, st, KID_MUL_LADDER);
    LAUNCHCK(launch::mul_ladder(g.opt_ladder_waves, st, sc, n, r->proj, r->proj_items, n, npts, skip_bits, top_or, zero_next));
  }
  return KYB_OK;
}

// the ceil(log2 t) halving passes that leave the sum of each group of t staging records in the group's first record
int launch_pair_sums(Ctx& g, StreamRes* r, size_t m, size_t t, hipStream_t st) {
  for (size_t len = t; len > 1;) {
    const size_t half = (len + 1) / 2;
    ProfScope ps(g, st, KID_PAIR_SUM);
    LAUNCHCK(launch::pair_sum(st, r->proj, r->proj_items, m, t, len, half));
    len = half;
  }
  return KYB_OK;
}

// out[g] = sum_j scalars[g*t + j] * P[g*t + j]  (shared == false)  or  * P[j]  (shared == true)
int launch_lincomb(Ctx& g, const uint8_t* sc, const uint8_t* penc, const int32_t* pext, bool shared, size_t m, size_t t, uint8_t* ok,
                   uint8_t* oenc, int32_t* oext, hipStream_t st) {
  if (m == 0) return KYB_OK;
  std::lock_guard<std::mutex> launch_lock(g.launch_mu);
  StreamRes* r = nullptr;
  { int rc = res_for(g, st, &r); if (rc) return rc; }
  SlotUse use(r, st);
  const size_t n = m * t;
  if (n <= coop_lim(g, g.opt_coop_max) && n <= coop_lim(g, g.opt_coop_ladder_max) && g.opt_mul_algo == 1) {
    // few products: one per wavefront, handed over projective (no inversion) to the halving passes
    const size_t np = shared ? t : n;
    int rc = ensure_proj(g, r, n); if (rc) return rc;
    if (penc != nullptr) {
      rc = ensure_enc(g, r, 160 * np + 256); if (rc) return rc;
      int32_t* tmp = reinterpret_cast<int32_t*>(r->enc);
      { ProfScope ps(g, st, KID_DECODE_COOP); LAUNCHCK(launch::decode_coop(st, penc, np, tmp, ok, true)); }
      pext = tmp;
    } else if (ok != nullptr) {
      HIPCK(hipMemsetAsync(ok, 1, np, st));
    }
    if (t <= 32) {
      // short sums: the products as extended quads, one wavefront adds a group up and finishes (two launches in all)
      rc = ensure_ws_part(g, r, n); if (rc) return rc;
      { ProfScope ps(g, st, KID_MUL_COOP);
        LAUNCHCK(launch::mul_coop(st, sc, pext, n, nullptr, nullptr, 0, nullptr, 0, 0, launch::DoneFlag{}, shared ? t : 0, 1, false, r->part)); }
      ProfScope ps(g, st, KID_FINISH_COOP);
      LAUNCHCK(launch::sum_coop(st, r->part, nullptr, m, t, oenc, oext, ext_projective(g), take_done_flag(g, st, m)));
      return KYB_OK;
    }
    ProfScope ps(g, st, KID_MUL_COOP);
    LAUNCHCK(launch::mul_coop(st, sc, pext, n, nullptr, nullptr, 0, r->proj, r->proj_items, 0, launch::DoneFlag{}, shared ? t : 0));
  } else {
    int rc = launch_ladder_core(g, sc, penc, pext, n, ok, r, st, shared ? t : 0); if (rc) return rc;
  }
  { int rc = launch_pair_sums(g, r, m, t, st); if (rc) return rc; }
  return launch_finish(g, r, m, oenc, oext, st, t, true);
}

// (launch_mu held, slot in use)  ext_item_major: pext holds t rows of m points (point j of group g = record j*m + g), transposed on the way
int sum_locked(Ctx& g, StreamRes* r, const int32_t* pext, const uint8_t* penc, uint8_t* ok, bool item_major, size_t m, size_t t, uint8_t* oenc, int32_t* oext,
               hipStream_t st, bool ext_item_major = false) {
  const size_t n = m * t;
  const bool small = t <= 32 && m <= coop_lim(g, g.opt_coop_base_max);
  if (penc != nullptr && small && !item_major) {
    int rc = ensure_ws_part(g, r, n); if (rc) return rc;
    int32_t* dec = reinterpret_cast<int32_t*>(r->part);
    if (n <= coop_lim(g, g.opt_coop_decode_max)) {
      ProfScope ps(g, st, KID_DECODE_COOP);
      LAUNCHCK(launch::decode_coop(st, penc, n, dec, ok, true));
    } else {
      ProfScope ps(g, st, KID_DECODE);
      LAUNCHCK(launch::decode_or_identity(st, penc, n, dec, ok));
    }
    pext = dec;
    penc = nullptr;
  }
  if (penc == nullptr && small && !ext_item_major) {
    // short sums of few groups: one group per wavefront, one launch
    ProfScope ps(g, st, KID_FINISH_COOP);
    LAUNCHCK(launch::sum_coop(st, nullptr, pext, m, t, oenc, oext, ext_projective(g), take_done_flag(g, st, m)));
    return KYB_OK;
  }
  { int rc = ensure_proj(g, r, n); if (rc) return rc; }
  if (penc != nullptr) {
    ProfScope ps(g, st, KID_DECODE);
    LAUNCHCK(launch::decode_to_proj(st, penc, n, r->proj, r->proj_items, ok, item_major ? t : 0, item_major ? m : 0));
  } else {
    LAUNCHCK(launch::ext_to_proj(st, pext, n, r->proj, r->proj_items, ext_item_major ? t : 0, ext_item_major ? m : 0));
  }
  { int rc = launch_pair_sums(g, r, m, t, st); if (rc) return rc; }
  return launch_finish(g, r, m, oenc, oext, st, t, true);
}
// kyb_lincomb_public_batch: the same sums for scalars the caller declares PUBLIC.  Over shared points with enough outputs to pay for them,
// every point gets a table of the multiples 1 .. 32 of its 43 radix-64 window bases (kernels_msm.hip) and a product costs 43 table
// additions instead of a 255-step ladder; anything else takes the constant-time path.
int launch_lincomb_public(Ctx& g, const uint8_t* sc, const uint8_t* penc, const int32_t* pext, bool shared, size_t m, size_t t, uint8_t* ok,
                          uint8_t* oenc, int32_t* oext, hipStream_t st) {
  if (m == 0) return KYB_OK;
  const bool tables = shared && g.opt_mul_algo == 1 && m >= 16 && t >= 2 && t <= 8192 && m * t >= 8192;
  if (!tables) return launch_lincomb(g, sc, penc, pext, shared, m, t, ok, oenc, oext, st);
  std::lock_guard<std::mutex> launch_lock(g.launch_mu);
  StreamRes* r = nullptr;
  { int rc = res_for(g, st, &r); if (rc) return rc; }
  SlotUse use(r, st);
  if (penc != nullptr) {           // unmarshal_binary of the shared points first (ok flags; failed decodes become the neutral element)
    int rc = ensure_enc(g, r, 160 * t + 256); if (rc) return rc;
    int32_t* tmp = reinterpret_cast<int32_t*>(r->enc);
    if (t <= coop_lim(g, g.opt_coop_decode_max)) { ProfScope ps(g, st, KID_DECODE_COOP); LAUNCHCK(launch::decode_coop(st, penc, t, tmp, ok, true)); }
    else { ProfScope ps(g, st, KID_DECODE); LAUNCHCK(launch::decode_or_identity(st, penc, t, tmp, ok)); }
    pext = tmp;
  } else if (ok != nullptr) {
    HIPCK(hipMemsetAsync(ok, 1, t, st));
  }
  int rc = ensure_msm(g, r, t); if (rc) return rc;
  uint32_t* bases = r->msm;
  uint32_t* tab = r->msm + r->msm_points * MSM_BASE_WORDS;
  // lanes of the accumulation: about two wavefronts per SIMD; a lane takes `chunk` points of one output
  size_t chunk = (m * t + 131071) / 131072;
  if (chunk < 2) chunk = 2;
  if (chunk > t) chunk = t;
  const size_t nchunks = (t + chunk - 1) / chunk;
  rc = ensure_proj(g, r, m * nchunks); if (rc) return rc;
  { ProfScope ps(g, st, KID_MSM_TABLES);
    LAUNCHCK(launch::msm_bases_coop(st, pext, t, bases));
    LAUNCHCK(launch::msm_tables(st, bases, t, tab)); }
  { ProfScope ps(g, st, KID_MSM_ACCUMULATE);
    LAUNCHCK(launch::msm_accumulate(st, sc, tab, m, t, (int)chunk, nchunks, r->proj, r->proj_items)); }
  rc = launch_pair_sums(g, r, m, nchunks, st); if (rc) return rc;
  return launch_finish(g, r, m, oenc, oext, st, nchunks, true);
}

// out[g] = sum_j P[g*t + j]: the halving passes of launch_lincomb on the points themselves.
// penc != nullptr: the points come as wire encodings (decoded here, ok[i] per encoding, failed decodes = neutral element);
// item_major (encodings only): point j of group g is encoding j*m + g — t dealers' polynomials of m coefficients each, as received.
int launch_sum(Ctx& g, const int32_t* pext, const uint8_t* penc, uint8_t* ok, bool item_major, size_t m, size_t t, uint8_t* oenc, int32_t* oext, hipStream_t st) {
  if (m == 0) return KYB_OK;
  std::lock_guard<std::mutex> launch_lock(g.launch_mu);
  StreamRes* r = nullptr;
  { int rc = res_for(g, st, &r); if (rc) return rc; }
  SlotUse use(r, st);
  return sum_locked(g, r, pext, penc, ok, item_major, m, t, oenc, oext, st);
}

// skip_hint: leading zero bits EVERY scalar of the call has (host-pointer calls of a few items look; 0 = unknown).  multipliers_public
// (kyb_mul_public_batch, or the option mul.short_scalars for callers that cannot say): when all of them are below 2^64 — share indices, the
// cofactor — the ladder of the one-item-per-wavefront kernel starts below the zeros: Point::mul(x_i, Some(v)) of PubPoly::eval takes 10 steps,
// not 255.  Without that declaration the step count never depends on the scalars' values beyond the canonical-range test (ladder.skip_canonical).
int launch_mul(Ctx& g, const uint8_t* sc, const uint8_t* penc, const int32_t* pext, size_t n, uint8_t* oenc, int32_t* oext, uint8_t* ok, hipStream_t st,
               int skip_hint = 0, bool multipliers_public = false) {
  if (n == 0) return KYB_OK;
  std::lock_guard<std::mutex> launch_lock(g.launch_mu);
  StreamRes* r = nullptr;
  { int rc = res_for(g, st, &r); if (rc) return rc; }
  SlotUse use(r, st);
  const size_t coop_to = penc != nullptr && skip_hint < 192 ? coop_lim(g, g.opt_coop_ladder_enc_max) : (penc == nullptr && skip_hint < 192 ? ladder_coop_lim(g) : coop_lim(g, g.opt_coop_ladder_max));
  if (n <= coop_lim(g, g.opt_coop_max) && n <= coop_to && g.opt_mul_algo == 1) {
    // small batch: one item per wavefront, the whole multiplication in one launch (kernels_coop.hip)
    if (penc != nullptr && 4 * n <= coop_lim(g, g.opt_coop_max)) {
      // from the wire encoding, two wavefronts per item: the ladder starts on y while the decode is still looking for x
      ProfScope ps(g, st, KID_MUL_COOP);
      LAUNCHCK(launch::mul_enc_coop(st, sc, penc, n, oenc, oext, ok, take_done_flag(g, st, n)));
      return KYB_OK;
    }
    if (penc != nullptr) {
      int rc = ensure_enc(g, r, 160 * n + 256); if (rc) return rc;
      int32_t* tmp = reinterpret_cast<int32_t*>(r->enc);
      { ProfScope ps(g, st, KID_DECODE_COOP); LAUNCHCK(launch::decode_coop(st, penc, n, tmp, ok, true)); }
      pext = tmp;
    } else if (ok != nullptr) {
      HIPCK(hipMemsetAsync(ok, 1, n, st));
    }
    ProfScope ps(g, st, KID_MUL_COOP);
    const bool short_scalars = skip_hint >= 192 && (multipliers_public || g.opt_mul_short_scalars != 0);
    // (canonical scalars, all below 2^252: the top wavefront's piece starts four bits lower, as the batch ladder does)
    const int canon_skip = (skip_hint >= 4 && g.opt_ladder_skip_canonical != 0) ? 4 : 0;
    // very few items: four single-wavefront workgroups share an item's scalar — while the 4 n workgroups find compute units of their own, or nearly
    // (measured on 256 CUs, tools/mul_coop_pieces_probe.py, with the row moves on permlane swaps: 142 against 182 us up to 16 items, 164 / 188 at 128,
    // 184 / 192 at 256, 216 / 195 at 384 — profiles/r06/coop_pieces_crossover.log; round 5, rows through the LDS crossbar: 191 / 199 at 128, 210 / 200 at 192)
    const bool in_pieces = !short_scalars && 2 * n <= coop_lim(g, g.opt_coop_verify_max) && n <= COOP_PIECES_ITEMS;
    if (in_pieces) { int rc = ensure_pieces(g, r); if (rc) return rc; }
    LAUNCHCK(launch::mul_coop(st, sc, pext, n, oenc, oext, short_scalars ? (skip_hint > 255 ? 255 : skip_hint) : canon_skip, nullptr, 0, 0, take_done_flag(g, st, n), 0,
                              in_pieces ? 4 : 1, ext_projective(g), nullptr, in_pieces ? r->pieces : nullptr));
    return KYB_OK;
  }
#ifdef KYB_CROSSCHECK
  if (g.opt_mul_algo == 1)
#endif
  {
    int rc = launch_ladder_core(g, sc, penc, pext, n, ok, r, st); if (rc) return rc;
    return launch_finish(g, r, n, oenc, oext, st, 1, true);
  }
#ifdef KYB_CROSSCHECK      // mul.algo = 0: the windowed-table kernel (ge.rs structure), kept as a cross-check of the ladder
  { int rc = ensure_ws(g, r); if (rc) return rc; }
  const size_t nchunks = (n + KYB_BLOCK - 1) / KYB_BLOCK;
  const int grid = (int)(nchunks < (size_t)g.grid_mul ? nchunks : (size_t)g.grid_mul);
  const bool split = use_split(g, n);
  if (split) { int rc = ensure_proj(g, r, n); if (rc) return rc; }
  {
    ProfScope ps(g, st, KID_MUL);
    LAUNCHCK(launch::mul_window(g.opt_mul_select, penc != nullptr, split, grid, st, sc, penc, pext, n, oenc, oext, ok, r->ws, r->proj, r->proj_items));
  }
  if (split) return launch_finish(g, r, n, oenc, oext, st, 1, true);
  return KYB_OK;
#endif
}

// fixed-base multiplication of n scalars; split leaves the points in r->proj at [offset, offset + n)
// sc_b != nullptr: a second array of n_b scalars follows the first in the same launch (radix-64 kernel), their
// results land behind the first n
// parts_at: where the mid-size form (base_quarters) may put its 3 n partial points; NO_PARTS: behind the results when offset == 0, else that form is not used
constexpr size_t NO_PARTS = ~(size_t)0;
int launch_base(Ctx& g, bool split, const uint8_t* sc, size_t n, uint8_t* oenc, int32_t* oext, StreamRes* r, size_t offset, hipStream_t st,
                const uint8_t* sc_b = nullptr, size_t n_b = 0, size_t parts_at = NO_PARTS) {
  const int radix = g.opt_base_radix, finish_min = g.opt_finish_min;
  if (sc_b != nullptr && !(radix == 64 && n + n_b >= (size_t)finish_min)) {
    int rc = launch_base(g, split, sc, n, oenc, oext, r, offset, st); if (rc) return rc;
    return launch_base(g, split, sc_b, n_b, oenc ? oenc + 32 * n : nullptr, oext ? oext + 40 * n : nullptr, r, offset + n, st);
  }
  if (radix == 64 && n + n_b >= (size_t)finish_min) {
    const size_t n_a = n;
    n += n_b;
    // one workgroup per CU (the table is its whole LDS); 256-thread workgroups while that leaves CUs idle
    const uint4* img64 = reinterpret_cast<const uint4*>(g.table + KYB_BASE_TABLE_WORDS + KYB_BASE32_TABLE_WORDS);
    const bool small = n <= (size_t)256 * (size_t)g.cus * (size_t)g.opt_base_small_chunks;
    const int b64 = g.opt_base_block64;
    const int block = small ? 256 : ((b64 == 512 || b64 == 768) ? b64 : 1024);
    const size_t nchunks64 = (n + block - 1) / block;                   // workgroups' worth of items (the kernel deals them out per wavefront)
    const int grid64 = (int)(nchunks64 < (size_t)g.cus ? nchunks64 : (size_t)g.cus);
    ProfScope ps(g, st, KID_MUL_BASE);
    const size_t parts = parts_at != NO_PARTS ? parts_at : n;
    if (split && (offset == 0 || parts_at != NO_PARTS) && base_quarters(g, n) && r->proj_items >= parts + 3 * n) {      // (the callers that want this form ask ensure_proj for the room)
      const size_t groups = (n + 63) / 64;
      LAUNCHCK(launch::mul_base64_quarters((int)(groups < (size_t)g.cus ? groups : (size_t)g.cus), st, sc, sc_b, n_a, n, img64, r->proj, r->proj_items, offset, parts));
      return KYB_OK;
    }
    LAUNCHCK(launch::mul_base64(split, block, grid64, st, sc, sc_b, n_a, n, oenc, oext, img64, r->proj, r->proj_items, offset));
    return KYB_OK;
  }
#ifndef KYB_CROSSCHECK
  return fail(KYB_E_BAD_ARG, "fixed base: only the radix-64 kernel is part of this build");      // unreachable: the options that lead here exist in the cross-check build only
#else
  if (radix == 32 && n >= (size_t)finish_min) {
    const uint4* img32 = reinterpret_cast<const uint4*>(g.table + KYB_BASE_TABLE_WORDS);
    const size_t nchunks32 = (n + KYB_BLOCK32 - 1) / KYB_BLOCK32;
    const int grid32 = (int)(nchunks32 < (size_t)g.cus ? nchunks32 : (size_t)g.cus);     // one workgroup per CU: the table fills its LDS
    ProfScope ps(g, st, KID_MUL_BASE);
    LAUNCHCK(launch::mul_base32(split, grid32, st, sc, n, oenc, oext, img32, r->proj, r->proj_items, offset));
    return KYB_OK;
  }
  const uint4* img = reinterpret_cast<const uint4*>(g.table);
  const int block = g.opt_base_block;
  const size_t nchunks = (n + block - 1) / block;
  const size_t cap = (size_t)g.cus * 2;                       // 2 blocks per CU: LDS holds two 64 KiB tables
  const int grid = (int)(nchunks < cap ? nchunks : cap);
  ProfScope ps(g, st, KID_MUL_BASE);
  LAUNCHCK(launch::mul_base16(g.opt_base_select, block, split, grid, st, sc, n, oenc, oext, img, r->proj, r->proj_items, offset));
  return KYB_OK;
#endif
}
int launch_mul_base(Ctx& g, const uint8_t* sc, size_t n, uint8_t* oenc, int32_t* oext, hipStream_t st) {
  if (n == 0) return KYB_OK;
  std::lock_guard<std::mutex> launch_lock(g.launch_mu);
  StreamRes* r = nullptr;
  { int rc = res_for(g, st, &r); if (rc) return rc; }
  SlotUse use(r, st);
  if (n <= base_coop_lim(g)) {
    ProfScope ps(g, st, KID_MUL_BASE_COOP);
    LAUNCHCK(launch::mul_base_coop(st, sc, n, oenc, oext, coop_table(g), nullptr, 0, 0, nullptr, 0, take_done_flag(g, st, n),
                                   n <= 2 * coop_lim(g, g.opt_coop_verify_max) ? 4 : 1,      // few items (measured: up to 1,024): four wavefronts share an item's 43 windows
                                   ext_projective(g)));
    return KYB_OK;
  }
  if (use_split(g, n)) {
    int rc = ensure_proj(g, r, base_quarters(g, n) ? 4 * n : n); if (rc) return rc;
    rc = launch_base(g, true, sc, n, nullptr, nullptr, r, 0, st); if (rc) return rc;
    return launch_finish(g, r, n, oenc, oext, st, 1, true);
  }
  return launch_base(g, false, sc, n, oenc, oext, r, 0, st);
}
// marshal_binary of n extended points: one shared inversion per FINISH_K points
int launch_encode(Ctx& g, const int32_t* pext, size_t n, uint8_t* oenc, hipStream_t st) {
  if (n == 0) return KYB_OK;
  if (n <= finish_coop_lim(g)) {
    ProfScope ps(g, st, KID_FINISH_COOP);
    LAUNCHCK(launch::finish_coop(st, nullptr, 0, pext, n, oenc, nullptr, 1, take_done_flag(g, st, n)));
    return KYB_OK;
  }
  if (g.opt_encode_batched == 1) {
    ProfScope ps(g, st, KID_ENCODE);
    if (finish_wave(g, n)) LAUNCHCK(launch::finish_wave(st, nullptr, 0, pext, n, oenc, nullptr, 1));
    else LAUNCHCK(launch::encode_batched(st, pext, n, oenc, finish_four(g, n)));
  } else {
    LAUNCHCK(launch::encode(st, pext, n, oenc));
  }
  return KYB_OK;
}
// PointCanCheckCanonicalAndSmallOrder for n points (point.rs:286-337): from received encodings (bytes only), or from the limbs of points the caller
// holds — has_small_order(&self) marshals the point first (point.rs:287-290), so do we, into the stream's scratch
int launch_point_checks(Ctx& g, const uint8_t* enc, const int32_t* pext, size_t n, uint8_t* flags, hipStream_t st) {
  if (n == 0) return KYB_OK;
  if (pext == nullptr) { LAUNCHCK(launch::point_checks(st, enc, n, flags)); return KYB_OK; }
  std::lock_guard<std::mutex> launch_lock(g.launch_mu);
  StreamRes* r = nullptr;
  { int rc = res_for(g, st, &r); if (rc) return rc; }
  SlotUse use(r, st);
  { int rc = ensure_enc(g, r, 32 * n + 256); if (rc) return rc; }
  if (n <= finish_coop_lim(g)) {
    ProfScope ps(g, st, KID_FINISH_COOP);
    LAUNCHCK(launch::finish_coop(st, nullptr, 0, pext, n, r->enc, nullptr, 1, launch::DoneFlag{}));      // not the call's last kernel: no completion flag
  } else {
    ProfScope ps(g, st, KID_ENCODE);
    if (finish_wave(g, n)) LAUNCHCK(launch::finish_wave(st, nullptr, 0, pext, n, r->enc, nullptr, 1));
    else LAUNCHCK(launch::encode_batched(st, pext, n, r->enc, finish_four(g, n)));
  }
  LAUNCHCK(launch::point_checks(st, r->enc, n, flags));
  return KYB_OK;
}
// schnorr::sign for n (x, k, msg) triples.  pub_in != nullptr: the callers' stored public keys enc(x*B) are
// hashed as they are and A is not recomputed (EdDSA::sign, eddsa_sig.rs:132-137; DSS long-term keys);
// pub_out != nullptr: receives enc(x*B).
int sign_locked(Ctx& g, StreamRes* r, const uint8_t* x, const uint8_t* k, const uint8_t* pub_in, const uint8_t* msgs, const uint32_t* off, size_t n,
                uint8_t* sig, uint8_t* pub_out, hipStream_t st) {
  if (n <= coop_lim(g, g.opt_coop_verify_max) && 4 * n <= 3 * base_coop_lim(g)) {
    // few signatures: one launch, two wavefronts each (kernels_coop.hip)
    ProfScope ps(g, st, KID_SIGN_COOP);
    LAUNCHCK(launch::sign_coop(st, x, k, pub_in, msgs, off, n, sig, pub_out, coop_table(g), take_done_flag(g, st, n)));
    return KYB_OK;
  }
  if (pub_in != nullptr) {
    // R = k*B only
    int rc = ensure_enc(g, r, 32 * n); if (rc) return rc;
    if (n <= base_coop_lim(g)) {
      ProfScope ps(g, st, KID_MUL_BASE_COOP);
      LAUNCHCK(launch::mul_base_coop(st, k, n, r->enc, nullptr, coop_table(g)));
    } else if (use_split(g, n)) {
      rc = ensure_proj(g, r, base_quarters(g, n) ? 4 * n : n); if (rc) return rc;
      rc = launch_base(g, true, k, n, nullptr, nullptr, r, 0, st); if (rc) return rc;
      rc = launch_finish(g, r, n, r->enc, nullptr, st); if (rc) return rc;
    } else {
      rc = launch_base(g, false, k, n, r->enc, nullptr, r, 0, st); if (rc) return rc;
    }
    {
      ProfScope ps(g, st, KID_SIGN_HASH);
      const bool last = pub_out == nullptr || pub_out == pub_in;
      LAUNCHCK(launch::sign_hash(st, x, k, msgs, off, n, r->enc, pub_in, sig, last ? take_done_flag(g, st, n) : launch::DoneFlag{}));
    }
    if (pub_out != nullptr && pub_out != pub_in) HIPCK(hipMemcpyAsync(pub_out, pub_in, 32 * n, hipMemcpyDeviceToDevice, st));
    return KYB_OK;
  }
  if (4 * n <= 3 * base_coop_lim(g)) {      // (signing crosses a little later than a bare multiplication: 118 / 122 us at 768 signatures, 123 / 124 at 1,024, 150 / 130 at 1,536)
    // small batch: the 2n fixed-base multiplications as 2n wavefronts of the cooperative kernel, encodings straight out
    int rc = ensure_enc(g, r, 64 * n); if (rc) return rc;
    {
      ProfScope ps(g, st, KID_MUL_BASE_COOP);
      LAUNCHCK(launch::mul_base_coop(st, k, n, r->enc, nullptr, coop_table(g), nullptr, 0, 0, x, n));      // R in [0, n), A in [n, 2n)
    }
    {
      ProfScope ps(g, st, KID_SIGN_HASH);
      LAUNCHCK(launch::sign_hash(st, x, k, msgs, off, n, r->enc, r->enc + 32 * n, sig, pub_out == nullptr ? take_done_flag(g, st, n) : launch::DoneFlag{}));
    }
    if (pub_out != nullptr) HIPCK(hipMemcpyAsync(pub_out, r->enc + 32 * n, 32 * n, hipMemcpyDeviceToDevice, st));
    return KYB_OK;
  }
  if (use_split(g, 2 * n)) {
    // R = k*B -> proj[0, n), A = x*B -> proj[n, 2n); one batched finish; then hash + scalar arithmetic
    int rc = ensure_proj(g, r, base_quarters(g, 2 * n) ? 8 * n : 2 * n); if (rc) return rc;
    rc = ensure_enc(g, r, 64 * n); if (rc) return rc;
    rc = launch_base(g, true, k, n, nullptr, nullptr, r, 0, st, x, n); if (rc) return rc;
    rc = launch_finish(g, r, 2 * n, r->enc, nullptr, st); if (rc) return rc;
    {
      ProfScope ps(g, st, KID_SIGN_HASH);
      LAUNCHCK(launch::sign_hash(st, x, k, msgs, off, n, r->enc, r->enc + 32 * n, sig));
    }
    if (pub_out != nullptr) HIPCK(hipMemcpyAsync(pub_out, r->enc + 32 * n, 32 * n, hipMemcpyDeviceToDevice, st));
    return KYB_OK;
  }
#ifndef KYB_CROSSCHECK
  return fail(KYB_E_BAD_ARG, "signing: only the split form is part of this build");      // unreachable: finish.batched = 0 exists in the cross-check build only
#else
  const size_t nchunks = (n + KYB_BLOCK - 1) / KYB_BLOCK;
  const size_t cap = (size_t)g.cus * 2;
  const int grid = (int)(nchunks < cap ? nchunks : cap);
  const uint4* img = reinterpret_cast<const uint4*>(g.table);
  {
    ProfScope ps(g, st, KID_SIGN);
    LAUNCHCK(launch::sign_fused(g.opt_base_select, grid, st, x, k, msgs, off, n, sig, img));
  }
  if (pub_out != nullptr) return launch_base(g, false, x, n, pub_out, nullptr, r, 0, st);
  return KYB_OK;
#endif
}
int launch_sign(Ctx& g, const uint8_t* x, const uint8_t* k, const uint8_t* pub_in, const uint8_t* msgs, const uint32_t* off, size_t n, uint8_t* sig, hipStream_t st) {
  if (n == 0) return KYB_OK;
  std::lock_guard<std::mutex> launch_lock(g.launch_mu);
  StreamRes* r = nullptr;
  { int rc = res_for(g, st, &r); if (rc) return rc; }
  SlotUse use(r, st);
  return sign_locked(g, r, x, k, pub_in, msgs, off, n, sig, nullptr, st);
}

// PriPoly::eval / shares (poly.rs:133-152) of m secret polynomials at k public indices: the Horner chains cut into segments so that
// evaluations x segments fill the chip (a lane per segment; kernels_verify.hip), partial values in the stream's scratch, cleared by the sum.
int launch_pripoly_eval(Ctx& g, const uint8_t* coeffs, size_t m, size_t t, const uint32_t* indices, size_t k, uint8_t* out, hipStream_t st) {
  if (m == 0 || k == 0) return KYB_OK;
  std::lock_guard<std::mutex> launch_lock(g.launch_mu);
  StreamRes* r = nullptr;
  { int rc = res_for(g, st, &r); if (rc) return rc; }
  SlotUse use(r, st);
  const size_t evals = m * k;
  size_t segs = evals >= 65536 ? 1 : 65536 / evals;
  if (segs > t / 8) segs = t / 8;                     // a segment is worth its 25 multiplications of x^(s len) only when it is a few coefficients long
  if (segs > 64) segs = 64;
  if (segs < 1) segs = 1;
  if (segs > 1) { int rc = ensure_enc(g, r, 32 * evals * segs + 256); if (rc) return rc; }
  ProfScope ps(g, st, KID_PRIPOLY_EVAL);
  LAUNCHCK(launch::pripoly_eval(st, coeffs, m, t, indices, k, (uint32_t)segs, static_cast<uint8_t*>(r->enc), out));
  return KYB_OK;
}

// EdDSA::sign for n (seed, msg) pairs: expansion + nonce, then the Schnorr pipeline.  pub_in: the public keys
// the EdDSA objects hold (eddsa_sig.rs:22-29), or nullptr = derive them here; pub_out: optional copy of them.
int launch_eddsa_sign(Ctx& g, const uint8_t* seeds, const uint8_t* pub_in, const uint8_t* msgs, const uint32_t* off, size_t n, uint8_t* sig, uint8_t* pub_out, hipStream_t st) {
  if (n == 0) return KYB_OK;
  std::lock_guard<std::mutex> launch_lock(g.launch_mu);
  StreamRes* r = nullptr;
  int rc = res_for(g, st, &r); if (rc) return rc;
  SlotUse use(r, st);
  // the signing pipeline uses r->enc[0, 64n) for the encodings of R and A: keep x and k behind that
  rc = ensure_enc(g, r, up256(64 * n) + 2 * up256(32 * n)); if (rc) return rc;
  uint8_t* xbuf = r->enc + up256(64 * n);
  uint8_t* kbuf = xbuf + up256(32 * n);
  {
    ProfScope ps(g, st, KID_EDDSA_PREP);
    LAUNCHCK(launch::eddsa_prep(st, seeds, msgs, off, n, xbuf, kbuf));
  }
  rc = sign_locked(g, r, xbuf, kbuf, pub_in, msgs, off, n, sig, pub_out, st);
  // the expanded secret scalars and the nonces do not outlive the call
  HIPCK(hipMemsetAsync(xbuf, 0, 2 * up256(32 * n), st));
  return rc;
}

// verification pipeline on one stream: prep -> ladder (h, A) -> fixed base (s) -> final
// pubs_ext != nullptr (pubs == nullptr): the public keys are the callers' POINTS (schnorr::verify / eddsa::verify, verify.h): they are
// marshalled here (one shared inversion per 8), and the batch path then skips the square root of unmarshalling them again; the small-batch
// kernels simply run on the encodings.
int launch_verify(Ctx& g, const uint8_t* pubs, const uint8_t* msgs, const uint32_t* off, const uint8_t* sigs, size_t n, int flavor,
                  uint8_t* status, hipStream_t st, const int32_t* pubs_ext = nullptr) {
  if (n == 0) return KYB_OK;
  std::lock_guard<std::mutex> launch_lock(g.launch_mu);
  StreamRes* r = nullptr;
  { int rc = res_for(g, st, &r); if (rc) return rc; }
  SlotUse use(r, st);
  if (pubs_ext != nullptr) {
    int rc = ensure_pub_enc(g, r, n); if (rc) return rc;
    if (n <= finish_coop_lim(g)) {
      ProfScope ps(g, st, KID_FINISH_COOP);
      LAUNCHCK(launch::finish_coop(st, nullptr, 0, pubs_ext, n, r->pub_enc, nullptr, 1));          // (not the call's last kernel: no completion flag)
    } else {
      ProfScope ps(g, st, KID_ENCODE);
      if (finish_wave(g, n)) LAUNCHCK(launch::finish_wave(st, nullptr, 0, pubs_ext, n, r->pub_enc, nullptr, 1));
      else LAUNCHCK(launch::encode_batched(st, pubs_ext, n, r->pub_enc, finish_four(g, n)));
    }
    pubs = r->pub_enc;
  }
  if (g.opt_mul_algo == 1 && n <= coop_lim(g, g.opt_coop_verify_max)) {
    // few signatures: the whole verification in one launch, three wavefronts per signature (kernels_coop.hip)
    ProfScope ps(g, st, KID_VERIFY_COOP);
    LAUNCHCK(launch::verify_coop(st, pubs, sigs, msgs, off, n, flavor, coop_table(g), status, take_done_flag(g, st, n)));
    return KYB_OK;
  }
  int rc = ensure_proj(g, r, base_quarters(g, n) ? 6 * n : 3 * n); if (rc) return rc;      // (h A | s B | R, and room for the partial points of s B's mid-size form)
  const size_t o_h = 0, o_s = up256(32 * n), o_a = o_s + up256(32 * n), o_fa = o_a + up256(160 * n), o_fr = o_fa + up256(n), o_ok = o_fr + up256(n);
  rc = ensure_enc(g, r, o_ok + up256(n)); if (rc) return rc;
  uint8_t* hbuf = r->enc + o_h; uint8_t* sbuf = r->enc + o_s; int32_t* a_ext = reinterpret_cast<int32_t*>(r->enc + o_a);
  uint8_t* flags_a = r->enc + o_fa; uint8_t* flags_r = r->enc + o_fr;
  // the R half (decode of R, s*B) is independent of the A half (decode of A, hash, h*A): while the batch leaves most of
  // the chip idle it runs on the side stream
  const size_t fork_max = g.opt_ladder_y_only != 0 && pair_lim(g, g.opt_ladder_pair_max) > (size_t)64 * (size_t)g.cus ? pair_lim(g, g.opt_ladder_pair_max) : (size_t)64 * (size_t)g.cus;
  const bool fork = g.opt_verify_overlap && n <= fork_max && host_load(g) < 4;      // with several calls in flight the other calls fill the idle SIMDs; a side stream only adds queue traffic
  const bool coop = g.opt_mul_algo == 1 && n <= coop_lim(g, g.opt_coop_max) && n <= coop_lim(g, g.opt_coop_base_max) &&
                    (pubs_ext == nullptr ? 6 * n <= 5 * coop_lim(g, g.opt_coop_ladder_enc_max) : 8 * n <= 7 * coop_lim(g, g.opt_coop_ladder_max));      // (from bytes: 260 / 350 us at 1,024 signatures, 380 / 352 at 1,536)     // small batch: one item per wavefront
  hipStream_t side = st;
  if (fork) {
    rc = ensure_aux(g, r); if (rc) return rc;
    side = r->aux;
    HIPCK(hipEventRecord(r->ev_fork, st));                 // behind whatever the caller queued before this call
    HIPCK(hipStreamWaitEvent(side, r->ev_fork, 0));
  }
  if (coop && pubs_ext == nullptr && g.opt_ladder_y_only != 0 && 4 * n <= coop_lim(g, g.opt_coop_max) && n <= coop_lim(g, 4 * g.cus)) {      // four wavefronts per signature in flight: up to four per SIMD (1,536 signatures: 0.455 against 0.40 ms)
    // One item per wavefront, from key BYTES: as in the DKG-sized path below, the 252 squarings of A's decode leave the critical path.  k_verify_hash gives
    // h and the flags the bytes decide; k_mul_enc_coop walks the ladder on A's y with a second wavefront per signature decoding A beside it, and hands
    // h A on projective; R's decode and s B run on the side stream (k_verify_prep_coop in front of k_mul_coop cost 0.07 of a 0.29 ms call).
    { ProfScope ps(g, side, KID_VERIFY_PREP_R); LAUNCHCK(launch::verify_prep_r_coop(side, sigs, n, flags_r, r->proj, r->proj_items, 2 * n)); }
    { ProfScope ps(g, st, KID_VERIFY_PREP); LAUNCHCK(launch::verify_hash(st, pubs, sigs, msgs, off, n, flags_a, hbuf, sbuf)); }
    if (fork) {
      HIPCK(hipEventRecord(r->ev_fork, st));               // s*B reads sbuf
      HIPCK(hipStreamWaitEvent(side, r->ev_fork, 0));
    }
    { ProfScope ps(g, side, KID_MUL_BASE_COOP); LAUNCHCK(launch::mul_base_coop(side, sbuf, n, nullptr, nullptr, coop_table(g), r->proj, r->proj_items, n)); }
    if (fork) HIPCK(hipEventRecord(r->ev_join, side));
    { ProfScope ps(g, st, KID_MUL_COOP); LAUNCHCK(launch::mul_enc_coop(st, hbuf, pubs, n, nullptr, nullptr, nullptr, launch::DoneFlag{}, r->proj, r->proj_items, flags_a, 3)); }      // h < L < 2^253
    if (fork) HIPCK(hipStreamWaitEvent(st, r->ev_join, 0));
    ProfScope ps(g, st, KID_VERIFY_FINAL);
    LAUNCHCK(launch::verify_final(st, r->proj, r->proj_items, n, flags_a, flags_r, flavor, status, take_done_flag(g, st, n)));
    return KYB_OK;
  }
  // keys given as POINTS, DKG-sized batch: nothing to decode on the A side; R is decoded by further workgroups of the ladder's launch (k_mul_ladder_pair_r) and
  // the equation is tested on projective coordinates — no inversion at the end, a failing signature costs what a valid one does (as the path from key bytes above)
  const bool r_role = !coop && pubs_ext != nullptr && fork && g.opt_mul_algo == 1 && g.opt_ladder_y_only == 2 && n <= pair_lim(g, g.opt_ladder_pair_max);
  // large batches compare encodings (kernels_verify.hip, k_verify_final_enc): R is only decoded for signatures that fail
  const bool by_enc = !coop && !r_role && g.opt_verify_by_enc != 0;
  if (by_enc && fork && pubs_ext == nullptr && g.opt_mul_algo == 1 && g.opt_ladder_y_only != 0 && n <= pair_lim(g, g.opt_ladder_pair_max)) {
    // DKG-sized batch from key BYTES: the decode of A (252 dependent squarings) leaves the critical path.  k_verify_hash gives h and every flag the
    // bytes decide; the two-lane ladder starts on A's y at once; the side stream decodes A, then multiplies s*B; k_ladder_recover joins (round 4).
    rc = ensure_ws_part(g, r, n); if (rc) return rc;
    uint4* state = reinterpret_cast<uint4*>(r->part);
    if (g.opt_ladder_y_only == 2) {
      // two launches on the critical path: hash + ladder with both decodes as workgroups of their own, then recovery + equation (kernels_ladder.hip,
      // k_verify_ladder_y); the side stream gathers s and multiplies s*B beside the first one.  R arrives decoded, so no inversion at the end.
      uint8_t* a_ok = r->enc + o_ok;
      { ProfScope ps(g, side, KID_VERIFY_PREP_R); LAUNCHCK(launch::sig_scalars(side, sigs, n, sbuf)); }
      rc = launch_base(g, true, sbuf, n, nullptr, nullptr, r, n, side, nullptr, 0, 3 * n); if (rc) return rc;
      HIPCK(hipEventRecord(r->ev_join, side));
      { ProfScope ps(g, st, KID_MUL_LADDER_PAIR); LAUNCHCK(launch::verify_ladder_y(st, pubs, sigs, msgs, off, n, flags_a, flags_r, a_ok, hbuf, state, a_ext, r->proj, r->proj_items, ladder_lanes(g, n))); }
      HIPCK(hipStreamWaitEvent(st, r->ev_join, 0));
      ProfScope ps(g, st, KID_VERIFY_FINAL);
      LAUNCHCK(launch::verify_recover_final(st, hbuf, n, a_ext, state, r->proj, r->proj_items, flags_a, a_ok, flags_r, flavor, status, take_done_flag(g, st, n)));
      return KYB_OK;
    }
    { ProfScope ps(g, side, KID_DECODE); LAUNCHCK(launch::decode_or_identity(side, pubs, n, a_ext, flags_r)); }      // flags_r: free in this path, holds "A decodes"
    HIPCK(hipEventRecord(r->ev_mid, side));
    { ProfScope ps(g, st, KID_VERIFY_PREP); LAUNCHCK(launch::verify_hash(st, pubs, sigs, msgs, off, n, flags_a, hbuf, sbuf)); }
    HIPCK(hipEventRecord(r->ev_fork, st));                 // s*B reads sbuf
    HIPCK(hipStreamWaitEvent(side, r->ev_fork, 0));
    rc = launch_base(g, true, sbuf, n, nullptr, nullptr, r, n, side, nullptr, 0, 3 * n); if (rc) return rc;
    HIPCK(hipEventRecord(r->ev_join, side));
    { ProfScope ps(g, st, KID_MUL_LADDER_PAIR); LAUNCHCK(launch::mul_ladder_pair_y(st, hbuf, n, pubs, state, 3, ladder_lanes(g, n))); }      // h < L < 2^253
    HIPCK(hipStreamWaitEvent(st, r->ev_mid, 0));
    { ProfScope ps(g, st, KID_LADDER_RECOVER); LAUNCHCK(launch::ladder_recover(st, hbuf, n, a_ext, state, r->proj, r->proj_items, flags_a, flags_r)); }
    HIPCK(hipStreamWaitEvent(st, r->ev_join, 0));
    ProfScope ps(g, st, KID_VERIFY_FINAL);
    LAUNCHCK(launch::verify_tail_enc(st, r->proj, r->proj_items, n, sigs, flags_a, flavor, status, take_done_flag(g, st, n), finish_four(g, n)));
    return KYB_OK;
  }
  if (!by_enc && !r_role) {
    ProfScope ps(g, side, KID_VERIFY_PREP_R);
    if (coop) LAUNCHCK(launch::verify_prep_r_coop(side, sigs, n, flags_r, r->proj, r->proj_items, 2 * n));
    else LAUNCHCK(launch::verify_prep_r(side, sigs, n, flags_r, r->proj, r->proj_items, 2 * n));
  }
  {
    ProfScope ps(g, st, KID_VERIFY_PREP);
    // (keys as points need no square root: the one-lane-per-item kernel — hash and an on-curve test, 0.036 ms — also in the one-item-per-wavefront regime)
    // (up to four wavefronts per SIMD in flight: beyond that the slower cooperative kernel, which lets the side stream's work finish first, wins — 2,048 signatures 0.467 against 0.496 ms)
    if (pubs_ext != nullptr && (!coop || n <= coop_lim(g, 4 * g.cus))) LAUNCHCK(launch::verify_prep_pts(st, pubs, pubs_ext, sigs, msgs, off, n, flags_a, hbuf, sbuf, a_ext));
    else if (coop) LAUNCHCK(launch::verify_prep_coop(st, pubs, sigs, msgs, off, n, flags_a, hbuf, sbuf, a_ext));
    else LAUNCHCK(launch::verify_prep(st, pubs, sigs, msgs, off, n, flags_a, hbuf, sbuf, a_ext));
  }
  if (fork) {
    HIPCK(hipEventRecord(r->ev_fork, st));                 // s*B reads sbuf, which the A-half kernel has just been asked to write
    HIPCK(hipStreamWaitEvent(side, r->ev_fork, 0));
    if (coop) { ProfScope ps(g, side, KID_MUL_BASE_COOP); LAUNCHCK(launch::mul_base_coop(side, sbuf, n, nullptr, nullptr, coop_table(g), r->proj, r->proj_items, n)); }
    else { rc = launch_base(g, true, sbuf, n, nullptr, nullptr, r, n, side, nullptr, 0, 3 * n); if (rc) return rc; }
    HIPCK(hipEventRecord(r->ev_join, side));
  }
  if (coop) {
    ProfScope ps(g, st, KID_MUL_COOP);
    LAUNCHCK(launch::mul_coop(st, hbuf, a_ext, n, nullptr, nullptr, 3, r->proj, r->proj_items, 0));      // h < L < 2^253
  } else if (r_role) {
    ProfScope ps(g, st, KID_MUL_LADDER_PAIR);
    LAUNCHCK(launch::mul_ladder_pair_r(st, hbuf, n, a_ext, r->proj, r->proj_items, 3, sigs, flags_r, 2 * n, ladder_lanes(g, n)));      // h < L < 2^253
  }
#ifdef KYB_CROSSCHECK
  else if (g.opt_mul_algo != 1) {
    rc = ensure_ws(g, r); if (rc) return rc;
    const size_t nchunks = (n + KYB_BLOCK - 1) / KYB_BLOCK;
    const int grid = (int)(nchunks < (size_t)g.grid_mul ? nchunks : (size_t)g.grid_mul);
    ProfScope ps(g, st, KID_MUL);
    LAUNCHCK(launch::mul_window(g.opt_mul_select, false, true, grid, st, hbuf, nullptr, a_ext, n, nullptr, nullptr, nullptr, r->ws, r->proj, r->proj_items));
  }
#endif
  else {
    rc = launch_ladder_core(g, hbuf, nullptr, a_ext, n, nullptr, r, st, 0, 3); if (rc) return rc;      // h < L < 2^253
  }
  if (fork) HIPCK(hipStreamWaitEvent(st, r->ev_join, 0));
  else if (coop) { ProfScope ps(g, st, KID_MUL_BASE_COOP); LAUNCHCK(launch::mul_base_coop(st, sbuf, n, nullptr, nullptr, coop_table(g), r->proj, r->proj_items, n)); }
  else { rc = launch_base(g, true, sbuf, n, nullptr, nullptr, r, n, st, nullptr, 0, 3 * n); if (rc) return rc; }
  {
    ProfScope ps(g, st, KID_VERIFY_FINAL);
    if (by_enc) LAUNCHCK(launch::verify_tail_enc(st, r->proj, r->proj_items, n, sigs, flags_a, flavor, status, take_done_flag(g, st, n), finish_four(g, n)));
    else LAUNCHCK(launch::verify_final(st, r->proj, r->proj_items, n, flags_a, flags_r, flavor, status, take_done_flag(g, st, n)));
  }
  return KYB_OK;
}

// PubPoly::eval has three launch shapes (kernel times on an MI355X: profiles/r02/poly_eval_multi_probe.log, poly_eval_batch_segments.log,
// poly_eval_shares_probe.log):
//   one evaluation per wavefront (kernels_coop.hip), its chain optionally cut over several wavefronts   — few evaluations
//   one evaluation per lane (k_poly_eval)                                                               — >= 10^5 evaluations
//   one SEGMENT per lane, recombined by the variable-base ladder (k_poly_eval_part)                      — long polynomials between the two
// Returns the number of segments per evaluation for the third shape, or 1 when one of the other two is expected to be faster.
// Cost model in microseconds.  A Horner step with an nbits-bit multiplier is nbits doublings, one addition per set bit (all of them
// when the lanes of a wavefront disagree, none for the bits no lane has set) and the coefficient's addition: ~1.5 nbits + 1 point
// operations; the constants were measured with 10-bit indices (16 operations).  The lane-per-item kernels keep a lane's latency
// up to one wavefront per SIMD (65,536 lanes) and take one more "round" for every further 65,536.
// Wavefronts per evaluation in the one-evaluation-per-wavefront regime (k_poly_eval_seg).  A segment's wavefront pays one full-length multiplication
// (x^(s len) of its partial value: ~0.195 ms) on top of its share of the chain, whose steps cost ~0.45 us per unit of (1.5 nbits + 1) — so cutting
// pays from the chain length at which t steps cost more than that, and then as many segments as there are idle SIMDs for are best (two coefficients per
// segment at least; profiles/r04/poly_segments_probe.log: t = 43 at a 10-bit index 0.348 -> 0.245 ms, at a 6-bit index one wavefront stays the best).
int coop_poly_segments(const Ctx& g, size_t n, size_t t, int nbits) {
  int cs = g.opt_poly_segments;
  if (cs == 0) {
    if (nbits <= 1) return 1;                                          // x = 1: the chain is t additions, nothing to gain
    size_t m = t / 2;
    const size_t by_room = 2048 / (n ? n : 1);                         // at most two wavefronts per SIMD: both still run near single-wavefront speed
    if (m > by_room) m = by_room;
    if (m > 32) m = 32;
    if (m < 2) return 1;
    const double step = 0.42 * (1.5 * (double)nbits + 1.0);
    const double one = (double)t * step, cut = (double)((t + m - 1) / m) * step + 195.0;
    cs = cut + 10.0 < one ? (int)m : 1;
  }
  if (cs > 32) cs = 32;
  if (cs < 1) cs = 1;
  return cs;
}
int poly_batch_segments(const Ctx& g, size_t n, size_t t, int nbits) {
  const int forced = g.opt_poly_batch_segments;
  if (forced != 0) return forced;
  if (nbits <= 1) return 1;                                            // x = 1: the chain is t additions
  const double f = (1.5 * (double)nbits + 1.0) / 16.0;
  double best;
  if (n <= coop_lim(g, g.opt_coop_max)) {
    const int cs = coop_poly_segments(g, n, t, nbits);
    const double waves = (double)n * cs, crowd = waves > 1280.0 ? waves / 1280.0 : 1.0;      // (1,536 until the two-lane ladder moved the boundary: profiles/r03/poly_eval_pair_probe.log)
    best = ((double)((t + cs - 1) / cs) * 6.2 * f + (cs > 1 ? 160.0 : 0.0)) * crowd + 40.0;
  } else {
    best = (double)((n + 65535) / 65536) * (double)t * 31.0 * f + 100.0;
  }
  int segs = 1;
  const size_t cand[] = {65536 / n, (65536 + n - 1) / n, 32768 / n, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48, 64, 96, 128, 192, 256};
  for (size_t sgs : cand) {
    if (sgs < 2 || sgs > 256 || sgs > t / 4) continue;                // at least four coefficients per segment
    const size_t len = (t + sgs - 1) / sgs;
    const double rounds = (double)((n * sgs + 65535) / 65536);
    // Horner chain + 255-step ladder per round; images, sums, finish.  Up to ladder.pair_max_items products the two-lane ladder runs them
    // (0.40 ms, no image kernel in front) instead of k_mont_prep + k_mul_ladder (0.10 + 0.61 ms)
    const bool pair = n * sgs <= pair_lim(g, g.opt_ladder_pair_max);
    const double cost = (rounds * ((double)len * 35.0 * f + (pair ? 400.0 : 700.0)) + (pair ? 200.0 : 300.0)) * 1.05;
    if (cost < best) { best = cost; segs = (int)sgs; }
  }
  return segs;
}

// commits_enc != nullptr: the commitments as wire encodings (what a Deal carries), decoded here first; ok[i] per commitment,
// a failed decode counts as the neutral element
// (launch_mu held, slot in use)  keep_commits: leave the decoded commitments in r->part for the caller (the combined DKG call sums them next);
// last: nothing is queued behind this call's kernels (they may carry the completion flag)
int poly_eval_locked(Ctx& g, StreamRes* r, const int32_t* commits, size_t t, const uint32_t* idx, size_t n, uint32_t max_index, uint8_t* oenc, int32_t* oext,
                     hipStream_t st, size_t per_poly, const uint8_t* commits_enc, uint8_t* ok, const int32_t** decoded = nullptr, bool last = true) {
  if (commits_enc != nullptr) {
    const size_t np = t * (per_poly ? (n + per_poly - 1) / per_poly : 1);
    int rc = ensure_ws_part(g, r, np); if (rc) return rc;
    int32_t* dec = reinterpret_cast<int32_t*>(r->part);
    if (np <= coop_lim(g, g.opt_coop_decode_max)) {
      ProfScope ps(g, st, KID_DECODE_COOP);
      LAUNCHCK(launch::decode_coop(st, commits_enc, np, dec, ok, true));
    } else {
      ProfScope ps(g, st, KID_DECODE);
      LAUNCHCK(launch::decode_or_identity(st, commits_enc, np, dec, ok));
    }
    commits = dec;
    if (decoded != nullptr) *decoded = dec;
  }
  int nbits = 1;
  while (nbits < 32 && ((uint64_t)max_index + 1) >> nbits) ++nbits;      // bit length of max x = max_index + 1
  {
    const int bsegs = poly_batch_segments(g, n, t, nbits);
    if (bsegs >= 2 && (size_t)bsegs <= t && g.opt_mul_algo == 1) {
      const int len = (int)((t + (size_t)bsegs - 1) / (size_t)bsegs);
      const int segs = (int)((t + (size_t)len - 1) / (size_t)len);     // no empty segment
      const size_t N = n * (size_t)segs;
      if (segs >= 2 && N <= (size_t(1) << 26)) {
        int rc = ensure_enc(g, r, 192 * N + 256); if (rc) return rc;
        int32_t* part_ext = reinterpret_cast<int32_t*>(r->enc);
        uint8_t* part_sc = r->enc + 160 * N;
        {
          ProfScope ps(g, st, KID_POLY_EVAL);
          LAUNCHCK(launch::poly_eval_part(st, commits, (int)t, idx, n, nbits, per_poly, len, segs, part_ext, part_sc));
        }
        rc = launch_ladder_core(g, part_sc, nullptr, part_ext, N, nullptr, r, st, 0, 1); if (rc) return rc;      // |multiplier| < 4L < 2^255
        rc = launch_pair_sums(g, r, n, (size_t)segs, st); if (rc) return rc;
        return launch_finish(g, r, n, oenc, oext, st, (size_t)segs, last);
      }
    }
  }
  if (n <= coop_lim(g, g.opt_coop_max)) {
    // few evaluations: one per wavefront (kernels_coop.hip); a long polynomial at very few indices: several wavefronts per evaluation.
    // A segment costs its wavefront one 255-step multiplication (~26 Horner steps of a 10-bit index) on top of its share of the chain,
    // and the segments of all evaluations should find idle SIMDs (2,048 wavefronts).
    const int segs = coop_poly_segments(g, n, t, nbits);
    if (segs >= 2 && (size_t)segs <= t) {
      const int len = (int)((t + (size_t)segs - 1) / (size_t)segs);
      int rc = ensure_enc(g, r, 160 * n * (size_t)segs + 256); if (rc) return rc;
      ProfScope ps(g, st, KID_POLY_EVAL_COOP);
      LAUNCHCK(launch::poly_eval_seg(st, commits, (int)t, idx, n, per_poly, len, segs, reinterpret_cast<uint32_t*>(r->enc), oenc, oext, (last ? take_done_flag(g, st, n) : launch::DoneFlag{}),
                                     ext_projective(g)));
      return KYB_OK;
    }
    ProfScope ps(g, st, KID_POLY_EVAL_COOP);
    LAUNCHCK(launch::poly_eval_coop(st, commits, (int)t, idx, n, nbits, per_poly, oenc, oext, (last ? take_done_flag(g, st, n) : launch::DoneFlag{}), ext_projective(g)));
    return KYB_OK;
  }
  const bool split = use_split(g, n);
  if (split) { int rc = ensure_proj(g, r, n); if (rc) return rc; }
  {
    ProfScope ps(g, st, KID_POLY_EVAL);
    LAUNCHCK(launch::poly_eval(split, st, commits, (int)t, idx, n, nbits, per_poly, oenc, oext, r->proj, r->proj_items));
  }
  if (split) return launch_finish(g, r, n, oenc, oext, st, 1, last);
  return KYB_OK;
}


// commits_enc != nullptr: the commitments as wire encodings (what a Deal carries), decoded here first; ok[i] per commitment,
// a failed decode counts as the neutral element
int launch_poly_eval(Ctx& g, const int32_t* commits, size_t t, const uint32_t* idx, size_t n, uint32_t max_index, uint8_t* oenc, int32_t* oext, hipStream_t st,
                     size_t per_poly = 0, const uint8_t* commits_enc = nullptr, uint8_t* ok = nullptr) {
  if (n == 0) return KYB_OK;
  std::lock_guard<std::mutex> launch_lock(g.launch_mu);
  StreamRes* r = nullptr;
  { int rc = res_for(g, st, &r); if (rc) return rc; }
  SlotUse use(r, st);
  return poly_eval_locked(g, r, commits, t, idx, n, max_index, oenc, oext, st, per_poly, commits_enc, ok);
}

// The verifier's curve work of one DKG round on the deals as they arrive (vss/pedersen/vss.rs:904-909 + dkg.rs:905-953): the m dealers'
// t commitments each are decoded ONCE; dealer g's polynomial is evaluated at `index` (eval outputs, m of them) and coefficient j is summed
// over the dealers (the distributed public polynomial, t outputs).  idx_dev: m copies of `index` in device memory.
int launch_dkg_round(Ctx& g, const uint8_t* commits_enc, size_t t, size_t m, const uint32_t* idx_dev, uint32_t index, uint8_t* eval_enc, int32_t* eval_ext,
                     uint8_t* sum_enc, int32_t* sum_ext, uint8_t* ok, hipStream_t st) {
  if (m == 0) return KYB_OK;
  std::lock_guard<std::mutex> launch_lock(g.launch_mu);
  StreamRes* r = nullptr;
  { int rc = res_for(g, st, &r); if (rc) return rc; }
  SlotUse use(r, st);
  const int32_t* dec = nullptr;
  const bool want_sum = sum_enc != nullptr || sum_ext != nullptr;
  int rc = poly_eval_locked(g, r, nullptr, t, idx_dev, m, index, eval_enc, eval_ext, st, 1, commits_enc, ok, &dec, !want_sum);
  if (rc || !want_sum) return rc;
  // the decoded commitments are still in r->part (dealer-major): sum the columns — groups of m records after the transposition
  return sum_locked(g, r, dec, nullptr, nullptr, false, t, m, sum_enc, sum_ext, st, true);
}

}  // namespace

#include "c_abi.inc"
#include "engine_group.inc"
#include "defer.inc"
