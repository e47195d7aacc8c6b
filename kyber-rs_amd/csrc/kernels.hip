// MI355X (gfx950) batched Ed25519 engine: the one translation unit behind libkyber_ed25519_hip.so.
//
// One scalar(-point pair) per lane, 64 lanes per wavefront, field elements as ten 32-bit VGPRs,
// products on v_mad_u64_u32 (see fe25519.h).  The path is integer-VALU bound: algorithmic HBM
// traffic is 64..224 B per operation against ~2*10^5 multiply-adds, so there is no MFMA and no
// LDS tiling of operands; LDS holds only the shared base-point table of the fixed-base kernels.
//
// Layout of the sources (all included here; templates and __global__ definitions must be visible
// to the launches, so this stays one TU):
//   fe25519.h ge25519.h ge_scalarmult.h ge_ladder.h sc25519.h sha512.h schnorr.h verify.h
//                        arithmetic, shared with the host-compiled check build (tests/hostcheck)
//   device_tables.h      table policies (workspace / LDS selection), 16-byte load/store helpers,
//                        the projective staging buffer of the split finish
//   device_kernels.h     every __global__ kernel
//   host_copy_pool.h     threaded memcpy for pageable caller buffers (plain C++, host only)
//   kernels.hip (this)   engine context, per-stream scratch, launch sequences, host-pointer pipeline
//   c_abi.inc            the extern "C" entry points of include/kyber_ed25519.h
//
//   k_mul_ladder   Point::mul(s, Some(P))  ge.rs:508-568   Montgomery ladder + y-recovery (default)
//   k_mul          the same, windowed      ge.rs:508-568   per-lane table 1P..8P in an L2/MALL-resident
//                                                          workspace, [entry][quad][lane] (mul.algo=0)
//   k_mul_base64   Point::mul(s, None)     ge.rs:442-486   42x32+16 affine table = the whole LDS (163,200 B), batches
//   k_mul_base32   the same                ge.rs:442-486   52x16 affine table in LDS (106,496 B)   (mul_base.radix=32)
//   k_mul_base     the same                ge.rs:442-486   64x8 affine table in LDS (65,536 B), small n
//   k_finish       batched inversion + encode (ge.rs:112-122)
//   k_sign / k_sign_hash / k_eddsa_prep    schnorr_sig.rs:25-47, eddsa_sig.rs:120-152
//   k_verify_prep / k_verify_final         eddsa_sig.rs:159-212, schnorr_sig.rs:53-110
//   k_poly_eval / k_pair_sum               share/poly.rs:457-469, 566-634
//   k_add / k_equal / k_encode / k_decode  point.rs:179-241 / 35-51
//   k_base_table / k_base_table32 / k_base_table64   build the LDS table images on the GPU at init
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/kyber_ed25519.h"
#include "schnorr.h"
namespace kyb {
// one out-of-line copy of the decompression (255 S + 20 M): called twice per item by k_verify_prep
__device__ __noinline__ uint32_t ge_decode_outlined(ge_p3& h, const uint32_t w[8]) { return ge_decode(h, w); }
__host__ inline uint32_t ge_decode_outlined_host(ge_p3& h, const uint32_t w[8]) { return ge_decode(h, w); }
}
#if defined(__HIP_DEVICE_COMPILE__)
#define KYB_GE_DECODE ge_decode_outlined
#else
#define KYB_GE_DECODE ge_decode_outlined_host
#endif
#include "verify.h"
#include "ge_ladder.h"

using namespace kyb;


#include "device_tables.h"
#include "device_kernels.h"
#include "host_copy_pool.h"

// ------------------------------------------------------------------------------------------------
// host side: context, staging, C ABI
// ------------------------------------------------------------------------------------------------
namespace {

thread_local std::string g_err;

// optional per-launch timing (bench.py): HIP events recorded on the launch stream around each kernel
enum KernelId { KID_MUL, KID_MUL_BASE, KID_FINISH, KID_SIGN, KID_SIGN_HASH, KID_VERIFY_PREP, KID_VERIFY_FINAL, KID_POLY_EVAL, KID_MONT_PREP,
                KID_MUL_LADDER, KID_DECODE, KID_EDDSA_PREP, KID_PAIR_SUM, KID_VERIFY_PREP_R, KID_COUNT };
const char* const KERNEL_NAMES[KID_COUNT] = {"k_mul", "k_mul_base", "k_finish", "k_sign", "k_sign_hash", "k_verify_prep", "k_verify_final", "k_poly_eval",
                                             "k_mont_prep", "k_mul_ladder", "k_decode", "k_eddsa_prep", "k_pair_sum", "k_verify_prep_r"};
struct ProfRec { int id; hipEvent_t a, b; };
struct Prof {
  bool on = false;
  int cap = 0, used = 0;
  ProfRec* recs = nullptr;
} g_prof;
struct ProfScope {
  hipStream_t st; int slot;
  ProfScope(hipStream_t s, int id) : st(s), slot(-1) {
    if (g_prof.on && g_prof.used < g_prof.cap) { slot = g_prof.used++; g_prof.recs[slot].id = id; (void)hipEventRecord(g_prof.recs[slot].a, st); }
  }
  ~ProfScope() { if (slot >= 0) (void)hipEventRecord(g_prof.recs[slot].b, st); }
};

struct Ctx {
  bool ready = false;
  int device = -1;
  int cus = 0;
  char name[128] = {0};
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;   // second lane of the pipelined host-pointer path
  uint32_t* table = nullptr;      // KYB_BASE_TABLE_BYTES: radix-16 image (65,536 B), radix-32 image (106,496 B), radix-64 image (163,200 B)
  bool table_ready = false;
  // per-stream device scratch (two launches that overlap on different streams must not share it):
  //   ws    variable-base table workspace (fixed size)
  //   proj  projective staging of the split finish (grows with the largest batch seen)
  //   enc   encodings of R and A between the stages of the split signing path
  //   aux   internal side stream (+ fork/join events) on which a small verification batch runs s*B next to the ladder
  struct StreamRes { hipStream_t stream; uint4* ws; uint4* proj; size_t proj_items; uint8_t* enc; size_t enc_bytes;
                     hipStream_t aux; hipEvent_t ev_fork, ev_join; };
  StreamRes res[8] = {};
  int res_count = 0;
  size_t ws_bytes = 0;
  int grid_mul = 0;
  uint8_t* stage = nullptr;       // device staging for the host-pointer API
  size_t stage_bytes = 0;
  uint8_t* stage2 = nullptr;      // staging of the second pipeline lane
  size_t stage2_bytes = 0;
  uint8_t* pin[2] = {nullptr, nullptr};   // page-locked bounce buffers of the two lanes (pageable caller memory)
  size_t pin_bytes[2] = {0, 0};
  int opt_copy_threads = 0;       // host threads that move pageable batches through the bounce buffers (0 = auto)
  int opt_mul_select = 1;         // 0 cndmask, 1 and/or mask
  int opt_base_select = 1;        // 0 LDS broadcast scan, 1 bpermute
  int opt_base_block = 256;       // 256 (2 waves/SIMD) or 512 (4 waves/SIMD, 128 VGPRs)   [radix-16 kernel]
  int opt_base_radix = 64;        // 64 / 32: 43- / 52-window kernel for batches >= finish.min_items; 16: always the 64-window kernel
  int opt_base_block64 = 1024;    // radix-64 kernel, full batches: 1024 (4 waves/SIMD, <= 128 VGPRs) or 512 (2 waves/SIMD)
  int opt_base_small_chunks = 2;  // radix-64 kernel: 256-thread workgroups up to this many chunks per CU, 1024-thread beyond
  int opt_verify_overlap = 1;     // small verification batches: s*B on a side stream next to the ladder
  int opt_mul_algo = 1;           // 0 windowed table (ge.rs structure), 1 Montgomery ladder (table-free, 1.33x faster: profiles/r01/sweep_mul_algo.log)
  int opt_ladder_waves = 3;       // launch bound of k_mul_ladder: waves per SIMD the register allocator must allow
  int opt_finish = 1;             // 0 fused inversion per item, 1 split + batched inversion (n >= finish_min)
  int opt_finish_min = 1;         // batches below it: fused per-item inversion in the radix-16 kernels (slower at every size, tools/midsize_bench.py; kept as a cross-check)
  std::mutex mu;          // host-pointer API: staging buffer + engine stream
  std::mutex launch_mu;   // every launch_* entry: per-stream scratch bookkeeping (calls from any thread, any stream)
};
Ctx g;

int fail(int code, const char* what, hipError_t e = hipSuccess) {
  char buf[256];
  if (e != hipSuccess) snprintf(buf, sizeof(buf), "%s: %s", what, hipGetErrorString(e));
  else snprintf(buf, sizeof(buf), "%s", what);
  g_err = buf;
  return code;
}
#define HIPCK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(KYB_E_HIP, #x, e_); } while (0)

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline hipStream_t pick(void* s) { return s ? reinterpret_cast<hipStream_t>(s) : g.stream; }

int ensure_stage(size_t bytes) {
  if (bytes <= g.stage_bytes) return KYB_OK;
  if (g.stage) { HIPCK(hipFree(g.stage)); g.stage = nullptr; g.stage_bytes = 0; }
  size_t want = bytes + (bytes >> 2) + 4096;
  hipError_t e = hipMalloc(&g.stage, want);
  if (e != hipSuccess) return fail(KYB_E_NOMEM, "staging allocation", e);
  g.stage_bytes = want;
  return KYB_OK;
}
inline size_t up256(size_t x) { return (x + 255) & ~(size_t)255; }
int ensure_stage2(size_t bytes) {
  if (bytes <= g.stage2_bytes) return KYB_OK;
  if (g.stage2) { HIPCK(hipFree(g.stage2)); g.stage2 = nullptr; g.stage2_bytes = 0; }
  size_t want = bytes + (bytes >> 2) + 4096;
  hipError_t e = hipMalloc(&g.stage2, want);
  if (e != hipSuccess) return fail(KYB_E_NOMEM, "staging allocation", e);
  g.stage2_bytes = want;
  return KYB_OK;
}

int ensure_pin(int lane, size_t bytes) {
  if (bytes <= g.pin_bytes[lane]) return KYB_OK;
  if (g.pin[lane]) { HIPCK(hipHostFree(g.pin[lane])); g.pin[lane] = nullptr; g.pin_bytes[lane] = 0; }
  size_t want = bytes + (bytes >> 2) + 4096;
  hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&g.pin[lane]), want, hipHostMallocDefault);
  if (e != hipSuccess) return fail(KYB_E_NOMEM, "pinned bounce buffer allocation", e);
  g.pin_bytes[lane] = want;
  return KYB_OK;
}

kyb::CopyPool g_copy;

int copy_threads() {
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

// Host-pointer batches of fixed-size records: the batch is cut into chunks that alternate between two
// streams (each with its own staging and scratch) so that the H2D copy of chunk c+1 and the D2H copy
// of chunk c-1 overlap the kernels of chunk c.  Page-locked caller buffers (kyb_host_alloc) are handed to
// the DMA engines as they are.  Pageable ones would make every hipMemcpyAsync a blocking, single-threaded
// staging copy inside the runtime (~7 GB/s); they go through the engine's own page-locked bounce buffers
// instead, filled and drained by CopyPool threads while the GPU works on the neighbouring chunk.
struct HostArr { const void* in; void* out; size_t bytes; };    // per-item size; exactly one of in/out, or neither = absent
constexpr size_t PIPE_MIN_ITEMS = (size_t)1 << 16;
constexpr int PIPE_CHUNKS = 8;
template <class Fn>
int run_host_batch(size_t n, const HostArr* arrs, int na, Fn launch) {
  std::lock_guard<std::mutex> lk(g.mu);
  HIPCK(hipSetDevice(g.device));
  const int nchunks = n >= PIPE_MIN_ITEMS ? PIPE_CHUNKS : 1;
  const size_t cap = (((n + nchunks - 1) / nchunks) + 1023) & ~(size_t)1023;      // items per chunk
  size_t off[8], total = 0;
  for (int k = 0; k < na; ++k) { off[k] = total; total += up256(arrs[k].bytes * cap); }
  int rc = ensure_stage(total);
  if (rc) return rc;
  if (nchunks > 1) { rc = ensure_stage2(total); if (rc) return rc; }
  hipStream_t streams[2] = {g.stream, g.stream2};
  uint8_t* stages[2] = {g.stage, g.stage2};
  bool pinned = true;
  for (int k = 0; k < na; ++k) {
    if (arrs[k].in) pinned = pinned && is_pinned(arrs[k].in);
    if (arrs[k].out) pinned = pinned && is_pinned(arrs[k].out);
  }
  if (nchunks > 1 && !pinned) {
    rc = ensure_pin(0, total); if (rc) return rc;
    rc = ensure_pin(1, total); if (rc) return rc;
    const int threads = copy_threads();
    kyb::CopyPool::Job jobs[8];
    auto chunk_items = [&](int c) { const size_t lo = (size_t)c * cap; return lo >= n ? (size_t)0 : ((lo + cap <= n) ? cap : n - lo); };
    auto copy_out = [&](int c) -> int {             // chunk c has been queued on its lane: wait for it, hand the results over
      const int lane = c & 1;
      HIPCK(hipStreamSynchronize(streams[lane]));
      const size_t lo = (size_t)c * cap, cn = chunk_items(c);
      int nj = 0;
      for (int k = 0; k < na; ++k)
        if (arrs[k].out) jobs[nj++] = kyb::CopyPool::Job{static_cast<uint8_t*>(arrs[k].out) + arrs[k].bytes * lo, g.pin[lane] + off[k], arrs[k].bytes * cn};
      g_copy.run(jobs, nj, threads);
      return KYB_OK;
    };
    int queued = -1;
    for (int c = 0; c < nchunks && chunk_items(c) > 0; ++c) {
      const int lane = c & 1;
      const size_t lo = (size_t)c * cap, cn = chunk_items(c);
      if (c >= 2) { rc = copy_out(c - 2); if (rc) return rc; }     // frees this lane's bounce and staging buffers
      int nj = 0;
      for (int k = 0; k < na; ++k)
        if (arrs[k].in) jobs[nj++] = kyb::CopyPool::Job{g.pin[lane] + off[k], static_cast<const uint8_t*>(arrs[k].in) + arrs[k].bytes * lo, arrs[k].bytes * cn};
      g_copy.run(jobs, nj, threads);
      uint8_t* dptr[8];
      for (int k = 0; k < na; ++k) {
        dptr[k] = (arrs[k].in || arrs[k].out) ? stages[lane] + off[k] : nullptr;
        if (arrs[k].in) HIPCK(hipMemcpyAsync(dptr[k], g.pin[lane] + off[k], arrs[k].bytes * cn, hipMemcpyHostToDevice, streams[lane]));
      }
      rc = launch(streams[lane], cn, dptr);
      if (rc) return rc;
      for (int k = 0; k < na; ++k)
        if (arrs[k].out) HIPCK(hipMemcpyAsync(g.pin[lane] + off[k], stages[lane] + off[k], arrs[k].bytes * cn, hipMemcpyDeviceToHost, streams[lane]));
      queued = c;
    }
    if (queued >= 1) { rc = copy_out(queued - 1); if (rc) return rc; }
    if (queued >= 0) { rc = copy_out(queued); if (rc) return rc; }
    return KYB_OK;
  }
  auto d2h = [&](int c) -> int {
    const int lane = c & 1;
    const size_t lo = (size_t)c * cap, cn = (lo + cap <= n) ? cap : n - lo;
    for (int k = 0; k < na; ++k)
      if (arrs[k].out) HIPCK(hipMemcpyAsync(static_cast<uint8_t*>(arrs[k].out) + arrs[k].bytes * lo, stages[lane] + off[k], arrs[k].bytes * cn, hipMemcpyDeviceToHost, streams[lane]));
    return KYB_OK;
  };
  int last = -1;
  for (int c = 0; c < nchunks; ++c) {
    const size_t lo = (size_t)c * cap;
    if (lo >= n) break;
    const size_t cn = (lo + cap <= n) ? cap : n - lo;
    const int lane = c & 1;
    if (c >= 2) HIPCK(hipStreamSynchronize(streams[lane]));      // chunk c-2 has left this lane's staging
    uint8_t* dptr[8];
    for (int k = 0; k < na; ++k) {
      dptr[k] = (arrs[k].in || arrs[k].out) ? stages[lane] + off[k] : nullptr;
      if (arrs[k].in) HIPCK(hipMemcpyAsync(dptr[k], static_cast<const uint8_t*>(arrs[k].in) + arrs[k].bytes * lo, arrs[k].bytes * cn, hipMemcpyHostToDevice, streams[lane]));
    }
    rc = launch(streams[lane], cn, dptr);
    if (rc) return rc;
    if (c >= 1) { rc = d2h(c - 1); if (rc) return rc; }
    last = c;
  }
  if (last >= 0) { rc = d2h(last); if (rc) return rc; }
  HIPCK(hipStreamSynchronize(g.stream));
  if (nchunks > 1) HIPCK(hipStreamSynchronize(g.stream2));
  return KYB_OK;
}

static_assert(KYB_BASE_TABLE_BYTES == 4u * (KYB_BASE_TABLE_WORDS + KYB_BASE32_TABLE_WORDS + KYB_BASE64_TABLE_WORDS), "table image layout");

// One synchronous host-pointer call of the small (non-pipelined) kind: the caller's arrays are laid out in the
// engine's device staging buffer, inputs copied in, `body` queues the kernels on the engine stream, outputs copied
// back, stream synchronised.  An array whose host pointer is null takes no space and maps to a null device pointer.
class HostCall {
 public:
  int in(const void* p, size_t bytes, size_t pad = 0) { return add(p, nullptr, bytes, pad); }
  int out(void* p, size_t bytes) { return add(nullptr, p, bytes, 0); }
  int inout(const void* p_in, void* p_out, size_t bytes) { return add(p_in, p_out, bytes, 0); }      // one device array, filled from p_in and/or returned to p_out
  template <class T = uint8_t>
  T* dev(int slot) const { return a_[slot].present ? reinterpret_cast<T*>(g.stage + a_[slot].off) : nullptr; }
  template <class Body>
  int run(Body body) {
    std::lock_guard<std::mutex> lk(g.mu);
    HIPCK(hipSetDevice(g.device));
    int rc = ensure_stage(total_);
    if (rc) return rc;
    for (int i = 0; i < n_; ++i)
      if (a_[i].src && a_[i].bytes) HIPCK(hipMemcpyAsync(g.stage + a_[i].off, a_[i].src, a_[i].bytes, hipMemcpyHostToDevice, g.stream));
    rc = body(g.stream);
    if (rc) return rc;
    for (int i = 0; i < n_; ++i)
      if (a_[i].dst && a_[i].bytes) HIPCK(hipMemcpyAsync(a_[i].dst, g.stage + a_[i].off, a_[i].bytes, hipMemcpyDeviceToHost, g.stream));
    HIPCK(hipStreamSynchronize(g.stream));
    return KYB_OK;
  }

 private:
  struct Arr { const void* src; void* dst; size_t bytes, off; bool present; };
  int add(const void* src, void* dst, size_t bytes, size_t pad) {
    const bool present = src != nullptr || dst != nullptr;
    a_[n_] = Arr{src, dst, bytes, total_, present};
    if (present) total_ += up256(bytes + pad);
    return n_++;
  }
  Arr a_[12];
  int n_ = 0;
  size_t total_ = 0;
};
// message blobs: offsets must not decrease; returns the blob size through *mbytes
int check_messages(const uint8_t* msgs, const uint32_t* msg_off, size_t n, size_t* mbytes) {
  *mbytes = msg_off[n];
  if (*mbytes && !msgs) return fail(KYB_E_BAD_ARG, "null message buffer");
  for (size_t i = 0; i < n; ++i) if (msg_off[i + 1] < msg_off[i]) return fail(KYB_E_BAD_ARG, "msg_off must be non-decreasing");
  return KYB_OK;
}

int do_init(int device, bool build_table) {
  std::lock_guard<std::mutex> lk(g.mu);
  if (g.ready) {
    if (g.device != device) return fail(KYB_E_BAD_ARG, "already initialised on another device (one process per GPU)");
    return KYB_OK;
  }
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
  g.device = device;
  g.cus = prop.multiProcessorCount;
  snprintf(g.name, sizeof(g.name), "%s", prop.name);
  HIPCK(hipStreamCreateWithFlags(&g.stream, hipStreamNonBlocking));
  HIPCK(hipStreamCreateWithFlags(&g.stream2, hipStreamNonBlocking));
  HIPCK(hipMalloc(&g.table, KYB_BASE_TABLE_BYTES));
  // persistent grids: 2 blocks of 256 threads per CU = 2 waves per SIMD (needed to saturate
  // v_mad_u64_u32 issue, profiles/r01_valu_rates_mi355x.jsonl)
  g.grid_mul = g.cus * 2;
  g.ws_bytes = (size_t)g.grid_mul * (KYB_BLOCK / 64) * (8 * 10 * 64) * sizeof(uint4);
  g.res[0] = Ctx::StreamRes{g.stream, nullptr, nullptr, 0, nullptr, 0};   // scratch is allocated on first use
  g.res_count = 1;
  if (build_table) {
    hipLaunchKernelGGL(k_base_table, dim3(8), dim3(64), 0, g.stream, g.table);
    HIPCK(hipGetLastError());
    hipLaunchKernelGGL(k_base_table32, dim3(13), dim3(64), 0, g.stream, g.table + KYB_BASE_TABLE_WORDS);
    HIPCK(hipGetLastError());
    hipLaunchKernelGGL(k_base_table64, dim3(22), dim3(64), 0, g.stream, g.table + KYB_BASE_TABLE_WORDS + KYB_BASE32_TABLE_WORDS);
    HIPCK(hipGetLastError());
    HIPCK(hipStreamSynchronize(g.stream));
    g.table_ready = true;
  }
  g.ready = true;
  return KYB_OK;
}

#define REQUIRE_READY() do { if (!g.ready) return fail(KYB_E_NOT_INIT, "kyb_init has not succeeded in this process"); } while (0)
#define REQUIRE_TABLE() do { if (!g.table_ready) return fail(KYB_E_NOT_INIT, "base table not built or imported"); } while (0)

// scratch bound to a stream (allocated on first use; at most 8 streams)
int res_for(hipStream_t st, Ctx::StreamRes** out) {
  static std::mutex mu;
  std::lock_guard<std::mutex> lk(mu);
  for (int i = 0; i < g.res_count; ++i) if (g.res[i].stream == st) { *out = &g.res[i]; return KYB_OK; }
  if (g.res_count == 8) return fail(KYB_E_NOMEM, "kernels have been launched on more than 8 distinct streams");
  g.res[g.res_count] = Ctx::StreamRes{st, nullptr, nullptr, 0, nullptr, 0};
  *out = &g.res[g.res_count++];
  return KYB_OK;
}
// the windowed-table kernel's per-wave table slots (160 MiB): only allocated if that kernel is used
int ensure_ws(Ctx::StreamRes* r) {
  if (r->ws) return KYB_OK;
  hipError_t e = hipMalloc(&r->ws, g.ws_bytes);
  if (e != hipSuccess) return fail(KYB_E_NOMEM, "table workspace allocation", e);
  return KYB_OK;
}
// grow-only; growth synchronises the stream first because earlier launches may still use the old buffer
int ensure_proj(Ctx::StreamRes* r, size_t items) {
  if (items <= r->proj_items) return KYB_OK;
  HIPCK(hipStreamSynchronize(r->stream));
  if (r->proj) HIPCK(hipFree(r->proj));
  r->proj = nullptr; r->proj_items = 0;
  const size_t want = ((items + (items >> 3)) + 1023) & ~(size_t)1023;
  hipError_t e = hipMalloc(&r->proj, want * 8 * sizeof(uint4));
  if (e != hipSuccess) return fail(KYB_E_NOMEM, "projective staging allocation", e);
  r->proj_items = want;
  return KYB_OK;
}
int ensure_enc(Ctx::StreamRes* r, size_t bytes) {
  if (bytes <= r->enc_bytes) return KYB_OK;
  HIPCK(hipStreamSynchronize(r->stream));
  if (r->enc) HIPCK(hipFree(r->enc));
  r->enc = nullptr; r->enc_bytes = 0;
  const size_t want = bytes + (bytes >> 3) + 4096;
  hipError_t e = hipMalloc(&r->enc, want);
  if (e != hipSuccess) return fail(KYB_E_NOMEM, "encoding staging allocation", e);
  r->enc_bytes = want;
  return KYB_OK;
}
int ensure_aux(Ctx::StreamRes* r) {
  if (r->aux) return KYB_OK;
  HIPCK(hipStreamCreateWithFlags(&r->aux, hipStreamNonBlocking));
  HIPCK(hipEventCreateWithFlags(&r->ev_fork, hipEventDisableTiming));
  HIPCK(hipEventCreateWithFlags(&r->ev_join, hipEventDisableTiming));
  return KYB_OK;
}
inline bool use_split(size_t n) { return g.opt_finish == 1 && n >= (size_t)g.opt_finish_min; }

int launch_finish(Ctx::StreamRes* r, size_t n, uint8_t* oenc, int32_t* oext, hipStream_t st, size_t src_mul = 1) {
  const size_t M = (n + FINISH_K - 1) / FINISH_K;
  ProfScope ps(st, KID_FINISH);
  hipLaunchKernelGGL(k_finish, dim3((unsigned)((M + KYB_BLOCK - 1) / KYB_BLOCK)), dim3(KYB_BLOCK), 0, st, r->proj, r->proj_items, n, oenc, oext, src_mul);
  HIPCK(hipGetLastError());
  return KYB_OK;
}

template <bool SPLIT>
void launch_mul_t(int sel, bool enc, int grid, hipStream_t st, const uint8_t* sc, const uint8_t* penc, const int32_t* pext, size_t n,
                  uint8_t* oenc, int32_t* oext, uint8_t* ok, Ctx::StreamRes* r) {
#define KYB_L(M_, E_) hipLaunchKernelGGL((k_mul<M_, E_, SPLIT>), dim3(grid), dim3(KYB_BLOCK), 0, st, sc, penc, pext, n, oenc, oext, ok, r->ws, r->proj, r->proj_items)
  if (sel == 0) { if (enc) KYB_L(0, true); else KYB_L(0, false); }
  else          { if (enc) KYB_L(1, true); else KYB_L(1, false); }
#undef KYB_L
}
// leaves the results projective in r->proj[0, n): prep (batched inversion) -> 256-step ladder.
// npts == 0: item i multiplies point i.  npts > 0: the npts points are shared, item i multiplies point
// i mod npts (their Montgomery images live in records [n, n + npts)).
// skip_bits: leading zero bits every scalar of the launch has for public reasons (3 for values reduced mod L).
int launch_ladder_core(const uint8_t* sc, const uint8_t* penc, const int32_t* pext, size_t n, uint8_t* ok, Ctx::StreamRes* r, hipStream_t st,
                       size_t npts = 0, int skip_bits = 0) {
  const size_t np = npts ? npts : n;
  int rc = ensure_proj(r, n + npts); if (rc) return rc;
  const unsigned blocks = (unsigned)((n + KYB_BLOCK - 1) / KYB_BLOCK);
  if (penc != nullptr) {           // unmarshal_binary of the operands first (ok flags; failed decodes become the neutral element)
    rc = ensure_enc(r, 160 * np + 256); if (rc) return rc;
    int32_t* tmp = reinterpret_cast<int32_t*>(r->enc);
    ProfScope ps(st, KID_DECODE);
    hipLaunchKernelGGL(k_decode_or_identity, dim3((unsigned)((np + KYB_BLOCK - 1) / KYB_BLOCK)), dim3(KYB_BLOCK), 0, st, penc, np, tmp, ok);
    pext = tmp;
  } else if (ok != nullptr) {
    HIPCK(hipMemsetAsync(ok, 1, np, st));          // extended operands are taken as they are (k_mul does the same)
  }
  HIPCK(hipGetLastError());
  {
    const size_t M = (np + FINISH_K - 1) / FINISH_K;
    ProfScope ps(st, KID_MONT_PREP);
    hipLaunchKernelGGL(k_mont_prep, dim3((unsigned)((M + KYB_BLOCK - 1) / KYB_BLOCK)), dim3(KYB_BLOCK), 0, st, pext, np, r->proj + (npts ? n : 0), r->proj_items);
  }
  HIPCK(hipGetLastError());
  {
    ProfScope ps(st, KID_MUL_LADDER);
    if (g.opt_ladder_waves >= 4)      hipLaunchKernelGGL((k_mul_ladder<4>), dim3(blocks), dim3(KYB_BLOCK), 0, st, sc, n, r->proj, r->proj_items, n, npts, skip_bits);
    else if (g.opt_ladder_waves == 3) hipLaunchKernelGGL((k_mul_ladder<3>), dim3(blocks), dim3(KYB_BLOCK), 0, st, sc, n, r->proj, r->proj_items, n, npts, skip_bits);
    else                              hipLaunchKernelGGL((k_mul_ladder<2>), dim3(blocks), dim3(KYB_BLOCK), 0, st, sc, n, r->proj, r->proj_items, n, npts, skip_bits);
  }
  HIPCK(hipGetLastError());
  return KYB_OK;
}

// out[g] = sum_j scalars[g*t + j] * P[g*t + j]  (shared == false)  or  * P[j]  (shared == true)
int launch_lincomb(const uint8_t* sc, const uint8_t* penc, const int32_t* pext, bool shared, size_t m, size_t t, uint8_t* ok,
                   uint8_t* oenc, int32_t* oext, hipStream_t st) {
  if (m == 0) return KYB_OK;
  std::lock_guard<std::mutex> launch_lock(g.launch_mu);
  Ctx::StreamRes* r = nullptr;
  { int rc = res_for(st, &r); if (rc) return rc; }
  const size_t n = m * t;
  { int rc = launch_ladder_core(sc, penc, pext, n, ok, r, st, shared ? t : 0); if (rc) return rc; }
  for (size_t len = t; len > 1;) {
    const size_t half = (len + 1) / 2, lanes = m * (len - half);
    ProfScope ps(st, KID_PAIR_SUM);
    hipLaunchKernelGGL(k_pair_sum, dim3((unsigned)((lanes + KYB_BLOCK - 1) / KYB_BLOCK)), dim3(KYB_BLOCK), 0, st, r->proj, r->proj_items, m, t, len, half);
    HIPCK(hipGetLastError());
    len = half;
  }
  return launch_finish(r, m, oenc, oext, st, t);
}

// out[g] = sum_j P[g*t + j]: the halving passes of launch_lincomb on the points themselves
int launch_sum(const int32_t* pext, size_t m, size_t t, uint8_t* oenc, int32_t* oext, hipStream_t st) {
  if (m == 0) return KYB_OK;
  std::lock_guard<std::mutex> launch_lock(g.launch_mu);
  Ctx::StreamRes* r = nullptr;
  { int rc = res_for(st, &r); if (rc) return rc; }
  const size_t n = m * t;
  { int rc = ensure_proj(r, n); if (rc) return rc; }
  hipLaunchKernelGGL(k_ext_to_proj, dim3((unsigned)((n + KYB_BLOCK - 1) / KYB_BLOCK)), dim3(KYB_BLOCK), 0, st, pext, n, r->proj, r->proj_items);
  HIPCK(hipGetLastError());
  for (size_t len = t; len > 1;) {
    const size_t half = (len + 1) / 2, lanes = m * (len - half);
    ProfScope ps(st, KID_PAIR_SUM);
    hipLaunchKernelGGL(k_pair_sum, dim3((unsigned)((lanes + KYB_BLOCK - 1) / KYB_BLOCK)), dim3(KYB_BLOCK), 0, st, r->proj, r->proj_items, m, t, len, half);
    HIPCK(hipGetLastError());
    len = half;
  }
  return launch_finish(r, m, oenc, oext, st, t);
}

int launch_mul(const uint8_t* sc, const uint8_t* penc, const int32_t* pext, size_t n, uint8_t* oenc, int32_t* oext, uint8_t* ok, hipStream_t st) {
  if (n == 0) return KYB_OK;
  std::lock_guard<std::mutex> launch_lock(g.launch_mu);
  Ctx::StreamRes* r = nullptr;
  { int rc = res_for(st, &r); if (rc) return rc; }
  if (g.opt_mul_algo == 1) {
    int rc = launch_ladder_core(sc, penc, pext, n, ok, r, st); if (rc) return rc;
    return launch_finish(r, n, oenc, oext, st);
  }
  { int rc = ensure_ws(r); if (rc) return rc; }
  const size_t nchunks = (n + KYB_BLOCK - 1) / KYB_BLOCK;
  const int grid = (int)(nchunks < (size_t)g.grid_mul ? nchunks : (size_t)g.grid_mul);
  const bool split = use_split(n);
  if (split) { int rc = ensure_proj(r, n); if (rc) return rc; }
  {
    ProfScope ps(st, KID_MUL);
    if (split) launch_mul_t<true>(g.opt_mul_select, penc != nullptr, grid, st, sc, penc, pext, n, oenc, oext, ok, r);
    else       launch_mul_t<false>(g.opt_mul_select, penc != nullptr, grid, st, sc, penc, pext, n, oenc, oext, ok, r);
  }
  HIPCK(hipGetLastError());
  if (split) return launch_finish(r, n, oenc, oext, st);
  return KYB_OK;
}

// fixed-base multiplication of n scalars; SPLIT leaves the points in r->proj at [offset, offset + n)
// sc_b != nullptr: a second array of n_b scalars follows the first in the same launch (radix-64 kernel), their
// results land behind the first n
template <bool SPLIT>
int launch_base_t(const uint8_t* sc, size_t n, uint8_t* oenc, int32_t* oext, Ctx::StreamRes* r, size_t offset, hipStream_t st,
                  const uint8_t* sc_b = nullptr, size_t n_b = 0) {
  if (sc_b != nullptr && !(g.opt_base_radix == 64 && n + n_b >= (size_t)g.opt_finish_min)) {
    int rc = launch_base_t<SPLIT>(sc, n, oenc, oext, r, offset, st); if (rc) return rc;
    return launch_base_t<SPLIT>(sc_b, n_b, oenc ? oenc + 32 * n : nullptr, oext ? oext + 40 * n : nullptr, r, offset + n, st);
  }
  if (g.opt_base_radix == 64 && n + n_b >= (size_t)g.opt_finish_min) {
    const size_t n_a = n;
    n += n_b;
    // one workgroup per CU (the table is its whole LDS); 256-thread workgroups while that leaves CUs idle
    const uint4* img64 = reinterpret_cast<const uint4*>(g.table + KYB_BASE_TABLE_WORDS + KYB_BASE32_TABLE_WORDS);
    const bool small = n <= (size_t)256 * (size_t)g.cus * (size_t)g.opt_base_small_chunks;
    const size_t block = small ? 256 : (size_t)g.opt_base_block64;
    const size_t nchunks64 = (n + block - 1) / block;
    const int grid64 = (int)(nchunks64 < (size_t)g.cus ? nchunks64 : (size_t)g.cus);
    ProfScope ps(st, KID_MUL_BASE);
    if (small) hipLaunchKernelGGL((k_mul_base64<SPLIT, 256>), dim3(grid64), dim3(256), 0, st, sc, sc_b, n_a, n, oenc, oext, img64, r->proj, r->proj_items, offset);
    else if (g.opt_base_block64 == 512) hipLaunchKernelGGL((k_mul_base64<SPLIT, 512>), dim3(grid64), dim3(512), 0, st, sc, sc_b, n_a, n, oenc, oext, img64, r->proj, r->proj_items, offset);
    else       hipLaunchKernelGGL((k_mul_base64<SPLIT, 1024>), dim3(grid64), dim3(1024), 0, st, sc, sc_b, n_a, n, oenc, oext, img64, r->proj, r->proj_items, offset);
    HIPCK(hipGetLastError());
    return KYB_OK;
  }
  if (g.opt_base_radix == 32 && n >= (size_t)g.opt_finish_min) {
    const uint4* img32 = reinterpret_cast<const uint4*>(g.table + KYB_BASE_TABLE_WORDS);
    const size_t nchunks32 = (n + KYB_BLOCK32 - 1) / KYB_BLOCK32;
    const int grid32 = (int)(nchunks32 < (size_t)g.cus ? nchunks32 : (size_t)g.cus);     // one workgroup per CU: the table fills its LDS
    ProfScope ps(st, KID_MUL_BASE);
    hipLaunchKernelGGL((k_mul_base32<SPLIT>), dim3(grid32), dim3(KYB_BLOCK32), 0, st, sc, n, oenc, oext, img32, r->proj, r->proj_items, offset);
    HIPCK(hipGetLastError());
    return KYB_OK;
  }
  const uint4* img = reinterpret_cast<const uint4*>(g.table);
  const int block = g.opt_base_block;
  const size_t nchunks = (n + block - 1) / block;
  const size_t cap = (size_t)g.cus * 2;                       // 2 blocks per CU: LDS holds two 64 KiB tables
  const int grid = (int)(nchunks < cap ? nchunks : cap);
  ProfScope ps(st, KID_MUL_BASE);
#define KYB_L(M_, B_) hipLaunchKernelGGL((k_mul_base<M_, B_, SPLIT>), dim3(grid), dim3(B_), 0, st, sc, n, oenc, oext, img, r->proj, r->proj_items, offset)
  if (g.opt_base_select == 0) { if (block == 512) KYB_L(0, 512); else KYB_L(0, 256); }
  else                        { if (block == 512) KYB_L(1, 512); else KYB_L(1, 256); }
#undef KYB_L
  HIPCK(hipGetLastError());
  return KYB_OK;
}
int launch_mul_base(const uint8_t* sc, size_t n, uint8_t* oenc, int32_t* oext, hipStream_t st) {
  if (n == 0) return KYB_OK;
  std::lock_guard<std::mutex> launch_lock(g.launch_mu);
  Ctx::StreamRes* r = nullptr;
  { int rc = res_for(st, &r); if (rc) return rc; }
  if (use_split(n)) {
    int rc = ensure_proj(r, n); if (rc) return rc;
    rc = launch_base_t<true>(sc, n, nullptr, nullptr, r, 0, st); if (rc) return rc;
    return launch_finish(r, n, oenc, oext, st);
  }
  return launch_base_t<false>(sc, n, oenc, oext, r, 0, st);
}
// schnorr::sign for n (x, k, msg) triples.  pub_in != nullptr: the callers' stored public keys enc(x*B) are
// hashed as they are and A is not recomputed (EdDSA::sign, eddsa_sig.rs:132-137; DSS long-term keys);
// pub_out != nullptr: receives enc(x*B).
int sign_locked(Ctx::StreamRes* r, const uint8_t* x, const uint8_t* k, const uint8_t* pub_in, const uint8_t* msgs, const uint32_t* off, size_t n,
                uint8_t* sig, uint8_t* pub_out, hipStream_t st) {
  const unsigned blocks = (unsigned)((n + KYB_BLOCK - 1) / KYB_BLOCK);
  if (pub_in != nullptr) {
    // R = k*B only
    int rc = ensure_enc(r, 32 * n); if (rc) return rc;
    if (use_split(n)) {
      rc = ensure_proj(r, n); if (rc) return rc;
      rc = launch_base_t<true>(k, n, nullptr, nullptr, r, 0, st); if (rc) return rc;
      rc = launch_finish(r, n, r->enc, nullptr, st); if (rc) return rc;
    } else {
      rc = launch_base_t<false>(k, n, r->enc, nullptr, r, 0, st); if (rc) return rc;
    }
    {
      ProfScope ps(st, KID_SIGN_HASH);
      hipLaunchKernelGGL(k_sign_hash, dim3(blocks), dim3(KYB_BLOCK), 0, st, x, k, msgs, off, n, r->enc, pub_in, sig);
      HIPCK(hipGetLastError());
    }
    if (pub_out != nullptr && pub_out != pub_in) HIPCK(hipMemcpyAsync(pub_out, pub_in, 32 * n, hipMemcpyDeviceToDevice, st));
    return KYB_OK;
  }
  if (use_split(2 * n)) {
    // R = k*B -> proj[0, n), A = x*B -> proj[n, 2n); one batched finish; then hash + scalar arithmetic
    int rc = ensure_proj(r, 2 * n); if (rc) return rc;
    rc = ensure_enc(r, 64 * n); if (rc) return rc;
    rc = launch_base_t<true>(k, n, nullptr, nullptr, r, 0, st, x, n); if (rc) return rc;
    rc = launch_finish(r, 2 * n, r->enc, nullptr, st); if (rc) return rc;
    {
      ProfScope ps(st, KID_SIGN_HASH);
      hipLaunchKernelGGL(k_sign_hash, dim3(blocks), dim3(KYB_BLOCK), 0, st, x, k, msgs, off, n, r->enc, r->enc + 32 * n, sig);
      HIPCK(hipGetLastError());
    }
    if (pub_out != nullptr) HIPCK(hipMemcpyAsync(pub_out, r->enc + 32 * n, 32 * n, hipMemcpyDeviceToDevice, st));
    return KYB_OK;
  }
  const size_t nchunks = (n + KYB_BLOCK - 1) / KYB_BLOCK;
  const size_t cap = (size_t)g.cus * 2;
  const int grid = (int)(nchunks < cap ? nchunks : cap);
  const uint4* img = reinterpret_cast<const uint4*>(g.table);
  {
    ProfScope ps(st, KID_SIGN);
    if (g.opt_base_select == 0) hipLaunchKernelGGL((k_sign<0, KYB_BLOCK>), dim3(grid), dim3(KYB_BLOCK), 0, st, x, k, msgs, off, n, sig, img);
    else                        hipLaunchKernelGGL((k_sign<1, KYB_BLOCK>), dim3(grid), dim3(KYB_BLOCK), 0, st, x, k, msgs, off, n, sig, img);
    HIPCK(hipGetLastError());
  }
  if (pub_out != nullptr) return launch_base_t<false>(x, n, pub_out, nullptr, r, 0, st);
  return KYB_OK;
}
int launch_sign(const uint8_t* x, const uint8_t* k, const uint8_t* pub_in, const uint8_t* msgs, const uint32_t* off, size_t n, uint8_t* sig, hipStream_t st) {
  if (n == 0) return KYB_OK;
  std::lock_guard<std::mutex> launch_lock(g.launch_mu);
  Ctx::StreamRes* r = nullptr;
  { int rc = res_for(st, &r); if (rc) return rc; }
  return sign_locked(r, x, k, pub_in, msgs, off, n, sig, nullptr, st);
}

// EdDSA::sign for n (seed, msg) pairs: expansion + nonce, then the Schnorr pipeline.  pub_in: the public keys
// the EdDSA objects hold (eddsa_sig.rs:22-29), or nullptr = derive them here; pub_out: optional copy of them.
int launch_eddsa_sign(const uint8_t* seeds, const uint8_t* pub_in, const uint8_t* msgs, const uint32_t* off, size_t n, uint8_t* sig, uint8_t* pub_out, hipStream_t st) {
  if (n == 0) return KYB_OK;
  std::lock_guard<std::mutex> launch_lock(g.launch_mu);
  Ctx::StreamRes* r = nullptr;
  int rc = res_for(st, &r); if (rc) return rc;
  // the signing pipeline uses r->enc[0, 64n) for the encodings of R and A: keep x and k behind that
  rc = ensure_enc(r, up256(64 * n) + 2 * up256(32 * n)); if (rc) return rc;
  uint8_t* xbuf = r->enc + up256(64 * n);
  uint8_t* kbuf = xbuf + up256(32 * n);
  {
    ProfScope ps(st, KID_EDDSA_PREP);
    hipLaunchKernelGGL(k_eddsa_prep, dim3((unsigned)((n + KYB_BLOCK - 1) / KYB_BLOCK)), dim3(KYB_BLOCK), 0, st, seeds, msgs, off, n, xbuf, kbuf);
    HIPCK(hipGetLastError());
  }
  return sign_locked(r, xbuf, kbuf, pub_in, msgs, off, n, sig, pub_out, st);
}

// verification pipeline on one stream: prep -> k_mul (h, A) -> k_mul_base (s) -> final
int launch_verify(const uint8_t* pubs, const uint8_t* msgs, const uint32_t* off, const uint8_t* sigs, size_t n, int flavor,
                  uint8_t* status, hipStream_t st) {
  if (n == 0) return KYB_OK;
  std::lock_guard<std::mutex> launch_lock(g.launch_mu);
  Ctx::StreamRes* r = nullptr;
  { int rc = res_for(st, &r); if (rc) return rc; }
  int rc = ensure_proj(r, 3 * n); if (rc) return rc;
  const size_t o_h = 0, o_s = up256(32 * n), o_a = o_s + up256(32 * n), o_fa = o_a + up256(160 * n), o_fr = o_fa + up256(n);
  rc = ensure_enc(r, o_fr + up256(n)); if (rc) return rc;
  uint8_t* hbuf = r->enc + o_h; uint8_t* sbuf = r->enc + o_s; int32_t* a_ext = reinterpret_cast<int32_t*>(r->enc + o_a);
  uint8_t* flags_a = r->enc + o_fa; uint8_t* flags_r = r->enc + o_fr;
  const unsigned blocks = (unsigned)((n + KYB_BLOCK - 1) / KYB_BLOCK);
  // the R half (decode of R, s*B) is independent of the A half (decode of A, hash, h*A): while the batch leaves most of
  // the chip idle it runs on the side stream
  const bool fork = g.opt_verify_overlap && n <= (size_t)64 * (size_t)g.cus;
  hipStream_t side = st;
  if (fork) {
    rc = ensure_aux(r); if (rc) return rc;
    side = r->aux;
    HIPCK(hipEventRecord(r->ev_fork, st));                 // behind whatever the caller queued before this call
    HIPCK(hipStreamWaitEvent(side, r->ev_fork, 0));
  }
  {
    ProfScope ps(side, KID_VERIFY_PREP_R);
    hipLaunchKernelGGL(k_verify_prep_r, dim3(blocks), dim3(KYB_BLOCK), 0, side, sigs, n, flags_r, r->proj, r->proj_items, 2 * n);
  }
  HIPCK(hipGetLastError());
  {
    ProfScope ps(st, KID_VERIFY_PREP);
    hipLaunchKernelGGL(k_verify_prep, dim3(blocks), dim3(KYB_BLOCK), 0, st, pubs, sigs, msgs, off, n, flags_a, hbuf, sbuf, a_ext);
  }
  HIPCK(hipGetLastError());
  if (fork) {
    HIPCK(hipEventRecord(r->ev_fork, st));                 // s*B reads sbuf, which the A-half kernel has just been asked to write
    HIPCK(hipStreamWaitEvent(side, r->ev_fork, 0));
    rc = launch_base_t<true>(sbuf, n, nullptr, nullptr, r, n, side); if (rc) return rc;
    HIPCK(hipEventRecord(r->ev_join, side));
  }
  if (g.opt_mul_algo == 1) {
    rc = launch_ladder_core(hbuf, nullptr, a_ext, n, nullptr, r, st, 0, 3); if (rc) return rc;      // h < L < 2^253
  } else {
    rc = ensure_ws(r); if (rc) return rc;
    const size_t nchunks = (n + KYB_BLOCK - 1) / KYB_BLOCK;
    const int grid = (int)(nchunks < (size_t)g.grid_mul ? nchunks : (size_t)g.grid_mul);
    ProfScope ps(st, KID_MUL);
    launch_mul_t<true>(g.opt_mul_select, false, grid, st, hbuf, nullptr, a_ext, n, nullptr, nullptr, nullptr, r);
    HIPCK(hipGetLastError());
  }
  if (fork) HIPCK(hipStreamWaitEvent(st, r->ev_join, 0));
  else { rc = launch_base_t<true>(sbuf, n, nullptr, nullptr, r, n, st); if (rc) return rc; }
  {
    ProfScope ps(st, KID_VERIFY_FINAL);
    hipLaunchKernelGGL(k_verify_final, dim3(blocks), dim3(KYB_BLOCK), 0, st, r->proj, r->proj_items, n, flags_a, flags_r, flavor, status);
  }
  HIPCK(hipGetLastError());
  return KYB_OK;
}

int launch_poly_eval(const int32_t* commits, size_t t, const uint32_t* idx, size_t n, uint32_t max_index, uint8_t* oenc, int32_t* oext, hipStream_t st,
                     size_t per_poly = 0) {
  if (n == 0) return KYB_OK;
  std::lock_guard<std::mutex> launch_lock(g.launch_mu);
  Ctx::StreamRes* r = nullptr;
  { int rc = res_for(st, &r); if (rc) return rc; }
  int nbits = 1;
  while (nbits < 32 && ((uint64_t)max_index + 1) >> nbits) ++nbits;      // bit length of max x = max_index + 1
  const unsigned blocks = (unsigned)((n + KYB_BLOCK - 1) / KYB_BLOCK);
  const bool split = use_split(n);
  if (split) { int rc = ensure_proj(r, n); if (rc) return rc; }
  {
    ProfScope ps(st, KID_POLY_EVAL);
    if (split) hipLaunchKernelGGL((k_poly_eval<true>), dim3(blocks), dim3(KYB_BLOCK), 0, st, commits, (int)t, idx, n, nbits, per_poly, oenc, oext, r->proj, r->proj_items);
    else       hipLaunchKernelGGL((k_poly_eval<false>), dim3(blocks), dim3(KYB_BLOCK), 0, st, commits, (int)t, idx, n, nbits, per_poly, oenc, oext, r->proj, r->proj_items);
  }
  HIPCK(hipGetLastError());
  if (split) return launch_finish(r, n, oenc, oext, st);
  return KYB_OK;
}

}  // namespace

#include "c_abi.inc"
