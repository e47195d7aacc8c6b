// Fixed-base kernels of the MI355X Ed25519 engine (one of the translation units mapped in launch.h).
//   k_base_table / 32 / 64   build the LDS table images on the GPU at init (role of constants.rs:89 BASE)
//   k_table_checksum         checksum embedded in / checked against an image that travelled between GPUs
//   k_mul_base64             Point::mul(s, None)  ge.rs:442-486   42x32+16 affine table = the whole LDS (163,200 B), batches
//   k_mul_base64_quarters    the same for mid-size batches: the four wavefronts of a workgroup take a quarter of an item's 43 windows each
#include <hip/hip_runtime.h>
#include "launch.h"
#include "ge_scalarmult.h"
#include "diag_stamp.h"
using namespace kyb;
#include "device_tables.h"
KYB_DEFINE_STAMP_SLOT()

__global__ void __launch_bounds__(64) k_base_table(uint32_t* image) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;   // 0..511
  if (e < 512) ge_base_table_entry(image, e >> 3, e & 7);
}

__global__ void __launch_bounds__(64) k_base_table32(uint32_t* image) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;   // 0..831
  if (e < KYB_BASE32_POS * 16) ge_base32_table_entry(image, e >> 4, e & 15);
}

__global__ void __launch_bounds__(64) k_base_table64(uint32_t* image) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;   // 0..1359: 42 windows x 32 entries, then the top window's 16
  if (e < 42 * 32) ge_base64_table_entry(image, e >> 5, e & 31);
  else if (e < 42 * 32 + 16) ge_base64_table_entry(image, 42, e - 42 * 32);
}

// entry-major copy for the cooperative kernels: one workgroup of 32 threads per entry (KYB_COOP_TABLE_WORDS, ge_scalarmult.h)
__global__ void __launch_bounds__(32) k_base_table_coop(const uint32_t* __restrict__ image64, uint32_t* __restrict__ tc) {
  const int e = blockIdx.x, w = threadIdx.x;                     // e: 0 .. 42 * 32 + 16 - 1
  const int pos = e < 42 * 32 ? (e >> 5) : 42, j = e < 42 * 32 ? (e & 31) : e - 42 * 32;
  tc[e * 32 + w] = w < 30 ? image64[KYB_BT64_IDX(pos, j, w)] : 0u;
}

// Fixed base, signed radix 64: one workgroup per CU owns the whole LDS (163,200 B table); 43 mixed additions
// per item.  BLOCK = 1024 / 768 / 512 (4 / 3 / 2 waves per SIMD; the register budget is 128 / 168 / 256 VGPRs) when the batch fills
// the chip, 256 (1 wave/SIMD, four times as many CUs busy) for batches that do not.
// Work is dealt out per WAVEFRONT in chunks of 64 items — chunk (round, wave, workgroup) = round * waves_in_grid + wave * gridDim.x +
// blockIdx.x — not per workgroup: a batch that is not a whole number of rounds leaves its last chunks with the LOW wave numbers of
// every workgroup, i.e. spread evenly over all CUs and over the four SIMDs of each (2^20 items on 256 CUs x 12 waves: 16 chunks per
// SIMD, as 6 + 5 + 5; dealt per workgroup the same batch would leave a third of the CUs with a round more than the others).
// Two scalar arrays may be multiplied in one launch (signing: the nonces and the private keys): items
// [0, n_a) come from `scalars`, items [n_a, n) from `scalars_b`.
template <bool SPLIT, int BLOCK>
__global__ void __launch_bounds__(BLOCK, (BLOCK + 255) / 256)
k_mul_base64(const uint8_t* __restrict__ scalars, const uint8_t* __restrict__ scalars_b, size_t n_a, size_t n,
             uint8_t* __restrict__ out_enc, int32_t* __restrict__ out_ext,
             const uint4* __restrict__ table_image, uint4* __restrict__ proj, size_t proj_stride, size_t proj_offset) {
  // kyb_diag_wave_stamps (diag_stamp.h): off unless a benchmark asked for the in-kernel clock
  KYB_STAMP_BEGIN();
  __shared__ uint4 lds_tbl[KYB_BASE64_TABLE_WORDS / 4];
  for (int k = threadIdx.x; k < KYB_BASE64_TABLE_WORDS / 4; k += BLOCK) lds_tbl[k] = table_image[k];
  __syncthreads();
  tbl_lds64 tbl{reinterpret_cast<const uint32_t*>(lds_tbl)};
  const size_t nchunks = (n + 63) / 64;
  const size_t per_round = (size_t)gridDim.x * (BLOCK / 64);
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave-uniform, and the compiler knows it
  // Addressing: everything the 43-window loop does not need stays out of VGPRs across it.  The chunk's first item i0 is
  // wave-uniform, so record addresses are (uniform base in SGPRs) + (32-bit lane offset), not 64-bit per-lane pointers
  // (which the compiler hoists out of the chunk loop as invariants and then has to spill in the 128-register build).
  for (size_t chunk = (size_t)wave * gridDim.x + blockIdx.x; chunk < nchunks; chunk += per_round) {
    const size_t i0 = chunk * 64;
    const uint32_t cnt = (uint32_t)((n - i0 < (size_t)64) ? n - i0 : (size_t)64);             // items of this chunk (>= 1)
    const uint32_t lane = threadIdx.x & 63u;
    const bool live = lane < cnt;
    const uint32_t t = live ? lane : 0u;                                                        // dead lanes redo the chunk's first item
    uint32_t a[8];
    if (scalars_b == nullptr || i0 + 64 <= n_a) load_words8(a, scalars + 32 * i0, t);          // the usual case: one array (uniform branch)
    else if (i0 >= n_a) load_words8(a, scalars_b + 32 * (i0 - n_a), t);
    else if (i0 + t < n_a) load_words8(a, scalars + 32 * i0, t);                               // the one chunk that straddles the two arrays
    else load_words8(a, scalars_b, (size_t)(i0 + t - n_a));
    ge_p3 h;
    ge_scalarmult_base64(h, a, tbl);
    uint32_t tl = threadIdx.x & 63u;
    asm volatile("" : "+v"(tl));         // the lane's store address is formed here, after the loop (not hoisted out of the chunk loop and spilled)
    if (SPLIT) { if (tl < cnt) store_proj(proj + (proj_offset + i0), proj_stride, tl, h.X, h.Y, h.Z); }
    else finish_point(h.X, h.Y, h.Z, out_enc ? out_enc + 32 * i0 : nullptr, out_ext ? out_ext + 40 * i0 : nullptr, t, live);
  }
  KYB_STAMP_END();
}

// The same for MID-SIZE batches (above the one-item-per-wavefront sizes, up to 128 items per CU): k_mul_base64 is then one lane's chain of 43
// additions with a wavefront per SIMD at most and most of the chip idle (105 us whatever the size).  The radix-64 sum has no doublings, so here
// the FOUR WAVEFRONTS of a 256-thread workgroup share 64 items: wavefront w adds windows [11 w, 11 w + 11) (the last one 33..42) for all of them,
// the partial points meet through the staging records (the table is the whole LDS) and two additions — wavefronts 0 and 1 side by side, then
// wavefront 0 — leave the sum in the item's own record, where k_finish* expects it.  Staging: part q of item i in record offset + q n + i
// (q = 1, 2, 3: the caller provides 4 n records), the result in record offset + i.  Same group element as k_mul_base64, hence the same bytes.
__device__ __forceinline__ void add_staged_part(ge_p3& h, const uint4* proj, size_t stride, size_t rec) {
  ge_p2 b;
  load_proj_xy(b.X, b.Y, proj, stride, rec); load_proj_z(b.Z, proj, stride, rec);
  ge_p3 B;
  ge_p2_to_p3(B, b);
  ge_cached c;
  ge_p3_to_cached(c, B);
  ge_p1p1 t;
  ge_add(t, h, c);
  ge_p1p1_to_p3(h, t);
}
__global__ void __launch_bounds__(256, 1)
k_mul_base64_quarters(const uint8_t* __restrict__ scalars, const uint8_t* __restrict__ scalars_b, size_t n_a, size_t n,
                      const uint4* __restrict__ table_image, uint4* __restrict__ proj, size_t proj_stride, size_t proj_offset, size_t parts_offset) {
  __shared__ uint4 lds_tbl[KYB_BASE64_TABLE_WORDS / 4];
  for (int k = threadIdx.x; k < KYB_BASE64_TABLE_WORDS / 4; k += 256) lds_tbl[k] = table_image[k];
  __syncthreads();
  tbl_lds64 tbl{reinterpret_cast<const uint32_t*>(lds_tbl)};
  const size_t ngroups = (n + 63) / 64;
  const int part = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave-uniform, and the compiler knows it
  const uint32_t lane = threadIdx.x & 63u;
  for (size_t grp = blockIdx.x; grp < ngroups; grp += gridDim.x) {          // (uniform per workgroup: every wavefront reaches every barrier)
    const size_t i0 = grp * 64;
    const uint32_t cnt = (uint32_t)((n - i0 < (size_t)64) ? n - i0 : (size_t)64);
    const bool live = lane < cnt;
    const uint32_t t = live ? lane : 0u;                                     // dead lanes redo the group's first item
    uint32_t a[8];
    if (scalars_b == nullptr || i0 + 64 <= n_a) load_words8(a, scalars + 32 * i0, t);
    else if (i0 >= n_a) load_words8(a, scalars_b + 32 * (i0 - n_a), t);
    else if (i0 + t < n_a) load_words8(a, scalars + 32 * i0, t);             // the one group that straddles the two arrays
    else load_words8(a, scalars_b, (size_t)(i0 + t - n_a));
    ge_p3 h;
    ge_scalarmult_base64_part(h, a, tbl, 11 * part, part == 3 ? KYB_BASE64_POS : 11 * part + 11);
    const size_t rec = proj_offset + i0 + t, prec = parts_offset + i0 + t;      // part q = 1, 2, 3 of the item: record prec + (q - 1) n
    if (part >= 2 && live) store_proj(proj, proj_stride, prec + (size_t)(part - 1) * n, h.X, h.Y, h.Z);
    __threadfence_block();
    __syncthreads();
    if (part < 2) add_staged_part(h, proj, proj_stride, prec + (size_t)(part + 1) * n);      // 0 + 2 | 1 + 3
    if (part == 1 && live) store_proj(proj, proj_stride, prec, h.X, h.Y, h.Z);
    __threadfence_block();
    __syncthreads();
    if (part == 0) {
      add_staged_part(h, proj, proj_stride, prec);
      if (live) store_proj(proj, proj_stride, rec, h.X, h.Y, h.Z);
    }
  }
}

// ---- table image checksum ---------------------------------------------------------------------------
// 64-bit position-weighted sum of the image's words, the two padding words of radix-16 entry (0, 0) excluded: that is
// where the checksum itself travels (KYB_BT_IDX(0, 0, 30 / 31); no kernel reads those words as data).  A wrong or
// truncated broadcast is caught at import instead of producing wrong points on 7 of 8 GPUs.
constexpr int KYB_TABLE_WORDS_ALL = KYB_BASE_TABLE_WORDS + KYB_BASE32_TABLE_WORDS + KYB_BASE64_TABLE_WORDS;
constexpr int KYB_CK_LO = KYB_BT_IDX(0, 0, 30), KYB_CK_HI = KYB_BT_IDX(0, 0, 31);
__global__ void __launch_bounds__(1024) k_table_checksum(const uint32_t* __restrict__ image, uint64_t* __restrict__ out, uint32_t* __restrict__ embed_into) {
  __shared__ unsigned long long part[1024 / 64];
  unsigned long long h = 0;
  for (int i = threadIdx.x; i < KYB_TABLE_WORDS_ALL; i += 1024) {
    if (i == KYB_CK_LO || i == KYB_CK_HI) continue;
    h += ((unsigned long long)image[i] + 0x9e3779b97f4a7c15ull) * (2ull * (unsigned long long)i + 1ull);
  }
  for (int o = 32; o > 0; o >>= 1) h += __shfl_down(h, o);
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = h;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long t = 0;
    for (int w = 0; w < 1024 / 64; ++w) t += part[w];
    if (out != nullptr) out[0] = t;
    if (embed_into != nullptr) { embed_into[KYB_CK_LO] = (uint32_t)t; embed_into[KYB_CK_HI] = (uint32_t)(t >> 32); }
  }
}

namespace kyb { namespace launch {
hipError_t build_tables(uint32_t* table, hipStream_t st) {
  hipLaunchKernelGGL(k_base_table, dim3(8), dim3(64), 0, st, table);
  hipLaunchKernelGGL(k_base_table32, dim3(13), dim3(64), 0, st, table + KYB_BASE_TABLE_WORDS);
  hipLaunchKernelGGL(k_base_table64, dim3(22), dim3(64), 0, st, table + KYB_BASE_TABLE_WORDS + KYB_BASE32_TABLE_WORDS);
  hipLaunchKernelGGL(k_table_checksum, dim3(1), dim3(1024), 0, st, table, (uint64_t*)nullptr, table);
  return hipGetLastError();
}
hipError_t build_coop_table(const uint32_t* image64, uint32_t* table_coop, hipStream_t st) {
  hipLaunchKernelGGL(k_base_table_coop, dim3(42 * 32 + 16), dim3(32), 0, st, image64, table_coop);
  return hipGetLastError();
}
hipError_t table_checksum(const uint32_t* table, uint64_t* out_dev, hipStream_t st) {
  hipLaunchKernelGGL(k_table_checksum, dim3(1), dim3(1024), 0, st, table, out_dev, (uint32_t*)nullptr);
  return hipGetLastError();
}
hipError_t mul_base64(bool split, int block, int grid, hipStream_t st, const uint8_t* sc, const uint8_t* sc_b, size_t n_a, size_t n,
                      uint8_t* oenc, int32_t* oext, const uint4* img64, uint4* proj, size_t stride, size_t offset) {
#define KYB_L(S_, B_) hipLaunchKernelGGL((k_mul_base64<S_, B_>), dim3(grid), dim3(B_), 0, st, sc, sc_b, n_a, n, oenc, oext, img64, proj, stride, offset)
#ifdef KYB_CROSSCHECK      // mul_base.block64 = 512 / 768 (2 / 3 wavefronts per SIMD): cross-check build only
  if (block == 512) { if (split) KYB_L(true, 512); else KYB_L(false, 512); return hipGetLastError(); }
  if (block == 768) { if (split) KYB_L(true, 768); else KYB_L(false, 768); return hipGetLastError(); }
#endif
  if (!split) {        // finish.batched = 0 (an inversion per item inside the kernel): cross-check build only
#ifdef KYB_CROSSCHECK
    if (block == 256) KYB_L(false, 256); else KYB_L(false, 1024);
    return hipGetLastError();
#else
    return hipErrorInvalidValue;
#endif
  }
  if (block == 256) KYB_L(true, 256); else KYB_L(true, 1024);
#undef KYB_L
  return hipGetLastError();
}
hipError_t mul_base64_quarters(int grid, hipStream_t st, const uint8_t* sc, const uint8_t* sc_b, size_t n_a, size_t n, const uint4* img64, uint4* proj, size_t stride,
                               size_t offset, size_t parts_offset) {
  hipLaunchKernelGGL(k_mul_base64_quarters, dim3(grid), dim3(256), 0, st, sc, sc_b, n_a, n, img64, proj, stride, offset, parts_offset);
  return hipGetLastError();
}
hipError_t diag_stamps_base(uint64_t* buf) { return kyb_set_stamp_slot(buf); }
}}  // namespace kyb::launch
