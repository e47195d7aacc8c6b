// MI355X (gfx950) kernels and C-ABI implementation of the batched Ed25519 engine.
//
// One scalar(-point pair) per lane, 64 lanes per wavefront, field elements as ten 32-bit VGPRs,
// products on v_mad_u64_u32 (see fe25519.h).  The path is integer-VALU bound: algorithmic HBM
// traffic is 64..224 B per operation against ~2*10^5 multiply-adds, so there is no MFMA and no
// LDS tiling of operands; LDS holds only the shared base-point table of the fixed-base kernels.
//
//   k_mul       Point::mul(s, Some(P))   ge.rs:508-568   per-lane table 1P..8P in an L2/MALL-resident
//                                                        workspace, [entry][quad][lane] so that every
//                                                        scan load is one coalesced 1 KiB request
//   k_mul_base  Point::mul(s, None)      ge.rs:442-486   64x8 affine table in LDS (65,536 B)
//   k_sign      schnorr::sign            schnorr_sig.rs:25-47
//   k_add / k_encode / k_decode          point.rs:179-197 / 35-51
//   k_base_table                         builds the LDS table image on the GPU at init
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>

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

// ------------------------------------------------------------------------------------------------
// device table policies
// ------------------------------------------------------------------------------------------------

// Variable-base table of one wave: uint4 [8 entries][10 quads][64 lanes] = 81,920 B.
// A cached point is 40 dwords: YpX[10] YmX[10] Z[10] T2d[10] -> 10 quads.
// MASKED = 0: merge with v_cndmask_b32; 1: merge with (x & m) | acc.
template <int MASKED>
struct tbl_global {
  uint4* p;  // wave base + lane
  struct scan { uint32_t f[40]; uint32_t mag; };
  struct slice { uint4 q[2][10]; };

  __device__ __forceinline__ static void flatten(uint32_t f[40], const ge_cached& c) {
#pragma unroll
    for (int i = 0; i < 10; ++i) { f[i] = c.YpX.v[i]; f[10 + i] = c.YmX.v[i]; f[20 + i] = c.Z.v[i]; f[30 + i] = c.T2d.v[i]; }
  }
  __device__ __forceinline__ static void unflatten(ge_cached& c, const uint32_t f[40]) {
#pragma unroll
    for (int i = 0; i < 10; ++i) { c.YpX.v[i] = f[i]; c.YmX.v[i] = f[10 + i]; c.Z.v[i] = f[20 + i]; c.T2d.v[i] = f[30 + i]; }
  }
  __device__ __forceinline__ void store(int e, const ge_cached& c) {
    uint32_t f[40];
    flatten(f, c);
#pragma unroll
    for (int q = 0; q < 10; ++q) p[(e * 10 + q) * 64] = make_uint4(f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]);
  }
  __device__ __forceinline__ static void merge_entry(uint32_t f[40], const uint4 q[10], uint32_t hit) {
    if (MASKED) {
      const uint32_t m = 0u - hit;
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        f[4 * i] |= q[i].x & m; f[4 * i + 1] |= q[i].y & m; f[4 * i + 2] |= q[i].z & m; f[4 * i + 3] |= q[i].w & m;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        f[4 * i] = hit ? q[i].x : f[4 * i]; f[4 * i + 1] = hit ? q[i].y : f[4 * i + 1];
        f[4 * i + 2] = hit ? q[i].z : f[4 * i + 2]; f[4 * i + 3] = hit ? q[i].w : f[4 * i + 3];
      }
    }
  }
  __device__ __forceinline__ void scan_begin(scan& st, uint32_t mag) {
    st.mag = mag;
#pragma unroll
    for (int i = 0; i < 40; ++i) st.f[i] = 0;
  }
  __device__ __forceinline__ void scan_issue(slice& sl, int k) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < 10; ++q) sl.q[h][q] = p[((2 * k + h) * 10 + q) * 64];
  }
  __device__ __forceinline__ void scan_merge(scan& st, slice& sl, int k) {
    merge_entry(st.f, sl.q[0], st.mag == (uint32_t)(2 * k + 1));
    merge_entry(st.f, sl.q[1], st.mag == (uint32_t)(2 * k + 2));
  }
  __device__ __forceinline__ void scan_end(ge_cached& c, scan& st) {
    const uint32_t z = (st.mag == 0);   // neutral element in cached form: (1, 1, 1, 0)
    st.f[0] |= z; st.f[10] |= z; st.f[20] |= z;
    unflatten(c, st.f);
  }
  __device__ __forceinline__ void select(ge_cached& c, uint32_t mag) {
    scan st;
    scan_begin(st, mag);
#pragma unroll 1
    for (int k = 0; k < 4; ++k) { slice sl; scan_issue(sl, k); scan_merge(st, sl, k); }
    scan_end(c, st);
  }
};

// Fixed-base table in LDS, image layout [pos][quad][entry][4] (KYB_BT_IDX).
// MODE 0: every lane reads all 8 entries (uniform address -> LDS broadcast) and merges under a mask.
// MODE 1: lane l fetches entry (l & 7) with eight conflict-free ds_read_b128, then each of the 30
//         limbs is pulled from the lane that holds the wanted entry with ds_bpermute_b32
//         (data-independent instruction stream and addresses; the only per-lane quantity is the
//         bpermute source lane, which goes through the conflict-free crossbar).
template <int MODE>
struct tbl_lds {
  const uint32_t* t;  // LDS
  __device__ __forceinline__ void select(ge_precomp& c, int pos, uint32_t mag) {
    uint32_t f[32];
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 32; ++i) f[i] = 0;
#pragma unroll 2
      for (int j = 0; j < 8; ++j) {
        const uint32_t m = 0u - (uint32_t)(mag == (uint32_t)(j + 1));
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const uint4 v = *reinterpret_cast<const uint4*>(t + ((pos * 8 + q) * 8 + j) * 4);
          f[4 * q] |= v.x & m; f[4 * q + 1] |= v.y & m; f[4 * q + 2] |= v.z & m; f[4 * q + 3] |= v.w & m;
        }
      }
    } else {
      const uint32_t lane = threadIdx.x & 63u;
      const uint32_t mine = lane & 7u;
      uint32_t own[32];
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const uint4 v = *reinterpret_cast<const uint4*>(t + ((pos * 8 + q) * 8 + mine) * 4);
        own[4 * q] = v.x; own[4 * q + 1] = v.y; own[4 * q + 2] = v.z; own[4 * q + 3] = v.w;
      }
      const uint32_t want = (mag - 1u) & 7u;                   // mag == 0 reads entry 7, masked below
      const int src = (int)(((lane & ~7u) | want) << 2);       // byte address of the source lane
      const uint32_t m = 0u - (uint32_t)(mag != 0);
#pragma unroll
      for (int i = 0; i < 30; ++i) f[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)own[i]) & m;
    }
    const uint32_t z = (mag == 0);       // neutral element in precomputed form: (1, 1, 0)
    f[0] |= z; f[10] |= z;
#pragma unroll
    for (int i = 0; i < 10; ++i) { c.ypx.v[i] = f[i]; c.ymx.v[i] = f[10 + i]; c.xy2d.v[i] = f[20 + i]; }
  }
};

// Radix-32 fixed-base table in LDS, image layout [pos][quad][16 entries][4] (KYB_BT32_IDX): lane l holds
// entry (l & 15) after eight conflict-free ds_read_b128, the wanted one is pulled with ds_bpermute_b32.
struct tbl_lds32 {
  const uint32_t* t;  // LDS
  __device__ __forceinline__ void select(ge_precomp& c, int pos, uint32_t mag) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t mine = lane & 15u;
    uint32_t own[32], f[30];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const uint4 v = *reinterpret_cast<const uint4*>(t + ((pos * 8 + q) * 16 + mine) * 4);
      own[4 * q] = v.x; own[4 * q + 1] = v.y; own[4 * q + 2] = v.z; own[4 * q + 3] = v.w;
    }
    const uint32_t want = (mag - 1u) & 15u;                  // mag == 0 reads entry 15, masked below
    const int src = (int)(((lane & ~15u) | want) << 2);
    const uint32_t m = 0u - (uint32_t)(mag != 0);
#pragma unroll
    for (int i = 0; i < 30; ++i) f[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)own[i]) & m;
    const uint32_t z = (mag == 0);
    f[0] |= z; f[10] |= z;
#pragma unroll
    for (int i = 0; i < 10; ++i) { c.ypx.v[i] = f[i]; c.ymx.v[i] = f[10 + i]; c.xy2d.v[i] = f[20 + i]; }
  }
};

// ------------------------------------------------------------------------------------------------
// load / store helpers (16-byte vector accesses; batches are arrays of 32- or 160-byte records)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_words8(uint32_t w[8], const uint8_t* base, size_t i) {
  const uint4* p = reinterpret_cast<const uint4*>(base) + 2 * i;
  const uint4 a = p[0], b = p[1];
  w[0] = a.x; w[1] = a.y; w[2] = a.z; w[3] = a.w; w[4] = b.x; w[5] = b.y; w[6] = b.z; w[7] = b.w;
}
__device__ __forceinline__ void store_words8(uint8_t* base, size_t i, const uint32_t w[8]) {
  uint4* p = reinterpret_cast<uint4*>(base) + 2 * i;
  p[0] = make_uint4(w[0], w[1], w[2], w[3]);
  p[1] = make_uint4(w[4], w[5], w[6], w[7]);
}
__device__ __forceinline__ void load_ext(ge_p3& P, const int32_t* base, size_t i) {
  const uint4* p = reinterpret_cast<const uint4*>(base) + 10 * i;
  int32_t s[40];
#pragma unroll
  for (int q = 0; q < 10; ++q) { const uint4 v = p[q]; s[4 * q] = (int32_t)v.x; s[4 * q + 1] = (int32_t)v.y; s[4 * q + 2] = (int32_t)v.z; s[4 * q + 3] = (int32_t)v.w; }
  fe_from_ref10(P.X, s); fe_from_ref10(P.Y, s + 10); fe_from_ref10(P.Z, s + 20); fe_from_ref10(P.T, s + 30);
}
__device__ __forceinline__ void store_ext(int32_t* base, size_t i, const fe& X, const fe& Y, const fe& Z, const fe& T) {
  int32_t s[40];
  fe_to_ref10(s, X); fe_to_ref10(s + 10, Y); fe_to_ref10(s + 20, Z); fe_to_ref10(s + 30, T);
  uint4* p = reinterpret_cast<uint4*>(base) + 10 * i;
#pragma unroll
  for (int q = 0; q < 10; ++q) p[q] = make_uint4((uint32_t)s[4 * q], (uint32_t)s[4 * q + 1], (uint32_t)s[4 * q + 2], (uint32_t)s[4 * q + 3]);
}
// encode (and optionally emit affine extended limbs, Z = 1) from a projective result
__device__ __forceinline__ void finish_point(const fe& X, const fe& Y, const fe& Z, uint8_t* out_enc, int32_t* out_ext, size_t i, bool live) {
  fe zi, x, y;
  fe_invert(zi, Z);
  fe_mul(x, X, zi);
  fe_mul(y, Y, zi);
  if (out_enc != nullptr) {
    uint32_t w[8];
    fe_to_words(w, y);
    w[7] ^= fe_is_negative(x) << 31;
    if (live) store_words8(out_enc, i, w);
  }
  if (out_ext != nullptr) {
    fe one, t;
    fe_one(one);
    fe_mul(t, x, y);
    if (live) store_ext(out_ext, i, x, y, one, t);
  }
}

// ------------------------------------------------------------------------------------------------
// projective staging buffer for the split finish: uint4 [8 quads][stride items]
//   dwords 0..9 X, 10..19 Y, 20..29 Z (tight limbs), 30..31 unused.  Item-minor so that both the
//   producer (lane = item) and the batched finish (lane j takes items j, j+M, j+2M, ...) are coalesced.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void store_proj(uint4* proj, size_t stride, size_t i, const fe& X, const fe& Y, const fe& Z) {
  uint32_t f[32];
#pragma unroll
  for (int k = 0; k < 10; ++k) { f[k] = X.v[k]; f[10 + k] = Y.v[k]; f[20 + k] = Z.v[k]; }
  f[30] = 0; f[31] = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) proj[q * stride + i] = make_uint4(f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]);
}
__device__ __forceinline__ void load_proj_z(fe& Z, const uint4* proj, size_t stride, size_t i) {
  const uint4 a = proj[5 * stride + i], b = proj[6 * stride + i], c = proj[7 * stride + i];
  Z.v[0] = a.x; Z.v[1] = a.y; Z.v[2] = a.z; Z.v[3] = a.w; Z.v[4] = b.x; Z.v[5] = b.y; Z.v[6] = b.z; Z.v[7] = b.w; Z.v[8] = c.x; Z.v[9] = c.y;
}
__device__ __forceinline__ void load_proj_xy(fe& X, fe& Y, const uint4* proj, size_t stride, size_t i) {
  uint32_t f[20];
#pragma unroll
  for (int q = 0; q < 5; ++q) { const uint4 v = proj[q * stride + i]; f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w; }
#pragma unroll
  for (int k = 0; k < 10; ++k) { X.v[k] = f[k]; Y.v[k] = f[10 + k]; }
}

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
constexpr int KYB_BLOCK = 256;

__global__ void __launch_bounds__(64) k_base_table(uint32_t* image) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;   // 0..511
  if (e < 512) ge_base_table_entry(image, e >> 3, e & 7);
}

__global__ void __launch_bounds__(64) k_base_table32(uint32_t* image) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;   // 0..831
  if (e < KYB_BASE32_POS * 16) ge_base32_table_entry(image, e >> 4, e & 15);
}

// Fixed base, signed radix 32: one 1024-thread workgroup per CU shares the 106,496-byte table in LDS
// (4 waves per SIMD, <= 128 VGPRs); 52 mixed additions per item.
constexpr int KYB_BLOCK32 = 1024;
template <bool SPLIT>
__global__ void __launch_bounds__(KYB_BLOCK32, 4)
k_mul_base32(const uint8_t* __restrict__ scalars, size_t n, uint8_t* __restrict__ out_enc, int32_t* __restrict__ out_ext,
             const uint4* __restrict__ table_image, uint4* __restrict__ proj, size_t proj_stride, size_t proj_offset) {
  __shared__ uint4 lds_tbl[KYB_BASE32_TABLE_WORDS / 4];
  for (int k = threadIdx.x; k < KYB_BASE32_TABLE_WORDS / 4; k += KYB_BLOCK32) lds_tbl[k] = table_image[k];
  __syncthreads();
  tbl_lds32 tbl{reinterpret_cast<const uint32_t*>(lds_tbl)};
  const size_t nchunks = (n + KYB_BLOCK32 - 1) / KYB_BLOCK32;
  for (size_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const size_t i = chunk * KYB_BLOCK32 + threadIdx.x;
    const bool live = i < n;
    const size_t ii = live ? i : 0;
    uint32_t a[8];
    load_words8(a, scalars, ii);
    ge_p3 h;
    ge_scalarmult_base32(h, a, tbl);
    if (SPLIT) { if (live) store_proj(proj, proj_stride, proj_offset + i, h.X, h.Y, h.Z); }
    else finish_point(h.X, h.Y, h.Z, out_enc, out_ext, ii, live);
  }
}

// Variable base.  Persistent grid: block b handles chunks b, b+grid, ...; its four waves own four
// table slots of the workspace for the whole launch.  SPLIT: leave the result projective in `proj`
// for k_finish (one field inversion per FINISH_K items instead of one per item).
template <int MASKED, bool FROM_ENC, bool SPLIT>
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_mul(const uint8_t* __restrict__ scalars, const uint8_t* __restrict__ pts_enc, const int32_t* __restrict__ pts_ext,
      size_t n, uint8_t* __restrict__ out_enc, int32_t* __restrict__ out_ext, uint8_t* __restrict__ ok_out, uint4* __restrict__ ws,
      uint4* __restrict__ proj, size_t proj_stride) {
  const uint32_t lane = threadIdx.x & 63u;
  const size_t wave_slot = (size_t)blockIdx.x * (KYB_BLOCK / 64) + (threadIdx.x >> 6);
  tbl_global<MASKED> tbl{ws + wave_slot * (8 * 10 * 64) + lane};
  const size_t nchunks = (n + KYB_BLOCK - 1) / KYB_BLOCK;
  for (size_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const size_t i = chunk * KYB_BLOCK + threadIdx.x;
    const bool live = i < n;
    const size_t ii = live ? i : 0;       // dead lanes redo item 0 and store nothing
    uint32_t a[8];
    load_words8(a, scalars, ii);
    ge_p3 P;
    uint32_t ok = 1;
    if (FROM_ENC) {
      uint32_t w[8];
      load_words8(w, pts_enc, ii);
      ok = ge_decode(P, w);
      ge_p3 id;
      ge_p3_0(id);
      fe_cmov(P.X, id.X, 1u - ok); fe_cmov(P.Y, id.Y, 1u - ok); fe_cmov(P.Z, id.Z, 1u - ok); fe_cmov(P.T, id.T, 1u - ok);
    } else {
      load_ext(P, pts_ext, ii);
    }
    ge_p2 r;
    ge_scalarmult(r, a, P, tbl);
    if (SPLIT) { if (live) store_proj(proj, proj_stride, i, r.X, r.Y, r.Z); }
    else finish_point(r.X, r.Y, r.Z, out_enc, out_ext, ii, live);
    if (ok_out != nullptr && live) ok_out[i] = (uint8_t)ok;
  }
}

// Fixed base.  BLOCK = 256 (2 waves/SIMD, <= 256 VGPRs) or 512 (one 64 KiB LDS table shared by 8 waves,
// 2 blocks per CU = 4 waves/SIMD, 128 VGPRs).
template <int MODE, int BLOCK, bool SPLIT>
__global__ void __launch_bounds__(BLOCK, BLOCK == 512 ? 4 : 2)
k_mul_base(const uint8_t* __restrict__ scalars, size_t n, uint8_t* __restrict__ out_enc, int32_t* __restrict__ out_ext,
           const uint4* __restrict__ table_image, uint4* __restrict__ proj, size_t proj_stride, size_t proj_offset) {
  __shared__ uint4 lds_tbl[KYB_BASE_TABLE_WORDS / 4];
  for (int k = threadIdx.x; k < KYB_BASE_TABLE_WORDS / 4; k += BLOCK) lds_tbl[k] = table_image[k];
  __syncthreads();
  tbl_lds<MODE> tbl{reinterpret_cast<const uint32_t*>(lds_tbl)};
  const size_t nchunks = (n + BLOCK - 1) / BLOCK;
  for (size_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const size_t i = chunk * BLOCK + threadIdx.x;
    const bool live = i < n;
    const size_t ii = live ? i : 0;
    uint32_t a[8];
    load_words8(a, scalars, ii);
    ge_p3 h;
    ge_scalarmult_base(h, a, tbl);
    if (SPLIT) { if (live) store_proj(proj, proj_stride, proj_offset + i, h.X, h.Y, h.Z); }
    else finish_point(h.X, h.Y, h.Z, out_enc, out_ext, ii, live);
  }
}

// Montgomery's trick over K values with the running prefixes held in locals of a template recursion (an
// indexed `fe pre[K]` array ends up in scratch): level T multiplies d_T onto the prefix, the innermost level
// inverts once, and on the way back every level peels its own 1/d_T off.
//   load(t, d)   supplies d_t (already forced non-zero)        emit(t, dinv)   consumes 1/d_t
template <int T, int K, class Load, class Emit>
__device__ __forceinline__ void batch_invert(const fe& prefix_prev, fe& inv_prev, Load& load, Emit& emit) {
  fe d, pre, inv, di;
  load(T, d);
  if (T == 0) fe_copy(pre, d); else fe_mul(pre, prefix_prev, d);
  if constexpr (T + 1 < K) batch_invert<T + 1, K>(pre, inv, load, emit); else fe_invert(inv, pre);
  if (T == 0) fe_copy(di, inv); else fe_mul(di, inv, prefix_prev);
  emit(T, di);
  if (T > 0) { load(T, d); fe_mul(inv_prev, inv, d); }
}

// Batched finish: lane j owns items j, j+M, ..., j+(K-1)M (M = ceil(n/K)) and inverts the product of
// their Z's once (Montgomery's trick): 3(K-1) M + one inversion per K items instead of 254 S + 11 M
// per item.  A zero Z (only reachable from invalid extended inputs) is replaced by 1 in the product
// and gets the reference's own answer for it (0^(p-2) = 0 -> x = y = 0), so one bad item cannot
// disturb its K-1 neighbours.  Item i is read from record i * src_mul (src_mul = group length after a
// segmented sum, 1 otherwise).
constexpr int FINISH_K = 8;
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_finish(const uint4* __restrict__ proj, size_t stride, size_t n, uint8_t* __restrict__ out_enc, int32_t* __restrict__ out_ext, size_t src_mul) {
  const size_t M = (n + FINISH_K - 1) / FINISH_K;
  const size_t j = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (j >= M) return;
  auto load = [&](int t, fe& z) {
    const size_t i = j + (size_t)t * M;
    fe one;
    fe_one(one);
    if (i < n) load_proj_z(z, proj, stride, i * src_mul); else fe_one(z);
    fe_cmov(z, one, 1u - fe_is_nonzero(z));
  };
  auto emit = [&](int t, const fe& zinv) {
    const size_t i = j + (size_t)t * M;
    if (i >= n) return;
    fe z, zi, zero, one, X, Y, x, y;
    fe_zero(zero); fe_one(one);
    load_proj_z(z, proj, stride, i * src_mul);
    fe_copy(zi, zinv);
    fe_cmov(zi, zero, 1u - fe_is_nonzero(z));          // Z == 0: the reference's 0^(p-2) = 0
    load_proj_xy(X, Y, proj, stride, i * src_mul);
    fe_mul(x, X, zi);
    fe_mul(y, Y, zi);
    if (out_enc != nullptr) {
      uint32_t w[8];
      fe_to_words(w, y);
      w[7] ^= fe_is_negative(x) << 31;
      store_words8(out_enc, i, w);
    }
    if (out_ext != nullptr) {
      fe tt;
      fe_mul(tt, x, y);
      store_ext(out_ext, i, x, y, one, tt);
    }
  };
  fe unused_prefix, unused_inv;
  fe_one(unused_prefix);
  batch_invert<0, FINISH_K>(unused_prefix, unused_inv, load, emit);
}

// fused signing kernel (small batches)
template <int MODE, int BLOCK>
__global__ void __launch_bounds__(BLOCK, 2)
k_sign(const uint8_t* __restrict__ x, const uint8_t* __restrict__ k, const uint8_t* __restrict__ msgs,
       const uint32_t* __restrict__ msg_off, size_t n, uint8_t* __restrict__ sig, const uint4* __restrict__ table_image) {
  __shared__ uint4 lds_tbl[KYB_BASE_TABLE_WORDS / 4];
  for (int q = threadIdx.x; q < KYB_BASE_TABLE_WORDS / 4; q += BLOCK) lds_tbl[q] = table_image[q];
  __syncthreads();
  tbl_lds<MODE> tbl{reinterpret_cast<const uint32_t*>(lds_tbl)};
  const size_t nchunks = (n + BLOCK - 1) / BLOCK;
  for (size_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const size_t i = chunk * BLOCK + threadIdx.x;
    const bool live = i < n;
    const size_t ii = live ? i : 0;
    uint32_t wx[8], wk[8], s[16];
    load_words8(wx, x, ii);
    load_words8(wk, k, ii);
    const uint32_t off = msg_off[ii], len = msg_off[ii + 1] - off;
    schnorr_sign(s, wx, wk, msgs + off, len, tbl);
    if (live) { store_words8(sig, 2 * ii, s); store_words8(sig, 2 * ii + 1, s + 8); }
  }
}

// EdDSA front end: secret scalar and deterministic nonce of every (seed, msg) pair (curve.rs:74-87,
// eddsa_sig.rs:120-131); the signing pipeline proper follows
__global__ void __launch_bounds__(KYB_BLOCK)
k_eddsa_prep(const uint8_t* __restrict__ seeds, const uint8_t* __restrict__ msgs, const uint32_t* __restrict__ msg_off, size_t n,
             uint8_t* __restrict__ xbuf, uint8_t* __restrict__ kbuf) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t seed[8], x[8], r[8];
  load_words8(seed, seeds, i);
  const uint32_t off = msg_off[i], len = msg_off[i + 1] - off;
  eddsa_expand_and_nonce(x, r, seed, msgs + off, len);
  store_words8(xbuf, i, x);
  store_words8(kbuf, i, r);
}

// split signing, last stage: enc holds enc(R_i) at record i and enc(A_i) at record n + i (produced by
// two split fixed-base launches + k_finish); h = SHA-512(R || A || msg) mod L, s = k + x h mod L.
__global__ void __launch_bounds__(KYB_BLOCK)
k_sign_hash(const uint8_t* __restrict__ x, const uint8_t* __restrict__ k, const uint8_t* __restrict__ msgs,
            const uint32_t* __restrict__ msg_off, size_t n, const uint8_t* __restrict__ enc, uint8_t* __restrict__ sig) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t wx[8], wk[8], ra[16];
  load_words8(wx, x, i);
  load_words8(wk, k, i);
  load_words8(ra, enc, i);
  load_words8(ra + 8, enc, n + i);
  sha512_ctx c;
  sha512_init(c);
  sha512_words64(c, ra);
  const uint32_t off = msg_off[i], len = msg_off[i + 1] - off;
  sha512_bytes(c, msgs + off, len);
  uint32_t dig[16], h[8], s[8];
  sha512_final(dig, c);
  sc_reduce512(h, dig);
  sc_muladd(s, wx, h, wk);
  store_words8(sig, 2 * i, ra);
  store_words8(sig, 2 * i + 1, s);
}

// unmarshal_binary for the ladder path: extended limbs out, failed decodes replaced by the neutral element
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_decode_or_identity(const uint8_t* __restrict__ enc, size_t n, int32_t* __restrict__ out_ext, uint8_t* __restrict__ ok_out) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
  load_words8(w, enc, i);
  ge_p3 P, id;
  const uint32_t ok = ge_decode(P, w);
  ge_p3_0(id);
  fe_cmov(P.X, id.X, 1u - ok); fe_cmov(P.Y, id.Y, 1u - ok); fe_cmov(P.Z, id.Z, 1u - ok); fe_cmov(P.T, id.T, 1u - ok);
  store_ext(out_ext, i, P.X, P.Y, P.Z, P.T);
  if (ok_out != nullptr) ok_out[i] = (uint8_t)ok;
}

// ---- table-free variable base (ge_ladder.h) -------------------------------------------------------
// Montgomery images of the input points, one field inversion per FINISH_K items.  Output record of item
// i in the staging buffer: quads 0..4 = u[10] v[10], quad 5.x = flags (the ladder kernel later overwrites
// the same record with the projective result).
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_mont_prep(const int32_t* __restrict__ pts_ext, size_t n, uint4* __restrict__ proj, size_t stride) {
  const size_t M = (n + FINISH_K - 1) / FINISH_K;
  const size_t j = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (j >= M) return;
  auto load = [&](int t, fe& d) {
    const size_t i = j + (size_t)t * M;
    if (i < n) { ge_p3 P; uint32_t fl; load_ext(P, pts_ext, i); mont_prep_den(d, fl, P); }
    else fe_one(d);
  };
  auto emit = [&](int t, const fe& dinv) {
    const size_t i = j + (size_t)t * M;
    if (i >= n) return;
    ge_p3 P;
    load_ext(P, pts_ext, i);
    fe d;
    uint32_t fl;
    mont_prep_den(d, fl, P);
    mont_point m;
    mont_prep_finish(m, P, dinv, fl);
    uint32_t f[24];
#pragma unroll
    for (int k = 0; k < 10; ++k) { f[k] = m.u.v[k]; f[10 + k] = m.v.v[k]; }
    f[20] = m.flags; f[21] = f[22] = f[23] = 0;
#pragma unroll
    for (int q = 0; q < 6; ++q) proj[q * stride + i] = make_uint4(f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]);
  };
  fe unused_prefix, unused_inv;
  fe_one(unused_prefix);
  batch_invert<0, FINISH_K>(unused_prefix, unused_inv, load, emit);
}
template <int WAVES>
__global__ void __launch_bounds__(KYB_BLOCK, WAVES)
k_mul_ladder(const uint8_t* __restrict__ scalars, size_t n, uint4* __restrict__ proj, size_t stride, size_t img_offset, size_t img_mod) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t a[8];
  load_words8(a, scalars, i);
  // Montgomery image of the operand: record i itself, or (shared operands) record img_offset + i mod img_mod
  const size_t src = img_mod ? img_offset + i % img_mod : i;
  uint32_t f[24];
#pragma unroll
  for (int q = 0; q < 6; ++q) { const uint4 v = proj[q * stride + src]; f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w; }
  mont_point m;
#pragma unroll
  for (int k = 0; k < 10; ++k) { m.u.v[k] = f[k]; m.v.v[k] = f[10 + k]; }
  m.flags = f[20];
  ge_p2 r;
  ge_scalarmult_ladder(r, a, m);
  store_proj(proj, stride, i, r.X, r.Y, r.Z);
}

// One halving pass of the segmented sum behind kyb_lincomb_batch: in each of the m groups (group g
// starts at record g * gstride and currently holds `len` partial sums) record j + half is added onto
// record j for j < len - half.  ceil(log2 t) passes leave the group total in the group's first record.
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_pair_sum(uint4* __restrict__ proj, size_t stride, size_t m, size_t gstride, size_t len, size_t half) {
  const size_t cnt = len - half;
  const size_t idx = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (idx >= m * cnt) return;
  const size_t g = idx / cnt, j = idx - g * cnt;
  const size_t ia = g * gstride + j, ib = ia + half;
  ge_p2 a, b, r;
  load_proj_xy(a.X, a.Y, proj, stride, ia); load_proj_z(a.Z, proj, stride, ia);
  load_proj_xy(b.X, b.Y, proj, stride, ib); load_proj_z(b.Z, proj, stride, ib);
  ge_p2_add(r, a, b);
  store_proj(proj, stride, ia, r.X, r.Y, r.Z);
}

// verification stage 1: checks, decode R and A, h = SHA-512(R || A || msg) mod L.
// Writes h and s as contiguous 32-byte records, A in reference limbs (input of k_mul), R into the
// projective staging buffer at [proj_offset, proj_offset + n).
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_verify_prep(const uint8_t* __restrict__ pubs, const uint8_t* __restrict__ sigs, const uint8_t* __restrict__ msgs,
              const uint32_t* __restrict__ msg_off, size_t n, int flavor, uint8_t* __restrict__ status,
              uint8_t* __restrict__ hbuf, uint8_t* __restrict__ sbuf, int32_t* __restrict__ a_ext,
              uint4* __restrict__ proj, size_t stride, size_t proj_offset) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t pub[8], sig[16], h[8];
  load_words8(pub, pubs, i);
  load_words8(sig, sigs, 2 * i);
  load_words8(sig + 8, sigs, 2 * i + 1);
  const uint32_t off = msg_off[i], len = msg_off[i + 1] - off;
  ge_p3 R, A;
  const uint32_t st = verify_prep(h, R, A, pub, sig, msgs + off, len, flavor);
  status[i] = (uint8_t)st;
  store_words8(hbuf, i, h);
  store_words8(sbuf, i, sig + 8);
  store_ext(a_ext, i, A.X, A.Y, A.Z, A.T);
  store_proj(proj, stride, proj_offset + i, R.X, R.Y, R.Z);
}
// verification stage 4: hA at proj[i], sB at proj[n + i], R at proj[2n + i]
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_verify_final(const uint4* __restrict__ proj, size_t stride, size_t n, uint8_t* __restrict__ status) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  ge_p2 hA, sB;
  fe RX, RY;
  load_proj_xy(hA.X, hA.Y, proj, stride, i);           load_proj_z(hA.Z, proj, stride, i);
  load_proj_xy(sB.X, sB.Y, proj, stride, n + i);       load_proj_z(sB.Z, proj, stride, n + i);
  load_proj_xy(RX, RY, proj, stride, 2 * n + i);
  const uint32_t eq = verify_final(RX, RY, hA, sB);
  const uint8_t st = status[i];
  status[i] = (st == 0 && !eq) ? (uint8_t)9 : st;
}

// PubPoly::eval for one polynomial at n share indices (poly.rs:457-469, shares :472-478)
template <bool SPLIT>
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_poly_eval(const int32_t* __restrict__ commits_ext, int t, const uint32_t* __restrict__ indices, size_t n, int nbits,
            uint8_t* __restrict__ out_enc, int32_t* __restrict__ out_ext, uint4* __restrict__ proj, size_t stride) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  const bool live = i < n;
  const size_t ii = live ? i : 0;
  const uint32_t x = indices[ii] + 1u;
  ge_p2 r;
  ge_poly_eval(r, [&](int j, ge_p3& c) { load_ext(c, commits_ext, (size_t)j); }, t, x, nbits);
  if (SPLIT) { if (live) store_proj(proj, stride, i, r.X, r.Y, r.Z); }
  else finish_point(r.X, r.Y, r.Z, out_enc, out_ext, ii, live);
}
// batched Point::eq (point.rs:227-241) without inversions
__global__ void __launch_bounds__(KYB_BLOCK)
k_equal(const int32_t* __restrict__ a_ext, const int32_t* __restrict__ b_ext, size_t n, uint8_t* __restrict__ eq_out) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  ge_p3 A, B;
  load_ext(A, a_ext, i);
  load_ext(B, b_ext, i);
  eq_out[i] = (uint8_t)ge_equal(A, B);
}

__global__ void __launch_bounds__(KYB_BLOCK)
k_add(const int32_t* __restrict__ a_ext, const int32_t* __restrict__ b_ext, size_t n, int32_t* __restrict__ out_ext, int subtract) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  ge_p3 A, B, R;
  load_ext(A, a_ext, i);
  load_ext(B, b_ext, i);
  ge_cached c;
  ge_p3_to_cached(c, B);
  ge_cached_cneg(c, subtract ? 1u : 0u);
  ge_p1p1 r;
  ge_add(r, A, c);
  ge_p1p1_to_p3(R, r);
  store_ext(out_ext, i, R.X, R.Y, R.Z, R.T);
}

__global__ void __launch_bounds__(KYB_BLOCK)
k_encode(const int32_t* __restrict__ pts_ext, size_t n, uint8_t* __restrict__ out_enc) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  ge_p3 P;
  load_ext(P, pts_ext, i);
  uint32_t w[8];
  ge_encode(w, P.X, P.Y, P.Z);
  store_words8(out_enc, i, w);
}

__global__ void __launch_bounds__(KYB_BLOCK)
k_decode(const uint8_t* __restrict__ enc, size_t n, int32_t* __restrict__ out_ext, uint8_t* __restrict__ ok_out) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
  load_words8(w, enc, i);
  ge_p3 P;
  const uint32_t ok = ge_decode(P, w);
  store_ext(out_ext, i, P.X, P.Y, P.Z, P.T);
  if (ok_out != nullptr) ok_out[i] = (uint8_t)ok;
}

// ------------------------------------------------------------------------------------------------
// host side: context, staging, C ABI
// ------------------------------------------------------------------------------------------------
namespace {

thread_local std::string g_err;

// optional per-launch timing (bench.py): HIP events recorded on the launch stream around each kernel
enum KernelId { KID_EDDSA_PREP = 11, KID_MUL = 0, KID_MUL_BASE = 1, KID_FINISH = 2, KID_SIGN = 3, KID_SIGN_HASH = 4, KID_VERIFY_PREP = 5, KID_VERIFY_FINAL = 6, KID_POLY_EVAL = 7, KID_MONT_PREP = 8, KID_MUL_LADDER = 9, KID_DECODE = 10, KID_PAIR_SUM = 12, KID_COUNT = 13 };
const char* const KERNEL_NAMES[KID_COUNT] = {"k_mul", "k_mul_base", "k_finish", "k_sign", "k_sign_hash", "k_verify_prep", "k_verify_final", "k_poly_eval", "k_mont_prep", "k_mul_ladder", "k_decode", "k_eddsa_prep", "k_pair_sum"};
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
  uint32_t* table = nullptr;      // KYB_BASE_TABLE_BYTES: radix-16 image (65,536 B) followed by the radix-32 image (106,496 B)
  bool table_ready = false;
  // per-stream device scratch (two launches that overlap on different streams must not share it):
  //   ws    variable-base table workspace (fixed size)
  //   proj  projective staging of the split finish (grows with the largest batch seen)
  //   enc   encodings of R and A between the stages of the split signing path
  struct StreamRes { hipStream_t stream; uint4* ws; uint4* proj; size_t proj_items; uint8_t* enc; size_t enc_bytes; };
  StreamRes res[8] = {};
  int res_count = 0;
  size_t ws_bytes = 0;
  int grid_mul = 0;
  uint8_t* stage = nullptr;       // device staging for the host-pointer API
  size_t stage_bytes = 0;
  uint8_t* stage2 = nullptr;      // staging of the second pipeline lane
  size_t stage2_bytes = 0;
  int opt_mul_select = 1;         // 0 cndmask, 1 and/or mask
  int opt_base_select = 1;        // 0 LDS broadcast scan, 1 bpermute
  int opt_base_block = 256;       // 256 (2 waves/SIMD) or 512 (4 waves/SIMD, 128 VGPRs)   [radix-16 kernel]
  int opt_base_radix = 32;        // 32: 52-window kernel for batches >= finish.min_items; 16: always the 64-window kernel
  int opt_mul_algo = 1;           // 0 windowed table (ge.rs structure), 1 Montgomery ladder (table-free, 1.33x faster: profiles/r01/sweep_mul_algo.log)
  int opt_ladder_waves = 3;       // launch bound of k_mul_ladder: waves per SIMD the register allocator must allow
  int opt_finish = 1;             // 0 fused inversion per item, 1 split + batched inversion (n >= finish_min)
  int opt_finish_min = 4096;
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

// Host-pointer batches of fixed-size records: the batch is cut into chunks that alternate between two
// streams (each with its own staging and scratch) so that the H2D copy of chunk c+1 and the D2H copy
// of chunk c-1 overlap the kernels of chunk c.  (Caller buffers are pageable, so the copies block the
// calling thread; the order below is what creates the overlap.)
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
int launch_ladder_core(const uint8_t* sc, const uint8_t* penc, const int32_t* pext, size_t n, uint8_t* ok, Ctx::StreamRes* r, hipStream_t st,
                       size_t npts = 0) {
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
    if (g.opt_ladder_waves >= 4)      hipLaunchKernelGGL((k_mul_ladder<4>), dim3(blocks), dim3(KYB_BLOCK), 0, st, sc, n, r->proj, r->proj_items, n, npts);
    else if (g.opt_ladder_waves == 3) hipLaunchKernelGGL((k_mul_ladder<3>), dim3(blocks), dim3(KYB_BLOCK), 0, st, sc, n, r->proj, r->proj_items, n, npts);
    else                              hipLaunchKernelGGL((k_mul_ladder<2>), dim3(blocks), dim3(KYB_BLOCK), 0, st, sc, n, r->proj, r->proj_items, n, npts);
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
template <bool SPLIT>
int launch_base_t(const uint8_t* sc, size_t n, uint8_t* oenc, int32_t* oext, Ctx::StreamRes* r, size_t offset, hipStream_t st) {
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
int launch_sign(const uint8_t* x, const uint8_t* k, const uint8_t* msgs, const uint32_t* off, size_t n, uint8_t* sig, hipStream_t st) {
  if (n == 0) return KYB_OK;
  std::lock_guard<std::mutex> launch_lock(g.launch_mu);
  Ctx::StreamRes* r = nullptr;
  { int rc = res_for(st, &r); if (rc) return rc; }
  if (use_split(2 * n)) {
    // R = k*B -> proj[0, n), A = x*B -> proj[n, 2n); one batched finish; then hash + scalar arithmetic
    int rc = ensure_proj(r, 2 * n); if (rc) return rc;
    rc = ensure_enc(r, 64 * n); if (rc) return rc;
    rc = launch_base_t<true>(k, n, nullptr, nullptr, r, 0, st); if (rc) return rc;
    rc = launch_base_t<true>(x, n, nullptr, nullptr, r, n, st); if (rc) return rc;
    rc = launch_finish(r, 2 * n, r->enc, nullptr, st); if (rc) return rc;
    ProfScope ps(st, KID_SIGN_HASH);
    hipLaunchKernelGGL(k_sign_hash, dim3((unsigned)((n + KYB_BLOCK - 1) / KYB_BLOCK)), dim3(KYB_BLOCK), 0, st, x, k, msgs, off, n, r->enc, sig);
    HIPCK(hipGetLastError());
    return KYB_OK;
  }
  const size_t nchunks = (n + KYB_BLOCK - 1) / KYB_BLOCK;
  const size_t cap = (size_t)g.cus * 2;
  const int grid = (int)(nchunks < cap ? nchunks : cap);
  const uint4* img = reinterpret_cast<const uint4*>(g.table);
  ProfScope ps(st, KID_SIGN);
  if (g.opt_base_select == 0) hipLaunchKernelGGL((k_sign<0, KYB_BLOCK>), dim3(grid), dim3(KYB_BLOCK), 0, st, x, k, msgs, off, n, sig, img);
  else                        hipLaunchKernelGGL((k_sign<1, KYB_BLOCK>), dim3(grid), dim3(KYB_BLOCK), 0, st, x, k, msgs, off, n, sig, img);
  HIPCK(hipGetLastError());
  return KYB_OK;
}

// EdDSA::sign for n (seed, msg) pairs: expansion + nonce, then the Schnorr pipeline; optionally the public keys
int launch_eddsa_sign(const uint8_t* seeds, const uint8_t* msgs, const uint32_t* off, size_t n, uint8_t* sig, uint8_t* pub, hipStream_t st) {
  if (n == 0) return KYB_OK;
  uint8_t *xbuf = nullptr, *kbuf = nullptr;
  {
    std::lock_guard<std::mutex> launch_lock(g.launch_mu);
    Ctx::StreamRes* r = nullptr;
    int rc = res_for(st, &r); if (rc) return rc;
    // the signing pipeline uses r->enc[0, 64n) for the encodings of R and A: keep x and k behind that
    rc = ensure_enc(r, up256(64 * n) + 2 * up256(32 * n)); if (rc) return rc;
    xbuf = r->enc + up256(64 * n); kbuf = xbuf + up256(32 * n);
    ProfScope ps(st, KID_EDDSA_PREP);
    hipLaunchKernelGGL(k_eddsa_prep, dim3((unsigned)((n + KYB_BLOCK - 1) / KYB_BLOCK)), dim3(KYB_BLOCK), 0, st, seeds, msgs, off, n, xbuf, kbuf);
    HIPCK(hipGetLastError());
  }
  int rc = launch_sign(xbuf, kbuf, msgs, off, n, sig, st); if (rc) return rc;
  if (pub != nullptr) return launch_mul_base(xbuf, n, pub, nullptr, st);
  return KYB_OK;
}

// verification pipeline on one stream: prep -> k_mul (h, A) -> k_mul_base (s) -> final
int launch_verify(const uint8_t* pubs, const uint8_t* msgs, const uint32_t* off, const uint8_t* sigs, size_t n, int flavor,
                  uint8_t* status, hipStream_t st) {
  if (n == 0) return KYB_OK;
  std::lock_guard<std::mutex> launch_lock(g.launch_mu);
  Ctx::StreamRes* r = nullptr;
  { int rc = res_for(st, &r); if (rc) return rc; }
  int rc = ensure_proj(r, 3 * n); if (rc) return rc;
  const size_t o_h = 0, o_s = up256(32 * n), o_a = o_s + up256(32 * n);
  rc = ensure_enc(r, o_a + 160 * n); if (rc) return rc;
  uint8_t* hbuf = r->enc + o_h; uint8_t* sbuf = r->enc + o_s; int32_t* a_ext = reinterpret_cast<int32_t*>(r->enc + o_a);
  const unsigned blocks = (unsigned)((n + KYB_BLOCK - 1) / KYB_BLOCK);
  {
    ProfScope ps(st, KID_VERIFY_PREP);
    hipLaunchKernelGGL(k_verify_prep, dim3(blocks), dim3(KYB_BLOCK), 0, st, pubs, sigs, msgs, off, n, flavor, status, hbuf, sbuf, a_ext, r->proj, r->proj_items, 2 * n);
  }
  HIPCK(hipGetLastError());
  if (g.opt_mul_algo == 1) {
    rc = launch_ladder_core(hbuf, nullptr, a_ext, n, nullptr, r, st); if (rc) return rc;
  } else {
    rc = ensure_ws(r); if (rc) return rc;
    const size_t nchunks = (n + KYB_BLOCK - 1) / KYB_BLOCK;
    const int grid = (int)(nchunks < (size_t)g.grid_mul ? nchunks : (size_t)g.grid_mul);
    ProfScope ps(st, KID_MUL);
    launch_mul_t<true>(g.opt_mul_select, false, grid, st, hbuf, nullptr, a_ext, n, nullptr, nullptr, nullptr, r);
    HIPCK(hipGetLastError());
  }
  rc = launch_base_t<true>(sbuf, n, nullptr, nullptr, r, n, st); if (rc) return rc;
  {
    ProfScope ps(st, KID_VERIFY_FINAL);
    hipLaunchKernelGGL(k_verify_final, dim3(blocks), dim3(KYB_BLOCK), 0, st, r->proj, r->proj_items, n, status);
  }
  HIPCK(hipGetLastError());
  return KYB_OK;
}

int launch_poly_eval(const int32_t* commits, size_t t, const uint32_t* idx, size_t n, uint32_t max_index, uint8_t* oenc, int32_t* oext, hipStream_t st) {
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
    if (split) hipLaunchKernelGGL((k_poly_eval<true>), dim3(blocks), dim3(KYB_BLOCK), 0, st, commits, (int)t, idx, n, nbits, oenc, oext, r->proj, r->proj_items);
    else       hipLaunchKernelGGL((k_poly_eval<false>), dim3(blocks), dim3(KYB_BLOCK), 0, st, commits, (int)t, idx, n, nbits, oenc, oext, r->proj, r->proj_items);
  }
  HIPCK(hipGetLastError());
  if (split) return launch_finish(r, n, oenc, oext, st);
  return KYB_OK;
}

}  // namespace

extern "C" {

int kyb_init(int device) { return do_init(device, true); }
int kyb_init_no_table(int device) { return do_init(device, false); }

void kyb_shutdown(void) {
  std::lock_guard<std::mutex> lk(g.mu);
  if (!g.ready) return;
  (void)hipSetDevice(g.device);
  (void)hipStreamSynchronize(g.stream);
  if (g.stage) (void)hipFree(g.stage);
  if (g.stage2) (void)hipFree(g.stage2);
  g.stage2 = nullptr; g.stage2_bytes = 0;
  if (g.stream2) { (void)hipStreamSynchronize(g.stream2); (void)hipStreamDestroy(g.stream2); g.stream2 = nullptr; }
  for (int i = 0; i < g.res_count; ++i) {
    if (g.res[i].ws) (void)hipFree(g.res[i].ws);
    if (g.res[i].proj) (void)hipFree(g.res[i].proj);
    if (g.res[i].enc) (void)hipFree(g.res[i].enc);
    g.res[i] = Ctx::StreamRes{};
  }
  g.res_count = 0;
  if (g.table) (void)hipFree(g.table);
  (void)hipStreamDestroy(g.stream);
  g.stage = nullptr; g.stage_bytes = 0; g.table = nullptr; g.stream = nullptr;
  g.table_ready = false; g.ready = false; g.device = -1;
}

const char* kyb_last_error(void) { return g_err.c_str(); }

int kyb_device_info(char* name, size_t name_cap, int* compute_units, size_t* workspace_bytes) {
  REQUIRE_READY();
  if (name && name_cap) snprintf(name, name_cap, "%s", g.name);
  if (compute_units) *compute_units = g.cus;
  if (workspace_bytes) *workspace_bytes = g.ws_bytes;
  return KYB_OK;
}

int kyb_sync(void* stream) {
  REQUIRE_READY();
  HIPCK(hipStreamSynchronize(pick(stream)));
  return KYB_OK;
}

int kyb_base_table_export_dev(void* dst_dev, void* stream) {
  REQUIRE_READY(); REQUIRE_TABLE();
  if (!dst_dev) return fail(KYB_E_BAD_ARG, "null destination");
  HIPCK(hipMemcpyAsync(dst_dev, g.table, KYB_BASE_TABLE_BYTES, hipMemcpyDeviceToDevice, pick(stream)));
  return KYB_OK;
}
int kyb_base_table_import_dev(const void* src_dev, void* stream) {
  REQUIRE_READY();
  if (!src_dev) return fail(KYB_E_BAD_ARG, "null source");
  HIPCK(hipMemcpyAsync(g.table, src_dev, KYB_BASE_TABLE_BYTES, hipMemcpyDeviceToDevice, pick(stream)));
  HIPCK(hipStreamSynchronize(pick(stream)));
  g.table_ready = true;
  return KYB_OK;
}
int kyb_base_table_export(uint8_t* dst_host) {
  REQUIRE_READY(); REQUIRE_TABLE();
  if (!dst_host) return fail(KYB_E_BAD_ARG, "null destination");
  HIPCK(hipMemcpy(dst_host, g.table, KYB_BASE_TABLE_BYTES, hipMemcpyDeviceToHost));
  return KYB_OK;
}

// ---- device-pointer API ----
int kyb_mul_base_batch_dev(const uint8_t* scalars, size_t n, uint8_t* out_enc, int32_t* out_ext, void* stream) {
  REQUIRE_READY(); REQUIRE_TABLE();
  if (n && (!scalars || (!out_enc && !out_ext))) return fail(KYB_E_BAD_ARG, "null buffer");
  if (!aligned16(scalars) || !aligned16(out_enc) || !aligned16(out_ext)) return fail(KYB_E_BAD_ARG, "device buffers must be 16-byte aligned");
  return launch_mul_base(scalars, n, out_enc, out_ext, pick(stream));
}
int kyb_mul_batch_dev(const uint8_t* scalars, const uint8_t* pts_enc, const int32_t* pts_ext, size_t n,
                      uint8_t* out_enc, int32_t* out_ext, uint8_t* ok, void* stream) {
  REQUIRE_READY();
  if (n && (!scalars || (!out_enc && !out_ext))) return fail(KYB_E_BAD_ARG, "null buffer");
  if (n && ((pts_enc == nullptr) == (pts_ext == nullptr))) return fail(KYB_E_BAD_ARG, "give exactly one of pts_enc / pts_ext");
  if (!aligned16(scalars) || !aligned16(pts_enc) || !aligned16(pts_ext) || !aligned16(out_enc) || !aligned16(out_ext))
    return fail(KYB_E_BAD_ARG, "device buffers must be 16-byte aligned");
  return launch_mul(scalars, pts_enc, pts_ext, n, out_enc, out_ext, ok, pick(stream));
}
int kyb_add_batch_dev(const int32_t* a_ext, const int32_t* b_ext, size_t n, int32_t* out_ext, int subtract, void* stream) {
  REQUIRE_READY();
  if (n == 0) return KYB_OK;
  if (!a_ext || !b_ext || !out_ext) return fail(KYB_E_BAD_ARG, "null buffer");
  if (!aligned16(a_ext) || !aligned16(b_ext) || !aligned16(out_ext)) return fail(KYB_E_BAD_ARG, "device buffers must be 16-byte aligned");
  hipLaunchKernelGGL(k_add, dim3((unsigned)((n + KYB_BLOCK - 1) / KYB_BLOCK)), dim3(KYB_BLOCK), 0, pick(stream), a_ext, b_ext, n, out_ext, subtract);
  HIPCK(hipGetLastError());
  return KYB_OK;
}
int kyb_encode_batch_dev(const int32_t* pts_ext, size_t n, uint8_t* out_enc, void* stream) {
  REQUIRE_READY();
  if (n == 0) return KYB_OK;
  if (!pts_ext || !out_enc) return fail(KYB_E_BAD_ARG, "null buffer");
  if (!aligned16(pts_ext) || !aligned16(out_enc)) return fail(KYB_E_BAD_ARG, "device buffers must be 16-byte aligned");
  hipLaunchKernelGGL(k_encode, dim3((unsigned)((n + KYB_BLOCK - 1) / KYB_BLOCK)), dim3(KYB_BLOCK), 0, pick(stream), pts_ext, n, out_enc);
  HIPCK(hipGetLastError());
  return KYB_OK;
}
int kyb_decode_batch_dev(const uint8_t* enc, size_t n, int32_t* out_ext, uint8_t* ok, void* stream) {
  REQUIRE_READY();
  if (n == 0) return KYB_OK;
  if (!enc || !out_ext) return fail(KYB_E_BAD_ARG, "null buffer");
  if (!aligned16(enc) || !aligned16(out_ext)) return fail(KYB_E_BAD_ARG, "device buffers must be 16-byte aligned");
  hipLaunchKernelGGL(k_decode, dim3((unsigned)((n + KYB_BLOCK - 1) / KYB_BLOCK)), dim3(KYB_BLOCK), 0, pick(stream), enc, n, out_ext, ok);
  HIPCK(hipGetLastError());
  return KYB_OK;
}
int kyb_schnorr_sign_batch_dev(const uint8_t* x, const uint8_t* k, const uint8_t* msgs, const uint32_t* msg_off, size_t n, uint8_t* sig, void* stream) {
  REQUIRE_READY(); REQUIRE_TABLE();
  if (n == 0) return KYB_OK;
  if (!x || !k || !msg_off || !sig) return fail(KYB_E_BAD_ARG, "null buffer");
  if (!aligned16(x) || !aligned16(k) || !aligned16(sig)) return fail(KYB_E_BAD_ARG, "device buffers must be 16-byte aligned");
  return launch_sign(x, k, msgs, msg_off, n, sig, pick(stream));
}

// ---- host-pointer API: stage through one device buffer, run on the engine stream, copy back ----
int kyb_mul_base_batch(const uint8_t* scalars, size_t n, uint8_t* out_enc, int32_t* out_ext) {
  REQUIRE_READY(); REQUIRE_TABLE();
  if (n == 0) return KYB_OK;
  if (!scalars || (!out_enc && !out_ext)) return fail(KYB_E_BAD_ARG, "null buffer");
  const HostArr arrs[3] = {{scalars, nullptr, 32}, {nullptr, out_enc, 32}, {nullptr, out_ext, 160}};
  return run_host_batch(n, arrs, 3, [&](hipStream_t st, size_t cn, uint8_t** d) {
    return launch_mul_base(d[0], cn, d[1], reinterpret_cast<int32_t*>(d[2]), st);
  });
}
int kyb_mul_batch(const uint8_t* scalars, const uint8_t* pts_enc, const int32_t* pts_ext, size_t n,
                  uint8_t* out_enc, int32_t* out_ext, uint8_t* ok) {
  REQUIRE_READY();
  if (n == 0) return KYB_OK;
  if (!scalars || (!out_enc && !out_ext)) return fail(KYB_E_BAD_ARG, "null buffer");
  if ((pts_enc == nullptr) == (pts_ext == nullptr)) return fail(KYB_E_BAD_ARG, "give exactly one of pts_enc / pts_ext");
  const HostArr arrs[6] = {{scalars, nullptr, 32}, {pts_enc, nullptr, 32}, {pts_ext, nullptr, 160},
                           {nullptr, out_enc, 32}, {nullptr, out_ext, 160}, {nullptr, ok, 1}};
  return run_host_batch(n, arrs, 6, [&](hipStream_t st, size_t cn, uint8_t** d) {
    return launch_mul(d[0], d[1], reinterpret_cast<const int32_t*>(d[2]), cn, d[3], reinterpret_cast<int32_t*>(d[4]), d[5], st);
  });
}
int kyb_add_batch(const int32_t* a_ext, const int32_t* b_ext, size_t n, int32_t* out_ext, int subtract) {
  REQUIRE_READY();
  if (n == 0) return KYB_OK;
  if (!a_ext || !b_ext || !out_ext) return fail(KYB_E_BAD_ARG, "null buffer");
  std::lock_guard<std::mutex> lk(g.mu);
  HIPCK(hipSetDevice(g.device));
  const size_t o_a = 0, o_b = up256(160 * n), o_o = 2 * up256(160 * n), total = 3 * up256(160 * n);
  int rc = ensure_stage(total);
  if (rc) return rc;
  uint8_t* d = g.stage;
  HIPCK(hipMemcpyAsync(d + o_a, a_ext, 160 * n, hipMemcpyHostToDevice, g.stream));
  HIPCK(hipMemcpyAsync(d + o_b, b_ext, 160 * n, hipMemcpyHostToDevice, g.stream));
  hipLaunchKernelGGL(k_add, dim3((unsigned)((n + KYB_BLOCK - 1) / KYB_BLOCK)), dim3(KYB_BLOCK), 0, g.stream,
                     reinterpret_cast<const int32_t*>(d + o_a), reinterpret_cast<const int32_t*>(d + o_b), n, reinterpret_cast<int32_t*>(d + o_o), subtract);
  HIPCK(hipGetLastError());
  HIPCK(hipMemcpyAsync(out_ext, d + o_o, 160 * n, hipMemcpyDeviceToHost, g.stream));
  HIPCK(hipStreamSynchronize(g.stream));
  return KYB_OK;
}
int kyb_encode_batch(const int32_t* pts_ext, size_t n, uint8_t* out_enc) {
  REQUIRE_READY();
  if (n == 0) return KYB_OK;
  if (!pts_ext || !out_enc) return fail(KYB_E_BAD_ARG, "null buffer");
  std::lock_guard<std::mutex> lk(g.mu);
  HIPCK(hipSetDevice(g.device));
  const size_t o_p = 0, o_e = up256(160 * n), total = o_e + up256(32 * n);
  int rc = ensure_stage(total);
  if (rc) return rc;
  uint8_t* d = g.stage;
  HIPCK(hipMemcpyAsync(d + o_p, pts_ext, 160 * n, hipMemcpyHostToDevice, g.stream));
  hipLaunchKernelGGL(k_encode, dim3((unsigned)((n + KYB_BLOCK - 1) / KYB_BLOCK)), dim3(KYB_BLOCK), 0, g.stream, reinterpret_cast<const int32_t*>(d + o_p), n, d + o_e);
  HIPCK(hipGetLastError());
  HIPCK(hipMemcpyAsync(out_enc, d + o_e, 32 * n, hipMemcpyDeviceToHost, g.stream));
  HIPCK(hipStreamSynchronize(g.stream));
  return KYB_OK;
}
int kyb_decode_batch(const uint8_t* enc, size_t n, int32_t* out_ext, uint8_t* ok) {
  REQUIRE_READY();
  if (n == 0) return KYB_OK;
  if (!enc || !out_ext) return fail(KYB_E_BAD_ARG, "null buffer");
  std::lock_guard<std::mutex> lk(g.mu);
  HIPCK(hipSetDevice(g.device));
  const size_t o_e = 0, o_p = up256(32 * n), o_ok = o_p + up256(160 * n), total = o_ok + up256(n);
  int rc = ensure_stage(total);
  if (rc) return rc;
  uint8_t* d = g.stage;
  HIPCK(hipMemcpyAsync(d + o_e, enc, 32 * n, hipMemcpyHostToDevice, g.stream));
  hipLaunchKernelGGL(k_decode, dim3((unsigned)((n + KYB_BLOCK - 1) / KYB_BLOCK)), dim3(KYB_BLOCK), 0, g.stream, d + o_e, n, reinterpret_cast<int32_t*>(d + o_p), d + o_ok);
  HIPCK(hipGetLastError());
  HIPCK(hipMemcpyAsync(out_ext, d + o_p, 160 * n, hipMemcpyDeviceToHost, g.stream));
  if (ok) HIPCK(hipMemcpyAsync(ok, d + o_ok, n, hipMemcpyDeviceToHost, g.stream));
  HIPCK(hipStreamSynchronize(g.stream));
  return KYB_OK;
}
int kyb_schnorr_sign_batch(const uint8_t* x, const uint8_t* k, const uint8_t* msgs, const uint32_t* msg_off, size_t n, uint8_t* sig) {
  REQUIRE_READY(); REQUIRE_TABLE();
  if (n == 0) return KYB_OK;
  if (!x || !k || !msg_off || !sig) return fail(KYB_E_BAD_ARG, "null buffer");
  const size_t mbytes = msg_off[n];
  if (mbytes && !msgs) return fail(KYB_E_BAD_ARG, "null message buffer");
  for (size_t i = 0; i < n; ++i) if (msg_off[i + 1] < msg_off[i]) return fail(KYB_E_BAD_ARG, "msg_off must be non-decreasing");
  std::lock_guard<std::mutex> lk(g.mu);
  HIPCK(hipSetDevice(g.device));
  const size_t o_x = 0, o_k = up256(32 * n), o_m = o_k + up256(32 * n), o_off = o_m + up256(mbytes + 16), o_sig = o_off + up256(4 * (n + 1)), total = o_sig + up256(64 * n);
  int rc = ensure_stage(total);
  if (rc) return rc;
  uint8_t* d = g.stage;
  HIPCK(hipMemcpyAsync(d + o_x, x, 32 * n, hipMemcpyHostToDevice, g.stream));
  HIPCK(hipMemcpyAsync(d + o_k, k, 32 * n, hipMemcpyHostToDevice, g.stream));
  if (mbytes) HIPCK(hipMemcpyAsync(d + o_m, msgs, mbytes, hipMemcpyHostToDevice, g.stream));
  HIPCK(hipMemcpyAsync(d + o_off, msg_off, 4 * (n + 1), hipMemcpyHostToDevice, g.stream));
  rc = launch_sign(d + o_x, d + o_k, d + o_m, reinterpret_cast<const uint32_t*>(d + o_off), n, d + o_sig, g.stream);
  if (rc) return rc;
  HIPCK(hipMemcpyAsync(sig, d + o_sig, 64 * n, hipMemcpyDeviceToHost, g.stream));
  HIPCK(hipStreamSynchronize(g.stream));
  return KYB_OK;
}

int kyb_eddsa_sign_batch_dev(const uint8_t* seeds, const uint8_t* msgs, const uint32_t* msg_off, size_t n, uint8_t* sig, uint8_t* pub, void* stream) {
  REQUIRE_READY(); REQUIRE_TABLE();
  if (n == 0) return KYB_OK;
  if (!seeds || !msg_off || !sig) return fail(KYB_E_BAD_ARG, "null buffer");
  if (!aligned16(seeds) || !aligned16(sig) || !aligned16(pub)) return fail(KYB_E_BAD_ARG, "device buffers must be 16-byte aligned");
  return launch_eddsa_sign(seeds, msgs, msg_off, n, sig, pub, pick(stream));
}
int kyb_eddsa_sign_batch(const uint8_t* seeds, const uint8_t* msgs, const uint32_t* msg_off, size_t n, uint8_t* sig, uint8_t* pub) {
  REQUIRE_READY(); REQUIRE_TABLE();
  if (n == 0) return KYB_OK;
  if (!seeds || !msg_off || !sig) return fail(KYB_E_BAD_ARG, "null buffer");
  const size_t mbytes = msg_off[n];
  if (mbytes && !msgs) return fail(KYB_E_BAD_ARG, "null message buffer");
  for (size_t i = 0; i < n; ++i) if (msg_off[i + 1] < msg_off[i]) return fail(KYB_E_BAD_ARG, "msg_off must be non-decreasing");
  std::lock_guard<std::mutex> lk(g.mu);
  HIPCK(hipSetDevice(g.device));
  const size_t o_s = 0, o_m = up256(32 * n), o_off = o_m + up256(mbytes + 16), o_sig = o_off + up256(4 * (n + 1)), o_pub = o_sig + up256(64 * n), total = o_pub + up256(32 * n);
  int rc = ensure_stage(total);
  if (rc) return rc;
  uint8_t* d = g.stage;
  HIPCK(hipMemcpyAsync(d + o_s, seeds, 32 * n, hipMemcpyHostToDevice, g.stream));
  if (mbytes) HIPCK(hipMemcpyAsync(d + o_m, msgs, mbytes, hipMemcpyHostToDevice, g.stream));
  HIPCK(hipMemcpyAsync(d + o_off, msg_off, 4 * (n + 1), hipMemcpyHostToDevice, g.stream));
  rc = launch_eddsa_sign(d + o_s, d + o_m, reinterpret_cast<const uint32_t*>(d + o_off), n, d + o_sig, pub ? d + o_pub : nullptr, g.stream);
  if (rc) return rc;
  HIPCK(hipMemcpyAsync(sig, d + o_sig, 64 * n, hipMemcpyDeviceToHost, g.stream));
  if (pub) HIPCK(hipMemcpyAsync(pub, d + o_pub, 32 * n, hipMemcpyDeviceToHost, g.stream));
  HIPCK(hipStreamSynchronize(g.stream));
  return KYB_OK;
}

int kyb_verify_batch_dev(const uint8_t* pubs, const uint8_t* msgs, const uint32_t* msg_off, const uint8_t* sigs, size_t n, int flavor,
                         uint8_t* status, void* stream) {
  REQUIRE_READY(); REQUIRE_TABLE();
  if (n == 0) return KYB_OK;
  if (!pubs || !msg_off || !sigs || !status) return fail(KYB_E_BAD_ARG, "null buffer");
  if (flavor != 0 && flavor != 1) return fail(KYB_E_BAD_ARG, "flavor: 0 = eddsa check order, 1 = schnorr check order");
  if (!aligned16(pubs) || !aligned16(sigs)) return fail(KYB_E_BAD_ARG, "device buffers must be 16-byte aligned");
  return launch_verify(pubs, msgs, msg_off, sigs, n, flavor, status, pick(stream));
}
int kyb_verify_batch(const uint8_t* pubs, const uint8_t* msgs, const uint32_t* msg_off, const uint8_t* sigs, size_t n, int flavor, uint8_t* status) {
  REQUIRE_READY(); REQUIRE_TABLE();
  if (n == 0) return KYB_OK;
  if (!pubs || !msg_off || !sigs || !status) return fail(KYB_E_BAD_ARG, "null buffer");
  if (flavor != 0 && flavor != 1) return fail(KYB_E_BAD_ARG, "flavor: 0 = eddsa check order, 1 = schnorr check order");
  const size_t mbytes = msg_off[n];
  if (mbytes && !msgs) return fail(KYB_E_BAD_ARG, "null message buffer");
  for (size_t i = 0; i < n; ++i) if (msg_off[i + 1] < msg_off[i]) return fail(KYB_E_BAD_ARG, "msg_off must be non-decreasing");
  std::lock_guard<std::mutex> lk(g.mu);
  HIPCK(hipSetDevice(g.device));
  const size_t o_p = 0, o_s = up256(32 * n), o_m = o_s + up256(64 * n), o_off = o_m + up256(mbytes + 16), o_st = o_off + up256(4 * (n + 1)), total = o_st + up256(n);
  int rc = ensure_stage(total);
  if (rc) return rc;
  uint8_t* d = g.stage;
  HIPCK(hipMemcpyAsync(d + o_p, pubs, 32 * n, hipMemcpyHostToDevice, g.stream));
  HIPCK(hipMemcpyAsync(d + o_s, sigs, 64 * n, hipMemcpyHostToDevice, g.stream));
  if (mbytes) HIPCK(hipMemcpyAsync(d + o_m, msgs, mbytes, hipMemcpyHostToDevice, g.stream));
  HIPCK(hipMemcpyAsync(d + o_off, msg_off, 4 * (n + 1), hipMemcpyHostToDevice, g.stream));
  rc = launch_verify(d + o_p, d + o_m, reinterpret_cast<const uint32_t*>(d + o_off), d + o_s, n, flavor, d + o_st, g.stream);
  if (rc) return rc;
  HIPCK(hipMemcpyAsync(status, d + o_st, n, hipMemcpyDeviceToHost, g.stream));
  HIPCK(hipStreamSynchronize(g.stream));
  return KYB_OK;
}

int kyb_pubpoly_eval_batch_dev(const int32_t* commits_ext, size_t t, const uint32_t* indices, size_t n, uint32_t max_index,
                               uint8_t* out_enc, int32_t* out_ext, void* stream) {
  REQUIRE_READY();
  if (n == 0) return KYB_OK;
  if (!commits_ext || !indices || (!out_enc && !out_ext) || t == 0 || t > (1u << 20)) return fail(KYB_E_BAD_ARG, "bad argument");
  if (max_index == 0xffffffffu) return fail(KYB_E_BAD_ARG, "index + 1 must fit 32 bits");
  if (!aligned16(commits_ext) || !aligned16(out_enc) || !aligned16(out_ext)) return fail(KYB_E_BAD_ARG, "device buffers must be 16-byte aligned");
  return launch_poly_eval(commits_ext, t, indices, n, max_index, out_enc, out_ext, pick(stream));
}
int kyb_pubpoly_eval_batch(const int32_t* commits_ext, size_t t, const uint32_t* indices, size_t n, uint8_t* out_enc, int32_t* out_ext) {
  REQUIRE_READY();
  if (n == 0) return KYB_OK;
  if (!commits_ext || !indices || (!out_enc && !out_ext) || t == 0 || t > (1u << 20)) return fail(KYB_E_BAD_ARG, "bad argument");
  uint32_t mx = 0;
  for (size_t i = 0; i < n; ++i) mx = indices[i] > mx ? indices[i] : mx;
  if (mx == 0xffffffffu) return fail(KYB_E_BAD_ARG, "index + 1 must fit 32 bits");
  std::lock_guard<std::mutex> lk(g.mu);
  HIPCK(hipSetDevice(g.device));
  const size_t o_c = 0, o_i = up256(160 * t), o_e = o_i + up256(4 * n), o_x = o_e + up256(32 * n), total = o_x + up256(160 * n);
  int rc = ensure_stage(total);
  if (rc) return rc;
  uint8_t* d = g.stage;
  HIPCK(hipMemcpyAsync(d + o_c, commits_ext, 160 * t, hipMemcpyHostToDevice, g.stream));
  HIPCK(hipMemcpyAsync(d + o_i, indices, 4 * n, hipMemcpyHostToDevice, g.stream));
  rc = launch_poly_eval(reinterpret_cast<const int32_t*>(d + o_c), t, reinterpret_cast<const uint32_t*>(d + o_i), n, mx,
                        out_enc ? d + o_e : nullptr, out_ext ? reinterpret_cast<int32_t*>(d + o_x) : nullptr, g.stream);
  if (rc) return rc;
  if (out_enc) HIPCK(hipMemcpyAsync(out_enc, d + o_e, 32 * n, hipMemcpyDeviceToHost, g.stream));
  if (out_ext) HIPCK(hipMemcpyAsync(out_ext, d + o_x, 160 * n, hipMemcpyDeviceToHost, g.stream));
  HIPCK(hipStreamSynchronize(g.stream));
  return KYB_OK;
}
int kyb_lincomb_batch_dev(const uint8_t* scalars, const uint8_t* pts_enc, const int32_t* pts_ext, int shared_points,
                          size_t m, size_t t, uint8_t* out_enc, int32_t* out_ext, uint8_t* ok, void* stream) {
  REQUIRE_READY();
  if (m == 0) return KYB_OK;
  if (t == 0 || t > (size_t(1) << 24) || m > (size_t(1) << 28) / t) return fail(KYB_E_BAD_ARG, "m * t out of range");
  if (!scalars || (!out_enc && !out_ext)) return fail(KYB_E_BAD_ARG, "null buffer");
  if ((pts_enc == nullptr) == (pts_ext == nullptr)) return fail(KYB_E_BAD_ARG, "give exactly one of pts_enc / pts_ext");
  if (!aligned16(scalars) || !aligned16(pts_enc) || !aligned16(pts_ext) || !aligned16(out_enc) || !aligned16(out_ext))
    return fail(KYB_E_BAD_ARG, "device buffers must be 16-byte aligned");
  return launch_lincomb(scalars, pts_enc, pts_ext, shared_points != 0, m, t, ok, out_enc, out_ext, pick(stream));
}
int kyb_lincomb_batch(const uint8_t* scalars, const uint8_t* pts_enc, const int32_t* pts_ext, int shared_points,
                      size_t m, size_t t, uint8_t* out_enc, int32_t* out_ext, uint8_t* ok) {
  REQUIRE_READY();
  if (m == 0) return KYB_OK;
  if (t == 0 || t > (size_t(1) << 24) || m > (size_t(1) << 28) / t) return fail(KYB_E_BAD_ARG, "m * t out of range");
  if (!scalars || (!out_enc && !out_ext)) return fail(KYB_E_BAD_ARG, "null buffer");
  if ((pts_enc == nullptr) == (pts_ext == nullptr)) return fail(KYB_E_BAD_ARG, "give exactly one of pts_enc / pts_ext");
  std::lock_guard<std::mutex> lk(g.mu);
  HIPCK(hipSetDevice(g.device));
  const size_t n = m * t, np = shared_points ? t : n, pt_bytes = (pts_enc ? 32 : 160) * np;
  const size_t o_s = 0, o_p = up256(32 * n), o_e = o_p + up256(pt_bytes), o_x = o_e + up256(32 * m), o_k = o_x + up256(160 * m), total = o_k + up256(np);
  int rc = ensure_stage(total);
  if (rc) return rc;
  uint8_t* d = g.stage;
  HIPCK(hipMemcpyAsync(d + o_s, scalars, 32 * n, hipMemcpyHostToDevice, g.stream));
  HIPCK(hipMemcpyAsync(d + o_p, pts_enc ? static_cast<const void*>(pts_enc) : static_cast<const void*>(pts_ext), pt_bytes, hipMemcpyHostToDevice, g.stream));
  rc = launch_lincomb(d + o_s, pts_enc ? d + o_p : nullptr, pts_enc ? nullptr : reinterpret_cast<const int32_t*>(d + o_p), shared_points != 0, m, t,
                      ok ? d + o_k : nullptr, out_enc ? d + o_e : nullptr, out_ext ? reinterpret_cast<int32_t*>(d + o_x) : nullptr, g.stream);
  if (rc) return rc;
  if (out_enc) HIPCK(hipMemcpyAsync(out_enc, d + o_e, 32 * m, hipMemcpyDeviceToHost, g.stream));
  if (out_ext) HIPCK(hipMemcpyAsync(out_ext, d + o_x, 160 * m, hipMemcpyDeviceToHost, g.stream));
  if (ok) HIPCK(hipMemcpyAsync(ok, d + o_k, np, hipMemcpyDeviceToHost, g.stream));
  HIPCK(hipStreamSynchronize(g.stream));
  return KYB_OK;
}
int kyb_equal_batch_dev(const int32_t* a_ext, const int32_t* b_ext, size_t n, uint8_t* eq, void* stream) {
  REQUIRE_READY();
  if (n == 0) return KYB_OK;
  if (!a_ext || !b_ext || !eq) return fail(KYB_E_BAD_ARG, "null buffer");
  if (!aligned16(a_ext) || !aligned16(b_ext)) return fail(KYB_E_BAD_ARG, "device buffers must be 16-byte aligned");
  hipLaunchKernelGGL(k_equal, dim3((unsigned)((n + KYB_BLOCK - 1) / KYB_BLOCK)), dim3(KYB_BLOCK), 0, pick(stream), a_ext, b_ext, n, eq);
  HIPCK(hipGetLastError());
  return KYB_OK;
}
int kyb_equal_batch(const int32_t* a_ext, const int32_t* b_ext, size_t n, uint8_t* eq) {
  REQUIRE_READY();
  if (n == 0) return KYB_OK;
  if (!a_ext || !b_ext || !eq) return fail(KYB_E_BAD_ARG, "null buffer");
  std::lock_guard<std::mutex> lk(g.mu);
  HIPCK(hipSetDevice(g.device));
  const size_t o_a = 0, o_b = up256(160 * n), o_e = 2 * up256(160 * n), total = o_e + up256(n);
  int rc = ensure_stage(total);
  if (rc) return rc;
  uint8_t* d = g.stage;
  HIPCK(hipMemcpyAsync(d + o_a, a_ext, 160 * n, hipMemcpyHostToDevice, g.stream));
  HIPCK(hipMemcpyAsync(d + o_b, b_ext, 160 * n, hipMemcpyHostToDevice, g.stream));
  hipLaunchKernelGGL(k_equal, dim3((unsigned)((n + KYB_BLOCK - 1) / KYB_BLOCK)), dim3(KYB_BLOCK), 0, g.stream,
                     reinterpret_cast<const int32_t*>(d + o_a), reinterpret_cast<const int32_t*>(d + o_b), n, d + o_e);
  HIPCK(hipGetLastError());
  HIPCK(hipMemcpyAsync(eq, d + o_e, n, hipMemcpyDeviceToHost, g.stream));
  HIPCK(hipStreamSynchronize(g.stream));
  return KYB_OK;
}

void* kyb_host_alloc(size_t bytes) {
  if (!g.ready) { (void)fail(KYB_E_NOT_INIT, "kyb_init has not succeeded in this process"); return nullptr; }
  void* p = nullptr;
  hipError_t e = hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault);
  if (e != hipSuccess) { (void)fail(KYB_E_NOMEM, "pinned host allocation", e); return nullptr; }
  return p;
}
void kyb_host_free(void* p) { if (p) (void)hipHostFree(p); }

int kyb_profile_begin(int max_launches) {
  REQUIRE_READY();
  if (max_launches < 0 || max_launches > 65536) return fail(KYB_E_BAD_ARG, "max_launches out of range");
  for (int i = 0; i < g_prof.cap; ++i) { (void)hipEventDestroy(g_prof.recs[i].a); (void)hipEventDestroy(g_prof.recs[i].b); }
  delete[] g_prof.recs;
  g_prof = Prof{};
  if (max_launches == 0) return KYB_OK;
  g_prof.recs = new ProfRec[max_launches];
  for (int i = 0; i < max_launches; ++i) { HIPCK(hipEventCreate(&g_prof.recs[i].a)); HIPCK(hipEventCreate(&g_prof.recs[i].b)); g_prof.cap = i + 1; }
  g_prof.on = true;
  return KYB_OK;
}
int kyb_profile_read(int* kernel_ids, float* ms, int cap, int* count) {
  REQUIRE_READY();
  if (!kernel_ids || !ms || !count) return fail(KYB_E_BAD_ARG, "null argument");
  g_prof.on = false;
  int nrec = g_prof.used < cap ? g_prof.used : cap;
  for (int i = 0; i < nrec; ++i) {
    HIPCK(hipEventSynchronize(g_prof.recs[i].b));
    kernel_ids[i] = g_prof.recs[i].id;
    HIPCK(hipEventElapsedTime(&ms[i], g_prof.recs[i].a, g_prof.recs[i].b));
  }
  *count = nrec;
  return KYB_OK;
}
const char* kyb_kernel_name(int kernel_id) { return (kernel_id >= 0 && kernel_id < KID_COUNT) ? KERNEL_NAMES[kernel_id] : ""; }

int kyb_set_option(const char* key, int value) {
  if (!key) return fail(KYB_E_BAD_ARG, "null key");
  if (!strcmp(key, "mul.select")) { if (value < 0 || value > 1) return fail(KYB_E_BAD_ARG, "mul.select in {0,1}"); g.opt_mul_select = value; return KYB_OK; }
  if (!strcmp(key, "mul_base.select")) { if (value < 0 || value > 1) return fail(KYB_E_BAD_ARG, "mul_base.select in {0,1}"); g.opt_base_select = value; return KYB_OK; }
  if (!strcmp(key, "mul.algo")) { if (value < 0 || value > 1) return fail(KYB_E_BAD_ARG, "mul.algo in {0 window table, 1 ladder}"); g.opt_mul_algo = value; return KYB_OK; }
  if (!strcmp(key, "mul.ladder_waves")) { if (value < 2 || value > 4) return fail(KYB_E_BAD_ARG, "mul.ladder_waves in 2..4"); g.opt_ladder_waves = value; return KYB_OK; }
  if (!strcmp(key, "mul_base.radix")) { if (value != 16 && value != 32) return fail(KYB_E_BAD_ARG, "mul_base.radix in {16,32}"); g.opt_base_radix = value; return KYB_OK; }
  if (!strcmp(key, "mul_base.block")) { if (value != 256 && value != 512) return fail(KYB_E_BAD_ARG, "mul_base.block in {256,512}"); g.opt_base_block = value; return KYB_OK; }
  if (!strcmp(key, "finish.batched")) { if (value < 0 || value > 1) return fail(KYB_E_BAD_ARG, "finish.batched in {0,1}"); g.opt_finish = value; return KYB_OK; }
  if (!strcmp(key, "finish.min_items")) { if (value < 1) return fail(KYB_E_BAD_ARG, "finish.min_items >= 1"); g.opt_finish_min = value; return KYB_OK; }
  if (!strcmp(key, "mul.grid_per_cu")) { if (value < 1 || value > 8 || !g.ready) return fail(KYB_E_BAD_ARG, "mul.grid_per_cu in 1..8 after init");
    if (value > 2) return fail(KYB_E_BAD_ARG, "workspace is sized for 2 blocks per CU");
    g.grid_mul = g.cus * value; return KYB_OK; }
  return fail(KYB_E_BAD_ARG, "unknown option");
}
int kyb_get_option(const char* key, int* value) {
  if (!key || !value) return fail(KYB_E_BAD_ARG, "null argument");
  if (!strcmp(key, "mul.select")) { *value = g.opt_mul_select; return KYB_OK; }
  if (!strcmp(key, "mul_base.select")) { *value = g.opt_base_select; return KYB_OK; }
  if (!strcmp(key, "mul.algo")) { *value = g.opt_mul_algo; return KYB_OK; }
  if (!strcmp(key, "mul.ladder_waves")) { *value = g.opt_ladder_waves; return KYB_OK; }
  if (!strcmp(key, "mul_base.radix")) { *value = g.opt_base_radix; return KYB_OK; }
  if (!strcmp(key, "mul_base.block")) { *value = g.opt_base_block; return KYB_OK; }
  if (!strcmp(key, "finish.batched")) { *value = g.opt_finish; return KYB_OK; }
  if (!strcmp(key, "finish.min_items")) { *value = g.opt_finish_min; return KYB_OK; }
  if (!strcmp(key, "mul.grid_per_cu")) { *value = g.cus ? g.grid_mul / g.cus : 0; return KYB_OK; }
  return fail(KYB_E_BAD_ARG, "unknown option");
}

}  // extern "C"
