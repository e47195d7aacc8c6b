// Alternative fixed-base kernels of the MI355X Ed25519 engine (one of the translation units, see launch.h): the
// radix-32 and radix-16 tables that the radix-64 kernel superseded (kept selectable as cross-checks, mul_base.radix)
// and the fused signing kernel on the radix-16 table.
//   k_mul_base32   Point::mul(s, None)  ge.rs:442-486   52x16 affine table in LDS (106,496 B)
//   k_mul_base     the same                             64x8 affine table in LDS (65,536 B)
//   k_sign         fused schnorr::sign (schnorr_sig.rs:25-47)
#include <hip/hip_runtime.h>
#include "launch.h"
#include "schnorr.h"
using namespace kyb;
#include "device_tables.h"

// Fixed base, signed radix 32: one 1024-thread workgroup per CU shares the 106,496-byte table in LDS
// (4 waves per SIMD, <= 128 VGPRs); 52 mixed additions per item.
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


namespace kyb { namespace launch {
hipError_t mul_base32(bool split, int grid, hipStream_t st, const uint8_t* sc, size_t n, uint8_t* oenc, int32_t* oext, const uint4* img32,
                      uint4* proj, size_t stride, size_t offset) {
  if (split) hipLaunchKernelGGL((k_mul_base32<true>), dim3(grid), dim3(KYB_BLOCK32), 0, st, sc, n, oenc, oext, img32, proj, stride, offset);
  else       hipLaunchKernelGGL((k_mul_base32<false>), dim3(grid), dim3(KYB_BLOCK32), 0, st, sc, n, oenc, oext, img32, proj, stride, offset);
  return hipGetLastError();
}
hipError_t mul_base16(int mode, int block, bool split, int grid, hipStream_t st, const uint8_t* sc, size_t n, uint8_t* oenc, int32_t* oext,
                      const uint4* img, uint4* proj, size_t stride, size_t offset) {
#define KYB_L(M_, B_, S_) hipLaunchKernelGGL((k_mul_base<M_, B_, S_>), dim3(grid), dim3(B_), 0, st, sc, n, oenc, oext, img, proj, stride, offset)
  if (split) { if (mode == 0) { if (block == 512) KYB_L(0, 512, true); else KYB_L(0, 256, true); } else { if (block == 512) KYB_L(1, 512, true); else KYB_L(1, 256, true); } }
  else       { if (mode == 0) { if (block == 512) KYB_L(0, 512, false); else KYB_L(0, 256, false); } else { if (block == 512) KYB_L(1, 512, false); else KYB_L(1, 256, false); } }
#undef KYB_L
  return hipGetLastError();
}
hipError_t sign_fused(int mode, int grid, hipStream_t st, const uint8_t* x, const uint8_t* k, const uint8_t* msgs, const uint32_t* off, size_t n,
                      uint8_t* sig, const uint4* img) {
  if (mode == 0) hipLaunchKernelGGL((k_sign<0, KYB_BLOCK>), dim3(grid), dim3(KYB_BLOCK), 0, st, x, k, msgs, off, n, sig, img);
  else           hipLaunchKernelGGL((k_sign<1, KYB_BLOCK>), dim3(grid), dim3(KYB_BLOCK), 0, st, x, k, msgs, off, n, sig, img);
  return hipGetLastError();
}
}}  // namespace kyb::launch
