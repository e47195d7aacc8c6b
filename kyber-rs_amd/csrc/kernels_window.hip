// Windowed variable-base kernel of the MI355X Ed25519 engine (one of the translation units, see launch.h).
//   k_mul   Point::mul(s, Some(P)) with the reference's own structure (ge.rs:508-568): per-lane table 1P..8P in an
//           L2/MALL-resident workspace [entry][quad][lane].  Superseded by the ladder (mul.algo=1, 1.33x faster,
//           profiles/r01/sweep_mul_algo.log); kept selectable (mul.algo=0) as an independent cross-check.
#include <hip/hip_runtime.h>
#include "launch.h"
#include "ge_scalarmult.h"
using namespace kyb;
#include "device_tables.h"

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

namespace kyb { namespace launch {
hipError_t mul_window(int masked, bool from_enc, bool split, int grid, hipStream_t st, const uint8_t* sc, const uint8_t* penc, const int32_t* pext, size_t n,
                      uint8_t* oenc, int32_t* oext, uint8_t* ok, uint4* ws, uint4* proj, size_t stride) {
#define KYB_L(M_, E_, S_) hipLaunchKernelGGL((k_mul<M_, E_, S_>), dim3(grid), dim3(KYB_BLOCK), 0, st, sc, penc, pext, n, oenc, oext, ok, ws, proj, stride)
  if (split) { if (masked == 0) { if (from_enc) KYB_L(0, true, true); else KYB_L(0, false, true); } else { if (from_enc) KYB_L(1, true, true); else KYB_L(1, false, true); } }
  else       { if (masked == 0) { if (from_enc) KYB_L(0, true, false); else KYB_L(0, false, false); } else { if (from_enc) KYB_L(1, true, false); else KYB_L(1, false, false); } }
#undef KYB_L
  return hipGetLastError();
}
}}  // namespace kyb::launch
