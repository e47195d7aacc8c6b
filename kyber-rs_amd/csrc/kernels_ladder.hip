// Variable-base kernels of the MI355X Ed25519 engine (one of the translation units mapped in launch.h).
//   k_decode_or_identity   unmarshal_binary of the operands (ge.rs:124-179), failed decodes -> neutral element
//   k_mont_prep            Montgomery images of the operands, one field inversion per FINISH_K items
//   k_mul_ladder           Point::mul(s, Some(P))  ge.rs:508-568   Montgomery ladder + y-recovery (ge_ladder.h)
//   k_decode_to_proj       unmarshal_binary into projective staging records, optionally transposed (kyb_sum_enc_batch)
//   k_pair_sum             one halving pass of the segmented sums behind kyb_lincomb_batch / kyb_sum_batch
//   k_ext_to_proj          extended limbs -> projective staging records
//   k_mul_ladder_pair_y[_dec] / k_ladder_recover   the two-lane ladder on the y of a wire encoding, the decode beside it (ge_ladder_pair.h)
//   k_verify_ladder_y / k_verify_recover_final      a DKG-sized verification as two launches (eddsa_sig.rs:159-212, schnorr_sig.rs:53-110; verify.h)
#include <hip/hip_runtime.h>
#include "launch.h"
#include "ge_scalarmult.h"
#include "ge_ladder.h"
#include "ge_ladder_pair.h"
#include "ge_ladder_quad.h"
#include "schnorr.h"
#include "verify.h"
#include "device_batch_invert.h"
#include "diag_stamp.h"
using namespace kyb;
#include "device_tables.h"
KYB_DEFINE_STAMP_SLOT()

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

// unmarshal_binary straight into projective staging records (kyb_sum_enc_batch): no extended-limb copy in between.
// rows > 0: the encodings form a rows x cols matrix (row-major) and the records its transpose, so that the sums over the
// COLUMNS of the source (coefficient g of every dealer's polynomial) become sums over contiguous records.
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_decode_to_proj(const uint8_t* __restrict__ enc, size_t n, uint4* __restrict__ proj, size_t stride, uint8_t* __restrict__ ok_out, size_t rows, size_t cols) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
  load_words8(w, enc, i);
  ge_p3 P, id;
  const uint32_t ok = ge_decode(P, w);
  ge_p3_0(id);
  fe_cmov(P.X, id.X, 1u - ok); fe_cmov(P.Y, id.Y, 1u - ok); fe_cmov(P.Z, id.Z, 1u - ok);
  size_t d = i;
  if (rows != 0) { const size_t rr = i / cols, cc = i - rr * cols; d = cc * rows + rr; }
  store_proj(proj, stride, d, P.X, P.Y, P.Z);
  if (ok_out != nullptr) ok_out[i] = (uint8_t)ok;
}

// ---- table-free variable base (ge_ladder.h) -------------------------------------------------------
// Montgomery images of the input points, one field inversion per FINISH_K items.  Output record of item
// i in the staging buffer: quads 0..4 = u[10] v[10], quad 5.x = flags (the ladder kernel later overwrites
// the same record with the projective result).
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_mont_prep(const int32_t* __restrict__ pts_ext, size_t n, uint4* __restrict__ proj, size_t stride, const uint8_t* __restrict__ scalars,
            uint32_t* __restrict__ top_or) {
  KYB_SHORT_KERNEL_PRIORITY();
  const size_t M = (n + FINISH_K - 1) / FINISH_K;
  const size_t j = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (j >= M) return;
  if (top_or != nullptr) {
    // On the way: the OR of the top four bits of this lane's scalars.  The word stays 0 iff every scalar of the launch is below 2^252, as
    // a value reduced mod L is (L = 2^252 + 2.8 x 10^37: all but 2^-127 of them); the ladder kernel then skips those four steps (1.6 % of
    // its time).  The test says nothing about a canonical scalar; for the unreduced inputs the reference also accepts (clamped keys,
    // quirk vectors) the word becomes non-zero and all 256 bits are walked.
    uint32_t top = 0;
    KYB_UNROLL for (int t = 0; t < FINISH_K; ++t) {
      const size_t i = j + (size_t)t * M;
      if (i < n) top |= reinterpret_cast<const uint32_t*>(scalars)[8 * i + 7] >> 28;
    }
    // one unconditional atomic per wavefront, by its first lane, with the wave-wide answer as DATA: no branch and no address depends on
    // a scalar (tools/ct_check.py); what the launch learns — "some scalar of it is not canonical" — is the documented exception
    const uint32_t any = __ballot(top != 0u) != 0ull ? 1u : 0u;
    if ((threadIdx.x & 63u) == 0u) atomicOr(top_or, any);
  }
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
k_mul_ladder(const uint8_t* __restrict__ scalars, size_t n, uint4* __restrict__ proj, size_t stride, size_t img_offset, size_t img_mod, int skip_bits_arg,
             const uint32_t* __restrict__ top_or, uint32_t* __restrict__ zero_next) {
  // kyb_diag_wave_stamps (diag_stamp.h): off unless a benchmark asked for the in-kernel clock
  KYB_STAMP_BEGIN();
  // top_or (written by k_mont_prep of this call): 0 iff every scalar of the launch is below 2^252 — the ladder then starts below the four
  // leading zeros.  zero_next: the word the NEXT call on this stream will collect into (the two alternate), cleared here.
  const int skip_bits = top_or != nullptr ? ((*top_or == 0u) ? 4 : 0) : skip_bits_arg;
  if (zero_next != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *zero_next = 0u;
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
  ge_scalarmult_ladder(r, a, m, skip_bits);
  store_proj(proj, stride, i, r.X, r.Y, r.Z);
  KYB_STAMP_END();
}

// The same multiplication with two lanes per item (ge_ladder_pair.h): for batches that leave SIMDs idle.  Lanes 2i and 2i+1 load the
// same scalar and the same extended point (item i's, or point i mod pts_mod of a shared set), build its projective Montgomery image,
// walk the ladder together, both recover the point, the even lane stores it.  No k_mont_prep in front (so no launch-wide canonical test:
// the step count is 256 - skip_bits for public skip_bits only).
__global__ void __launch_bounds__(KYB_BLOCK, 2)      // launched for at most one wavefront per SIMD (ladder.pair_max_items): registers are free, no spills
k_mul_ladder_pair(const uint8_t* __restrict__ scalars, size_t n, const int32_t* __restrict__ pts_ext, size_t pts_mod, uint4* __restrict__ proj, size_t stride, int skip_bits) {
  const size_t lane = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  const size_t i = lane >> 1;
  const uint32_t odd = threadIdx.x & 1u;
  if (i >= n) return;
  uint32_t a[8];
  load_words8(a, scalars, i);
  ge_p3 P;
  load_ext(P, pts_ext, pts_mod ? i % pts_mod : i);
  ge_p2 r;
  ge_scalarmult_ladder_pair(r, a, P, skip_bits, odd);
  if (odd == 0u) store_proj(proj, stride, i, r.X, r.Y, r.Z);
}

// Four lanes per item (ge_ladder_quad.h): lanes 4i .. 4i+3 walk the ladder of item i three products deep per step, all four recover the point, lane 4i
// stores it.  For launches of at most a wavefront per SIMD (ladder.quad_max_items).
__global__ void __launch_bounds__(KYB_BLOCK, 1)
k_mul_ladder_quad(const uint8_t* __restrict__ scalars, size_t n, const int32_t* __restrict__ pts_ext, size_t pts_mod, uint4* __restrict__ proj, size_t stride, int skip_bits) {
  const size_t lane = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  const size_t i = lane >> 2;
  const uint32_t q = threadIdx.x & 3u;
  if (i >= n) return;                                  // (n is a whole number of quads: the four lanes of an item leave together)
  uint32_t a[8];
  load_words8(a, scalars, i);
  ge_p3 P;
  load_ext(P, pts_ext, pts_mod ? i % pts_mod : i);
  ge_p2 r;
  ge_scalarmult_ladder_quad(r, a, P, skip_bits, q);
  if (q == 0u) store_proj(proj, stride, i, r.X, r.Y, r.Z);
}

// The same launch with the R half of a verification as further workgroups (keys given as POINTS: nothing to decode on the A side, but R's square
// root would otherwise be paid at the end, inside the encode-and-compare tail): workgroups [0, ladder_blocks) are k_mul_ladder_pair, the ones behind
// them k_verify_prep_r — checks and decode of R into record r_offset + i — on CUs of their own (ladder.y_only = 2).
template <int LG>      // (two or four lanes per item: ge_ladder_quad.h)
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_mul_ladder_pair_r(const uint8_t* __restrict__ scalars, size_t n, const int32_t* __restrict__ pts_ext, uint4* __restrict__ proj, size_t stride, int skip_bits,
                    unsigned ladder_blocks, const uint8_t* __restrict__ sigs, uint8_t* __restrict__ flags_r, size_t r_offset) {
  if (blockIdx.x < ladder_blocks) {
    __builtin_amdgcn_s_setprio(3);         // s*B shares these CUs from the side stream; the ladder is the critical path
    const size_t lane = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
    const size_t i = lane >> LG;
    const uint32_t sub = threadIdx.x & ((1u << LG) - 1u);
    if (i >= n) return;
    uint32_t a[8];
    load_words8(a, scalars, i);
    ge_p3 P;
    load_ext(P, pts_ext, i);
    ge_p2 r;
    ge_scalarmult_ladder_lanes<LG>(r, a, P, skip_bits, sub);
    if (sub == 0u) store_proj(proj, stride, i, r.X, r.Y, r.Z);
    return;
  }
  const size_t i = (size_t)(blockIdx.x - ladder_blocks) * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t sig[16];
  load_words8(sig, sigs, 2 * i);
  load_words8(sig + 8, sigs, 2 * i + 1);
  ge_p3 R;
  flags_r[i] = (uint8_t)verify_prep_r(R, sig);
  store_proj(proj, stride, r_offset + i, R.X, R.Y, R.Z);
}

// The two-lane ladder from the wire encoding (ge_ladder_pair.h, "from the WIRE encoding"): the ladder on (1 + y : 1 - y), its x-only state to a
// 160-byte record per item (x2, z2, x3, z3 as raw tight limbs); k_decode_or_identity runs beside it on a side stream; k_ladder_recover joins them.
__device__ __forceinline__ void store_state(uint4* base, size_t i, const fe& a, const fe& b, const fe& c, const fe& d) {
  uint32_t f[40];
#pragma unroll
  for (int k = 0; k < 10; ++k) { f[k] = a.v[k]; f[10 + k] = b.v[k]; f[20 + k] = c.v[k]; f[30 + k] = d.v[k]; }
  uint4* p = base + 10 * i;
#pragma unroll
  for (int q = 0; q < 10; ++q) p[q] = make_uint4(f[4 * q], f[4 * q + 1], f[4 * q + 2], f[4 * q + 3]);
}
__device__ __forceinline__ void load_state(fe& a, fe& b, fe& c, fe& d, const uint4* base, size_t i) {
  uint32_t f[40];
  const uint4* p = base + 10 * i;
#pragma unroll
  for (int q = 0; q < 10; ++q) { const uint4 v = p[q]; f[4 * q] = v.x; f[4 * q + 1] = v.y; f[4 * q + 2] = v.z; f[4 * q + 3] = v.w; }
#pragma unroll
  for (int k = 0; k < 10; ++k) { a.v[k] = f[k]; b.v[k] = f[10 + k]; c.v[k] = f[20 + k]; d.v[k] = f[30 + k]; }
}
template <int LG>
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_mul_ladder_pair_y(const uint8_t* __restrict__ scalars, size_t n, const uint8_t* __restrict__ pts_enc, uint4* __restrict__ state, int skip_bits) {
  // the decode kernel runs beside this one and its workgroups land on the same CUs (both grids start at CU 0): the ladder is the critical path, so its
  // wavefronts win the issue arbitration and the decode takes the slots a lone ladder wavefront leaves empty anyway (profiles/r04/mid_size_kernels.log)
  __builtin_amdgcn_s_setprio(3);
  const size_t lane = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  const size_t i = lane >> LG;
  const uint32_t sub = threadIdx.x & ((1u << LG) - 1u);
  if (i >= n) return;
  uint32_t a[8], w[8];
  load_words8(a, scalars, i);
  load_words8(w, pts_enc, i);
  fe x2, z2, x3, z3;
  mont_ladder_lanes_from_y<LG>(x2, z2, x3, z3, a, w, skip_bits, sub);
  if (sub == 0u) store_state(state, i, x2, z2, x3, z3);
}
// The two kernels above and k_decode_or_identity as ONE launch (ladder.y_only = 2): workgroups [0, ladder_blocks) walk the ladder, two lanes per item;
// the workgroups behind them decode the same encodings, one lane per item.  Workgroups of one launch are dealt out to the CUs in order, so the decoding
// ones land on CUs of their own instead of on the SIMDs the ladder occupies — which is where a second kernel on a side stream puts them (both grids start
// at the same CU; profiles/r04/side_cu_mask_probe.log) — and no stream has to be forked and joined.  The role depends on blockIdx only.
template <int LG>
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_mul_ladder_pair_y_dec(const uint8_t* __restrict__ scalars, size_t n, const uint8_t* __restrict__ pts_enc, uint4* __restrict__ state, int skip_bits,
                        unsigned ladder_blocks, int32_t* __restrict__ out_ext, uint8_t* __restrict__ ok_out) {
  if (blockIdx.x < ladder_blocks) {
    const size_t lane = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
    const size_t i = lane >> LG;
    const uint32_t sub = threadIdx.x & ((1u << LG) - 1u);
    if (i >= n) return;
    uint32_t a[8], w[8];
    load_words8(a, scalars, i);
    load_words8(w, pts_enc, i);
    fe x2, z2, x3, z3;
    mont_ladder_lanes_from_y<LG>(x2, z2, x3, z3, a, w, skip_bits, sub);
    if (sub == 0u) store_state(state, i, x2, z2, x3, z3);
    return;
  }
  const size_t i = (size_t)(blockIdx.x - ladder_blocks) * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
  load_words8(w, pts_enc, i);
  ge_p3 P, id;
  const uint32_t ok = ge_decode(P, w);
  ge_p3_0(id);
  fe_cmov(P.X, id.X, 1u - ok); fe_cmov(P.Y, id.Y, 1u - ok); fe_cmov(P.Z, id.Z, 1u - ok); fe_cmov(P.T, id.T, 1u - ok);
  store_ext(out_ext, i, P.X, P.Y, P.Z, P.T);
  if (ok_out != nullptr) ok_out[i] = (uint8_t)ok;
}
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_ladder_recover(const uint8_t* __restrict__ scalars, size_t n, const int32_t* __restrict__ pts_ext, const uint4* __restrict__ state, uint4* __restrict__ proj, size_t stride,
                 uint8_t* __restrict__ flags, const uint8_t* __restrict__ dec_ok) {
  KYB_SHORT_KERNEL_PRIORITY();
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  if (flags != nullptr) flags[i] = (uint8_t)(flags[i] | ((dec_ok[i] & 1u) << 2));      // verification: "the key decodes" joins the flags k_verify_hash wrote
  uint32_t a[8];
  load_words8(a, scalars, i);
  ge_p3 P;
  load_ext(P, pts_ext, i);
  fe x2, z2, x3, z3;
  load_state(x2, z2, x3, z3, state, i);
  ge_p2 r;
  ge_recover_from_state(r, a, P, x2, z2, x3, z3);
  store_proj(proj, stride, i, r.X, r.Y, r.Z);
}

// ---- verification of a DKG-sized batch from key BYTES in two launches (ladder.y_only = 2) ---------------------------------------------------------
// Everything that does not need the other multiplication, as ONE launch whose workgroups take one of three roles by blockIdx:
//   [0, ladder_blocks)                 two lanes per signature: s < L, the byte checks of A, h = SHA-512(R || A || msg) mod L (both lanes hash: both need
//                                      every bit of h), then the two-lane ladder on A's y with h -> the 160-byte state record
//   the next item_blocks workgroups    one lane per signature: decode A (the 252-squaring square root) -> extended limbs, "A decodes"
//   the last item_blocks workgroups    one lane per signature: checks and decode of R (verify_prep_r) -> flags_r, record 2n + i
// The decoding workgroups are dealt to CUs the ladder does not occupy (up to 128 items per CU in total), so neither square root is on the critical
// path and R arrives decoded: the equation is checked projectively by k_verify_recover_final, without the field inversion k_verify_final_enc
// pays to compare encodings.  s*B runs beside this launch on the side stream (k_sig_scalars gathers its scalars first).
template <int LG>
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_verify_ladder_y(const uint8_t* __restrict__ pubs, const uint8_t* __restrict__ sigs, const uint8_t* __restrict__ msgs, const uint32_t* __restrict__ msg_off, size_t n,
                  uint8_t* __restrict__ flags_a, uint8_t* __restrict__ flags_r, uint8_t* __restrict__ a_ok, uint8_t* __restrict__ hbuf, uint4* __restrict__ state,
                  int32_t* __restrict__ a_ext, uint4* __restrict__ proj, size_t stride, unsigned ladder_blocks, unsigned item_blocks) {
  if (blockIdx.x < ladder_blocks) {
    __builtin_amdgcn_s_setprio(3);         // s*B shares these CUs from the side stream; the ladder is the critical path
    const size_t lane = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
    const size_t i = lane >> LG;
    const uint32_t sub = threadIdx.x & ((1u << LG) - 1u);
    if (i >= n) return;
    uint32_t pub[8], ra[16], h[8];
    load_words8(pub, pubs, i);
    load_words8(ra, sigs, 2 * i);
    load_words8(ra + 8, sigs, 2 * i + 1);
    const uint32_t off = msg_off[i], len = msg_off[i + 1] - off;
    fe y;
    fe_from_words(y, pub);
    const uint32_t fl = sc_is_canonical_w(ra + 8) | (pt_is_canonical_w(pub) << 1) | (pt_has_small_order(y) << 3);
    for (int k = 0; k < 8; ++k) ra[8 + k] = pub[k];
    sha512_ctx c;
    sha512_init(c);
    sha512_words64(c, ra);
    sha512_bytes(c, msgs + off, len);
    uint32_t dig[16];
    sha512_final(dig, c);
    sc_reduce512(h, dig);
    if (sub == 0u) { flags_a[i] = (uint8_t)fl; store_words8(hbuf, i, h); }
    fe x2, z2, x3, z3;
    mont_ladder_lanes_from_y<LG>(x2, z2, x3, z3, h, pub, 3, sub);      // h < L < 2^253
    if (sub == 0u) store_state(state, i, x2, z2, x3, z3);
    return;
  }
  const unsigned b = blockIdx.x - ladder_blocks;
  const bool is_r = b >= item_blocks;                             // uniform in the workgroup
  const size_t i = (size_t)(is_r ? b - item_blocks : b) * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
  load_words8(w, is_r ? sigs : pubs, is_r ? 2 * i : i);
  ge_p3 P, id;
  const uint32_t ok = ge_decode(P, w);
  const uint32_t small = pt_has_small_order(P.Y);                 // as verify_prep_r_with: on what the decode left
  ge_p3_0(id);
  fe_cmov(P.X, id.X, 1u - ok); fe_cmov(P.Y, id.Y, 1u - ok); fe_cmov(P.Z, id.Z, 1u - ok); fe_cmov(P.T, id.T, 1u - ok);
  if (is_r) {
    flags_r[i] = (uint8_t)(pt_is_canonical_w(w) | (ok << 1) | (small << 2));
    store_proj(proj, stride, 2 * n + i, P.X, P.Y, P.Z);
  } else {
    a_ok[i] = (uint8_t)ok;
    store_ext(a_ext, i, P.X, P.Y, P.Z, P.T);
  }
}
// s of every signature as contiguous 32-byte records (the scalars of the fixed-base multiplication on the side stream)
__global__ void __launch_bounds__(KYB_BLOCK)
k_sig_scalars(const uint8_t* __restrict__ sigs, size_t n, uint8_t* __restrict__ sbuf) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t s[8];
  load_words8(s, sigs, 2 * i + 1);
  store_words8(sbuf, i, s);
}
// ... and the join: h*A from the ladder's state and the decoded key (ge_recover_from_state), then the equation R + h*A == s*B on projective
// coordinates and the status (verify_final, verify_status) — k_ladder_recover and k_verify_final in one pass over the item
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_verify_recover_final(const uint8_t* __restrict__ hbuf, size_t n, const int32_t* __restrict__ a_ext, const uint4* __restrict__ state, const uint4* __restrict__ proj, size_t stride,
                       const uint8_t* __restrict__ flags_a, const uint8_t* __restrict__ a_ok, const uint8_t* __restrict__ flags_r, int flavor, uint8_t* __restrict__ status,
                       kyb::launch::DoneFlag df) {
  KYB_SHORT_KERNEL_PRIORITY();
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t h[8];
  load_words8(h, hbuf, i);
  ge_p3 A;
  load_ext(A, a_ext, i);
  fe x2, z2, x3, z3;
  load_state(x2, z2, x3, z3, state, i);
  ge_p2 hA, sB;
  ge_recover_from_state(hA, h, A, x2, z2, x3, z3);
  fe RX, RY;
  load_proj_xy(sB.X, sB.Y, proj, stride, n + i);       load_proj_z(sB.Z, proj, stride, n + i);
  load_proj_xy(RX, RY, proj, stride, 2 * n + i);
  const uint32_t eq = verify_final(RX, RY, hA, sB);
  const uint32_t st = verify_status((uint32_t)flags_a[i] | (((uint32_t)a_ok[i] & 1u) << 2), flags_r[i], flavor);
  status[i] = (st == 0 && !eq) ? (uint8_t)9 : (uint8_t)st;
  signal_done(df);
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

// extended limbs -> projective staging records (input of the k_pair_sum passes of kyb_sum_batch)
// rows > 0: the points form a rows x cols matrix (row-major) and the records its transpose (as k_decode_to_proj)
__global__ void __launch_bounds__(KYB_BLOCK)
k_ext_to_proj(const int32_t* __restrict__ pts_ext, size_t n, uint4* __restrict__ proj, size_t stride, size_t rows, size_t cols) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  ge_p3 P;
  load_ext(P, pts_ext, i);
  size_t d = i;
  if (rows != 0) { const size_t rr = i / cols, cc = i - rr * cols; d = cc * rows + rr; }
  store_proj(proj, stride, d, P.X, P.Y, P.Z);
}


namespace kyb { namespace launch {
static inline unsigned blocks_for(size_t n) { return (unsigned)((n + KYB_BLOCK - 1) / KYB_BLOCK); }
hipError_t decode_or_identity(hipStream_t st, const uint8_t* enc, size_t n, int32_t* out_ext, uint8_t* ok) {
  hipLaunchKernelGGL(k_decode_or_identity, dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, enc, n, out_ext, ok);
  return hipGetLastError();
}
hipError_t decode_to_proj(hipStream_t st, const uint8_t* enc, size_t n, uint4* proj, size_t stride, uint8_t* ok, size_t rows, size_t cols) {
  hipLaunchKernelGGL(k_decode_to_proj, dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, enc, n, proj, stride, ok, rows, cols);
  return hipGetLastError();
}
hipError_t mont_prep(hipStream_t st, const int32_t* pext, size_t n, uint4* proj, size_t stride, const uint8_t* scalars, uint32_t* top_or) {
  const size_t M = (n + FINISH_K - 1) / FINISH_K;
  hipLaunchKernelGGL(k_mont_prep, dim3(blocks_for(M)), dim3(KYB_BLOCK), 0, st, pext, n, proj, stride, scalars, top_or);
  return hipGetLastError();
}
hipError_t mul_ladder(int waves, hipStream_t st, const uint8_t* sc, size_t n, uint4* proj, size_t stride, size_t img_offset, size_t img_mod, int skip_bits,
                      const uint32_t* top_or, uint32_t* zero_next) {
#ifdef KYB_CROSSCHECK      // mul.ladder_waves: the 4- and 2-wavefronts-per-SIMD register budgets (both slower: profiles/r02) exist in the cross-check build only
  if (waves >= 4)      { hipLaunchKernelGGL((k_mul_ladder<4>), dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, sc, n, proj, stride, img_offset, img_mod, skip_bits, top_or, zero_next); return hipGetLastError(); }
  else if (waves == 2) { hipLaunchKernelGGL((k_mul_ladder<2>), dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, sc, n, proj, stride, img_offset, img_mod, skip_bits, top_or, zero_next); return hipGetLastError(); }
#endif
  (void)waves;
  hipLaunchKernelGGL((k_mul_ladder<3>), dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, sc, n, proj, stride, img_offset, img_mod, skip_bits, top_or, zero_next);
  return hipGetLastError();
}
hipError_t mul_ladder_pair(hipStream_t st, const uint8_t* sc, size_t n, const int32_t* pext, size_t pts_mod, uint4* proj, size_t stride, int skip_bits) {
  hipLaunchKernelGGL(k_mul_ladder_pair, dim3(blocks_for(2 * n)), dim3(KYB_BLOCK), 0, st, sc, n, pext, pts_mod, proj, stride, skip_bits);
  return hipGetLastError();
}
hipError_t mul_ladder_quad(hipStream_t st, const uint8_t* sc, size_t n, const int32_t* pext, size_t pts_mod, uint4* proj, size_t stride, int skip_bits) {
  hipLaunchKernelGGL(k_mul_ladder_quad, dim3(blocks_for(4 * n)), dim3(KYB_BLOCK), 0, st, sc, n, pext, pts_mod, proj, stride, skip_bits);
  return hipGetLastError();
}
hipError_t mul_ladder_pair_r(hipStream_t st, const uint8_t* sc, size_t n, const int32_t* pext, uint4* proj, size_t stride, int skip_bits, const uint8_t* sigs, uint8_t* flags_r, size_t r_offset,
                             int lanes) {
  const unsigned lb = blocks_for((size_t)lanes * n);
  if (lanes == 4) hipLaunchKernelGGL(k_mul_ladder_pair_r<2>, dim3(lb + blocks_for(n)), dim3(KYB_BLOCK), 0, st, sc, n, pext, proj, stride, skip_bits, lb, sigs, flags_r, r_offset);
  else hipLaunchKernelGGL(k_mul_ladder_pair_r<1>, dim3(lb + blocks_for(n)), dim3(KYB_BLOCK), 0, st, sc, n, pext, proj, stride, skip_bits, lb, sigs, flags_r, r_offset);
  return hipGetLastError();
}
hipError_t mul_ladder_pair_y(hipStream_t st, const uint8_t* sc, size_t n, const uint8_t* penc, uint4* state, int skip_bits, int lanes) {
  if (lanes == 4) hipLaunchKernelGGL(k_mul_ladder_pair_y<2>, dim3(blocks_for(4 * n)), dim3(KYB_BLOCK), 0, st, sc, n, penc, state, skip_bits);
  else hipLaunchKernelGGL(k_mul_ladder_pair_y<1>, dim3(blocks_for(2 * n)), dim3(KYB_BLOCK), 0, st, sc, n, penc, state, skip_bits);
  return hipGetLastError();
}
hipError_t mul_ladder_pair_y_dec(hipStream_t st, const uint8_t* sc, size_t n, const uint8_t* penc, uint4* state, int skip_bits, int32_t* out_ext, uint8_t* ok, int lanes) {
  const unsigned lb = blocks_for((size_t)lanes * n);
  if (lanes == 4) hipLaunchKernelGGL(k_mul_ladder_pair_y_dec<2>, dim3(lb + blocks_for(n)), dim3(KYB_BLOCK), 0, st, sc, n, penc, state, skip_bits, lb, out_ext, ok);
  else hipLaunchKernelGGL(k_mul_ladder_pair_y_dec<1>, dim3(lb + blocks_for(n)), dim3(KYB_BLOCK), 0, st, sc, n, penc, state, skip_bits, lb, out_ext, ok);
  return hipGetLastError();
}
hipError_t ladder_recover(hipStream_t st, const uint8_t* sc, size_t n, const int32_t* pext, const uint4* state, uint4* proj, size_t stride, uint8_t* flags, const uint8_t* dec_ok) {
  hipLaunchKernelGGL(k_ladder_recover, dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, sc, n, pext, state, proj, stride, flags, dec_ok);
  return hipGetLastError();
}
hipError_t verify_ladder_y(hipStream_t st, const uint8_t* pubs, const uint8_t* sigs, const uint8_t* msgs, const uint32_t* off, size_t n, uint8_t* flags_a, uint8_t* flags_r,
                           uint8_t* a_ok, uint8_t* hbuf, uint4* state, int32_t* a_ext, uint4* proj, size_t stride, int lanes) {
  const unsigned lb = blocks_for((size_t)lanes * n), ib = blocks_for(n);
  if (lanes == 4) hipLaunchKernelGGL(k_verify_ladder_y<2>, dim3(lb + 2 * ib), dim3(KYB_BLOCK), 0, st, pubs, sigs, msgs, off, n, flags_a, flags_r, a_ok, hbuf, state, a_ext, proj, stride, lb, ib);
  else hipLaunchKernelGGL(k_verify_ladder_y<1>, dim3(lb + 2 * ib), dim3(KYB_BLOCK), 0, st, pubs, sigs, msgs, off, n, flags_a, flags_r, a_ok, hbuf, state, a_ext, proj, stride, lb, ib);
  return hipGetLastError();
}
hipError_t sig_scalars(hipStream_t st, const uint8_t* sigs, size_t n, uint8_t* sbuf) {
  hipLaunchKernelGGL(k_sig_scalars, dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, sigs, n, sbuf);
  return hipGetLastError();
}
hipError_t verify_recover_final(hipStream_t st, const uint8_t* hbuf, size_t n, const int32_t* a_ext, const uint4* state, const uint4* proj, size_t stride, const uint8_t* flags_a,
                                const uint8_t* a_ok, const uint8_t* flags_r, int flavor, uint8_t* status, DoneFlag df) {
  hipLaunchKernelGGL(k_verify_recover_final, dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, hbuf, n, a_ext, state, proj, stride, flags_a, a_ok, flags_r, flavor, status, df);
  return hipGetLastError();
}
hipError_t pair_sum(hipStream_t st, uint4* proj, size_t stride, size_t m, size_t gstride, size_t len, size_t half) {
  const size_t lanes = m * (len - half);
  hipLaunchKernelGGL(k_pair_sum, dim3(blocks_for(lanes)), dim3(KYB_BLOCK), 0, st, proj, stride, m, gstride, len, half);
  return hipGetLastError();
}
hipError_t ext_to_proj(hipStream_t st, const int32_t* pext, size_t n, uint4* proj, size_t stride, size_t rows, size_t cols) {
  hipLaunchKernelGGL(k_ext_to_proj, dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, pext, n, proj, stride, rows, cols);
  return hipGetLastError();
}
hipError_t diag_stamps_ladder(uint64_t* buf) { return kyb_set_stamp_slot(buf); }
}}  // namespace kyb::launch
