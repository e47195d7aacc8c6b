// Remaining kernels of the MI355X Ed25519 engine (one of the translation units mapped in launch.h).
//   k_finish            batched inversion + encode of projective staging records (ge.rs:112-122)
//   k_encode_batched    marshal_binary of extended points with the same shared inversion (point.rs:35-41)
//   k_add / k_equal / k_encode / k_decode   point.rs:179-241 / 35-51
//   k_poly_eval         share/poly.rs:457-469
//   k_poly_eval_part    the same for long polynomials at 10^3..10^4 evaluations: partial Horner chains, recombined by the ladder
#include <hip/hip_runtime.h>
#include "launch.h"
#include "ge_scalarmult.h"
#include "sc25519.h"
#include "device_batch_invert.h"
#include "diag_stamp.h"
using namespace kyb;
#include "device_tables.h"

// Batched finish: lane j owns items j, j+M, ..., j+(K-1)M (M = ceil(n/K)) and inverts the product of
// their Z's once (Montgomery's trick): 3(K-1) M + one inversion per K items instead of 254 S + 11 M
// per item.  A zero Z (only reachable from invalid extended inputs) is replaced by 1 in the product
// and gets the reference's own answer for it (0^(p-2) = 0 -> x = y = 0), so one bad item cannot
// disturb its K-1 neighbours.  Item i is read from record i * src_mul (src_mul = group length after a
// segmented sum, 1 otherwise).
template <int K>
__device__ __forceinline__ void finish_body(const uint4* __restrict__ proj, size_t stride, size_t n, uint8_t* __restrict__ out_enc, int32_t* __restrict__ out_ext, size_t src_mul) {
  KYB_SHORT_KERNEL_PRIORITY();
  const size_t M = (n + K - 1) / K;
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
  batch_invert<0, K>(unused_prefix, unused_inv, load, emit);
}
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_finish(const uint4* __restrict__ proj, size_t stride, size_t n, uint8_t* __restrict__ out_enc, int32_t* __restrict__ out_ext, size_t src_mul) {
  finish_body<FINISH_K>(proj, stride, n, out_enc, out_ext, src_mul);
}
// The same with 4 items per inversion, for launches of at most a wavefront per SIMD (the DKG-sized calls, where this kernel is one lane's
// chain of 600 divsteps plus 3 (K - 1) + 2 K products and nothing else runs beside it): half the products of the chain (profiles/r03/ab_finish_k.log
// measured -2..7 % per call for these sizes; full batches keep 8).
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_finish4(const uint4* __restrict__ proj, size_t stride, size_t n, uint8_t* __restrict__ out_enc, int32_t* __restrict__ out_ext, size_t src_mul) {
  finish_body<4>(proj, stride, n, out_enc, out_ext, src_mul);
}


// marshal_binary of n extended points (point.rs:35-41 -> ge.rs:112-122) with one field inversion per FINISH_K
// points: lane j owns points j, j+M, ... exactly as k_finish does, reading X, Y, Z straight from the 160-byte records.
template <int K>
__device__ __forceinline__ void encode_batched_body(const int32_t* __restrict__ pts_ext, size_t n, uint8_t* __restrict__ out_enc) {
  const size_t M = (n + K - 1) / K;
  const size_t j = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (j >= M) return;
  auto load_fe = [&](fe& h, size_t i, int which) {
    const uint4* p = reinterpret_cast<const uint4*>(pts_ext) + 10 * i;
    int32_t s[12];
    // limbs [10*which, 10*which + 10) of the record: quads (10*which)/4 .. cover them with three 16-byte loads
    const int first = (10 * which) >> 2, skip = (10 * which) & 3;
#pragma unroll
    for (int q = 0; q < 3; ++q) { const uint4 v = p[first + q]; s[4 * q] = (int32_t)v.x; s[4 * q + 1] = (int32_t)v.y; s[4 * q + 2] = (int32_t)v.z; s[4 * q + 3] = (int32_t)v.w; }
    int32_t t[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) t[k] = s[skip + k];
    fe_from_ref10(h, t);
  };
  auto load = [&](int t, fe& z) {
    const size_t i = j + (size_t)t * M;
    fe one;
    fe_one(one);
    if (i < n) load_fe(z, i, 2); else fe_one(z);
    fe_cmov(z, one, 1u - fe_is_nonzero(z));
  };
  auto emit = [&](int t, const fe& zinv) {
    const size_t i = j + (size_t)t * M;
    if (i >= n) return;
    fe z, zi, zero, X, Y, x, y;
    fe_zero(zero);
    load_fe(z, i, 2);
    fe_copy(zi, zinv);
    fe_cmov(zi, zero, 1u - fe_is_nonzero(z));          // Z == 0: the reference's 0^(p-2) = 0
    load_fe(X, i, 0);
    load_fe(Y, i, 1);
    fe_mul(x, X, zi);
    fe_mul(y, Y, zi);
    uint32_t w[8];
    fe_to_words(w, y);
    w[7] ^= fe_is_negative(x) << 31;
    store_words8(out_enc, i, w);
  };
  fe unused_prefix, unused_inv;
  fe_one(unused_prefix);
  batch_invert<0, K>(unused_prefix, unused_inv, load, emit);
}
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_encode_batched(const int32_t* __restrict__ pts_ext, size_t n, uint8_t* __restrict__ out_enc) {
  encode_batched_body<FINISH_K>(pts_ext, n, out_enc);
}
// four points per inversion for launches of at most a wavefront per SIMD (as k_finish4: a mid-size marshal_binary batch 0.069 -> 0.056 ms)
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_encode_batched4(const int32_t* __restrict__ pts_ext, size_t n, uint8_t* __restrict__ out_enc) {
  encode_batched_body<4>(pts_ext, n, out_enc);
}

// PubPoly::eval (poly.rs:457-469, shares :472-478) at n share indices: of one polynomial (per_poly == 0) or of
// polynomial i / per_poly for item i (a verifier checking the deals of many dealers at its own index)
template <bool SPLIT>
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_poly_eval(const int32_t* __restrict__ commits_ext, int t, const uint32_t* __restrict__ indices, size_t n, int nbits, size_t per_poly,
            uint8_t* __restrict__ out_enc, int32_t* __restrict__ out_ext, uint4* __restrict__ proj, size_t stride) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  const bool live = i < n;
  const size_t ii = live ? i : 0;
  const uint32_t x = indices[ii] + 1u;
  const size_t first = per_poly ? (ii / per_poly) * (size_t)t : 0;      // first commitment of this item's polynomial
  ge_p2 r;
  ge_poly_eval(r, [&](int j, ge_p3& c) { load_ext(c, commits_ext, first + (size_t)j); }, t, x, nbits);
  if (SPLIT) { if (live) store_proj(proj, stride, i, r.X, r.Y, r.Z); }
  else finish_point(r.X, r.Y, r.Z, out_enc, out_ext, ii, live);
}
// Long polynomials at a middling number of evaluations (about 10^3 .. 6x10^4: more than one wavefront each can serve, fewer than
// one lane each needs to fill the chip, whose t-step Horner chain would then run at one lane's latency).  The chain is cut as in
// k_poly_eval_seg:  P(x) = sum_s x^(s len) Q_s(x),  Q_s(x) = sum_{j < len} x^j C_{s len + j}.
// Lane (evaluation e, segment s) evaluates Q_s by Horner and leaves it as extended limbs — negated when the representative of the
// multiplier is negative — next to |x^(s len) mod 8L| (sc_pow_mod8L_signed: exact on points with small-order components); the
// variable-base ladder multiplies, k_pair_sum adds an evaluation's segs products up, k_finish encodes.  segs * len >= t and
// (segs - 1) * len < t: no segment is empty.
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_poly_eval_part(const int32_t* __restrict__ commits_ext, int t, const uint32_t* __restrict__ indices, size_t n, int nbits, size_t per_poly, int len, int segs,
                 int32_t* __restrict__ part_ext, uint8_t* __restrict__ part_sc) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  const bool live = i < n * (size_t)segs;
  const size_t ii = live ? i : 0;
  const size_t e = ii / (size_t)segs;
  const int sg = (int)(ii - e * (size_t)segs);
  const uint32_t x = indices[e] + 1u;
  const int lo = sg * len, cnt = t - lo < len ? t - lo : len;
  const size_t first = (per_poly ? (e / per_poly) * (size_t)t : 0) + (size_t)lo;
  ge_p3 v;
  ge_poly_eval_p3(v, [&](int j, ge_p3& c) { load_ext(c, commits_ext, first + (size_t)j); }, cnt, x, nbits);
  uint32_t mag[8], neg;
  sc_pow_mod8L_signed(mag, neg, x, (uint32_t)lo);
  fe nx, nt;
  fe_reduce_weak(v.X, v.X); fe_reduce_weak(v.T, v.T);
  fe_neg(nx, v.X); fe_neg(nt, v.T);
  fe_reduce_weak(nx, nx); fe_reduce_weak(nt, nt);
  fe_cmov(v.X, nx, neg); fe_cmov(v.T, nt, neg);
  if (live) {
    store_ext(part_ext, i, v.X, v.Y, v.Z, v.T);
    store_words8(part_sc, i, mag);
  }
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


// Benchmark diagnostic (kyb_diag_mad_peak): the integer-multiply roofline of THIS chip, measured in the run that quotes it.
// Every wavefront issues nothing but v_mad_u64_u32 — eight independent accumulator chains, so the 4-cycle issue of a
// quarter-rate instruction is never waiting on a result — and stamps its own lifetime (diag_stamp.h).  Two 1024-thread
// workgroups per CU = 8 wavefronts per SIMD.
constexpr int MAD_PEAK_CHAINS = launch::MAD_PEAK_CHAINS_HOST, MAD_PEAK_UNROLL = launch::MAD_PEAK_UNROLL_HOST;
// SGPR_CARRY: the (unused) carry-out goes to an SGPR pair instead of VCC — in a pure stream that form issues a few per cent faster; the
// roofline takes the faster of the two, whichever the product kernels use.
template <bool SGPR_CARRY>
__global__ void __launch_bounds__(1024)
k_diag_mad_peak(int iters, uint64_t* stamps, uint32_t* __restrict__ sink) {
  uint64_t acc[MAD_PEAK_CHAINS];
  const uint32_t a = threadIdx.x * 2654435761u + 12345u + blockIdx.x, b = threadIdx.x * 40503u + 977u;
#pragma unroll
  for (int c = 0; c < MAD_PEAK_CHAINS; ++c) acc[c] = a + c * 7919u;
  WaveClock::stamp(stamps, 0);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < MAD_PEAK_UNROLL; ++u) {
#pragma unroll
      for (int c = 0; c < MAD_PEAK_CHAINS; ++c) {
        if (SGPR_CARRY) { uint64_t cy; asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(acc[c]), "=&s"(cy) : "v"(a), "v"(b)); }
        else asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
      }
    }
  }
  WaveClock::stamp(stamps, 1);
  uint64_t s = 0;
#pragma unroll
  for (int c = 0; c < MAD_PEAK_CHAINS; ++c) s += acc[c];
  if (s == 0x1234567u) sink[0] = (uint32_t)s;      // keeps the chains alive; practically never true
}

namespace kyb { namespace launch {
hipError_t diag_mad_peak(hipStream_t st, int grid, int iters, uint64_t* stamps, uint32_t* sink, bool sgpr_carry) {
  if (sgpr_carry) hipLaunchKernelGGL((k_diag_mad_peak<true>), dim3(grid), dim3(1024), 0, st, iters, stamps, sink);
  else            hipLaunchKernelGGL((k_diag_mad_peak<false>), dim3(grid), dim3(1024), 0, st, iters, stamps, sink);
  return hipGetLastError();
}
static inline unsigned blocks_for(size_t n) { return (unsigned)((n + KYB_BLOCK - 1) / KYB_BLOCK); }
hipError_t finish(hipStream_t st, const uint4* proj, size_t stride, size_t n, uint8_t* oenc, int32_t* oext, size_t src_mul, bool four) {
  if (four) {
    const size_t M4 = (n + 3) / 4;
    hipLaunchKernelGGL(k_finish4, dim3(blocks_for(M4)), dim3(KYB_BLOCK), 0, st, proj, stride, n, oenc, oext, src_mul);
    return hipGetLastError();
  }
  const size_t M = (n + FINISH_K - 1) / FINISH_K;
  hipLaunchKernelGGL(k_finish, dim3(blocks_for(M)), dim3(KYB_BLOCK), 0, st, proj, stride, n, oenc, oext, src_mul);
  return hipGetLastError();
}
hipError_t encode_batched(hipStream_t st, const int32_t* pext, size_t n, uint8_t* oenc, bool four) {
  if (four) {
    hipLaunchKernelGGL(k_encode_batched4, dim3(blocks_for((n + 3) / 4)), dim3(KYB_BLOCK), 0, st, pext, n, oenc);
    return hipGetLastError();
  }
  const size_t M = (n + FINISH_K - 1) / FINISH_K;
  hipLaunchKernelGGL(k_encode_batched, dim3(blocks_for(M)), dim3(KYB_BLOCK), 0, st, pext, n, oenc);
  return hipGetLastError();
}
hipError_t add(hipStream_t st, const int32_t* a, const int32_t* b, size_t n, int32_t* out, int subtract) {
  hipLaunchKernelGGL(k_add, dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, a, b, n, out, subtract);
  return hipGetLastError();
}
hipError_t equal(hipStream_t st, const int32_t* a, const int32_t* b, size_t n, uint8_t* eq) {
  hipLaunchKernelGGL(k_equal, dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, a, b, n, eq);
  return hipGetLastError();
}
hipError_t encode(hipStream_t st, const int32_t* pext, size_t n, uint8_t* oenc) {
  hipLaunchKernelGGL(k_encode, dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, pext, n, oenc);
  return hipGetLastError();
}
hipError_t decode(hipStream_t st, const uint8_t* enc, size_t n, int32_t* out_ext, uint8_t* ok) {
  hipLaunchKernelGGL(k_decode, dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, enc, n, out_ext, ok);
  return hipGetLastError();
}
hipError_t poly_eval(bool split, hipStream_t st, const int32_t* commits, int t, const uint32_t* idx, size_t n, int nbits, size_t per_poly,
                     uint8_t* oenc, int32_t* oext, uint4* proj, size_t stride) {
  if (split) hipLaunchKernelGGL((k_poly_eval<true>), dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, commits, t, idx, n, nbits, per_poly, oenc, oext, proj, stride);
  else       hipLaunchKernelGGL((k_poly_eval<false>), dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, commits, t, idx, n, nbits, per_poly, oenc, oext, proj, stride);
  return hipGetLastError();
}
hipError_t poly_eval_part(hipStream_t st, const int32_t* commits, int t, const uint32_t* idx, size_t n, int nbits, size_t per_poly, int len, int segs,
                          int32_t* part_ext, uint8_t* part_sc) {
  hipLaunchKernelGGL(k_poly_eval_part, dim3(blocks_for(n * (size_t)segs)), dim3(KYB_BLOCK), 0, st, commits, t, idx, n, nbits, per_poly, len, segs, part_ext, part_sc);
  return hipGetLastError();
}
}}  // namespace kyb::launch
