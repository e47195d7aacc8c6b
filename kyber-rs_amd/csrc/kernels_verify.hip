// SHA-512 users of the MI355X Ed25519 engine (one of the translation units mapped in launch.h).
//   k_verify_prep / k_verify_prep_r / k_verify_final   eddsa_sig.rs:159-212, schnorr_sig.rs:53-110 (verify.h)
//   k_verify_diff / k_verify_final_enc / k_verify_fixup   the same equation checked on encodings: R is decoded only when it fails
//   k_sign_hash                                        last stage of the split signing path (schnorr_sig.rs:25-47)
//   k_eddsa_prep                                       key expansion + deterministic nonce (curve.rs:74-87, eddsa_sig.rs:120-131)
#include <hip/hip_runtime.h>
#include "launch.h"
#include "schnorr.h"
// (round 1 routed the point decompression through one out-of-line copy because a single kernel decoded both A and R; with
// the two halves in kernels of their own each decodes once, inline — the call's stack frame was the kernels' 208 / 240 B
// of scratch and the reason for their 248 VGPRs)
#include "verify.h"
#include "device_batch_invert.h"
using namespace kyb;
#include "device_tables.h"

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

// split signing, last stage: r_enc[i] = enc(R_i), a_enc[i] = enc(A_i) (from the fixed-base launches + k_finish,
// or the caller's stored public keys); h = SHA-512(R || A || msg) mod L, s = k + x h mod L.
__global__ void __launch_bounds__(KYB_BLOCK)
k_sign_hash(const uint8_t* __restrict__ x, const uint8_t* __restrict__ k, const uint8_t* __restrict__ msgs,
            const uint32_t* __restrict__ msg_off, size_t n, const uint8_t* __restrict__ r_enc, const uint8_t* __restrict__ a_enc,
            uint8_t* __restrict__ sig, kyb::launch::DoneFlag df) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t wx[8], wk[8], ra[16];
  load_words8(wx, x, i);
  load_words8(wk, k, i);
  load_words8(ra, r_enc, i);
  load_words8(ra + 8, a_enc, i);
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
  signal_done(df);
}

// PriPoly::eval / shares (share/poly.rs:133-152): out[g][i] = sum_j coeffs[g][j] * x_i^j mod L, x_i = indices[i] + 1 — a dealer's n private
// shares of its secret polynomial (n x t scalar multiply-adds on one core in the reference).  The coefficients are SECRET: everything
// below is sc_muladd (the signing kernel's arithmetic: fixed instruction stream, no address or branch from the operands) under loop bounds
// that depend on t, the segment length and the public indices only.
// Stage 1: lane (g, i, s) evaluates segment s of the chain, Q_s(x) = sum_{j < len} c[s*len + j] x^j, and multiplies it by x^(s*len)
// (square-and-multiply over the 24 bits of the public exponent); stage 2 adds the S partial values of an evaluation and clears them.
__global__ void __launch_bounds__(KYB_BLOCK)
k_pripoly_eval_part(const uint8_t* __restrict__ coeffs, size_t m, size_t t, const uint32_t* __restrict__ indices, size_t k, uint32_t segs, uint32_t len,
                    uint8_t* __restrict__ part) {
  const size_t id = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (id >= m * k * segs) return;
  const size_t s = id % segs, gi = id / segs, i = gi % k, g = gi / k;
  const size_t lo = s * (size_t)len, hi = (lo + len < t) ? lo + len : t;
  uint32_t x[8], v[8], r[8];
  const uint64_t xv = (uint64_t)indices[i] + 1u;
  x[0] = (uint32_t)xv; x[1] = (uint32_t)(xv >> 32);
  KYB_UNROLL for (int q = 2; q < 8; ++q) x[q] = 0u;
  KYB_UNROLL for (int q = 0; q < 8; ++q) v[q] = 0u;
#pragma unroll 1
  for (size_t j = hi; j-- > lo;) {
    uint32_t c[8];
    load_words8(c, coeffs, g * t + j);
    sc_muladd(r, v, x, c);
    KYB_UNROLL for (int q = 0; q < 8; ++q) v[q] = r[q];
  }
  if (segs > 1u) {                                   // x^(s * len) mod L, exponent below 2^24 (t <= 2^24)
    const uint32_t zero[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    uint32_t pw[8] = {1u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
    const uint32_t e = (uint32_t)lo;
#pragma unroll 1
    for (int b = 23; b >= 0; --b) {
      uint32_t sq[8], sx[8];
      sc_muladd(sq, pw, pw, zero);
      sc_muladd(sx, sq, x, zero);
      const uint32_t bit = (e >> b) & 1u;
      KYB_UNROLL for (int q = 0; q < 8; ++q) pw[q] = bit ? sx[q] : sq[q];
    }
    sc_muladd(r, v, pw, zero);
    KYB_UNROLL for (int q = 0; q < 8; ++q) v[q] = r[q];
  }
  store_words8(part, id, v);
}
__global__ void __launch_bounds__(KYB_BLOCK)
k_pripoly_eval_sum(uint8_t* __restrict__ part, size_t n, uint32_t segs, uint8_t* __restrict__ out) {
  const size_t gi = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (gi >= n) return;
  const uint32_t one[8] = {1u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, zero[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
  uint32_t acc[8], r[8];
  load_words8(acc, part, gi * segs);
  store_words8(part, gi * segs, zero);
#pragma unroll 1
  for (uint32_t s = 1; s < segs; ++s) {
    uint32_t p[8];
    load_words8(p, part, gi * segs + s);
    store_words8(part, gi * segs + s, zero);        // partial values of a secret polynomial do not stay in the scratch buffer
    sc_muladd(r, p, one, acc);
    KYB_UNROLL for (int q = 0; q < 8; ++q) acc[q] = r[q];
  }
  store_words8(out, gi, acc);
}

// verification, A half: s < L, checks and decode of the public key, h = SHA-512(R || A || msg) mod L.
// Writes h and s as contiguous 32-byte records and A in reference limbs (inputs of the two multiplications).
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_verify_prep(const uint8_t* __restrict__ pubs, const uint8_t* __restrict__ sigs, const uint8_t* __restrict__ msgs,
              const uint32_t* __restrict__ msg_off, size_t n, uint8_t* __restrict__ flags_a,
              uint8_t* __restrict__ hbuf, uint8_t* __restrict__ sbuf, int32_t* __restrict__ a_ext) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t pub[8], sig[16], h[8];
  load_words8(pub, pubs, i);
  load_words8(sig, sigs, 2 * i);
  load_words8(sig + 8, sigs, 2 * i + 1);
  const uint32_t off = msg_off[i], len = msg_off[i + 1] - off;
  ge_p3 A;
  flags_a[i] = (uint8_t)verify_prep_a(h, A, pub, sig, msgs + off, len);
  store_words8(hbuf, i, h);
  store_words8(sbuf, i, sig + 8);
  store_ext(a_ext, i, A.X, A.Y, A.Z, A.T);
}
// The A half WITHOUT the decode (DKG-sized batches, round 4): everything the bytes give — s < L, is_canonical, has_small_order (it looks at y mod p
// only, verify_prep_a_point_with), the challenge h — so that the two-lane ladder can start on the y of the key at once (ge_ladder_pair.h, "from the
// WIRE encoding") while k_decode_or_identity looks for x on the side stream; k_ladder_recover adds bit 2 (A decodes) to the flags when it joins them.
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_verify_hash(const uint8_t* __restrict__ pubs, const uint8_t* __restrict__ sigs, const uint8_t* __restrict__ msgs, const uint32_t* __restrict__ msg_off, size_t n,
              uint8_t* __restrict__ flags_a, uint8_t* __restrict__ hbuf, uint8_t* __restrict__ sbuf) {
  KYB_SHORT_KERNEL_PRIORITY();
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t pub[8], sig[16], h[8];
  load_words8(pub, pubs, i);
  load_words8(sig, sigs, 2 * i);
  load_words8(sig + 8, sigs, 2 * i + 1);
  const uint32_t off = msg_off[i], len = msg_off[i + 1] - off;
  fe y;
  fe_from_words(y, pub);
  const uint32_t fl = sc_is_canonical_w(sig + 8) | (pt_is_canonical_w(pub) << 1) | (pt_has_small_order(y) << 3);
  uint32_t ra[16];
  for (int k = 0; k < 8; ++k) { ra[k] = sig[k]; ra[8 + k] = pub[k]; }
  sha512_ctx c;
  sha512_init(c);
  sha512_words64(c, ra);
  sha512_bytes(c, msgs + off, len);
  uint32_t dig[16];
  sha512_final(dig, c);
  sc_reduce512(h, dig);
  flags_a[i] = (uint8_t)fl;
  store_words8(hbuf, i, h);
  store_words8(sbuf, i, sig + 8);
}
// the same with the public keys given as points (schnorr::verify / eddsa::verify, verify.h): pub_enc = their marshal_binary, made by
// k_encode_batched in front of this kernel; no square root unless the limbs are not a point of the curve
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_verify_prep_pts(const uint8_t* __restrict__ pub_enc, const int32_t* __restrict__ pubs_ext, const uint8_t* __restrict__ sigs, const uint8_t* __restrict__ msgs,
                  const uint32_t* __restrict__ msg_off, size_t n, uint8_t* __restrict__ flags_a,
                  uint8_t* __restrict__ hbuf, uint8_t* __restrict__ sbuf, int32_t* __restrict__ a_ext) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t pub[8], sig[16], h[8];
  load_words8(pub, pub_enc, i);
  load_words8(sig, sigs, 2 * i);
  load_words8(sig + 8, sigs, 2 * i + 1);
  const uint32_t off = msg_off[i], len = msg_off[i + 1] - off;
  ge_p3 P, A;
  load_ext(P, pubs_ext, i);
  flags_a[i] = (uint8_t)verify_prep_a_point_with(h, A, P, pub, sig, msgs + off, len, ge_decode_fn());
  store_words8(hbuf, i, h);
  store_words8(sbuf, i, sig + 8);
  store_ext(a_ext, i, A.X, A.Y, A.Z, A.T);
}
// verification, R half: checks and decode of R into the projective staging buffer at [proj_offset, proj_offset + n)
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_verify_prep_r(const uint8_t* __restrict__ sigs, size_t n, uint8_t* __restrict__ flags_r, uint4* __restrict__ proj, size_t stride, size_t proj_offset) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t sig[16];
  load_words8(sig, sigs, 2 * i);
  load_words8(sig + 8, sigs, 2 * i + 1);
  ge_p3 R;
  flags_r[i] = (uint8_t)verify_prep_r(R, sig);
  store_proj(proj, stride, proj_offset + i, R.X, R.Y, R.Z);
}
// verification, last stage: hA at proj[i], sB at proj[n + i], R at proj[2n + i]; status = first failing check, else the equation
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_verify_final(const uint4* __restrict__ proj, size_t stride, size_t n, const uint8_t* __restrict__ flags_a, const uint8_t* __restrict__ flags_r,
               int flavor, uint8_t* __restrict__ status, kyb::launch::DoneFlag df) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  ge_p2 hA, sB;
  fe RX, RY;
  load_proj_xy(hA.X, hA.Y, proj, stride, i);           load_proj_z(hA.Z, proj, stride, i);
  load_proj_xy(sB.X, sB.Y, proj, stride, n + i);       load_proj_z(sB.Z, proj, stride, n + i);
  load_proj_xy(RX, RY, proj, stride, 2 * n + i);
  const uint32_t eq = verify_final(RX, RY, hA, sB);
  const uint32_t st = verify_status(flags_a[i], flags_r[i], flavor);
  status[i] = (st == 0 && !eq) ? (uint8_t)9 : (uint8_t)st;
  signal_done(df);
}

// ---- the equation without decoding R (large batches) ----------------------------------------------------------------------------
// R + h A == s B holds iff the encoding of D = s B - h A is the 32 bytes of R: in both check orders a non-canonical R is rejected
// before the equation is looked at, and the encoding of a curve point decodes.  So the 252-squaring square root of R's
// decompression is only needed for signatures whose bytes do NOT match — to tell "R is not a point" (status 4) from "the equation
// fails" (9) — and k_verify_fixup does it for exactly those; k_verify_prep_r (0.8 ms per 2^20) drops out of the common path.
// D goes through the projective staging buffer: k_verify_diff overwrites record i (h A) with D, k_verify_final_enc encodes with one
// field inversion per FINISH_K items (k_finish's scheme) and compares.
constexpr uint8_t KYB_VERIFY_TODO = 0xff;
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_verify_diff(uint4* __restrict__ proj, size_t stride, size_t n) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  ge_p2 hA, sB, D;
  load_proj_xy(hA.X, hA.Y, proj, stride, i);           load_proj_z(hA.Z, proj, stride, i);
  load_proj_xy(sB.X, sB.Y, proj, stride, n + i);       load_proj_z(sB.Z, proj, stride, n + i);
  ge_p3 H, S;
  ge_p2_to_p3(H, hA);
  ge_p2_to_p3(S, sB);
  ge_cached c;
  ge_p3_to_cached(c, H);
  ge_cached_cneg(c, 1u);
  ge_p1p1 t;
  ge_add(t, S, c);
  ge_p1p1_to_p2(D, t);
  store_proj(proj, stride, i, D.X, D.Y, D.Z);
}
template <int K>
__device__ __forceinline__ void verify_final_enc_body(const uint4* __restrict__ proj, size_t stride, size_t n, const uint8_t* __restrict__ sigs, const uint8_t* __restrict__ flags_a,
                                                      int flavor, uint8_t* __restrict__ status) {
  const size_t M = (n + K - 1) / K;
  const size_t j = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (j >= M) return;
  auto load = [&](int t, fe& z) {
    const size_t i = j + (size_t)t * M;
    fe one;
    fe_one(one);
    if (i < n) load_proj_z(z, proj, stride, i); else fe_one(z);
    fe_cmov(z, one, 1u - fe_is_nonzero(z));
  };
  auto emit = [&](int t, const fe& zinv) {
    const size_t i = j + (size_t)t * M;
    if (i >= n) return;
    fe z, zi, zero, X, Y, x, y;
    fe_zero(zero);
    load_proj_z(z, proj, stride, i);
    fe_copy(zi, zinv);
    fe_cmov(zi, zero, 1u - fe_is_nonzero(z));
    load_proj_xy(X, Y, proj, stride, i);
    fe_mul(x, X, zi);
    fe_mul(y, Y, zi);
    uint32_t w[8], r[8];
    fe_to_words(w, y);
    w[7] ^= fe_is_negative(x) << 31;
    load_words8(r, sigs, 2 * i);
    uint32_t diff = 0;
    for (int q = 0; q < 8; ++q) diff |= w[q] ^ r[q];
    if (diff != 0) { status[i] = KYB_VERIFY_TODO; return; }
    fe RY;
    fe_from_words(RY, r);
    const uint32_t fr = pt_is_canonical_w(r) | (1u << 1) | (pt_has_small_order(RY) << 2);      // R decodes: it is the encoding of D
    status[i] = (uint8_t)verify_status(flags_a[i], fr, flavor);
  };
  fe unused_prefix, unused_inv;
  fe_one(unused_prefix);
  batch_invert<0, K>(unused_prefix, unused_inv, load, emit);
}
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_verify_final_enc(const uint4* __restrict__ proj, size_t stride, size_t n, const uint8_t* __restrict__ sigs, const uint8_t* __restrict__ flags_a,
                   int flavor, uint8_t* __restrict__ status) {
  verify_final_enc_body<FINISH_K>(proj, stride, n, sigs, flags_a, flavor, status);
}
// four items per inversion for launches of at most a wavefront per SIMD (as k_finish4)
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_verify_final_enc4(const uint4* __restrict__ proj, size_t stride, size_t n, const uint8_t* __restrict__ sigs, const uint8_t* __restrict__ flags_a,
                    int flavor, uint8_t* __restrict__ status) {
  verify_final_enc_body<4>(proj, stride, n, sigs, flags_a, flavor, status);
}
// the signatures whose R bytes are not the encoding of s B - h A: the reference's first failing check, else 9 (equation)
__global__ void __launch_bounds__(KYB_BLOCK, 2)
k_verify_fixup(const uint8_t* __restrict__ sigs, size_t n, const uint8_t* __restrict__ flags_a, int flavor, uint8_t* __restrict__ status, kyb::launch::DoneFlag df) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  if (status[i] == KYB_VERIFY_TODO) {
    uint32_t sig[16];
    load_words8(sig, sigs, 2 * i);
    load_words8(sig + 8, sigs, 2 * i + 1);
    ge_p3 R;
    const uint32_t st = verify_status(flags_a[i], verify_prep_r(R, sig), flavor);
    status[i] = st == 0 ? (uint8_t)9 : (uint8_t)st;
  }
  signal_done(df);
}

// PointCanCheckCanonicalAndSmallOrder for n encodings (group.rs:71-78, point.rs:286-337), no multiplication: bit 0 = is_canonical(bytes) as the
// reference evaluates it (pt_is_canonical_w), bit 1 = has_small_order() of the point the bytes decode to.  has_small_order compares the point's
// canonical re-encoding, sign bit masked, with the five WEAK_KEYS, i.e. looks at y mod p only — which the bytes give without a square root; all five
// y values do decode, so for bytes that decode to no point the bit is 0.
__global__ void __launch_bounds__(KYB_BLOCK)
k_point_checks(const uint8_t* __restrict__ enc, size_t n, uint8_t* __restrict__ flags) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  uint32_t w[8];
  load_words8(w, enc, i);
  fe y;
  fe_from_words(y, w);                       // bit 255 ignored, y >= p accepted (fe_from_bytes); pt_has_small_order canonicalises
  flags[i] = (uint8_t)(pt_is_canonical_w(w) | (pt_has_small_order(y) << 1));
}

namespace kyb { namespace launch {
hipError_t point_checks(hipStream_t st, const uint8_t* enc, size_t n, uint8_t* flags) {
  hipLaunchKernelGGL(k_point_checks, dim3((unsigned)((n + KYB_BLOCK - 1) / KYB_BLOCK)), dim3(KYB_BLOCK), 0, st, enc, n, flags);
  return hipGetLastError();
}
static inline unsigned blocks_for(size_t n) { return (unsigned)((n + KYB_BLOCK - 1) / KYB_BLOCK); }
hipError_t verify_prep(hipStream_t st, const uint8_t* pubs, const uint8_t* sigs, const uint8_t* msgs, const uint32_t* off, size_t n,
                       uint8_t* flags_a, uint8_t* hbuf, uint8_t* sbuf, int32_t* a_ext) {
  hipLaunchKernelGGL(k_verify_prep, dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, pubs, sigs, msgs, off, n, flags_a, hbuf, sbuf, a_ext);
  return hipGetLastError();
}
hipError_t verify_hash(hipStream_t st, const uint8_t* pubs, const uint8_t* sigs, const uint8_t* msgs, const uint32_t* off, size_t n, uint8_t* flags_a, uint8_t* hbuf, uint8_t* sbuf) {
  hipLaunchKernelGGL(k_verify_hash, dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, pubs, sigs, msgs, off, n, flags_a, hbuf, sbuf);
  return hipGetLastError();
}
hipError_t verify_prep_pts(hipStream_t st, const uint8_t* pub_enc, const int32_t* pubs_ext, const uint8_t* sigs, const uint8_t* msgs, const uint32_t* off, size_t n,
                           uint8_t* flags_a, uint8_t* hbuf, uint8_t* sbuf, int32_t* a_ext) {
  hipLaunchKernelGGL(k_verify_prep_pts, dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, pub_enc, pubs_ext, sigs, msgs, off, n, flags_a, hbuf, sbuf, a_ext);
  return hipGetLastError();
}
hipError_t verify_prep_r(hipStream_t st, const uint8_t* sigs, size_t n, uint8_t* flags_r, uint4* proj, size_t stride, size_t offset) {
  hipLaunchKernelGGL(k_verify_prep_r, dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, sigs, n, flags_r, proj, stride, offset);
  return hipGetLastError();
}
hipError_t verify_final(hipStream_t st, const uint4* proj, size_t stride, size_t n, const uint8_t* flags_a, const uint8_t* flags_r, int flavor, uint8_t* status, DoneFlag df) {
  hipLaunchKernelGGL(k_verify_final, dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, proj, stride, n, flags_a, flags_r, flavor, status, df);
  return hipGetLastError();
}
hipError_t verify_tail_enc(hipStream_t st, uint4* proj, size_t stride, size_t n, const uint8_t* sigs, const uint8_t* flags_a, int flavor, uint8_t* status, DoneFlag df, bool four) {
  const size_t M = (n + FINISH_K - 1) / FINISH_K;
  hipLaunchKernelGGL(k_verify_diff, dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, proj, stride, n);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  if (four) hipLaunchKernelGGL(k_verify_final_enc4, dim3(blocks_for((n + 3) / 4)), dim3(KYB_BLOCK), 0, st, proj, stride, n, sigs, flags_a, flavor, status);
  else hipLaunchKernelGGL(k_verify_final_enc, dim3(blocks_for(M)), dim3(KYB_BLOCK), 0, st, proj, stride, n, sigs, flags_a, flavor, status);
  e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_verify_fixup, dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, sigs, n, flags_a, flavor, status, df);
  return hipGetLastError();
}
hipError_t pripoly_eval(hipStream_t st, const uint8_t* coeffs, size_t m, size_t t, const uint32_t* indices, size_t k, uint32_t segs, uint8_t* part, uint8_t* out) {
  const uint32_t len = (uint32_t)((t + segs - 1) / segs);
  hipLaunchKernelGGL(k_pripoly_eval_part, dim3(blocks_for(m * k * segs)), dim3(KYB_BLOCK), 0, st, coeffs, m, t, indices, k, segs, len, segs > 1u ? part : out);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess || segs == 1u) return e;
  hipLaunchKernelGGL(k_pripoly_eval_sum, dim3(blocks_for(m * k)), dim3(KYB_BLOCK), 0, st, part, m * k, segs, out);
  return hipGetLastError();
}
hipError_t sign_hash(hipStream_t st, const uint8_t* x, const uint8_t* k, const uint8_t* msgs, const uint32_t* off, size_t n,
                     const uint8_t* r_enc, const uint8_t* a_enc, uint8_t* sig, DoneFlag df) {
  hipLaunchKernelGGL(k_sign_hash, dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, x, k, msgs, off, n, r_enc, a_enc, sig, df);
  return hipGetLastError();
}
hipError_t eddsa_prep(hipStream_t st, const uint8_t* seeds, const uint8_t* msgs, const uint32_t* off, size_t n, uint8_t* xbuf, uint8_t* kbuf) {
  hipLaunchKernelGGL(k_eddsa_prep, dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, seeds, msgs, off, n, xbuf, kbuf);
  return hipGetLastError();
}
}}  // namespace kyb::launch
