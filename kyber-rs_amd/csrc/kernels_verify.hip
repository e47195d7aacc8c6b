// SHA-512 users of the MI355X Ed25519 engine (one of the five translation units, see launch.h).
//   k_verify_prep / k_verify_prep_r / k_verify_final   eddsa_sig.rs:159-212, schnorr_sig.rs:53-110 (verify.h)
//   k_sign_hash                                        last stage of the split signing path (schnorr_sig.rs:25-47)
//   k_eddsa_prep                                       key expansion + deterministic nonce (curve.rs:74-87, eddsa_sig.rs:120-131)
#include <hip/hip_runtime.h>
#include "launch.h"
#include "schnorr.h"
// (round 1 routed the point decompression through one out-of-line copy because a single kernel decoded both A and R; with
// the two halves in kernels of their own each decodes once, inline — the call's stack frame was the kernels' 208 / 240 B
// of scratch and the reason for their 248 VGPRs)
#include "verify.h"
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


namespace kyb { namespace launch {
static inline unsigned blocks_for(size_t n) { return (unsigned)((n + KYB_BLOCK - 1) / KYB_BLOCK); }
hipError_t verify_prep(hipStream_t st, const uint8_t* pubs, const uint8_t* sigs, const uint8_t* msgs, const uint32_t* off, size_t n,
                       uint8_t* flags_a, uint8_t* hbuf, uint8_t* sbuf, int32_t* a_ext) {
  hipLaunchKernelGGL(k_verify_prep, dim3(blocks_for(n)), dim3(KYB_BLOCK), 0, st, pubs, sigs, msgs, off, n, flags_a, hbuf, sbuf, a_ext);
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
