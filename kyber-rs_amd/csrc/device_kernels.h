// Every __global__ kernel of the engine (included by kernels.hip only; see the map at its top).
#pragma once

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

__global__ void __launch_bounds__(64) k_base_table64(uint32_t* image) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;   // 0..1359: 42 windows x 32 entries, then the top window's 16
  if (e < 42 * 32) ge_base64_table_entry(image, e >> 5, e & 31);
  else if (e < 42 * 32 + 16) ge_base64_table_entry(image, 42, e - 42 * 32);
}

// Fixed base, signed radix 64: one workgroup per CU owns the whole LDS (163,200 B table); 43 mixed additions
// per item.  BLOCK = 1024 (4 waves/SIMD) when the batch fills the chip, 256 (1 wave/SIMD, four times as many
// CUs busy) for batches that do not.
// Two scalar arrays may be multiplied in one launch (signing: the nonces and the private keys): items
// [0, n_a) come from `scalars`, items [n_a, n) from `scalars_b`.
template <bool SPLIT, int BLOCK>
__global__ void __launch_bounds__(BLOCK, BLOCK / 256)
k_mul_base64(const uint8_t* __restrict__ scalars, const uint8_t* __restrict__ scalars_b, size_t n_a, size_t n,
             uint8_t* __restrict__ out_enc, int32_t* __restrict__ out_ext,
             const uint4* __restrict__ table_image, uint4* __restrict__ proj, size_t proj_stride, size_t proj_offset) {
  __shared__ uint4 lds_tbl[KYB_BASE64_TABLE_WORDS / 4];
  for (int k = threadIdx.x; k < KYB_BASE64_TABLE_WORDS / 4; k += BLOCK) lds_tbl[k] = table_image[k];
  __syncthreads();
  tbl_lds64 tbl{reinterpret_cast<const uint32_t*>(lds_tbl)};
  const size_t nchunks = (n + BLOCK - 1) / BLOCK;
  for (size_t chunk = blockIdx.x; chunk < nchunks; chunk += gridDim.x) {
    const size_t i = chunk * BLOCK + threadIdx.x;
    const bool live = i < n;
    const size_t ii = live ? i : 0;
    uint32_t a[8];
    if (ii < n_a) load_words8(a, scalars, ii); else load_words8(a, scalars_b, ii - n_a);
    ge_p3 h;
    ge_scalarmult_base64(h, a, tbl);
    if (SPLIT) { if (live) store_proj(proj, proj_stride, proj_offset + i, h.X, h.Y, h.Z); }
    else finish_point(h.X, h.Y, h.Z, out_enc, out_ext, ii, live);
  }
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

// split signing, last stage: r_enc[i] = enc(R_i), a_enc[i] = enc(A_i) (from the fixed-base launches + k_finish,
// or the caller's stored public keys); h = SHA-512(R || A || msg) mod L, s = k + x h mod L.
__global__ void __launch_bounds__(KYB_BLOCK)
k_sign_hash(const uint8_t* __restrict__ x, const uint8_t* __restrict__ k, const uint8_t* __restrict__ msgs,
            const uint32_t* __restrict__ msg_off, size_t n, const uint8_t* __restrict__ r_enc, const uint8_t* __restrict__ a_enc,
            uint8_t* __restrict__ sig) {
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
#if defined(KYB_DIAG_STAMPS)
// diagnostic build only: per-wave (cycles, 100 MHz ticks) of the ladder loop go to a buffer of their own
__device__ uint64_t* kyb_diag_stamp_buf = nullptr;
__device__ size_t kyb_diag_stamp_cap = 0;
#endif
template <int WAVES>
__global__ void __launch_bounds__(KYB_BLOCK, WAVES)
k_mul_ladder(const uint8_t* __restrict__ scalars, size_t n, uint4* __restrict__ proj, size_t stride, size_t img_offset, size_t img_mod, int skip_bits) {
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
#if defined(KYB_DIAG_STAMPS)
  uint64_t stamp[2];
  ge_scalarmult_ladder_stamped(r, a, m, skip_bits, stamp);
  const size_t wave = i >> 6;
  if ((threadIdx.x & 63u) == 0 && kyb_diag_stamp_buf != nullptr && wave < kyb_diag_stamp_cap) { kyb_diag_stamp_buf[2 * wave] = stamp[0]; kyb_diag_stamp_buf[2 * wave + 1] = stamp[1]; }
#else
  ge_scalarmult_ladder(r, a, m, skip_bits);
#endif
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

// extended limbs -> projective staging records (input of the k_pair_sum passes of kyb_sum_batch)
__global__ void __launch_bounds__(KYB_BLOCK)
k_ext_to_proj(const int32_t* __restrict__ pts_ext, size_t n, uint4* __restrict__ proj, size_t stride) {
  const size_t i = (size_t)blockIdx.x * KYB_BLOCK + threadIdx.x;
  if (i >= n) return;
  ge_p3 P;
  load_ext(P, pts_ext, i);
  store_proj(proj, stride, i, P.X, P.Y, P.Z);
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
               int flavor, uint8_t* __restrict__ status) {
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

