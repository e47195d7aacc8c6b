// Device-side table policies and memory helpers of the engine (included by kernels.hip only).
#pragma once

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

// Radix-64 fixed-base table in LDS (kyb_bt64_in_win: separate planes for ypx / ymx / xy2d).  Lane l of a wave holds
// entry l mod 32 — lanes 32..63 hold it NEGATED, (ymx, ypx, -xy2d): the exchange of ypx and ymx is an exchange of
// the two plane addresses (free), the negation of xy2d is 2p - x done as (x ^ m) + (m & (2p + 1)) with the lane's
// constant mask m.  The wanted entry with the wanted sign is then pulled from lane (neg << 5 | idx) with
// ds_bpermute_b32: the requesting lane needs no conditional negation and, the digits being odd, no neutral-element
// case.  The top window (16 entries, digit always positive) has no negated copies.
struct tbl_lds64 {
  const uint32_t* t;  // LDS
  // The selection in two halves, so that a caller can put arithmetic between them:
  //   rows():    this lane's own entry of the window (entry lane mod E, lanes 32..63 the negated one) — nine LDS reads whose addresses depend
  //              on nothing but the window and the lane
  //   permute(): the wanted entry pulled from the lane that holds it — thirty ds_bpermute_b32, in place
  template <int E, bool SIGNED>
  __device__ __forceinline__ void rows(uint32_t own[30], const uint32_t* win, bool once = false) {
    uint32_t lane = threadIdx.x & 63u;
    if (once) asm volatile("" : "+v"(lane));     // a fetch outside the window loop: recompute the lane terms there instead of keeping them live across it
    const uint32_t mine = lane & (uint32_t)(E - 1);
    const uint32_t s = SIGNED ? (lane >> 5) & 1u : 0u;                 // this lane holds the negated entry
    const uint32_t* pa = win + (s ? 8 * E : 0);                        // plane A (ypx 0..7) or B (ymx 0..7)
    const uint32_t* pb = win + (s ? 0 : 8 * E);
    const uint32_t* qa = win + (s ? 18 * E : 16 * E);                  // plane a (ypx 8, 9) or b
    const uint32_t* qb = win + (s ? 16 * E : 18 * E);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const uint4 va = *reinterpret_cast<const uint4*>(pa + (q * E + mine) * 4);
      const uint4 vb = *reinterpret_cast<const uint4*>(pb + (q * E + mine) * 4);
      const uint4 vc = *reinterpret_cast<const uint4*>(win + 20 * E + (q * E + mine) * 4);
      own[4 * q] = va.x; own[4 * q + 1] = va.y; own[4 * q + 2] = va.z; own[4 * q + 3] = va.w;
      own[10 + 4 * q] = vb.x; own[10 + 4 * q + 1] = vb.y; own[10 + 4 * q + 2] = vb.z; own[10 + 4 * q + 3] = vb.w;
      own[20 + 4 * q] = vc.x; own[20 + 4 * q + 1] = vc.y; own[20 + 4 * q + 2] = vc.z; own[20 + 4 * q + 3] = vc.w;
    }
    const uint2 wa = *reinterpret_cast<const uint2*>(qa + mine * 2);
    const uint2 wb = *reinterpret_cast<const uint2*>(qb + mine * 2);
    const uint2 wc = *reinterpret_cast<const uint2*>(win + 28 * E + mine * 2);
    own[8] = wa.x; own[9] = wa.y; own[18] = wb.x; own[19] = wb.y; own[28] = wc.x; own[29] = wc.y;
  }
  template <bool SIGNED>
  __device__ __forceinline__ void permute(ge_precomp& c, uint32_t own[30], uint32_t src_lane) {
    const uint32_t lane = threadIdx.x & 63u;
    if (SIGNED) {
      const uint32_t p2[10] = KYB_FE_2P;
      const uint32_t m = 0u - ((lane >> 5) & 1u);
#pragma unroll
      for (int i = 0; i < 10; ++i) own[20 + i] = (own[20 + i] ^ m) + (m & (p2[i] + 1u));        // this lane holds the negated entry ? 2p - x : x
    }
    const int src = (int)(((lane & ~63u) | src_lane) << 2);
    uint32_t f[30];
#pragma unroll
    for (int i = 0; i < 30; ++i) f[i] = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)own[i]);
#pragma unroll
    for (int i = 0; i < 10; ++i) { c.ypx.v[i] = f[i]; c.ymx.v[i] = f[10 + i]; c.xy2d.v[i] = f[20 + i]; }
  }
  template <int E, bool SIGNED>
  __device__ __forceinline__ void fetch(ge_precomp& c, const uint32_t* win, uint32_t src_lane, bool once = false) {
    uint32_t own[30];
    rows<E, SIGNED>(own, win, once);
    permute<SIGNED>(c, own, src_lane);
  }
  // split form for the pipelined loop of ge_scalarmult_base64 (pos < 42: signed digits; the top window: 16 entries, no negatives)
  __device__ __forceinline__ void rows_of(uint32_t own[30], int pos) { rows<32, true>(own, t + pos * KYB_BASE64_WIN_WORDS); }
  __device__ __forceinline__ void pick(ge_precomp& c, uint32_t own[30], uint32_t idx, uint32_t neg) { permute<true>(c, own, (neg << 5) | idx); }
  __device__ __forceinline__ void rows_of_top(uint32_t own[30]) {
    uint32_t top = KYB_BASE64_TOP_BASE;
    asm volatile("" : "+v"(top));
    rows<16, false>(own, t + top, true);
  }
  __device__ __forceinline__ void pick_top(ge_precomp& c, uint32_t own[30], uint32_t idx) { permute<false>(c, own, ((threadIdx.x & 63u) & 48u) | idx); }
  __device__ __forceinline__ void select(ge_precomp& c, int pos, uint32_t idx, uint32_t neg) {
    fetch<32, true>(c, t + pos * KYB_BASE64_WIN_WORDS, (neg << 5) | idx);
  }
  __device__ __forceinline__ void select_top(ge_precomp& c, uint32_t idx) {
    const uint32_t lane = threadIdx.x & 63u;
    // The top window sits 161,280 bytes into LDS, beyond the 16-bit offset field of ds_read: folded as constants its
    // nine plane addresses become nine loop-invariant VGPRs (which then spill in the 128-register build).  Behind an
    // opaque offset they are one base register + small immediate offsets.
    uint32_t top = KYB_BASE64_TOP_BASE;
    asm volatile("" : "+v"(top));
    fetch<16, false>(c, t + top, (lane & 48u) | idx, true);
  }
};

// Fixed base, software-pipelined (kernels_base.hip): the SAME additions in the same order as ge_scalarmult_base64, with the table traffic
// of window pos + 1 placed inside the arithmetic of window pos —
//     A, B of the mixed addition   |  the lane's nine LDS row reads of window pos + 1 are issued here ...
//     C and the sums               |  ... and land under this product;  then the thirty ds_bpermute of the selection are issued ...
//     conversion p1p1 -> p3        |  ... and land under its four products
// — instead of in front of the window's first product, where every wavefront of a SIMD used to wait for two dependent LDS round trips
// at the same point of the loop.  Order of issue is pinned with scheduling barriers; the rows cost 30 more live registers during C.
__device__ __forceinline__ void ge_scalarmult_base64_pipelined(ge_p3& h, const uint32_t a[8], tbl_lds64& tbl) {
  sc_digits64 dg;
  sc_recode64(dg, a);
  ge_p3_0(h);
  ge_precomp c;
  {
    uint32_t mag, neg;
    sc_next_digit64(mag, neg, dg, false);
    tbl.select(c, 0, mag, neg);
  }
#pragma unroll 1
  for (int pos = 1; pos < KYB_BASE64_POS - 1; ++pos) {
    fe A, B;
    ge_p1p1 t;
    uint32_t own[30], mag, neg;
    ge_madd_lazy_t_ab(A, B, h, c);                // window pos - 1 ...
    __builtin_amdgcn_sched_barrier(0);
    tbl.rows_of(own, pos);
    __builtin_amdgcn_sched_barrier(0);
    ge_madd_lazy_t_c(t, A, B, h, c);              // ... its last product hides the row reads
    sc_next_digit64(mag, neg, dg, false);
    __builtin_amdgcn_sched_barrier(0);
    tbl.pick(c, own, mag, neg);
    __builtin_amdgcn_sched_barrier(0);
    ge_p1p1_to_p3_lazy_t(h, t);                   // ... and the conversion hides the selection
  }
  {
    fe A, B;
    ge_p1p1 t;
    uint32_t own[30], mag, neg;
    ge_madd_lazy_t_ab(A, B, h, c);                // window 41
    tbl.rows_of_top(own);
    ge_madd_lazy_t_c(t, A, B, h, c);
    sc_next_digit64(mag, neg, dg, true);
    tbl.pick_top(c, own, mag);
    ge_p1p1_to_p3_lazy_t(h, t);
    ge_madd_lazy_t(t, h, c);                      // top window
    ge_p1p1_to_p3_lazy_t(h, t);
  }
  fe nx, nt;
  fe_neg(nx, h.X); fe_reduce_weak(nx, nx);
  fe_neg(nt, h.T); fe_reduce_weak(nt, nt);
  fe_cmov(h.X, nx, dg.neg);
  fe_cmov(h.T, nt, dg.neg);
}

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
  fe_inv(zi, Z);
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

// completion signal (launch.h, DoneFlag): called by the thread that has just stored an item's results
__device__ __forceinline__ void signal_done(const kyb::launch::DoneFlag& df) {
  if (df.flag == nullptr) return;
  __threadfence_system();                                                  // this item's results first, system-wide
  const uint32_t before = atomicAdd(df.counter, 1u);
  if (before + 1u == df.total) {
    atomicExch(df.counter, 0u);                                            // ready for the next call
    __threadfence_system();
    __hip_atomic_store(df.flag, df.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
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

