// Scalar multiplication drivers: variable base (ge.rs:508-568) and fixed base (ge.rs:442-486).
//
// Both are written against a small "table policy" so the same source runs
//   * on the device with the per-lane table in an HBM/L2-resident workspace (variable base) or the
//     shared base-point table in LDS (fixed base), and
//   * in the g++ host-test build with plain arrays (tests/ only).
// Selection is constant-time in the reference's sense (ge.rs:423-434, 488-500): every entry of the
// window is read by every lane and merged under a mask; no lane-dependent address is formed.
#pragma once
#include "ge25519.h"

namespace kyb {

// Variable base.  Tbl must provide:
//   void store(int e, const ge_cached& c)          e = 0..7 (wave-uniform)
//   void select(ge_cached& c, uint32_t mag)        c = (mag ? entry[mag-1] : identity), full scan
//   scan_begin / scan_issue(k) / scan_merge(k) / scan_end: the same scan cut into four slices
// out = a * P as P2 (X:Y:Z), digit handling exactly as ge.rs:519-567 (see sc_recode).
template <class Tbl>
KYB_HD void ge_scalarmult(ge_p2& out, const uint32_t a[8], const ge_p3& P, Tbl& tbl) {
  sc_digits dg;
  sc_recode(dg, a);

  // table 1P..8P in cached form (ge.rs:537-543): 1M + 7 x (4M + 4M + 1M)
  ge_cached c;
  ge_p3_to_cached(c, P);
  tbl.store(0, c);
#if defined(__HIPCC__)
#pragma unroll 1
#endif
  for (int i = 1; i < 8; ++i) {
    ge_p1p1 t;
    ge_p3 u;
    ge_add(t, P, c);
    ge_p1p1_to_p3(u, t);
    ge_p3_to_cached(c, u);
    tbl.store(i, c);
  }

  // Main loop.  The table scan for digit i-1 does not depend on the accumulator, so it is split into
  // four slices (two entries each) that ride along the four doublings of window i: the loads of a
  // slice are issued before a doubling and merged after it, which hides their L2/MALL latency
  // behind ~2,500 cycles of field arithmetic without holding more than two entries in flight.
  ge_p1p1 t;
  ge_p1p1_0(t);
  uint32_t neg = 0;
  tbl.select(c, dg.top);                            // digit 63: not recentred, never negative
#if defined(__HIPCC__)
#pragma unroll 1
#endif
  for (int i = 63; i >= 0; --i) {
    ge_p3 u;
    ge_p1p1_to_p3(u, t);                           // 4M
    ge_cached_cneg(c, neg);
    ge_add(t, u, c);                               // 4M
    if (i > 0) {
      uint32_t mag;
      sc_digit(mag, neg, dg, i - 1);
      typename Tbl::scan st;
      tbl.scan_begin(st, mag);
#if defined(__HIPCC__)
#pragma unroll 1
#endif
      for (int k = 0; k < 4; ++k) {                // t <<= 4: 4 x (3M + 4S)
        typename Tbl::slice sl;
        tbl.scan_issue(sl, k);
        ge_p2 r;
        ge_p1p1_to_p2(r, t);
        ge_p2_dbl(t, r.X, r.Y, r.Z);
        tbl.scan_merge(st, sl, k);
      }
      tbl.scan_end(c, st);
    }
  }
  ge_p1p1_to_p2(out, t);
}

// Fixed base.  Tbl must provide:
//   void select(ge_precomp& c, int pos, uint32_t mag)   c = (mag ? T[pos][mag-1] : identity)
// with T[pos][j] = (j+1) * 16^pos * B in affine precomputed form, pos = 0..63.  The reference keeps
// only the 32 even positions (constants.rs:89) and spends 4 doublings to reach the odd ones
// (ge.rs:470-479); a 64-position table (65,536 B, one LDS image per workgroup) removes the
// doublings and makes the loop body one mixed addition.
template <class Tbl>
KYB_HD void ge_scalarmult_base(ge_p3& h, const uint32_t a[8], Tbl& tbl) {
  sc_digits dg;
  sc_recode(dg, a);
  ge_p3_0(h);
#if defined(__HIPCC__)
#pragma unroll 1
#endif
  for (int pos = 0; pos < 64; ++pos) {
    uint32_t mag, neg;
    sc_digit(mag, neg, dg, pos);
    if (pos == 63) { mag = dg.top; neg = 0; }
    ge_precomp c;
    tbl.select(c, pos, mag);
    ge_precomp_cneg(c, neg);
    ge_p1p1 t;
    ge_madd(t, h, c);                              // 3M
    ge_p1p1_to_p3_after_add(h, t);                 // 4M
  }
}

// Multiplication by a SMALL PUBLIC scalar x < 2^nbits (PubPoly::eval's x_i = i + 1, poly.rs:457-469:
// the share index is public, so neither the bit length nor the operation count needs hiding; lanes
// still run the same nbits steps and merge the conditional addition under a mask because the bits
// differ between lanes of a wave).  Left-to-right binary: nbits x (double + masked add).
// The reference routes this through the full 64-window ge_scalar_mult (point.rs:214-220).
KYB_HD void ge_small_mul(ge_p3& out, const ge_p3& in, uint32_t x, int nbits) {
  ge_cached cin;
  ge_p3_to_cached(cin, in);
  ge_p3 acc;
  ge_p3_0(acc);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int b = nbits - 1; b >= 0; --b) {
    ge_p1p1 t;
    ge_p3 d, a;
    ge_p2_dbl(t, acc.X, acc.Y, acc.Z);
    ge_p1p1_to_p3(d, t);
    const uint32_t bit = (x >> b) & 1u;
#if defined(__HIP_DEVICE_COMPILE__)
    // share indices are public: when no lane of the wavefront has this bit set (a verifier evaluates every dealer's polynomial at
    // its OWN index, so the lanes usually agree) the addition is skipped for the whole wavefront
    if (__builtin_amdgcn_ballot_w64(bit != 0u) == 0ull) { acc = d; continue; }
#endif
    ge_add(t, d, cin);
    ge_p1p1_to_p3(a, t);
    fe_select(acc.X, d.X, a.X, bit); fe_select(acc.Y, d.Y, a.Y, bit);
    fe_select(acc.Z, d.Z, a.Z, bit); fe_select(acc.T, d.T, a.T, bit);
  }
  out = acc;
}

// PubPoly::eval (poly.rs:457-469): v = sum_j commit_j * x^j by Horner, x = index + 1.
// `load_commit(j, P)` supplies commitment j as an extended point.
template <class LoadCommit>
KYB_HD void ge_poly_eval_p3(ge_p3& v, LoadCommit load_commit, int t, uint32_t x, int nbits) {
  ge_p3_0(v);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int j = t - 1; j >= 0; --j) {
    ge_p3 m, c;
    ge_small_mul(m, v, x, nbits);
    load_commit(j, c);
    ge_cached cc;
    ge_p3_to_cached(cc, c);
    ge_p1p1 r;
    ge_add(r, m, cc);
    ge_p1p1_to_p3(v, r);
  }
}
template <class LoadCommit>
KYB_HD void ge_poly_eval(ge_p2& out, LoadCommit load_commit, int t, uint32_t x, int nbits) {
  ge_p3 v;
  ge_poly_eval_p3(v, load_commit, t, x, nbits);
  fe_copy(out.X, v.X); fe_copy(out.Y, v.Y); fe_copy(out.Z, v.Z);
}

// Point equality without inversions (the reference's eq encodes both sides, point.rs:227-241):
// X1 Z2 == X2 Z1 and Y1 Z2 == Y2 Z1.  Total like the reference's: a record with Z = 0 (e.g. `Point::default()`, all-zero
// limbs) encodes as x = y = 0 there (0^(p-2) = 0, ge.rs:112-122), so it equals another Z = 0 record and a Z != 0 record
// only if that one's X and Y are both 0.
KYB_HD uint32_t ge_equal(const ge_p3& a, const ge_p3& b) {
  fe l, r, d;
  fe_mul(l, a.X, b.Z); fe_mul(r, b.X, a.Z); fe_sub(d, l, r);
  const uint32_t xne = fe_is_nonzero(d);
  fe_mul(l, a.Y, b.Z); fe_mul(r, b.Y, a.Z); fe_sub(d, l, r);
  const uint32_t yne = fe_is_nonzero(d);
  const uint32_t cross = 1u - (xne | yne);
  const uint32_t za = 1u - fe_is_nonzero(a.Z), zb = 1u - fe_is_nonzero(b.Z);
  const uint32_t a00 = (1u - fe_is_nonzero(a.X)) & (1u - fe_is_nonzero(a.Y)), b00 = (1u - fe_is_nonzero(b.X)) & (1u - fe_is_nonzero(b.Y));
  // za & zb -> equal; za only -> b must be (0, 0); zb only -> a must be (0, 0); neither -> the cross-multiplication
  return (za & zb) | (za & (1u - zb) & b00) | (zb & (1u - za) & a00) | ((1u - za) & (1u - zb) & cross);
}

// --- plain-array policies (host-test build, and the one-off device table generator) ---
struct tbl_array_cached {
  ge_cached e[8];
  struct scan { ge_cached c; uint32_t mag; };
  struct slice {};
  KYB_HD void store(int i, const ge_cached& c) { e[i] = c; }
  KYB_HD void merge(ge_cached& c, int i, uint32_t mag) {
    uint32_t m = (mag == (uint32_t)(i + 1));
    fe_cmov(c.YpX, e[i].YpX, m); fe_cmov(c.YmX, e[i].YmX, m);
    fe_cmov(c.Z, e[i].Z, m); fe_cmov(c.T2d, e[i].T2d, m);
  }
  KYB_HD void select(ge_cached& c, uint32_t mag) {
    fe_one(c.YpX); fe_one(c.YmX); fe_one(c.Z); fe_zero(c.T2d);
    for (int i = 0; i < 8; ++i) merge(c, i, mag);
  }
  KYB_HD void scan_begin(scan& st, uint32_t mag) {
    st.mag = mag;
    fe_one(st.c.YpX); fe_one(st.c.YmX); fe_one(st.c.Z); fe_zero(st.c.T2d);
  }
  KYB_HD void scan_issue(slice&, int) {}
  KYB_HD void scan_merge(scan& st, slice&, int k) { merge(st.c, 2 * k, st.mag); merge(st.c, 2 * k + 1, st.mag); }
  KYB_HD void scan_end(ge_cached& c, scan& st) { c = st.c; }
};

// base table image: uint32 [64 pos][8 quads][8 entries][4] = 65,536 B, limbs canonical.  Dword k
// (0..29: ypx[10] ymx[10] xy2d[10]; 30,31 = 0) of entry (pos, j) sits in quad k/4; the eight entries
// of one quad are contiguous, so eight lanes can fetch eight different entries with one
// conflict-free ds_read_b128 (see tbl_lds_bperm in kernels.hip).
#define KYB_BASE_TABLE_WORDS (64 * 8 * 8 * 4)
#define KYB_BT_IDX(pos, j, k) ((((pos) * 8 + ((k) >> 2)) * 8 + (j)) * 4 + ((k) & 3))
struct tbl_base_words {
  const uint32_t* w;
  KYB_HD void select(ge_precomp& c, int pos, uint32_t mag) {
    fe_one(c.ypx); fe_one(c.ymx); fe_zero(c.xy2d);
    for (int j = 0; j < 8; ++j) {
      uint32_t m = (mag == (uint32_t)(j + 1));
      for (int k = 0; k < 10; ++k) {
        c.ypx.v[k] = m ? w[KYB_BT_IDX(pos, j, k)] : c.ypx.v[k];
        c.ymx.v[k] = m ? w[KYB_BT_IDX(pos, j, 10 + k)] : c.ymx.v[k];
        c.xy2d.v[k] = m ? w[KYB_BT_IDX(pos, j, 20 + k)] : c.xy2d.v[k];
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------
// Fixed base, signed radix 32: 52 windows x 16 multiples, T32[pos][j] = (j+1) * 32^pos * B.
// 52 mixed additions instead of 64 (-19 %); the table image is 106,496 B and lives in the LDS of a
// 1024-thread workgroup.  The scalar goes through sc_effective first (the reference's top-digit quirk,
// possibly a negative integer), so the digits below recode a plain 256-bit magnitude:
//     b = mag + sum_{i<51} 16*32^i  (9 words);  digit_i = group_i(b) - 16 in [-16, 15], digit_51 = group_51(b) in {0,1,2}
#define KYB_BASE32_POS 52
#define KYB_BASE32_TABLE_WORDS (KYB_BASE32_POS * 8 * 16 * 4)
#define KYB_BT32_IDX(pos, j, k) ((((pos) * 8 + ((k) >> 2)) * 16 + (j)) * 4 + ((k) & 3))
struct sc_digits32 {
  uint32_t w[9];     // b, consumed 5 bits at a time from the bottom
  uint32_t neg;      // the whole scalar is negative: negate the result
};
KYB_HD void sc_recode32(sc_digits32& d, const uint32_t a[8]) {
  const uint32_t c32[8] = KYB_W_RECODE32;
  uint32_t mag[8];
  sc_effective(d.neg, mag, a);
  uint64_t c = 0;
  KYB_UNROLL for (int i = 0; i < 8; ++i) {
    c += (uint64_t)mag[i] + c32[i];
    d.w[i] = (uint32_t)c;
    c >>= 32;
  }
  d.w[8] = (uint32_t)c;
}
// next digit (lowest 5-bit group), then shift the register down by 5 bits
KYB_HD void sc_next_digit32(uint32_t& mag, uint32_t& neg, sc_digits32& d, bool top) {
  const int v = (int)(d.w[0] & 31u) - (top ? 0 : 16);
  neg = v < 0;
  mag = neg ? (uint32_t)(-v) : (uint32_t)v;
  KYB_UNROLL for (int i = 0; i < 8; ++i) d.w[i] = (d.w[i] >> 5) | (d.w[i + 1] << 27);
  d.w[8] >>= 5;
}
// Tbl: void select(ge_precomp& c, int pos, uint32_t mag)  with mag in 0..16
template <class Tbl>
KYB_HD void ge_scalarmult_base32(ge_p3& h, const uint32_t a[8], Tbl& tbl) {
  sc_digits32 dg;
  sc_recode32(dg, a);
  ge_p3_0(h);
#if defined(__HIPCC__)
#pragma unroll 1
#endif
  for (int pos = 0; pos < KYB_BASE32_POS; ++pos) {
    uint32_t mag, neg;
    sc_next_digit32(mag, neg, dg, pos == KYB_BASE32_POS - 1);
    ge_precomp c;
    tbl.select(c, pos, mag);
    ge_precomp_cneg(c, neg);
    ge_p1p1 t;
    ge_madd(t, h, c);
    ge_p1p1_to_p3_after_add(h, t);
  }
  // a' < 0: negate (x -> -x, t -> -t)
  fe nx, nt;
  fe_neg(nx, h.X); fe_reduce_weak(nx, nx);
  fe_neg(nt, h.T); fe_reduce_weak(nt, nt);
  fe_cmov(h.X, nx, dg.neg);
  fe_cmov(h.T, nt, dg.neg);
}
struct tbl_base32_words {
  const uint32_t* w;
  KYB_HD void select(ge_precomp& c, int pos, uint32_t mag) {
    fe_one(c.ypx); fe_one(c.ymx); fe_zero(c.xy2d);
    for (int j = 0; j < 16; ++j) {
      uint32_t m = (mag == (uint32_t)(j + 1));
      for (int k = 0; k < 10; ++k) {
        c.ypx.v[k] = m ? w[KYB_BT32_IDX(pos, j, k)] : c.ypx.v[k];
        c.ymx.v[k] = m ? w[KYB_BT32_IDX(pos, j, 10 + k)] : c.ymx.v[k];
        c.xy2d.v[k] = m ? w[KYB_BT32_IDX(pos, j, 20 + k)] : c.xy2d.v[k];
      }
    }
  }
};

// ---------------------------------------------------------------------------------------------
// Fixed base, radix 64 with ODD signed digits: 42 windows x 32 odd multiples + a top window of 16,
// T64[pos][j] = (2j+1) * 64^pos * B.  43 mixed additions instead of 52 (-17 %).  The image packs an
// entry into its 30 words (no padding): 42 * 3840 + 1920 = 163,200 B, i.e. the whole 160 KiB LDS of
// a CU, owned by one 1024-thread workgroup.  Per window the three field elements of an entry live in
// separately addressable planes (kyb_bt64_in_win), read with six conflict-free 16-byte and three 8-byte
// LDS loads; the 64 lanes of a wave hold the 32 entries (lanes 0..31) and their negatives (lanes 32..63).
// Recoding (regular, Joye-Tunstall style: no zero digit, hence no neutral-element case in the loop):
//     k = mag, made odd by adding L when it is even (L*B is the neutral element; mag < 10 * 2^252 stays < 2^256);
//     c_i = 6-bit groups of k >> 1;  digit_i = 2 c_i - 63 in {-63, -61, .., 63} for i < 42;
//     digit_42 = 2 c_42 + 1 in {1, 3, .., 11}            (sum_i digit_i 64^i = 2 (k >> 1) + 1 = k)
// (mag < 9 * 2^252: sc_effective keeps a top radix-16 digit of at most 8).
#define KYB_BASE64_POS 43
#define KYB_BASE64_WIN_WORDS 960
#define KYB_BASE64_TOP_BASE (42 * KYB_BASE64_WIN_WORDS)
#define KYB_BASE64_TABLE_WORDS (KYB_BASE64_TOP_BASE + 480)
// Entry-major copy of the same table for the one-item-per-wavefront kernels (kernels_coop.hip, coop_table_entry): 32 words = one
// 128-byte line per entry (its 30 words, two zeros), 32 entries per window, the top window's 16 behind the 42 full ones.  Each context
// derives it from the image on its own GPU (k_base_table_coop) when the image is built or imported; it is not part of what travels.
#define KYB_COOP_WIN_WORDS (32 * 32)
#define KYB_COOP_TABLE_WORDS (42 * KYB_COOP_WIN_WORDS + 16 * 32)
// word k (0..9 ypx, 10..19 ymx, 20..29 xy2d) of entry j of a window with E entries, relative to the window:
//   plane A  words [ 0E,  8E)  [2 quads][E][4]  ypx 0..7        plane a  words [16E, 18E)  [E][2]  ypx 8, 9
//   plane B  words [ 8E, 16E)  [2 quads][E][4]  ymx 0..7        plane b  words [18E, 20E)  [E][2]  ymx 8, 9
//   plane C  words [20E, 28E)  [2 quads][E][4]  xy2d 0..7       plane c  words [28E, 30E)  [E][2]  xy2d 8, 9
// ypx and ymx sit in separately addressable planes: a lane that holds the NEGATED entry (ymx, ypx, -xy2d) reads
// A/a and B/b with exchanged base addresses instead of swapping registers afterwards.
KYB_HD constexpr int kyb_bt64_in_win(int E, int j, int k) {
  const int g = k / 10, r = k % 10;                  // g: 0 ypx, 1 ymx, 2 xy2d
  const int big = g == 0 ? 0 : (g == 1 ? 8 * E : 20 * E), small = g == 0 ? 16 * E : (g == 1 ? 18 * E : 28 * E);
  return r < 8 ? big + ((r >> 2) * E + j) * 4 + (r & 3) : small + j * 2 + (r - 8);
}
#define KYB_BT64_IDX(pos, j, k) \
  ((pos) < 42 ? (pos) * KYB_BASE64_WIN_WORDS + kyb_bt64_in_win(32, j, k) : KYB_BASE64_TOP_BASE + kyb_bt64_in_win(16, j, k))
struct sc_digits64 {
  uint32_t w[8];     // k >> 1, consumed 6 bits at a time from the bottom
  uint32_t neg;      // the whole scalar is negative: negate the result
};
KYB_HD void sc_recode64(sc_digits64& d, const uint32_t a[8]) {
  const uint32_t lw[8] = KYB_W_L;
  uint32_t mag[8];
  sc_effective(d.neg, mag, a);
  const uint32_t even = 1u - (mag[0] & 1u);
  uint64_t c = 0;
  KYB_UNROLL for (int i = 0; i < 8; ++i) {
    c += (uint64_t)mag[i] + (even ? lw[i] : 0u);
    mag[i] = (uint32_t)c;
    c >>= 32;
  }
  KYB_UNROLL for (int i = 0; i < 7; ++i) d.w[i] = (mag[i] >> 1) | (mag[i + 1] << 31);
  d.w[7] = mag[7] >> 1;
}
// next digit as (table index = (|digit| - 1) / 2, negative flag), then shift the register down by 6 bits
KYB_HD void sc_next_digit64(uint32_t& idx, uint32_t& neg, sc_digits64& d, bool top) {
  const uint32_t c = d.w[0] & 63u;
  neg = top ? 0u : (uint32_t)(c < 32u);
  idx = top ? c : (neg ? 31u - c : c - 32u);
  KYB_UNROLL for (int i = 0; i < 7; ++i) d.w[i] = (d.w[i] >> 6) | (d.w[i + 1] << 26);
  d.w[7] >>= 6;
}
// Tbl: void select(ge_precomp& c, int pos, uint32_t idx, uint32_t neg) for pos < 42: entry idx in 0..31, negated when neg
//      void select_top(ge_precomp& c, uint32_t idx) for the top window (idx in 0..15, never negative)
// (One loop for all 43 windows, the top window's 16-entry selection behind a uniform branch: with the top window peeled off behind the
// loop, as it was, the 1024-thread kernel of this round needs 12 bytes of scratch; this form needs none — tests/test_build_resources.py.)
template <class Tbl>
KYB_HD void ge_scalarmult_base64(ge_p3& h, const uint32_t a[8], Tbl& tbl) {
  sc_digits64 dg;
  sc_recode64(dg, a);
  ge_p3_0(h);
#if defined(__HIPCC__)
#pragma unroll 1
#endif
  for (int pos = 0; pos < KYB_BASE64_POS; ++pos) {
    uint32_t mag, neg;
    sc_next_digit64(mag, neg, dg, pos == KYB_BASE64_POS - 1);
    ge_precomp c;
    if (pos < KYB_BASE64_POS - 1) tbl.select(c, pos, mag, neg); else tbl.select_top(c, mag);
    ge_p1p1 t;
    ge_madd_lazy_t(t, h, c);
    ge_p1p1_to_p3_lazy_t(h, t);
  }
  fe nx, nt;
  fe_neg(nx, h.X); fe_reduce_weak(nx, nx);
  fe_neg(nt, h.T); fe_reduce_weak(nt, nt);
  fe_cmov(h.X, nx, dg.neg);
  fe_cmov(h.T, nt, dg.neg);
}
// Windows [p0, p1) of the same sum: the radix-64 form has no doublings, so the 43 additions split freely — four lanes (wavefronts) take a quarter
// of the windows each and the four partial points are added up (k_mul_base64_quarters, kernels_base.hip).  The sign of the whole scalar goes onto
// every part.
template <class Tbl>
KYB_HD void ge_scalarmult_base64_part(ge_p3& h, const uint32_t a[8], Tbl& tbl, int p0, int p1) {
  sc_digits64 dg;
  sc_recode64(dg, a);
#if defined(__HIPCC__)
#pragma unroll 1
#endif
  for (int pos = 0; pos < p0; ++pos) {            // (uniform trip count: the part is the wavefront's number)
    KYB_UNROLL for (int i = 0; i < 7; ++i) dg.w[i] = (dg.w[i] >> 6) | (dg.w[i + 1] << 26);
    dg.w[7] >>= 6;
  }
  ge_p3_0(h);
#if defined(__HIPCC__)
#pragma unroll 1
#endif
  for (int pos = p0; pos < p1; ++pos) {
    uint32_t mag, neg;
    sc_next_digit64(mag, neg, dg, pos == KYB_BASE64_POS - 1);
    ge_precomp c;
    if (pos < KYB_BASE64_POS - 1) tbl.select(c, pos, mag, neg); else tbl.select_top(c, mag);
    ge_p1p1 t;
    ge_madd_lazy_t(t, h, c);
    ge_p1p1_to_p3_lazy_t(h, t);
  }
  fe nx, nt;
  fe_neg(nx, h.X); fe_reduce_weak(nx, nx);
  fe_neg(nt, h.T); fe_reduce_weak(nt, nt);
  fe_cmov(h.X, nx, dg.neg);
  fe_cmov(h.T, nt, dg.neg);
}
struct tbl_base64_words {
  const uint32_t* w;
  KYB_HD void scan(ge_precomp& c, int pos, uint32_t idx, int entries) {
    fe_one(c.ypx); fe_one(c.ymx); fe_zero(c.xy2d);
    for (int j = 0; j < entries; ++j) {
      uint32_t m = (idx == (uint32_t)j);
      for (int k = 0; k < 10; ++k) {
        c.ypx.v[k] = m ? w[KYB_BT64_IDX(pos, j, k)] : c.ypx.v[k];
        c.ymx.v[k] = m ? w[KYB_BT64_IDX(pos, j, 10 + k)] : c.ymx.v[k];
        c.xy2d.v[k] = m ? w[KYB_BT64_IDX(pos, j, 20 + k)] : c.xy2d.v[k];
      }
    }
  }
  KYB_HD void select(ge_precomp& c, int pos, uint32_t idx, uint32_t neg) { scan(c, pos, idx, 32); ge_precomp_cneg(c, neg); }
  KYB_HD void select_top(ge_precomp& c, uint32_t idx) { scan(c, 42, idx, 16); }
};

// One entry of the base table: (j+1) * 16^pos * B, normalised to affine (y+x, y-x, 2dxy).
// Used by the init kernel (one thread per entry); B is decoded from its RFC 8032 encoding.
KYB_HD void ge_base_table_entry(uint32_t* image, int pos, int j) {
  const uint32_t benc[8] = KYB_W_BASE_ENC;
  const fe d2 = {KYB_FE_D2};
  ge_p3 B;
  ge_decode(B, benc);
  uint32_t a[8];
  for (int i = 0; i < 8; ++i) a[i] = 0;
  a[pos >> 3] = (uint32_t)(j + 1) << ((pos & 7) * 4);   // (j+1) * 16^pos; 8*16^63 = 2^255 is a top digit of 8
  tbl_array_cached tbl;
  ge_p2 r;
  ge_scalarmult(r, a, B, tbl);
  fe recip, x, y, t;
  fe_invert(recip, r.Z);
  fe_mul(x, r.X, recip);
  fe_mul(y, r.Y, recip);
  fe ypx, ymx, xy2d;
  fe_add(ypx, y, x);
  fe_sub(ymx, y, x);
  fe_mul(t, x, y);
  fe_mul(xy2d, t, d2);
  fe_canon(ypx, ypx);
  fe_canon(ymx, ymx);
  fe_canon(xy2d, xy2d);
  for (int k = 0; k < 10; ++k) {
    image[KYB_BT_IDX(pos, j, k)] = ypx.v[k];
    image[KYB_BT_IDX(pos, j, 10 + k)] = ymx.v[k];
    image[KYB_BT_IDX(pos, j, 20 + k)] = xy2d.v[k];
  }
  image[KYB_BT_IDX(pos, j, 30)] = 0;
  image[KYB_BT_IDX(pos, j, 31)] = 0;
}

// One entry of the radix-32 table: (j+1) * 32^pos * B = (j+1) * (2^(5 pos) * B); 2^255 (pos 51) is a top
// radix-16 digit of 8, which the windowed routine honours.
KYB_HD void ge_base32_table_entry(uint32_t* image, int pos, int j) {
  const uint32_t benc[8] = KYB_W_BASE_ENC;
  const fe d2 = {KYB_FE_D2};
  ge_p3 B;
  ge_decode(B, benc);
  uint32_t a[8];
  for (int i = 0; i < 8; ++i) a[i] = 0;
  a[(5 * pos) >> 5] = 1u << ((5 * pos) & 31);
  tbl_array_cached tbl;
  ge_p2 r;
  ge_scalarmult(r, a, B, tbl);
  ge_p3 Ppos, Q;                                  // P2 -> P3
  fe_mul(Ppos.X, r.X, r.Z); fe_mul(Ppos.Y, r.Y, r.Z); fe_sq(Ppos.Z, r.Z); fe_mul(Ppos.T, r.X, r.Y);
  ge_small_mul(Q, Ppos, (uint32_t)(j + 1), 5);
  fe recip, x, y, t, ypx, ymx, xy2d;
  fe_invert(recip, Q.Z);
  fe_mul(x, Q.X, recip);
  fe_mul(y, Q.Y, recip);
  fe_add(ypx, y, x);
  fe_sub(ymx, y, x);
  fe_mul(t, x, y);
  fe_mul(xy2d, t, d2);
  fe_canon(ypx, ypx); fe_canon(ymx, ymx); fe_canon(xy2d, xy2d);
  for (int k = 0; k < 10; ++k) {
    image[KYB_BT32_IDX(pos, j, k)] = ypx.v[k];
    image[KYB_BT32_IDX(pos, j, 10 + k)] = ymx.v[k];
    image[KYB_BT32_IDX(pos, j, 20 + k)] = xy2d.v[k];
  }
  image[KYB_BT32_IDX(pos, j, 30)] = 0;
  image[KYB_BT32_IDX(pos, j, 31)] = 0;
}

// One entry of the radix-64 table: (2j+1) * 64^pos * B = (2j+1) * (2^(6 pos) * B); 2^252 (pos 42) is a top
// radix-16 digit of 1.
KYB_HD void ge_base64_table_entry(uint32_t* image, int pos, int j) {
  const uint32_t benc[8] = KYB_W_BASE_ENC;
  const fe d2 = {KYB_FE_D2};
  ge_p3 B;
  ge_decode(B, benc);
  uint32_t a[8];
  for (int i = 0; i < 8; ++i) a[i] = 0;
  a[(6 * pos) >> 5] = 1u << ((6 * pos) & 31);
  tbl_array_cached tbl;
  ge_p2 r;
  ge_scalarmult(r, a, B, tbl);
  ge_p3 Ppos, Q;
  ge_p2_to_p3(Ppos, r);
  ge_small_mul(Q, Ppos, (uint32_t)(2 * j + 1), 6);
  fe recip, x, y, t, ypx, ymx, xy2d;
  fe_invert(recip, Q.Z);
  fe_mul(x, Q.X, recip);
  fe_mul(y, Q.Y, recip);
  fe_add(ypx, y, x);
  fe_sub(ymx, y, x);
  fe_mul(t, x, y);
  fe_mul(xy2d, t, d2);
  fe_canon(ypx, ypx); fe_canon(ymx, ymx); fe_canon(xy2d, xy2d);
  for (int k = 0; k < 10; ++k) {
    image[KYB_BT64_IDX(pos, j, k)] = ypx.v[k];
    image[KYB_BT64_IDX(pos, j, 10 + k)] = ymx.v[k];
    image[KYB_BT64_IDX(pos, j, 20 + k)] = xy2d.v[k];
  }
}

}  // namespace kyb
