// Small-batch (latency) kernels of the MI355X Ed25519 engine: ONE ITEM PER WAVEFRONT, field elements spread over lanes
// (coop25519.h).  One of the translation units of the library (map: launch.h).
//   k_mul_coop        Point::mul(s, Some(P))  ge.rs:508-568   the Montgomery ladder of k_mul_ladder with a PROJECTIVE base point
//                     (mont_ladder_proj, ge_ladder.h: no inversion in front), its 256 steps as three cooperative multiplication
//                     levels each; image, y-recovery and encoding replicated on all lanes around it, the one field inversion
//                     cooperative.  One launch does the whole multiplication.
//   k_mul_base_coop   Point::mul(s, None)     ge.rs:442-486   the radix-64 table of k_mul_base64 read from global memory (every
//                     line of a limb's 32 entries is touched whatever the digit), 43 cooperative mixed additions, cooperative inversion.
// Used for batches that leave the chip idle (engine.hip: `coop.max_items`); results are bit-identical to the batch kernels.
#include <hip/hip_runtime.h>
#include "launch.h"
#include "ge_scalarmult.h"
#include "ge_ladder.h"
#include "coop25519.h"
using namespace kyb;
using namespace kyb::coop;
#include "device_tables.h"

namespace {

struct row_masks {
  bool odd, r0, r1, r2, r3;
};

// encode (and optionally affine limbs) from a point (X : Y : Z) held replicated; Z^-1 computed cooperatively
// proj != nullptr: the affine point also goes to staging record proj_offset + i (X, Y, Z = 1) for k_verify_final
__device__ __forceinline__ void coop_finish(const lane_consts& c, const fe& X, const fe& Y, const fe& Z, uint32_t negate_x,
                                            uint8_t* out_enc, int32_t* out_ext, size_t i, uint4* proj = nullptr, size_t proj_stride = 0, size_t proj_offset = 0) {
  cq q = 0;
  q = quad_row_from_fe(c, q, 0, X);
  q = quad_row_from_fe(c, q, 1, Y);
  q = quad_row_from_fe(c, q, 2, Z);
  const cq inv = cinv(c, q);                                         // row 2 = 1/Z (0 when Z = 0: the reference's 0^(p-2))
  const cq zi = bperm(rowperm_idx(c, 2, 2, 2, 2), inv);
  const cq xy = cmul4(c, q, zi);                                     // rows 0, 1 = x, y
  fe x, y;
  fe_from_quad_row(c, x, xy, 0);
  fe_from_quad_row(c, y, xy, 1);
  fe nx;
  fe_neg(nx, x);
  fe_reduce_weak(nx, nx);
  fe_cmov(x, nx, negate_x);
  if (out_enc != nullptr) {
    uint32_t w[8];
    fe_to_words(w, y);
    w[7] ^= fe_is_negative(x) << 31;
    if (c.lane == 0) store_words8(out_enc, i, w);
  }
  if (out_ext != nullptr) {
    fe one, t;
    fe_one(one);
    fe_mul(t, x, y);
    if (c.lane == 0) store_ext(out_ext, i, x, y, one, t);
  }
  if (proj != nullptr) {
    fe one;
    fe_one(one);
    if (c.lane == 0) store_proj(proj, proj_stride, proj_offset + i, x, y, one);
  }
}

}  // namespace

// One ladder step on the state S = (x2, z2, x3, z3) (rows 0..3) with the base point's u = U1 / W1 kept projective
// (mont_ladder_proj, ge_ladder.h): UWQ holds U1 in row 0 and W1 in row 2.  `swap` = the pending conditional swap XOR this
// step's scalar bit (the swap only exchanges (a, b) with (c, d)).  Three cooperative multiplication levels.
struct ladder_idx { int I_0022, I_1133, I_F1, I_G1, I_2200, I_3311, I_1300, I_3333, I_0000, I_1120, x128; };
__device__ __forceinline__ ladder_idx ladder_idx_init(const lane_consts& c) {
  return ladder_idx{rowperm_idx(c, 0, 0, 2, 2), rowperm_idx(c, 1, 1, 3, 3), rowperm_idx(c, 0, 1, 3, 1), rowperm_idx(c, 0, 1, 0, 2), rowperm_idx(c, 2, 2, 0, 0),
                    rowperm_idx(c, 3, 3, 1, 1), rowperm_idx(c, 1, 3, 0, 0), rowperm_idx(c, 3, 3, 3, 3), rowperm_idx(c, 0, 0, 0, 0), rowperm_idx(c, 1, 1, 2, 0),
                    c.row < 2 ? 128 : 0};             // x128: rows 0,1 read rows 2,3 of the source when the swap is set
}
__device__ __forceinline__ cq coop_ladder_step(const lane_consts& c, const ladder_idx& li, cq S, cq UWQ, uint32_t swap) {
  const bool rodd = (c.row & 1u) != 0, r0 = c.row == 0, r1 = c.row == 1, r2 = c.row == 2, r3 = c.row == 3;
  const cq A24Q = (r3 && c.k == 0) ? 121665u : 0u;
  // level 0: a = x2 + z2, b = x2 - z2, c = x3 + z3, d = x3 - z3
  const cq U = bperm(li.I_0022, S), V = bperm(li.I_1133, S);
  const cq AB = cnorm(c, rodd ? csub(c, U, V) : cadd(U, V));                      // (a, b, c, d), tight
  // level 1: (aa, bb, da, cb) = (sa^2, sb^2, d*a, b*c) with (sa, sb) = swap ? (c, d) : (a, b)
  const int sx = (0 - (int)swap) & li.x128;
  const cq L1 = cmul4(c, bperm(li.I_F1 ^ sx, AB), bperm(li.I_G1 ^ sx, AB));
  // level 2: (s, t', x2', a24*e) = ((da+cb)^2, (da-cb)^2, aa*bb, e*a24),  e = aa - bb
  const cq W = bperm(li.I_2200, L1), Z = bperm(li.I_3311, L1);                    // (da, da, aa, aa), (cb, cb, bb, bb)
  const cq F2 = cnorm(c, rodd ? csub(c, W, Z) : (r2 ? W : cadd(W, Z)));            // (da+cb, da-cb, aa, e), tight
  const cq G2 = r3 ? A24Q : (r2 ? Z : F2);
  const cq L2 = cmul4(c, F2, G2);
  // level 3: (z3', z2', x3') = (t' * U1, e * (a24*e + aa), s * W1)
  const cq T3 = bperm(li.I_1300, L2);                                              // (t', a24*e, s, s)
  const cq E1 = bperm(li.I_3333, F2), A1 = bperm(li.I_0000, L1);                   // e, aa in every row
  const cq L3 = cmul4(c, r1 ? E1 : T3, r1 ? cadd(T3, A1) : UWQ);                   // row 3: * 0
  // new state (x2', z2', x3', z3') = (L2 row 2, L3 row 1, L3 row 2, L3 row 0)
  // (both cross-lane reads are issued by EVERY lane before the select: `cond ? bperm() : bperm()` would run each under a
  // partial EXEC mask, and a ds_bpermute that reads a disabled lane gets 0)
  const cq fromL3 = bperm(li.I_1120, L3), fromL2 = bperm(li.I_2200, L2);
  return r0 ? fromL2 : fromL3;
}

__global__ void __launch_bounds__(64)
k_mul_coop(const uint8_t* __restrict__ scalars, const int32_t* __restrict__ pts_ext, size_t n, uint8_t* __restrict__ out_enc, int32_t* __restrict__ out_ext,
           int skip_bits, uint4* __restrict__ proj, size_t proj_stride, size_t proj_offset) {
  const size_t i = blockIdx.x;
  if (i >= n) return;
  lane_consts c;
  lane_consts_init(c);

  // ---- replicated on all lanes: operands and their projective Montgomery image (ge_ladder.h) ----
  uint32_t a[8];
  load_words8(a, scalars, i);
  ge_p3 P;
  load_ext(P, pts_ext, i);
  mont_point_proj m;
  mont_prep_proj(m, P);                                                // u = U / W, v = V / W: no inversion in front of the ladder
  uint32_t neg, mag[8];
  sc_effective(neg, mag, a);

  // ---- the ladder: state S = (x2, z2, x3, z3) in rows 0..3 ----
  const cq UWQ = quad_row_from_fe(c, quad_row_from_fe(c, 0, 0, m.U), 2, m.W);      // U1 in row 0, W1 in row 2 (second operands of level 3)
  const cq ONE0 = (c.k == 0 && c.active) ? 1u : 0u;
  cq S = quad_row_from_fe(c, quad_row_from_fe(c, c.row == 0 ? ONE0 : 0u, 2, m.U), 3, m.W);      // (1, 0, U1, W1)
  const ladder_idx li = ladder_idx_init(c);
  const int I_own = (int)(c.lane << 2);
  uint32_t swap = 0;
#pragma unroll 1
  for (int w = 7; w >= 0; --w) {
    uint32_t word = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) word = (w == q) ? mag[q] : word;
    const int first = (w == 7) ? skip_bits : 0;
    word <<= first;
#pragma unroll 1
    for (int j = first; j < 32; ++j) {
      const uint32_t bit = word >> 31;
      word <<= 1;
      swap ^= bit;
      S = coop_ladder_step(c, li, S, UWQ, swap);
      swap = bit;
    }
  }
  S = bperm(I_own ^ ((0 - (int)swap) & 128), S);                       // final conditional swap (x2, z2) <-> (x3, z3)

  // ---- replicated: y-recovery and the exceptional cases (mont_recover_to_edwards), then the encoding ----
  fe x2, z2, x3, z3;
  fe_from_quad_row(c, x2, S, 0);
  fe_from_quad_row(c, z2, S, 1);
  fe_from_quad_row(c, x3, S, 2);
  fe_from_quad_row(c, z3, S, 3);
  ge_p2 r;
  mont_recover_to_edwards_proj(r, m, x2, z2, x3, z3, mag[0] & 1u, neg);
  coop_finish(c, r.X, r.Y, r.Z, 0u, out_enc, out_ext, i, proj, proj_stride, proj_offset);
}

// h + E for h = (X : Y : Z : T) in rows 0..3 and an affine table entry E = (y+x, y-x, 2dxy, 0): ge_madd followed by
// ge_p1p1_to_p3 (ge25519.h), two cooperative multiplication levels.
struct madd_idx { int I_1133, I_0000, I_2222, I_1122; };
__device__ __forceinline__ madd_idx madd_idx_init(const lane_consts& c) {
  return madd_idx{rowperm_idx(c, 1, 1, 3, 3), rowperm_idx(c, 0, 0, 0, 0), rowperm_idx(c, 2, 2, 2, 2), rowperm_idx(c, 1, 1, 2, 2)};
}
__device__ __forceinline__ cq coop_madd(const lane_consts& c, const madd_idx& mi, cq h, cq E) {
  const bool r0 = c.row == 0, r1 = c.row == 1, r2 = c.row == 2, r3 = c.row == 3;
  // a = Y + X, b = Y - X; A = a ypx, B = b ymx, C = xy2d T
  const cq U = bperm(mi.I_1133, h), V = bperm(mi.I_0000, h);         // (Y, Y, T, T), (X, X, X, X)
  const cq FA = cnorm(c, r0 ? cadd(U, V) : (r1 ? csub(c, U, V) : (r2 ? U : 0u)));
  const cq LA = cmul4(c, FA, E);                                       // (A, B, C, 0)
  // X3 = A - B, Y3 = A + B, Z3 = D + C, T3 = D - C with D = 2Z
  const cq H2 = cadd(h, h);
  const cq qa = bperm(mi.I_0000, LA), qd = bperm(mi.I_2222, H2);       // every lane issues both reads, then selects
  const cq Q1 = (c.row < 2) ? qa : qd;                                 // (A, A, D, D)
  const cq Q2 = bperm(mi.I_1122, LA);                                  // (B, B, C, C)
  const cq SUM = cadd(Q1, Q2), DIF = csub(c, Q1, Q2);                  // (Y3, Y3, Z3, Z3), (X3, X3, T3, T3)
  // (X3 T3, Z3 Y3, Z3 T3, X3 Y3)
  const cq x3 = bperm(mi.I_0000, DIF), z3 = bperm(mi.I_2222, SUM), t3 = bperm(mi.I_2222, DIF), y3 = bperm(mi.I_0000, SUM);
  const cq FB = cnorm(c, (r0 || r3) ? x3 : z3);                        // (X3, Z3, Z3, X3)
  const cq GB = (r0 || r2) ? t3 : y3;                                  // (T3, Y3, T3, Y3)
  return cmul4(c, FB, GB);
}

// One entry of the radix-64 image for the cooperative layout: lane (row g < 3, limb r) gets word 10 g + r of entry idx of
// the window at `win` (E entries).  All LINES of a limb's entries are read whatever idx is (four loads 8 entries apart, the
// wanted one kept by a uniform select): the cache sees the same lines for every digit.
template <int E>
__device__ __forceinline__ cq coop_table_entry(const lane_consts& c, const uint32_t* __restrict__ win, uint32_t idx, uint32_t negate) {
  const uint32_t g = c.row < 3 ? c.row : 0u;
  const uint32_t g_eff = (g < 2u) ? (g ^ negate) : g;                 // a negated entry is (ymx, ypx, -xy2d)
  const uint32_t kk = c.active ? c.k : 0u;
  uint32_t v = 0;
  const uint32_t want = idx >> 3;
#pragma unroll
  for (uint32_t t = 0; t < (uint32_t)(E / 8); ++t) {
    const uint32_t j = (idx & 7u) | (t << 3);
    const uint32_t word = win[kyb_bt64_in_win(E, (int)j, (int)(10u * g_eff + kk))];
    v = (t == want) ? word : v;
  }
  v = (c.active && c.row < 3) ? v : 0u;
  const uint32_t nv = c.p2 - v;                                        // 2p - xy2d
  return (c.row == 2 && negate) ? nv : v;
}

__global__ void __launch_bounds__(64)
k_mul_base_coop(const uint8_t* __restrict__ scalars, const uint8_t* __restrict__ scalars_b, size_t n_a, size_t n, uint8_t* __restrict__ out_enc,
                int32_t* __restrict__ out_ext, const uint32_t* __restrict__ image64, uint4* __restrict__ proj, size_t proj_stride, size_t proj_offset) {
  const size_t i = blockIdx.x;
  if (i >= n) return;
  lane_consts c;
  lane_consts_init(c);
  const bool r1 = c.row == 1, r2 = c.row == 2;
  uint32_t a[8];
  if (i < n_a) load_words8(a, scalars, i); else load_words8(a, scalars_b, i - n_a);      // two arrays in one launch (signing: nonces, then keys)
  sc_digits64 dg;
  sc_recode64(dg, a);
  const cq ONE0 = (c.k == 0 && c.active) ? 1u : 0u;
  cq h = (r1 || r2) ? ONE0 : 0u;                                       // neutral element (0 : 1 : 1 : 0)
  const madd_idx mi = madd_idx_init(c);
  auto madd = [&](cq E) { h = coop_madd(c, mi, h, E); };
#pragma unroll 1
  for (int pos = 0; pos < KYB_BASE64_POS - 1; ++pos) {
    uint32_t idx, neg;
    sc_next_digit64(idx, neg, dg, false);
    madd(coop_table_entry<32>(c, image64 + pos * KYB_BASE64_WIN_WORDS, idx, neg));
  }
  {
    uint32_t idx, neg;
    sc_next_digit64(idx, neg, dg, true);
    madd(coop_table_entry<16>(c, image64 + KYB_BASE64_TOP_BASE, idx, 0u));
  }
  fe X, Y, Z;
  fe_from_quad_row(c, X, h, 0);
  fe_from_quad_row(c, Y, h, 1);
  fe_from_quad_row(c, Z, h, 2);
  coop_finish(c, X, Y, Z, dg.neg, out_enc, out_ext, i, proj, proj_stride, proj_offset);
}

// Test hook (tests/test_gpu_coop.py, against the lane-level model tools/coop_model.py): one wavefront applies ONE
// cooperative primitive to caller-supplied quads.  op: 0 cmul4(A, B), 1 cnorm(A), 2 cinv(A), 3 mixed addition h = A, entry = B,
// 4 table entry (window, idx, negate) = (B[0], B[1], B[2]) of the radix-64 image, 5 csub(A, B), 6 one ladder step S = A,
// U1 / W1 in rows 0 / 2 of B, swap / bit in B lanes 16 / 17 (returns S'), 7 quad -> fe -> quad round trip of every row.
__global__ void __launch_bounds__(64)
k_coop_selftest(int op, const uint32_t* __restrict__ A, const uint32_t* __restrict__ B, uint32_t* __restrict__ out, const uint32_t* __restrict__ image64) {
  lane_consts c;
  lane_consts_init(c);
  const cq a = A[c.lane], b = B[c.lane];
  cq r = 0;
  if (op == 0) r = cmul4(c, a, b);
  else if (op == 1) r = cnorm(c, a);
  else if (op == 2) r = cinv(c, a);
  else if (op == 5) r = csub(c, a, b);
  else if (op == 4) {
    const uint32_t pos = B[0], idx = B[1], neg = B[2];
    r = pos < 42 ? coop_table_entry<32>(c, image64 + pos * KYB_BASE64_WIN_WORDS, idx, neg) : coop_table_entry<16>(c, image64 + KYB_BASE64_TOP_BASE, idx, 0u);
  } else if (op == 3) {
    r = coop_madd(c, madd_idx_init(c), a, b);
  } else if (op == 6) {
    const uint32_t swap = B[16] ^ B[17];                              // pending swap XOR this step's bit
    r = coop_ladder_step(c, ladder_idx_init(c), a, ((c.row == 0 || c.row == 2) && c.active) ? b : 0u, swap);
  } else if (op == 7) {
    fe f[4];
    for (uint32_t q = 0; q < 4; ++q) fe_from_quad_row(c, f[q], a, q);
    for (uint32_t q = 0; q < 4; ++q) r = quad_row_from_fe(c, r, q, f[q]);
  }
  out[c.lane] = r;
}

namespace kyb { namespace launch {
hipError_t coop_selftest(hipStream_t st, int op, const uint32_t* A, const uint32_t* B, uint32_t* out, const uint32_t* image64) {
  hipLaunchKernelGGL(k_coop_selftest, dim3(1), dim3(64), 0, st, op, A, B, out, image64);
  return hipGetLastError();
}
hipError_t mul_coop(hipStream_t st, const uint8_t* sc, const int32_t* pext, size_t n, uint8_t* oenc, int32_t* oext, int skip_bits,
                    uint4* proj, size_t proj_stride, size_t proj_offset) {
  hipLaunchKernelGGL(k_mul_coop, dim3((unsigned)n), dim3(64), 0, st, sc, pext, n, oenc, oext, skip_bits, proj, proj_stride, proj_offset);
  return hipGetLastError();
}
hipError_t mul_base_coop(hipStream_t st, const uint8_t* sc, size_t n, uint8_t* oenc, int32_t* oext, const uint32_t* image64,
                         uint4* proj, size_t proj_stride, size_t proj_offset, const uint8_t* sc_b, size_t n_b) {
  hipLaunchKernelGGL(k_mul_base_coop, dim3((unsigned)(n + n_b)), dim3(64), 0, st, sc, sc_b, n, n + n_b, oenc, oext, image64, proj, proj_stride, proj_offset);
  return hipGetLastError();
}
}}  // namespace kyb::launch
