// Small-batch (latency) kernels of the MI355X Ed25519 engine: ONE ITEM PER WAVEFRONT, field elements spread over lanes
// (coop25519.h).  One of the translation units of the library (map: launch.h).
//   k_mul_coop        Point::mul(s, Some(P))  ge.rs:508-568   the Montgomery ladder of k_mul_ladder with a PROJECTIVE base point
//                     (mont_ladder_proj, ge_ladder.h: no inversion in front), its 256 steps as three cooperative multiplication
//                     levels each; operand image, y-recovery with its exceptional cases and the field inversion of the encoding
//                     cooperative as well (only the canonical byte encoding runs replicated).  One launch does the whole multiplication.
//   k_mul_base_coop   Point::mul(s, None)     ge.rs:442-486   the radix-64 table of k_mul_base64 read from global memory (every
//                     line of a limb's 32 entries is touched whatever the digit), 43 cooperative mixed additions, cooperative inversion.
//   k_sum_coop        short sums of points (kyb_sum_batch, the tail of a linear combination), one group per wavefront
//   k_finish_coop     marshal_binary / the finish of a projective staging record, one point per wavefront (cooperative inversion)
//   k_finish_wave     the same for mid-size batches: one point per lane, one cooperative inversion per wavefront (Montgomery's trick across the lanes)
//   k_decode_coop     unmarshal_binary          ge.rs:124-179   ge_decode replicated on all lanes, its square-root chain
//                     (252 of ~270 dependent multiplications) cooperative
//   k_verify_prep_coop / k_verify_prep_r_coop   the two front halves of a verification (verify.h) with that decode
//   k_sign_coop       schnorr::sign in one launch, eight wavefronts per signature: k B | x B (a quarter of the windows each), then hash and s = k + x h
//   k_mul_enc_coop    Point::mul on a wire encoding, two wavefronts per item: ladder on y alone | square root for x
//   k_verify_coop     one verification per workgroup of three wavefronts: hash + ladder | both decodes | s B, one barrier, one launch
//   k_poly_eval_seg / k_poly_eval_sum   the same for a long polynomial: up to 32 wavefronts per evaluation, segments combined by x^(s len) mod 8L
//   k_poly_eval_coop  PubPoly::eval             poly.rs:457-469 one evaluation per wavefront: Horner with cooperative doublings / additions
// Used for batches that leave the chip idle (engine.hip: `coop.max_items`); results are bit-identical to the batch kernels.
#include <hip/hip_runtime.h>
#include "launch.h"
#include "ge_scalarmult.h"
#include "ge_ladder.h"
#include "verify.h"
#include "sc25519.h"
#include "coop25519.h"
using namespace kyb;
using namespace kyb::coop;
#include "device_tables.h"

// per-lane constants and this wavefront's LDS scratch (NW = wavefronts per workgroup of the kernel)
#define KYB_COOP_CONSTS(c, NW) \
  __shared__ uint32_t coop_lds_[(NW) * KYB_COOP_LDS_WORDS]; \
  lane_consts c; \
  lane_consts_init(c, coop_lds_ + (threadIdx.x >> 6) * KYB_COOP_LDS_WORDS)

namespace {

// Phase stamps of the one-item kernels (CROSS-CHECK build only; tools/one_item_stamps.py, profiles/r06/one_item_stamps.log): while a buffer is
// set (kyb_diag_phase_stamps), lane 0 of the named wavefront of workgroup 0..3 writes the constant 100 MHz clock into its slot at the phase
// boundaries below — where the time of a one-item call goes, measured inside the kernel.  The product build compiles none of it.
#ifdef KYB_CROSSCHECK
static __device__ uint64_t* kyb_phase_buf = nullptr;
__device__ __forceinline__ void phase_stamp(int slot, bool mine) {
  uint64_t* b = *reinterpret_cast<uint64_t* const volatile*>(&kyb_phase_buf);
  if (b == nullptr) return;
  uint64_t rt;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(rt) :: "memory");
  if (mine && (threadIdx.x & 63u) == 0) b[slot] = rt;
}
#define KYB_PHASE(slot, mine) phase_stamp((slot), (mine))
#else
#define KYB_PHASE(slot, mine) ((void)0)
#endif

// ---- ONE field inversion by a whole wavefront (round 6) ----------------------------------------------------------------------------------
// The one-item kernels end in Z^-1, and a lone wavefront issues one instruction per ~5.3 cycles on whichever unit: the 254 cooperative squarings +
// 11 products of cinv (coop25519.h) are ~12,000 instructions, 30.6 us of a 41 us one-item fixed-base multiplication
// (profiles/r06/one_item_stamps.log).  Here: Bernstein-Yang safegcd (fe_invert_gcd.h: 20 batches of 30 divsteps, 2x2 matrix per batch) laid out over
// the lanes —
//   * the 30 divsteps of a batch in THREE lane roles (lane & 3 = 0: (f, g) low words, 1: (u, q), 2: (v, r)): the three updates of a divstep are one
//     vector instruction each, the decision bits uniform (g's low bit read from lane 0, zeta in scalar registers): 13 instructions per divstep
//     where one lane needs ~22;
//   * the nine 30-bit limbs of d, e, f, g in lanes 0..8 of every 16-lane row: the matrix step is two multiply-adds per lane and quantity, the
//     division by 2^30 a limb shift by DPP with two short carry passes (limbs stay LOOSELY normalised, in [-1, 2^30 + 1]: the products have the
//     room, the low 30 bits the next divsteps look at are exact, and an exact carry chain runs once at the end);
// ~10,600 instructions: fixed base + encoding 55.4 -> 49.7 us, encode of a projective point 46.4 -> 40.2, variable base 164.3 -> 159.6, signing
// 92.8 -> 90.3 (profiles/r06/ab_coop_inv_gcd.log).  Same result as cinv / fe_invert bit for bit (the inverse is unique; 0 -> 0).  Constant time:
// 600 divsteps whatever the input, every selection a mask, no table, no branch (tools/ct_check.py covers the kernels that use it).
// tools/gcd_lanes_model.py is the limb-exact model (every intermediate in the width used here, against x^(p-2)); the cross-check build runs it
// beside cinv on the device (k_coop_selftest op 9, tests/test_gpu_coop.py).
// ---- the nine 30-bit limbs of d, e, f, g in lanes 0..8 of every 16-lane row (lanes 9..15 hold 0) ---------------------------------------------
// One 64-bit column value per lane -> the number divided by 2^30, limb k again in lane k: x_k = lo_k + mid_k 2^30 + top_k 2^60, so the new limb k is
// lo_(k+1) + mid_k + top_(k-1); two short carry passes leave limbs 0..7 in [-1, 2^30 + 1] (LOOSELY normalised: the next products have the room, the
// low 30 bits the divsteps look at are exact) and everything above in the signed limb 8.
struct gcd_lane_consts { uint32_t k; uint32_t mk30, cmask, live; int32_t mod; };
__device__ __forceinline__ int32_t gcd_shr1(int32_t v) { return __builtin_amdgcn_update_dpp(0, v, KYB_DPP_ROW_SHR(1), 0xf, 0xf, true); }      // lane k reads lane k-1
__device__ __forceinline__ int32_t gcd_shl1(int32_t v) { return __builtin_amdgcn_update_dpp(0, v, KYB_DPP_ROW_SHL(1), 0xf, 0xf, true); }      // lane k reads lane k+1
__device__ __forceinline__ int32_t gcd_lane_div30(const gcd_lane_consts& L, int64_t x) {
  const int32_t lo = (int32_t)((uint32_t)x & KYB_GCD_M30);
  const int32_t mid = (int32_t)((uint32_t)(x >> 30) & L.mk30);           // lane 8: everything above bit 30 (|x_8| < 2^48), signed
  const int32_t top = (int32_t)(x >> 60);                                // -4 .. 3  (lane 8's goes to lane 9, which is cleared)
  const int32_t t1 = gcd_shl1(lo) + gcd_shr1(top);                       // [-4, 2^30 + 2]
  const int32_t c1 = (t1 >> 30) & (int32_t)L.cmask;                      // (no carry out of lane 8)
  const int32_t r1 = (int32_t)((uint32_t)t1 & L.mk30);
  const int32_t t2 = r1 + mid;                                           // lanes 0..7: [0, 2^31 - 2]
  const int32_t c2 = (int32_t)(((uint32_t)t2 >> 30) & L.cmask);
  const int32_t r2 = (int32_t)((uint32_t)t2 & L.mk30);
  return (r2 + gcd_shr1(c1 + c2)) & (int32_t)L.live;
}
// 30 divsteps on the low words in three lane roles, the matrix as uniform values (gcd_divsteps_30 of fe_invert_gcd.h is the one-lane form)
__device__ __forceinline__ int32_t gcd_divsteps_30_roles(int32_t zeta, uint32_t f0, uint32_t g0, int32_t& mu, int32_t& mv, int32_t& mq, int32_t& mr) {
  const uint32_t role = threadIdx.x & 3u;
  uint32_t F = role == 0 ? f0 : (role == 1 ? 1u : 0u);
  uint32_t G = role == 0 ? g0 : (role == 1 ? 0u : 1u);
  const uint32_t shG = role == 0 ? 1u : 0u, shF = role == 0 ? 0u : 1u;
#pragma unroll
  for (int i = 0; i < 30; ++i) {
    const uint32_t g_lo = (uint32_t)__builtin_amdgcn_readlane((int)G, 0);
    const uint32_t c2 = 0u - (g_lo & 1u);
    uint32_t c1 = (uint32_t)(zeta >> 31);
    const uint32_t x = (F ^ c1) - c1;
    G += x & c2;
    c1 &= c2;
    asm volatile("" : "+s"(c1));                                     // the combined mask stays a scalar operand (else c2 is copied into a vector register: 14 -> 13 instructions)
    zeta = (int32_t)((uint32_t)zeta ^ c1) - 1;
    F += G & c1;
    G >>= shG; F <<= shF;
  }
  mu = __builtin_amdgcn_readlane((int)F, 1); mv = __builtin_amdgcn_readlane((int)F, 2);
  mq = __builtin_amdgcn_readlane((int)G, 1); mr = __builtin_amdgcn_readlane((int)G, 2);
  return zeta;
}
__device__ __forceinline__ void gcd_carry_chain(int32_t r[9]) { KYB_UNROLL for (int i = 0; i < 8; ++i) { r[i + 1] += r[i] >> 30; r[i] &= KYB_GCD_M30; } }
__device__ __forceinline__ void fe_invert_gcd_wave(fe& h, const fe& z) {
  uint32_t w[8];
  fe_to_words(w, z);
  int32_t g0[9];
  g0[0] = (int32_t)(w[0] & KYB_GCD_M30);
  KYB_UNROLL for (int i = 1; i < 8; ++i) g0[i] = (int32_t)(((w[i - 1] >> (32 - 2 * i)) | (w[i] << (2 * i))) & KYB_GCD_M30);
  g0[8] = (int32_t)(w[7] >> 16);
  const int32_t modv[9] = KYB_GCD_MOD;
  gcd_lane_consts L;
  L.k = threadIdx.x & 15u;
  L.mk30 = L.k < 8u ? KYB_GCD_M30 : 0xffffffffu;
  L.cmask = L.k < 8u ? 0xffffffffu : 0u;
  L.live = L.k <= 8u ? 0xffffffffu : 0u;
  int32_t Gv = 0, Fv = 0;
  KYB_UNROLL for (int i = 0; i < 9; ++i) { Gv = L.k == (uint32_t)i ? g0[i] : Gv; Fv = L.k == (uint32_t)i ? modv[i] : Fv; }
  L.mod = Fv;
  int32_t D = 0, E = L.k == 0u ? 1 : 0;
  int32_t zeta = -1;
#pragma unroll 1
  for (int it = 0; it < 20; ++it) {
    int32_t u, v, q, r;
    zeta = gcd_divsteps_30_roles(zeta, (uint32_t)__builtin_amdgcn_readlane(Fv, 0), (uint32_t)__builtin_amdgcn_readlane(Gv, 0), u, v, q, r);
    // (d, e) <- (u d + v e + md p, q d + r e + me p) / 2^30: md, me clear the low 30 bits (gcd_update_de)
    const int64_t bd = (int64_t)u * D + (int64_t)v * E, be = (int64_t)q * D + (int64_t)r * E;
    const int32_t sd = __builtin_amdgcn_readlane(D, 8) >> 31, se = __builtin_amdgcn_readlane(E, 8) >> 31;
    int32_t md = (u & sd) + (v & se), me = (q & sd) + (r & se);
    md -= (int32_t)((KYB_GCD_MODINV30 * (uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)bd, 0) + (uint32_t)md) & KYB_GCD_M30);
    me -= (int32_t)((KYB_GCD_MODINV30 * (uint32_t)__builtin_amdgcn_readlane((int32_t)(uint32_t)be, 0) + (uint32_t)me) & KYB_GCD_M30);
    const int32_t nD = gcd_lane_div30(L, bd + (int64_t)L.mod * md);
    const int32_t nE = gcd_lane_div30(L, be + (int64_t)L.mod * me);
    // (f, g) <- (u f + v g, q f + r g) / 2^30, exact
    const int32_t nF = gcd_lane_div30(L, (int64_t)u * Fv + (int64_t)v * Gv);
    const int32_t nG = gcd_lane_div30(L, (int64_t)q * Fv + (int64_t)r * Gv);
    D = nD; E = nE; Fv = nF; Gv = nG;
  }
  int32_t d[9], f[9];
  KYB_UNROLL for (int i = 0; i < 9; ++i) { d[i] = __builtin_amdgcn_readlane(D, i); f[i] = __builtin_amdgcn_readlane(Fv, i); }
  gcd_carry_chain(f);                                                    // f = +-1 (+-p for z = 0): its sign from EXACT limbs
  // d (|d| < 2p up to the loose limbs' epsilon; anything below 8p is handled) -> sign(f) d + 8p in (0, 16p) -> a field element:
  // 8p = 2^258 - 152 in 30-bit limbs; what lies above bit 255 folds back as 19 (bit 255) and 38 (bits 256..258)
  const int32_t neg = f[8] >> 31;
  KYB_UNROLL for (int i = 0; i < 9; ++i) d[i] = (d[i] ^ neg) - neg;
  const int32_t p8[9] = {0x3fffff68, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3fffffff, 0x3ffff};
  KYB_UNROLL for (int i = 0; i < 9; ++i) d[i] += p8[i];
  gcd_carry_chain(d);
  const uint32_t above = (uint32_t)d[8] >> 16;                           // bits 256.. : 0..7
  d[8] &= 0xffff;
  uint32_t o[8];
  KYB_UNROLL for (int i = 0; i < 8; ++i) o[i] = ((uint32_t)d[i] >> (2 * i)) | ((uint32_t)d[i + 1] << (30 - 2 * i));
  fe_from_words(h, o);                                                   // (ignores bit 255)
  h.v[0] += 19u * (o[7] >> 31) + 38u * above;                            // <= 285 above the mask: tight
}
// affine (x, y) of the point (X : Y : Z) in rows 0..2 of a tight quad, in every lane; Z^-1 by the whole wavefront (0 when Z = 0: the reference's 0^(p-2))
__device__ __forceinline__ void coop_affine(const lane_consts& c, cq q, fe& x, fe& y) {
  fe Z, zinv;
  fe_from_quad_row(c, Z, q, 2);
  fe_invert_gcd_wave(zinv, Z);
  uint32_t v = 0;
  KYB_UNROLL for (int j = 0; j < 10; ++j) v = (c.k == (uint32_t)j) ? zinv.v[j] : v;
  const cq zi = c.active ? v : 0u;                                   // Z^-1 in every row
  const cq xy = cmul4(c, q, zi);                                     // rows 0, 1 = x, y
  fe_from_quad_row(c, x, xy, 0);
  fe_from_quad_row(c, y, xy, 1);
}
// encode (and optionally affine limbs) from a point (X : Y : Z) in rows 0..2 of a tight quad; Z^-1 computed cooperatively.
// proj != nullptr: the point also goes to staging record proj_offset + i for k_verify_final — affine (Z = 1) next to an
// encoding or limbs, as it is (projective, no inversion at all) when it is the only output.
// ext_proj (option ext.projective): a call that only asks for extended limbs gets the point as it is, (X : Y : Z : T) with Z != 1 —
// what the reference's own Point holds after a multiplication — and the 265-multiplication inversion waits for a marshal_binary;
// has_t: row 3 of q already holds T = X Y / Z.
__device__ __forceinline__ void coop_finish(const lane_consts& c, cq q, uint32_t negate_x,
                                            uint8_t* out_enc, int32_t* out_ext, size_t i, uint4* proj = nullptr, size_t proj_stride = 0, size_t proj_offset = 0,
                                            bool ext_proj = false, bool has_t = false, bool z_is_one = false) {
  if (ext_proj && out_enc == nullptr && out_ext != nullptr && proj == nullptr) {
    if (!has_t) {                                                        // (X : Y : Z) -> (X Z : Y Z : Z^2 : X Y)
      const cq zz = bperm(rowperm_idx(c, 2, 2, 2, 2), q), xy = bperm(rowperm_idx(c, 0, 1, 2, 0), q), yy = bperm(rowperm_idx(c, 1, 1, 1, 1), q);
      q = cmul4(c, xy, c.row == 3 ? yy : zz);
    }
    const cq ng = cnorm(c, c.p2 - q);                                    // rows 0, 3: -X, -T
    q = csel((uint32_t)(c.row == 0 || c.row == 3) & negate_x, ng, q);
    fe X, Y, Z, T;
    fe_from_quad_row(c, X, q, 0); fe_from_quad_row(c, Y, q, 1); fe_from_quad_row(c, Z, q, 2); fe_from_quad_row(c, T, q, 3);
    if (c.lane == 0) store_ext(out_ext, i, X, Y, Z, T);
    return;
  }
  const cq nq = cnorm(c, c.p2 - q);                                    // row 0: -X
  q = csel((uint32_t)(c.row == 0) & negate_x, nq, q);
  if (out_enc == nullptr && out_ext == nullptr) {
    fe X, Y, Z;
    fe_from_quad_row(c, X, q, 0);
    fe_from_quad_row(c, Y, q, 1);
    fe_from_quad_row(c, Z, q, 2);
    if (proj != nullptr && c.lane == 0) store_proj(proj, proj_stride, proj_offset + i, X, Y, Z);
    return;
  }
  fe x, y;
  if (z_is_one) { fe_from_quad_row(c, x, q, 0); fe_from_quad_row(c, y, q, 1); }      // (wave-uniform: see k_finish_coop)
  else { KYB_PHASE(30, blockIdx.x < 4); coop_affine(c, q, x, y); KYB_PHASE(31, blockIdx.x < 4); }      // (slots 30 / 31: the inversion of whichever workgroup finished last)
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
// 1 iff row r of a tight quad is 0 mod p (the row fetched into every lane, then fe_is_nonzero's canonical test)
__device__ __forceinline__ uint32_t coop_row_is_zero(const lane_consts& c, cq q, uint32_t r) {
  fe f;
  fe_from_quad_row(c, f, q, r);
  return 1u - fe_is_nonzero(f);
}

// ge_decode (ge25519.h) with z^((p-5)/8) computed cooperatively (row 0 of a quad)
struct coop_decode_fn {
  const lane_consts& c;
  __device__ __forceinline__ uint32_t operator()(ge_p3& P, const uint32_t w[8]) const {
    const lane_consts& lc = c;
    return ge_decode_with(P, w, [&lc](fe& o, const fe& z) { fe_from_quad_row(lc, o, cpow22523(lc, quad_row_from_fe(lc, 0u, 0, z)), 0); });
  }
};

}  // namespace

// One ladder step with the base point's u = U1 / W1 kept projective (mont_ladder_proj, ge_ladder.h).  The state is held as two
// quads with every coordinate twice, SX = (x2, x2, x3, x3) and SZ = (z2, z2, z3, z3): the additions in front of the first
// multiplication level then need no cross-row traffic.  UWQ holds U1 in row 0 and W1 in row 2.  `swap` = the pending
// conditional swap XOR this step's scalar bit (the swap only exchanges (a, b) with (c, d)).  Three cooperative multiplication
// levels.
struct ladder_idx { int I_F1, I_G1, I_2200, I_3311, I_1300, I_3333, I_0000, I_2222, I_1100, x128; };
struct ladder_state { cq SX, SZ; };
__device__ __forceinline__ ladder_idx ladder_idx_init(const lane_consts& c) {
  return ladder_idx{rowperm_idx(c, 0, 1, 3, 1), rowperm_idx(c, 0, 1, 0, 2), rowperm_idx(c, 2, 2, 0, 0), rowperm_idx(c, 3, 3, 1, 1), rowperm_idx(c, 1, 3, 0, 0),
                    rowperm_idx(c, 3, 3, 3, 3), rowperm_idx(c, 0, 0, 0, 0), rowperm_idx(c, 2, 2, 2, 2), rowperm_idx(c, 1, 1, 0, 0),
                    c.row < 2 ? 128 : 0};             // x128: rows 0,1 read rows 2,3 of the source when the swap is set
}
__device__ __forceinline__ ladder_state coop_ladder_step(const lane_consts& c, const ladder_idx& li, ladder_state st, cq UWQ, uint32_t swap) {
  const bool rodd = (c.row & 1u) != 0, r1 = c.row == 1, r2 = c.row == 2, r3 = c.row == 3;
  const cq A24Q = (r3 && c.k == 0) ? 121665u : 0u;
  // level 0: (a, b, c, d) = (x2 + z2, x2 - z2, x3 + z3, x3 - z3)
  const cq ABraw = rodd ? csub(c, st.SX, st.SZ) : cadd(st.SX, st.SZ);              // <= 3T: good enough for a second operand
  const cq AB = cnorm(c, ABraw);
  // level 1: (aa, bb, da, cb) = (sa^2, sb^2, d*a, b*c) with (sa, sb) = swap ? (c, d) : (a, b)
  const int sx = (0 - (int)swap) & li.x128;
  const cq L1 = cmul4(c, bperm(li.I_F1 ^ sx, AB), bperm(li.I_G1 ^ sx, ABraw));
  // (row moves by permlane swaps, as in coop_dbl — with several wavefronts per SIMD the LDS crossbar was the bottleneck: 2,048 items 334 -> 299 us,
  //  profiles/r06/ab_permlane_rows.log; only level 1's operands keep the ds_bpermute, whose lane index carries the conditional swap)
  const bool rlo = c.row < 2;
  // level 2: (s, t', x2', a24*e) = ((da+cb)^2, (da-cb)^2, aa*bb, e*a24),  e = aa - bb
  const auto l16 = __builtin_amdgcn_permlane16_swap(L1, L1, false, false);         // (aa, aa, da, da), (bb, bb, cb, cb)
  const auto le = __builtin_amdgcn_permlane32_swap(l16[0], l16[0], false, false);  // aa x 4, da x 4
  const auto lo = __builtin_amdgcn_permlane32_swap(l16[1], l16[1], false, false);  // bb x 4, cb x 4
  const cq W = rlo ? le[1] : le[0], Z = rlo ? lo[1] : lo[0];                       // (da, da, aa, aa), (cb, cb, bb, bb)
  const cq A1 = le[0];                                                             // aa in every row
  const cq F2raw = rodd ? csub(c, W, Z) : (r2 ? W : cadd(W, Z));                   // (da+cb, da-cb, aa, e), <= 3T
  const cq F2 = cnorm(c, F2raw);
  const auto f16 = __builtin_amdgcn_permlane16_swap(F2, F2, false, false);         // [1] = (F2.1, F2.1, e, e)
  const auto fo = __builtin_amdgcn_permlane32_swap(f16[1], f16[1], false, false);  // [1] = e x 4
  const cq E1 = fo[1];
  const cq L2 = cmul4(c, F2, r3 ? A24Q : (r2 ? Z : F2raw));
  // level 3: (z3', z2', x3') = (t' * U1, (a24*e + aa) * e, s * W1)
  const auto m16 = __builtin_amdgcn_permlane16_swap(L2, L2, false, false);         // (s, s, x2', x2'), (t', t', a24e, a24e)
  const auto me = __builtin_amdgcn_permlane32_swap(m16[0], m16[0], false, false);  // s x 4, x2' x 4
  const auto mo = __builtin_amdgcn_permlane32_swap(m16[1], m16[1], false, false);  // t' x 4, a24e x 4
  const cq T3 = c.row == 0 ? mo[0] : (r1 ? mo[1] : me[0]);                         // (t', a24*e, s, s)
  const cq X2 = me[1];                                                             // x2' in every row
  const cq F3 = r1 ? cnorm(c, cadd(T3, A1)) : T3;                                  // row 3: * 0
  const cq L3 = cmul4(c, F3, r1 ? E1 : UWQ);
  // new state: SX = (x2', x2', x3', x3') = (L2 row 2 twice, L3 row 2 twice), SZ = (z2', z2', z3', z3') = (L3 rows 1, 1, 0, 0)
  const auto n16 = __builtin_amdgcn_permlane16_swap(L3, L3, false, false);         // (z3', z3', x3', x3'), (z2', z2', ., .)
  const auto ne = __builtin_amdgcn_permlane32_swap(n16[0], n16[0], false, false);  // z3' x 4, x3' x 4
  const auto no = __builtin_amdgcn_permlane32_swap(n16[1], n16[1], false, false);  // z2' x 4
  return ladder_state{rlo ? X2 : ne[1], rlo ? no[0] : ne[0]};
}

// Head of k_mul_coop in quads (mont_prep_proj, ge_ladder.h): PQ = (X, Y, Z, T) -> M = (U, V, W, 0), the projective Montgomery
// image u = U / W, v = V / W with U = (Z+Y) X, V = c (Z+Y) Z, W = (Z-Y) X; flags as mont_point.  Two multiplication levels.
__device__ __forceinline__ cq coop_mont_prep(const lane_consts& c, cq PQ, uint32_t& flags) {
  const bool r0 = c.row == 0, r1 = c.row == 1, r2 = c.row == 2, r3 = c.row == 3;
  const uint32_t ccv[10] = KYB_FE_MONT_C;
  uint32_t cck = 0;
  KYB_UNROLL for (int j = 0; j < 10; ++j) cck = (c.k == (uint32_t)j) ? ccv[j] : cck;
  const cq ZZ = bperm(rowperm_idx(c, 2, 2, 2, 2), PQ), YY = bperm(rowperm_idx(c, 1, 1, 1, 1), PQ), XX = bperm(rowperm_idx(c, 0, 0, 0, 0), PQ);
  const cq Fp = cnorm(c, r0 ? csub(c, ZZ, YY) : cadd(ZZ, YY));           // (Z-Y, Z+Y, Z+Y, Z+Y)
  const cq M1 = cmul4(c, Fp, r2 ? ZZ : XX);                                // (W, U, (Z+Y) Z, -)
  const cq M2 = cmul4(c, M1, (r2 && c.active) ? cck : 0u);                 // row 2 = V
  const cq a = bperm(rowperm_idx(c, 1, 1, 0, 0), M1), b = bperm(rowperm_idx(c, 2, 2, 2, 2), M2);
  cq M = r1 ? b : (r3 ? 0u : a);                                           // (U, V, W, 0)
  const uint32_t x0 = coop_row_is_zero(c, PQ, 0);
  const uint32_t id = x0 & coop_row_is_zero(c, Fp, 0);
  const uint32_t o2 = x0 & coop_row_is_zero(c, Fp, 1);
  const uint32_t degenerate = coop_row_is_zero(c, M1, 0);                  // X == 0, or Z == Y with X != 0 (not on the curve): neutral element
  flags = (id | (degenerate & (1u - o2))) | (o2 << 1);
  const cq ONE0 = (c.k == 0 && c.active) ? 1u : 0u;
  M = csel(degenerate, r3 ? 0u : ONE0, M);
  return M;
}

// Tail of k_mul_coop in quads (mont_recover_to_edwards_proj, ge_ladder.h; lane-level model: tools/coop_model.py recover_quads):
// M = (U, V, W, 0), SX = (x2, x2, x3, x3), SZ = (z2, z2, z3, z3) after the ladder's last swap.  Okeya-Sakurai y-recovery
// scaled by W^2, the map back to Edwards and -P (for the result -P the formulas degenerate), eight multiplication levels
// where the one-lane form has 27 dependent multiplications; then the exceptional cases by uniform selects.  Returns (X : Y : Z).
__device__ __forceinline__ cq coop_mont_recover(const lane_consts& c, cq M, cq SX, cq SZ, uint32_t p_flags, uint32_t k_is_odd, uint32_t negate) {
  const bool r0 = c.row == 0, r1 = c.row == 1, r2 = c.row == 2, r3 = c.row == 3;
  const int I0 = rowperm_idx(c, 0, 0, 0, 0), I1 = rowperm_idx(c, 1, 1, 1, 1), I2 = rowperm_idx(c, 2, 2, 2, 2), I3 = rowperm_idx(c, 3, 3, 3, 3);
  const uint32_t ccv[10] = KYB_FE_MONT_C;
  uint32_t cck = 0;
  KYB_UNROLL for (int j = 0; j < 10; ++j) cck = (c.k == (uint32_t)j) ? ccv[j] : cck;
  const cq CCQ = c.active ? cck : 0u;
  const cq ONE0 = (c.k == 0 && c.active) ? 1u : 0u;
  const cq K2A = ONE0 * (2u * 486662u);
  const cq z2a = bperm(I0, SZ), x2a = bperm(I0, SX);
  const cq mu = bperm(I0, M), mv = bperm(I1, M), mw = bperm(I2, M);
  const cq up1 = cadd(mu, mw), um1 = csub(c, mu, mw);
  // level 1: (T1, Wx2, UX, Wz2) = (U z2, W x2, U x2, W z2)
  const cq L1 = cmul4(c, (r0 || r2) ? mu : mw, (r0 || r3) ? z2a : x2a);
  // level 2: (a_, TT1, t1n, W2) = (2A z2, 2V z2, U c, W^2)
  const cq L2 = cmul4(c, (r0 || r1) ? z2a : (r2 ? mu : mw), r0 ? K2A : (r1 ? cadd(mv, mv) : (r2 ? CCQ : mw)));
  // level 3: (T3s, Wa, b_, nXp) = ((Wx2 - T1)^2, W a_, a_ z2, t1n (U + W))
  const cq T3d = cnorm(c, csub(c, bperm(I1, L1), L1));                    // row 0
  const cq a_all = bperm(I0, L2), t1n_all = bperm(I2, L2);
  const cq L3 = cmul4(c, r0 ? T3d : (r1 ? mw : (r2 ? a_all : t1n_all)), r0 ? T3d : (r1 ? a_all : (r2 ? z2a : up1)));
  // level 4: (T3, T2T4, W2b, TT2) = (T3s x3, (Wx2 + T1 + Wa) (UX + Wz2), W2 b_, TT1 z3)
  const cq T2p = cnorm(c, cadd(cadd(L1, bperm(I0, L1)), L3));             // row 1
  const cq T4 = cadd(bperm(I2, L1), bperm(I3, L1));
  const cq fromL2 = bperm(rowperm_idx(c, 0, 0, 3, 1), L2), x3a = bperm(I2, SX);
  const cq L4 = cmul4(c, r0 ? L3 : (r1 ? T2p : fromL2), r0 ? x3a : (r1 ? T4 : (r2 ? L3 : SZ)));
  // level 5: (T2z3, t, nY, nZ) = ((T2T4 - W2b) z3, W TT2, V (U - W), V (U + W))
  const cq T2pp = cnorm(c, csub(c, bperm(I1, L4), bperm(I2, L4)));
  const cq z3a = bperm(I2, SZ), tt2 = bperm(I3, L4);
  const cq L5 = cmul4(c, r0 ? T2pp : (r1 ? mw : mv), r0 ? z3a : (r1 ? tt2 : (r2 ? um1 : up1)));
  const cq YPn = cnorm(c, csub(c, L5, L4));                               // row 0: W^2 Yp
  // level 6: (Uo, Wo) = (t x2, t z2)
  const cq L6 = cmul4(c, bperm(I1, L5), r0 ? SX : SZ);
  const cq uo = bperm(I0, L6), wo = bperm(I1, L6);
  const cq upw = cadd(uo, wo), umw = csub(c, uo, wo);
  // level 7: (t1, Y, Z) = (Uo c, (Uo - Wo) V', V' (Uo + Wo));  level 8: X = t1 (Uo + Wo)
  const cq vq = bperm(I0, YPn);
  const cq L7 = cmul4(c, r0 ? L6 : vq, r0 ? CCQ : (r1 ? umw : upw));
  const cq L8 = cmul4(c, L7, upw);
  cq RES = r0 ? L8 : L7;
  // -P = (-t1n (U + W) : (U - W) V : V (U + W))
  const cq nxp = bperm(I3, L3), nyz = bperm(rowperm_idx(c, 0, 2, 3, 3), L5);
  const cq NEG = r0 ? cnorm(c, c.p2 - nxp) : nyz;
  const cq ID = (r1 || r2) ? ONE0 : 0u;                                   // (0 : 1 : 1)
  const cq O2 = r1 ? c.p2 - ONE0 : (r2 ? ONE0 : 0u);                      // (0 : -1 : 1)
  const uint32_t z2_zero = coop_row_is_zero(c, SZ, 0), z3_zero = coop_row_is_zero(c, SZ, 2), x2_zero = coop_row_is_zero(c, SX, 0);
  const uint32_t res_inf = z2_zero, res_negp = z3_zero & (1u - z2_zero), res_o2 = x2_zero & (1u - z2_zero);
  const uint32_t p_id = p_flags & 1u, p_o2 = (p_flags >> 1) & 1u;
  // (csel, not ?: — these flags depend on the scalar, and a wave-uniform ?: becomes a branch)
  RES = csel(res_negp, NEG, RES);
  RES = csel(res_o2, O2, RES);
  RES = csel(res_inf, ID, RES);
  RES = csel(p_id, ID, RES);
  RES = csel(p_o2, csel(k_is_odd, O2, ID), RES);
  RES = cnorm(c, RES);                                                    // (the constant -1 is 2p - 1 limb-wise)
  const cq nres = cnorm(c, c.p2 - RES);
  return csel((uint32_t)r0 & negate, nres, RES);
}

// ---- row moves without the LDS crossbar (gfx950 v_permlane16_swap / v_permlane32_swap; tools/microbench/permlane_probe.hip) -------------------
//   swap16(A, B) = ((A0, B0, A2, B2), (A1, B1, A3, B3)),   swap32(A, B) = ((A0, A1, B0, B1), (A2, A3, B2, B3))
struct cq2 { cq a, b; };
__device__ __forceinline__ cq2 swap16(cq x, cq y) { const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false); return cq2{r[0], r[1]}; }
__device__ __forceinline__ cq2 swap32(cq x, cq y) { const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false); return cq2{r[0], r[1]}; }
// the tail shared by the two additions: from (A, B, C, .) and D in rows 2, 3 of `dd` to the operands of the second level
//   Q1 = (A, A, D, D), Q2 = (B, B, C, C);  SUM = (Y3, Y3, Z3, Z3), DIF = (X3, X3, T3, T3);  returns (X3 T3, Z3 Y3, Z3 T3, X3 Y3)
__device__ __forceinline__ cq coop_add_tail(const lane_consts& c, cq LA, cq dd) {
  const bool r0 = c.row == 0, r2 = c.row == 2, r3 = c.row == 3, rlo = c.row < 2;
  const cq2 l16 = swap16(LA, LA);                                       // (A, A, C, C), (B, B, ., .)
  const cq Q1 = rlo ? l16.a : dd, Q2 = rlo ? l16.b : l16.a;
  const cq SUM = cadd(Q1, Q2), DIF = csub(c, Q1, Q2);
  const cq2 sm = swap32(SUM, SUM), df = swap32(DIF, DIF);               // Y3 x 4, Z3 x 4;  X3 x 4, T3 x 4
  return cmul4(c, cnorm(c, (r0 || r3) ? df.a : sm.b), (r0 || r2) ? df.b : sm.a);
}

// ---- general point arithmetic in quads: PubPoly::eval for small batches (share/poly.rs:457-469) ----------------------------------
// (X : Y : Z : T) -> the cached form (Y+X, Y-X, 2d T, Z) the addition below takes as its second operand.  One multiplication level.
__device__ __forceinline__ cq coop_to_cached(const lane_consts& c, cq P) {
  const bool r0 = c.row == 0, r1 = c.row == 1, r2 = c.row == 2;
  const uint32_t d2v[10] = KYB_FE_D2;
  uint32_t d2k = 0;
  KYB_UNROLL for (int j = 0; j < 10; ++j) d2k = (c.k == (uint32_t)j) ? d2v[j] : d2k;
  const cq ONE0 = (c.k == 0 && c.active) ? 1u : 0u;
  const cq2 p16 = swap16(P, P);                                         // (X, X, Z, Z), (Y, Y, T, T)
  const cq a = c.row == 3 ? p16.a : p16.b, b = swap32(p16.a, p16.a).a;  // (Y, Y, T, Z), X
  const cq F = cnorm(c, r0 ? cadd(a, b) : (r1 ? csub(c, a, b) : a));
  return cmul4(c, F, r2 ? (c.active ? d2k : 0u) : ONE0);
}
// h + E for extended h and cached E: ge_add followed by ge_p1p1_to_p3 (ge25519.h).  Two multiplication levels.
__device__ __forceinline__ cq coop_add(const lane_consts& c, cq h, cq E) {
  const bool r0 = c.row == 0, r1 = c.row == 1, r2 = c.row == 2, r3 = c.row == 3;
  const cq2 h16 = swap16(h, h);                                                   // (X, X, Z, Z), (Y, Y, T, T)
  const cq U = r3 ? h16.a : h16.b, V = swap32(h16.a, h16.a).a;                    // (Y, Y, T, Z), X
  const cq FA = cnorm(c, r0 ? cadd(U, V) : (r1 ? csub(c, U, V) : U));
  const cq LA = cmul4(c, FA, E);                                                  // (A, B, C, ZZ)
  const cq2 z16 = swap16(LA, LA);                                                 // .b = (B, B, ZZ, ZZ)
  return coop_add_tail(c, LA, cadd(z16.b, z16.b));                                // D = 2 ZZ
}
// 2 h: ge_p2_dbl followed by ge_p1p1_to_p3.  One squaring level, one multiplication level.
__device__ __forceinline__ cq coop_dbl(const lane_consts& c, cq h) {
  const bool r0 = c.row == 0, r1 = c.row == 1, r2 = c.row == 2, r3 = c.row == 3;
  // Row moves by v_permlane16_swap / v_permlane32_swap (gfx950; semantics probed on the device: tools/microbench/permlane_probe.hip) —
  //   swap16(A, B) = ((A0, B0, A2, B2), (A1, B1, A3, B3)),   swap32(A, B) = ((A0, A1, B0, B1), (A2, A3, B2, B3))
  // — instead of ds_bpermute round trips through the LDS crossbar: a lone wavefront waits out every one of those (round 6: a doubling 0.43 ->
  // 0.375 us, the 241 doublings of a one-item product 104 -> 91 us; profiles/r06/ab_permlane_rows.log)
  const auto h16 = __builtin_amdgcn_permlane16_swap(h, h, false, false);            // (X, X, Z, Z), (Y, Y, T, T)
  const cq s = h16[0] + h16[1];                                                   // (X + Y, X + Y, ., .)
  const auto s32 = __builtin_amdgcn_permlane32_swap(s, s, false, false);            // X + Y in every row
  const cq xpy = cnorm(c, s32[0]);
  const cq Q = csq4(c, r3 ? xpy : h);                                             // (XX, YY, ZZ, (X+Y)^2)
  const auto q16 = __builtin_amdgcn_permlane16_swap(Q, Q, false, false);            // (Q0, Q0, Q2, Q2), (Q1, Q1, Q3, Q3)
  const auto qe = __builtin_amdgcn_permlane32_swap(q16[0], q16[0], false, false);   // Q0 x 4, Q2 x 4
  const auto qo = __builtin_amdgcn_permlane32_swap(q16[1], q16[1], false, false);   // Q1 x 4, Q3 x 4
  const cq xx = qe[0], zz = qe[1], yy = qo[0], aa = qo[1];
  // every row picks its first operand among the raw combinations and ONE carry pass makes it tight (a lone wavefront pays per instruction, not
  // per dependency: three passes side by side cost three times one); the second operands stay lazy: Y3 <= 2T, Z3 <= 3T, T3 <= 5T (columns
  // stay below 2^63).  4p - (yy + xx) >= 0 limb-wise.
  const cq Y3 = cadd(yy, xx);
  const cq Z3 = csub(c, yy, xx);
  const cq X3 = cadd(aa, (c.p2 << 1) - Y3);
  const cq T3 = csub(c, cadd(cadd(zz, zz), xx), yy);
  return cmul4(c, cnorm(c, (r0 || r3) ? X3 : (r1 ? Y3 : Z3)), (r0 || r2) ? T3 : (r1 ? Z3 : Y3));      // (X3 T3, Y3 Z3, Z3 T3, X3 Y3)
}

// The whole ladder: UWQ = the base point's u as U1 (row 0) / W1 (row 2), tight; |scalar| = mag (or its words w_hi .. w_lo taken as a
// number of their own), the top skip_bits bits of the words from w_hi downwards known to be 0 (any number up to the piece's length).
// Returns the state after the last conditional swap: SX = (x2, x2, x3, x3), SZ = (z2, z2, z3, z3).
__device__ __forceinline__ void coop_ladder_run(const lane_consts& c, const uint32_t mag[8], int skip_bits, cq UWQ, cq& SX, cq& SZ, int w_hi = 7, int w_lo = 0) {
  const cq ONE0 = (c.k == 0 && c.active) ? 1u : 0u;
  ladder_state st;                                                       // (x2, z2, x3, z3) = (1, 0, U1, W1)
  {
    const cq u1 = bperm(rowperm_idx(c, 0, 0, 0, 0), UWQ), w1 = bperm(rowperm_idx(c, 2, 2, 2, 2), UWQ);
    st.SX = c.row < 2 ? ONE0 : u1;
    st.SZ = c.row < 2 ? 0u : w1;
  }
  const ladder_idx li = ladder_idx_init(c);
  uint32_t swap = 0;
  const int w_top = w_hi - (skip_bits >> 5);                               // whole leading words known to be zero are not walked at all
#pragma unroll 1
  for (int w = w_top; w >= w_lo; --w) {                                    // the scalar's words w_hi .. w_lo (a piece of it, or all eight)
    uint32_t word = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) word = (w == q) ? mag[q] : word;
    const int first = (w == w_top) ? (skip_bits & 31) : 0;
    word <<= first;
#pragma unroll 1
    for (int j = first; j < 32; ++j) {
      const uint32_t bit = word >> 31;
      word <<= 1;
      swap ^= bit;
      st = coop_ladder_step(c, li, st, UWQ, swap);                       // (rows 1, 3 of UWQ are not read)
      swap = bit;
    }
  }
  // the final conditional swap (x2, z2) <-> (x3, z3): rows 0, 1 <-> 2, 3
  const int I_sw = (int)(c.lane << 2) ^ ((0 - (int)swap) & 128);
  SX = bperm(I_sw, st.SX);
  SZ = bperm(I_sw, st.SZ);
}

// One item per wavefront — or, for very few items, FOUR single-wavefront workgroups per item: the scalar is cut into pieces
// k = sum_j k_j 2^(b_j); workgroup j doubles P b_j times (Edwards doublings, 0.3 us each), runs a ladder of the piece's length for k_j on
// that point, recovers the full point and leaves it in device scratch; the workgroup that arrives LAST (one atomic counter per item) adds
// the four in a fixed order and encodes.  Measured in place, a doubling costs 0.79 of a ladder step (0.444 / 0.560 us:
// tools/mul_coop_pieces_probe.py, profiles/r05/mul_coop_pieces.log), so the chain of the top piece is hardly shorter than its doublings:
// with pieces of 144 / 66 / 31 / 15 bits from the bottom the four chains — b_j doublings, then the piece's steps — are as long as 144 / 180 /
// 197 / 205 steps (the optimum, 203 / 43 / 8 / 2 bits, is 203), where one wavefront alone walks 256 and the four 64-bit pieces of rounds
// 3 - 4 needed 216: one item 139 us (projective limbs out) against 154 before and 167 on one wavefront.  Separate workgroups land on
// separate compute units.  The cut is public (the workgroup's number), the instruction stream the same for every scalar; nobody waits for
// anybody: every workgroup runs to its end.
constexpr int KYB_COOP_CUT1 = 144, KYB_COOP_CUT2 = 210, KYB_COOP_CUT3 = 241;
__global__ void __launch_bounds__(64)
k_mul_coop(const uint8_t* __restrict__ scalars, const int32_t* __restrict__ pts_ext, size_t n, uint8_t* __restrict__ out_enc, int32_t* __restrict__ out_ext,
           int skip_bits, uint4* __restrict__ proj, size_t proj_stride, size_t proj_offset, size_t pt_mod, int pieces, int ext_proj, uint32_t* __restrict__ part,
           uint32_t* pieces_buf, kyb::launch::DoneFlag df) {
  const size_t i = pieces == 1 ? (size_t)blockIdx.x : (size_t)blockIdx.x >> 2;      // (the four pieces of an item: consecutive workgroups, dealt to different XCDs)
  if (i >= n) return;
  const int piece = pieces == 1 ? 0 : (int)(blockIdx.x & 3u);
  KYB_COOP_CONSTS(c, 1);
  const size_t ip = pt_mod ? i % pt_mod : i;                             // shared operands: item i multiplies point i mod pt_mod
  KYB_PHASE(0 + piece, blockIdx.x < 4);                                   // slots 0..3: a piece's wavefront starts

  // ---- operands: the scalar replicated on all lanes, the point as a quad (X, Y, Z, T) straight from its 40 reference limbs ----
  uint32_t a[8];
  load_words8(a, scalars, i);
  uint32_t neg, mag[8];
  sc_effective(neg, mag, a);
  const uint32_t word = c.active ? (uint32_t)pts_ext[40 * ip + 10 * c.row + c.k] : 0u;
  cq PQ = cnorm(c, c.active ? word + (c.p2 << 3) : 0u);                 // fe_from_ref10: signed limb + 16p, one carry pass
  if (pieces == 1) {
    uint32_t p_flags;
    const cq M = coop_mont_prep(c, PQ, p_flags);                         // u = U / W, v = V / W: no inversion in front of the ladder
    cq SX, SZ;
    coop_ladder_run(c, mag, skip_bits, M, SX, SZ);
    // y-recovery, exceptional cases, encoding
    const cq RES = coop_mont_recover(c, M, SX, SZ, p_flags, mag[0] & 1u, neg);
    if (part != nullptr) {                                               // a term of a linear combination: the extended point for k_sum_coop
      const cq zz = bperm(rowperm_idx(c, 2, 2, 2, 2), RES), xy = bperm(rowperm_idx(c, 0, 1, 2, 0), RES), yy = bperm(rowperm_idx(c, 1, 1, 1, 1), RES);
      const cq e = cmul4(c, xy, c.row == 3 ? yy : zz);
      if (c.active) part[i * 40 + 10 * c.row + c.k] = e;
      return;
    }
    coop_finish(c, RES, 0u, out_enc, out_ext, i, proj, proj_stride, proj_offset, ext_proj != 0, false);
    if (c.lane == 0) signal_done(df);
    return;
  }
  // ---- four workgroups per item: bits [lo, hi) of the scalar on 2^lo P ----
  const int lo = piece == 0 ? 0 : (piece == 1 ? KYB_COOP_CUT1 : (piece == 2 ? KYB_COOP_CUT2 : KYB_COOP_CUT3));
  const int hi = piece == 0 ? KYB_COOP_CUT1 : (piece == 1 ? KYB_COOP_CUT2 : (piece == 2 ? KYB_COOP_CUT3 : 256));
  const int len = hi - lo;
#pragma unroll 1
  for (int d = 0; d < lo; ++d) PQ = coop_dbl(c, PQ);
  KYB_PHASE(4 + piece, blockIdx.x < 4);                                   // slots 4..7: its doublings are done
  // the piece as a number of its own: (mag >> lo) mod 2^len.  Word and bit offsets are the workgroup's (public); the words are picked by selects
  uint32_t pm[8];
  {
    const int ws = lo >> 5, bs = lo & 31;
    KYB_UNROLL for (int q = 0; q < 8; ++q) {
      uint32_t w0 = 0, w1 = 0;
      KYB_UNROLL for (int k = 0; k < 8; ++k) { w0 = (q + ws == k) ? mag[k] : w0; w1 = (q + ws + 1 == k) ? mag[k] : w1; }
      const uint32_t v = bs ? ((w0 >> bs) | (w1 << (32 - bs))) : w0;
      const int keep = len - 32 * q;                                      // bits of this word that belong to the piece
      pm[q] = keep >= 32 ? v : (keep <= 0 ? 0u : (v & ((1u << keep) - 1u)));
    }
  }
  uint32_t p_flags;
  const cq M = coop_mont_prep(c, PQ, p_flags);
  cq SX, SZ;
  // leading bits of the 256-bit register known to be zero for the whole launch (skip_bits) shorten the piece they reach into
  int known_zero = hi - (256 - skip_bits);
  known_zero = known_zero < 0 ? 0 : (known_zero > len ? len : known_zero);
  coop_ladder_run(c, pm, 256 - len + known_zero, M, SX, SZ);
  const cq R = coop_mont_recover(c, M, SX, SZ, p_flags, pm[0] & 1u, 0u);
  // (X : Y : Z) -> extended (X Z : Y Z : Z^2 : X Y), into this item's four records; then the arrival counter
  const cq zz = bperm(rowperm_idx(c, 2, 2, 2, 2), R), xy = bperm(rowperm_idx(c, 0, 1, 2, 0), R), yy = bperm(rowperm_idx(c, 1, 1, 1, 1), R);
  cq q = cmul4(c, xy, c.row == 3 ? yy : zz);
  uint32_t* mine = pieces_buf + (4 * i + (size_t)piece) * 40;
  uint32_t* arrived = pieces_buf + 160 * n + i;                           // (behind the n * 4 records; zero between launches)
  if (c.active) mine[10 * c.row + c.k] = q;
  KYB_PHASE(8 + piece, blockIdx.x < 4);                                   // slots 8..11: ladder, recovery and record are done
  __threadfence();                                                         // release: the record before the count, device-wide (the others are on other XCDs)
  uint32_t before = 0;
  if (c.lane == 0) before = atomicAdd(arrived, 1u);
  before = (uint32_t)__builtin_amdgcn_readfirstlane((int)before);
  if (before != 3u) return;                                                // not the last: done
  __threadfence();                                                         // acquire: the three other records
  // the same sum whoever comes last: ((piece 0 + piece 1) + piece 2) + piece 3; the records held multiples of a secret — cleared here
  const uint32_t* rec = pieces_buf + 4 * i * 40;
  q = c.active ? __builtin_nontemporal_load(rec + 10 * c.row + c.k) : 0u;
#pragma unroll 1
  for (int w = 1; w < 4; ++w) {
    const cq o = c.active ? __builtin_nontemporal_load(rec + 40 * w + 10 * c.row + c.k) : 0u;
    q = coop_add(c, q, coop_to_cached(c, o));
  }
  if (c.lane < 40u) { KYB_UNROLL for (int w = 0; w < 4; ++w) pieces_buf[(4 * i + (size_t)w) * 40 + c.lane] = 0u; }
  if (c.lane == 0) *arrived = 0u;
  KYB_PHASE(12, blockIdx.x < 4);                                          // slot 12: the four pieces are added
  coop_finish(c, q, neg, out_enc, out_ext, i, proj, proj_stride, proj_offset, ext_proj != 0, true);
  KYB_PHASE(13, blockIdx.x < 4);                                          // slot 13: encoded and stored
  if (c.lane == 0) signal_done(df);
}

// kyb_lincomb_public_batch, shared points (kernels_msm.hip): the window bases 64^w P for w = 0 .. 42 of one point per wavefront — 252
// cooperative doublings, a dependent chain of 0.1 ms where one lane would need 0.6 — as raw extended quads (40 tight limbs each).
__global__ void __launch_bounds__(64)
k_msm_bases_coop(const int32_t* __restrict__ pts_ext, size_t t, uint32_t* __restrict__ bases) {
  const size_t j = blockIdx.x;
  if (j >= t) return;
  KYB_COOP_CONSTS(c, 1);
  const uint32_t word = c.active ? (uint32_t)pts_ext[40 * j + 10 * c.row + c.k] : 0u;
  cq PQ = cnorm(c, c.active ? word + (c.p2 << 3) : 0u);                 // fe_from_ref10: signed limb + 16p, one carry pass
  // (the T an input carries is not trusted to be X Y / Z: rebuilt from X, Y, Z as the product kernels do)
  {
    const cq zz = bperm(rowperm_idx(c, 2, 2, 2, 2), PQ), xy = bperm(rowperm_idx(c, 0, 1, 2, 0), PQ), yy = bperm(rowperm_idx(c, 1, 1, 1, 1), PQ);
    PQ = cmul4(c, xy, c.row == 3 ? yy : zz);                             // (X Z : Y Z : Z^2 : X Y)
  }
#pragma unroll 1
  for (int w = 0; w < 43; ++w) {
    if (c.active) bases[(j * 43 + (size_t)w) * 40 + 10 * c.row + c.k] = PQ;
    if (w == 42) break;
#pragma unroll 1
    for (int d = 0; d < 6; ++d) PQ = coop_dbl(c, PQ);
  }
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
  (void)mi;
  const cq2 h16 = swap16(h, h);                                        // (X, X, Z, Z), (Y, Y, T, T)
  const cq U = h16.b, V = swap32(h16.a, h16.a).a;                      // (Y, Y, T, T), (X, X, X, X)
  const cq FA = cnorm(c, r0 ? cadd(U, V) : (r1 ? csub(c, U, V) : (r2 ? U : 0u)));
  const cq LA = cmul4(c, FA, E);                                       // (A, B, C, 0)
  // X3 = A - B, Y3 = A + B, Z3 = D + C, T3 = D - C with D = 2Z
  return coop_add_tail(c, LA, cadd(h16.a, h16.a));                     // rows 2, 3 of (X, X, Z, Z) doubled: D
}

// sum_{j < count} x^j C_{first + j} by Horner (x >= 1 public and wave-uniform: the instruction stream follows its bits), every
// point operation two cooperative levels.  Returns the extended point (X : Y : Z : T).
__device__ __forceinline__ cq coop_horner(const lane_consts& c, const int32_t* __restrict__ commits_ext, size_t first, int count, uint32_t x) {
  const cq ONE0 = (c.k == 0 && c.active) ? 1u : 0u;
  cq v = (c.row == 1 || c.row == 2) ? ONE0 : 0u;                       // neutral element (0 : 1 : 1 : 0)
  const int top = 31 - __builtin_clz(x | 1u);
  // coefficient j - 1 is fetched while step j computes (the operands of a small host-pointer call sit in page-locked HOST memory:
  // ~1.5 us per dependent read over PCIe)
  auto fetch = [&](int j) { return (c.active && j >= 0) ? (uint32_t)commits_ext[40 * (first + (size_t)j) + 10 * c.row + c.k] : 0u; };
  uint32_t word = fetch(count - 1);
#pragma unroll 1
  for (int j = count - 1; j >= 0; --j) {
    const uint32_t word_next = fetch(j - 1);
    // v <- x v  (binary, most significant bit first; x = 1 leaves v alone)
    if (top > 0) {
      const cq vc = coop_to_cached(c, v);
      cq acc = v;
#pragma unroll 1
      for (int b = top - 1; b >= 0; --b) {
        acc = coop_dbl(c, acc);
        if ((x >> b) & 1u) acc = coop_add(c, acc, vc);
      }
      v = acc;
    }
    // v <- v + C_j
    const cq C = cnorm(c, c.active ? word + (c.p2 << 3) : 0u);         // fe_from_ref10: signed limb + 16p, one carry pass
    v = coop_add(c, v, coop_to_cached(c, C));
    word = word_next;
  }
  return v;
}

// PubPoly::eval at one share index per wavefront: v = sum_j x^j C_j, x = index + 1.  The batch kernel (k_poly_eval, one evaluation
// per lane) walks the same t (nbits + 1) point operations as ~150-instruction field multiplications of ONE lane; with a few thousand
// evaluations or fewer this is what a DKG node's verify_deal pass looks like (vss.rs:904-909: n polynomials at its own index).
__global__ void __launch_bounds__(64)
k_poly_eval_coop(const int32_t* __restrict__ commits_ext, int t, const uint32_t* __restrict__ indices, size_t n, int nbits, size_t per_poly,
                 uint8_t* __restrict__ out_enc, int32_t* __restrict__ out_ext, int ext_proj, kyb::launch::DoneFlag df) {
  const size_t i = blockIdx.x;
  if (i >= n) return;
  KYB_COOP_CONSTS(c, 1);
  const uint32_t x = (uint32_t)__builtin_amdgcn_readfirstlane((int)(indices[i] + 1u));      // wave-uniform: the branches on its bits are scalar
  const size_t first = per_poly ? (i / per_poly) * (size_t)t : 0;      // first commitment of this item's polynomial
  (void)nbits;
  const cq v = coop_horner(c, commits_ext, first, t, x);
  coop_finish(c, v, 0u, out_enc, out_ext, i, nullptr, 0, 0, ext_proj != 0, true);
  if (c.lane == 0) signal_done(df);
}

// The same for ONE long polynomial and few evaluations: the Horner chain of t steps is cut into `segs` segments of `len`
// coefficients, one wavefront (and workgroup: the wavefronts of one workgroup would share a CU's four SIMDs) each:
//     P(x) = sum_s x^(s len) Q_s(x),   Q_s(x) = sum_{j < len} x^j C_{s len + j}.
// Wavefront s evaluates Q_s by Horner with the small multiplier x and then multiplies it by x^(s len), reduced mod 8L
// (sc_pow_mod8L_signed: the commitments may carry small-order components, mod L alone would not be exact) — the variable-base
// ladder of k_mul_coop with its image and recovery in quads — and leaves the extended point in `part`; k_poly_eval_sum adds an
// evaluation's partial results up and encodes.  The chain per wavefront is len (bits(x) + 1) point operations + one
// multiplication (~26 Horner steps of a 10-bit index) instead of t (bits(x) + 1).
__global__ void __launch_bounds__(64)
k_poly_eval_seg(const int32_t* __restrict__ commits_ext, int t, const uint32_t* __restrict__ indices, size_t n, size_t per_poly, int len, int segs,
                uint32_t* __restrict__ part) {
  const size_t b = blockIdx.x;
  if (b >= n * (size_t)segs) return;
  const size_t i = b / (size_t)segs;
  const int sg = (int)(b % (size_t)segs);
  KYB_COOP_CONSTS(c, 1);
  const uint32_t x = (uint32_t)__builtin_amdgcn_readfirstlane((int)(indices[i] + 1u));
  const size_t first = per_poly ? (i / per_poly) * (size_t)t : 0;
  const int lo = sg * len, cnt = lo >= t ? 0 : (t - lo < len ? t - lo : len);
  cq q = coop_horner(c, commits_ext, first + (size_t)lo, cnt, x);       // (cnt == 0: the neutral element)
  if (sg > 0 && cnt > 0) {
    uint32_t mag[8], neg;
    sc_pow_mod8L_signed(mag, neg, x, (uint32_t)lo);
    uint32_t p_flags;
    const cq M = coop_mont_prep(c, q, p_flags);
    cq SX, SZ;
    coop_ladder_run(c, mag, 1, M, SX, SZ);                               // |multiplier| < 4L < 2^255
    const cq R = coop_mont_recover(c, M, SX, SZ, p_flags, mag[0] & 1u, neg);
    // (X : Y : Z) -> extended (X Z : Y Z : Z^2 : X Y)
    const cq zz = bperm(rowperm_idx(c, 2, 2, 2, 2), R), xy = bperm(rowperm_idx(c, 0, 1, 2, 0), R), yy = bperm(rowperm_idx(c, 1, 1, 1, 1), R);
    q = cmul4(c, xy, c.row == 3 ? yy : zz);
  }
  if (c.active) part[b * 40 + 10 * c.row + c.k] = q;
}
__global__ void __launch_bounds__(64)
k_poly_eval_sum(const uint32_t* __restrict__ part, size_t n, int segs, uint8_t* __restrict__ out_enc, int32_t* __restrict__ out_ext, int ext_proj,
                kyb::launch::DoneFlag df) {
  const size_t i = blockIdx.x;
  if (i >= n) return;
  KYB_COOP_CONSTS(c, 1);
  auto load = [&](int sg) { return c.active ? part[(i * (size_t)segs + (size_t)sg) * 40 + 10 * c.row + c.k] : 0u; };
  cq q = load(0);
#pragma unroll 1
  for (int sg = 1; sg < segs; ++sg) q = coop_add(c, q, coop_to_cached(c, load(sg)));
  coop_finish(c, q, 0u, out_enc, out_ext, i, nullptr, 0, 0, ext_proj != 0, true);
  if (c.lane == 0) signal_done(df);
}

// One entry of the radix-64 table for the cooperative layout: lane (row g < 3, limb r) gets word 10 g + r of entry idx of a window
// (E entries) — by a scan whose ADDRESSES depend on nothing secret.  `win` points into the context's entry-major copy of the table
// (k_base_table_coop: 32 words = one 128-byte line per entry, words 30 and 31 zero).  Lane (half h, l) reads word l of the entries
// 2m + h, m = 0 .. E/2 - 1: E/2 loads of 256 contiguous bytes each; a select tree over the upper bits of idx (opaque masks, v_bfi_b32)
// leaves it with word l of entry (idx & ~1) | h; ONE ds_bpermute_b32 then pulls word 10 g' + r from half (idx & 1), where g' swaps
// the y+x and y-x rows of a negated entry — the lane select of the crossbar is the only place the digit goes (tools/ct_check.py).
// (Round 2 read four words at addresses that carried idx & 7 and the sign: the same cache LINES whatever the digit, but not the same
// addresses.)
template <int E>
__device__ __forceinline__ cq coop_table_entry(const lane_consts& c, const uint32_t* __restrict__ win, uint32_t idx, uint32_t negate) {
  const uint32_t l = c.lane & 31u, half = c.lane >> 5;
  uint32_t cand[E / 2];
#pragma unroll
  for (int m = 0; m < E / 2; ++m) cand[m] = win[(2 * m + (int)half) * 32 + (int)l];
  constexpr int LEVELS = (E == 32) ? 4 : 3;
#pragma unroll
  for (int t = 0; t < LEVELS; ++t) {
    uint32_t mk = 0u - ((idx >> (t + 1)) & 1u);
    asm volatile("" : "+v"(mk));
#pragma unroll
    for (int m = 0; m < (E / 4) >> t; ++m) cand[m] = (cand[2 * m + 1] & mk) | (cand[2 * m] & ~mk);
  }
  const uint32_t g = c.row < 3 ? c.row : 0u;
  const uint32_t g_eff = g ^ (negate & (uint32_t)(g < 2u));            // a negated entry is (ymx, ypx, -xy2d)
  const uint32_t kk = c.active ? c.k : 0u;
  const uint32_t src = ((idx & 1u) << 5) | (10u * g_eff + kk);
  uint32_t v = bperm((int)(src << 2), cand[0]);
  v = (c.active && c.row < 3) ? v : 0u;
  const uint32_t nv = c.p2 - v;                                        // 2p - xy2d
  return csel((uint32_t)(c.row == 2) & negate, nv, v);
}

// a' B for the scalar words a (sc_recode64's signed radix-64 digits): the point (X : Y : Z : T) before the sign of the top digit
// (`neg`: negate X) is applied.  43 cooperative mixed additions — or the share [pos_lo, pos_hi) of the 43 windows (window 42 is the
// top one) when several wavefronts divide them between themselves.
__device__ __forceinline__ cq coop_base_mul(const lane_consts& c, const uint32_t a[8], const uint32_t* __restrict__ table_coop, uint32_t& neg,
                                            int pos_lo = 0, int pos_hi = KYB_BASE64_POS) {
  sc_digits64 dg;
  sc_recode64(dg, a);
  // drop the digits below pos_lo: whole words, then the remaining bits (pos_lo is wave-uniform)
  const int skip = 6 * pos_lo;
#pragma unroll 1
  for (int wd = 0; wd < (skip >> 5); ++wd) {
    KYB_UNROLL for (int q = 0; q < 7; ++q) dg.w[q] = dg.w[q + 1];
    dg.w[7] = 0;
  }
  if (skip & 31) {
    const uint32_t r = (uint32_t)(skip & 31);
    KYB_UNROLL for (int q = 0; q < 7; ++q) dg.w[q] = (dg.w[q] >> r) | (dg.w[q + 1] << (32u - r));
    dg.w[7] >>= r;
  }
  const cq ONE0 = (c.k == 0 && c.active) ? 1u : 0u;
  cq h = (c.row == 1 || c.row == 2) ? ONE0 : 0u;                         // neutral element (0 : 1 : 1 : 0)
  const madd_idx mi = madd_idx_init(c);
  const int last = pos_hi < KYB_BASE64_POS - 1 ? pos_hi : KYB_BASE64_POS - 1;
#pragma unroll 1
  for (int pos = pos_lo; pos < last; ++pos) {
    uint32_t idx, ng;
    sc_next_digit64(idx, ng, dg, false);
    h = coop_madd(c, mi, h, coop_table_entry<32>(c, table_coop + pos * KYB_COOP_WIN_WORDS, idx, ng));
  }
  if (pos_hi == KYB_BASE64_POS) {
    uint32_t idx, ng;
    sc_next_digit64(idx, ng, dg, true);
    h = coop_madd(c, mi, h, coop_table_entry<16>(c, table_coop + 42 * KYB_COOP_WIN_WORDS, idx, 0u));
  }
  neg = dg.neg;
  return h;
}

// One item per workgroup of `waves` wavefronts (1 or 4): with four, each adds up a quarter of the 43 windows and wavefront 0 adds the
// four partial sums (three general additions) before it encodes — the dependent chain of a one-item call is 11 + 3 additions
// instead of 43.
__global__ void __launch_bounds__(256)
k_mul_base_coop(const uint8_t* __restrict__ scalars, const uint8_t* __restrict__ scalars_b, size_t n_a, size_t n, uint8_t* __restrict__ out_enc,
                int32_t* __restrict__ out_ext, const uint32_t* __restrict__ table_coop, uint4* __restrict__ proj, size_t proj_stride, size_t proj_offset,
                int waves, int ext_proj, kyb::launch::DoneFlag df) {
  __shared__ uint32_t sh_part[3 * 40];
  const size_t i = blockIdx.x;
  if (i >= n) return;
  const int wave = (int)(threadIdx.x >> 6);
  KYB_COOP_CONSTS(c, 4);
  uint32_t a[8];
  if (i < n_a) load_words8(a, scalars, i); else load_words8(a, scalars_b, i - n_a);      // two arrays in one launch (signing: nonces, then keys)
  uint32_t neg;
  const int per = (KYB_BASE64_POS + waves - 1) / waves;
  const int lo = wave * per, hi = (wave + 1) * per < KYB_BASE64_POS ? (wave + 1) * per : KYB_BASE64_POS;
  KYB_PHASE(16, blockIdx.x == 0 && wave == 0);                            // slot 16: wavefront 0 starts (operands loaded)
  cq h = coop_base_mul(c, a, table_coop, neg, lo, hi);
  KYB_PHASE(17, blockIdx.x == 0 && wave == 0);                            // slot 17: its share of the 43 windows is added up
  if (waves > 1) {
    if (wave > 0 && c.active) sh_part[(wave - 1) * 40 + 10 * c.row + c.k] = h;
    __syncthreads();
    if (wave != 0) return;
#pragma unroll 1
    for (int w = 1; w < waves; ++w) {
      const cq o = c.active ? sh_part[(w - 1) * 40 + 10 * c.row + c.k] : 0u;
      h = coop_add(c, h, coop_to_cached(c, o));
    }
  }
  KYB_PHASE(18, blockIdx.x == 0);                                         // slot 18: the partial sums of the other wavefronts are in
  coop_finish(c, h, neg, out_enc, out_ext, i, proj, proj_stride, proj_offset, ext_proj != 0, true);
  KYB_PHASE(19, blockIdx.x == 0);                                         // slot 19: encoded and stored
  if (c.lane == 0) signal_done(df);
}

// k_finish / k_encode for a small batch: one point per wavefront, the inversion cooperative (csrc/kernels_misc.hip has the batch forms,
// one inversion per 8 points of ONE lane: 70 us however few the points).  Source: projective staging record i * src_mul, or the 40
// reference limbs of point i.
__global__ void __launch_bounds__(64)
k_finish_coop(const uint4* __restrict__ proj, size_t stride, const int32_t* __restrict__ pts_ext, size_t n, uint8_t* __restrict__ out_enc,
              int32_t* __restrict__ out_ext, size_t src_mul, int ext_proj, kyb::launch::DoneFlag df) {
  const size_t i = blockIdx.x;
  if (i >= n) return;
  KYB_COOP_CONSTS(c, 1);
  cq q;
  // A point handed over as limbs whose Z is LITERALLY one — (1, 0, ..., 0): what unmarshal_binary produces and what every entry point of this
  // library returns unless ext.projective is set — is affine already: marshal_binary is then a reduction of X and Y, no inversion (38 -> 11 us
  // for a lone point: the n marshals of the verifier keys in new_dealer, vss.rs:1075-1100).  The test looks at how the point is REPRESENTED,
  // which the caller's own call history decides; a projective result (a product the engine left with Z != 1) has Z = 1 with probability
  // 2^-255, so the branch tells an observer nothing about the point (include/kyber_ed25519.h, "timing").
  bool z_is_one = false;
  if (pts_ext != nullptr) {
    const uint32_t word = c.active ? (uint32_t)pts_ext[40 * i + 10 * c.row + c.k] : 0u;
    q = cnorm(c, c.active ? word + (c.p2 << 3) : 0u);
    const bool lane_ok = c.row != 2u || !c.active || word == (c.k == 0u ? 1u : 0u);
    z_is_one = __builtin_amdgcn_ballot_w64(lane_ok) == ~0ull;
  } else {
    const uint32_t w = 10u * (c.row < 3 ? c.row : 0u) + (c.active ? c.k : 0u);
    const uint32_t v = reinterpret_cast<const uint32_t*>(proj)[((size_t)(w >> 2) * stride + i * src_mul) * 4 + (w & 3u)];
    q = (c.active && c.row < 3) ? v : 0u;                                // staging records hold tight limbs
  }
  coop_finish(c, q, 0u, out_enc, out_ext, i, nullptr, 0, 0, ext_proj != 0, false, z_is_one);
  if (c.lane == 0) signal_done(df);
}

// k_finish / k_encode_batched for a MID-SIZE batch (above the one-point-per-wavefront sizes, up to a wavefront per SIMD): one point per LANE, ONE
// inversion per WAVEFRONT, and that one spread over its lanes (fe_invert_gcd_wave).  The batch forms of csrc/kernels_misc.hip share an inversion
// between 4 (8) points of one lane — a lane's 600 divsteps are the whole kernel, 52 us however few the points (profiles/r06/finish_crossover.log);
// here the 64 lanes multiply their Z's together in a butterfly (level j: lane l takes the sub-product of lane l ^ 2^j and multiplies it on — after six
// levels every lane holds the product of all 64, each having kept the six sub-products it received), the wavefront inverts that once, and every
// lane peels its own 1/Z off by multiplying the six kept sub-products back on: 1/P_j = 1/P_{j+1} * Q_j.  12 products + one cooperative inversion
// per 64 points instead of 9 + a one-lane inversion per 4.  The lanes' products agree mod p, not limb by limb: the inversion starts from the
// canonical words (fe_to_words) and everything stored is canonical, so the bytes and limbs are those of the other forms.  A zero Z (invalid
// extended input only) counts as 1 in the product and gets the reference's own 0^(p-2) = 0, as in finish_body.  Source: projective staging
// record i * src_mul, or the 40 reference limbs of point i.
__device__ __forceinline__ void fe_of_lane_xor(fe& h, const fe& f, int mask) {
  KYB_UNROLL for (int k = 0; k < 10; ++k) h.v[k] = (uint32_t)__shfl_xor((int)f.v[k], mask);
}
__global__ void __launch_bounds__(64)
k_finish_wave(const uint4* __restrict__ proj, size_t stride, const int32_t* __restrict__ pts_ext, size_t n, uint8_t* __restrict__ out_enc,
              int32_t* __restrict__ out_ext, size_t src_mul) {
  const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  const bool live = i < n;
  const size_t ii = live ? i : n - 1;                                     // dead lanes of the last wavefront redo its last point (nothing stored)
  fe X, Y, Z, one, zero;
  fe_one(one); fe_zero(zero);
  if (pts_ext != nullptr) {
    ge_p3 P;
    load_ext(P, pts_ext, ii);
    fe_copy(X, P.X); fe_copy(Y, P.Y); fe_copy(Z, P.Z);
  } else {
    load_proj_z(Z, proj, stride, ii * src_mul);
    load_proj_xy(X, Y, proj, stride, ii * src_mul);
  }
  const uint32_t z_zero = 1u - fe_is_nonzero(Z);
  fe P, Q0, Q1, Q2, Q3, Q4, Q5;
  fe_copy(P, Z);
  fe_cmov(P, one, z_zero);
  fe_of_lane_xor(Q0, P, 1);  fe_mul(P, P, Q0);
  fe_of_lane_xor(Q1, P, 2);  fe_mul(P, P, Q1);
  fe_of_lane_xor(Q2, P, 4);  fe_mul(P, P, Q2);
  fe_of_lane_xor(Q3, P, 8);  fe_mul(P, P, Q3);
  fe_of_lane_xor(Q4, P, 16); fe_mul(P, P, Q4);
  fe_of_lane_xor(Q5, P, 32); fe_mul(P, P, Q5);
  fe I;
  fe_invert_gcd_wave(I, P);                                               // (wave-uniform input: the canonical words of the 64 lanes' products are the same)
  fe_mul(I, I, Q5); fe_mul(I, I, Q4); fe_mul(I, I, Q3); fe_mul(I, I, Q2); fe_mul(I, I, Q1); fe_mul(I, I, Q0);
  fe_cmov(I, zero, z_zero);                                               // Z == 0: the reference's 0^(p-2) = 0
  fe x, y;
  fe_mul(x, X, I);
  fe_mul(y, Y, I);
  if (out_enc != nullptr) {
    uint32_t w[8];
    fe_to_words(w, y);
    w[7] ^= fe_is_negative(x) << 31;
    if (live) store_words8(out_enc, i, w);
  }
  if (out_ext != nullptr) {
    fe tt;
    fe_mul(tt, x, y);
    if (live) store_ext(out_ext, i, x, y, one, tt);
  }
}

// out[g] = sum_j P[g t + j] for short sums (kyb_sum_batch, the tail of kyb_lincomb_batch), one group per wavefront: t - 1 cooperative
// additions, then the finish.  Source: extended quads in `part` (k_mul_coop's products) or the 40 reference limbs per point.
__global__ void __launch_bounds__(64)
k_sum_coop(const uint32_t* __restrict__ part, const int32_t* __restrict__ pts_ext, size_t m, size_t t, uint8_t* __restrict__ out_enc,
           int32_t* __restrict__ out_ext, int ext_proj, kyb::launch::DoneFlag df) {
  const size_t g = blockIdx.x;
  if (g >= m) return;
  KYB_COOP_CONSTS(c, 1);
  auto load = [&](size_t j) -> cq {
    const size_t i = g * t + j;
    if (part != nullptr) return c.active ? part[i * 40 + 10 * c.row + c.k] : 0u;
    const uint32_t word = c.active ? (uint32_t)pts_ext[40 * i + 10 * c.row + c.k] : 0u;
    return cnorm(c, c.active ? word + (c.p2 << 3) : 0u);
  };
  cq q = load(0);
#pragma unroll 1
  for (size_t j = 1; j < t; ++j) q = coop_add(c, q, coop_to_cached(c, load(j)));
  coop_finish(c, q, 0u, out_enc, out_ext, g, nullptr, 0, 0, ext_proj != 0, true);
  if (c.lane == 0) signal_done(df);
}

// unmarshal_binary of a small batch: extended limbs and the ok flag; or_identity: failed decodes become the neutral element
__global__ void __launch_bounds__(64)
k_decode_coop(const uint8_t* __restrict__ enc, size_t n, int32_t* __restrict__ out_ext, uint8_t* __restrict__ ok_out, int or_identity, kyb::launch::DoneFlag df) {
  const size_t i = blockIdx.x;
  if (i >= n) return;
  KYB_COOP_CONSTS(c, 1);
  uint32_t w[8];
  load_words8(w, enc, i);
  ge_p3 P, id;
  const uint32_t ok = coop_decode_fn{c}(P, w);
  const uint32_t repl = or_identity ? 1u - ok : 0u;
  ge_p3_0(id);
  fe_cmov(P.X, id.X, repl); fe_cmov(P.Y, id.Y, repl); fe_cmov(P.Z, id.Z, repl); fe_cmov(P.T, id.T, repl);
  if (c.lane == 0) {
    store_ext(out_ext, i, P.X, P.Y, P.Z, P.T);
    if (ok_out != nullptr) ok_out[i] = (uint8_t)ok;
    signal_done(df);
  }
}

// the A half of a verification (k_verify_prep, kernels_verify.hip): checks, decode of A, h = SHA-512(R || A || msg) mod L
__global__ void __launch_bounds__(64)
k_verify_prep_coop(const uint8_t* __restrict__ pubs, const uint8_t* __restrict__ sigs, const uint8_t* __restrict__ msgs,
                   const uint32_t* __restrict__ msg_off, size_t n, uint8_t* __restrict__ flags_a,
                   uint8_t* __restrict__ hbuf, uint8_t* __restrict__ sbuf, int32_t* __restrict__ a_ext) {
  const size_t i = blockIdx.x;
  if (i >= n) return;
  KYB_COOP_CONSTS(c, 1);
  uint32_t pub[8], sig[16], h[8];
  load_words8(pub, pubs, i);
  load_words8(sig, sigs, 2 * i);
  load_words8(sig + 8, sigs, 2 * i + 1);
  const uint32_t off = msg_off[i], len = msg_off[i + 1] - off;
  ge_p3 A;
  const uint32_t fl = verify_prep_a_with(h, A, pub, sig, msgs + off, len, coop_decode_fn{c});
  if (c.lane == 0) {
    flags_a[i] = (uint8_t)fl;
    store_words8(hbuf, i, h);
    store_words8(sbuf, i, sig + 8);
    store_ext(a_ext, i, A.X, A.Y, A.Z, A.T);
  }
}
// the R half (k_verify_prep_r)
__global__ void __launch_bounds__(64)
k_verify_prep_r_coop(const uint8_t* __restrict__ sigs, size_t n, uint8_t* __restrict__ flags_r, uint4* __restrict__ proj, size_t stride, size_t proj_offset) {
  const size_t i = blockIdx.x;
  if (i >= n) return;
  KYB_COOP_CONSTS(c, 1);
  uint32_t sig[16];
  load_words8(sig, sigs, 2 * i);
  load_words8(sig + 8, sigs, 2 * i + 1);
  ge_p3 R;
  const uint32_t fl = verify_prep_r_with(R, sig, coop_decode_fn{c});
  if (c.lane == 0) {
    flags_r[i] = (uint8_t)fl;
    store_proj(proj, stride, proj_offset + i, R.X, R.Y, R.Z);
  }
}

// schnorr::sign (schnorr_sig.rs:25-47) in one launch, eight wavefronts per signature: R = k B in wavefronts 0..3 and A = x B in
// wavefronts 4..7 (idle when the signer's stored public key is given), each a quarter of the 43 windows as in k_mul_base_coop; then
// wavefront 0 hashes and forms s = k + x h (k_sign_hash's work).
__global__ void __launch_bounds__(512)
k_sign_coop(const uint8_t* __restrict__ x, const uint8_t* __restrict__ k, const uint8_t* __restrict__ pubs, const uint8_t* __restrict__ msgs,
            const uint32_t* __restrict__ msg_off, size_t n, uint8_t* __restrict__ sig, uint8_t* __restrict__ pub_out,
            const uint32_t* __restrict__ table_coop, kyb::launch::DoneFlag df) {
  __shared__ uint32_t sh_a[8], sh_part[6 * 40];
  const size_t i = blockIdx.x;
  if (i >= n) return;
  const int wave = (int)(threadIdx.x >> 6), grp = wave >> 2, sub = wave & 3;
  KYB_COOP_CONSTS(c, 8);
  const bool works = grp == 0 || pubs == nullptr;
  uint32_t ra[16], neg = 0;
  cq h = 0;
  if (works) {
    uint32_t a[8];
    load_words8(a, grp == 0 ? k : x, i);
    const int per = (KYB_BASE64_POS + 3) / 4;
    const int lo = sub * per, hi = (sub + 1) * per < KYB_BASE64_POS ? (sub + 1) * per : KYB_BASE64_POS;
    h = coop_base_mul(c, a, table_coop, neg, lo, hi);
    if (sub > 0 && c.active) sh_part[(grp * 3 + sub - 1) * 40 + 10 * c.row + c.k] = h;
  }
  __syncthreads();
  if (works && sub == 0) {
#pragma unroll 1
    for (int w = 1; w < 4; ++w) {
      const cq o = c.active ? sh_part[(grp * 3 + w - 1) * 40 + 10 * c.row + c.k] : 0u;
      h = coop_add(c, h, coop_to_cached(c, o));
    }
    const cq nq = cnorm(c, c.p2 - h);
    h = (c.row == 0 && neg) ? nq : h;
    fe ax, ay;
    coop_affine(c, h, ax, ay);
    fe_to_words(ra, ay);
    ra[7] ^= fe_is_negative(ax) << 31;
    if (grp == 1 && c.lane == 0) for (int j = 0; j < 8; ++j) sh_a[j] = ra[j];
  }
  __syncthreads();
  if (wave != 0) return;
  if (pubs != nullptr) load_words8(ra + 8, pubs, i);
  else for (int j = 0; j < 8; ++j) ra[8 + j] = sh_a[j];
  uint32_t wx[8], wk[8];
  load_words8(wx, x, i);
  load_words8(wk, k, i);
  sha512_ctx sc;
  sha512_init(sc);
  sha512_words64(sc, ra);
  const uint32_t off = msg_off[i], len = msg_off[i + 1] - off;
  sha512_bytes(sc, msgs + off, len);
  uint32_t dig[16], hh[8], ss[8];
  sha512_final(dig, sc);
  sc_reduce512(hh, dig);
  sc_muladd(ss, wx, hh, wk);
  if (c.lane == 0) {
    store_words8(sig, 2 * i, ra);
    store_words8(sig, 2 * i + 1, ss);
    if (pub_out != nullptr) store_words8(pub_out, i, ra + 8);
    signal_done(df);
  }
}

// Point::mul on a WIRE ENCODING (the Diffie-Hellman batch: unmarshal_binary + mul), one workgroup of two wavefronts per item:
// wavefront 0 runs the ladder on (1 + y : 1 - y) while wavefront 1 extracts the square root for x (see k_verify_coop below);
// a failed decode gives ok = 0 and the neutral element, as k_decode_or_identity + k_mul_coop do in two launches.
__global__ void __launch_bounds__(128)
k_mul_enc_coop(const uint8_t* __restrict__ scalars, const uint8_t* __restrict__ pts_enc, size_t n, uint8_t* __restrict__ out_enc, int32_t* __restrict__ out_ext,
               uint8_t* __restrict__ ok_out, kyb::launch::DoneFlag df, uint4* __restrict__ proj, size_t proj_stride, uint8_t* __restrict__ flags_or, int skip_bits) {
  __shared__ uint32_t sh_x[10], sh_ok[1];
  const size_t i = blockIdx.x;
  if (i >= n) return;
  const uint32_t wave = threadIdx.x >> 6;
  KYB_COOP_CONSTS(c, 2);
  uint32_t w[8];
  load_words8(w, pts_enc, i);
  const cq ONE0 = (c.k == 0 && c.active) ? 1u : 0u;
  fe Y;
  fe_from_words(Y, w);
  cq SX = 0, SZ = 0;
  uint32_t neg = 0, mag[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  if (wave == 0) {
    uint32_t a[8];
    load_words8(a, scalars, i);
    sc_effective(neg, mag, a);
    const cq yq = quad_row_from_fe(c, quad_row_from_fe(c, 0u, 0, Y), 2, Y);
    const cq UWQ = cnorm(c, c.row == 0 ? cadd(ONE0, yq) : (c.row == 2 ? csub(c, ONE0, yq) : 0u));
    coop_ladder_run(c, mag, skip_bits, UWQ, SX, SZ);
  } else {
    ge_p3 P;
    const uint32_t ok = coop_decode_fn{c}(P, w);
    if (c.lane == 0) {
      for (int j = 0; j < 10; ++j) sh_x[j] = P.X.v[j];
      sh_ok[0] = ok;
    }
  }
  __syncthreads();
  if (wave != 0) return;
  const uint32_t ok = sh_ok[0];
  const uint32_t xk = c.active ? sh_x[c.k] : 0u;
  cq PQ = c.row == 0 ? xk : (c.row == 1 ? quad_row_from_fe(c, 0u, 1, Y) : (c.row == 2 ? ONE0 : 0u));
  PQ = ok ? PQ : ((c.row == 1 || c.row == 2) ? ONE0 : 0u);
  uint32_t p_flags;
  const cq M = coop_mont_prep(c, PQ, p_flags);
  const cq RES = coop_mont_recover(c, M, SX, SZ, p_flags, mag[0] & 1u, neg);
  // (proj: the h A of a verification, handed on projective to k_verify_final; flags_or: "the key decodes" joins the flags k_verify_hash wrote)
  coop_finish(c, RES, 0u, out_enc, out_ext, i, proj, proj_stride, 0);
  if (c.lane == 0) {
    if (ok_out != nullptr) ok_out[i] = (uint8_t)ok;
    if (flags_or != nullptr) flags_or[i] = (uint8_t)(flags_or[i] | (ok << 2));
    signal_done(df);
  }
}

// One verification per WORKGROUP of three wavefronts (eddsa_sig.rs:159-212 / schnorr_sig.rs:53-110, verify.h), one launch:
//   wavefront 0   s < L, canonical A, h = SHA-512(R || A || msg) mod L, the ladder for h A — which needs only A's y, since
//                 u = (1 + y) / (1 - y): the x-only state it leaves is the same projective pair whatever common factor U1 and W1
//                 carry, so it runs on (1 + y : 1 - y) while wavefront 1 is still extracting the square root for x;
//   wavefront 1   decode of A (x, the small-order test), then decode and checks of R;
//   wavefront 2   s B.
// After one barrier wavefront 0 builds A's full image (U, V, W) with the x it was handed, recovers y(h A) and tests
// R + h A == s B.  Critical path: hash + 253 ladder steps + recovery + comparison; the five-kernel sequence this replaces
// for up to `coop.verify_max_items` signatures had the decode of A in front of the ladder and four launch gaps.
__global__ void __launch_bounds__(192)
k_verify_coop(const uint8_t* __restrict__ pubs, const uint8_t* __restrict__ sigs, const uint8_t* __restrict__ msgs, const uint32_t* __restrict__ msg_off,
              size_t n, int flavor, const uint32_t* __restrict__ table_coop, uint8_t* __restrict__ status, kyb::launch::DoneFlag df) {
  __shared__ uint32_t sh_ax[10], sh_rx[10], sh_ry[10], sh_sb[30], sh_fl[4];
  const size_t i = blockIdx.x;
  if (i >= n) return;
  const uint32_t wave = threadIdx.x >> 6;
  KYB_COOP_CONSTS(c, 3);
  uint32_t pub[8], sig[16];
  load_words8(pub, pubs, i);
  load_words8(sig, sigs, 2 * i);
  load_words8(sig + 8, sigs, 2 * i + 1);
  const cq ONE0 = (c.k == 0 && c.active) ? 1u : 0u;
  cq SX = 0, SZ = 0;
  uint32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  fe AY;
  fe_from_words(AY, pub);
  if (wave == 0) {
    uint32_t ra[16];
    for (int j = 0; j < 8; ++j) { ra[j] = sig[j]; ra[8 + j] = pub[j]; }
    const uint32_t off = msg_off[i], len = msg_off[i + 1] - off;
    sha512_ctx sc;
    sha512_init(sc);
    sha512_words64(sc, ra);
    sha512_bytes(sc, msgs + off, len);
    uint32_t dig[16];
    sha512_final(dig, sc);
    sc_reduce512(h, dig);
    // (1 + y : 1 - y) in rows 0 / 2
    const cq yq = quad_row_from_fe(c, quad_row_from_fe(c, 0u, 0, AY), 2, AY);
    const cq UWQ = cnorm(c, c.row == 0 ? cadd(ONE0, yq) : (c.row == 2 ? csub(c, ONE0, yq) : 0u));
    coop_ladder_run(c, h, 3, UWQ, SX, SZ);                               // h < L < 2^253
  } else if (wave == 1) {
    ge_p3 A, R;
    const uint32_t a_dec = coop_decode_fn{c}(A, pub);
    const uint32_t a_small = pt_has_small_order(A.Y);
    const uint32_t r_can = pt_is_canonical_w(sig);
    const uint32_t r_dec = coop_decode_fn{c}(R, sig);
    const uint32_t r_small = pt_has_small_order(R.Y);
    ge_p3 id;
    ge_p3_0(id);
    fe_cmov(R.X, id.X, 1u - r_dec); fe_cmov(R.Y, id.Y, 1u - r_dec);
    if (c.lane == 0) {
      for (int j = 0; j < 10; ++j) { sh_ax[j] = A.X.v[j]; sh_rx[j] = R.X.v[j]; sh_ry[j] = R.Y.v[j]; }
      sh_fl[0] = a_dec | (a_small << 1);
      sh_fl[1] = r_can | (r_dec << 1) | (r_small << 2);
    }
  } else {
    uint32_t neg;
    cq hq = coop_base_mul(c, sig + 8, table_coop, neg);
    const cq nq = cnorm(c, c.p2 - hq);
    hq = (c.row == 0 && neg) ? nq : hq;
    if (c.active && c.row < 3) sh_sb[10 * c.row + c.k] = hq;
  }
  __syncthreads();
  if (wave != 0) return;
  const uint32_t a_dec = sh_fl[0] & 1u, a_small = (sh_fl[0] >> 1) & 1u, fr = sh_fl[1];
  // A = (x, y, 1) or, after a failed decode, the neutral element (verify_prep_a_with)
  const uint32_t axk = c.active ? sh_ax[c.k] : 0u;
  cq PQ = c.row == 0 ? axk : (c.row == 1 ? quad_row_from_fe(c, 0u, 1, AY) : (c.row == 2 ? ONE0 : 0u));
  PQ = a_dec ? PQ : ((c.row == 1 || c.row == 2) ? ONE0 : 0u);
  uint32_t p_flags;
  const cq M = coop_mont_prep(c, PQ, p_flags);
  const cq HA = coop_mont_recover(c, M, SX, SZ, p_flags, h[0] & 1u, 0u);
  ge_p2 hA, sB;
  fe RX, RY;
  fe_from_quad_row(c, hA.X, HA, 0); fe_from_quad_row(c, hA.Y, HA, 1); fe_from_quad_row(c, hA.Z, HA, 2);
  for (int j = 0; j < 10; ++j) { RX.v[j] = sh_rx[j]; RY.v[j] = sh_ry[j]; sB.X.v[j] = sh_sb[j]; sB.Y.v[j] = sh_sb[10 + j]; sB.Z.v[j] = sh_sb[20 + j]; }
  const uint32_t eq = verify_final(RX, RY, hA, sB);
  const uint32_t fa = sc_is_canonical_w(sig + 8) | (pt_is_canonical_w(pub) << 1) | (a_dec << 2) | (a_small << 3);
  const uint32_t st = verify_status(fa, fr, flavor);
  if (c.lane == 0) {
    status[i] = (st == 0 && !eq) ? (uint8_t)9 : (uint8_t)st;
    signal_done(df);
  }
}

#ifdef KYB_CROSSCHECK      // the cross-check build only
// Test hook (tests/test_gpu_coop.py, against the lane-level model tools/coop_model.py): one wavefront applies ONE
// cooperative primitive to caller-supplied quads.  op: 0 cmul4(A, B), 1 cnorm(A), 2 cinv(A), 3 mixed addition h = A, entry = B,
// 4 table entry (window, idx, negate) = (B[0], B[1], B[2]) of the radix-64 image, 5 csub(A, B), 6 one ladder step S = A,
// U1 / W1 in rows 0 / 2 of B, swap / bit in B lanes 16 / 17 (returns S'), 7 quad -> fe -> quad round trip of every row, 8 csq4(A),
// 9 fe_invert_gcd_wave(row 0 of A) in every row.
__global__ void __launch_bounds__(64)
k_coop_selftest(int op, const uint32_t* __restrict__ A, const uint32_t* __restrict__ B, uint32_t* __restrict__ out, const uint32_t* __restrict__ table_coop) {
  KYB_COOP_CONSTS(c, 1);
  const cq a = A[c.lane], b = B[c.lane];
  cq r = 0;
  if (op == 0) r = cmul4(c, a, b);
  else if (op == 1) r = cnorm(c, a);
  else if (op == 2) r = cinv(c, a);
  else if (op == 9) {                                                 // the wavefront's safegcd inversion of row 0, result in every row
    fe Z, zi;
    fe_from_quad_row(c, Z, a, 0);
    fe_invert_gcd_wave(zi, Z);
    uint32_t v = 0;
    KYB_UNROLL for (int j = 0; j < 10; ++j) v = (c.k == (uint32_t)j) ? zi.v[j] : v;
    r = c.active ? v : 0u;
  }
  else if (op == 5) r = csub(c, a, b);
  else if (op == 8) r = csq4(c, a);
  else if (op == 4) {
    const uint32_t pos = B[0], idx = B[1], neg = B[2];
    r = pos < 42 ? coop_table_entry<32>(c, table_coop + pos * KYB_COOP_WIN_WORDS, idx, neg) : coop_table_entry<16>(c, table_coop + 42 * KYB_COOP_WIN_WORDS, idx, 0u);
  } else if (op == 3) {
    r = coop_madd(c, madd_idx_init(c), a, b);
  } else if (op == 6) {
    const uint32_t swap = B[16] ^ B[17];                              // pending swap XOR this step's bit
    ladder_state st{bperm(rowperm_idx(c, 0, 0, 2, 2), a), bperm(rowperm_idx(c, 1, 1, 3, 3), a)};
    st = coop_ladder_step(c, ladder_idx_init(c), st, ((c.row == 0 || c.row == 2) && c.active) ? b : 0u, swap);
    const cq fx = bperm(rowperm_idx(c, 0, 0, 2, 2), st.SX), fz = bperm(rowperm_idx(c, 0, 0, 2, 2), st.SZ);
    r = (c.row & 1u) ? fz : fx;                                       // (x2', z2', x3', z3')
  } else if (op >= 16 && op < 24) {
    // timing chains (tools/coop_primitive_times.py): B[3] dependent repetitions of one primitive
    const uint32_t reps = B[3];
    const ladder_idx li = ladder_idx_init(c);
    const madd_idx mi = madd_idx_init(c);
    const int Irot = rowperm_idx(c, 1, 2, 3, 0);
    ladder_state st{a, b};
    r = a;
#define KYB_CHAIN(stmt) _Pragma("unroll 1") for (uint32_t t = 0; t < reps; ++t) { stmt; }
    if (op == 16) { KYB_CHAIN(r = cmul4(c, r, b)) }
    else if (op == 17) { KYB_CHAIN(r = csq4(c, r)) }
    else if (op == 18) { KYB_CHAIN(r = cnorm(c, r + b)) }
    else if (op == 19) { KYB_CHAIN(st = coop_ladder_step(c, li, st, b, t & 1u)) }
    else if (op == 20) { KYB_CHAIN(r = bperm(Irot, r)) }
    else if (op == 21) { KYB_CHAIN(r = dpp0<KYB_DPP_ROW_SHR(1)>(r) + b) }
    else if (op == 22) { KYB_CHAIN(r = coop_madd(c, mi, r, b)) }
    else { KYB_CHAIN(r = r * b + t) }                                    // 23: one dependent v_mul_lo + add
#undef KYB_CHAIN
    if (op == 19) r = st.SX ^ st.SZ;
  } else if (op == 7) {
    fe f[4];
    for (uint32_t q = 0; q < 4; ++q) fe_from_quad_row(c, f[q], a, q);
    for (uint32_t q = 0; q < 4; ++q) r = quad_row_from_fe(c, r, q, f[q]);
  }
  out[c.lane] = r;
}
#endif

namespace kyb { namespace launch {
hipError_t msm_bases_coop(hipStream_t st, const int32_t* pts_ext, size_t t, uint32_t* bases) {
  hipLaunchKernelGGL(k_msm_bases_coop, dim3((unsigned)t), dim3(64), 0, st, pts_ext, t, bases);
  return hipGetLastError();
}
#ifdef KYB_CROSSCHECK
hipError_t diag_phase_stamps(uint64_t* buf) { return hipMemcpyToSymbol(HIP_SYMBOL(kyb_phase_buf), &buf, sizeof(buf)); }
hipError_t coop_selftest(hipStream_t st, int op, const uint32_t* A, const uint32_t* B, uint32_t* out, const uint32_t* table_coop) {
  hipLaunchKernelGGL(k_coop_selftest, dim3(1), dim3(64), 0, st, op, A, B, out, table_coop);
  return hipGetLastError();
}
#endif
hipError_t finish_coop(hipStream_t st, const uint4* proj, size_t stride, const int32_t* pts_ext, size_t n, uint8_t* oenc, int32_t* oext, size_t src_mul, DoneFlag df,
                       bool ext_proj) {
  hipLaunchKernelGGL(k_finish_coop, dim3((unsigned)n), dim3(64), 0, st, proj, stride, pts_ext, n, oenc, oext, src_mul, ext_proj ? 1 : 0, df);
  return hipGetLastError();
}
hipError_t finish_wave(hipStream_t st, const uint4* proj, size_t stride, const int32_t* pts_ext, size_t n, uint8_t* oenc, int32_t* oext, size_t src_mul) {
  hipLaunchKernelGGL(k_finish_wave, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, st, proj, stride, pts_ext, n, oenc, oext, src_mul);
  return hipGetLastError();
}
hipError_t sum_coop(hipStream_t st, const uint32_t* part, const int32_t* pts_ext, size_t m, size_t t, uint8_t* oenc, int32_t* oext, bool ext_proj, DoneFlag df) {
  hipLaunchKernelGGL(k_sum_coop, dim3((unsigned)m), dim3(64), 0, st, part, pts_ext, m, t, oenc, oext, ext_proj ? 1 : 0, df);
  return hipGetLastError();
}
hipError_t decode_coop(hipStream_t st, const uint8_t* enc, size_t n, int32_t* out_ext, uint8_t* ok, bool or_identity, DoneFlag df) {
  hipLaunchKernelGGL(k_decode_coop, dim3((unsigned)n), dim3(64), 0, st, enc, n, out_ext, ok, or_identity ? 1 : 0, df);
  return hipGetLastError();
}
hipError_t verify_prep_coop(hipStream_t st, const uint8_t* pubs, const uint8_t* sigs, const uint8_t* msgs, const uint32_t* off, size_t n,
                            uint8_t* flags_a, uint8_t* hbuf, uint8_t* sbuf, int32_t* a_ext) {
  hipLaunchKernelGGL(k_verify_prep_coop, dim3((unsigned)n), dim3(64), 0, st, pubs, sigs, msgs, off, n, flags_a, hbuf, sbuf, a_ext);
  return hipGetLastError();
}
hipError_t verify_prep_r_coop(hipStream_t st, const uint8_t* sigs, size_t n, uint8_t* flags_r, uint4* proj, size_t stride, size_t offset) {
  hipLaunchKernelGGL(k_verify_prep_r_coop, dim3((unsigned)n), dim3(64), 0, st, sigs, n, flags_r, proj, stride, offset);
  return hipGetLastError();
}
hipError_t poly_eval_coop(hipStream_t st, const int32_t* commits, int t, const uint32_t* idx, size_t n, int nbits, size_t per_poly,
                          uint8_t* oenc, int32_t* oext, DoneFlag df, bool ext_proj) {
  hipLaunchKernelGGL(k_poly_eval_coop, dim3((unsigned)n), dim3(64), 0, st, commits, t, idx, n, nbits, per_poly, oenc, oext, ext_proj ? 1 : 0, df);
  return hipGetLastError();
}
hipError_t sign_coop(hipStream_t st, const uint8_t* x, const uint8_t* k, const uint8_t* pubs, const uint8_t* msgs, const uint32_t* off, size_t n,
                     uint8_t* sig, uint8_t* pub_out, const uint32_t* table_coop, DoneFlag df) {
  hipLaunchKernelGGL(k_sign_coop, dim3((unsigned)n), dim3(512), 0, st, x, k, pubs, msgs, off, n, sig, pub_out, table_coop, df);
  return hipGetLastError();
}
hipError_t mul_enc_coop(hipStream_t st, const uint8_t* sc, const uint8_t* penc, size_t n, uint8_t* oenc, int32_t* oext, uint8_t* ok, DoneFlag df,
                        uint4* proj, size_t proj_stride, uint8_t* flags_or, int skip_bits) {
  hipLaunchKernelGGL(k_mul_enc_coop, dim3((unsigned)n), dim3(128), 0, st, sc, penc, n, oenc, oext, ok, df, proj, proj_stride, flags_or, skip_bits);
  return hipGetLastError();
}
hipError_t verify_coop(hipStream_t st, const uint8_t* pubs, const uint8_t* sigs, const uint8_t* msgs, const uint32_t* off, size_t n, int flavor,
                       const uint32_t* table_coop, uint8_t* status, DoneFlag df) {
  hipLaunchKernelGGL(k_verify_coop, dim3((unsigned)n), dim3(192), 0, st, pubs, sigs, msgs, off, n, flavor, table_coop, status, df);
  return hipGetLastError();
}
hipError_t poly_eval_seg(hipStream_t st, const int32_t* commits, int t, const uint32_t* idx, size_t n, size_t per_poly, int len, int segs,
                         uint32_t* part, uint8_t* oenc, int32_t* oext, DoneFlag df, bool ext_proj) {
  hipLaunchKernelGGL(k_poly_eval_seg, dim3((unsigned)(n * (size_t)segs)), dim3(64), 0, st, commits, t, idx, n, per_poly, len, segs, part);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(k_poly_eval_sum, dim3((unsigned)n), dim3(64), 0, st, part, n, segs, oenc, oext, ext_proj ? 1 : 0, df);
  return hipGetLastError();
}
hipError_t mul_coop(hipStream_t st, const uint8_t* sc, const int32_t* pext, size_t n, uint8_t* oenc, int32_t* oext, int skip_bits,
                    uint4* proj, size_t proj_stride, size_t proj_offset, DoneFlag df, size_t pt_mod, int pieces, bool ext_proj, uint32_t* part, uint32_t* pieces_buf) {
  if (pieces != 1 && (pieces != 4 || pieces_buf == nullptr || part != nullptr)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(k_mul_coop, dim3((unsigned)(n * (size_t)pieces)), dim3(64u), 0, st, sc, pext, n, oenc, oext, skip_bits, proj, proj_stride, proj_offset, pt_mod,
                     pieces, ext_proj ? 1 : 0, part, pieces_buf, df);
  return hipGetLastError();
}
hipError_t mul_base_coop(hipStream_t st, const uint8_t* sc, size_t n, uint8_t* oenc, int32_t* oext, const uint32_t* table_coop,
                         uint4* proj, size_t proj_stride, size_t proj_offset, const uint8_t* sc_b, size_t n_b, DoneFlag df, int waves, bool ext_proj) {
  hipLaunchKernelGGL(k_mul_base_coop, dim3((unsigned)(n + n_b)), dim3(64u * (unsigned)waves), 0, st, sc, sc_b, n, n + n_b, oenc, oext, table_coop, proj, proj_stride, proj_offset,
                     waves, ext_proj ? 1 : 0, df);
  return hipGetLastError();
}
}}  // namespace kyb::launch
