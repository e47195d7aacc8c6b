// Lane-cooperative GF(2^255-19) arithmetic for SMALL batches (device only).
//
// The batch kernels give every item one lane: a variable-base multiplication is then a serial chain of ~330,000
// instructions that ONE wavefront walks alone (0.9 ms), however few items the call carries — which is what unmodified
// protocol code does (Point::mul one at a time: poly.rs:198, vss.rs:300-303, schnorr_sig.rs:32-35, key_pair.rs:66).
// Here ONE ITEM OWNS A WAVEFRONT: a field element is spread over ten lanes (limb k in lane k of a 16-lane row), a
// wavefront holds four rows, and the four independent multiplications of a ladder level / of a point addition run side by
// side, so the dependent chain is ~60 instructions per level instead of ~140 per multiplication.
//
//   quad (`cq`)    one VGPR: row r = lanes 16r .. 16r+15 holds limbs 0..9 of element r in its first ten lanes; lanes 10..15 of a row
//                  are DON'T-CARE (nothing an active lane computes ever depends on them)
//   cmul4(F, G)    four products F_r * G_r.  Lane k forms column k: sum_i f_i * g_{(k-i) mod 10} * m(i,k) with the 2x / 19x / 38x factors
//                  m(i,k) of the radix-2^25.5 product (fe25519.h).  Both operands travel through a per-wavefront scratch in LDS MEMORY
//                  (320 words): F is stored in four scaled copies (x1, x2, x19, x38 — so F must be TIGHT, <= 1.01T: 38 f < 2^32; G may be
//                  lazy, <= 4T), lane k reads term i's f from the copy that carries m(i,k) at a per-lane constant address, and g_i by a
//                  broadcast read of its row.  That is 5 ds_write + 20 ds_read and, in the VALU stream, 3 instructions for the copies,
//                  10 multiply-adds and the carries.  LDS instructions of one wavefront execute in order, so no barrier is needed; a lone
//                  wavefront pays ~6 cycles per VALU instruction and far less per LDS instruction (tools/coop_primitive_times.py: 277
//                  cycles against 383 for the version that broadcast by DPP, rotated by ds_bpermute and multiplied the factors in).
//                  Carries travel between NEIGHBOURING lanes, i.e. by DPP row shifts (row_shr:1 / row_shr:2; the wrap from limbs 8, 9 to
//                  limbs 0, 1 by row_shl:8 / row_shl:9 times 19): a three-way split of the 64-bit column (own limb / next limb / limb
//                  after that) and one more light pass.
//   cnorm(V)       that light pass alone: any limbs < 2^31 -> tight
//   rows are moved with ds_bpermute_b32 and per-lane index constants; additions are ONE instruction for four elements.
//
// Measured for ONE wavefront on an idle chip (tools/coop_primitive_times.py, profiles/r02/coop_primitive_times.log): a lone wavefront
// pays ~6 cycles per VALU instruction whatever it does, ~45 for a dependent ds_bpermute round trip, and little for LDS reads and
// writes that are issued in a block — which is why the operands of the products live in LDS memory.
//
// Same limb format and bounds notation as fe25519.h; results are bit-identical to the one-lane code (tests compare both
// paths with the oracle).  Constant time, checked on the compiled code by tools/ct_check.py: no branch and no memory address (global
// or LDS) depends on a scalar.  Secret-dependent data moves in exactly two ways: the LANE SELECT of ds_bpermute_b32 (the ladder's
// conditional swap exchanges two rows; coop_table_entry pulls the wanted table word: a lone permute takes 60 cycles whatever the
// pattern, tools/microbench/bpermute_patterns.hip) and csel() below, a v_bfi_b32 under an opaque mask — never `flag ? a : b` on a
// secret-derived flag: all lanes of a wavefront serve ONE item, so every such flag is wave-uniform and the compiler turns the
// ternary into a scalar branch (round 2's y-recovery had them for its exceptional cases).
// All of this needs the FULL wavefront active: a DPP or ds_bpermute read of a lane that EXEC has switched off returns 0, so
// no cross-lane operation may sit under a lane-dependent branch (`cond ? move(a) : move(b)` must be written move, move, select).
#pragma once
#include "fe25519.h"

namespace kyb {
namespace coop {

typedef uint32_t cq;      // one lane's share of a quad

// Select by a flag (0 / 1) that may derive from secret data: flag ? a : b.  Every lane of a wavefront works on the SAME item here, so such
// flags are wave-uniform, and the compiler, seeing that, turns a plain `flag ? a : b` into a scalar BRANCH around one arm
// (tools/ct_check.py found them in the exceptional-case selects of the y-recovery).  The mask is made opaque first; what is left is one
// v_bfi_b32 whatever the flag.
__device__ __forceinline__ cq csel(uint32_t flag01, cq a, cq b) {
  uint32_t m = 0u - flag01;
  asm volatile("" : "+v"(m));
  return (a & m) | (b & ~m);
}

// per-lane constants of the cooperative arithmetic (computed once per kernel)
struct lane_consts {
  uint32_t lane, row, k;          // k = lane & 15
  uint32_t active;                // k < 10
  uint32_t mask, bits;            // of limb k
  uint32_t mask_next;             // of limb k+1 (mod 10)
  uint32_t p2;                    // limb k of 2p (0 in inactive lanes)
  uint32_t w19a, w19b;            // 19 where the carry from limb k-1 / k-2 crosses the wrap (k == 0 / k < 2), else 0
  uint32_t* wlds;                 // this wavefront's 384-word scratch in LDS (KYB_COOP_LDS_WORDS)
  int mf[10];                     // cmul4: word index of term i's first-operand limb, in the copy carrying m(i, k) (copies at 0 / 64 / 128 / 192)
  int mg;                         // cmul4: word index of limb 0 of this row's second operand (slot at 256)
  int sqa[6], sqb[6];             // csq4: word indices of limbs a_t = ceil(k/2) + t (in the copy carrying its factor) and b_t = floor(k/2) - t (x1 or x2)
};

constexpr int KYB_COOP_LDS_WORDS = 384;
__device__ __forceinline__ void lane_consts_init(lane_consts& c, uint32_t* wlds) {
  const uint32_t p2v[10] = KYB_FE_2P;
  c.lane = threadIdx.x & 63u;
  c.row = c.lane >> 4;
  c.k = c.lane & 15u;
  c.active = c.k < 10u;
  const uint32_t k = c.active ? c.k : 0u;
  c.bits = (k & 1u) ? 25u : 26u;
  c.mask = c.active ? ((1u << c.bits) - 1u) : 0u;
  c.mask_next = c.active ? ((k & 1u) ? 0x3ffffffu : 0x1ffffffu) : 0u;      // limb k+1 has the other width
  uint32_t p2 = 0;
  KYB_UNROLL for (int j = 0; j < 10; ++j) p2 = (k == (uint32_t)j) ? p2v[j] : p2;
  c.p2 = c.active ? p2 : 0u;
  c.w19a = c.k == 0u ? 19u : 0u;
  c.w19b = c.k < 2u ? 19u : 0u;
  const uint32_t base = c.row << 4;
  c.wlds = wlds;
  auto copy_of = [](uint32_t factor) { return factor == 1u ? 0u : (factor == 2u ? 64u : (factor == 19u ? 128u : (factor == 38u ? 192u : 320u))); };      // 320: zeros
  KYB_UNROLL for (int i = 0; i < 10; ++i) {
    // column k, term i: g_i (broadcast) times f_j, j = (k - i) mod 10, times m = 19 if i + j >= 10, 2 if i and j odd
    const uint32_t j = (k + 10u - (uint32_t)i) % 10u;
    const uint32_t wrap = (uint32_t)i > k, oo = ((uint32_t)i & 1u) & (j & 1u);
    c.mf[i] = c.active ? (int)(copy_of((wrap ? 19u : 1u) * (oo ? 2u : 1u)) + base + j) : 320;      // lanes 10..15 multiply by the zero words
  }
  c.mg = (int)(256u + base);
  KYB_UNROLL for (int t = 0; t < 6; ++t) {
    const uint32_t a = ((k + 1u) / 2u + (uint32_t)t) % 10u, b = (k / 2u + 10u - (uint32_t)t) % 10u;       // a + b = k (mod 10)
    const uint32_t wrap = a + b >= 10u, oo = (a & 1u) & (b & 1u);
    const uint32_t dup = (t == 5) & (k & 1u);                         // odd k: {a_5, b_5} = {a_4, b_4}
    const uint32_t twice0 = (t == 0) & (k & 1u);                      // t = 0, odd k: a proper pair, counted twice on the a side (no wrap, never both odd)
    const uint32_t fac = (c.active && !dup) ? ((wrap ? 19u : 1u) * (oo ? 2u : 1u) * (twice0 ? 2u : 1u)) : 0u;
    c.sqa[t] = (int)(copy_of(fac) + base + a);
    c.sqb[t] = (int)(((t >= 1 && t <= 4) ? 64u : 0u) + base + b);    // the pair's 2 rides on b in steps 1..4
  }
  if (c.lane < 64u) wlds[320 + c.lane] = 0u;                            // the zero words a duplicate term reads
}

__device__ __forceinline__ cq bperm(int idx, cq v) { return (cq)__builtin_amdgcn_ds_bpermute(idx, (int)v); }

// DPP moves inside a 16-lane row; a lane whose source falls outside the row gets 0.  Controls (ISA 'DPP_CTRL'):
//   row_shl:n  lane k reads lane k+n      row_shr:n  lane k reads lane k-n      row_newbcast:n  every lane reads lane n of its row
// (checked on the device by tools/microbench/dpp_probe.hip)
#define KYB_DPP_ROW_SHL(n) (0x100 + (n))
#define KYB_DPP_ROW_SHR(n) (0x110 + (n))
#define KYB_DPP_ROW_BCAST(n) (0x150 + (n))
template <int CTRL>
__device__ __forceinline__ cq dpp0(cq v) { return (cq)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xf, 0xf, true); }

// byte-index vector that makes row r read row p_r (same limb position)
__device__ __forceinline__ int rowperm_idx(const lane_consts& c, uint32_t p0, uint32_t p1, uint32_t p2, uint32_t p3) {
  const uint32_t src = c.row == 0 ? p0 : (c.row == 1 ? p1 : (c.row == 2 ? p2 : p3));
  return (int)(((src << 4) | c.k) << 2);
}

// light carry pass: limbs < 2^31 in, tight (<= mask + 19 * 64) out.  Five VALU instructions, no LDS.
__device__ __forceinline__ cq cnorm(const lane_consts& c, cq v) {
  const uint32_t lo = v & c.mask;
  const uint32_t cy = v >> c.bits;                                             // < 2^7
  const uint32_t r = lo + dpp0<KYB_DPP_ROW_SHR(1)>(cy);                        // carry of limb k-1 (v_add_u32_dpp)
  return r + __umul24(dpp0<KYB_DPP_ROW_SHL(9)>(cy), c.w19a);                   // limb 0: 19 x the carry of limb 9 (v_mad_u32_u24)
}

// carry propagation of four 64-bit column vectors -> tight quad
__device__ __forceinline__ cq ccarry(const lane_consts& c, uint64_t s) {
  const uint32_t lo = (uint32_t)s & c.mask;
  const uint64_t t = s >> c.bits;
  const uint32_t mid = (uint32_t)t & c.mask_next;            // belongs to limb k+1
  const uint32_t hi = (uint32_t)(s >> 51);                   // belongs to limb k+2 (26 + 25 bits up); < 2^13
  uint32_t v = lo + dpp0<KYB_DPP_ROW_SHR(1)>(mid);
  v += dpp0<KYB_DPP_ROW_SHR(2)>(hi);
  v += dpp0<KYB_DPP_ROW_SHL(9)>(mid) * c.w19a;               // limb 0 <- 19 x mid of limb 9
  v += __umul24(dpp0<KYB_DPP_ROW_SHL(8)>(hi), c.w19b);       // limbs 0, 1 <- 19 x hi of limbs 8, 9
  return cnorm(c, v);                                        // v < 2^26 + 19 * 2^26 + 20 * 2^13
}

// four products F_r * G_r; F tight, G <= 4T
__device__ __forceinline__ cq cmul4(const lane_consts& c, cq F, cq G) {
  uint32_t* w = c.wlds;
  const uint32_t f19 = F * 19u;
  w[c.lane] = F;
  w[64 + c.lane] = F + F;
  w[128 + c.lane] = f19;
  w[192 + c.lane] = f19 + f19;
  w[256 + c.lane] = G;
  __builtin_amdgcn_wave_barrier();                          // (compiler ordering only: the LDS queue of a wavefront is in order)
  uint32_t fm[10], gb[10];
  KYB_UNROLL for (int i = 0; i < 10; ++i) { gb[i] = w[c.mg + i]; fm[i] = w[c.mf[i]]; }
  __builtin_amdgcn_wave_barrier();
  uint64_t acc = (uint64_t)gb[0] * fm[0];
  KYB_UNROLL for (int i = 1; i < 10; ++i) acc += (uint64_t)gb[i] * fm[i];
  return ccarry(c, acc);
}
// four squares F_r^2, F tight; bit-identical to cmul4(c, F, F) (the same column sums) with the symmetric terms taken once:
// column k = sum over the unordered pairs {a, b}, a + b = k (mod 10), of f_a f_b m(a, b) x (2 if a != b).  Step t = 0..5 takes
// a = ceil(k/2) + t, b = floor(k/2) - t: for even k the steps 0 and 5 are the two squares and 1..4 the four pairs; for odd k
// 0..4 are the five pairs and step 5 repeats step 4 (read from the zero words).  The pair's 2 rides on b (read from the x2 copy)
// in steps 1..4 and on the a side in step 0 (where nothing wraps), so every a-side factor is one of 1, 2, 19, 38: 6
// multiply-adds instead of 10, 4 ds_write + 12 ds_read.
__device__ __forceinline__ cq csq4(const lane_consts& c, cq F) {
  uint32_t* w = c.wlds;
  const uint32_t f19 = F * 19u;
  w[c.lane] = F;
  w[64 + c.lane] = F + F;
  w[128 + c.lane] = f19;
  w[192 + c.lane] = f19 + f19;
  __builtin_amdgcn_wave_barrier();
  uint32_t fa[6], fb[6];
  KYB_UNROLL for (int t = 0; t < 6; ++t) { fa[t] = w[c.sqa[t]]; fb[t] = w[c.sqb[t]]; }
  __builtin_amdgcn_wave_barrier();
  uint64_t acc = (uint64_t)fa[0] * fb[0];
  KYB_UNROLL for (int t = 1; t < 6; ++t) acc += (uint64_t)fa[t] * fb[t];
  return ccarry(c, acc);
}

// element-wise on four elements at once
__device__ __forceinline__ cq cadd(cq a, cq b) { return a + b; }
__device__ __forceinline__ cq csub(const lane_consts& c, cq a, cq b) { return a + (c.p2 - b); }      // b <= 2T; result <= bound(a) + 2T

// ---- replicated one-lane form <-> quad ----------------------------------------------------------------------------
// every lane holds the same `fe` (the irregular parts of a multiplication run replicated on all lanes)
__device__ __forceinline__ cq quad_row_from_fe(const lane_consts& c, cq q, uint32_t r, const fe& f) {
  uint32_t v = 0;
  KYB_UNROLL for (int j = 0; j < 10; ++j) v = (c.k == (uint32_t)j) ? f.v[j] : v;
  return (c.row == r && c.active) ? v : q;
}
__device__ __forceinline__ void fe_from_quad_row(const lane_consts& c, fe& f, cq q, uint32_t r) {
  (void)c;                                                   // (ten readlane-style fetches; outside the loops)
  KYB_UNROLL for (int j = 0; j < 10; ++j) f.v[j] = bperm((int)(((r << 4) | (uint32_t)j) << 2), q);
}

__device__ __forceinline__ cq csqn(const lane_consts& c, cq f, int n) {
  cq h = csq4(c, f);
#pragma unroll 1
  for (int i = 1; i < n; ++i) h = csq4(c, h);
  return h;
}
// z^(2^250 - 1) and z^11 of all four rows: the shared prefix of both exponentiations (fe_pow_250_1, fe25519.h)
__device__ __forceinline__ void cpow_250_1(const lane_consts& c, cq& z250, cq& z11, cq z) {
  const cq z2 = csq4(c, z);
  cq t = csqn(c, z2, 2);
  const cq z9 = cmul4(c, t, z);
  z11 = cmul4(c, z9, z2);
  t = csq4(c, z11);
  const cq z5 = cmul4(c, t, z9);               // 2^5 - 1
  t = csqn(c, z5, 5);
  const cq z10 = cmul4(c, t, z5);              // 2^10 - 1
  t = csqn(c, z10, 10);
  const cq z20 = cmul4(c, t, z10);             // 2^20 - 1
  t = csqn(c, z20, 20);
  t = cmul4(c, t, z20);                        // 2^40 - 1
  t = csqn(c, t, 10);
  const cq z50 = cmul4(c, t, z10);             // 2^50 - 1
  t = csqn(c, z50, 50);
  const cq z100 = cmul4(c, t, z50);            // 2^100 - 1
  t = csqn(c, z100, 100);
  t = cmul4(c, t, z100);                       // 2^200 - 1
  t = csqn(c, t, 50);
  z250 = cmul4(c, t, z50);                     // 2^250 - 1
}
// z^(p-2) of all four rows (fe_invert): 254 squarings + 11 products.  Rows holding 0 stay 0.
__device__ __forceinline__ cq cinv(const lane_consts& c, cq z) {
  cq z250, z11;
  cpow_250_1(c, z250, z11, z);
  return cmul4(c, csqn(c, z250, 5), z11);      // 2^255 - 21
}
// z^((p-5)/8) = z^(2^252 - 3) of all four rows (fe_pow22523): the square-root chain of the point decode
__device__ __forceinline__ cq cpow22523(const lane_consts& c, cq z) {
  cq z250, z11;
  cpow_250_1(c, z250, z11, z);
  return cmul4(c, csqn(c, z250, 2), z);
}

}  // namespace coop
}  // namespace kyb
