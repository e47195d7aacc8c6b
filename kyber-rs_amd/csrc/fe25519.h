// Field arithmetic in GF(2^255-19) for the MI355X Ed25519 engine.
//
// Replaces (functionally) /root/reference src/group/edwards25519/fe.rs — fe_mul (fe.rs:299-535),
// fe_square (fe.rs:544-688), fe_invert (fe.rs:857-944), fe_pow22523 (fe.rs:946-1035),
// fe_from_bytes / fe_to_bytes (fe.rs:67-122,147-238) — but is NOT a transcription of it:
//
//  * limbs are UNSIGNED, radix 2^25.5 (26,25,26,25,... bits) in ten 32-bit VGPRs; products are
//    accumulated with v_mad_u64_u32 (32x32+64->64), the only wide integer multiply gfx950 has.
//    A "51-bit limb" (north_star wording) is one even/odd limb pair; gfx950 has no 64x64 multiplier,
//    so the pair is kept as two VGPRs and one 51x51 product is four v_mad_u64_u32.
//  * columns are evaluated serially and the carry of column k is the 64-bit addend of the first
//    multiply-add of column k+1, so carry propagation costs 3 VALU ops per limb instead of 8.
//  * subtraction adds a limb-wise multiple of p (no signed limbs, no rounding carries).
//
// Bounds notation: T_i = 2^26 (even i) / 2^25 (odd i).  "kT" means limb_i <= k*T_i for every i.
//   tight   : <= 1.01T   (output of fe_mul / fe_sq / fe_reduce_weak)
//   fe_mul(h, f, g): f <= 6T, g <= 3.3T   (19*g_i and 2*f_odd must fit 32 bits, column sums 64 bits)
//   fe_sq(h, f)    : f <= 3.3T
// The same source is compiled by g++ for tests/ (KYB_HOST_TEST) where every multiply is shadowed by a
// 128-bit overflow check; that host build is test infrastructure, never a product code path.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define KYB_HD __host__ __device__ __forceinline__
#define KYB_UNROLL _Pragma("unroll")
#else
#define KYB_HD inline
#define KYB_UNROLL
#endif

#include "consts.inc"

namespace kyb {

struct fe {
  uint32_t v[10];
};

#define KYB_BITS(i) (((i) & 1) ? 25 : 26)
#define KYB_MASK(i) (((i) & 1) ? 0x1ffffffu : 0x3ffffffu)

#if defined(KYB_HOST_TEST)
// overflow shadow check for the host-compiled test build
extern "C" void kyb_host_overflow(const char* what);
static inline uint64_t kyb_mad(uint32_t a, uint32_t b, uint64_t c) {
  unsigned __int128 w = (unsigned __int128)a * b + c;
  if (w >> 64) kyb_host_overflow("mad64");
  return (uint64_t)w;
}
static inline uint32_t kyb_mul32(uint32_t a, uint32_t b, const char* what) {
  uint64_t w = (uint64_t)a * b;
  if (w >> 32) kyb_host_overflow(what);
  return (uint32_t)w;
}
static inline uint32_t kyb_sub32(uint32_t a, uint32_t b, const char* what) {
  if (b > a) kyb_host_overflow(what);
  return a - b;
}
static inline uint32_t kyb_add32(uint32_t a, uint32_t b, const char* what) {
  uint64_t w = (uint64_t)a + b;
  if (w >> 32) kyb_host_overflow(what);
  return (uint32_t)w;
}
#else
// LLVM reassociates each column into "products first, carry last", which costs one v_lshl_add_u64 per
// column on top of the mads.  -DKYB_CHAIN_BARRIER makes every partial sum opaque (empty asm) so the
// carry becomes the addend of the column's first v_mad_u64_u32 and those adds disappear — but hipcc
// then pads every asm boundary that feeds the next VALU with an s_nop, and the A/B on MI355X
// (profiles/r01/ab_chain_barrier.log) came out 2-3 % SLOWER for k_mul and equal for k_mul_base, so it
// is off by default.
KYB_HD uint64_t kyb_mad(uint32_t a, uint32_t b, uint64_t c) {
  uint64_t r = (uint64_t)a * b + c;
#if defined(__HIP_DEVICE_COMPILE__) && defined(KYB_CHAIN_BARRIER)
  asm("" : "+v"(r));
#endif
  return r;
}
KYB_HD uint32_t kyb_mul32(uint32_t a, uint32_t b, const char*) { return a * b; }
KYB_HD uint32_t kyb_sub32(uint32_t a, uint32_t b, const char*) { return a - b; }
KYB_HD uint32_t kyb_add32(uint32_t a, uint32_t b, const char*) { return a + b; }
#endif

KYB_HD uint32_t kyb_x19(uint32_t a) { return kyb_mul32(a, 19u, "x19"); }
// 2*a: LLVM canonicalises x+x into v_lshlrev_b32 (half rate on gfx950,
// profiles/r01_valu_rates2_mi355x.jsonl).  -DKYB_X2_ADD hides one operand behind an empty asm to get a
// full-rate v_add_u32; same A/B as above: no measurable gain (the asm pad eats it), off by default.
KYB_HD uint32_t kyb_x2(uint32_t a, const char* what) {
#if defined(__HIP_DEVICE_COMPILE__) && defined(KYB_X2_ADD)
  uint32_t t = a;
  asm("" : "+v"(t));
  return a + t;
#else
  return kyb_add32(a, a, what);
#endif
}

// ---- device-only: one asm statement per COLUMN ---------------------------------------------------
// hipcc reassociates a column into "products first, carry last" and then needs a separate 64-bit add
// per column (v_lshl_add_u64, half rate).  Writing the column's mads as one asm statement keeps the
// running sum — carry of the previous column included — as the addend chain (no extra add), and costs
// one asm-boundary pad per column instead of one per mad (the per-mad barrier variant lost to its
// s_nops, profiles/r01/ab_chain_barrier.log).  (The host build of tests/hostcheck/ takes the plain C++ columns.)
#if defined(__HIP_DEVICE_COMPILE__)
#define KYB_ASM_COLUMNS 1
// The carry-out of v_mad_u64_u32 is never used (the sums stay below 2^64 by the bound calculus); it goes to VCC.  In a PURE multiply-add
// stream the form that writes an SGPR pair instead issues 7 % faster (profiles/r02/valu_rates_selfconsistent_mi355x.jsonl: 30.4 against
// 28.2 T mad/s); in these kernels, where a third of the instructions are not multiply-adds, it made no difference
// (profiles/r03/ab_carry_sgpr.log: 8.642 against 8.638 ms per 2^20 variable-base mults), so VCC it is.
// operands by name: %[acc]; the column's factors %[a1].. / %[b1]..
#define KYB_M1(i) "v_mad_u64_u32 %[acc], vcc, %[a" #i "], %[b" #i "], %[acc]\n\t"
__device__ __forceinline__ uint64_t kyb_col10(uint64_t acc, const uint32_t* A, const uint32_t* B) {
  asm(KYB_M1(1) KYB_M1(2) KYB_M1(3) KYB_M1(4) KYB_M1(5) KYB_M1(6) KYB_M1(7) KYB_M1(8) KYB_M1(9) KYB_M1(10)
      : [acc] "+v"(acc)
      : [a1] "v"(A[0]), [a2] "v"(A[1]), [a3] "v"(A[2]), [a4] "v"(A[3]), [a5] "v"(A[4]), [a6] "v"(A[5]), [a7] "v"(A[6]), [a8] "v"(A[7]), [a9] "v"(A[8]), [a10] "v"(A[9]),
        [b1] "v"(B[0]), [b2] "v"(B[1]), [b3] "v"(B[2]), [b4] "v"(B[3]), [b5] "v"(B[4]), [b6] "v"(B[5]), [b7] "v"(B[6]), [b8] "v"(B[7]), [b9] "v"(B[8]), [b10] "v"(B[9])
      : "vcc");
  return acc;
}
#define KYB_M0(i) "v_mad_u64_u32 %[acc], vcc, %[a" #i "], %[b" #i "], 0\n\t"
// first column of a product: the accumulator starts at the literal 0 (no v_mov_b64 to clear a register pair)
__device__ __forceinline__ uint64_t kyb_col10z(const uint32_t* A, const uint32_t* B) {
  uint64_t acc;
  asm(KYB_M0(1) KYB_M1(2) KYB_M1(3) KYB_M1(4) KYB_M1(5) KYB_M1(6) KYB_M1(7) KYB_M1(8) KYB_M1(9) KYB_M1(10)
      : [acc] "=&v"(acc)
      : [a1] "v"(A[0]), [a2] "v"(A[1]), [a3] "v"(A[2]), [a4] "v"(A[3]), [a5] "v"(A[4]), [a6] "v"(A[5]), [a7] "v"(A[6]), [a8] "v"(A[7]), [a9] "v"(A[8]), [a10] "v"(A[9]),
        [b1] "v"(B[0]), [b2] "v"(B[1]), [b3] "v"(B[2]), [b4] "v"(B[3]), [b5] "v"(B[4]), [b6] "v"(B[5]), [b7] "v"(B[6]), [b8] "v"(B[7]), [b9] "v"(B[8]), [b10] "v"(B[9])
      : "vcc");
  return acc;
}
__device__ __forceinline__ uint64_t kyb_col6z(const uint32_t* A, const uint32_t* B) {
  uint64_t acc;
  asm(KYB_M0(1) KYB_M1(2) KYB_M1(3) KYB_M1(4) KYB_M1(5) KYB_M1(6)
      : [acc] "=&v"(acc)
      : [a1] "v"(A[0]), [a2] "v"(A[1]), [a3] "v"(A[2]), [a4] "v"(A[3]), [a5] "v"(A[4]), [a6] "v"(A[5]),
        [b1] "v"(B[0]), [b2] "v"(B[1]), [b3] "v"(B[2]), [b4] "v"(B[3]), [b5] "v"(B[4]), [b6] "v"(B[5])
      : "vcc");
  return acc;
}
// 2*a as v_add_u32 a, a: LLVM canonicalises x + x into v_lshlrev_b32, which issues at 4.4 cycles per wave-instruction on
// gfx950 against 2.45 for the add (tools/microbench/valu_rates.hip).  One asm statement per GROUP of doublings, so there
// is one asm boundary, not one per limb.  Outputs are early-clobber: none may share a register with a later input.
__device__ __forceinline__ void kyb_dbl5(uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t& o3, uint32_t& o4,
                                         uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4) {
  asm("v_add_u32 %0, %5, %5\n\tv_add_u32 %1, %6, %6\n\tv_add_u32 %2, %7, %7\n\tv_add_u32 %3, %8, %8\n\tv_add_u32 %4, %9, %9"
      : "=&v"(o0), "=&v"(o1), "=&v"(o2), "=&v"(o3), "=&v"(o4) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4));
}
__device__ __forceinline__ void kyb_dbl3(uint32_t& o0, uint32_t& o1, uint32_t& o2, uint32_t a0, uint32_t a1, uint32_t a2) {
  asm("v_add_u32 %0, %3, %3\n\tv_add_u32 %1, %4, %4\n\tv_add_u32 %2, %5, %5" : "=&v"(o0), "=&v"(o1), "=&v"(o2) : "v"(a0), "v"(a1), "v"(a2));
}
__device__ __forceinline__ uint64_t kyb_col6(uint64_t acc, const uint32_t* A, const uint32_t* B) {
  asm(KYB_M1(1) KYB_M1(2) KYB_M1(3) KYB_M1(4) KYB_M1(5) KYB_M1(6)
      : [acc] "+v"(acc)
      : [a1] "v"(A[0]), [a2] "v"(A[1]), [a3] "v"(A[2]), [a4] "v"(A[3]), [a5] "v"(A[4]), [a6] "v"(A[5]),
        [b1] "v"(B[0]), [b2] "v"(B[1]), [b3] "v"(B[2]), [b4] "v"(B[3]), [b5] "v"(B[4]), [b6] "v"(B[5])
      : "vcc");
  return acc;
}
__device__ __forceinline__ uint64_t kyb_col5(uint64_t acc, const uint32_t* A, const uint32_t* B) {
  asm(KYB_M1(1) KYB_M1(2) KYB_M1(3) KYB_M1(4) KYB_M1(5)
      : [acc] "+v"(acc)
      : [a1] "v"(A[0]), [a2] "v"(A[1]), [a3] "v"(A[2]), [a4] "v"(A[3]), [a5] "v"(A[4]),
        [b1] "v"(B[0]), [b2] "v"(B[1]), [b3] "v"(B[2]), [b4] "v"(B[3]), [b5] "v"(B[4])
      : "vcc");
  return acc;
}
#endif

KYB_HD void fe_zero(fe& h) {
  KYB_UNROLL for (int i = 0; i < 10; ++i) h.v[i] = 0;
}
KYB_HD void fe_one(fe& h) {
  h.v[0] = 1;
  KYB_UNROLL for (int i = 1; i < 10; ++i) h.v[i] = 0;
}
KYB_HD void fe_copy(fe& h, const fe& f) {
  KYB_UNROLL for (int i = 0; i < 10; ++i) h.v[i] = f.v[i];
}
// 2p - g, limb-wise (g <= 1.99T)
KYB_HD void fe_neg_raw(fe& h, const fe& g) {
  const uint32_t p2[10] = KYB_FE_2P;
  KYB_UNROLL for (int i = 0; i < 10; ++i) h.v[i] = kyb_sub32(p2[i], g.v[i], "fe_sub");
}
// h = f + g, no carry: bound(h) = bound(f) + bound(g)
KYB_HD void fe_add(fe& h, const fe& f, const fe& g) {
  KYB_UNROLL for (int i = 0; i < 10; ++i) h.v[i] = kyb_add32(f.v[i], g.v[i], "fe_add");
}
// h = f - g computed as f + 2p - g; requires g <= 1.99T; bound(h) = bound(f) + 2T
KYB_HD void fe_sub(fe& h, const fe& f, const fe& g) {
  const uint32_t p2[10] = KYB_FE_2P;
  KYB_UNROLL for (int i = 0; i < 10; ++i) h.v[i] = kyb_add32(f.v[i], kyb_sub32(p2[i], g.v[i], "fe_sub"), "fe_sub+");
}
// The same two for the INNER LOOPS (ladder step, point doubling / addition): on the device two limbs share one 64-bit
// addition (no carry can cross a word: limb sums stay below 2^32), 5 v_lshl_add_u64 instead of 10 v_add_u32.  These kernels
// are bound by instruction COUNT, not by what an instruction does: the ladder step goes from 1275 to
// 1245 instructions and the same-box A/B gives -3.3 % (mul), -3.7 % (mul_base), -2.6 % (sign), -2.0 % (verify):
// profiles/r02/ab_add64w.log.  Register pairs cost allocation freedom, so code outside the loops (where the register
// peaks are) keeps the 32-bit forms, and so do the doublings inside fe_sq (measured: no gain there).
// -DKYB_NO_ADD64 turns these into the 32-bit forms as well.
KYB_HD void fe_addw(fe& h, const fe& f, const fe& g) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(KYB_NO_ADD64)
  KYB_UNROLL for (int j = 0; j < 5; ++j) {
    const uint64_t x = (uint64_t)f.v[2 * j] | ((uint64_t)f.v[2 * j + 1] << 32), y = (uint64_t)g.v[2 * j] | ((uint64_t)g.v[2 * j + 1] << 32);
    uint64_t s;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(s) : "v"(x), "v"(y));      // as asm: LLVM otherwise splits some of these back into 32-bit halves
    h.v[2 * j] = (uint32_t)s; h.v[2 * j + 1] = (uint32_t)(s >> 32);
  }
#else
  fe_add(h, f, g);
#endif
}
KYB_HD void fe_subw(fe& h, const fe& f, const fe& g) {
#if defined(__HIP_DEVICE_COMPILE__) && !defined(KYB_NO_ADD64)
  fe t;
  fe_neg_raw(t, g);
  KYB_UNROLL for (int i = 0; i < 10; ++i) asm("" : "+v"(t.v[i]));      // keep 2p - g as ten 32-bit subtractions (LLVM would widen them into borrow chains)
  fe_addw(h, f, t);
#else
  fe_sub(h, f, g);
#endif
}
// h = f - g computed as f + 4p - g; requires g <= 3.99T; bound(h) = bound(f) + 4T
KYB_HD void fe_sub4(fe& h, const fe& f, const fe& g) {
  const uint32_t p4[10] = KYB_FE_4P;
  KYB_UNROLL for (int i = 0; i < 10; ++i) h.v[i] = kyb_add32(f.v[i], kyb_sub32(p4[i], g.v[i], "fe_sub4"), "fe_sub4+");
}
// h = -f = 2p - f; requires f <= 1.99T; h <= 2T
KYB_HD void fe_neg(fe& h, const fe& f) {
  const uint32_t p2[10] = KYB_FE_2P;
  KYB_UNROLL for (int i = 0; i < 10; ++i) h.v[i] = kyb_sub32(p2[i], f.v[i], "fe_neg");
}
// h = c ? g : f  (lowered to v_cndmask_b32; c is 0/1)
KYB_HD void fe_select(fe& h, const fe& f, const fe& g, uint32_t c) {
  KYB_UNROLL for (int i = 0; i < 10; ++i) h.v[i] = c ? g.v[i] : f.v[i];
}
KYB_HD void fe_cmov(fe& h, const fe& g, uint32_t c) {
  KYB_UNROLL for (int i = 0; i < 10; ++i) h.v[i] = c ? g.v[i] : h.v[i];
}
KYB_HD void fe_cswap(fe& f, fe& g, uint32_t c) {
  KYB_UNROLL for (int i = 0; i < 10; ++i) {
    uint32_t a = f.v[i], b = g.v[i];
    f.v[i] = c ? b : a;
    g.v[i] = c ? a : b;
  }
}

// One parallel carry pass: any 32-bit limbs in, tight (<= 1.01T) out.  30 VALU ops.
KYB_HD void fe_reduce_weak(fe& h, const fe& f) {
  uint32_t c[10];
  KYB_UNROLL for (int i = 0; i < 10; ++i) c[i] = f.v[i] >> KYB_BITS(i);
  h.v[0] = (f.v[0] & KYB_MASK(0)) + 19u * c[9];  // c[9] <= 2^7
  KYB_UNROLL for (int i = 1; i < 10; ++i) h.v[i] = (f.v[i] & KYB_MASK(i)) + c[i - 1];
}

// Carry out of the top column folded back into limbs 0 and 1: h0 = (r0 + 19 c) mod 2^26, h1 = r1 + ((r0 + 19 c) >> 26).
// C32 = the caller guarantees c < 2^32 (column 9 stays below 2^57): then 19 c + r0 is ONE v_mad_u64_u32 on c's low word;
// otherwise c may have 34 bits and the high word is folded separately (v_lshrrev + v_mad_u32_u24 more per product).
template <bool C32>
KYB_HD void fe_fold(fe& h, const uint32_t r[10], uint64_t acc) {
  uint64_t t;
  if (C32) {
#if defined(KYB_HOST_TEST)
    if (acc >> 32) kyb_host_overflow("fold32: top carry needs more than 32 bits");
#endif
    t = kyb_mad((uint32_t)acc, 19u, (uint64_t)r[0]);
  } else {
    t = (uint64_t)r[0] + acc * 19u;       // acc < 2^39
  }
  h.v[0] = (uint32_t)t & KYB_MASK(0);
  h.v[1] = r[1] + (uint32_t)(t >> 26);
  KYB_UNROLL for (int i = 2; i < 10; ++i) h.v[i] = r[i];
}

// 19 * g, limbs 1..9 (limb 0 unused): the wrap-around multiples of a product's second operand.  Loop-invariant operands
// (the ladder's u(P)) compute it once.
KYB_HD void fe_x19(uint32_t g19[10], const fe& g) {
  g19[0] = 0;
  KYB_UNROLL for (int i = 1; i < 10; ++i) g19[i] = kyb_x19(g.v[i]);
}

// h = f*g with 19*g supplied.  Bounds as fe_mul.
template <bool C32>
KYB_HD void fe_mul_g19(fe& h, const fe& f, const fe& g, const uint32_t g19[10]) {
  uint32_t f2[10];
  f2[0] = f2[2] = f2[4] = f2[6] = f2[8] = 0;
#if defined(KYB_ASM_COLUMNS)
  kyb_dbl5(f2[1], f2[3], f2[5], f2[7], f2[9], f.v[1], f.v[3], f.v[5], f.v[7], f.v[9]);
#else
  KYB_UNROLL for (int i = 1; i < 10; i += 2) f2[i] = kyb_x2(f.v[i], "f2");
#endif
  uint64_t acc = 0;
  uint32_t r[10];
  KYB_UNROLL for (int k = 0; k < 10; ++k) {
#if defined(KYB_ASM_COLUMNS)
    uint32_t ca[10], cb[10];
    KYB_UNROLL for (int i = 0; i < 10; ++i) {
      const int j = (k - i + 10) % 10;
      ca[i] = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
      cb[i] = (i > k) ? g19[j] : g.v[j];
    }
    acc = (k == 0) ? kyb_col10z(ca, cb) : kyb_col10(acc, ca, cb);
#else
    KYB_UNROLL for (int i = 0; i < 10; ++i) {
      const int j = (k - i + 10) % 10;
      const bool wrap = i > k;
      const uint32_t fi = ((i & 1) && (j & 1)) ? f2[i] : f.v[i];
      const uint32_t gj = wrap ? g19[j] : g.v[j];
      acc = kyb_mad(fi, gj, acc);
    }
#endif
    r[k] = (uint32_t)acc & KYB_MASK(k);
    acc >>= KYB_BITS(k);
  }
  fe_fold<C32>(h, r, acc);
}

// h = f*g.  f <= 6T, g <= 3.3T.  100 v_mad_u64_u32 + ~45 VALU.  Output tight.
KYB_HD void fe_mul(fe& h, const fe& f, const fe& g) {
  uint32_t g19[10];
  fe_x19(g19, g);
  fe_mul_g19<false>(h, f, g, g19);
}
// The same for operands with bound(f) * bound(g) <= 6.3 (e.g. 3T x 2T, 6T x tight, tight x tight): column 9 then stays
// below 63 * 2^51 + 2^33 < 2^57, the top carry fits 32 bits and the fold is 3 instructions shorter.
KYB_HD void fe_mul_b6(fe& h, const fe& f, const fe& g) {
  uint32_t g19[10];
  fe_x19(g19, g);
  fe_mul_g19<true>(h, f, g, g19);
}

// h = f^2.  f <= 3.3T (C32: f <= 2.5T, column 9 <= 62.5 * 2^51).  55 v_mad_u64_u32.  Output tight.
// Operand multiples as the reference chooses them (fe.rs:544-688): 2 f_i for i <= 7, 38 f_j for the odd j >= 5, 19 f_j for
// j = 6, 8 — 8 doublings and 5 multiplications.  A wrapped term with odd j takes its whole factor (2 for the cross term or for
// odd-odd, times 19) from 38 f_j, and only the 76 of an odd-odd cross term also doubles f_i.
template <bool C32>
KYB_HD void fe_sq_t(fe& h, const fe& f) {
  uint32_t f2[10], fw[10];                    // fw[j]: the wrap-around multiple of f_j (38 f_j for odd j, 19 f_j for even j), j >= 5
  KYB_UNROLL for (int i = 0; i < 10; ++i) { f2[i] = 0; fw[i] = 0; }
  KYB_UNROLL for (int i = 5; i < 10; ++i) fw[i] = kyb_mul32(f.v[i], (i & 1) ? 38u : 19u, "sq fw");
#if defined(KYB_ASM_COLUMNS)
  kyb_dbl5(f2[0], f2[1], f2[2], f2[3], f2[4], f.v[0], f.v[1], f.v[2], f.v[3], f.v[4]);
  kyb_dbl3(f2[5], f2[6], f2[7], f.v[5], f.v[6], f.v[7]);
#else
  KYB_UNROLL for (int i = 0; i < 8; ++i) f2[i] = kyb_x2(f.v[i], "sq f2");
#endif
  uint64_t acc = 0;
  uint32_t r[10];
  KYB_UNROLL for (int k = 0; k < 10; ++k) {
#if defined(KYB_ASM_COLUMNS)
    uint32_t ca[6], cb[6];
    int nt = 0;
#endif
    KYB_UNROLL for (int i = 0; i < 10; ++i) {
      const int j = (k - i + 10) % 10;
      if (i > j) continue;                    // each unordered pair once
      const bool wrap = (i + j) >= 10;
      const bool cross = (i != j);
      const bool oo = (i & 1) && (j & 1);
      // multiplier = (cross?2:1) * (oo?2:1) * (wrap?19:1), split between the two operands
      uint32_t a, b;
      if (!wrap)      { a = cross ? f2[i] : f.v[i]; b = oo ? f2[j] : f.v[j]; }
      else if (j & 1) { a = (cross && oo) ? f2[i] : f.v[i]; b = fw[j]; }      // 38 f_j carries the 2 of a cross term with even i, or of f_j f_j
      else            { a = cross ? f2[i] : f.v[i]; b = fw[j]; }
#if defined(KYB_ASM_COLUMNS)
      ca[nt] = a; cb[nt] = b; ++nt;
#else
      acc = kyb_mad(a, b, acc);
#endif
    }
#if defined(KYB_ASM_COLUMNS)
    if (k == 0) acc = kyb_col6z(ca, cb);
    else if ((k & 1) == 0) acc = kyb_col6(acc, ca, cb); else acc = kyb_col5(acc, ca, cb);   // 6 pairs in even columns, 5 in odd ones
#endif
    r[k] = (uint32_t)acc & KYB_MASK(k);
    acc >>= KYB_BITS(k);
  }
  fe_fold<C32>(h, r, acc);
}
KYB_HD void fe_sq(fe& h, const fe& f) { fe_sq_t<false>(h, f); }
// f <= 2.5T (tight operands, sums of two tight ones)
KYB_HD void fe_sq_b2(fe& h, const fe& f) { fe_sq_t<true>(h, f); }

// h = f^(2^n), n >= 1 (rolled loop: one copy of the squaring body per call site)
KYB_HD void fe_sqn(fe& h, const fe& f, int n) {
  fe_sq_b2(h, f);
#if defined(__HIPCC__)
#pragma unroll 1
#endif
  for (int i = 1; i < n; ++i) fe_sq_b2(h, h);
}

// z^(2^250-1) and z^11, the shared prefix of both exponentiations (z <= 2.5T; everything after the first squaring is tight)
KYB_HD void fe_pow_250_1(fe& z250, fe& z11, const fe& z) {
  fe z2, z9, t, z5, z10, z20, z50, z100;
  fe_sq_b2(z2, z);               // 2
  fe_sqn(t, z2, 2);           // 8
  fe_mul_b6(z9, t, z);           // 9
  fe_mul_b6(z11, z9, z2);        // 11
  fe_sq_b2(t, z11);              // 22
  fe_mul_b6(z5, t, z9);          // 2^5-1
  fe_sqn(t, z5, 5);
  fe_mul_b6(z10, t, z5);         // 2^10-1
  fe_sqn(t, z10, 10);
  fe_mul_b6(z20, t, z10);        // 2^20-1
  fe_sqn(t, z20, 20);
  fe_mul_b6(t, t, z20);          // 2^40-1
  fe_sqn(t, t, 10);
  fe_mul_b6(z50, t, z10);        // 2^50-1
  fe_sqn(t, z50, 50);
  fe_mul_b6(z100, t, z50);       // 2^100-1
  fe_sqn(t, z100, 100);
  fe_mul_b6(t, t, z100);         // 2^200-1
  fe_sqn(t, t, 50);
  fe_mul_b6(z250, t, z50);       // 2^250-1
}
// h = z^(p-2) = z^(2^255-21): 254 squarings + 11 multiplications (same count as fe.rs:857-944)
KYB_HD void fe_invert(fe& h, const fe& z) {
  fe z250, z11;
  fe_pow_250_1(z250, z11, z);
  fe_sqn(z250, z250, 5);
  fe_mul_b6(h, z250, z11);
}
// h = z^((p-5)/8) = z^(2^252-3)  (fe.rs:946-1035)
KYB_HD void fe_pow22523(fe& h, const fe& z) {
  fe z250, z11;
  fe_pow_250_1(z250, z11, z);
  fe_sqn(z250, z250, 2);
  fe_mul_b6(h, z250, z);
}

// Canonical limbs (value in [0,p), limb_i < 2^bits_i).  Input: any limbs <= 2^31.
KYB_HD void fe_canon(fe& h, const fe& f) {
  uint32_t v[10];
  KYB_UNROLL for (int i = 0; i < 10; ++i) v[i] = f.v[i];
  // two sequential carry passes -> every limb within its mask, value < 2^255
  KYB_UNROLL for (int pass = 0; pass < 2; ++pass) {
    uint32_t c = 0;
    KYB_UNROLL for (int i = 0; i < 10; ++i) {
      v[i] += c;
      c = v[i] >> KYB_BITS(i);
      v[i] &= KYB_MASK(i);
    }
    v[0] += 19u * c;
  }
  // q = 1 iff value >= p  (value + 19 >= 2^255)
  uint32_t q = (v[0] + 19u) >> 26;
  KYB_UNROLL for (int i = 1; i < 10; ++i) q = (v[i] + q) >> KYB_BITS(i);
  uint32_t c = 19u * q;
  KYB_UNROLL for (int i = 0; i < 10; ++i) {
    v[i] += c;
    c = v[i] >> KYB_BITS(i);
    v[i] &= KYB_MASK(i);
  }
  KYB_UNROLL for (int i = 0; i < 10; ++i) h.v[i] = v[i];
}

// 8 little-endian 32-bit words of the canonical value (fe_to_bytes, fe.rs:147-238)
KYB_HD void fe_to_words(uint32_t w[8], const fe& f) {
  fe c;
  fe_canon(c, f);
  const uint32_t* v = c.v;
  // bit offsets 0,26,51,77,102,128,153,179,204,230
  w[0] = v[0] | (v[1] << 26);
  w[1] = (v[1] >> 6) | (v[2] << 19);
  w[2] = (v[2] >> 13) | (v[3] << 13);
  w[3] = (v[3] >> 19) | (v[4] << 6);
  w[4] = v[5] | (v[6] << 25);
  w[5] = (v[6] >> 7) | (v[7] << 19);
  w[6] = (v[7] >> 13) | (v[8] << 12);
  w[7] = (v[8] >> 20) | (v[9] << 6);
}

// limbs from 8 LE words; bit 255 ignored, values >= p accepted as-is (fe_from_bytes, fe.rs:67-122:
// "ignores bit 255 and does not reject y >= p").  Output limbs within masks (tight).
KYB_HD void fe_from_words(fe& h, const uint32_t w[8]) {
  h.v[0] = w[0] & 0x3ffffffu;
  h.v[1] = ((w[0] >> 26) | (w[1] << 6)) & 0x1ffffffu;
  h.v[2] = ((w[1] >> 19) | (w[2] << 13)) & 0x3ffffffu;
  h.v[3] = ((w[2] >> 13) | (w[3] << 19)) & 0x1ffffffu;
  h.v[4] = (w[3] >> 6) & 0x3ffffffu;
  h.v[5] = w[4] & 0x1ffffffu;
  h.v[6] = ((w[4] >> 25) | (w[5] << 7)) & 0x3ffffffu;
  h.v[7] = ((w[5] >> 19) | (w[6] << 13)) & 0x1ffffffu;
  h.v[8] = ((w[6] >> 12) | (w[7] << 20)) & 0x3ffffffu;
  h.v[9] = (w[7] >> 6) & 0x1ffffffu;
}

// fe_is_negative (fe.rs:240-245): low bit of the canonical value
KYB_HD uint32_t fe_is_negative(const fe& f) {
  fe c;
  fe_canon(c, f);
  return c.v[0] & 1u;
}
// fe_is_non_zero (fe.rs:247-257)
KYB_HD uint32_t fe_is_nonzero(const fe& f) {
  fe c;
  fe_canon(c, f);
  uint32_t o = 0;
  KYB_UNROLL for (int i = 0; i < 10; ++i) o |= c.v[i];
  return o != 0;
}

// Import a reference-layout field element: ten SIGNED int32 limbs, radix 2^25.5 (fe.rs:4-8).
// Accepts |limb_i| < 2^29 (the reference's own bound is 1.01*2^26): adds 16p limb-wise, then carries.
KYB_HD void fe_from_ref10(fe& h, const int32_t s[10]) {
  const uint32_t p4[10] = KYB_FE_4P;
  fe t;
  KYB_UNROLL for (int i = 0; i < 10; ++i) t.v[i] = (uint32_t)(s[i] + (int32_t)(4u * p4[i]));
  fe_reduce_weak(h, t);
}
// Export in the reference layout: the limbs fe_from_bytes (fe.rs:67-122) produces for the canonical
// value — signed, centred, |even| <= 2^25, |odd| <= 2^24.  Canonical NON-NEGATIVE limbs (up to 2^26)
// would break the reference's own bounds as soon as it adds two of them (19*(y+x) no longer fits
// an i32 in fe_mul, fe.rs:320-328), so the rounding-carry normal form is part of the ABI contract.
KYB_HD void fe_to_ref10(int32_t s[10], const fe& f) {
  uint32_t w[8];
  fe_to_words(w, f);
  // 32-bit modular arithmetic throughout (two's complement gives the signed limbs); only limbs 0 and 5
  // start as full 32-bit words and need a 64-bit intermediate for their rounding carry
  uint32_t h[10];
  h[0] = w[0];                                                       // bits   0..31
  h[1] = (w[1] & 0xffffffu) << 6;                                    // bits  32..55
  h[2] = (((w[1] >> 24) | (w[2] << 8)) & 0xffffffu) << 5;            // bits  56..79
  h[3] = (((w[2] >> 16) | (w[3] << 16)) & 0xffffffu) << 3;           // bits  80..103
  h[4] = (w[3] >> 8) << 2;                                           // bits 104..127
  h[5] = w[4];                                                       // bits 128..159
  h[6] = (w[5] & 0xffffffu) << 7;                                    // bits 160..183
  h[7] = (((w[5] >> 24) | (w[6] << 8)) & 0xffffffu) << 5;            // bits 184..207
  h[8] = (((w[6] >> 16) | (w[7] << 16)) & 0xffffffu) << 4;           // bits 208..231
  h[9] = ((w[7] >> 8) & 0x7fffffu) << 2;                             // bits 232..254
  // fe_from_bytes' carry order: 9, 1, 3, 5, 7, then 0, 2, 4, 6, 8 (fe.rs:79-108)
  uint32_t c;
  c = (h[9] + (1u << 24)) >> 25; const uint64_t h0w = (uint64_t)h[0] + 19u * c; h[9] -= c << 25;
  c = (h[1] + (1u << 24)) >> 25; h[2] += c; h[1] -= c << 25;
  c = (h[3] + (1u << 24)) >> 25; h[4] += c; h[3] -= c << 25;
  c = (uint32_t)(((uint64_t)h[5] + (1u << 24)) >> 25); h[6] += c; h[5] -= c << 25;
  c = (h[7] + (1u << 24)) >> 25; h[8] += c; h[7] -= c << 25;
  c = (uint32_t)((h0w + (1u << 25)) >> 26); h[1] += c; h[0] = (uint32_t)h0w - (c << 26);
  c = (h[2] + (1u << 25)) >> 26; h[3] += c; h[2] -= c << 26;
  c = (h[4] + (1u << 25)) >> 26; h[5] += c; h[4] -= c << 26;
  c = (h[6] + (1u << 25)) >> 26; h[7] += c; h[6] -= c << 26;
  c = (h[8] + (1u << 25)) >> 26; h[9] += c; h[8] -= c << 26;
  KYB_UNROLL for (int i = 0; i < 10; ++i) s[i] = (int32_t)h[i];
}

}  // namespace kyb
