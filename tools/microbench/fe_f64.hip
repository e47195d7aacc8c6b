// Field products on the FP64 pipe?  (DESIGN.md section 4 "The FP64 path".)
//
// The engine multiplies in ref10's radix 2^25.5: ten 32-bit limbs, 100 v_mad_u64_u32 per product, 55 per square, and about 38 more
// instructions per product around them (carry shifts and masks, 19x and 2x operand multiples, the fold).  BASELINE.json's north_star
// speaks of "51-bit-limb field arithmetic", and v_fma_f64 issues at the rate of v_mad_u64_u32, so a 5 x 51-bit product needs 25 limb
// products where the integer form needs 100.  This program builds that product, checks it bit for bit against the library's, and times
// both as dependent chains at 3 wavefronts per SIMD (the ladder's occupancy), as fe9.hip did for nine integer limbs.
//
// How a 102-bit limb product comes out of a 53-bit multiplier (Emmart, Zheng, Weems: "Faster modular exponentiation using double
// precision floating point arithmetic on the GPU", ARITH 2018): with C1 = 2^104 the sum a b + C1 lies in [2^104, 2^105) where doubles
// are 2^52 apart, so
//     hi = fma(a, b, C1)            = C1 + F 2^52,  F = a b / 2^52 rounded to an integer
//     lo = fma(a, b, C2 - hi)       = a b - F 2^52 + bias          (exact: the difference fits the mantissa)
// and the BIT PATTERNS of hi and lo are bits(C1) + F and bits(bias) + (a b - F 2^52): integers that add up column by column in 64-bit
// integer registers.  The FMAs must round TOWARD ZERO (F = floor, low part in [0, 2^52), bias 2^52 = C2 - C1): with round-to-nearest the
// low part is signed and its bias would have to be 1.5 x 2^52, which no difference of two multiples of 2^52 can supply.  The kernels set
// MODE.FP_ROUND for double precision once at their top (s_setreg); the host check uses fesetround.
// Per limb product that is 2 FMAs, 1 FP subtraction and 2 64-bit integer additions: FIVE instructions of the 4.4-cycle class on this
// chip (NVIDIA parts issue the integer additions down a separate pipe; CDNA4 has one VALU issue port per SIMD) — 125 for the 25
// products against the integer form's 100 multiply-adds — and then the columns still have to be recombined, folded (19 x 2^51 does not fit
// a double, so the wrap-around cannot be premultiplied into an operand), carried, and turned back into doubles.
//
// Also timed: the PRODUCT PART ONLY of a 6 x 43-bit all-floating-point form (where 152 g = 19 x 8 g still fits 53 bits and column sums
// of the low parts stay exact): high parts chained through the FMA addend, d = h_prev - h, lo = fma(a, b, d), low parts added: 23 FP
// instructions per column, 143 per product before any carry.  That kernel is a LOWER BOUND (its "carry" is two instructions per column
// and numerically meaningless), printed to show that the form loses before its normalisation is even written.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I kyber-rs_amd/csrc -o tools/microbench/fe_f64 tools/microbench/fe_f64.hip
//   tools/microbench/fe_f64 --host-only     (CPU: the same source against the ten-limb product, no GPU touched)
//   tools/microbench/fe_f64                 (GPU: bit-exactness on >= 10^6 operand pairs, then the timings)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <cfenv>
#include <vector>
#include <algorithm>
#include "fe25519.h"
using namespace kyb;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// ---- five limbs of 51 bits held as doubles ---------------------------------------------------------------------------------------------
struct fe5 { double v[5]; };            // limb i: a non-negative integer, weight 2^(51 i); "tight" = below 2^51 + 2^14

KYB_HD uint64_t dbits(double x) { return __builtin_bit_cast(uint64_t, x); }
KYB_HD double bdouble(uint64_t x) { return __builtin_bit_cast(double, x); }

KYB_HD void fe5_from_fe(fe5& h, const fe& f) {          // f within its masks
  KYB_UNROLL for (int i = 0; i < 5; ++i) h.v[i] = (double)((uint64_t)f.v[2 * i] | ((uint64_t)f.v[2 * i + 1] << 26));
}
KYB_HD void fe5_to_fe(fe& h, const fe5& f) {            // limbs below 2^52: the odd ten-limb gets up to 26 bits, which fe_canon accepts
  KYB_UNROLL for (int i = 0; i < 5; ++i) {
    const uint64_t l = (uint64_t)f.v[i];
    h.v[2 * i] = (uint32_t)l & 0x3ffffffu;
    h.v[2 * i + 1] = (uint32_t)(l >> 26);
  }
}

// h = f g (SQ: f^2, symmetric terms once on a doubled operand).  Operands: a_i b_j < 2^103.9 for every term (tight x tight, or a sum of
// two tight elements times a tight one).  Output tight.
template <bool SQ>
KYB_HD void fe5_mul_t(fe5& h, const fe5& f, const fe5& g) {
  const double C1 = 0x1p104, C2 = 0x1p104 + 0x1p52;
  const uint64_t B104 = 0x4670000000000000ull, B52 = 0x4330000000000000ull;       // bits(2^104), bits(2^52)
  double f2[5];
  if (SQ) { KYB_UNROLL for (int i = 0; i < 4; ++i) f2[i] = f.v[i] + f.v[i]; }
  uint64_t LO[10], HI[10];
  int n[10];
  KYB_UNROLL for (int c = 0; c < 10; ++c) { LO[c] = 0; HI[c] = 0; n[c] = 0; }
  KYB_UNROLL for (int i = 0; i < 5; ++i) {
    KYB_UNROLL for (int j = 0; j < 5; ++j) {
      if (SQ && j < i) continue;
      const double a = (SQ && i < j) ? f2[i] : f.v[i], b = g.v[j];
      const double hi = __builtin_fma(a, b, C1);
      const double lo = __builtin_fma(a, b, C2 - hi);
      HI[i + j] += dbits(hi);
      LO[i + j] += dbits(lo);
      ++n[i + j];
    }
  }
  // column c in units of 2^(51 c): its low parts, and twice the high parts of column c - 1 (2^52 = 2 x 2^51); below 2^57
  uint64_t T[10];
  KYB_UNROLL for (int c = 0; c < 10; ++c) {
    uint64_t t = LO[c] - (uint64_t)n[c] * B52;
    if (c) t += (HI[c - 1] - (uint64_t)n[c - 1] * B104) << 1;
    T[c] = t;
  }
  // 2^255 = 19: columns 5..9 come back times 19 (below 2^62); then one sequential carry
  const uint64_t M51 = (1ull << 51) - 1;
  uint64_t r[5];
  KYB_UNROLL for (int c = 0; c < 5; ++c) r[c] = T[c] + 19u * T[c + 5];
  KYB_UNROLL for (int c = 0; c < 4; ++c) { r[c + 1] += r[c] >> 51; r[c] &= M51; }
  const uint64_t top = r[4] >> 51;                      // < 2^11
  r[4] &= M51;
  r[0] += 19u * top;
  // integers below 2^52 back into doubles: exponent of 2^52 OR-ed on, 2^52 subtracted
  KYB_UNROLL for (int c = 0; c < 5; ++c) h.v[c] = bdouble(r[c] | B52) - 0x1p52;
}

// ---- six limbs of 43 bits, all floating point: the product part only (lower bound, see the header) ---------------------------------------
struct fe6 { double v[6]; };
KYB_HD void fe6_mul_lower_bound(fe6& h, const fe6& f, const fe6& g) {
  const double C = 0x1.8p98;                            // doubles 2^46 apart around it, room for signed sums below 2^97
  double g152[6];
  KYB_UNROLL for (int j = 1; j < 6; ++j) g152[j] = g.v[j] * 152.0;
  g152[0] = 0;
  KYB_UNROLL for (int c = 0; c < 6; ++c) {
    double hc = C, L = 0;
    KYB_UNROLL for (int i = 0; i < 6; ++i) {
      const int j = (c - i + 6) % 6;
      const double a = f.v[i], b = (i > c) ? g152[j] : g.v[j];
      const double hn = __builtin_fma(a, b, hc);        // hc + (a b truncated to a multiple of 2^46)
      const double d = hc - hn;
      const double lo = __builtin_fma(a, b, d);         // exact, 0 <= lo < 2^46
      L = (i == 0) ? lo : L + lo;
      hc = hn;
    }
    // NOT a carry: two instructions that keep the magnitudes of the next operand in range, so that the chain can be timed
    h.v[c] = __builtin_fma(hc - C, 0x1p-55, L * 0x1p-6);
  }
}

// ---- kernels: dependent chains, 2 operations per iteration (the harness of fe9.hip) ------------------------------------------------------
enum { V_MUL10, V_SQ10, V_MUL5, V_SQ5, V_MUL6_LB };
struct Stamp { unsigned long long cyc, rt; };
__device__ __forceinline__ unsigned long long memtime() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
__device__ __forceinline__ unsigned long long memrealtime() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
template <int V>
__global__ void __launch_bounds__(256, 3) k_chain(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, int iters, Stamp* __restrict__ stamps) {
  // MODE[3:2] = rounding of double (and half) precision: 3 = toward zero.  hwreg(HW_REG_MODE = 1, offset 2, size 2) = 1 | 2 << 6 | 1 << 11.
  // As inline asm: LLVM's SIModeRegister pass assumes that every double-precision instruction wants the DEFAULT rounding mode, sees a
  // __builtin_amdgcn_s_setreg of MODE, and puts the mode back in front of the first FP64 instruction (the first version of this file
  // did that and produced round-to-nearest results on the GPU while the CPU check of the same source passed).
  if (V == V_MUL5 || V == V_SQ5 || V == V_MUL6_LB) asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 2, 2), 3" ::: "memory");
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t wa[8], wb[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) { wa[k] = in[16 * i + k]; wb[k] = in[16 * i + 8 + k]; }
  fe a, b;
  fe_from_words(a, wa); fe_from_words(b, wb);
  const unsigned long long t0 = memtime(), r0 = memrealtime();
  if (V == V_MUL5 || V == V_SQ5) {
    fe5 x, y;
    fe5_from_fe(x, a); fe5_from_fe(y, b);
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
      if (V == V_MUL5) { fe5_mul_t<false>(x, x, y); fe5_mul_t<false>(y, y, x); }
      else { fe5_mul_t<true>(x, x, x); fe5_mul_t<true>(y, y, y); }
    }
    fe5_to_fe(a, x); fe5_to_fe(b, y);
  } else if (V == V_MUL6_LB) {
    fe6 x, y;
#pragma unroll
    for (int k = 0; k < 6; ++k) { x.v[k] = (double)(wa[k] & 0x3ffffffu) * 0x1p16; y.v[k] = (double)(wb[k] & 0x3ffffffu) * 0x1p16; }
#pragma unroll 1
    for (int it = 0; it < iters; ++it) { fe6_mul_lower_bound(x, x, y); fe6_mul_lower_bound(y, y, x); }
#pragma unroll
    for (int k = 0; k < 5; ++k) { a.v[2 * k] = (uint32_t)(int64_t)x.v[k] & 0x3ffffffu; b.v[2 * k] = (uint32_t)(int64_t)y.v[k] & 0x3ffffffu; }
  } else {
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
      if (V == V_MUL10) { fe_mul_b6(a, a, b); fe_mul_b6(b, b, a); }
      else { fe_sq_b2(a, a); fe_sq_b2(b, b); }
    }
  }
  asm volatile("" :: "v"(a.v[0]), "v"(b.v[0]));
  const unsigned long long t1 = memtime(), r1 = memrealtime();
  if (stamps && (threadIdx.x & 63) == 0) { stamps[i >> 6].cyc = t1 - t0; stamps[i >> 6].rt = r1 - r0; }
  fe_to_words(wa, a); fe_to_words(wb, b);
#pragma unroll
  for (int k = 0; k < 8; ++k) { out[16 * i + k] = wa[k]; out[16 * i + 8 + k] = wb[k]; }
}

static Stamp* g_stamps = nullptr;          // one per wavefront
static double g_last_mhz = 0;              // median in-kernel shader clock of the last timed launch (s_memtime / s_memrealtime, 100 MHz)
template <int V>
static double run(const uint32_t* d_in, uint32_t* d_out, int blocks, int iters, std::vector<uint32_t>* host = nullptr) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {
    CK(hipEventRecord(a));
    hipLaunchKernelGGL(k_chain<V>, dim3(blocks), dim3(256), 0, 0, d_in, d_out, iters, g_stamps);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    if (rep && ms < best) best = ms;
  }
  if (g_stamps) {
    std::vector<Stamp> st((size_t)blocks * 4);
    CK(hipMemcpy(st.data(), g_stamps, st.size() * sizeof(Stamp), hipMemcpyDeviceToHost));
    std::vector<double> mhz;
    for (auto& x : st) if (x.rt) mhz.push_back((double)x.cyc / (double)x.rt * 100.0);
    std::sort(mhz.begin(), mhz.end());
    g_last_mhz = mhz.empty() ? 0 : mhz[mhz.size() / 2];
  }
  if (host) CK(hipMemcpy(host->data(), d_out, host->size() * 4, hipMemcpyDeviceToHost));
  return best;
}

// operand words: random, with every 16th pair replaced by an edge pattern (limbs at their extremes, p - 1, 2^255 - 1, 0, 1, single bits)
static void fill_operands(std::vector<uint32_t>& in, unsigned seed) {
  srand(seed);
  for (auto& w : in) w = ((uint32_t)rand() << 16) ^ (uint32_t)rand();
  const size_t pairs = in.size() / 16;
  for (size_t p = 0; p < pairs; p += 16) {
    uint32_t* a = &in[16 * p];
    const unsigned kind = (unsigned)((p / 16) % 8);
    for (int half = 0; half < 2; ++half) {
      uint32_t* w = a + 8 * half;
      switch ((kind + 3 * half) % 8) {
        case 0: for (int k = 0; k < 8; ++k) w[k] = 0xffffffffu; w[7] = 0x7fffffffu; break;                   // 2^255 - 1: every limb at its maximum
        case 1: for (int k = 0; k < 8; ++k) w[k] = 0xffffffffu; w[0] = 0xffffffecu; w[7] = 0x7fffffffu; break; // p - 1
        case 2: for (int k = 0; k < 8; ++k) w[k] = 0; break;
        case 3: for (int k = 0; k < 8; ++k) w[k] = 0; w[0] = 1; break;
        case 4: for (int k = 0; k < 8; ++k) w[k] = 0; w[(p / 128) % 8] = 1u << ((p / 1024) % 32); break;       // one bit
        case 5: for (int k = 0; k < 8; ++k) w[k] = 0xffffffffu; w[7] = 0x7fffffffu; w[(p / 128) % 8] &= ~(1u << ((p / 1024) % 31)); break;
        case 6: for (int k = 0; k < 8; ++k) w[k] = (k & 1) ? 0xffffffffu : 0; w[7] &= 0x7fffffffu; break;
        default: break;                                                                                          // random
      }
    }
  }
}

static int host_check(size_t pairs) {
  fesetround(FE_TOWARDZERO);
  std::vector<uint32_t> in(pairs * 16);
  size_t bad = 0, total = 0;
  for (unsigned seed = 1; seed <= 2; ++seed) {
    fill_operands(in, seed);
    for (size_t p = 0; p < pairs; ++p) {
      uint32_t wa[8], wb[8], w10[8], w5[8];
      memcpy(wa, &in[16 * p], 32); memcpy(wb, &in[16 * p + 8], 32);
      fe a, b, m10, s10, t;
      fe_from_words(a, wa); fe_from_words(b, wb);
      fe_mul_b6(m10, a, b); fe_sq_b2(s10, a);
      fe5 x, y, m5, s5;
      fe5_from_fe(x, a); fe5_from_fe(y, b);
      fe5_mul_t<false>(m5, x, y); fe5_mul_t<true>(s5, x, x);
      fe_to_words(w10, m10); fe5_to_fe(t, m5); fe_to_words(w5, t);
      bad += memcmp(w10, w5, 32) != 0;
      fe_to_words(w10, s10); fe5_to_fe(t, s5); fe_to_words(w5, t);
      bad += memcmp(w10, w5, 32) != 0;
      // a lazy operand: (a + b) b, the shape of the ladder's (x + z)(x' - z')
      fe ab; fe_add(ab, a, b); fe_mul_b6(m10, ab, b);
      fe5 xy; for (int k = 0; k < 5; ++k) xy.v[k] = x.v[k] + y.v[k];
      fe5_mul_t<false>(m5, xy, y);
      fe_to_words(w10, m10); fe5_to_fe(t, m5); fe_to_words(w5, t);
      bad += memcmp(w10, w5, 32) != 0;
      total += 3;
    }
  }
  printf("host check (same source, CPU): %zu of %zu FP64 results differ from the ten-limb ones\n", bad, total);
  return bad ? 1 : 0;
}

int main(int argc, char** argv) {
  if (argc > 1 && !strcmp(argv[1], "--host-only")) return host_check(argc > 2 ? (size_t)atol(argv[2]) : 200000);
  if (host_check(20000)) return 1;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, blocks = cus * 3;                               // 3 workgroups of 4 wavefronts per CU = 3 per SIMD
  const size_t lanes = (size_t)blocks * 256;
  std::vector<uint32_t> in(lanes * 16), o10(lanes * 16), o5(lanes * 16);
  uint32_t *d_in, *d_out;
  CK(hipMalloc(&d_in, in.size() * 4)); CK(hipMalloc(&d_out, in.size() * 4));
  CK(hipMalloc(&g_stamps, (size_t)blocks * 4 * sizeof(Stamp)));
  size_t checked = 0, bad_total = 0;
  for (unsigned seed = 1; seed <= 3; ++seed) {
    fill_operands(in, seed);
    CK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice));
    for (int iters : {1, 7}) {
      run<V_MUL10>(d_in, d_out, blocks, iters, &o10); run<V_MUL5>(d_in, d_out, blocks, iters, &o5);
      size_t badm = 0, bads = 0;
      for (size_t i = 0; i < o10.size(); ++i) badm += o10[i] != o5[i];
      run<V_SQ10>(d_in, d_out, blocks, iters, &o10); run<V_SQ5>(d_in, d_out, blocks, iters, &o5);
      for (size_t i = 0; i < o10.size(); ++i) bads += o10[i] != o5[i];
      printf("check, seed %u, %d iterations: FP64 5x51 words differing from ten-limb: products %zu, squares %zu of %zu (%zu field elements each)\n",
             seed, iters, badm, bads, o10.size(), lanes * 2);
      checked += 4 * lanes * (size_t)iters; bad_total += badm + bads;
    }
  }
  printf("bit-exact field operations checked on the GPU: %zu, differing words: %zu\n", checked, bad_total);
  fill_operands(in, 7);
  CK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice));
  const int iters = 2000;
  const double ops = 2.0 * iters;
  // instructions per operation in the loop bodies: hipcc -S of this file, VALU instructions of each loop / 2
  const double m10 = run<V_MUL10>(d_in, d_out, blocks, iters), c_m10 = g_last_mhz, s10 = run<V_SQ10>(d_in, d_out, blocks, iters), c_s10 = g_last_mhz;
  const double m5 = run<V_MUL5>(d_in, d_out, blocks, iters), c_m5 = g_last_mhz, s5 = run<V_SQ5>(d_in, d_out, blocks, iters), c_s5 = g_last_mhz;
  const double m6 = run<V_MUL6_LB>(d_in, d_out, blocks, iters), c_m6 = g_last_mhz;
  printf("%s, %d CUs, 3 wavefronts per SIMD, %d dependent operations per lane; ns per operation of a lane (kernel time / operations), in-kernel shader clock, SIMD cycles per operation of a wavefront (ns x GHz / 3):\n", prop.name, cus, (int)ops);
  printf("  product   ten limbs (fe_mul_b6, the library's)               %8.2f   %5.0f MHz  %7.1f\n", m10 * 1e6 / ops, c_m10, m10 * 1e6 / ops * c_m10 * 1e-3 / 3);
  printf("  product   5 x 51 bits on v_fma_f64 (bit-exact)                %8.2f   %5.0f MHz  %7.1f   (x%.3f)\n", m5 * 1e6 / ops, c_m5, m5 * 1e6 / ops * c_m5 * 1e-3 / 3, m5 / m10);
  printf("  product   6 x 43 bits all-FP, product part only (lower bound) %8.2f   %5.0f MHz  %7.1f   (x%.3f)\n", m6 * 1e6 / ops, c_m6, m6 * 1e6 / ops * c_m6 * 1e-3 / 3, m6 / m10);
  printf("  square    ten limbs (fe_sq_b2, the library's)                 %8.2f   %5.0f MHz  %7.1f\n", s10 * 1e6 / ops, c_s10, s10 * 1e6 / ops * c_s10 * 1e-3 / 3);
  printf("  square    5 x 51 bits on v_fma_f64 (bit-exact)                %8.2f   %5.0f MHz  %7.1f   (x%.3f)\n", s5 * 1e6 / ops, c_s5, s5 * 1e6 / ops * c_s5 * 1e-3 / 3, s5 / s10);
  printf("  ladder step = 5 products + 4 squares: ten limbs %.2f, FP64 %.2f (x%.3f) — before the carry pass that FP64 limbs need after every addition of two lazy elements\n",
         (5 * m10 + 4 * s10) * 1e6 / ops, (5 * m5 + 4 * s5) * 1e6 / ops, (5 * m5 + 4 * s5) / (5 * m10 + 4 * s10));
  return bad_total ? 1 : 0;
}
