// Integer/VALU issue-rate microbenchmark for gfx950 (MI355X).
//
// Purpose: SURVEY.md §7 "Integer-multiply roofline is unpublished" — the roofline that bounds the
// Ed25519 kernels is the issue rate of v_mad_u64_u32 (32x32+64 -> 64) and of the plain 32-bit VALU
// ops around it.  This program measures, per instruction, the cycles one SIMD needs per
// wave-instruction at 1/2/4/8 waves per SIMD, using s_memtime (shader clock) inside the kernel
// and hipEvents outside.  Output: one JSON line per (instruction, occupancy).
//
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

constexpr int UNROLL = 4;     // asm groups per loop iteration
constexpr int CHAINS = 8;     // independent dependency chains per group

struct Stamp { unsigned long long cyc; unsigned long long rt; };

__device__ __forceinline__ unsigned long long memtime() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}
__device__ __forceinline__ unsigned long long memrealtime() {
  unsigned long long t;
  asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  return t;
}

// Every kernel: CHAINS independent accumulators, `iters` iterations of UNROLL*CHAINS instructions.
#define KERNEL_PROLOGUE(T)                                                     \
  T acc[CHAINS];                                                                \
  unsigned a = threadIdx.x * 2654435761u + 12345u + seed;                       \
  unsigned b = threadIdx.x * 40503u + 977u + seed;                              \
  for (int c = 0; c < CHAINS; ++c) acc[c] = (T)(a + c * 7919u);                 \
  unsigned long long t0 = memtime(); unsigned long long r0 = memrealtime();

#define KERNEL_EPILOGUE(T)                                                     \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                             \
  unsigned long long t1 = memtime(); unsigned long long r1 = memrealtime();     \
  T s = 0; for (int c = 0; c < CHAINS; ++c) s += acc[c];                         \
  if (s == (T)0x1234567) sink[0] = (unsigned)s;                                  \
  if ((threadIdx.x & 63) == 0) {                                                 \
    int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;                        \
    stamps[w].cyc = t1 - t0; stamps[w].rt = r1 - r0;                             \
  }

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

// ---- 64-bit accumulator kernels ----
#define DEF_K64(NAME, ASMSTR, ...)                                                             \
__global__ void NAME(int iters, unsigned seed, Stamp* stamps, unsigned* sink) {                 \
  KERNEL_PROLOGUE(unsigned long long)                                                           \
  for (int it = 0; it < iters; ++it) {                                                          \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                                        \
      _Pragma("unroll") for (int c = 0; c < CHAINS; ++c)                                        \
        asm volatile(ASMSTR : "+v"(acc[c]) : "v"(a), "v"(b) : __VA_ARGS__);                            \
    }                                                                                           \
  }                                                                                             \
  KERNEL_EPILOGUE(unsigned long long)                                                           \
}

// ---- 32-bit accumulator kernels ----
#define DEF_K32(NAME, ASMSTR, ...)                                                             \
__global__ void NAME(int iters, unsigned seed, Stamp* stamps, unsigned* sink) {                 \
  KERNEL_PROLOGUE(unsigned)                                                                     \
  for (int it = 0; it < iters; ++it) {                                                          \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                                        \
      _Pragma("unroll") for (int c = 0; c < CHAINS; ++c)                                        \
        asm volatile(ASMSTR : "+v"(acc[c]) : "v"(a), "v"(b) : __VA_ARGS__);                            \
    }                                                                                           \
  }                                                                                             \
  KERNEL_EPILOGUE(unsigned)                                                                     \
}

DEF_K64(k_mad_u64_u32, "v_mad_u64_u32 %0, vcc, %1, %2, %0", "vcc")
DEF_K64(k_mad_i64_i32, "v_mad_i64_i32 %0, vcc, %1, %2, %0", "vcc")
DEF_K64(k_mad_u64_u32_sgprcarry, "v_mad_u64_u32 %0, s[20:21], %1, %2, %0", "s20", "s21")
DEF_K64(k_lshl_add_u64, "v_lshl_add_u64 %0, %0, 1, %0", "memory")
DEF_K64(k_lshrrev_b64, "v_lshrrev_b64 %0, 3, %0", "memory")
DEF_K64(k_fma_f64, "v_fma_f64 %0, %0, %0, %0", "memory")
DEF_K64(k_mul_f64, "v_mul_f64 %0, %0, %0", "memory")
DEF_K64(k_add_f64, "v_add_f64 %0, %0, %0", "memory")
DEF_K64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %0, %0", "memory")
DEF_K64(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %0", "memory")

DEF_K32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1", "memory")
DEF_K32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1", "memory")
DEF_K32(k_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %2", "memory")
DEF_K32(k_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1", "memory")
DEF_K32(k_mul_hi_u32_u24, "v_mul_hi_u32_u24 %0, %0, %1", "memory")
DEF_K32(k_add_u32, "v_add_u32 %0, %0, %1", "memory")
DEF_K32(k_add_co_u32, "v_add_co_u32 %0, vcc, %0, %1", "vcc")
DEF_K32(k_addc_co_u32, "v_addc_co_u32 %0, vcc, %0, %1, vcc", "vcc")
DEF_K32(k_add3_u32, "v_add3_u32 %0, %0, %1, %2", "memory")
DEF_K32(k_lshl_add_u32, "v_lshl_add_u32 %0, %0, 1, %1", "memory")
DEF_K32(k_and_or_b32, "v_and_or_b32 %0, %0, %1, %2", "memory")
DEF_K32(k_alignbit_b32, "v_alignbit_b32 %0, %0, %1, 26", "memory")
DEF_K32(k_lshrrev_b32, "v_lshrrev_b32 %0, 3, %0", "memory")
DEF_K32(k_bfe_u32, "v_bfe_u32 %0, %0, 3, 26", "memory")
DEF_K32(k_cndmask_b32, "v_cndmask_b32 %0, %0, %1, vcc", "memory")
DEF_K32(k_xor_b32, "v_xor_b32 %0, %0, %1", "memory")
DEF_K32(k_fma_f32, "v_fma_f32 %0, %0, %0, %0", "memory")
DEF_K32(k_dot4_u32_u8, "v_dot4_u32_u8 %0, %0, %1, %0", "memory")
DEF_K32(k_pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1", "memory")
DEF_K32(k_pk_mad_u16, "v_pk_mad_u16 %0, %0, %1, %0", "memory")
DEF_K32(k_mad_u16, "v_mad_u16 %0, %0, %1, %0", "memory")
DEF_K32(k_mov_b32, "v_mov_b32 %0, %1", "memory")
DEF_K32(k_bpermute, "ds_bpermute_b32 %0, %1, %0\n\ts_waitcnt lgkmcnt(0)", "memory")
DEF_K32(k_accvgpr_rw, "v_accvgpr_write_b32 a0, %0\n\ts_nop 1\n\tv_accvgpr_read_b32 %0, a0", "a0")
DEF_K32(k_cndmask_e64_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]", "memory")
DEF_K32(k_cmp_cndmask, "v_cmp_eq_u32 vcc, %1, %2\n\tv_cndmask_b32 %0, %0, %1, vcc", "vcc")
DEF_K32(k_cmp_e64_cndmask, "v_cmp_eq_u32_e64 s[20:21], %1, %2\n\tv_cndmask_b32_e64 %0, %0, %1, s[20:21]", "s20")
DEF_K32(k_and_b32, "v_and_b32 %0, %0, %1", "memory")
DEF_K32(k_or_b32, "v_or_b32 %0, %0, %1", "memory")
DEF_K32(k_bfi_b32, "v_bfi_b32 %0, %1, %2, %0", "memory")
DEF_K32(k_sub_u32, "v_sub_u32 %0, %0, %1", "memory")
DEF_K32(k_lshlrev_b32, "v_lshlrev_b32 %0, 1, %0", "memory")
DEF_K32(k_bpermute_nowait, "ds_bpermute_b32 %0, %1, %0", "memory")
DEF_K32(k_mov_dpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,2,3,0] row_mask:0xf bank_mask:0xf", "memory")

// mixed stream approximating a field-multiply inner loop: 10 mads then 3 simple ops
__global__ void k_mix_femul(int iters, unsigned seed, Stamp* stamps, unsigned* sink) {
  KERNEL_PROLOGUE(unsigned long long)
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c)
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        unsigned lo = (unsigned)acc[c], hi = (unsigned)(acc[c] >> 32);
        asm volatile("v_alignbit_b32 %0, %1, %0, 26" : "+v"(lo) : "v"(hi));
        asm volatile("v_lshrrev_b32 %0, 26, %0" : "+v"(hi));
        acc[c] = ((unsigned long long)hi << 32) | lo;
      }
    }
  }
  KERNEL_EPILOGUE(unsigned long long)
}

// the same multiply-add on operands shaped like field limbs (26 significant bits): the clock the chip holds is power
// limited, so the data matters (MI355X_MICROARCH.md, DVFS give-back items 1 and 7)
#define DEF_K64_LIMB(NAME, ASMSTR)                                                                \
__global__ void NAME(int iters, unsigned seed, Stamp* stamps, unsigned* sink) {                 \
  KERNEL_PROLOGUE(unsigned long long)                                                           \
  a &= 0x3ffffffu; b &= 0x3ffffffu;                                                             \
  for (int c = 0; c < CHAINS; ++c) acc[c] &= 0xffffffffffffull;                                 \
  for (int it = 0; it < iters; ++it) {                                                          \
    _Pragma("unroll") for (int u = 0; u < UNROLL; ++u) {                                        \
      _Pragma("unroll") for (int c = 0; c < CHAINS; ++c)                                        \
        asm volatile(ASMSTR : "+v"(acc[c]) : "v"(a), "v"(b) : "vcc");                           \
    }                                                                                           \
    _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) acc[c] &= 0xffffffffffffull;             \
  }                                                                                             \
  KERNEL_EPILOGUE(unsigned long long)                                                           \
}
DEF_K64_LIMB(k_mad_u64_u32_limbs, "v_mad_u64_u32 %0, vcc, %1, %2, %0")
DEF_K64_LIMB(k_mad_i64_i32_limbs, "v_mad_i64_i32 %0, vcc, %1, %2, %0")

// one field-multiplication column as the kernels execute it: 10 chained mads, one mask, one 64-bit shift
__global__ void k_mix_column(int iters, unsigned seed, Stamp* stamps, unsigned* sink) {
  KERNEL_PROLOGUE(unsigned long long)
  a &= 0x3ffffffu; b &= 0x3ffffffu;
  unsigned r = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                     "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                     "v_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\tv_mad_u64_u32 %0, vcc, %1, %2, %0\n\t"
                     "v_mad_u64_u32 %0, vcc, %1, %2, %0"
                     : "+v"(acc[c]), "+v"(a), "+v"(b) :: "vcc");
        r = (unsigned)acc[c] & 0x3ffffffu;
        acc[c] >>= 26;
        asm volatile("" : "+v"(r), "+v"(acc[c]));
        b ^= r & 1u;
      }
    }
  }
  acc[0] += r;
  KERNEL_EPILOGUE(unsigned long long)
}

// LDS uniform-address (broadcast) b128 read throughput
__global__ void k_lds_bcast_b128(int iters, unsigned seed, Stamp* stamps, unsigned* sink) {
  __shared__ uint4 tbl[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) tbl[i] = make_uint4(i + seed, i * 3, i * 5, i * 7);
  __syncthreads();
  KERNEL_PROLOGUE(unsigned)
  unsigned idx = seed & 7;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
#pragma unroll
      for (int c = 0; c < CHAINS; ++c) {
        uint4 v = tbl[(idx + c * 16 + u * 128 + (it & 1)) & 1023];
        acc[c] += v.x ^ v.y ^ v.z ^ v.w;
      }
    }
  }
  KERNEL_EPILOGUE(unsigned)
}

typedef void (*kern_t)(int, unsigned, Stamp*, unsigned*);
struct Entry { const char* name; kern_t k; int instr_per_group; };

int main(int argc, char** argv) {
  int dev = 0; CK(hipSetDevice(dev));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, dev));
  int cus = prop.multiProcessorCount;
  printf("{\"device\": \"%s\", \"arch\": \"%s\", \"cus\": %d, \"clock_khz\": %d}\n", prop.name, prop.gcnArchName, cus, prop.clockRate);

  std::vector<Entry> es = {
    {"v_mad_u64_u32", k_mad_u64_u32, 1}, {"v_mad_i64_i32", k_mad_i64_i32, 1},
    {"v_mul_lo_u32", k_mul_lo_u32, 1}, {"v_mul_hi_u32", k_mul_hi_u32, 1},
    {"v_mad_u32_u24", k_mad_u32_u24, 1}, {"v_mul_u32_u24", k_mul_u32_u24, 1}, {"v_mul_hi_u32_u24", k_mul_hi_u32_u24, 1},
    {"v_add_u32", k_add_u32, 1}, {"v_add_co_u32", k_add_co_u32, 1}, {"v_addc_co_u32", k_addc_co_u32, 1},
    {"v_add3_u32", k_add3_u32, 1}, {"v_lshl_add_u32", k_lshl_add_u32, 1}, {"v_and_or_b32", k_and_or_b32, 1},
    {"v_lshl_add_u64", k_lshl_add_u64, 1}, {"v_lshrrev_b64", k_lshrrev_b64, 1},
    {"v_alignbit_b32", k_alignbit_b32, 1}, {"v_lshrrev_b32", k_lshrrev_b32, 1}, {"v_bfe_u32", k_bfe_u32, 1},
    {"v_cndmask_b32", k_cndmask_b32, 1}, {"v_xor_b32", k_xor_b32, 1}, {"v_mov_b32", k_mov_b32, 1},
    {"v_fma_f32", k_fma_f32, 1}, {"v_pk_fma_f32", k_pk_fma_f32, 1}, {"v_pk_mul_f32", k_pk_mul_f32, 1},
    {"v_fma_f64", k_fma_f64, 1}, {"v_mul_f64", k_mul_f64, 1}, {"v_add_f64", k_add_f64, 1},
    {"v_dot4_u32_u8", k_dot4_u32_u8, 1}, {"v_pk_mul_lo_u16", k_pk_mul_lo_u16, 1}, {"v_pk_mad_u16", k_pk_mad_u16, 1}, {"v_mad_u16", k_mad_u16, 1},
    {"ds_bpermute_b32", k_bpermute, 1}, {"v_accvgpr_write+read", k_accvgpr_rw, 2}, {"v_mov_b32_dpp", k_mov_dpp, 1},
    {"v_cndmask_b32_e64(sgpr)", k_cndmask_e64_sgpr, 1}, {"v_cmp+v_cndmask(vcc)", k_cmp_cndmask, 2}, {"v_cmp_e64+v_cndmask_e64", k_cmp_e64_cndmask, 2},
    {"v_and_b32", k_and_b32, 1}, {"v_or_b32", k_or_b32, 1}, {"v_bfi_b32", k_bfi_b32, 1}, {"v_sub_u32", k_sub_u32, 1}, {"v_lshlrev_b32", k_lshlrev_b32, 1},
    {"ds_bpermute_b32(nowait)", k_bpermute_nowait, 1},
    {"v_mad_u64_u32(sgpr carry-out)", k_mad_u64_u32_sgprcarry, 1}, {"v_mad_u64_u32(26-bit operands)", k_mad_u64_u32_limbs, 1},
    {"v_mad_i64_i32(26-bit operands)", k_mad_i64_i32_limbs, 1}, {"mix_column(10mad+and+lshr64)", k_mix_column, 12},
    {"mix_femul(8mad+4simple)", k_mix_femul, 1}, {"lds_bcast_b128(+xor3+add)", k_lds_bcast_b128, 1},
  };
  const char* only = argc > 1 ? argv[1] : nullptr;

  int iters = 4096;
  int max_waves = cus * 4 * 8;
  Stamp* d_st; unsigned* d_sink;
  CK(hipMalloc(&d_st, sizeof(Stamp) * max_waves)); CK(hipMalloc(&d_sink, 64));
  std::vector<Stamp> h_st(max_waves);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));

  for (auto& e : es) {
    if (only && !strstr(e.name, only)) continue;
    for (int wps : {1, 2, 4, 8}) {       // waves per SIMD
      int threads = 256;                  // 4 waves per block -> one per SIMD
      int blocks = cus * wps;             // wps blocks per CU
      int nw = blocks * 4;
      // warm
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, 64, 1u, d_st, d_sink);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(threads), 0, 0, iters, 1u, d_st, d_sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(h_st.data(), d_st, sizeof(Stamp) * nw, hipMemcpyDeviceToHost));
      std::vector<double> cyc(nw), mhz(nw);
      for (int i = 0; i < nw; ++i) { cyc[i] = (double)h_st[i].cyc; mhz[i] = h_st[i].rt ? (double)h_st[i].cyc / (double)h_st[i].rt * 100.0 : 0; }
      std::sort(cyc.begin(), cyc.end()); std::sort(mhz.begin(), mhz.end());
      double med_cyc = cyc[nw / 2], med_mhz = mhz[nw / 2];
      double instr_per_wave = (double)iters * UNROLL * CHAINS * e.instr_per_group;
      if (!strcmp(e.name, "mix_femul(8mad+4simple)")) instr_per_wave = (double)iters * UNROLL * (CHAINS + 4);
      // Cycles one SIMD spends per wave-instruction.  The waves of a SIMD are served oldest first, so they do not finish
      // together and the MEDIAN wave's lifetime understates the time the SIMD was busy (round 1 divided the median by the
      // wave count and got figures that disagreed with the chip-wide rate at 4 and 8 waves).  The self-consistent figure is
      // the kernel's wall time in shader cycles over the wave-instructions one SIMD issued:
      //   cyc = ms x shader clock / (instructions per wave x waves per SIMD),  and  chip rate = SIMDs x 64 x clock / cyc.
      double max_cyc = cyc[nw - 1];
      double cyc_wall = ms * 1e-3 * med_mhz * 1e6 / (instr_per_wave * wps);
      double total_lane_ops = instr_per_wave * 64.0 * nw;
      double gops = total_lane_ops / (ms * 1e-3) / 1e9;
      printf("{\"instr\": \"%s\", \"waves_per_simd\": %d, \"cyc_per_waveinstr_per_simd\": %.3f, \"cyc_from_longest_wave\": %.3f, \"median_wave_cycles\": %.0f, \"longest_wave_cycles\": %.0f, \"shader_mhz\": %.0f, \"ms\": %.4f, \"chip_Glaneops_s\": %.1f}\n",
             e.name, wps, cyc_wall, max_cyc / (instr_per_wave * wps), med_cyc, max_cyc, med_mhz, ms, gops);
      fflush(stdout);
    }
  }
  return 0;
}
