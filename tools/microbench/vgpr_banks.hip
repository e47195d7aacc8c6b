// Does v_mad_u64_u32's issue cost on gfx950 depend on which VGPR banks its operands sit in?
//
// The ladder kernel spends 5,130 SIMD cycles per step on 739 v_mad_u64_u32 + ~540 other VALU; removing 4 % of the
// estimated issue cycles of the OTHER instructions changed neither its cycle count (GRBM_GUI_ACTIVE) nor its time
// (profiles/r02/ab_field_microopts_same_box.log), so the multiply-adds cost more in the kernel than in the pure
// stream of tools/microbench/valu_rates.hip (4.55 cycles).  This program pins operands to explicit registers:
// bank = register index mod 4.  One JSON line per pattern at 2, 3, 4 and 8 waves per SIMD.
//
// build: hipcc --offload-arch=gfx950 -O3 -o vgpr_banks vgpr_banks.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

struct Stamp { unsigned long long cyc, rt; };
__device__ __forceinline__ unsigned long long memtime() { unsigned long long t; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }
__device__ __forceinline__ unsigned long long memrealtime() { unsigned long long t; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); return t; }

#define CLOB "v0","v1","v2","v3","v4","v5","v6","v7","v8","v9","v10","v11","v12","v13","v14","v15","v16","v17","v18","v19","v20","v21","v22","v23","v24","v25","v26","v27","v28","v29","v30","v31","v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","vcc"
#define INIT "v_mov_b32 v0, %0\n\tv_mov_b32 v1, %0\n\tv_mov_b32 v2, %0\n\tv_mov_b32 v3, %0\n\tv_mov_b32 v4, %0\n\tv_mov_b32 v5, %0\n\tv_mov_b32 v6, %0\n\tv_mov_b32 v7, %0\n\t" \
             "v_mov_b32 v8, %0\n\tv_mov_b32 v9, 0\n\tv_mov_b32 v10, %0\n\tv_mov_b32 v11, 0\n\tv_mov_b32 v12, %0\n\tv_mov_b32 v13, 0\n\tv_mov_b32 v14, %0\n\tv_mov_b32 v15, 0\n\t" \
             "v_mov_b32 v16, %0\n\tv_mov_b32 v17, 0\n\tv_mov_b32 v18, %0\n\tv_mov_b32 v19, 0\n\tv_mov_b32 v20, %0\n\tv_mov_b32 v21, 0\n\tv_mov_b32 v22, %0\n\tv_mov_b32 v23, 0\n\t" \
             "v_mov_b32 v24, %0\n\tv_mov_b32 v25, 0\n\tv_mov_b32 v26, %0\n\tv_mov_b32 v27, 0\n\tv_mov_b32 v28, %0\n\tv_mov_b32 v29, 0\n\tv_mov_b32 v30, %0\n\tv_mov_b32 v31, 0\n\t" \
             "v_mov_b32 v32, %0\n\tv_mov_b32 v33, 0\n\tv_mov_b32 v34, %0\n\tv_mov_b32 v35, 0\n\tv_mov_b32 v36, %0\n\tv_mov_b32 v37, 0\n\tv_mov_b32 v38, %0\n\tv_mov_b32 v39, 0\n\t" \
             "v_mov_b32 v40, %0\n\tv_mov_b32 v41, %0\n\tv_mov_b32 v42, %0\n\tv_mov_b32 v43, %0\n\tv_mov_b32 v44, %0\n\tv_mov_b32 v45, %0\n\tv_mov_b32 v46, %0\n\tv_mov_b32 v47, %0\n\tv_mov_b32 v48, %0\n\tv_mov_b32 v49, %0\n\t" \
             "v_mov_b32 v50, %0\n\tv_mov_b32 v51, %0\n\tv_mov_b32 v52, %0\n\tv_mov_b32 v53, %0\n\tv_mov_b32 v54, %0\n\tv_mov_b32 v55, %0\n\tv_mov_b32 v56, %0\n\tv_mov_b32 v57, %0\n\tv_mov_b32 v58, %0\n\tv_mov_b32 v59, %0\n\t"
// one mad: D, S0, S1, S2 as register strings
#define M(D, A, B, C) "v_mad_u64_u32 " D ", vcc, " A ", " B ", " C "\n\t"

// 8 accumulator pairs, all starting on bank 0: v[8:9] v[12:13] ... v[36:37]; sources chosen per pattern
#define BODY8(A, B) M("v[8:9]", A, B, "v[8:9]") M("v[12:13]", A, B, "v[12:13]") M("v[16:17]", A, B, "v[16:17]") M("v[20:21]", A, B, "v[20:21]") \
                    M("v[24:25]", A, B, "v[24:25]") M("v[28:29]", A, B, "v[28:29]") M("v[32:33]", A, B, "v[32:33]") M("v[36:37]", A, B, "v[36:37]")
// accumulator pairs starting on bank 2: v[10:11] v[14:15] ...
#define BODY8_B2(A, B) M("v[10:11]", A, B, "v[10:11]") M("v[14:15]", A, B, "v[14:15]") M("v[18:19]", A, B, "v[18:19]") M("v[22:23]", A, B, "v[22:23]") \
                       M("v[26:27]", A, B, "v[26:27]") M("v[30:31]", A, B, "v[30:31]") M("v[34:35]", A, B, "v[34:35]") M("v[38:39]", A, B, "v[38:39]")
// one dependent chain of 8 on ONE accumulator (what a product column is)
#define CHAIN8(ACC, A0, B0, A1, B1) M(ACC, A0, B0, ACC) M(ACC, A1, B1, ACC) M(ACC, A0, B1, ACC) M(ACC, A1, B0, ACC) M(ACC, A0, B0, ACC) M(ACC, A1, B1, ACC) M(ACC, A0, B1, ACC) M(ACC, A1, B0, ACC)

#define DEF(NAME, ASMBODY)                                                                         \
__global__ void NAME(int iters, unsigned seed, Stamp* stamps, unsigned* sink) {                   \
  unsigned a = (threadIdx.x * 2654435761u + seed) & 0x3ffffffu;                                    \
  unsigned long long t0 = memtime(), r0 = memrealtime();                                            \
  unsigned out;                                                                                     \
  asm volatile(INIT "s_mov_b32 s20, %2\n\t"                                                        \
               "1:\n\t" ASMBODY ASMBODY ASMBODY ASMBODY                                            \
               "s_sub_u32 s20, s20, 1\n\ts_cmp_lg_u32 s20, 0\n\ts_cbranch_scc1 1b\n\t"             \
               "v_xor_b32 %1, v8, v12\n\tv_xor_b32 %1, %1, v16\n\tv_xor_b32 %1, %1, v10\n\t"        \
               : "+v"(a), "=v"(out) : "s"(iters) : CLOB, "s20", "scc");                            \
  unsigned long long t1 = memtime(), r1 = memrealtime();                                            \
  if (out == 0x1234567u) sink[0] = out;                                                             \
  if ((threadIdx.x & 63) == 0) { int w = (blockIdx.x * blockDim.x + threadIdx.x) >> 6; stamps[w].cyc = t1 - t0; stamps[w].rt = r1 - r0; } \
}

DEF(k_banks_all_distinct, BODY8("v2", "v3"))          // S0 bank 2, S1 bank 3, S2/D banks 0,1
DEF(k_s0_hits_acc_lo, BODY8("v0", "v3"))              // S0 bank 0 = acc.lo's bank
DEF(k_s0_s1_hit_acc, BODY8("v0", "v1"))               // S0 bank 0, S1 bank 1 = acc.lo, acc.hi
DEF(k_s0_s1_same_bank, BODY8("v2", "v6"))             // S0, S1 both bank 2
DEF(k_all_bank0, BODY8("v0", "v4"))                   // S0, S1, acc.lo all bank 0
DEF(k_acc_on_bank2_srcs_01, BODY8_B2("v0", "v1"))     // accumulators on banks 2,3, sources on 0,1
DEF(k_chain_distinct, CHAIN8("v[8:9]", "v2", "v3", "v6", "v7") CHAIN8("v[12:13]", "v2", "v3", "v6", "v7"))        // dependent chains, 2 per group
DEF(k_chain_conflict, CHAIN8("v[8:9]", "v0", "v1", "v4", "v5") CHAIN8("v[12:13]", "v0", "v1", "v4", "v5"))

// --- operand reuse: does a multiply-add whose sources differ from the previous instruction's cost more? ---
// product column as the kernels execute it: one accumulator, ten DIFFERENT (S0, S1) pairs (f_i, g_{k-i})
#define COL10(ACC) M(ACC, "v40", "v59", ACC) M(ACC, "v41", "v58", ACC) M(ACC, "v42", "v57", ACC) M(ACC, "v43", "v56", ACC) M(ACC, "v44", "v55", ACC) \
                   M(ACC, "v45", "v54", ACC) M(ACC, "v46", "v53", ACC) M(ACC, "v47", "v52", ACC) M(ACC, "v48", "v51", ACC) M(ACC, "v49", "v50", ACC)
// the same ten sources every time (what valu_rates.hip measured): S0 = v40, S1 = v50
#define COL10_SAME(ACC) M(ACC, "v40", "v50", ACC) M(ACC, "v40", "v50", ACC) M(ACC, "v40", "v50", ACC) M(ACC, "v40", "v50", ACC) M(ACC, "v40", "v50", ACC) \
                        M(ACC, "v40", "v50", ACC) M(ACC, "v40", "v50", ACC) M(ACC, "v40", "v50", ACC) M(ACC, "v40", "v50", ACC) M(ACC, "v40", "v50", ACC)
// row-wise (operand scanning): S0 fixed over ten consecutive multiply-adds, ten different accumulators and S1
#define ROW10(F) M("v[8:9]", F, "v50", "v[8:9]") M("v[10:11]", F, "v51", "v[10:11]") M("v[12:13]", F, "v52", "v[12:13]") M("v[14:15]", F, "v53", "v[14:15]") M("v[16:17]", F, "v54", "v[16:17]") \
                M("v[18:19]", F, "v55", "v[18:19]") M("v[20:21]", F, "v56", "v[20:21]") M("v[22:23]", F, "v57", "v[22:23]") M("v[24:25]", F, "v58", "v[24:25]") M("v[26:27]", F, "v59", "v[26:27]")
// S0 and S1 both change, accumulators rotate (no dependence between neighbours)
#define DIAG10 M("v[8:9]", "v40", "v59", "v[8:9]") M("v[10:11]", "v41", "v58", "v[10:11]") M("v[12:13]", "v42", "v57", "v[12:13]") M("v[14:15]", "v43", "v56", "v[14:15]") M("v[16:17]", "v44", "v55", "v[16:17]") \
               M("v[18:19]", "v45", "v54", "v[18:19]") M("v[20:21]", "v46", "v53", "v[20:21]") M("v[22:23]", "v47", "v52", "v[22:23]") M("v[24:25]", "v48", "v51", "v[24:25]") M("v[26:27]", "v49", "v50", "v[26:27]")
DEF(k_col_distinct_sources, COL10("v[8:9]") COL10("v[12:13]"))
DEF(k_col_same_sources, COL10_SAME("v[8:9]") COL10_SAME("v[12:13]"))
DEF(k_row_fixed_s0, ROW10("v40") ROW10("v41"))
DEF(k_diag_all_change, DIAG10 DIAG10)
// the column followed by its mask and 64-bit shift, as in fe_mul
#define COLFULL(ACC, R) COL10(ACC) "v_and_b32 " R ", 0x3ffffff, v8\n\tv_lshrrev_b64 " ACC ", 26, " ACC "\n\t"
DEF(k_col_with_carry, COLFULL("v[8:9]", "v30") COLFULL("v[8:9]", "v31"))

typedef void (*kern_t)(int, unsigned, Stamp*, unsigned*);
struct Entry { const char* name; kern_t k; int mads_per_body; };

int main() {
  CK(hipSetDevice(0));
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  std::vector<Entry> es = {{"all operands on distinct banks (S0 b2, S1 b3, acc b0/b1)", k_banks_all_distinct, 8}, {"S0 on acc.lo's bank", k_s0_hits_acc_lo, 8},
                           {"S0, S1 on acc.lo / acc.hi banks", k_s0_s1_hit_acc, 8}, {"S0, S1 share a bank (not the acc's)", k_s0_s1_same_bank, 8},
                           {"S0, S1, acc.lo all on bank 0", k_all_bank0, 8}, {"acc on banks 2/3, S0 b0, S1 b1", k_acc_on_bank2_srcs_01, 8},
                           {"dependent chains of 8, distinct banks", k_chain_distinct, 16}, {"dependent chains of 8, sources on the acc's banks", k_chain_conflict, 16},
                           {"column: 10 dependent mads, 10 DIFFERENT source pairs", k_col_distinct_sources, 20}, {"column: 10 dependent mads, the SAME source pair", k_col_same_sources, 20},
                           {"row: S0 fixed over 10 mads, 10 accumulators", k_row_fixed_s0, 20}, {"10 independent mads, every operand changes", k_diag_all_change, 20},
                           {"column + v_and + v_lshrrev_b64 (cycles per INSTRUCTION)", k_col_with_carry, 24}};
  const char* only = getenv("VB_ONLY");
  const int iters = 2048;
  const int max_waves = cus * 4 * 8;
  Stamp* d_st; unsigned* d_sink;
  CK(hipMalloc(&d_st, sizeof(Stamp) * max_waves)); CK(hipMalloc(&d_sink, 64));
  std::vector<Stamp> h(max_waves);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (auto& e : es)
    for (int wps : {2, 3, 4, 8}) {
      if (only && !strstr(e.name, only)) continue;
      const int blocks = cus * wps, nw = blocks * 4;
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, 32, 1u, d_st, d_sink);
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.k, dim3(blocks), dim3(256), 0, 0, iters, 1u, d_st, d_sink);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      CK(hipMemcpy(h.data(), d_st, sizeof(Stamp) * nw, hipMemcpyDeviceToHost));
      std::vector<double> mhz(nw);
      for (int i = 0; i < nw; ++i) mhz[i] = h[i].rt ? (double)h[i].cyc / (double)h[i].rt * 100.0 : 0;
      std::sort(mhz.begin(), mhz.end());
      const double clk = mhz[nw / 2];
      const double instr = (double)iters * 4 * e.mads_per_body;
      printf("{\"pattern\": \"%s\", \"waves_per_simd\": %d, \"cyc_per_mad_per_simd\": %.3f, \"shader_mhz\": %.0f, \"ms\": %.4f, \"T_mads_per_s\": %.2f}\n",
             e.name, wps, ms * 1e-3 * clk * 1e6 / (instr * wps), clk, ms, instr * 64.0 * nw / (ms * 1e-3) / 1e12);
      fflush(stdout);
    }
  return 0;
}
