// Host-callable launchers of every kernel group.  The library is built from seven translation units that hipcc
// compiles in parallel (csrc/Makefile); the CROSS-CHECK build (CROSSCHECK=1, -DKYB_CROSSCHECK: test infrastructure) adds two:
//
//   kernels_base.hip      fixed base: table builders, table checksum, k_mul_base64 (the fixed-base kernel)
//   kernels_ladder.hip    variable base: k_decode_or_identity, k_decode_to_proj, k_mont_prep, k_mul_ladder, k_pair_sum, k_ext_to_proj
//   [kernels_base_alt.hip  cross-check build only: radix-32 / radix-16 fixed-base kernels and the fused k_sign]
//   [kernels_window.hip    cross-check build only: windowed variable-base k_mul (mul.algo=0)]
//   kernels_verify.hip    SHA-512 users: k_verify_prep / _r / _final, k_sign_hash, k_eddsa_prep
//   kernels_misc.hip      k_finish, k_encode_batched, k_add, k_equal, k_encode, k_decode, k_poly_eval, k_poly_eval_part
//   kernels_msm.hip       linear combinations over shared points with PUBLIC scalars: k_msm_tables, k_msm_accumulate (+ k_msm_bases_coop in kernels_coop.hip)
//   kernels_coop.hip      small batches: one item per wavefront (or two / three wavefronts per item), lane-cooperative field arithmetic:
//                         k_mul_coop, k_mul_enc_coop, k_mul_base_coop, k_decode_coop, k_finish_coop, k_verify_coop, k_verify_prep(_r)_coop, k_poly_eval_coop
//   engine.hip            contexts, per-stream scratch, launch sequences, host-pointer pipeline, multi-device groups, C ABI
//
// A kernel is defined in exactly one unit; the engine reaches it through the plain C++ function declared here
// (arguments = the kernel's own, plus stream and launch geometry), so no template or __global__ symbol crosses a
// unit boundary.  Every launcher returns hipGetLastError() of its launch.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace kyb {

constexpr int KYB_BLOCK = 256;      // threads per workgroup of every one-item-per-lane kernel
constexpr int KYB_BLOCK32 = 1024;   // radix-32 fixed-base kernel
constexpr int FINISH_K = 8;                    // items per shared field inversion (k_finish, k_mont_prep, k_encode_batched)

// The short kernels either side of the ladder (k_mont_prep, k_finish: one wavefront per 512 items, a 265-multiplication inversion chain
// each) raise their wavefronts' issue priority.  Alone on the chip it changes nothing.  In the pipelined host-pointer path they share
// SIMDs with the ladder of the neighbouring chunk: the arbiter serves the OLDEST wavefront first, so a young short-kernel wavefront
// starves behind two resident ladder wavefronts for a whole ladder lifetime (1.9 ms instead of 0.1) while its 180-230 registers keep
// a third ladder wavefront out — the SIMD runs at two thirds (tools/host_pipeline_trace.py; -DKYB_NO_SETPRIO for the A/B).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(KYB_NO_SETPRIO)
#define KYB_SHORT_KERNEL_PRIORITY() __builtin_amdgcn_s_setprio(3)
#else
#define KYB_SHORT_KERNEL_PRIORITY() ((void)0)
#endif

namespace launch {

// ---- kernels_base.hip / kernels_base_alt.hip ----
hipError_t build_tables(uint32_t* table, hipStream_t st);                       // radix-16, -32 and -64 images, then the embedded checksum
hipError_t table_checksum(const uint32_t* table, uint64_t* out_dev, hipStream_t st);   // out_dev[0] = checksum of the image (embed slot excluded)
hipError_t build_coop_table(const uint32_t* image64, uint32_t* table_coop, hipStream_t st);      // entry-major copy of the radix-64 table for kernels_coop.hip
hipError_t mul_base64(bool split, int block, int grid, hipStream_t st, const uint8_t* sc, const uint8_t* sc_b, size_t n_a, size_t n,
                      uint8_t* oenc, int32_t* oext, const uint4* img64, uint4* proj, size_t stride, size_t offset);
// mid-size batches: four wavefronts per 64 items, a quarter of the windows each; the sums land in records [offset, offset + n), three partial points per item pass
// through records [parts_offset, parts_offset + 3 n)
hipError_t mul_base64_quarters(int grid, hipStream_t st, const uint8_t* sc, const uint8_t* sc_b, size_t n_a, size_t n, const uint4* img64, uint4* proj, size_t stride,
                               size_t offset, size_t parts_offset);
hipError_t mul_base32(bool split, int grid, hipStream_t st, const uint8_t* sc, size_t n, uint8_t* oenc, int32_t* oext, const uint4* img32,
                      uint4* proj, size_t stride, size_t offset);
hipError_t mul_base16(int mode, int block, bool split, int grid, hipStream_t st, const uint8_t* sc, size_t n, uint8_t* oenc, int32_t* oext,
                      const uint4* img, uint4* proj, size_t stride, size_t offset);
hipError_t sign_fused(int mode, int grid, hipStream_t st, const uint8_t* x, const uint8_t* k, const uint8_t* msgs, const uint32_t* off, size_t n,
                      uint8_t* sig, const uint4* img);

// ---- kernels_ladder.hip ----
hipError_t decode_or_identity(hipStream_t st, const uint8_t* enc, size_t n, int32_t* out_ext, uint8_t* ok);
// rows != 0: encoding i = (r, c) of a rows x cols matrix lands in record c * rows + r
hipError_t decode_to_proj(hipStream_t st, const uint8_t* enc, size_t n, uint4* proj, size_t stride, uint8_t* ok, size_t rows, size_t cols);
// scalars / top_or != nullptr (one scalar per point): the OR of the scalars' top four bits is collected into *top_or on the way
hipError_t mont_prep(hipStream_t st, const int32_t* pext, size_t n, uint4* proj, size_t stride, const uint8_t* scalars = nullptr, uint32_t* top_or = nullptr);
// top_or != nullptr: the word mont_prep collected into — the kernel takes skip_bits = 4 when it is 0, else 0; zero_next: cleared for the next call
hipError_t mul_ladder(int waves, hipStream_t st, const uint8_t* sc, size_t n, uint4* proj, size_t stride, size_t img_offset, size_t img_mod, int skip_bits,
                      const uint32_t* top_or = nullptr, uint32_t* zero_next = nullptr);
hipError_t mul_ladder_pair(hipStream_t st, const uint8_t* sc, size_t n, const int32_t* pext, size_t pts_mod, uint4* proj, size_t stride, int skip_bits);
// the same with FOUR lanes per item (ge_ladder_quad.h): three products deep per step
hipError_t mul_ladder_quad(hipStream_t st, const uint8_t* sc, size_t n, const int32_t* pext, size_t pts_mod, uint4* proj, size_t stride, int skip_bits);
// the same from wire encodings: the ladder on (1 + y : 1 - y) leaves its x-only state (160 bytes per item) while the decode runs elsewhere;
// ladder_recover turns state + decoded point into the projective result (ge_ladder_pair.h)
// k_mul_ladder_pair with the R half of a verification (k_verify_prep_r: flags_r, record r_offset + i) as further workgroups of the launch
hipError_t mul_ladder_pair_r(hipStream_t st, const uint8_t* sc, size_t n, const int32_t* pext, uint4* proj, size_t stride, int skip_bits, const uint8_t* sigs, uint8_t* flags_r, size_t r_offset,
                             int lanes = 2);      // lanes: 2, or 4 lanes per item (ge_ladder_quad.h)
hipError_t mul_ladder_pair_y(hipStream_t st, const uint8_t* sc, size_t n, const uint8_t* penc, uint4* state, int skip_bits, int lanes = 2);
// ladder and decode as one launch: the first workgroups walk the ladder, the ones behind them decode (out_ext / ok as decode_or_identity)
hipError_t mul_ladder_pair_y_dec(hipStream_t st, const uint8_t* sc, size_t n, const uint8_t* penc, uint4* state, int skip_bits, int32_t* out_ext, uint8_t* ok, int lanes = 2);
hipError_t ladder_recover(hipStream_t st, const uint8_t* sc, size_t n, const int32_t* pext, const uint4* state, uint4* proj, size_t stride,
                          uint8_t* flags = nullptr, const uint8_t* dec_ok = nullptr);       // flags != nullptr: flags[i] |= dec_ok[i] << 2 (verification)
// verification, the A half without the decode: flags (bits 0, 1, 3), h, s from the bytes alone (kernels_verify.hip)
hipError_t verify_hash(hipStream_t st, const uint8_t* pubs, const uint8_t* sigs, const uint8_t* msgs, const uint32_t* off, size_t n, uint8_t* flags_a, uint8_t* hbuf, uint8_t* sbuf);
// ladder.y_only = 2, verification (kernels_ladder.hip): hash + ladder on A's y, decode of A, decode of R as workgroup roles of one launch; the join checks the equation
hipError_t verify_ladder_y(hipStream_t st, const uint8_t* pubs, const uint8_t* sigs, const uint8_t* msgs, const uint32_t* off, size_t n, uint8_t* flags_a, uint8_t* flags_r,
                           uint8_t* a_ok, uint8_t* hbuf, uint4* state, int32_t* a_ext, uint4* proj, size_t stride, int lanes = 2);
hipError_t sig_scalars(hipStream_t st, const uint8_t* sigs, size_t n, uint8_t* sbuf);
hipError_t pair_sum(hipStream_t st, uint4* proj, size_t stride, size_t m, size_t gstride, size_t len, size_t half);
hipError_t ext_to_proj(hipStream_t st, const int32_t* pext, size_t n, uint4* proj, size_t stride, size_t rows = 0, size_t cols = 0);      // rows != 0: transposed, as decode_to_proj

// Completion signal of a small host-pointer call (engine.hip, HostCall): the LAST kernel of the call's launch sequence counts
// its finished items (threads or wavefronts, `total` of them) in *counter; the one that completes the count resets the counter and
// stores `seq` into *flag, a word of coherent page-locked host memory the calling thread is spinning on — ~6 us sooner than
// hipStreamSynchronize notices the end of the kernel (tools/microbench/launch_latency.hip).  flag == nullptr: no signal.
struct DoneFlag { uint32_t* counter; uint32_t* flag; uint32_t seq; uint32_t total; };

// ---- kernels_coop.hip ----
// proj != nullptr: the result is also written to staging record proj_offset + i (projective when it is the only output)
hipError_t decode_coop(hipStream_t st, const uint8_t* enc, size_t n, int32_t* out_ext, uint8_t* ok, bool or_identity, DoneFlag df = DoneFlag{});
hipError_t verify_prep_coop(hipStream_t st, const uint8_t* pubs, const uint8_t* sigs, const uint8_t* msgs, const uint32_t* off, size_t n,
                            uint8_t* flags_a, uint8_t* hbuf, uint8_t* sbuf, int32_t* a_ext);
hipError_t verify_prep_r_coop(hipStream_t st, const uint8_t* sigs, size_t n, uint8_t* flags_r, uint4* proj, size_t stride, size_t offset);
hipError_t poly_eval_coop(hipStream_t st, const int32_t* commits, int t, const uint32_t* idx, size_t n, int nbits, size_t per_poly,
                          uint8_t* oenc, int32_t* oext, DoneFlag df = DoneFlag{}, bool ext_proj = false);
hipError_t sign_coop(hipStream_t st, const uint8_t* x, const uint8_t* k, const uint8_t* pubs, const uint8_t* msgs, const uint32_t* off, size_t n,
                     uint8_t* sig, uint8_t* pub_out, const uint32_t* table_coop, DoneFlag df = DoneFlag{});
hipError_t mul_enc_coop(hipStream_t st, const uint8_t* sc, const uint8_t* penc, size_t n, uint8_t* oenc, int32_t* oext, uint8_t* ok, DoneFlag df = DoneFlag{},
                        uint4* proj = nullptr, size_t proj_stride = 0, uint8_t* flags_or = nullptr, int skip_bits = 0);      // proj / flags_or: the h A of a verification (record i, projective; flags_or[i] |= decodes << 2)
hipError_t verify_coop(hipStream_t st, const uint8_t* pubs, const uint8_t* sigs, const uint8_t* msgs, const uint32_t* off, size_t n, int flavor,
                       const uint32_t* table_coop, uint8_t* status, DoneFlag df = DoneFlag{});
// segs wavefronts (2..32) per evaluation, len coefficients each (segs * len >= t); part: n * segs * 40 words of device scratch
hipError_t poly_eval_seg(hipStream_t st, const int32_t* commits, int t, const uint32_t* idx, size_t n, size_t per_poly, int len, int segs,
                         uint32_t* part, uint8_t* oenc, int32_t* oext, DoneFlag df = DoneFlag{}, bool ext_proj = false);
hipError_t mul_coop(hipStream_t st, const uint8_t* sc, const int32_t* pext, size_t n, uint8_t* oenc, int32_t* oext, int skip_bits,
                    uint4* proj = nullptr, size_t proj_stride = 0, size_t proj_offset = 0, DoneFlag df = DoneFlag{}, size_t pt_mod = 0,
                    int pieces = 1, bool ext_proj = false, uint32_t* part = nullptr, uint32_t* pieces_buf = nullptr);
                    // pieces: 1, or 4 single-wavefront workgroups per item (the scalar in four pieces, the last to arrive adds them): then pieces_buf =
                    // KYB_COOP_PIECES_WORDS(n) words of device scratch, ZERO between launches (the kernel leaves it so);
                    // ext_proj: option ext.projective; part != nullptr (pieces == 1): the product goes there as an extended quad (40 words) for sum_coop
#define KYB_COOP_PIECES_WORDS(n) ((size_t)(n) * 161)      /* four 40-word records and one arrival counter per item */
// out[g] = sum of t points per group, from extended quads (part) or reference limbs (pts_ext)
hipError_t sum_coop(hipStream_t st, const uint32_t* part, const int32_t* pts_ext, size_t m, size_t t, uint8_t* oenc, int32_t* oext, bool ext_proj,
                    DoneFlag df = DoneFlag{});
// proj != nullptr: projective staging record i * src_mul, else the 40 reference limbs of point i
hipError_t finish_coop(hipStream_t st, const uint4* proj, size_t stride, const int32_t* pts_ext, size_t n, uint8_t* oenc, int32_t* oext, size_t src_mul,
                       DoneFlag df = DoneFlag{}, bool ext_proj = false);
// the same, one point per lane and one cooperative inversion per wavefront (mid-size batches)
hipError_t finish_wave(hipStream_t st, const uint4* proj, size_t stride, const int32_t* pts_ext, size_t n, uint8_t* oenc, int32_t* oext, size_t src_mul);
// sc_b != nullptr: n_b more scalars follow the first n in the same launch (their results behind the first n)
hipError_t mul_base_coop(hipStream_t st, const uint8_t* sc, size_t n, uint8_t* oenc, int32_t* oext, const uint32_t* table_coop,
                         uint4* proj = nullptr, size_t proj_stride = 0, size_t proj_offset = 0, const uint8_t* sc_b = nullptr, size_t n_b = 0,
                         DoneFlag df = DoneFlag{}, int waves = 1, bool ext_proj = false);      // waves: 1, or 4 wavefronts per item sharing the 43 windows
hipError_t coop_selftest(hipStream_t st, int op, const uint32_t* A, const uint32_t* B, uint32_t* out, const uint32_t* table_coop);
hipError_t diag_phase_stamps(uint64_t* buf);      // (cross-check build only, like coop_selftest)

// ---- kernels_verify.hip ----
hipError_t verify_prep_pts(hipStream_t st, const uint8_t* pub_enc, const int32_t* pubs_ext, const uint8_t* sigs, const uint8_t* msgs, const uint32_t* off, size_t n,
                           uint8_t* flags_a, uint8_t* hbuf, uint8_t* sbuf, int32_t* a_ext);      // public keys as points (+ their encodings)
hipError_t verify_prep(hipStream_t st, const uint8_t* pubs, const uint8_t* sigs, const uint8_t* msgs, const uint32_t* off, size_t n,
                       uint8_t* flags_a, uint8_t* hbuf, uint8_t* sbuf, int32_t* a_ext);
hipError_t verify_prep_r(hipStream_t st, const uint8_t* sigs, size_t n, uint8_t* flags_r, uint4* proj, size_t stride, size_t offset);
hipError_t verify_recover_final(hipStream_t st, const uint8_t* hbuf, size_t n, const int32_t* a_ext, const uint4* state, const uint4* proj, size_t stride, const uint8_t* flags_a,
                                const uint8_t* a_ok, const uint8_t* flags_r, int flavor, uint8_t* status, DoneFlag df);      // kernels_ladder.hip
hipError_t verify_final(hipStream_t st, const uint4* proj, size_t stride, size_t n, const uint8_t* flags_a, const uint8_t* flags_r, int flavor, uint8_t* status,
                        DoneFlag df = DoneFlag{});
// the equation of n verifications on encodings (h A at proj[i], s B at proj[n + i]; record i is overwritten): three launches
hipError_t verify_tail_enc(hipStream_t st, uint4* proj, size_t stride, size_t n, const uint8_t* sigs, const uint8_t* flags_a, int flavor, uint8_t* status,
                           DoneFlag df = DoneFlag{}, bool four = false);      // four: 4 items per shared inversion in the encode (k_verify_final_enc4)
hipError_t sign_hash(hipStream_t st, const uint8_t* x, const uint8_t* k, const uint8_t* msgs, const uint32_t* off, size_t n,
                     const uint8_t* r_enc, const uint8_t* a_enc, uint8_t* sig, DoneFlag df = DoneFlag{});
hipError_t eddsa_prep(hipStream_t st, const uint8_t* seeds, const uint8_t* msgs, const uint32_t* off, size_t n, uint8_t* xbuf, uint8_t* kbuf);

// ---- kernels_misc.hip / kernels_window.hip ----
hipError_t finish(hipStream_t st, const uint4* proj, size_t stride, size_t n, uint8_t* oenc, int32_t* oext, size_t src_mul, bool four = false);      // four: 4 items per shared inversion (k_finish4)
hipError_t encode_batched(hipStream_t st, const int32_t* pext, size_t n, uint8_t* oenc, bool four = false);      // four: 4 instead of FINISH_K points per inversion
hipError_t mul_window(int masked, bool from_enc, bool split, int grid, hipStream_t st, const uint8_t* sc, const uint8_t* penc, const int32_t* pext, size_t n,
                      uint8_t* oenc, int32_t* oext, uint8_t* ok, uint4* ws, uint4* proj, size_t stride);
hipError_t add(hipStream_t st, const int32_t* a, const int32_t* b, size_t n, int32_t* out, int subtract);
hipError_t equal(hipStream_t st, const int32_t* a, const int32_t* b, size_t n, uint8_t* eq);
hipError_t encode(hipStream_t st, const int32_t* pext, size_t n, uint8_t* oenc);
hipError_t decode(hipStream_t st, const uint8_t* enc, size_t n, int32_t* out_ext, uint8_t* ok);
hipError_t poly_eval(bool split, hipStream_t st, const int32_t* commits, int t, const uint32_t* idx, size_t n, int nbits, size_t per_poly,
                     uint8_t* oenc, int32_t* oext, uint4* proj, size_t stride);

// ---- kernels_msm.hip (+ msm_bases_coop in kernels_coop.hip): kyb_lincomb_public_batch over shared points ----
// bases: t x 43 x 40 words (64^w P_j as raw extended quads); tab: t x 43 x 32 x 40 words (cached multiples 1 .. 32 of every base);
// partial sums of output g land in staging records g * nchunks + c
hipError_t msm_bases_coop(hipStream_t st, const int32_t* pts_ext, size_t t, uint32_t* bases);
hipError_t msm_tables(hipStream_t st, const uint32_t* bases, size_t t, uint32_t* tab);
// out[g][i] = poly_g(indices[i] + 1) mod L for m secret polynomials of t coefficients at k public indices; part: m * k * segs * 32 bytes of scratch when segs > 1
hipError_t pripoly_eval(hipStream_t st, const uint8_t* coeffs, size_t m, size_t t, const uint32_t* indices, size_t k, uint32_t segs, uint8_t* part, uint8_t* out);
hipError_t lagrange_at_zero(hipStream_t st, const uint32_t* idx, size_t m, size_t t, uint8_t* out);      // out: m x t scalars, 32 bytes each
hipError_t msm_accumulate(hipStream_t st, const uint8_t* scalars, const uint32_t* tab, size_t m, size_t t, int chunk, size_t nchunks, uint4* proj, size_t stride);

// flags[i] = is_canonical(enc_i) | has_small_order(point of enc_i) << 1  (point.rs:286-337); bytes only, no field multiplication
hipError_t point_checks(hipStream_t st, const uint8_t* enc, size_t n, uint8_t* flags);
// benchmark diagnostic (kyb_diag_wave_stamps, diag_stamp.h): the five 64-bit sums into which the wavefronts of k_mul_ladder /
// k_mul_base64 on the CURRENT device add their start / end (shader cycles, 100 MHz ticks); nullptr = off.  Synchronous.
hipError_t diag_stamps_ladder(uint64_t* buf);
hipError_t diag_stamps_base(uint64_t* buf);
// benchmark diagnostic (kyb_diag_mad_peak): `grid` workgroups of 1024 threads issue nothing but v_mad_u64_u32, iters x 32 per wavefront;
// stamps[0..4] = the sums of diag_stamp.h
hipError_t diag_mad_peak(hipStream_t st, int grid, int iters, uint64_t* stamps, uint32_t* sink, bool sgpr_carry);
constexpr int MAD_PEAK_CHAINS_HOST = 8, MAD_PEAK_UNROLL_HOST = 4;      // = the kernel's chains x unroll (kernels_misc.hip)

// segs lanes per evaluation (len coefficients each): partial results as extended limbs in part_ext[n * segs], multipliers in part_sc
hipError_t poly_eval_part(hipStream_t st, const int32_t* commits, int t, const uint32_t* idx, size_t n, int nbits, size_t per_poly, int len, int segs,
                          int32_t* part_ext, uint8_t* part_sc);

}  // namespace launch
}  // namespace kyb
