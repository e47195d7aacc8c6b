// Wave clock stamps for benchmarks (kyb_diag_wave_stamps, include/kyber_ed25519.h).
//
// While a stamp buffer is set, each wavefront of k_mul_ladder / k_mul_base64 reads s_memtime (shader cycles) and
// s_memrealtime (constant 100 MHz) when it starts and when it ends and adds them into four 64-bit sums:
//   buf[0] += c0, buf[1] += r0   at wavefront start        buf[2] += c1, buf[3] += r1, buf[4] += 1   at wavefront end
// so (buf[2] - buf[0]) / (buf[3] - buf[1]) x 100 MHz is the clock the kernel really ran at (the chip's DVFS settles at a
// different clock for every instruction mix) — what bench.py needs to turn a kernel duration into SIMD cycles — and
// (buf[2] - buf[0]) / buf[4] the mean lifetime of a wavefront in cycles.  The sums wrap modulo 2^64; their differences do not.
// With no buffer set (the default) a kernel executes one scalar load and one uniform branch each way and nothing else.
//
// Nothing of the stamp is live across the kernel body, and it needs no wavefront index: the fixed-base kernel sits exactly at
// its SGPR and VGPR budgets (96 / 128, 0 scratch) and both a kernel argument carried across its loop and a workgroup id kept
// for the end spill (measured: 8-16 B of scratch).  The buffer pointer therefore lives in a __device__ variable of the
// kernel's translation unit: a device has ONE buffer per kernel unit, shared by every context on it (a diagnostic, not a
// product path).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kyb {

struct WaveClock {
  __device__ __forceinline__ static void read(uint64_t& cyc, uint64_t& rt) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(cyc), "=s"(rt) :: "memory");
#else
    cyc = rt = 0;      // host pass of hipcc only parses this
#endif
  }
  // buf: the sums' buffer as loaded from the unit's slot (nullptr = off)
  __device__ __forceinline__ static void stamp(uint64_t* buf, int at_end) {
    if (buf == nullptr) return;
    uint64_t c, r;
    read(c, r);
    uint32_t t = threadIdx.x;
    asm volatile("" : "+v"(t));      // opaque: otherwise the start's (threadIdx.x & 63) is kept for the end — in a spill slot
    if ((t & 63u) == 0) {
      atomicAdd(reinterpret_cast<unsigned long long*>(buf + 2 * at_end), (unsigned long long)c);
      atomicAdd(reinterpret_cast<unsigned long long*>(buf + 2 * at_end + 1), (unsigned long long)r);
      if (at_end) atomicAdd(reinterpret_cast<unsigned long long*>(buf + 4), 1ull);
    }
  }
  __device__ __forceinline__ static uint64_t* load_slot(uint64_t* const* slot) { return *reinterpret_cast<uint64_t* const volatile*>(slot); }
};
constexpr size_t KYB_STAMP_WORDS = 5;

// One slot PAIR per translation unit that carries stamps (static: each unit has its own).  Two variables holding the same pointer, one
// read at wavefront start and one at its end: with a single variable the compiler keeps its address in two SGPRs across the kernel.
#define KYB_DEFINE_STAMP_SLOT()                                     \
  static __device__ uint64_t* kyb_stamp_buf_begin = nullptr;       \
  static __device__ uint64_t* kyb_stamp_buf_end = nullptr;         \
  static hipError_t kyb_set_stamp_slot(uint64_t* buf) {            \
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(kyb_stamp_buf_begin), &buf, sizeof(buf)); \
    if (e != hipSuccess) return e;                                 \
    return hipMemcpyToSymbol(HIP_SYMBOL(kyb_stamp_buf_end), &buf, sizeof(buf)); \
  }
#define KYB_STAMP_BEGIN() ::kyb::WaveClock::stamp(::kyb::WaveClock::load_slot(&kyb_stamp_buf_begin), 0)
#define KYB_STAMP_END() ::kyb::WaveClock::stamp(::kyb::WaveClock::load_slot(&kyb_stamp_buf_end), 1)

}  // namespace kyb
