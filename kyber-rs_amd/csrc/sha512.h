// SHA-512 (FIPS 180-4) for the Schnorr/EdDSA challenge h = SHA-512(R || A || msg)
// (/root/reference src/sign/schnorr/schnorr_sig.rs:128-141, src/sign/eddsa/eddsa_sig.rs:120-152;
// the reference takes it from the sha2 crate).  Streaming interface over byte fragments so the
// device kernel can hash R (32 B, registers), A (32 B, registers) and a per-lane message in HBM
// without staging a contiguous buffer.
#pragma once
#include <stdint.h>
#include "fe25519.h"  // KYB_HD

namespace kyb {

#if defined(__HIPCC__)
#define KYB_SHA_CONST __constant__
#else
#define KYB_SHA_CONST
#endif

static KYB_SHA_CONST const uint64_t kyb_sha512_k[80] = {
    0x428a2f98d728ae22ULL, 0x7137449123ef65cdULL, 0xb5c0fbcfec4d3b2fULL, 0xe9b5dba58189dbbcULL, 0x3956c25bf348b538ULL,
    0x59f111f1b605d019ULL, 0x923f82a4af194f9bULL, 0xab1c5ed5da6d8118ULL, 0xd807aa98a3030242ULL, 0x12835b0145706fbeULL,
    0x243185be4ee4b28cULL, 0x550c7dc3d5ffb4e2ULL, 0x72be5d74f27b896fULL, 0x80deb1fe3b1696b1ULL, 0x9bdc06a725c71235ULL,
    0xc19bf174cf692694ULL, 0xe49b69c19ef14ad2ULL, 0xefbe4786384f25e3ULL, 0x0fc19dc68b8cd5b5ULL, 0x240ca1cc77ac9c65ULL,
    0x2de92c6f592b0275ULL, 0x4a7484aa6ea6e483ULL, 0x5cb0a9dcbd41fbd4ULL, 0x76f988da831153b5ULL, 0x983e5152ee66dfabULL,
    0xa831c66d2db43210ULL, 0xb00327c898fb213fULL, 0xbf597fc7beef0ee4ULL, 0xc6e00bf33da88fc2ULL, 0xd5a79147930aa725ULL,
    0x06ca6351e003826fULL, 0x142929670a0e6e70ULL, 0x27b70a8546d22ffcULL, 0x2e1b21385c26c926ULL, 0x4d2c6dfc5ac42aedULL,
    0x53380d139d95b3dfULL, 0x650a73548baf63deULL, 0x766a0abb3c77b2a8ULL, 0x81c2c92e47edaee6ULL, 0x92722c851482353bULL,
    0xa2bfe8a14cf10364ULL, 0xa81a664bbc423001ULL, 0xc24b8b70d0f89791ULL, 0xc76c51a30654be30ULL, 0xd192e819d6ef5218ULL,
    0xd69906245565a910ULL, 0xf40e35855771202aULL, 0x106aa07032bbd1b8ULL, 0x19a4c116b8d2d0c8ULL, 0x1e376c085141ab53ULL,
    0x2748774cdf8eeb99ULL, 0x34b0bcb5e19b48a8ULL, 0x391c0cb3c5c95a63ULL, 0x4ed8aa4ae3418acbULL, 0x5b9cca4f7763e373ULL,
    0x682e6ff3d6b2b8a3ULL, 0x748f82ee5defb2fcULL, 0x78a5636f43172f60ULL, 0x84c87814a1f0ab72ULL, 0x8cc702081a6439ecULL,
    0x90befffa23631e28ULL, 0xa4506cebde82bde9ULL, 0xbef9a3f7b2c67915ULL, 0xc67178f2e372532bULL, 0xca273eceea26619cULL,
    0xd186b8c721c0c207ULL, 0xeada7dd6cde0eb1eULL, 0xf57d4f7fee6ed178ULL, 0x06f067aa72176fbaULL, 0x0a637dc5a2c898a6ULL,
    0x113f9804bef90daeULL, 0x1b710b35131c471bULL, 0x28db77f523047d84ULL, 0x32caab7b40c72493ULL, 0x3c9ebe0a15c9bebcULL,
    0x431d67c49c100d4cULL, 0x4cc5d4becb3e42b6ULL, 0x597f299cfc657e2aULL, 0x5fcb6fab3ad6faecULL, 0x6c44198c4a475817ULL};

struct sha512_ctx {
  uint64_t h[8];
  uint64_t w[16];    // current block, big-endian words being filled
  uint32_t fill;     // bytes in the current block
  uint64_t total;    // total bytes absorbed
};

// 64-bit rotation (n is a literal at every call site).  On the device: two v_alignbit_b32 on the halves — the generic form costs two
// 64-bit shifts and two ORs, and the rotations are a third of SHA-512's instructions.
KYB_HD uint64_t kyb_rotr64(uint64_t x, int n) {
#if defined(__HIP_DEVICE_COMPILE__)
  const uint32_t lo = (uint32_t)x, hi = (uint32_t)(x >> 32);
  uint32_t nl, nh;
  if (n < 32) { nl = __builtin_amdgcn_alignbit(hi, lo, (uint32_t)n); nh = __builtin_amdgcn_alignbit(lo, hi, (uint32_t)n); }
  else if (n == 32) { nl = hi; nh = lo; }
  else { nl = __builtin_amdgcn_alignbit(lo, hi, (uint32_t)(n - 32)); nh = __builtin_amdgcn_alignbit(hi, lo, (uint32_t)(n - 32)); }
  return ((uint64_t)nh << 32) | nl;
#else
  return (x >> n) | (x << (64 - n));
#endif
}

KYB_HD void sha512_init(sha512_ctx& c) {
  c.h[0] = 0x6a09e667f3bcc908ULL; c.h[1] = 0xbb67ae8584caa73bULL; c.h[2] = 0x3c6ef372fe94f82bULL; c.h[3] = 0xa54ff53a5f1d36f1ULL;
  c.h[4] = 0x510e527fade682d1ULL; c.h[5] = 0x9b05688c2b3e6c1fULL; c.h[6] = 0x1f83d9abfb41bd6bULL; c.h[7] = 0x5be0cd19137e2179ULL;
  for (int i = 0; i < 16; ++i) c.w[i] = 0;
  c.fill = 0;
  c.total = 0;
}

KYB_HD void sha512_compress(sha512_ctx& c) {
  uint64_t a = c.h[0], b = c.h[1], cc = c.h[2], d = c.h[3], e = c.h[4], f = c.h[5], g = c.h[6], h = c.h[7];
  uint64_t w[16];
  for (int i = 0; i < 16; ++i) w[i] = c.w[i];
#if defined(__HIPCC__)
#pragma unroll 1
#endif
  for (int r = 0; r < 80; r += 16) {
    KYB_UNROLL for (int i = 0; i < 16; ++i) {
      if (r) {
        uint64_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
        uint64_t s0 = kyb_rotr64(w15, 1) ^ kyb_rotr64(w15, 8) ^ (w15 >> 7);
        uint64_t s1 = kyb_rotr64(w2, 19) ^ kyb_rotr64(w2, 61) ^ (w2 >> 6);
        w[i] = w[i] + s0 + w[(i + 9) & 15] + s1;
      }
      uint64_t S1 = kyb_rotr64(e, 14) ^ kyb_rotr64(e, 18) ^ kyb_rotr64(e, 41);
      uint64_t ch = (e & f) ^ (~e & g);
      uint64_t t1 = h + S1 + ch + kyb_sha512_k[r + i] + w[i];
      uint64_t S0 = kyb_rotr64(a, 28) ^ kyb_rotr64(a, 34) ^ kyb_rotr64(a, 39);
      uint64_t mj = (a & b) ^ (a & cc) ^ (b & cc);
      uint64_t t2 = S0 + mj;
      h = g; g = f; f = e; e = d + t1; d = cc; cc = b; b = a; a = t1 + t2;
    }
  }
  c.h[0] += a; c.h[1] += b; c.h[2] += cc; c.h[3] += d; c.h[4] += e; c.h[5] += f; c.h[6] += g; c.h[7] += h;
  for (int i = 0; i < 16; ++i) c.w[i] = 0;
  c.fill = 0;
}

// absorb one byte (block position is data-independent within a wave only if lengths agree; the
// device kernel pads every lane to the wave's longest message, see sign kernel)
KYB_HD void sha512_byte(sha512_ctx& c, uint32_t byte) {
  uint32_t wi = c.fill >> 3, sh = (7u - (c.fill & 7u)) * 8u;
  uint64_t v = (uint64_t)(byte & 0xffu) << sh;
  for (int k = 0; k < 16; ++k) c.w[k] |= (wi == (uint32_t)k) ? v : 0ULL;
  c.fill += 1;
  c.total += 1;
  if (c.fill == 128) sha512_compress(c);
}
// absorb 32 bytes given as 8 little-endian 32-bit words (an encoded point or scalar)
KYB_HD void sha512_words32(sha512_ctx& c, const uint32_t w[8]) {
  for (int i = 0; i < 8; ++i)
    for (int b = 0; b < 4; ++b) sha512_byte(c, (w[i] >> (8 * b)) & 0xffu);
}
// absorb 64 bytes given as 16 little-endian 32-bit words into an EMPTY block position (fill == 0):
// the R || A prefix of the signing hash; no per-byte select
KYB_HD void sha512_words64(sha512_ctx& c, const uint32_t w[16]) {
  for (int i = 0; i < 8; ++i)
    c.w[i] = ((uint64_t)__builtin_bswap32(w[2 * i]) << 32) | (uint64_t)__builtin_bswap32(w[2 * i + 1]);
  c.fill = 64;
  c.total += 64;
}
// the same for 32 bytes (8 words): a seed or a prefix at the start of a block
KYB_HD void sha512_words32_at0(sha512_ctx& c, const uint32_t w[8]) {
  for (int i = 0; i < 4; ++i)
    c.w[i] = ((uint64_t)__builtin_bswap32(w[2 * i]) << 32) | (uint64_t)__builtin_bswap32(w[2 * i + 1]);
  c.fill = 32;
  c.total += 32;
}
// absorb eight bytes, given as the big-endian word they form, at a block position that is a multiple of 8: ONE 16-way select instead of eight
KYB_HD void sha512_word64_be(sha512_ctx& c, uint64_t v) {
  const uint32_t wi = c.fill >> 3;
  for (int k = 0; k < 16; ++k) c.w[k] = (wi == (uint32_t)k) ? v : c.w[k];      // (nothing has been absorbed into that word yet: it is zero)
  c.fill += 8;
  c.total += 8;
  if (c.fill == 128) sha512_compress(c);
}
// a message of n bytes at any address: byte by byte up to the next multiple of 8 of the block position, then eight bytes per step — their loads are
// issued together, so a lane waits for memory once per eight bytes, not once per byte (a 32-byte digest behind R || A: 4 steps; absorbing it byte
// by byte was 22 of the 84 us of a one-signature call, round 6) — and the rest byte by byte
KYB_HD void sha512_bytes(sha512_ctx& c, const uint8_t* p, uint32_t n) {
  uint32_t i = 0;
  for (; i < n && (c.fill & 7u) != 0u; ++i) sha512_byte(c, p[i]);
  for (; i + 8u <= n; i += 8u) {
    uint32_t b[8];
    for (int j = 0; j < 8; ++j) b[j] = p[i + j];
    const uint32_t hi = (b[0] << 24) | (b[1] << 16) | (b[2] << 8) | b[3], lo = (b[4] << 24) | (b[5] << 16) | (b[6] << 8) | b[7];
    sha512_word64_be(c, ((uint64_t)hi << 32) | (uint64_t)lo);
  }
  for (; i < n; ++i) sha512_byte(c, p[i]);
}
// finish; digest returned as 16 little-endian 32-bit words of the 64-byte digest string
KYB_HD void sha512_final(uint32_t out[16], sha512_ctx& c) {
  uint64_t bits = c.total * 8u;
  sha512_byte(c, 0x80u);
  c.total -= 1;
  if (c.fill > 112) sha512_compress(c);   // also covers fill == 0 after an exact block
  c.w[15] = bits;                         // 128-bit length, high half zero
  sha512_compress(c);
  for (int i = 0; i < 8; ++i) {
    uint64_t v = c.h[i];                  // big-endian bytes of v -> digest[8i .. 8i+7]
    uint32_t hi = (uint32_t)(v >> 32), lo = (uint32_t)v;
    out[2 * i] = __builtin_bswap32(hi);
    out[2 * i + 1] = __builtin_bswap32(lo);
  }
}

}  // namespace kyb
