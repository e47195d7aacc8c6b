// Host-side look at the scalars of a small call (c_abi.inc, kyb_mul_batch): the number of leading zero bits ALL of them have as 256-bit
// little-endian integers (256 when every scalar is zero).  The scan stops early once a scalar reaches bit `enough` (the caller only
// cares whether all scalars are short); the value returned is then merely <= 255 - enough.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace kyb {

inline int common_leading_zero_bits(const uint8_t* scalars, size_t n, int enough = 64) {
  int top = -1;                                    // highest set bit over all scalars seen so far
  for (size_t i = 0; i < n && top < enough; ++i) {
    const uint8_t* s = scalars + 32 * i;
    for (int b = 31; b >= 0; --b) {
      if (s[b] == 0) continue;
      int hb = 7;
      while (!((s[b] >> hb) & 1)) --hb;
      if (8 * b + hb > top) top = 8 * b + hb;
      break;
    }
  }
  return 255 - top;
}

}  // namespace kyb
