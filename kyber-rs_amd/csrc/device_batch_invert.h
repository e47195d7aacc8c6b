// Montgomery's trick on the device (included by the kernel units that share one field inversion between FINISH_K items).
#pragma once
#include "fe_invert_gcd.h"

namespace kyb {

// Montgomery's trick over K values with the running prefixes held in locals of a template recursion (an
// indexed `fe pre[K]` array ends up in scratch): level T multiplies d_T onto the prefix, the innermost level
// inverts once, and on the way back every level peels its own 1/d_T off.
//   load(t, d)   supplies d_t (already forced non-zero)        emit(t, dinv)   consumes 1/d_t
template <int T, int K, class Load, class Emit>
__device__ __forceinline__ void batch_invert(const fe& prefix_prev, fe& inv_prev, Load& load, Emit& emit) {
  fe d, pre, inv, di;
  load(T, d);
  if (T == 0) fe_copy(pre, d); else fe_mul(pre, prefix_prev, d);
  if constexpr (T + 1 < K) batch_invert<T + 1, K>(pre, inv, load, emit); else fe_inv(inv, pre);
  if (T == 0) fe_copy(di, inv); else fe_mul(di, inv, prefix_prev);
  emit(T, di);
  if (T > 0) { load(T, d); fe_mul(inv_prev, inv, d); }
}

}  // namespace kyb
