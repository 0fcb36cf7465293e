// Race-hunting mode of the fused multi-rank kernels (TDIST_DEBUG_RACES=1 in the environment).
//
// The hand-over protocols (payload stores, drain, epoch flag; relaxed poll, one acquire, loads) are
// correct or wrong regardless of timing, but a wrong one usually passes on an idle machine.  This mode
// makes the timing hostile and the evidence loud:
//   * producers sleep a pseudo-random 0 .. ~60 us before they store a payload and again before they raise
//     its flag, so consumers really do arrive first and really do have to wait;
//   * on entry every kernel checks the flags of its parity: a word must hold zero or a legal epoch, and
//     never the epoch of the NEXT call (a peer that runs one call ahead writes that one into the other
//     parity) - anything else sets bit 2 of the error word;
//   * the host poisons a parity's payload buffer with NaNs right after the op that consumed it
//     (ops/*.py), so a read that is not ordered behind this call's flag returns NaNs, not last time's
//     plausible numbers.
// The reference gets the same effect from sleep kernels in its producers when `for_correctness` is set
// and from a straggler option (reference: python/triton_dist/kernels/nvidia/allgather.py:74-78,
// allgather_gemm.py:602-603, test/nvidia/test_allreduce.py:185-197).
#pragma once
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "tdist/atomics.hpp"
#include "tdist/call_state.hpp"

namespace tdist {

// host side: read once per process
inline int race_debug_enabled() {
  static const int on = [] {
    const char* e = getenv("TDIST_DEBUG_RACES");
    return (e && e[0] == '1') ? 1 : 0;
  }();
  return on;
}

// Wave-uniform pseudo-random delay (whole workgroup sleeps together when every wave calls it).
__device__ __forceinline__ void race_debug_delay(int enabled, unsigned salt) {
  if (!enabled) return;
  unsigned r = (unsigned)wall_ticks() * 2654435761u ^ (blockIdx.x * 40503u + salt * 69069u);
  r ^= r >> 13;
  const int n = __builtin_amdgcn_readfirstlane((int)(r & 15u));
  for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);      // ~127 * 64 cycles each
}

// Workgroup 0 checks `n` flags of this call's parity (see the list above); bit 2 of *err on a violation.
// (Older epochs are legal: the flag layout depends on the call's shape, so a word may have been written last
// by a call long ago.)
__device__ __forceinline__ void race_debug_check_flags(int enabled, const int* flags, int n, int epoch, int* err) {
  if (!enabled || blockIdx.x != 0) return;
  const int next = epoch % CALL_EPOCH_MOD + 1;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int v = load_relaxed_system(flags + i);
    if (v < 0 || v > CALL_EPOCH_MOD || v == next) atomicOr(err, 4);
  }
}

}  // namespace tdist
