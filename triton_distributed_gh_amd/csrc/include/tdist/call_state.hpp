// Device-resident call numbering for the fused multi-rank kernels.
//
// Every fused op double-buffers its symmetric workspaces and flags by call parity and tags its flags
// with a per-call epoch.  When the host passes both as launch arguments, a HIP graph that captured
// the launch would replay one parity and one epoch forever.  Here the call number lives in device
// memory instead (`state[0]`, one word per context and rank, zeroed when the context is created):
//   * every workgroup reads it on entry (call_begin) and derives parity and flag value from it;
//   * every workgroup reports on exit (call_end); the last one re-arms the count and advances the
//     call number, so the next launch on the stream - eager or replayed - sees the next call.
// All ranks issue the same sequence of launches, so their counters stay equal without any exchange.
// A workgroup can only finish after it has read the call number, hence nobody sees the new value
// early; the next launch starts after this kernel's end-of-kernel release.
#pragma once
#include <hip/hip_runtime.h>

#include "tdist/atomics.hpp"

namespace tdist {

constexpr int CALL_EPOCH_MOD = 4093;   // flag values are 1 .. CALL_EPOCH_MOD (0 = never written)

// `slot` is one int of LDS owned by the caller.  Ends with a workgroup barrier.
__device__ __forceinline__ int call_begin(const int* state, volatile int* slot) {
  if (threadIdx.x == 0) *slot = load_relaxed_agent(state);
  __syncthreads();
  return __builtin_amdgcn_readfirstlane(*slot);
}

__device__ __forceinline__ int call_parity(int call) { return call & 1; }
__device__ __forceinline__ int call_epoch(int call) { return call % CALL_EPOCH_MOD + 1; }

// Starts with a workgroup barrier (all of the workgroup's work is issued; completion of its stores is
// the kernel's own business - the counter only orders launches on one stream).
__device__ __forceinline__ void call_end(int* state, int call) {
  __syncthreads();
  if (threadIdx.x == 0) {
    const int prev = atom_add_relaxed_agent(state + 1, 1);
    if (prev == (int)gridDim.x - 1) {
      store_relaxed_agent(state + 1, 0);
      store_relaxed_agent(state, (call + 1) % (2 * CALL_EPOCH_MOD));
    }
  }
}

}  // namespace tdist
