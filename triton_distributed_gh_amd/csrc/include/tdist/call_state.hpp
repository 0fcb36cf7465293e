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
//
// The same "last workgroup out" also re-arms the kernel's work counters (ticket, arrival counters): it
// zeroes them for the next launch, so a launch is ONE kernel node - no memset node in front of it -
// and costs one dispatch instead of two.  The words are zero when the context is created; a launch
// that gave up on a timeout leaves them undefined, which is why the error word is sticky.
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
// the kernel's own business - the counter only orders launches on one stream).  `fin` counts finished
// workgroups; the last one zeroes `counters[0 .. n_counters)` (every other workgroup has drawn its last
// ticket before it reported, so nobody reads them any more in this launch) and, when `state` is set,
// advances the call number.
__device__ __forceinline__ void call_end(int* state, int call, int* fin, int* counters, int n_counters) {
  __syncthreads();
  const int prev = threadIdx.x == 0 ? atom_add_relaxed_agent(fin, 1) : 0;
  if (__builtin_amdgcn_readfirstlane(prev) != (int)gridDim.x - 1) return;     // (only wave 0 can pass)
  if (threadIdx.x < 64) {
    for (int i = threadIdx.x; i < n_counters; i += 64) store_relaxed_agent(counters + i, 0);
    if (threadIdx.x == 0) {
      store_relaxed_agent(fin, 0);
      if (state) store_relaxed_agent(state, (call + 1) % (2 * CALL_EPOCH_MOD));
    }
  }
}

__device__ __forceinline__ void call_end(int* state, int call) { call_end(state, call, state + 1, nullptr, 0); }

}  // namespace tdist
