// Barrier over all workgroups of one launch (every workgroup must be resident: grid <= what the
// chip holds at the kernel's occupancy, as in the persistent kernels of this library).
//
// Role in the reference: `barrier_on_this_grid` / `unsafe_barrier_on_this_grid` /
// `cooperative_barrier_on_this_grid` (reference: python/triton_dist/kernels/nvidia/common_ops.py:56-130),
// used by the persistent flash-decode kernel (kernels/nvidia/flash_decode.py:719).  The reference
// flips bit 31 of one counter; here `state[0]` counts arrivals and `state[1]` is a generation
// number, so the barrier is reusable without a reset and a spin can be bounded.
//
// Memory: data written before the barrier by any workgroup is visible to every workgroup after it
// (one agent-scope release by the arriving lane after the workgroup drained its stores, one
// agent-scope acquire after the generation moved).
#pragma once
#include "tdist/atomics.hpp"
#include "tdist/common.hpp"
#include "tdist/primitives.hpp"

namespace tdist {

// `state`: two zero-initialised uint32 words, used by this grid only.  Returns false on timeout
// (and sets bit 1 of *err_word when given).  Call from ALL threads of ALL workgroups.
__device__ __forceinline__ bool barrier_on_this_grid(unsigned* state, int* err_word = nullptr) {
  __shared__ int s_ok;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0 && threadIdx.z == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned nblocks = gridDim.x * gridDim.y * gridDim.z;
    const unsigned gen = load_relaxed_agent(state + 1);   // cannot move before this block arrives
    bool ok = true;
    if (atom_add_relaxed_agent(state, 1u) == nblocks - 1) {
      store_relaxed_agent(state, 0u);                     // re-arm, then open the gate
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      store_relaxed_agent(state + 1, gen + 1);
    } else {
      const uint64_t t0 = wall_ticks();
      int backoff = 0;
      while (load_relaxed_agent(state + 1) == gen) {
        if (++backoff > 16) {
          __builtin_amdgcn_s_sleep(2);
          if ((backoff & 255) == 0 && wall_ticks() - t0 > TDIST_SPIN_TIMEOUT_TICKS) {
            ok = false;
            break;
          }
        }
      }
    }
    if (!ok && err_word) atomicOr(err_word, 2);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    s_ok = ok ? 1 : 0;
  }
  __syncthreads();
  return s_ok != 0;
}

}  // namespace tdist
