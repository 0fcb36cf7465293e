// Tile-centric distributed primitives as HIP device functions.
//
// Semantics follow the reference's `distributed` dialect ops
// (reference: include/TritonDistributed/Dialect/Distributed/IR/DistributedOps.td:45-189,
//  python/triton_dist/language/distributed_ops.py:56-111, docs/primitives.md:1-59) and the AMD
// lowering of `wait` (lib/Conversion/TritonDistributedToLLVM/AMD/DistributedOpToLLVM.cpp:96-215):
//   wait(flags, n, scope, semantic, value)  - workgroup-collective: threads stride over n flags,
//       each spins with an atomic load until flag == value, then the workgroup barriers.
//   consume_token(x, token)                 - identity that orders later uses of x after the wait.
//   notify(ptr, rank, signal, op, scope)    - single-thread: make prior stores visible, then
//       set/add `signal` on rank `rank`'s copy of the symmetric word `ptr` (no AMD lowering exists
//       in the reference; defined here from the NVIDIA semantics, DistributedOpToLLVM.cpp:233-342).
//   symm_at(ptr, rank) / rank() / num_ranks()
// Enum values are the dialect's numeric values (DistributedAttrDefs.td:36-53).
#pragma once
#include "tdist/atomics.hpp"
#include "tdist/common.hpp"

namespace tdist {

enum SignalOp : int { SIGNAL_SET = 1, SIGNAL_ADD = 2 };
enum CommScope : int { COMM_GPU = 1, COMM_INTRA_NODE = 2, COMM_INTER_NODE = 3 };
enum MemScope : int { SCOPE_WORKGROUP = 0, SCOPE_GPU = 1, SCOPE_SYS = 2 };
enum MemSemantic : int { SEM_RELAXED = 0, SEM_ACQUIRE = 1 };

__device__ __forceinline__ int rank(const SymmCtx& ctx) { return ctx.rank; }
__device__ __forceinline__ int num_ranks(const SymmCtx& ctx) { return ctx.num_ranks; }

// Translate a pointer into this rank's symmetric arena (data segment, signal arena or the
// virtual-memory-management arena) to the same object inside `peer`'s arena.  Pointers outside both arenas are returned unchanged
// when peer == rank and are otherwise invalid.
template <typename T>
__device__ __forceinline__ T* symm_at(const SymmCtx& ctx, T* ptr, int peer) {
  const uint64_t p = reinterpret_cast<uint64_t>(ptr);
  const uint64_t hb = ctx.heap_base[ctx.rank];
  if (p - hb < ctx.heap_bytes) return reinterpret_cast<T*>(ctx.heap_base[peer] + (p - hb));
  const uint64_t sb = ctx.sig_base[ctx.rank];
  if (p - sb < ctx.sig_bytes) return reinterpret_cast<T*>(ctx.sig_base[peer] + (p - sb));
  for (int k = 0; k < ctx.num_extra; ++k) {
    const uint64_t eb = ctx.extra_base[k][ctx.rank];
    if (p - eb < ctx.heap_bytes) return reinterpret_cast<T*>(ctx.extra_base[k][peer] + (p - eb));
  }
  const uint64_t vb = ctx.vmm_base[ctx.rank];
  if (p - vb < ctx.vmm_bytes) return reinterpret_cast<T*>(ctx.vmm_base[peer] + (p - vb));
  return ptr;
}

template <typename T>
__device__ __forceinline__ T load_scoped(const T* p, int scope, int semantic) {
  if (semantic == SEM_ACQUIRE) {
    if (scope == SCOPE_SYS) return load_acquire_system(p);
    if (scope == SCOPE_GPU) return load_acquire_agent(p);
    return load_acquire_workgroup(p);
  }
  if (scope == SCOPE_SYS) return load_relaxed_system(p);
  if (scope == SCOPE_GPU) return load_relaxed_agent(p);
  return load_relaxed_workgroup(p);
}

// Spin until *p == value.  Polls relaxed (cheap, L2/fabric served), then issues ONE acquire fence
// of the requested scope.  Returns false if the spin budget ran out.
template <typename T>
__device__ __forceinline__ bool spin_until_eq(const T* p, T value, int scope,
                                              uint64_t budget = TDIST_SPIN_TIMEOUT_TICKS) {
  const uint64_t t0 = wall_ticks();
  int backoff = 0;
  while (true) {
    T v = (scope == SCOPE_SYS) ? load_relaxed_system(p) : load_relaxed_agent(p);
    if (v == value) break;
    if (++backoff > 16) {
      __builtin_amdgcn_s_sleep(2);
      if ((backoff & 255) == 0 && wall_ticks() - t0 > budget) return false;
    }
  }
  return true;
}

template <typename T>
__device__ __forceinline__ bool spin_until_ge(const T* p, T value, int scope,
                                              uint64_t budget = TDIST_SPIN_TIMEOUT_TICKS) {
  const uint64_t t0 = wall_ticks();
  int backoff = 0;
  while (true) {
    T v = (scope == SCOPE_SYS) ? load_relaxed_system(p) : load_relaxed_agent(p);
    if (v >= value) break;
    if (++backoff > 16) {
      __builtin_amdgcn_s_sleep(2);
      if ((backoff & 255) == 0 && wall_ticks() - t0 > budget) return false;
    }
  }
  return true;
}

__device__ __forceinline__ void acquire_fence(int scope) {
  if (scope == SCOPE_SYS)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  else if (scope == SCOPE_GPU)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  else
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ void release_fence(int scope) {
  if (scope == SCOPE_SYS)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  else if (scope == SCOPE_GPU)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  else
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
}

// Workgroup-collective wait on `num_barriers` consecutive flags.  Returns a token (the number of
// flags that timed out; 0 on success) that callers thread through consume_token().
template <typename T>
__device__ __forceinline__ int wait(const T* barrier_ptrs, int num_barriers, int scope, int semantic,
                                    T wait_value = 1, int* err_word = nullptr) {
  int failed = 0;
  for (int i = threadIdx.x; i < num_barriers; i += blockDim.x) {
    // the common case - the flag is already there - costs one load
    const T* fp = barrier_ptrs + i;
    const T v = (scope == SCOPE_SYS) ? load_relaxed_system(fp) : load_relaxed_agent(fp);
    if (v == wait_value) continue;
    // Once any wait of this launch has timed out the error word is sticky and later waits give up
    // at once, so a lost peer costs one spin budget, not one per tile.
    const bool dead = err_word && load_relaxed_agent(err_word) != 0;
    if (dead) {
      failed = 1;
    } else if (!spin_until_eq(fp, wait_value, scope)) {
      failed = 1;
      if (err_word) atomicOr(err_word, 1);
    }
  }
  // Only waves that polled need the cache invalidate; it covers the whole CU's L1, and the
  // barrier below orders every other wave's loads behind it.
  const int wave_first = __builtin_amdgcn_readfirstlane((int)(threadIdx.x & ~63u));
  if (semantic == SEM_ACQUIRE && wave_first < num_barriers) acquire_fence(scope);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  return failed;
}

// Single-wave variant: lanes stride over the flags, no workgroup barrier.
template <typename T>
__device__ __forceinline__ int wait_wave(const T* barrier_ptrs, int num_barriers, int scope, int semantic,
                                         T wait_value = 1) {
  int failed = 0;
  for (int i = lane_id(); i < num_barriers; i += 64) {
    if (!spin_until_eq(barrier_ptrs + i, wait_value, scope)) failed = 1;
  }
  if (semantic == SEM_ACQUIRE) acquire_fence(scope);
  return failed;
}

// Identity on `value`; an opaque asm use of the token keeps the compiler from hoisting loads of
// `value` above the wait that produced the token.
template <typename T>
__device__ __forceinline__ T* consume_token(T* value, int token) {
  asm volatile("" : "+v"(token)::"memory");
  return value;
}

// Workgroup-uniform dynamic work fetch for persistent kernels: one thread draws the next ticket
// from a device-wide counter and the value is returned in a scalar register to every wave.
// The leading barrier is load-bearing: without a convergent operation between a trailing
// `if (threadIdx.x == 0) {...}` block of the previous iteration and this one, hipcc threads thread
// 0's path through both blocks, the loop becomes divergent and thread 0 falls out of step with the
// workgroup's barriers.  `slot` is 4 bytes of LDS inside the kernel's single dynamic array.
__device__ __forceinline__ int fetch_ticket_block(int* counter, volatile int* slot) {
  __syncthreads();
  if (threadIdx.x == 0) *slot = atomicAdd(counter, 1);
  __syncthreads();
  return __builtin_amdgcn_readfirstlane(*slot);
}

// Single-thread notify.  Call from ONE thread after the workgroup's stores were drained
// (`s_waitcnt vmcnt(0)` in every storing wave + __syncthreads()).
template <typename T>
__device__ __forceinline__ void notify(const SymmCtx& ctx, T* ptr, int peer, T signal = 1,
                                       int sig_op = SIGNAL_SET, int comm_scope = COMM_INTRA_NODE) {
  T* dst = (comm_scope == COMM_GPU) ? ptr : symm_at(ctx, ptr, peer);
  if (comm_scope == COMM_GPU) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (sig_op == SIGNAL_SET)
      store_relaxed_agent(dst, signal);
    else
      (void)atom_add_relaxed_agent(dst, signal);
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (sig_op == SIGNAL_SET)
      store_relaxed_system(dst, signal);
    else
      (void)atom_add_relaxed_system(dst, signal);
  }
}

}  // namespace tdist
