// Scoped atomic helpers for gfx950 device code.
//
// Same surface as the reference's AMD `libdevice` atomics
// (reference: backends/amd/language/hip/libdevice.py:518-734 and their LLVM
// lowering lib/Conversion/TritonDistributedToLLVM/AMD/LibDeviceToLLVM.cpp:129-308):
//   load_{acquire,relaxed}_{workgroup,agent,system}
//   store_{release,relaxed}_{workgroup,agent,system}
//   red_add_release_{agent,system}
//   atom_add_{acquire,relaxed,acqrel}_{agent,system}
//   atom_cas_{acquire,release,relaxed,acqrel}_relaxed_{agent,system}
//   syncthreads (= release fence + s_barrier + acquire fence at workgroup scope)
// Here they are ordinary inline HIP functions over the clang __hip_atomic builtins, so the
// compiler emits the sc0/sc1 cache-policy bits and buffer_wbl2 / buffer_inv for the scope.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tdist {

#define TDIST_SCOPE_WG __HIP_MEMORY_SCOPE_WORKGROUP
#define TDIST_SCOPE_AGENT __HIP_MEMORY_SCOPE_AGENT
#define TDIST_SCOPE_SYS __HIP_MEMORY_SCOPE_SYSTEM

#define TDIST_DEF_LOAD(name, order, scope)                                   \
  template <typename T>                                                      \
  __device__ __forceinline__ T name(const T* p) {                            \
    return __hip_atomic_load(const_cast<T*>(p), order, scope);               \
  }
TDIST_DEF_LOAD(load_acquire_workgroup, __ATOMIC_ACQUIRE, TDIST_SCOPE_WG)
TDIST_DEF_LOAD(load_acquire_agent, __ATOMIC_ACQUIRE, TDIST_SCOPE_AGENT)
TDIST_DEF_LOAD(load_acquire_system, __ATOMIC_ACQUIRE, TDIST_SCOPE_SYS)
TDIST_DEF_LOAD(load_relaxed_workgroup, __ATOMIC_RELAXED, TDIST_SCOPE_WG)
TDIST_DEF_LOAD(load_relaxed_agent, __ATOMIC_RELAXED, TDIST_SCOPE_AGENT)
TDIST_DEF_LOAD(load_relaxed_system, __ATOMIC_RELAXED, TDIST_SCOPE_SYS)
#undef TDIST_DEF_LOAD

#define TDIST_DEF_STORE(name, order, scope)                                  \
  template <typename T>                                                      \
  __device__ __forceinline__ void name(T* p, T v) {                          \
    __hip_atomic_store(p, v, order, scope);                                  \
  }
TDIST_DEF_STORE(store_release_workgroup, __ATOMIC_RELEASE, TDIST_SCOPE_WG)
TDIST_DEF_STORE(store_release_agent, __ATOMIC_RELEASE, TDIST_SCOPE_AGENT)
TDIST_DEF_STORE(store_release_system, __ATOMIC_RELEASE, TDIST_SCOPE_SYS)
TDIST_DEF_STORE(store_relaxed_workgroup, __ATOMIC_RELAXED, TDIST_SCOPE_WG)
TDIST_DEF_STORE(store_relaxed_agent, __ATOMIC_RELAXED, TDIST_SCOPE_AGENT)
TDIST_DEF_STORE(store_relaxed_system, __ATOMIC_RELAXED, TDIST_SCOPE_SYS)
#undef TDIST_DEF_STORE

// Reductions (no return value needed by the caller).
template <typename T>
__device__ __forceinline__ void red_add_release_agent(T* p, T v) {
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELEASE, TDIST_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ void red_add_release_system(T* p, T v) {
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELEASE, TDIST_SCOPE_SYS);
}

#define TDIST_DEF_ADD(name, order, scope)                                    \
  template <typename T>                                                      \
  __device__ __forceinline__ T name(T* p, T v) {                             \
    return __hip_atomic_fetch_add(p, v, order, scope);                       \
  }
TDIST_DEF_ADD(atom_add_acquire_agent, __ATOMIC_ACQUIRE, TDIST_SCOPE_AGENT)
TDIST_DEF_ADD(atom_add_relaxed_agent, __ATOMIC_RELAXED, TDIST_SCOPE_AGENT)
TDIST_DEF_ADD(atom_add_acqrel_agent, __ATOMIC_ACQ_REL, TDIST_SCOPE_AGENT)
TDIST_DEF_ADD(atom_add_release_agent, __ATOMIC_RELEASE, TDIST_SCOPE_AGENT)
TDIST_DEF_ADD(atom_add_acquire_system, __ATOMIC_ACQUIRE, TDIST_SCOPE_SYS)
TDIST_DEF_ADD(atom_add_relaxed_system, __ATOMIC_RELAXED, TDIST_SCOPE_SYS)
TDIST_DEF_ADD(atom_add_acqrel_system, __ATOMIC_ACQ_REL, TDIST_SCOPE_SYS)
TDIST_DEF_ADD(atom_add_release_system, __ATOMIC_RELEASE, TDIST_SCOPE_SYS)
#undef TDIST_DEF_ADD

// Compare-and-swap: returns the value found at *p (== expected on success).
#define TDIST_DEF_CAS(name, succ, scope)                                     \
  template <typename T>                                                      \
  __device__ __forceinline__ T name(T* p, T expected, T desired) {           \
    __hip_atomic_compare_exchange_strong(p, &expected, desired, succ,        \
                                         __ATOMIC_RELAXED, scope);           \
    return expected;                                                         \
  }
TDIST_DEF_CAS(atom_cas_acquire_relaxed_agent, __ATOMIC_ACQUIRE, TDIST_SCOPE_AGENT)
TDIST_DEF_CAS(atom_cas_release_relaxed_agent, __ATOMIC_RELEASE, TDIST_SCOPE_AGENT)
TDIST_DEF_CAS(atom_cas_relaxed_relaxed_agent, __ATOMIC_RELAXED, TDIST_SCOPE_AGENT)
TDIST_DEF_CAS(atom_cas_acqrel_relaxed_agent, __ATOMIC_ACQ_REL, TDIST_SCOPE_AGENT)
TDIST_DEF_CAS(atom_cas_acquire_relaxed_system, __ATOMIC_ACQUIRE, TDIST_SCOPE_SYS)
TDIST_DEF_CAS(atom_cas_release_relaxed_system, __ATOMIC_RELEASE, TDIST_SCOPE_SYS)
TDIST_DEF_CAS(atom_cas_relaxed_relaxed_system, __ATOMIC_RELAXED, TDIST_SCOPE_SYS)
TDIST_DEF_CAS(atom_cas_acqrel_relaxed_system, __ATOMIC_ACQ_REL, TDIST_SCOPE_SYS)
#undef TDIST_DEF_CAS

// Workgroup barrier with release/acquire fences at workgroup scope.
__device__ __forceinline__ void syncthreads() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

__device__ __forceinline__ int thread_idx(int axis) {
  return axis == 0 ? threadIdx.x : (axis == 1 ? threadIdx.y : threadIdx.z);
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (64 - 1); }

// max(x, x of lane ^ 16) / max(x, x of lane ^ 32) with the gfx950 row-swap instructions (register file only;
// a cross-lane shuffle would take a round trip through the LDS pipeline on the softmax's critical path)
// The two results are copied into scalars before the bit casts: a bit cast applied directly to an element of
// the builtin's vector result (`__builtin_bit_cast(float, r[1])`) reads element 0 with this compiler, and the
// max silently degenerates to "the partner's value on half of the lanes".
template <int XOR>
__device__ __forceinline__ float xor_max(float x) {
  const unsigned u = __builtin_bit_cast(unsigned, x);
  const auto r = XOR == 16 ? __builtin_amdgcn_permlane16_swap(u, u, false, false)
                           : __builtin_amdgcn_permlane32_swap(u, u, false, false);
  const unsigned a = r[0], b = r[1];
  return fmaxf(__builtin_bit_cast(float, a), __builtin_bit_cast(float, b));
}

// Constant 100 MHz wall clock, independent of the shader clock.
__device__ __forceinline__ uint64_t wall_ticks() { return __builtin_amdgcn_s_memrealtime(); }

}  // namespace tdist
