// Low-level device helpers for user kernels: the gfx950 forms of the reference's `language_extra`
// (reference: backends/nvidia/language/cuda/language_extra.py:109-929 - PTX snippets for packed moves, vector
// loads / stores, multimem stores and load-reduces, thread / lane ids, scoped ld / st / atomics, red_release,
// arrive_inc, atomic_add_per_warp, wait_eq, the shuffle family, ballot and CAS).
//
// Differences that come from the hardware, not from taste:
//   * a wavefront has 64 lanes: masks are 64-bit, `laneid()` runs to 63, "per warp" means per wave;
//   * there is no multicast address space on xGMI: `multimem_st_*` is a loop of write-through stores to the
//     same symmetric offset on every rank, `multimem_ld_reduce_*` a loop of loads over the ranks that sums in
//     fp32 - both take the SymmCtx that gives the peers' base addresses;
//   * scopes are "workgroup" / "gpu" (agent) / "sys" (system) and select the sc0 / sc1 cache-policy bits of
//     the access plus the buffer_wbl2 / buffer_inv that release / acquire need at that scope.
// Everything is an inline function; `#include "tdist/language_extra.hpp"` is all a user kernel needs (see
// tools/user_kernel.py for building and launching one).
#pragma once
#include "tdist/atomics.hpp"
#include "tdist/cache_policy.hpp"
#include "tdist/primitives.hpp"

namespace tdist {
namespace ext {

enum Scope : int { WORKGROUP = 0, GPU = 1, SYS = 2 };
enum Semantic : int { RELAXED = 0, ACQUIRE = 1, RELEASE = 2, ACQ_REL = 3 };

// ---- packed moves ---------------------------------------------------------------------------------------
// two 16-bit values -> one 32-bit register (low half = val0)
__device__ __forceinline__ uint32_t pack_b32_v2(uint16_t val0, uint16_t val1) { return (uint32_t)val0 | ((uint32_t)val1 << 16); }

// ---- barriers and fences ----------------------------------------------------------------------------------
__device__ __forceinline__ void __syncthreads_() { tdist::syncthreads(); }

template <int SCOPE = GPU>
__device__ __forceinline__ void fence() {
  if constexpr (SCOPE == WORKGROUP) {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
  } else if constexpr (SCOPE == GPU) {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "agent");
  } else {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "");
  }
}

// ---- vector loads / stores (one global_load_dwordx4 / dwordx2 per lane) -----------------------------------
__device__ __forceinline__ uint4 load_v4_u32(const void* p) { return *(const uint4*)p; }
__device__ __forceinline__ uint4 load_v4_b32(const void* p) { return *(const uint4*)p; }
__device__ __forceinline__ int4 load_v4_s32(const void* p) { return *(const int4*)p; }
__device__ __forceinline__ ulonglong2 load_v2_b64(const void* p) { return *(const ulonglong2*)p; }
__device__ __forceinline__ void st_v4_u32(void* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) { *(uint4*)p = uint4{a, b, c, d}; }
__device__ __forceinline__ void st_v4_b32(void* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) { *(uint4*)p = uint4{a, b, c, d}; }
__device__ __forceinline__ void st_v2_u32(void* p, uint32_t a, uint32_t b) { *(uint2*)p = uint2{a, b}; }

// ---- "multimem": the same symmetric offset on every rank ----------------------------------------------------
// Stores leave this GPU's caches at once (sc0 sc1); drain_stores() + a flag publish them, no release fence.
__device__ __forceinline__ void multimem_st_v4(const SymmCtx& ctx, void* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
  for (int r = 0; r < ctx.num_ranks; ++r) st_wt128_sys(symm_at(ctx, (char*)p, r), uint4{a, b, c, d});
}
__device__ __forceinline__ void multimem_st_v2(const SymmCtx& ctx, void* p, uint32_t a, uint32_t b) {
  for (int r = 0; r < ctx.num_ranks; ++r) st_wt64_sys(symm_at(ctx, (char*)p, r), uint2{a, b});
}
__device__ __forceinline__ void multimem_st_b64(const SymmCtx& ctx, void* p, uint64_t v) {
  multimem_st_v2(ctx, p, (uint32_t)v, (uint32_t)(v >> 32));
}
__device__ __forceinline__ void multimem_st_b32(const SymmCtx& ctx, void* p, uint32_t v) {
  for (int r = 0; r < ctx.num_ranks; ++r) store_relaxed_system((uint32_t*)symm_at(ctx, (char*)p, r), v);
}
// predicated form: lanes with mask == 0 do nothing
__device__ __forceinline__ void multimem_st_p_b32(const SymmCtx& ctx, void* p, uint32_t v, bool mask) {
  if (mask) multimem_st_b32(ctx, p, v);
}

// Load 16 bytes from the same offset on every rank and add them up: eight bf16 values (fp32 accumulation,
// rounded once) or four fp32 values.
__device__ __forceinline__ uint4 multimem_ld_reduce_v4_bf16(const SymmCtx& ctx, const void* p) {
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < ctx.num_ranks; ++r) {
    const uint4 v = *(const uint4*)symm_at(ctx, (const char*)p, r);
    const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc[2 * i] += __builtin_bit_cast(float, x[i] << 16);
      acc[2 * i + 1] += __builtin_bit_cast(float, x[i] & 0xffff0000u);
    }
  }
  uint32_t o[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    const b2 v = {(__bf16)acc[2 * i], (__bf16)acc[2 * i + 1]};
    o[i] = __builtin_bit_cast(uint32_t, v);
  }
  return uint4{o[0], o[1], o[2], o[3]};
}
__device__ __forceinline__ float4 multimem_ld_reduce_v4_f32(const SymmCtx& ctx, const void* p) {
  float4 acc = {0.f, 0.f, 0.f, 0.f};
  for (int r = 0; r < ctx.num_ranks; ++r) {
    const float4 v = *(const float4*)symm_at(ctx, (const char*)p, r);
    acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
  }
  return acc;
}
__device__ __forceinline__ uint4 multimem_ld_reduce_p_v4_bf16(const SymmCtx& ctx, const void* p, bool mask) {
  return mask ? multimem_ld_reduce_v4_bf16(ctx, p) : uint4{0, 0, 0, 0};
}

// ---- thread / lane ids ------------------------------------------------------------------------------------
__device__ __forceinline__ int tid(int axis) { return tdist::thread_idx(axis); }
__device__ __forceinline__ int ntid(int axis) { return axis == 0 ? blockDim.x : (axis == 1 ? blockDim.y : blockDim.z); }
__device__ __forceinline__ int laneid() { return tdist::lane_id(); }

// ---- scoped loads / stores ----------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T ld(const T* p, int scope = GPU, int semantic = RELAXED) {
  return load_scoped(p, scope, semantic == ACQUIRE ? SEM_ACQUIRE : SEM_RELAXED);
}
__device__ __forceinline__ uint32_t ld_b32(const void* p) { return *(const volatile uint32_t*)p; }
template <typename T>
__device__ __forceinline__ T ld_acquire(const T* p, int scope = GPU) { return ld(p, scope, ACQUIRE); }

template <typename T>
__device__ __forceinline__ void st(T* p, T v, int scope = GPU, int semantic = RELAXED) {
  if (semantic == RELEASE) {
    if (scope == SYS) store_release_system(p, v);
    else if (scope == GPU) store_release_agent(p, v);
    else store_release_workgroup(p, v);
  } else {
    if (scope == SYS) store_relaxed_system(p, v);
    else if (scope == GPU) store_relaxed_agent(p, v);
    else store_relaxed_workgroup(p, v);
  }
}
__device__ __forceinline__ void st_b32(void* p, uint32_t v) { *(volatile uint32_t*)p = v; }

// ---- atomics ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T atomic_add(T* p, T v, int scope = GPU, int semantic = RELAXED) {
  const bool sys = scope == SYS;
  switch (semantic) {
    case ACQUIRE: return sys ? atom_add_acquire_system(p, v) : atom_add_acquire_agent(p, v);
    case RELEASE: return sys ? atom_add_release_system(p, v) : atom_add_release_agent(p, v);
    case ACQ_REL: return sys ? atom_add_acqrel_system(p, v) : atom_add_acqrel_agent(p, v);
    default: return sys ? atom_add_relaxed_system(p, v) : atom_add_relaxed_agent(p, v);
  }
}

template <typename T>
__device__ __forceinline__ T atomic_cas(T* p, T expected, T desired, int scope = GPU, int semantic = RELAXED) {
  const bool sys = scope == SYS;
  switch (semantic) {
    case ACQUIRE: return sys ? atom_cas_acquire_relaxed_system(p, expected, desired) : atom_cas_acquire_relaxed_agent(p, expected, desired);
    case RELEASE: return sys ? atom_cas_release_relaxed_system(p, expected, desired) : atom_cas_release_relaxed_agent(p, expected, desired);
    case ACQ_REL: return sys ? atom_cas_acqrel_relaxed_system(p, expected, desired) : atom_cas_acqrel_relaxed_agent(p, expected, desired);
    default: return sys ? atom_cas_relaxed_relaxed_system(p, expected, desired) : atom_cas_relaxed_relaxed_agent(p, expected, desired);
  }
}

// add with release semantics, no return value (a reduction: the memory system need not send the old value back)
__device__ __forceinline__ void red_release(int* barrier_ptr, int value, int scope = GPU) {
  if (scope == SYS) red_add_release_system(barrier_ptr, value);
  else red_add_release_agent(barrier_ptr, value);
}

// Thread `thread_idx` of the workgroup adds `value` with release semantics after a workgroup barrier: every
// thread's earlier writes are ordered before the arrival becomes visible.
__device__ __forceinline__ void arrive_inc(int* barrier_ptr, int thread_idx, int value, int scope = GPU) {
  tdist::syncthreads();
  if ((int)threadIdx.x == thread_idx) red_release(barrier_ptr, value, scope);
}

// One atomic per wave: the first active lane adds `value` on behalf of the wave and the old value is handed to
// every lane.
__device__ __forceinline__ int atomic_add_per_warp(int* barrier_ptr, int value, int scope = GPU, int semantic = RELAXED) {
  const unsigned long long active = __ballot(1);
  const int leader = __builtin_ctzll(active);
  int old = 0;
  if (laneid() == leader) old = atomic_add(barrier_ptr, value, scope, semantic);
  return __shfl(old, leader);
}

// Thread `thread_idx` spins until *barrier_ptr == value (bounded: see primitives.hpp), then the workgroup
// meets; returns false if the spin budget ran out.
__device__ __forceinline__ bool wait_eq(const int* barrier_ptr, int thread_idx, int value, int scope = GPU) {
  bool ok = true;
  if ((int)threadIdx.x == thread_idx) {
    ok = spin_until_eq(barrier_ptr, value, scope == SYS ? SCOPE_SYS : SCOPE_GPU);
    acquire_fence(scope == SYS ? SCOPE_SYS : SCOPE_GPU);
  }
  tdist::syncthreads();
  return ok;
}

// ---- shuffles over the 64-lane wave (masks are accepted for source compatibility and ignored: all lanes of a
// wave execute in lockstep) --------------------------------------------------------------------------------
__device__ __forceinline__ int __shfl_sync_i32(unsigned long long, int value, int lane) { return __shfl(value, lane); }
__device__ __forceinline__ int __shfl_up_sync_i32(unsigned long long, int value, int delta) { return __shfl_up(value, delta); }
__device__ __forceinline__ int __shfl_down_sync_i32(unsigned long long, int value, int delta) { return __shfl_down(value, delta); }
__device__ __forceinline__ int __shfl_xor_sync_i32(unsigned long long, int value, int delta) { return __shfl_xor(value, delta); }
__device__ __forceinline__ unsigned long long __ballot_sync(unsigned long long, bool pred) { return __ballot(pred); }

// ---- gfx950 extras the reference has no counterpart for -----------------------------------------------------
// max over lane pairs (l, l ^ 16) / (l, l ^ 32) without touching LDS (v_permlane16_swap / v_permlane32_swap):
// xor_max<16>(x), xor_max<32>(x); and the 100 MHz wall clock wall_ticks()
using tdist::wall_ticks;
using tdist::xor_max;

}  // namespace ext
}  // namespace tdist
