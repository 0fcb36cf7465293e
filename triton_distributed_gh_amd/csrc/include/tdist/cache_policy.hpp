// Write-through 16-byte stores for data that another workgroup (or another GPU) reads inside the
// same launch.
//
// gfx950 has eight XCDs with private, mutually non-coherent L2s.  A plain store leaves a dirty line
// in the storing XCD's L2 and only a release fence (buffer_wbl2, which walks the whole L2: ~1.7 us
// clean, several us once a 128 KiB output tile is dirty) pushes it out.  A store carrying the sc1
// (agent) or sc0 sc1 (system) cache-policy bits leaves the L2 at once, so publishing needs no
// release fence: every storing wave drains its stores (s_waitcnt vmcnt(0)), the workgroup
// barriers, and one lane raises the flag or counter.  The consumer side is unchanged: relaxed poll,
// ONE acquire fence of the matching scope, then plain loads.
//
// The trailing s_nop keeps the compiler's next instruction from overwriting the data registers
// before the store has read them.  The compiler does not count these stores in vmcnt, so callers
// must drain with an explicit s_waitcnt vmcnt(0) before signalling (they do anyway).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tdist {

typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;

// system scope (sc0 sc1): payload bound for a peer GPU's memory or read by a peer GPU
__device__ __forceinline__ void st_wt128_sys(void* p, const uint4& v) {
  const u32x4_t d = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
}

// agent scope (sc1): payload read by other workgroups of this GPU
__device__ __forceinline__ void st_wt128_agent(void* p, const uint4& v) {
  const u32x4_t d = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
}

__device__ __forceinline__ void st_wt128_agent(void* p, const __attribute__((ext_vector_type(4))) float& v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}

// 8-byte form (four 16-bit values: one lane's share of a 16x16 accumulator tile)
__device__ __forceinline__ void st_wt64_sys(void* p, const uint2& v) {
  typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
  const u32x2_t d = {v.x, v.y};
  asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1\n\ts_nop 1" ::"v"(p), "v"(d) : "memory");
}

// Every storing wave calls this after its last write-through store and before the workgroup
// barrier that precedes the flag / counter update.
//
// TDIST_SAFE_PUBLISH (the second build of the library, libtdist_hip_safe.so): the zero-fence publish above rests on
// the sc0 / sc1 policy bits pushing every line out of the L2 on this GPU, fabric and firmware.  The safe build pays
// for not relying on it: the drain is followed by a system-scope release fence (write-back of the whole L2), so the
// flag store that follows is ordered after the payload by the memory model alone.  Consumers already poll relaxed
// and issue a system-scope acquire (L1 / L2 invalidate) before their first payload load.
#ifndef TDIST_SAFE_PUBLISH
#define TDIST_SAFE_PUBLISH 0
#endif
__device__ __forceinline__ void drain_stores() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#if TDIST_SAFE_PUBLISH
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
#endif
}

}  // namespace tdist
