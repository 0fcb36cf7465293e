// Cross-device coherence probe for the symmetric heap.
//
// Every fused op of this library hands data from one GPU to another the same way: the producer
// stores the payload into the consumer's arena with write-through system-scope stores
// (st_wt128_sys), drains them, and raises an epoch-valued flag in the consumer's signal arena; the
// consumer polls the flag, issues ONE system-scope acquire and then reads the payload with plain
// loads or direct-to-LDS loads.  Buffers are reused every second call, so when a payload arrives the
// consumer's caches (the vector L1 of every CU, the L2 of each of the eight XCDs) can still hold the
// lines of the call before last.  Whether the acquire is enough to keep those lines from being
// read depends on how the arena is mapped (coarse-grained hipMalloc memory, fine-grained or uncached
// memory), on the driver and on the fabric - so it is measured, once, when the heap is created, and
// the heap falls back to the next memory mode if a single stale or torn word is seen.
//
// One launch = one round on every rank (all ranks run the same kernel):
//   1. WARM   every workgroup reads every peer slot of this round's parity buffer with plain loads, so
//             the old lines sit in its CU's L1 and its XCD's L2;
//   2. READY  the last workgroup to finish the warm-up tells every peer "my old lines are cached";
//   3. WRITE  every workgroup waits for the peer's READY, then stores its slice of a round-specific
//             pattern into that peer's slot [this rank]; the last slice raises DATA on the peer;
//   4. CHECK  every workgroup waits for each peer's DATA (system-scope acquire) and compares the whole
//             slot twice: through plain global loads and through global_load_lds + ds_read;
//   5. ATOM   one lane adds 1 to a counter word on every peer (system-scope atomic over the fabric).
// All waits are bounded (2 s); nothing in here can hang a queue.
//
// Role in the reference: none - it checks data only with --check loops at the op level
// (reference: python/triton_dist/test/amd/test_ag_gemm_intra_node.py:229-275).
#include <stdio.h>

#include "tdist/cache_policy.hpp"
#include "tdist/common.hpp"
#include "tdist/primitives.hpp"

using namespace tdist;

namespace {

constexpr int PROBE_THREADS = 256;
constexpr uint64_t PROBE_BUDGET = 200ull * 1000ull * 1000ull;   // 2 s of 100 MHz ticks

struct ProbeParams {
  SymmCtx ctx;
  uint32_t* buf;     // symmetric, data arena: [2 parities][w sources][words]
  int* ready;        // symmetric, signal arena: [2][w]
  int* data;         // symmetric, signal arena: [2][w]
  int* counter;      // symmetric, signal arena: [1]
  int* local;        // device-local: [0] plain mismatches [1] LDS-DMA mismatches [2] timeouts [3] warm arrivals
                     //               [4] first bad word index [5] sink [8 .. 8 + w) slices stored per peer
  int words;         // per slot, multiple of 1024
  int round;
};

__device__ __forceinline__ uint32_t pattern(int src, int round, int i) {
  uint32_t x = (uint32_t)i * 0x9E3779B1u + (uint32_t)round * 0x85EBCA6Bu + (uint32_t)src * 0xC2B2AE35u + 0x27D4EB2Fu;
  x ^= x >> 15;
  x *= 0x2C1B3C6Du;
  x ^= x >> 12;
  return x;
}

// thread 0 polls, everybody leaves together; returns false on timeout
__device__ __forceinline__ bool block_wait_eq(const int* flag, int value, int* local, volatile int* s_ok) {
  if (threadIdx.x == 0) {
    const bool ok = spin_until_eq(flag, value, SCOPE_SYS, PROBE_BUDGET);
    if (!ok) atomicAdd(local + 2, 1);
    *s_ok = ok ? 1 : 0;
  }
  __syncthreads();
  const bool ok = *s_ok != 0;
  __syncthreads();
  return ok;
}

__global__ __launch_bounds__(PROBE_THREADS) void coherence_probe_kernel(ProbeParams p) {
  __shared__ __attribute__((aligned(16))) uint4 stage[PROBE_THREADS];
  __shared__ int s_ok;
  const int w = p.ctx.num_ranks, me = p.ctx.rank;
  const int G = gridDim.x, b = blockIdx.x, tid = threadIdx.x;
  const int par = p.round & 1, epoch = p.round + 1;
  uint32_t* mine = p.buf + (long long)par * w * p.words;
  int* ready = p.ready + par * w;
  int* data = p.data + par * w;

  // ---- 1. WARM: old lines of every peer slot into this CU's L1 and this XCD's L2 ---------------------
  uint32_t sink = 0;
  for (int s = 0; s < w; ++s) {
    if (s == me) continue;
    const uint4* q = (const uint4*)(mine + (long long)s * p.words);
    for (int i = tid; i < p.words / 4; i += PROBE_THREADS) {
      const uint4 v = q[i];
      sink += v.x ^ v.y ^ v.z ^ v.w;
    }
  }
  if (sink == 0x12345678u) p.local[5] = (int)sink;     // keeps the loads alive
  // ---- 2. READY -------------------------------------------------------------------------------------
  __syncthreads();
  if (tid == 0) {
    const int prev = atom_add_relaxed_agent(p.local + 3, 1);
    if (prev == G - 1) {
      store_relaxed_agent(p.local + 3, 0);
      for (int j = 1; j < w; ++j) {
        const int peer = (me + j) % w;
        store_relaxed_system(symm_at(p.ctx, ready, peer) + me, epoch);
      }
    }
  }
  // ---- 3. WRITE: slice b of my pattern into slot [me] of every peer ---------------------------------
  const int per = ((p.words / 4 + G - 1) / G);               // 16-byte pieces per workgroup
  const int i0 = min(b * per, p.words / 4), i1 = min(i0 + per, p.words / 4);
  for (int j = 1; j < w; ++j) {
    const int peer = (me + j) % w;
    if (!block_wait_eq(ready + peer, epoch, p.local, &s_ok)) continue;
    uint32_t* dst = symm_at(p.ctx, p.buf, peer) + ((long long)par * w + me) * p.words;
    for (int i = i0 + tid; i < i1; i += PROBE_THREADS) {
      uint4 v;
      v.x = pattern(me, p.round, 4 * i);
      v.y = pattern(me, p.round, 4 * i + 1);
      v.z = pattern(me, p.round, 4 * i + 2);
      v.w = pattern(me, p.round, 4 * i + 3);
      st_wt128_sys(dst + 4 * (long long)i, v);
    }
    drain_stores();
    __syncthreads();
    if (tid == 0) {
      const int prev = atom_add_relaxed_agent(p.local + 8 + peer, 1);
      if (prev == G - 1) {
        store_relaxed_agent(p.local + 8 + peer, 0);
        store_relaxed_system(symm_at(p.ctx, data, peer) + me, epoch);
      }
    }
  }
  // ---- 4. CHECK: every workgroup verifies every whole slot, by plain loads and by LDS-DMA ------------
  for (int j = 1; j < w; ++j) {
    const int s = (me + w - j) % w;
    if (!block_wait_eq(data + s, epoch, p.local, &s_ok)) continue;
    if (tid < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const uint32_t* slot = mine + (long long)s * p.words;
    int bad = 0, first_bad = -1;
    for (int i = tid; i < p.words / 4; i += PROBE_THREADS) {
      const uint4 v = ((const uint4*)slot)[i];
      const bool ok = v.x == pattern(s, p.round, 4 * i) && v.y == pattern(s, p.round, 4 * i + 1) &&
                      v.z == pattern(s, p.round, 4 * i + 2) && v.w == pattern(s, p.round, 4 * i + 3);
      if (!ok) {
        ++bad;
        if (first_bad < 0) first_bad = 4 * i;
      }
    }
    if (bad) {
      atomicAdd(p.local + 0, bad);
      atomicMax(p.local + 4, first_bad);
    }
    bad = 0;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int base = 0; base < p.words / 4; base += PROBE_THREADS) {
      const int i = base + tid;                               // words is a multiple of 1024: always in range
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const uint4*)slot + i),
                                       (__attribute__((address_space(3))) void*)(stage + wave * 64), 16, 0, 0);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const uint4 v = stage[tid];
      const bool ok = v.x == pattern(s, p.round, 4 * i) && v.y == pattern(s, p.round, 4 * i + 1) &&
                      v.z == pattern(s, p.round, 4 * i + 2) && v.w == pattern(s, p.round, 4 * i + 3);
      if (!ok) ++bad;
      __syncthreads();
    }
    if (bad) atomicAdd(p.local + 1, bad);
  }
  // ---- 5. ATOM ----------------------------------------------------------------------------------------
  if (b == 0 && tid == 0) {
    for (int j = 1; j < w; ++j) (void)atom_add_relaxed_system(symm_at(p.ctx, p.counter, (me + j) % w), 1);
  }
}

}  // namespace

// One probe round on `stream`.  `buf` holds 2 * w * words uint32 in the data arena, `ready` / `data`
// 2 * w int32 each and `counter` one int32 in the signal arena (all symmetric, zero-initialised);
// `local` is >= 8 + w device-local int32, zero-initialised.
TDIST_API int tdist_coherence_probe(const SymmCtx* ctx, void* buf, int* ready, int* data, int* counter, int* local,
                                    int words, int round, int num_ctas, void* stream) {
  TDIST_REQUIRE(ctx->num_ranks >= 2 && ctx->num_ranks <= TDIST_MAX_RANKS, "the probe needs 2..16 ranks");
  TDIST_REQUIRE(words > 0 && words % 1024 == 0, "words must be a positive multiple of 1024");
  TDIST_REQUIRE(num_ctas >= 1 && num_ctas <= 256, "num_ctas must be in [1, 256] (all workgroups must be resident)");
  TDIST_REQUIRE((((uintptr_t)buf) & 15) == 0, "buffer must be 16-byte aligned");
  ProbeParams p;
  p.ctx = *ctx;
  p.buf = (uint32_t*)buf;
  p.ready = ready;
  p.data = data;
  p.counter = counter;
  p.local = local;
  p.words = words;
  p.round = round;
  hipLaunchKernelGGL(coherence_probe_kernel, dim3(num_ctas), dim3(PROBE_THREADS), 0, (hipStream_t)stream, p);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
