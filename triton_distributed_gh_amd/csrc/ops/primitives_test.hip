// Small kernels that exercise the device primitives one at a time; used by the test-suite and by
// the Python `language` facade demos.
//
// Scenarios follow the reference's primitive tests
// (reference: python/triton_dist/test/amd/test_distributed-notify-wait.py:39-212 ring queue,
//  test_wait_and_notify.py, test_barrier_all.py, test_libdevice.py, test_rocshmem_api.py).
#include "tdist/common.hpp"
#include "tdist/grid_barrier.hpp"
#include "tdist/language_extra.hpp"
#include "tdist/primitives.hpp"
#include "tdist/shmem_device.hpp"

using namespace tdist;

// status[0] = 1 when the flag was seen, 2 on timeout; status[1] = ticks waited
__global__ void wait_flag_kernel(const int* flag, int value, int scope, long long* status) {
  const uint64_t t0 = wall_ticks();
  const int failed = wait(flag, 1, scope, SEM_ACQUIRE, value);
  if (threadIdx.x == 0) {
    status[0] = failed ? 2 : 1;
    status[1] = (long long)(wall_ticks() - t0);
  }
}

__global__ void notify_kernel(SymmCtx ctx, int* flag, int peer, int value, int sig_op, int comm_scope) {
  if (threadIdx.x == 0) notify(ctx, flag, peer, value, sig_op, comm_scope);
}

__global__ void barrier_all_kernel(SymmCtx ctx, int* slots, int* ok) {
  const bool r = shmem::barrier_all_block(ctx, slots);
  if (threadIdx.x == 0 && ok) *ok = r ? 1 : 0;
}

// Ring queue (protocol P1): `num_items` items of `item_elems` ints are pushed to `peer`'s queue
// by producer workgroups (blockIdx < num_producers) and popped from the local queue by consumer
// workgroups.  signal slot value 2r = free for round r, 2r+1 = full in round r.
__global__ void ring_queue_kernel(SymmCtx ctx, int* queue, int* signal, const int* input, int* output, int peer,
                                  int queue_size, int item_elems, int num_items, int num_producers,
                                  int num_consumers, int* err) {
  const int bid = blockIdx.x;
  if (bid < num_producers) {
    for (int i = bid; i < num_items; i += num_producers) {
      const int slot = i % queue_size, round = i / queue_size;
      int* rsig = symm_at(ctx, signal, peer) + slot;
      wait(rsig, 1, SCOPE_SYS, SEM_ACQUIRE, 2 * round, err);
      int* rq = symm_at(ctx, queue, peer) + (long long)slot * item_elems;
      for (int e = threadIdx.x; e < item_elems; e += blockDim.x) rq[e] = input[(long long)i * item_elems + e];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) notify(ctx, signal + slot, peer, 2 * round + 1, SIGNAL_SET, COMM_INTRA_NODE);
      __syncthreads();
    }
  } else {
    const int cid = bid - num_producers;
    for (int i = cid; i < num_items; i += num_consumers) {
      const int slot = i % queue_size, round = i / queue_size;
      wait(signal + slot, 1, SCOPE_SYS, SEM_ACQUIRE, 2 * round + 1, err);
      const int* q = queue + (long long)slot * item_elems;
      for (int e = threadIdx.x; e < item_elems; e += blockDim.x) output[(long long)i * item_elems + e] = q[e];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) notify(ctx, signal + slot, ctx.rank, 2 * round + 2, SIGNAL_SET, COMM_INTRA_NODE);
      __syncthreads();
    }
  }
}

// putmem / getmem / signal smoke: every rank puts `n` ints into slot [rank] of each peer's buffer
// with putmem_signal_block, waits for all signals, then reads a peer's slot back with getmem_block.
__global__ void shmem_api_kernel(SymmCtx ctx, int* buf, uint64_t* sig, const int* src, int* readback, int n,
                                 uint64_t sig_value) {
  const int peer = blockIdx.x;
  if (peer >= ctx.num_ranks) return;
  shmem::putmem_signal_block(ctx, buf + (long long)ctx.rank * n, src, (size_t)n * sizeof(int), sig + ctx.rank,
                             sig_value, shmem::SIGNAL_SET_OP, peer);
  if (threadIdx.x == 0) shmem::signal_wait_until(sig + peer, shmem::CMP_EQ, sig_value);
  __syncthreads();
  // read peer's copy of OUR slot back (it was written by us)
  shmem::getmem_block(ctx, readback + (long long)peer * n, buf + (long long)ctx.rank * n, (size_t)n * sizeof(int), peer);
}

// libdevice-style atomics on one word; results land in out[0..].
__global__ void atomics_kernel(int* word, int* out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    store_release_agent(word, 5);
    out[0] = load_acquire_agent(word);
    out[1] = atom_add_acqrel_agent(word, 3);          // returns 5, word = 8
    out[2] = atom_cas_acquire_relaxed_agent(word, 8, 11);  // returns 8, word = 11
    out[3] = atom_cas_relaxed_relaxed_system(word, 0, 1);  // fails, returns 11
    red_add_release_system(word, 1);                  // word = 12
    out[4] = load_relaxed_system(word);
    store_relaxed_workgroup(word, 20);
    out[5] = load_acquire_workgroup(word);
  }
}

// language_extra helpers, one workgroup of 128 threads (two waves) per rank.  `buf` is a symmetric
// [num_ranks + 2][128] uint4 area: slot r holds rank r's vector, slot num_ranks the "multimem" broadcast target
// and slot num_ranks + 1 bf16 values for the load-reduce.  `sync` is a symmetric int[4]; out[0] / out[1]: all
// checks of wave 0 / wave 1 passed, out[2]: the arrivals came in, out[3]: one bit per failed check.
__global__ void language_extra_kernel(SymmCtx ctx, uint4* buf, int* sync, int* out) {
  using namespace tdist::ext;
  const int t = tid(0), w = ctx.num_ranks, me = ctx.rank;
  // packed move, vector store / load round trip
  const uint32_t packed = pack_b32_v2((uint16_t)t, (uint16_t)(t + 1));
  st_v4_u32(buf + me * 128 + t, packed, (uint32_t)t, (uint32_t)me, 7u);
  const uint4 back = load_v4_u32(buf + me * 128 + t);
  bool ok = true; { const bool c_ = (back.x == ((uint32_t)t | ((uint32_t)(t + 1) << 16)) && back.y == (uint32_t)t && back.z == (uint32_t)me && back.w == 7u); if (!c_) atomicOr(out + 3, 1); ok &= c_; }
  // shuffles and ballot over the 64-lane wave
  const int l = laneid();
  { const bool c_ = (__shfl_sync_i32(~0ull, l, 5) == 5 && __shfl_xor_sync_i32(~0ull, l, 32) == (l ^ 32)); if (!c_) atomicOr(out + 3, 1 << 1); ok &= c_; }
  { const bool c_ = (__shfl_down_sync_i32(~0ull, l, 3) == (l + 3 < 64 ? l + 3 : l) && __shfl_up_sync_i32(~0ull, l, 3) == (l >= 3 ? l - 3 : l)); if (!c_) atomicOr(out + 3, 1 << 2); ok &= c_; }
  { const bool c_ = (__ballot_sync(~0ull, (l & 1) != 0) == 0xaaaaaaaaaaaaaaaaull); if (!c_) atomicOr(out + 3, 1 << 3); ok &= c_; }
  { const bool c_ = (ntid(0) == 128 && xor_max<32>((float)l) == (float)(l | 32) && xor_max<16>((float)l) == (float)(l | 16)); if (!c_) atomicOr(out + 3, 1 << 4); ok &= c_; }
  // one atomic per wave: two waves -> the counter ends at 2, every lane of a wave sees the same old value
  __shared__ int s_cnt;
  if (t == 0) s_cnt = 0;
  __syncthreads_();
  const int old = atomic_add_per_warp(&s_cnt, 1, WORKGROUP, RELAXED);
  { const bool c_ = (old == __shfl_sync_i32(~0ull, old, 0)); if (!c_) atomicOr(out + 3, 1 << 5); ok &= c_; }
  __syncthreads_();
  { const bool c_ = (s_cnt == 2); if (!c_) atomicOr(out + 3, 1 << 6); ok &= c_; }
  // multimem store: the same slot on every rank gets this rank's lane values (rank-tagged), then everybody
  // arrives at every rank's counter and waits for all arrivals
  multimem_st_v4(ctx, buf + (w + 2 + me) * 128 + t, (uint32_t)(me + 1), (uint32_t)t, 0u, 0u);
  drain_stores();
  for (int r = 0; r < w; ++r) arrive_inc(symm_at(ctx, sync, r), 0, 1, SYS);
  const bool arrived = wait_eq(sync, 0, w, SYS);
  // every rank's tagged vector must be here now
  for (int r = 0; r < w; ++r) {
    const uint4 v = load_v4_b32(buf + (w + 2 + r) * 128 + t);
    { const bool c_ = (v.x == (uint32_t)(r + 1) && v.y == (uint32_t)t); if (!c_) atomicOr(out + 3, 1 << 7); ok &= c_; }
  }
  // CAS / scoped ld / st / red_release on a private word
  if (t == 0) {
    int* word = sync + 1;
    st(word, 5, GPU, RELEASE);
    { const bool c_ = (ld_acquire(word, GPU) == 5 && atomic_cas(word, 5, 9, SYS, ACQ_REL) == 5 && atomic_cas(word, 5, 1, GPU, RELAXED) == 9); if (!c_) atomicOr(out + 3, 1 << 8); ok &= c_; }
    red_release(word, 3, GPU);
    { const bool c_ = (atomic_add(word, 0, GPU, ACQUIRE) == 12); if (!c_) atomicOr(out + 3, 1 << 9); ok &= c_; }
    st_b32(word, 77u);
    { const bool c_ = (ld_b32(word) == 77u); if (!c_) atomicOr(out + 3, 1 << 10); ok &= c_; }
  }
  const unsigned long long good = __ballot_sync(~0ull, ok);
  if (l == 0) out[t >> 6] = good == ~0ull ? 1 : 0;
  if (t == 0) out[2] = arrived ? 1 : 0;
}

// Load-reduce on its own (after a host-side barrier: every rank's slot is final): sums eight bf16 per lane over ranks.
__global__ void language_extra_ld_reduce_kernel(SymmCtx ctx, const uint4* src, uint4* dst) {
  dst[threadIdx.x] = tdist::ext::multimem_ld_reduce_v4_bf16(ctx, src + threadIdx.x);
}

// Grid barrier check: in round r every workgroup publishes r+1 in its own word, crosses the
// barrier, and must then read r+1 from EVERY workgroup's word; a second barrier keeps round r+1's
// stores behind round r's reads.  bad[0] counts stale reads, bad[1] timeouts.
__global__ void grid_barrier_kernel(unsigned* state, int* words, int rounds, int* bad) {
  for (int r = 0; r < rounds; ++r) {
    if (threadIdx.x == 0) words[blockIdx.x] = r + 1;
    if (!barrier_on_this_grid(state)) {
      if (threadIdx.x == 0) atomicAdd(bad + 1, 1);
      return;
    }
    int stale = 0;
    for (int b = threadIdx.x; b < (int)gridDim.x; b += blockDim.x) stale += words[b] != r + 1;
    if (stale) atomicAdd(bad, stale);
    if (!barrier_on_this_grid(state)) {
      if (threadIdx.x == 0) atomicAdd(bad + 1, 1);
      return;
    }
  }
}

TDIST_API int tdist_test_grid_barrier(unsigned* state, int* words, int rounds, int* bad, int grid, void* stream) {
  TDIST_REQUIRE(grid >= 1 && grid <= 2048, "grid must fit the chip (all workgroups resident)");
  hipLaunchKernelGGL(grid_barrier_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, state, words, rounds, bad);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

TDIST_API int tdist_test_wait_flag(const int* flag, int value, int scope, long long* status, void* stream) {
  hipLaunchKernelGGL(wait_flag_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, flag, value, scope, status);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
TDIST_API int tdist_test_notify(const SymmCtx* ctx, int* flag, int peer, int value, int sig_op, int comm_scope,
                                void* stream) {
  hipLaunchKernelGGL(notify_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *ctx, flag, peer, value, sig_op,
                     comm_scope);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
TDIST_API int tdist_barrier_all(const SymmCtx* ctx, int* slots, int* ok, void* stream) {
  hipLaunchKernelGGL(barrier_all_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, *ctx, slots, ok);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
TDIST_API int tdist_test_ring_queue(const SymmCtx* ctx, int* queue, int* signal, const int* input, int* output,
                                    int peer, int queue_size, int item_elems, int num_items, int num_producers,
                                    int num_consumers, int* err, void* stream) {
  hipLaunchKernelGGL(ring_queue_kernel, dim3(num_producers + num_consumers), dim3(256), 0, (hipStream_t)stream, *ctx,
                     queue, signal, input, output, peer, queue_size, item_elems, num_items, num_producers,
                     num_consumers, err);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
TDIST_API int tdist_test_shmem_api(const SymmCtx* ctx, int* buf, uint64_t* sig, const int* src, int* readback, int n,
                                   unsigned long long sig_value, void* stream) {
  hipLaunchKernelGGL(shmem_api_kernel, dim3(ctx->num_ranks), dim3(256), 0, (hipStream_t)stream, *ctx, buf, sig, src,
                     readback, n, (uint64_t)sig_value);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
TDIST_API int tdist_test_atomics(int* word, int* out, void* stream) {
  hipLaunchKernelGGL(atomics_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, word, out);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

// buf: symmetric uint4 [(2 * num_ranks + 2) * 128]; sync: symmetric int[4], zero; out: int[4].
TDIST_API int tdist_test_language_extra(const SymmCtx* ctx, void* buf, int* sync, int* out, void* stream) {
  hipLaunchKernelGGL(language_extra_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, *ctx, (uint4*)buf, sync, out);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
TDIST_API int tdist_test_ld_reduce(const SymmCtx* ctx, const void* src, void* dst, void* stream) {
  hipLaunchKernelGGL(language_extra_ld_reduce_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, *ctx, (const uint4*)src,
                     (uint4*)dst);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
