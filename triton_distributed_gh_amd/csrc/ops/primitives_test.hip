// Small kernels that exercise the device primitives one at a time; used by the test-suite and by
// the Python `language` facade demos.
//
// Scenarios follow the reference's primitive tests
// (reference: python/triton_dist/test/amd/test_distributed-notify-wait.py:39-212 ring queue,
//  test_wait_and_notify.py, test_barrier_all.py, test_libdevice.py, test_rocshmem_api.py).
#include "tdist/common.hpp"
#include "tdist/grid_barrier.hpp"
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
