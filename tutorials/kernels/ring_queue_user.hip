// A user kernel: producer / consumer ring queue between neighbouring ranks, written against the public
// device header only (tdist/primitives.hpp: wait, consume_token, notify, symm_at).  It is compiled on its
// own (`hipcc --genco`, or `python -m triton_dist.tools.compile_aot --kernels this.hip:ring_queue_user
// --build`) and launched from Python through triton_dist.tools.load_module - nothing of it is linked into
// the library.
//
// Protocol (the reference's tutorial kernel, tutorials/01-distributed-notify-wait.py:62-145): item i uses
// slot i % Q in round r = i / Q; the slot's signal is 2r when free for round r and 2r + 1 when full.
//   producer workgroups (blockIdx < num_producers) write into the RIGHT neighbour's queue,
//   consumer workgroups pop this rank's queue into `output` and hand the slot back.
// `base_round` lets one launch continue where the previous one stopped, so a test can run thousands of
// rounds over the same symmetric queue without resetting a signal.
#include "tdist/primitives.hpp"

using namespace tdist;

extern "C" __global__ void ring_queue_user(SymmCtx ctx, int* queue, int* signal, const int* input, int* output,
                                           int peer, int queue_size, int item_elems, int num_items, int base_round,
                                           int num_producers, int num_consumers, int* err) {
  const int bid = blockIdx.x;
  if (bid < num_producers) {
    for (int i = bid; i < num_items; i += num_producers) {
      const int slot = i % queue_size, round = base_round + i / queue_size;
      int* remote_signal = symm_at(ctx, signal, peer) + slot;
      const int token = wait(remote_signal, 1, SCOPE_SYS, SEM_ACQUIRE, 2 * round, err);     // slot free?
      int* remote_queue = consume_token(symm_at(ctx, queue, peer), token) + (long long)slot * item_elems;
      for (int e = threadIdx.x; e < item_elems; e += blockDim.x) remote_queue[e] = input[(long long)i * item_elems + e];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) notify(ctx, signal + slot, peer, 2 * round + 1, SIGNAL_SET, COMM_INTRA_NODE);
      __syncthreads();
    }
  } else if (bid < num_producers + num_consumers) {
    const int cid = bid - num_producers;
    for (int i = cid; i < num_items; i += num_consumers) {
      const int slot = i % queue_size, round = base_round + i / queue_size;
      const int token = wait(signal + slot, 1, SCOPE_SYS, SEM_ACQUIRE, 2 * round + 1, err);  // slot full?
      const int* q = consume_token(queue, token) + (long long)slot * item_elems;
      for (int e = threadIdx.x; e < item_elems; e += blockDim.x) output[(long long)i * item_elems + e] = q[e];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (threadIdx.x == 0) notify(ctx, signal + slot, ctx.rank, 2 * round + 2, SIGNAL_SET, COMM_INTRA_NODE);
      __syncthreads();
    }
  }
}
