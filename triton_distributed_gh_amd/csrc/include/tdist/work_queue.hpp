// Work distribution for the persistent fused kernels: eight queues, one per XCD.
//
// Workgroups are dealt to the eight XCDs round-robin by index (XCD = blockIdx.x & 7), and each XCD has an
// L2 of its own.  A kernel therefore cuts its items into eight queues so that the items of one queue share
// operands (a band of B panels, a residue class of A row blocks), and workgroup b draws from queue b & 7.
//
// Drawing an item costs nothing on the critical path:
//   * the FIRST item of every worker is static - worker number k of XCD x owns item k of queue x - so a
//     launch starts computing without a single atomic;
//   * the draw for the NEXT item (one relaxed device-scope atomic by thread 0) is issued ahead of time -
//     prefetch() - and collected later, so its round trip (1-2 us idle, 5 us when all workgroups of an XCD
//     hit the word at once) hides under the item's work.  Where items of very different length share a
//     queue the draw should be issued late in the item (after the main loop, under the epilogue): a
//     workgroup that commits to its next item before a long one takes work away from idle workgroups.
//     stash() parks the drawn ticket in LDS at a point where the thread's memory operations have drained
//     anyway (the end of a main loop), so that next() does not wait for stores issued after it;
//   * a workgroup whose queue has run dry reads all eight counters with ONE vector load and goes straight
//     to a queue that still has items (or leaves) instead of probing seven more counters one by one.
// Ticket t of queue q stands for item n_static(q) + t, n_static(q) being the number of workers on XCD q.
//
// The counters must be zero at launch; the kernels re-arm them through call_end (call_state.hpp).  The
// first `first_worker` workgroups of the grid are not workers from the start (dedicated producers); they may
// join later through next().  All members are workgroup-uniform except `pf`, thread 0's in-flight ticket.
#pragma once
#include <hip/hip_runtime.h>

#include "tdist/atomics.hpp"

namespace tdist {

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global store the
// wave has in flight (its release fence covers global memory), which would stall a workgroup on the tile
// it has just written out before it may even look at its next ticket.
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// An int in LDS, typed as such: through a generic pointer the compiler emits flat loads, which count as
// vector-memory operations and would again wait for every global store in flight.
typedef __attribute__((address_space(3))) int lds_int;

template <class QSize>
struct WorkQueues {
  int* tickets;          // [8]
  volatile lds_int* slot;    // LDS mailbox of the current item: slot0 and slot0 + 3 ints in turn, so that a wave
                             // still reading one item's ticket can never see the next item's early stash()
  volatile lds_int* slot0;
  QSize qsize;           // int qsize(int q): items in queue q
  int first_worker;
  int q;                 // queue currently drawn from
  int pf;                // thread 0: ticket drawn ahead
  bool pending, stashed;

  __device__ __forceinline__ WorkQueues(int* tickets_, volatile int* slot_, int first_worker_, QSize qsize_)
      : tickets(tickets_), slot((volatile lds_int*)slot_), slot0((volatile lds_int*)slot_), qsize(qsize_), first_worker(first_worker_), q((int)(blockIdx.x & 7)), pf(0),
        pending(false), stashed(false) {}

  static __device__ __forceinline__ int count_on_xcd(int n, int x) { return n > x ? (n - x + 7) >> 3 : 0; }
  __device__ __forceinline__ int n_static(int x) const {
    return count_on_xcd((int)gridDim.x, x) - count_on_xcd(first_worker, x);
  }

  // The static item of this workgroup: true with its index in queue `q` (= own XCD), or false if this
  // workgroup is not a worker from the start or its queue is shorter than the XCD's worker count.
  __device__ __forceinline__ bool first(int& local) const {
    if ((int)blockIdx.x < first_worker) return false;
    const int x = (int)(blockIdx.x & 7);
    local = (int)(blockIdx.x >> 3) - count_on_xcd(first_worker, x);
    return local < qsize(x);
  }

  // Draw ahead for the item after the current one.  Call once per item, before its long-running part.
  __device__ __forceinline__ void prefetch() {
    if (threadIdx.x == 0) pf = atom_add_relaxed_agent(tickets + q, 1);
    pending = true;
  }

  // Park the ticket drawn by prefetch() in LDS (waits for the draw; nobody may be reading the slot).
  __device__ __forceinline__ void stash() {
    if (pending && threadIdx.x == 0) *slot = pf;
    stashed = pending;
  }

  // Next item: true with (q, local).  Workgroup-collective (barriers inside).
  __device__ __forceinline__ bool next(int& local) {
    if (!stashed) {
      lds_barrier();
      if (threadIdx.x == 0) {
        if (!pending) pf = atom_add_relaxed_agent(tickets + q, 1);
        *slot = pf;
      }
    }
    pending = false;
    stashed = false;
    lds_barrier();
    local = n_static(q) + __builtin_amdgcn_readfirstlane(*slot);
    volatile lds_int* const scan_slot = slot;
    slot = (slot == slot0) ? slot0 + 3 : slot0;         // the next item's mailbox
    if (local < qsize(q)) return true;
    // own queue is dry: one look at all eight counters
    while (true) {
      lds_barrier();
      if (threadIdx.x < 64) {
        const int lane = (int)threadIdx.x;
        bool has = false;
        if (lane < 8) has = n_static(lane) + load_relaxed_agent(tickets + lane) < qsize(lane);
        const unsigned long long m = __ballot(has);
        if (lane == 0) *scan_slot = (int)m;
      }
      lds_barrier();
      const int mask = __builtin_amdgcn_readfirstlane(*scan_slot) & ~(1 << q);
      if (mask == 0) return false;
      const int x = (int)(blockIdx.x & 7);
      bool found = false;
      for (int hop = 1; hop < 8 && !found; ++hop) {
        const int cand = (x + hop) & 7;
        if (!((mask >> cand) & 1)) continue;
        lds_barrier();
        if (threadIdx.x == 0) *scan_slot = atom_add_relaxed_agent(tickets + cand, 1);
        lds_barrier();
        local = n_static(cand) + __builtin_amdgcn_readfirstlane(*scan_slot);
        if (local < qsize(cand)) {
          q = cand;
          found = true;
        }
      }
      if (found) return true;
      // every candidate was taken in the meantime: look again (counters only grow, so this ends)
    }
  }
};

}  // namespace tdist
