// Tutorial 02 as a user kernel: push all-gather with one arrival flag per source rank.
//
//   gathered : symmetric [w][n16] 16-byte pieces - slot s of every rank receives rank s's shard
//   flags    : symmetric [w] ints                - flags[s] == epoch  <=>  slot s of THIS call has landed here
//   done     : local [w] ints, zero between launches (the last slice of a destination re-arms its word)
//
// Grid = w * slices workgroups.  Workgroup (d, s) stores slice s of the local shard into rank d's slot [rank]
// (d == rank: the local copy).  The slices of one destination run on different XCDs, whose L2s are not coherent
// with each other: a plain store would sit dirty in the storing XCD's L2, and the release fence of the ONE thread
// that finally notifies only writes back its own XCD's.  So the payload leaves as write-through stores
// (tdist/cache_policy.hpp: the sc0 sc1 bits push every line out of the L2 at once); every storing wave drains
// them, the workgroup barriers, one thread counts the slice on done[d], and the slice that completes destination d
// calls notify() - flags[rank] := epoch ON RANK d.  Slice 0 of every group then waits until rank d's shard has
// landed HERE, so the kernel's completion means "all w slots are valid".  `epoch` changes every call, so no flag is
// ever reset (the reference zeroes its barrier buffers between calls: tutorials/02-intra-node-allgather.py:70-160).
//
// Built on its own with hipcc --genco against tdist/primitives.hpp; nothing of it is linked into the library.
#include "tdist/cache_policy.hpp"
#include "tdist/primitives.hpp"

using namespace tdist;

extern "C" __global__ void allgather_push_user(SymmCtx ctx, const uint4* shard, uint4* gathered, int* flags, int* done,
                                               int n16, int slices, int epoch, int* err) {
  const int w = num_ranks(ctx), me = rank(ctx);
  const int d = blockIdx.x / slices, s = blockIdx.x % slices;
  const int per = (n16 + slices - 1) / slices;
  const int i0 = min(s * per, n16), i1 = min(i0 + per, n16);
  uint4* dst = symm_at(ctx, gathered, d) + (long long)me * n16;
  for (int i = i0 + threadIdx.x; i < i1; i += blockDim.x) st_wt128_sys(dst + i, shard[i]);
  drain_stores();                                       // this wave's stores have reached memory
  __syncthreads();                                      // ... and so have the other waves'
  if (threadIdx.x == 0) {
    if (atomicAdd(done + d, 1) == slices - 1) {         // all slices for rank d are out
      done[d] = 0;
      notify(ctx, flags + me, d, epoch, SIGNAL_SET, COMM_INTRA_NODE);
    }
  }
  if (s == 0) wait(flags + d, 1, SCOPE_SYS, SEM_ACQUIRE, epoch, err);    // rank d's shard has landed here
}
