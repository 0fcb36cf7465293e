// Small-message communication kernels over xGMI peer mappings:
//   * low-latency all-gather: push (data + epoch flag), pull (ready flag, then peer reads) and the
//     LL protocol (every 4 payload bytes travel with a 4-byte flag in one 8-byte store)
//   * DeepSeek-style low-latency all-to-all (EP dispatch / combine for decode)
//   * EP normal-mode dispatch / combine (token-granular puts with slot allocation)
//   * p2p copies for pipeline parallelism
//
// Role in the reference: kernels/nvidia/low_latency_allgather.py:47-777 (`_forward_pull_kernel`,
// `_forward_push_2d_kernel`, `_pack_ll_block` :548-566 / `_recv_ll_block` :530-545),
// kernels/nvidia/low_latency_all_to_all.py:35-118 (`all_to_all_kernel`),
// kernels/nvidia/ep_a2a.py:36-320 (dispatch / combine / splits / bincount),
// kernels/nvidia/p2p.py:30-85.  NUMA / multi-node / multimem variants have nothing to map to inside
// one full-mesh xGMI node and are served by the same kernels.
#include "tdist/cache_policy.hpp"
#include "tdist/call_state.hpp"
#include "tdist/common.hpp"
#include "tdist/primitives.hpp"
#include "tdist/shmem_device.hpp"

using namespace tdist;

// ------------------------------------------------------------------------------------------------
// all-gather.  `buf` is symmetric [w][bytes_per_rank]; slot `rank` holds the local shard on entry.
// grid = (w - 1) * slices.  flags: symmetric int32 [w] (push) / [w] (pull ready flags).
// ------------------------------------------------------------------------------------------------
struct AgParams {
  SymmCtx ctx;
  char* buf;
  int* flags;       // [2][w]: [0] data-arrived (push) / shard-ready (pull); [1] reader-done (pull)
  int* local_cnt;   // local int32 [w], zeroed before launch (push with slices > 1)
  int* err;
  long long bytes_per_rank;
  int slices;
  int epoch;
  // Device-side call numbering (tdist/call_state.hpp): with `state` set, `flags` is the base of BOTH halves (half h
  // at flags + h * half_stride), half and flag value come from the call number on the device and the last workgroup
  // re-arms `local_cnt`, so the launch is one kernel node and the same every time.
  int* state;
  long long half_stride;
};

// the call's flag half and flag value (uniform; ends with a workgroup barrier when the numbering is on the device)
__device__ __forceinline__ int ag_call_begin(AgParams& p, int* s_call) {
  if (!p.state) return 0;
  const int call = call_begin(p.state, s_call);
  p.flags += call_parity(call) * p.half_stride;
  p.epoch = call_epoch(call);
  return call;
}

__global__ __launch_bounds__(512) void allgather_push_kernel(AgParams p) {
  __shared__ int s_call;
  const int call = ag_call_begin(p, &s_call);
  const int w = p.ctx.num_ranks, rank = p.ctx.rank;
  const int pi = blockIdx.x / p.slices, slice = blockIdx.x % p.slices;
  const int peer = (rank + 1 + pi) % w;
  const long long per = ((p.bytes_per_rank + p.slices - 1) / p.slices + 15) & ~15ll;
  const long long b0 = min((long long)slice * per, p.bytes_per_rank), b1 = min(b0 + per, p.bytes_per_rank);
  char* src = p.buf + rank * p.bytes_per_rank;
  char* dst = symm_at(p.ctx, p.buf, peer) + rank * p.bytes_per_rank;
  // write-through stores: publishing is drain + barrier + counter (no release fence unless an
  // unaligned shard forced plain stores)
  const bool plain = shmem::copy_bytes_wt_sys(dst + b0, src + b0, (size_t)(b1 - b0), threadIdx.x, blockDim.x);
  if (plain) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  drain_stores();
  __syncthreads();
  if (threadIdx.x == 0) {
    const int prev = atom_add_relaxed_agent(p.local_cnt + peer, 1);
    if (prev == p.slices - 1) store_relaxed_system(symm_at(p.ctx, p.flags, peer) + rank, p.epoch);
  }
  __syncthreads();
  // every workgroup waits for the shard of "its" peer, so kernel completion = all shards present
  if (slice == 0) wait(p.flags + peer, 1, SCOPE_SYS, SEM_ACQUIRE, p.epoch, p.err);
  if (p.state) call_end(p.state, call, p.state + 1, p.local_cnt, TDIST_MAX_RANKS);
}

__global__ __launch_bounds__(512) void allgather_pull_kernel(AgParams p) {
  __shared__ int s_call;
  const int call = ag_call_begin(p, &s_call);
  const int w = p.ctx.num_ranks, rank = p.ctx.rank;
  const int pi = blockIdx.x / p.slices, slice = blockIdx.x % p.slices;
  const int peer = (rank + 1 + pi) % w;
  // announce "my shard is ready" to the peer this workgroup serves, then wait for the peer's
  if (slice == 0 && threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_relaxed_system(symm_at(p.ctx, p.flags, peer) + rank, p.epoch);
  }
  wait(p.flags + peer, 1, SCOPE_SYS, SEM_ACQUIRE, p.epoch, p.err);
  const long long per = ((p.bytes_per_rank + p.slices - 1) / p.slices + 15) & ~15ll;
  const long long b0 = min((long long)slice * per, p.bytes_per_rank), b1 = min(b0 + per, p.bytes_per_rank);
  const char* src = symm_at(p.ctx, p.buf, peer) + peer * p.bytes_per_rank;
  shmem::copy_bytes(p.buf + peer * p.bytes_per_rank + b0, src + b0, (size_t)(b1 - b0), threadIdx.x, blockDim.x);
  // tell the peer we are done reading its shard (it may overwrite it once everyone has), and do
  // not leave before every peer has finished reading ours
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int prev = atom_add_acqrel_agent(p.local_cnt + peer, 1);
    if (prev == p.slices - 1) store_relaxed_system(symm_at(p.ctx, p.flags, peer) + w + rank, p.epoch);
  }
  __syncthreads();
  if (slice == 0) wait(p.flags + w + peer, 1, SCOPE_SYS, SEM_RELAXED, p.epoch, p.err);
  if (p.state) call_end(p.state, call, p.state + 1, p.local_cnt, TDIST_MAX_RANKS);
}

// LL protocol: staging is symmetric uint64 [w][words_per_rank]; word = (flag << 32) | payload32.
// Sender workgroup pi packs and stores the local shard into peer's staging slot [rank]; receiver
// workgroup (same kernel, blockIdx >= w-1) spins per word and unpacks into buf slot [src].
struct AgLLParams {
  SymmCtx ctx;
  char* buf;                  // [w][bytes_per_rank], local shard at slot rank (bytes multiple of 4)
  unsigned long long* stage;  // symmetric [w][words]
  int* err;
  long long words;
  unsigned flag;              // != 0, changes every call
  // Device-side call numbering (tdist/call_state.hpp): with `state` set, `stage` is the base of BOTH halves
  // (half h at stage + h * half_stride words) and half / flag come from the call number on the device, so
  // the launch is the same every time and replays from a HIP graph.
  int* state;
  long long half_stride;
};

__global__ __launch_bounds__(512) void allgather_ll_kernel(AgLLParams p) {
  __shared__ int s_call;
  const int w = p.ctx.num_ranks, rank = p.ctx.rank;
  const int b = blockIdx.x;
  int call = 0;
  if (p.state) {
    call = call_begin(p.state, &s_call);
    p.stage += (long long)call_parity(call) * p.half_stride;
    p.flag = (unsigned)call_epoch(call);
  }
  if (b < w - 1) {
    const int peer = (rank + 1 + b) % w;
    const unsigned* src = (const unsigned*)(p.buf + (long long)rank * p.words * 4);
    unsigned long long* dst = symm_at(p.ctx, p.stage, peer) + (long long)rank * p.words;
    for (long long i = threadIdx.x; i < p.words; i += blockDim.x)
      store_relaxed_system(dst + i, ((unsigned long long)p.flag << 32) | src[i]);
  } else {
    const int src_rank = (rank + 1 + (b - (w - 1))) % w;
    const unsigned long long* st = p.stage + (long long)src_rank * p.words;
    unsigned* dst = (unsigned*)(p.buf + (long long)src_rank * p.words * 4);
    const uint64_t t0 = wall_ticks();
    for (long long i = threadIdx.x; i < p.words; i += blockDim.x) {
      unsigned long long v;
      int spins = 0;
      while (true) {
        v = load_relaxed_system(st + i);
        if ((unsigned)(v >> 32) == p.flag) break;
        if (((++spins) & 1023) == 0 && wall_ticks() - t0 > TDIST_SPIN_TIMEOUT_TICKS) {
          atomicOr(p.err, 1);
          break;
        }
      }
      dst[i] = (unsigned)v;
    }
  }
  if (p.state) call_end(p.state, call);
}

// ------------------------------------------------------------------------------------------------
// Low-latency all-to-all.  `slices` workgroups serve each destination rank `pid`, each moving a
// 16-byte-aligned byte range of
//   rows [cum[pid*EPR], cum[(pid+1)*EPR]) of send_buf -> peer recv[half][rank*max_m ...]
//   optional per-row scales (scale_elems f32) -> peer scale[half][rank*max_m ...]
// slice 0 also writes the per-expert splits (EPR ints) -> peer split[half][rank*EPR ...].
// One workgroup moves 12-16 GB/s, an xGMI link several times that, so a destination needs a few
// workgroups to be link-bound.  The slice that finishes last (local counter `done[pid]`, zero at
// rest: the last slice re-arms it, so the launch needs no memset) raises signal[half][rank] := call_count on the peer; slice 0 of every destination
// then waits for signal[half][pid] == call_count, so kernel completion = everything has arrived.
// ------------------------------------------------------------------------------------------------
struct A2AParams {
  SymmCtx ctx;
  const char* send_buf;     // [<= max_m, row_bytes]
  char* recv_buf;           // symmetric [2][w*max_m, row_bytes]
  const float* send_scale;  // [<= max_m, scale_elems] or null
  float* recv_scale;        // symmetric [2][w*max_m, scale_elems]
  const int* cumsum;        // [num_tot_experts + 1] exclusive cumsum of tokens per expert
  int* split_buf;           // symmetric [2][w*EPR]
  unsigned long long* signal;  // symmetric [2][w]
  int* err;
  int* done;                // local [w], zero between launches
  long long row_bytes;
  int max_m, epr, scale_elems, slices;
  int n_rows;               // rows present in send_buf
  unsigned long long call_count;   // used when `state` is null (host-numbered calls)
  // Device-side call numbering (tdist/call_state.hpp): the half of the receive buffers and the signal value come
  // from state[0], so the launch is the same every call and can be replayed from a HIP graph.  What arrived is then
  // copied into buffers whose address does not depend on the half (out_* below), which is what the caller reads.
  int* state;               // [0] call number, [1] finished workgroups
  char* out_rows;           // local [w*max_m, row_bytes]
  float* out_scale;         // local [w*max_m, scale_elems] (when scales are sent)
  int* out_splits;          // local [w*EPR]
};

// Returns true when part of the slice went out as plain (not write-through) stores.
__device__ __forceinline__ bool a2a_copy_slice(char* dst, const char* src, long long total, int slice, int slices) {
  const long long per = ((total + slices - 1) / slices + 15) & ~15ll;
  const long long b0 = min((long long)slice * per, total), b1 = min(b0 + per, total);
  if (b1 <= b0) return false;
  return shmem::copy_bytes_wt_sys(dst + b0, src + b0, (size_t)(b1 - b0), threadIdx.x, blockDim.x);
}

__global__ __launch_bounds__(1024) void all_to_all_kernel(A2AParams p) {
  __shared__ int s_call, s_rows;
  const int w = p.ctx.num_ranks, rank = p.ctx.rank;
  const int pid = blockIdx.x / p.slices, slice = blockIdx.x % p.slices;
  int call = 0;
  unsigned long long sigval = p.call_count;
  if (p.state) {
    call = call_begin(p.state, &s_call);
    sigval = (unsigned long long)call_epoch(call);
  }
  const int half = p.state ? call_parity(call) : (int)(p.call_count & 1);
  const int c_st = p.cumsum[pid * p.epr], c_ed = p.cumsum[(pid + 1) * p.epr];
  // a malformed cumsum must not turn into an out-of-range read or peer write
  const int m_st = min(max(c_st, 0), p.n_rows);
  const int rows = min(min(max(c_ed - m_st, 0), p.max_m), p.n_rows - m_st);
  if ((m_st != c_st || rows != c_ed - c_st) && threadIdx.x == 0) atomicOr(p.err, 2);
  const long long slot_row = (long long)half * w * p.max_m + (long long)rank * p.max_m;
  char* dst = symm_at(p.ctx, p.recv_buf, pid) + slot_row * p.row_bytes;
  // Payload, scales and splits all leave as write-through system-scope stores, so publishing is
  // drain + barrier + counter; a release fence (a walk over the whole L2) is only paid when an
  // unaligned operand forced plain stores.
  bool plain = a2a_copy_slice(dst, p.send_buf + (long long)m_st * p.row_bytes, (long long)rows * p.row_bytes, slice,
                              p.slices);
  if (p.send_scale) {
    float* sd = symm_at(p.ctx, p.recv_scale, pid) + slot_row * p.scale_elems;
    plain |= a2a_copy_slice((char*)sd, (const char*)(p.send_scale + (long long)m_st * p.scale_elems),
                            (long long)rows * p.scale_elems * (long long)sizeof(float), slice, p.slices);
  }
  if (slice == 0) {
    int* sp = symm_at(p.ctx, p.split_buf, pid) + half * w * p.epr + rank * p.epr;
    for (int e = threadIdx.x; e < p.epr; e += blockDim.x)
      store_relaxed_system(sp + e, p.cumsum[pid * p.epr + e + 1] - p.cumsum[pid * p.epr + e]);
  }
  if (plain) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  drain_stores();
  __syncthreads();
  if (threadIdx.x == 0) {
    // relaxed: every slice's stores were drained before its add, and the add that completes the
    // destination is ordered after all earlier adds on the word
    const int prev = atom_add_relaxed_agent(p.done + pid, 1);
    if (prev == p.slices - 1) {
      // last slice of this destination: re-arm the counter for the next call (nobody else touches
      // it any more in this launch), then publish
      store_relaxed_agent(p.done + pid, 0);
      store_relaxed_system(symm_at(p.ctx, p.signal, pid) + half * w + rank, sigval);
    }
  }
  if (!p.state) {
    if (threadIdx.x == 0 && slice == 0) {
      if (!spin_until_eq(p.signal + half * w + pid, sigval, SCOPE_SYS)) atomicOr(p.err, 1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    return;
  }
  // Receive side of source `pid` (device-numbered calls): EVERY slice waits for that source's signal, then the
  // slices share the copy of what it sent - rows, scales, splits - out of this call's half into the caller's buffers.
  if (threadIdx.x == 0) {
    if (!spin_until_eq(p.signal + half * w + pid, sigval, SCOPE_SYS)) atomicOr(p.err, 1);
  }
  if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int* sp_in = p.split_buf + half * w * p.epr + pid * p.epr;
  if (threadIdx.x == 0) {
    int n = 0;
    for (int e = 0; e < p.epr; ++e) n += sp_in[e];
    s_rows = min(max(n, 0), p.max_m);
  }
  __syncthreads();
  const int got = s_rows;
  const long long src_row = (long long)half * w * p.max_m + (long long)pid * p.max_m;
  {
    const long long total = (long long)got * p.row_bytes;
    const long long per = ((total + p.slices - 1) / p.slices + 15) & ~15ll;
    const long long b0 = min((long long)slice * per, total), b1 = min(b0 + per, total);
    if (b1 > b0)
      shmem::copy_bytes(p.out_rows + (long long)pid * p.max_m * p.row_bytes + b0, p.recv_buf + src_row * p.row_bytes + b0,
                        (size_t)(b1 - b0), threadIdx.x, blockDim.x);
  }
  if (p.send_scale && slice == p.slices - 1) {
    const long long n = (long long)got * p.scale_elems;
    const float* ss = p.recv_scale + src_row * p.scale_elems;
    float* sd = p.out_scale + (long long)pid * p.max_m * p.scale_elems;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) sd[i] = ss[i];
  }
  if (slice == 0)
    for (int e = threadIdx.x; e < p.epr; e += blockDim.x) p.out_splits[pid * p.epr + e] = sp_in[e];
  call_end(p.state, call);
}

// Compaction of what fast_all_to_all received (`rows` [w, max_m, row_bytes], `splits` [w, EPR], optional `scale`):
// the rows of source s move to out[offset(s) ..] with offset = rows of the sources before it, tokens_per_expert[e] =
// sum over sources, total[0] = all rows.  One workgroup per (source, row block); no host round trip, capturable.
struct A2ACompactParams {
  const char* rows;
  const float* scale;
  const int* splits;
  char* out;
  float* out_scale;
  int* tokens_per_expert;   // [EPR]
  int* total;               // [1]
  long long row_bytes;
  int w, max_m, epr, scale_elems, rows_per_block;
};

__global__ __launch_bounds__(256) void a2a_compact_kernel(A2ACompactParams p) {
  __shared__ int s_off, s_cnt;
  const int blocks_per_src = (p.max_m + p.rows_per_block - 1) / p.rows_per_block;
  const int src = blockIdx.x / blocks_per_src, rb = blockIdx.x % blocks_per_src;
  if (threadIdx.x == 0) {
    int off = 0, cnt = 0;
    for (int s2 = 0; s2 <= src; ++s2) {
      int n = 0;
      for (int e = 0; e < p.epr; ++e) n += p.splits[s2 * p.epr + e];
      n = min(max(n, 0), p.max_m);
      if (s2 < src) off += n; else cnt = n;
    }
    s_off = off;
    s_cnt = cnt;
  }
  if (blockIdx.x == 0) {
    for (int e = threadIdx.x; e < p.epr; e += blockDim.x) {
      int n = 0;
      for (int s2 = 0; s2 < p.w; ++s2) n += p.splits[s2 * p.epr + e];
      p.tokens_per_expert[e] = n;
    }
    if (threadIdx.x == 0) {
      int n = 0;
      for (int s2 = 0; s2 < p.w; ++s2) {
        int m = 0;
        for (int e = 0; e < p.epr; ++e) m += p.splits[s2 * p.epr + e];
        n += min(max(m, 0), p.max_m);
      }
      p.total[0] = n;
    }
  }
  __syncthreads();
  const int r0 = rb * p.rows_per_block, r1 = min(r0 + p.rows_per_block, s_cnt);
  if (r1 <= r0) return;
  shmem::copy_bytes(p.out + (long long)(s_off + r0) * p.row_bytes, p.rows + ((long long)src * p.max_m + r0) * p.row_bytes,
                    (size_t)((long long)(r1 - r0) * p.row_bytes), threadIdx.x, blockDim.x);
  if (p.scale) {
    const long long n = (long long)(r1 - r0) * p.scale_elems;
    const float* ss = p.scale + ((long long)src * p.max_m + r0) * p.scale_elems;
    float* sd = p.out_scale + (long long)(s_off + r0) * p.scale_elems;
    for (long long i = threadIdx.x; i < n; i += blockDim.x) sd[i] = ss[i];
  }
}

// ------------------------------------------------------------------------------------------------
// EP normal mode.  All three kernels take the call's epoch from a device word (state[0]): the counts kernel uses
// the value after state[0] and its last workgroup stores it back, dispatch and combine - later in stream order - read it.
// Launch arguments are therefore the same every call and the whole dispatch -> combine round trip can be replayed
// from a HIP graph.
//
// counts + offsets (one launch; the reference's `kernel_get_ag_splits_and_recv_offset`, ep_a2a.py:243-306, plus its
// `kernel_bincount` :309-320): every workgroup counts this rank's (token, k) pairs per expert in LDS; workgroup b
// pushes the counts [E] to peer b's table row [rank], raises the peer's flag and waits for peer b's counts to land
// here.  The last workgroup to finish then has the full [source rank, expert] table and computes
//   dst_base[e]        first row, inside the owner's receive buffer, of the (expert e, source = this rank) segment
//                      (segments ordered by local expert, then source rank),
//   recv_per_expert[l] rows this rank receives for its local expert l,   num_recv  their sum
// (num_recv also goes to a pinned host word for callers that want an exactly sized view), zeroes the dispatch
// kernel's cursors and advances the epoch.
// ------------------------------------------------------------------------------------------------
struct EpCountParams {
  SymmCtx ctx;
  const int* topk_idx; // local [n_pairs] expert of every (token, k) pair
  int n_pairs;
  int* table;          // symmetric [w][E]
  int* flags;          // symmetric [w]
  int* err;
  int* state;          // [0] epoch of the latest dispatch, [1] finished workgroups of this launch
  int* dst_base;       // local [E]
  int* recv_per_expert;// local [EPR]
  int* num_recv;       // local [1]
  int* num_recv_host;  // pinned host [1] or null
  int* cursor;         // local [E]: slot allocators of the dispatch kernel, zeroed here
  int E, epr;
};

__global__ __launch_bounds__(256) void ep_exchange_counts_kernel(EpCountParams p) {
  extern __shared__ int s_counts[];          // [E]
  __shared__ int s_last;
  const int w = p.ctx.num_ranks, rank = p.ctx.rank, peer = blockIdx.x;
  const int epoch = load_relaxed_agent(p.state) % CALL_EPOCH_MOD + 1;      // flag values stay in 1 .. CALL_EPOCH_MOD
  for (int e = threadIdx.x; e < p.E; e += blockDim.x) s_counts[e] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < p.n_pairs; i += blockDim.x) {
    const int e = p.topk_idx[i];
    if (e >= 0 && e < p.E) atomicAdd(&s_counts[e], 1);
  }
  __syncthreads();
  int* dst = symm_at(p.ctx, p.table, peer) + rank * p.E;
  for (int e = threadIdx.x; e < p.E; e += blockDim.x) store_relaxed_system(dst + e, s_counts[e]);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) store_relaxed_system(symm_at(p.ctx, p.flags, peer) + rank, epoch);
  wait(p.flags + peer, 1, SCOPE_SYS, SEM_ACQUIRE, epoch, p.err);
  if (threadIdx.x == 0) s_last = atom_add_relaxed_agent(p.state + 1, 1) == (int)gridDim.x - 1;
  __syncthreads();
  if (!s_last) return;
  // every peer's row has landed (each workgroup acquired its own peer's row before counting itself)
  if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int e = threadIdx.x; e < p.E; e += blockDim.x) {
    const int o = e / p.epr, le = e - o * p.epr;
    int base = 0;
    for (int l2 = 0; l2 < le; ++l2)
      for (int s2 = 0; s2 < w; ++s2) base += load_relaxed_system(p.table + s2 * p.E + o * p.epr + l2);
    for (int s2 = 0; s2 < rank; ++s2) base += load_relaxed_system(p.table + s2 * p.E + e);
    p.dst_base[e] = base;
    p.cursor[e] = 0;
  }
  for (int l = threadIdx.x; l < p.epr; l += blockDim.x) {
    int n = 0;
    for (int s2 = 0; s2 < w; ++s2) n += load_relaxed_system(p.table + s2 * p.E + rank * p.epr + l);
    p.recv_per_expert[l] = n;
    s_counts[l] = n;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int n = 0;
    for (int l = 0; l < p.epr; ++l) n += s_counts[l];
    p.num_recv[0] = n;
    if (p.num_recv_host) *(volatile int*)p.num_recv_host = n;
    store_relaxed_agent(p.state + 1, 0);
    store_relaxed_agent(p.state, epoch);
  }
}

// dispatch: one wave per (token, k).  dst_base[e] = first row, inside the destination rank's
// receive buffer, of the (expert e, this source rank) segment; cursor[e] is a local atomic slot
// allocator (zeroed by the counts kernel).  Writes the row and records scatter_idx[token, k] = destination row.
// Rows beyond the destination's capacity are dropped and reported (error bit 1).
struct EpDispatchParams {
  SymmCtx ctx;
  const char* tokens;      // [T, row_bytes]
  const int* topk_idx;     // [T, topk]
  const int* dst_base;     // [E]
  int* cursor;             // [E]
  char* recv_buf;          // symmetric [cap_rows, row_bytes]
  int* scatter_idx;        // [T, topk]
  int* flags;              // symmetric [w]: source rank's tokens have landed
  int* done;               // local counter, zero at launch (re-armed by the last workgroup)
  int* err;
  const int* state;        // [0] epoch of this dispatch (stored by the counts kernel)
  long long row_bytes;
  int T, topk, E, epr, cap_rows;
};

__global__ __launch_bounds__(512) void ep_dispatch_kernel(EpDispatchParams p) {
  const int w = p.ctx.num_ranks;
  const int waves_per_block = blockDim.x / 64;
  const int lane = threadIdx.x & 63;
  const long long total = (long long)p.T * p.topk;
  const int epoch = load_relaxed_agent(p.state);
  bool plain = false;
  for (long long item = (long long)blockIdx.x * waves_per_block + (threadIdx.x >> 6); item < total;
       item += (long long)gridDim.x * waves_per_block) {
    const int t = (int)(item / p.topk);
    const int e = p.topk_idx[item];
    if (e < 0 || e >= p.E) {                         // not routed (the counts kernel skipped it too)
      if (lane == 0) p.scatter_idx[item] = -1;
      continue;
    }
    const int d = e / p.epr;
    int slot = 0;
    if (lane == 0) slot = p.dst_base[e] + atomicAdd(p.cursor + e, 1);
    slot = __shfl(slot, 0);
    if (slot >= p.cap_rows) {
      if (lane == 0) {
        atomicOr(p.err, 2);
        p.scatter_idx[item] = -1;
      }
      continue;
    }
    char* dst = symm_at(p.ctx, p.recv_buf, d) + (long long)slot * p.row_bytes;
    plain |= shmem::copy_bytes_wt_sys(dst, p.tokens + (long long)t * p.row_bytes, (size_t)p.row_bytes, lane, 64);
    if (lane == 0) p.scatter_idx[item] = slot;
  }
  // token rows went out write-through; a release fence only if unaligned rows forced plain stores
  if (plain) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  drain_stores();
  __syncthreads();
  if (threadIdx.x == 0) {
    const int prev = atom_add_relaxed_agent(p.done, 1);
    if (prev == (int)gridDim.x - 1) {
      store_relaxed_agent(p.done, 0);
      for (int r = 0; r < w; ++r) store_relaxed_system(symm_at(p.ctx, p.flags, r) + p.ctx.rank, epoch);
    }
  }
  __syncthreads();
  // kernel completion = every source's tokens are present locally
  wait(p.flags, w, SCOPE_SYS, SEM_ACQUIRE, epoch, p.err);
}

// combine: one workgroup per token; out[t] = sum_k weight[t,k] * expert_out(owner rank)[scatter_idx[t,k]]
struct EpCombineParams {
  SymmCtx ctx;
  const uint16_t* expert_out;   // symmetric [cap_rows, hidden] (bf16)
  const int* topk_idx;
  const int* scatter_idx;
  const float* weights;         // [T, topk] or null (=1)
  uint16_t* out;                // [T, hidden]
  int* flags;                   // symmetric [w]: rank's expert outputs are final
  int* err;
  const int* state;             // [0] epoch of the dispatch being combined
  int T, topk, epr, hidden;
};

__global__ __launch_bounds__(256) void ep_combine_kernel(EpCombineParams p) {
  const int w = p.ctx.num_ranks;
  const int epoch = load_relaxed_agent(p.state);
  // phase 0: tell everyone that this rank's expert_out is complete (it was produced by earlier
  // kernels on this stream), then wait for all ranks.
  if (blockIdx.x == 0 && threadIdx.x < w) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_relaxed_system(symm_at(p.ctx, p.flags, (int)threadIdx.x) + p.ctx.rank, epoch);
  }
  wait(p.flags, w, SCOPE_SYS, SEM_ACQUIRE, epoch, p.err);
  for (int t = blockIdx.x; t < p.T; t += gridDim.x) {
    for (int c = threadIdx.x * 8; c < p.hidden; c += blockDim.x * 8) {
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int k = 0; k < p.topk; ++k) {
        const int e = p.topk_idx[t * p.topk + k];
        const int d = e / p.epr;
        const int slot = p.scatter_idx[t * p.topk + k];
        if (slot < 0) continue;                       // (dropped at dispatch: over capacity)
        const float wt = p.weights ? p.weights[t * p.topk + k] : 1.f;
        const uint4 v = *(const uint4*)(symm_at(p.ctx, const_cast<uint16_t*>(p.expert_out), d) + (long long)slot * p.hidden + c);
        const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[2 * i] += wt * __builtin_bit_cast(float, x[i] << 16);
          acc[2 * i + 1] += wt * __builtin_bit_cast(float, x[i] & 0xffff0000u);
        }
      }
      typedef __attribute__((ext_vector_type(2))) __bf16 b2;
      uint32_t o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        b2 v = {(__bf16)acc[2 * i], (__bf16)acc[2 * i + 1]};
        o[i] = __builtin_bit_cast(uint32_t, v);
      }
      *(uint4*)(p.out + (long long)t * p.hidden + c) = uint4{o[0], o[1], o[2], o[3]};
    }
  }
}

// p2p copy (pull): local_dst <- peer's symmetric src
__global__ __launch_bounds__(512) void p2p_pull_kernel(SymmCtx ctx, char* local_dst, char* symm_src, long long bytes,
                                                       int peer) {
  const long long per = ((bytes + gridDim.x - 1) / gridDim.x + 15) & ~15ll;
  const long long b0 = min((long long)blockIdx.x * per, bytes), b1 = min(b0 + per, bytes);
  shmem::copy_bytes(local_dst + b0, symm_at(ctx, symm_src, peer) + b0, (size_t)(b1 - b0), threadIdx.x, blockDim.x);
}

// ------------------------------------------------------------------------------------------------
// `state` != null: int32 [2] = {call number, finished workgroups}, zero when created, `local_cnt` zero at the first
// launch; `flags` is then the base of both halves, `half_stride` their distance in ints, and `epoch` is ignored.
TDIST_API int tdist_allgather(const SymmCtx* ctx, void* buf, int* flags, int* local_cnt, int* err,
                              long long bytes_per_rank, int slices, int epoch, int mode, void* stream, int* state,
                              long long half_stride) {
  const int w = ctx->num_ranks;
  if (w <= 1) return 0;
  AgParams p;
  p.ctx = *ctx;
  p.buf = (char*)buf;
  p.flags = flags;
  p.local_cnt = local_cnt;
  p.err = err;
  p.bytes_per_rank = bytes_per_rank;
  p.slices = slices > 0 ? slices : 1;
  p.epoch = epoch;
  p.state = state;
  p.half_stride = half_stride;
  hipStream_t s = (hipStream_t)stream;
  if (!state) TDIST_HIP_TRY(hipMemsetAsync(local_cnt, 0, TDIST_MAX_RANKS * sizeof(int), s));
  if (mode == 0) {
    hipLaunchKernelGGL(allgather_push_kernel, dim3((w - 1) * p.slices), dim3(512), 0, s, p);
  } else {
    hipLaunchKernelGGL(allgather_pull_kernel, dim3((w - 1) * p.slices), dim3(512), 0, s, p);
  }
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

// `state` != null: int32 [2] = {call number, finished workgroups}, zero when created; `stage` is then the base of
// both halves, `half_stride` their distance in 8-byte words, and `flag` is ignored.
TDIST_API int tdist_allgather_ll(const SymmCtx* ctx, void* buf, void* stage, int* err, long long bytes_per_rank,
                                 unsigned flag, void* stream, int* state, long long half_stride) {
  const int w = ctx->num_ranks;
  if (w <= 1) return 0;
  TDIST_REQUIRE(bytes_per_rank % 4 == 0 && (flag != 0 || state), "LL all-gather needs 4-byte multiples and a non-zero flag");
  AgLLParams p;
  p.ctx = *ctx;
  p.buf = (char*)buf;
  p.stage = (unsigned long long*)stage;
  p.err = err;
  p.words = bytes_per_rank / 4;
  p.flag = flag;
  p.state = state;
  p.half_stride = half_stride;
  hipLaunchKernelGGL(allgather_ll_kernel, dim3(2 * (w - 1)), dim3(512), 0, (hipStream_t)stream, p);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

// scratch: local int32 [4 + TDIST_MAX_RANKS]: [0] sticky error word (bit 0 wait timeout, bit 1 bad
// cumsum), [4 ...] per-destination slice counters (zero-initialised by the caller, self-resetting).
TDIST_API int tdist_all_to_all(const SymmCtx* ctx, const void* send_buf, void* recv_buf, const float* send_scale,
                               float* recv_scale, const int* cumsum, int* split_buf, void* signal, int* scratch,
                               long long row_bytes, int n_rows, int max_m, int epr, int scale_elems, int slices,
                               unsigned long long call_count, void* stream, int* state, void* out_rows, float* out_scale,
                               int* out_splits) {
  const int w = ctx->num_ranks;
  TDIST_REQUIRE(w >= 1 && w <= TDIST_MAX_RANKS, "bad world size");
  TDIST_REQUIRE(slices >= 1 && slices <= 32, "slices must be in [1, 32]");
  TDIST_REQUIRE(row_bytes > 0 && max_m > 0 && epr > 0 && scale_elems >= 0 && n_rows >= 0 && n_rows <= max_m,
                "bad a2a shape");
  TDIST_REQUIRE(!state || (out_rows && out_splits && (!send_scale || out_scale)),
                "device-numbered calls need the parity-independent output buffers");
  A2AParams p;
  p.ctx = *ctx;
  p.send_buf = (const char*)send_buf;
  p.recv_buf = (char*)recv_buf;
  p.send_scale = send_scale;
  p.recv_scale = recv_scale;
  p.cumsum = cumsum;
  p.split_buf = split_buf;
  p.signal = (unsigned long long*)signal;
  p.err = scratch;
  p.done = scratch + 4;
  p.row_bytes = row_bytes;
  p.max_m = max_m;
  p.epr = epr;
  p.scale_elems = scale_elems;
  p.slices = slices;
  p.n_rows = n_rows;
  p.call_count = call_count;
  p.state = state;
  p.out_rows = (char*)out_rows;
  p.out_scale = out_scale;
  p.out_splits = out_splits;
  hipLaunchKernelGGL(all_to_all_kernel, dim3(w * slices), dim3(1024), 0, (hipStream_t)stream, p);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

TDIST_API int tdist_a2a_compact(const void* rows, const float* scale, const int* splits, void* out, float* out_scale,
                                int* tokens_per_expert, int* total, long long row_bytes, int w, int max_m, int epr,
                                int scale_elems, void* stream) {
  TDIST_REQUIRE(rows && splits && out && tokens_per_expert && total && w >= 1 && max_m > 0 && epr > 0, "bad arguments");
  A2ACompactParams p;
  p.rows = (const char*)rows;
  p.scale = scale;
  p.splits = splits;
  p.out = (char*)out;
  p.out_scale = out_scale;
  p.tokens_per_expert = tokens_per_expert;
  p.total = total;
  p.row_bytes = row_bytes;
  p.w = w;
  p.max_m = max_m;
  p.epr = epr;
  p.scale_elems = scale_elems;
  p.rows_per_block = 16;
  const int blocks = w * ((max_m + p.rows_per_block - 1) / p.rows_per_block);
  hipLaunchKernelGGL(a2a_compact_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

// `state`: int32 [2], zero when created (epoch of the latest dispatch, finished workgroups).  `cursor` [E] is
// zeroed by the kernel.  `num_recv_host`: optional pinned host word.
TDIST_API int tdist_ep_exchange_counts(const SymmCtx* ctx, const int* topk_idx, int n_pairs, int* table, int* flags,
                                       int* err, int* state, int* dst_base, int* recv_per_expert, int* num_recv,
                                       int* num_recv_host, int* cursor, int E, int epr, void* stream) {
  TDIST_REQUIRE(E > 0 && epr > 0 && E == epr * ctx->num_ranks && E <= 8192, "bad expert count");
  TDIST_REQUIRE(topk_idx && table && flags && err && state && dst_base && recv_per_expert && num_recv && cursor, "null argument");
  EpCountParams p;
  p.ctx = *ctx;
  p.topk_idx = topk_idx;
  p.n_pairs = n_pairs;
  p.table = table;
  p.flags = flags;
  p.err = err;
  p.state = state;
  p.dst_base = dst_base;
  p.recv_per_expert = recv_per_expert;
  p.num_recv = num_recv;
  p.num_recv_host = num_recv_host;
  p.cursor = cursor;
  p.E = E;
  p.epr = epr;
  hipLaunchKernelGGL(ep_exchange_counts_kernel, dim3(ctx->num_ranks), dim3(256), (size_t)E * sizeof(int), (hipStream_t)stream, p);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

TDIST_API int tdist_ep_dispatch(const SymmCtx* ctx, const void* tokens, const int* topk_idx, const int* dst_base,
                                int* cursor, void* recv_buf, int* scatter_idx, int* flags, int* done, int* err,
                                const int* state, long long row_bytes, int T, int topk, int E, int epr, int cap_rows,
                                int num_ctas, void* stream) {
  EpDispatchParams p;
  p.ctx = *ctx;
  p.tokens = (const char*)tokens;
  p.topk_idx = topk_idx;
  p.dst_base = dst_base;
  p.cursor = cursor;
  p.recv_buf = (char*)recv_buf;
  p.scatter_idx = scatter_idx;
  p.flags = flags;
  p.done = done;
  p.err = err;
  p.state = state;
  p.row_bytes = row_bytes;
  p.T = T;
  p.topk = topk;
  p.E = E;
  p.epr = epr;
  p.cap_rows = cap_rows;
  hipLaunchKernelGGL(ep_dispatch_kernel, dim3(num_ctas > 0 ? num_ctas : 40), dim3(512), 0, (hipStream_t)stream, p);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

TDIST_API int tdist_ep_combine(const SymmCtx* ctx, const void* expert_out, const int* topk_idx, const int* scatter_idx,
                               const float* weights, void* out, int* flags, int* err, const int* state, int T, int topk,
                               int epr, int hidden, int num_ctas, void* stream) {
  TDIST_REQUIRE(hidden % 8 == 0, "hidden must be a multiple of 8");
  EpCombineParams p;
  p.ctx = *ctx;
  p.expert_out = (const uint16_t*)expert_out;
  p.topk_idx = topk_idx;
  p.scatter_idx = scatter_idx;
  p.weights = weights;
  p.out = (uint16_t*)out;
  p.flags = flags;
  p.err = err;
  p.state = state;
  p.T = T;
  p.topk = topk;
  p.epr = epr;
  p.hidden = hidden;
  hipLaunchKernelGGL(ep_combine_kernel, dim3(num_ctas > 0 ? num_ctas : 64), dim3(256), 0, (hipStream_t)stream, p);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

TDIST_API int tdist_p2p_pull(const SymmCtx* ctx, void* local_dst, void* symm_src, long long bytes, int peer, int num_ctas,
                             void* stream) {
  hipLaunchKernelGGL(p2p_pull_kernel, dim3(num_ctas > 0 ? num_ctas : 4), dim3(512), 0, (hipStream_t)stream, *ctx,
                     (char*)local_dst, (char*)symm_src, bytes, peer);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

// 1 in the library built with -DTDIST_SAFE_PUBLISH=1 (release fence before every flag store), 0 in the default one.
TDIST_API int tdist_build_safe_publish() { return TDIST_SAFE_PUBLISH; }
