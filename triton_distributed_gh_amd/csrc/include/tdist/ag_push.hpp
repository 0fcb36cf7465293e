// In-kernel all-gather producer shared by AG + GEMM (ops/ag_gemm.hip) and AG + MoE (ops/moe.hip).
//
// `Params` provides: ctx (SymmCtx), a_local, lda_local, K, M_per_rank, rows_per_chunk, rank, world, push_wgs,
// push_done (int[chunks], zero at launch), straddle (1: the own shard is copied into the own workspace too),
// race_debug.  `CallView` provides the call's ws (symmetric [M, K] workspace), flags and epoch.
#pragma once
#include <hip/hip_runtime.h>

#include "tdist/atomics.hpp"
#include "tdist/cache_policy.hpp"
#include "tdist/primitives.hpp"
#include "tdist/race_debug.hpp"

namespace tdist {

// Producer work of workgroup `pw` (of `p.push_wgs`): the shard is a run of 64 KiB batches (chunk after chunk, the last
// batch of a chunk may be short) and workgroup pw copies every push_wgs-th of them to every peer's workspace.
// Batches are dealt round-robin, so at any moment the producers read and write ONE contiguous window of
// push_wgs x 64 KiB that sweeps through the shard: every HBM channel is busy.  (Contiguous per-workgroup slices -
// workgroup k owning bytes [k, k + 1) * chunk / push_wgs - put all producers 128 KiB apart at every instant; that
// stride is a multiple of the channel interleave and the write-through stores of 24 - 64 producers moved only
// 270 - 430 GB/s chip-wide, no faster with more of them.)  The deal runs over the whole shard, not chunk by
// chunk: a chunk of 2 MiB has only 32 batches, and a producer that drained and synchronised after its single
// batch of every chunk never had a load in flight behind its stores.
// Each thread keeps two batches of eight 16-byte pieces in flight (the next batch is loaded before the
// current one is stored); every piece is loaded once and stored w-1 times with write-through stores, peers
// rotated per workgroup so all xGMI links carry traffic at any moment.  A producer counts itself on a chunk once
// all its batches up to and including that chunk are stored and drained (also when it had none in it); the
// workgroup that completes a chunk raises that chunk's flag on every peer.  Chunks complete in order, so the
// consumers of every peer can start on chunk 0 while chunk 1 is still on the wire.
template <class Params, class CallView>
__device__ __forceinline__ void push_item(const Params& p, const CallView& cv, int pw) {
  constexpr int U = 8;
  const int STEP = U * (int)blockDim.x;                                    // pieces per batch
  const int w = p.world, rank = p.rank;
  const int cpr = p.M_per_rank / p.rows_per_chunk;
  const long long row_bytes = (long long)p.K * 2;
  const long long chunk_bytes = (long long)p.rows_per_chunk * row_bytes;   // multiple of 16 (K % 64 == 0)
  const int n16 = (int)(chunk_bytes >> 4);
  const int nb = (n16 + STEP - 1) / STEP;                                  // batches in a chunk
  const int total = cpr * nb;
  const int pieces_per_row = (int)(row_bytes >> 4);
  const int first_peer = p.straddle ? 0 : 1;
  const int P = p.push_wgs;
  auto load_batch = [&](uint4(&v)[U], int g) {
    const int chunk = g / nb, b = g - chunk * nb;
    const long long src_row = (long long)chunk * p.rows_per_chunk;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int i = b * STEP + u * (int)blockDim.x + (int)threadIdx.x;
      if (i < n16) {
        // the local shard may be strided (lda_local >= K): address it row by row
        const int r = i / pieces_per_row;
        v[u] = *(const uint4*)((const char*)(p.a_local + (src_row + r) * p.lda_local) +
                               ((long long)(i - r * pieces_per_row) << 4));
      }
    }
  };
  auto store_batch = [&](const uint4(&v)[U], int g) {
    const int chunk = g / nb, b = g - chunk * nb;
    const long long dst_off = ((long long)rank * p.M_per_rank + (long long)chunk * p.rows_per_chunk) * row_bytes;
    for (int j = first_peer; j < w; ++j) {
      const int peer = j == 0 ? rank : (rank + 1 + (j - 1 + pw) % (w - 1)) % w;
      char* dst = (char*)(peer == rank ? cv.ws : symm_at(p.ctx, cv.ws, peer)) + dst_off;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = b * STEP + u * (int)blockDim.x + (int)threadIdx.x;
        if (i < n16) {
          if (p.push_store == 0)
            st_wt128_sys(dst + ((long long)i << 4), v[u]);
          else if (p.push_store == 1)
            st_wt128_agent(dst + ((long long)i << 4), v[u]);
          else
            *(uint4*)(dst + ((long long)i << 4)) = v[u];
        }
      }
    }
  };
  int counted = 0;                       // chunks [0, counted) carry this producer's count
  auto count_chunks = [&](int upto) {    // (uniform over the workgroup)
    if (upto <= counted) return;
    if (p.push_store == 2) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");     // plain stores: system-scope release
    drain_stores();
    race_debug_delay(p.race_debug, 2 * counted + 1);
    __syncthreads();
    if (threadIdx.x == 0) {
      for (int chunk = counted; chunk < upto; ++chunk) {
        // every workgroup's stores completed before its own add; the completing add is ordered after
        // all earlier adds on the word, so the flags follow all of the chunk's data
        const int prev = atom_add_relaxed_agent(p.push_done + chunk, 1);
        if (prev == p.push_wgs - 1) {
          for (int j = first_peer; j < w; ++j) {
            const int peer = (rank + j) % w;
            int* f = (peer == rank ? cv.flags : symm_at(p.ctx, cv.flags, peer)) + rank * cpr + chunk;
            store_relaxed_system(f, cv.epoch);
          }
        }
      }
    }
    counted = upto;
  };
  race_debug_delay(p.race_debug, 0);
  uint4 va[U], vb[U];
  if (pw < total) load_batch(va, pw);
  for (int g = pw; g < total; g += 2 * P) {
    const bool second = g + P < total;
    if (second) load_batch(vb, g + P);
    store_batch(va, g);
    int last = g;
    if (second) {
      if (g + 2 * P < total) load_batch(va, g + 2 * P);
      store_batch(vb, g + P);
      last = g + P;
    }
    // every chunk below the one of this producer's next batch is complete as far as it is concerned
    const int next = last + P;
    count_chunks(next < total ? next / nb : cpr);
  }
  count_chunks(cpr);
}

}  // namespace tdist
