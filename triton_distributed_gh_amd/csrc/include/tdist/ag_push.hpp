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

// Producer work of workgroup `pw` (of `p.push_wgs`): for every chunk in order, copy slice pw of the chunk
// to every peer's workspace.  Each thread keeps two batches of eight 16-byte pieces in flight (the next
// batch is loaded before the current one is stored); every piece is loaded once and stored w-1 times
// with write-through stores, peers rotated per workgroup so all xGMI links carry traffic at any moment.
// The workgroup that completes a chunk raises that chunk's flag on every peer.  Chunks go out in order so
// the consumers of every peer can start on chunk 0 while chunk 1 is still on the wire.
template <class Params, class CallView>
__device__ __forceinline__ void push_item(const Params& p, const CallView& cv, int pw) {
  constexpr int U = 8;
  const int STEP = U * (int)blockDim.x;
  const int w = p.world, rank = p.rank;
  const int cpr = p.M_per_rank / p.rows_per_chunk;
  const long long row_bytes = (long long)p.K * 2;
  const long long chunk_bytes = (long long)p.rows_per_chunk * row_bytes;   // multiple of 16 (K % 64 == 0)
  const long long per = (((chunk_bytes + p.push_wgs - 1) / p.push_wgs) + 15) & ~15ll;
  const long long b0 = min((long long)pw * per, chunk_bytes);
  const long long b1 = min(b0 + per, chunk_bytes);
  const int n16 = (int)((b1 - b0) >> 4);
  const int first_piece = (int)(b0 >> 4);
  const int pieces_per_row = (int)(row_bytes >> 4);
  const int first_peer = p.straddle ? 0 : 1;
  for (int chunk = 0; chunk < cpr; ++chunk) {
    race_debug_delay(p.race_debug, 2 * chunk);
    const long long src_row = (long long)chunk * p.rows_per_chunk;
    const long long dst_off = ((long long)rank * p.M_per_rank + src_row) * row_bytes + b0;
    auto load_batch = [&](uint4(&v)[U], int base) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int i = base + u * (int)blockDim.x + (int)threadIdx.x;
        if (i < n16) {
          // the local shard may be strided (lda_local >= K): address it row by row
          const int q = first_piece + i, r = q / pieces_per_row;
          v[u] = *(const uint4*)((const char*)(p.a_local + (src_row + r) * p.lda_local) +
                                 ((long long)(q - r * pieces_per_row) << 4));
        }
      }
    };
    auto store_batch = [&](const uint4(&v)[U], int base) {
      for (int j = first_peer; j < w; ++j) {
        const int peer = j == 0 ? rank : (rank + 1 + (j - 1 + pw) % (w - 1)) % w;
        char* dst = (char*)(peer == rank ? cv.ws : symm_at(p.ctx, cv.ws, peer)) + dst_off;
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const int i = base + u * (int)blockDim.x + (int)threadIdx.x;
          if (i < n16) st_wt128_sys(dst + ((long long)i << 4), v[u]);
        }
      }
    };
    uint4 va[U], vb[U];
    if (n16 > 0) load_batch(va, 0);
    for (int base = 0; base < n16; base += 2 * STEP) {
      const bool second = base + STEP < n16;
      if (second) load_batch(vb, base + STEP);
      store_batch(va, base);
      if (second) {
        if (base + 2 * STEP < n16) load_batch(va, base + 2 * STEP);
        store_batch(vb, base + STEP);
      }
    }
    drain_stores();
    race_debug_delay(p.race_debug, 2 * chunk + 1);
    __syncthreads();
    if (threadIdx.x == 0) {
      // every slice's stores completed before its own add; the completing add is ordered after
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
}

}  // namespace tdist
