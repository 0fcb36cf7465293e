// Tutorials 07 / 09 as a user kernel: all-gather overlapped with a GEMM.
//   C[M, N] = AllGather_M(A_local[M/w, K]) @ B[N, K]^T      (bf16, fp32 accumulate; B is this rank's weight shard)
//
//   ws    : symmetric [M, K] bf16  - the gathered A; rows [s * M/w, (s+1) * M/w) come from rank s
//   flags : symmetric [w * cpr]    - one per (source rank, 256-row chunk): == epoch when that chunk has landed
//
// Grid = push_wgs + tiles workgroups of 512 threads (the tile engine's shape, tdist/gemm_gfx950.hpp):
//   * workgroups [0, push_wgs) are the producers: chunk by chunk they store this rank's rows into EVERY rank's
//     workspace (the own one too) with write-through stores (tdist/cache_policy.hpp - the producers sit on different
//     XCDs with private L2s, so nothing may stay dirty in an L2 behind a counter another workgroup completes), and
//     the producer that completes a chunk notifies every rank's flag for it;
//   * workgroup push_wgs + t computes one 256 x 256 tile of C.  Row blocks are visited starting with this rank's own
//     rows (they arrive first), and a tile waits only for the chunk flag of its row block - so the first tiles start
//     at once and the transfer of the remaining chunks hides behind them.
// The wait sits directly in front of the tile engine's main loop (mainloop_256_2ph: LDS-DMA staging + MFMA); its
// token is threaded into the operand pointer with consume_token so no load of A can move above it.
//
// A production kernel (csrc/ops/ag_gemm.hip) is persistent - a bounded grid that draws tiles from per-XCD queues -
// because workgroups that spin on flags must all be resident together with the producers they wait for.  Here the
// grid is small enough for that by construction (the launcher checks it).  Reference:
// tutorials/07-overlapping-allgather-gemm.py:76-214, :307-440 (persistent consumer GEMM, copy-engine producer).
#include "tdist/cache_policy.hpp"
#include "tdist/gemm_gfx950.hpp"
#include "tdist/primitives.hpp"

using namespace tdist;
using namespace tdist::gemm;

extern "C" __global__ __launch_bounds__(512, 2) void ag_gemm_user(SymmCtx ctx, const uint16_t* a_local,
                                                                   const uint16_t* B, uint16_t* C, uint16_t* ws,
                                                                   int* flags, int* done, int M, int N, int K,
                                                                   int push_wgs, int epoch, int* err) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int w = num_ranks(ctx), me = rank(ctx);
  const int m_per_rank = M / w, cpr = m_per_rank / BM;          // chunks (= row blocks) per rank
  if ((int)blockIdx.x < push_wgs) {
    const long long chunk16 = (long long)BM * K / 8;            // 16-byte pieces per chunk
    for (int c = 0; c < cpr; ++c) {
      const uint4* src = (const uint4*)(a_local + (long long)c * BM * K);
      for (int j = 0; j < w; ++j) {
        const int peer = (me + j) % w;
        uint4* dst = (uint4*)(symm_at(ctx, ws, peer) + ((long long)me * m_per_rank + (long long)c * BM) * K);
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < chunk16; i += (long long)push_wgs * blockDim.x)
          st_wt128_sys(dst + i, src[i]);
      }
      drain_stores();
      __syncthreads();
      if (threadIdx.x == 0 && atomicAdd(done + c, 1) == push_wgs - 1) {
        done[c] = 0;
        for (int j = 0; j < w; ++j) notify(ctx, flags + me * cpr + c, (me + j) % w, epoch, SIGNAL_SET, COMM_INTRA_NODE);
      }
      __syncthreads();
    }
    return;
  }
  const int num_pid_n = (N + BN - 1) / BN;
  const int t = blockIdx.x - push_wgs;
  const int pid_n = t % num_pid_n;
  const int visit = t / num_pid_n;                              // 0 .. M / 256 - 1, own rows first
  const int src_rank = (me + visit / cpr) % w, chunk = visit % cpr;
  const int pid_m = src_rank * cpr + chunk;
  const int token = wait(flags + src_rank * cpr + chunk, 1, SCOPE_SYS, SEM_ACQUIRE, epoch, err);
  TileArgs ta;
  ta.A = consume_token(ws, token) + (long long)pid_m * BM * K;
  ta.B = B + (long long)pid_n * BN * K;
  ta.lda = ta.ldb = K;
  ta.a_rows = M - pid_m * BM;
  ta.b_rows = N - pid_n * BN;
  ta.K = K;
  Acc<DT_BF16> acc;
  mainloop_256_2ph<DT_BF16, true, 3>(ta, smem, acc);            // (PERM_B: accumulators laid out for the direct epilogue)
  const int nrem = N - pid_n * BN;
  uint16_t* cbase = C + (long long)pid_m * BM * N + pid_n * BN;
  epilogue_256_direct<true>(acc, [&](int r, int c, const uint4& v) {
    if (c < nrem) *(uint4*)(cbase + (long long)r * N + c) = v;
  });
}
