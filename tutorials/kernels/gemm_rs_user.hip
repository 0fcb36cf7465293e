// Tutorials 08 / 10 as a user kernel: GEMM overlapped with a reduce-scatter.
//   out[M/w, N] = ReduceScatter_M( A[M, K/w] @ B[N, K/w]^T )      (bf16, fp32 accumulate)
//
//   scatter : symmetric [w][M/w, N] bf16 - slot s of rank d receives rank s's partial of d's rows
//   flags   : symmetric [w] ints         - flags[s] == epoch when ALL tiles of slot s have landed
//   done    : local [w] ints, zero between launches
//
// Grid = tiles + R workgroups of 512 threads.
//   * workgroup t < tiles computes one 256 x 256 tile of this rank's partial product and stores it from the
//     epilogue STRAIGHT into the owner's scatter slot [rank] over xGMI (peer-store epilogue: the partial never
//     exists in local memory; write-through stores, tdist/cache_policy.hpp, because the tile that finally notifies
//     an owner runs on another XCD than most of the tiles it vouches for).  Row blocks are visited starting with the right neighbour's rows, own rows last, so
//     all links carry traffic from the first tile on.  The tile that completes an owner's rows notifies it.
//   * the last R workgroups wait for all w flags and sum the w slots into `out` (bf16 partials, fp32 sum).
// Tiles never wait, so the reducing workgroups can only be waiting for other ranks.  Reference:
// tutorials/08-overlapping-gemm-reduce-scatter.py:136-352 (GEMM notifies per tile, scatter + ring reduce on
// other streams); csrc/ops/gemm_rs.hip is the persistent production form with write-through stores.
#include "tdist/cache_policy.hpp"
#include "tdist/gemm_gfx950.hpp"
#include "tdist/primitives.hpp"

using namespace tdist;
using namespace tdist::gemm;

extern "C" __global__ __launch_bounds__(512, 2) void gemm_rs_user(SymmCtx ctx, const uint16_t* A, const uint16_t* B,
                                                                   uint16_t* scatter, int* flags, int* done,
                                                                   uint16_t* out, int M, int N, int K, int tiles,
                                                                   int epoch, int* err) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int w = num_ranks(ctx), me = rank(ctx);
  const int m_per_rank = M / w, bpr = m_per_rank / BM;          // row blocks per owner
  const int num_pid_n = (N + BN - 1) / BN;
  if ((int)blockIdx.x < tiles) {
    const int t = blockIdx.x;
    const int pid_n = t % num_pid_n, visit = t / num_pid_n;
    const int owner = (me + 1 + visit / bpr) % w, blk = visit % bpr;
    const int pid_m = owner * bpr + blk;
    TileArgs ta;
    ta.A = A + (long long)pid_m * BM * K;
    ta.B = B + (long long)pid_n * BN * K;
    ta.lda = ta.ldb = K;
    ta.a_rows = M - pid_m * BM;
    ta.b_rows = N - pid_n * BN;
    ta.K = K;
    Acc<DT_BF16> acc;
    mainloop_256_2ph<DT_BF16, true, 3>(ta, smem, acc);
    const int nrem = N - pid_n * BN;
    uint16_t* dst = symm_at(ctx, scatter, owner) + ((long long)me * m_per_rank + (long long)blk * BM) * N + pid_n * BN;
    epilogue_256_direct<true>(acc, [&](int r, int c, const uint4& v) {
      if (c < nrem) st_wt128_sys(dst + (long long)r * N + c, v);
    });
    drain_stores();
    __syncthreads();
    if (threadIdx.x == 0 && atomicAdd(done + owner, 1) == bpr * num_pid_n - 1) {
      done[owner] = 0;
      notify(ctx, flags + me, owner, epoch, SIGNAL_SET, COMM_INTRA_NODE);
    }
    return;
  }
  const int token = wait(flags, w, SCOPE_SYS, SEM_ACQUIRE, epoch, err);
  const uint16_t* slots = consume_token(scatter, token);
  const int R = gridDim.x - tiles, r = blockIdx.x - tiles;
  const long long n8 = (long long)m_per_rank * N / 8;           // 16-byte pieces of the output
  for (long long i = (long long)r * blockDim.x + threadIdx.x; i < n8; i += (long long)R * blockDim.x) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < w; ++s) {
      const uint4 v = *(const uint4*)(slots + (long long)s * m_per_rank * N + i * 8);
      const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[2 * k] += __builtin_bit_cast(float, x[k] << 16);
        acc[2 * k + 1] += __builtin_bit_cast(float, x[k] & 0xffff0000u);
      }
    }
    uint4 o;
    o.x = Frag<DT_BF16>::pack2(acc[0], acc[1]);
    o.y = Frag<DT_BF16>::pack2(acc[2], acc[3]);
    o.z = Frag<DT_BF16>::pack2(acc[4], acc[5]);
    o.w = Frag<DT_BF16>::pack2(acc[6], acc[7]);
    *(uint4*)(out + i * 8) = o;
  }
}
