// GEMM + AllReduce in one persistent kernel:  out[M, N] = sum over ranks of A_r[M, K_local] * B_r[N, K_local]^T.
//
// Two-shot over 256x256 output tiles: tile t is OWNED by rank t % w.
//   GEMM(t)   : every rank computes its partial tile and stores it as one contiguous 128 KiB block
//               into the owner's slot [src_rank][t / w] (remote store over xGMI, or local), then
//               raises flag1[t / w][src_rank] = epoch on the owner.
//   REDUCE(t) : the owner waits for the w partial blocks, sums them in fp32 (rank order, so the
//               result is bitwise identical everywhere), writes the tile into EVERY rank's
//               symmetric output [M, N] and raises flag2[t] = epoch on every rank.
//   DRAIN     : before leaving, each workgroup waits until every tile's flag2 has arrived, so
//               kernel completion means the full output is present locally.
// Work order: all GEMM tiles (owners visited round-robin, so each step loads all links evenly),
// then this rank's REDUCE items.  Epoch flags + call-parity double buffering: no reset, no barrier.
//
// Role in the reference: kernels/nvidia/gemm_allreduce.py - `kernel_fused_gemm_allreduce` :232-315
// with NUM_COMM_SMS communication CTAs doing `multimem.ld_reduce` (:123-207) and the LL
// double-buffered context (:73-91).  xGMI has no in-fabric reduction, hence owner-reduces-then-
// broadcasts; the low-latency entry point is the same kernel because every context is already
// double-buffered.
#include "tdist/cache_policy.hpp"
#include "tdist/common.hpp"
#include "tdist/gemm_gfx950.hpp"
#include "tdist/primitives.hpp"

using namespace tdist;
using namespace tdist::gemm;

struct GemmArParams {
  SymmCtx ctx;
  const uint16_t* A;     // [M, K_local]
  const uint16_t* B;     // [N, K_local]
  uint16_t* slots;       // symmetric, this parity: [w][tiles_per_owner][256*256]
  uint16_t* out;         // symmetric, this parity: [M, N]
  int* flags;            // symmetric (signal arena), this parity: flag1 [tiles_per_owner][w], then flag2 [tiles]
  int* ticket;           // local, zeroed before launch
  int* err;
  long long lda, ldb;
  int M, N, K;
  int epoch;
};

constexpr int AR_TICKET_SLOT = LDS_BYTES;
constexpr int TILE_ELEMS = BM * BN;

template <bool IS_BF16>
__device__ __forceinline__ void reduce_tile(const GemmArParams& p, int t, int tiles_per_owner, int total_tiles) {
  const int w = p.ctx.num_ranks, rank = p.ctx.rank;
  const int num_pid_n = (p.N + BN - 1) / BN;
  const int local = t / w;
  {
    const bool dead = load_relaxed_agent(p.err) != 0;
    int failed = 0;
    if ((int)threadIdx.x < w) {
      if (dead || !spin_until_eq(p.flags + local * w + threadIdx.x, p.epoch, SCOPE_SYS)) failed = 1;
    }
    if (failed && !dead) atomicOr(p.err, 1);
    if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  const int pid_m = t / num_pid_n, pid_n = t - pid_m * num_pid_n;
  const int rows = min(BM, p.M - pid_m * BM), cols = min(BN, p.N - pid_n * BN);
  const long long slot_stride = (long long)tiles_per_owner * TILE_ELEMS;
  const uint16_t* base = p.slots + (long long)local * TILE_ELEMS;
  // 32 threads per row (8 elements each), 16 rows per pass
  const int r0 = threadIdx.x >> 5, c8 = (threadIdx.x & 31) * 8;
  for (int r = r0; r < rows; r += 16) {
    if (c8 >= cols) continue;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < w; ++s) {
      const uint4 v = *(const uint4*)(base + s * slot_stride + r * BN + c8);
      const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        if (IS_BF16) {
          acc[2 * e] += __builtin_bit_cast(float, x[e] << 16);
          acc[2 * e + 1] += __builtin_bit_cast(float, x[e] & 0xffff0000u);
        } else {
          typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
          const f16x2 h = __builtin_bit_cast(f16x2, x[e]);
          acc[2 * e] += (float)h[0];
          acc[2 * e + 1] += (float)h[1];
        }
      }
    }
    uint4 o;
    o.x = Frag<IS_BF16>::pack2(acc[0], acc[1]);
    o.y = Frag<IS_BF16>::pack2(acc[2], acc[3]);
    o.z = Frag<IS_BF16>::pack2(acc[4], acc[5]);
    o.w = Frag<IS_BF16>::pack2(acc[6], acc[7]);
    const long long off = (long long)(pid_m * BM + r) * p.N + pid_n * BN + c8;
    for (int i = 0; i < w; ++i) {
      const int peer = (rank + i) % w;
      uint16_t* dst = (i == 0) ? p.out : symm_at(p.ctx, p.out, peer);
      st_wt128_sys(dst + off, o);
    }
  }
  // write-through stores: drain + barrier publish them, no release fence
  drain_stores();
  __syncthreads();
  if ((int)threadIdx.x < w) {
    int* f2 = ((int)threadIdx.x == rank ? p.flags : symm_at(p.ctx, p.flags, (int)threadIdx.x)) + tiles_per_owner * w + t;
    store_relaxed_system(f2, p.epoch);
  }
}

template <bool IS_BF16>
__global__ __launch_bounds__(THREADS, 2) void gemm_ar_kernel(GemmArParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  volatile int* s_ticket = (volatile int*)(smem + AR_TICKET_SLOT);
  const int w = p.ctx.num_ranks, rank = p.ctx.rank;
  const int num_pid_m = (p.M + BM - 1) / BM, num_pid_n = (p.N + BN - 1) / BN;
  const int tiles = num_pid_m * num_pid_n;
  const int tiles_per_owner = (tiles + w - 1) / w;
  const int owned = (tiles - rank + w - 1) / w;   // tiles t with t % w == rank
  const int total = tiles + owned;

  while (true) {
    const int item = fetch_ticket_block(p.ticket, s_ticket);
    if (item >= total) break;
    if (item >= tiles) {
      reduce_tile<IS_BF16>(p, (item - tiles) * w + rank, tiles_per_owner, tiles);
      continue;
    }
    // rotate so that at any moment different ranks are writing to different owners
    const int t = (item + rank) % tiles;
    const int owner = t % w, local = t / w;
    const int pid_m = t / num_pid_n, pid_n = t - pid_m * num_pid_n;
    const int rows = min(BM, p.M - pid_m * BM);
    TileArgs ta;
    ta.A = p.A + (long long)pid_m * BM * p.lda;
    ta.B = p.B + (long long)pid_n * BN * p.ldb;
    ta.lda = p.lda;
    ta.ldb = p.ldb;
    ta.a_rows = rows;
    ta.b_rows = p.N - pid_n * BN;
    ta.K = p.K;
    Acc<IS_BF16> acc;
    mainloop_256_2ph<IS_BF16>(ta, smem, acc);
    uint16_t* slot_base = (owner == rank) ? p.slots : symm_at(p.ctx, p.slots, owner);
    uint16_t* dst = slot_base + ((long long)rank * tiles_per_owner + local) * TILE_ELEMS;
    epilogue_256<IS_BF16>(acc, smem, [&](int r, int c, const uint4& v) {
      if (r < rows) st_wt128_sys(dst + r * BN + c, v);
    });
    drain_stores();
    __syncthreads();
    if (threadIdx.x == 0) {
      int* f1 = ((owner == rank) ? p.flags : symm_at(p.ctx, p.flags, owner)) + local * w + rank;
      store_relaxed_system(f1, p.epoch);
    }
  }
  // DRAIN: all tiles' results must have landed locally before the kernel completes
  {
    const int* f2 = p.flags + tiles_per_owner * w;
    const bool dead = load_relaxed_agent(p.err) != 0;
    int failed = 0;
    for (int t = threadIdx.x; t < tiles; t += blockDim.x)
      if (dead || !spin_until_eq(f2 + t, p.epoch, SCOPE_SYS)) failed = 1;
    if (failed && !dead) atomicOr(p.err, 1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  }
}

TDIST_API int tdist_gemm_ar(const tdist::SymmCtx* ctx, const void* A, const void* B, void* slots, void* out, int* flags,
                            int* ticket, int* err, long long lda, long long ldb, int M, int N, int K, int epoch,
                            int num_ctas, int dtype, void* stream) {
  TDIST_REQUIRE(K >= 128 && K % 64 == 0, "K must be a multiple of 64 and >= 128");
  TDIST_REQUIRE(N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "N/lda/ldb must be multiples of 8");
  TDIST_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)out | (uintptr_t)slots) & 15) == 0,
                "operands must be 16-byte aligned");
  TDIST_REQUIRE(M > 0 && N > 0, "empty problem");
  GemmArParams p;
  p.ctx = *ctx;
  p.A = (const uint16_t*)A;
  p.B = (const uint16_t*)B;
  p.slots = (uint16_t*)slots;
  p.out = (uint16_t*)out;
  p.flags = flags;
  p.ticket = ticket;
  p.err = err;
  p.lda = lda;
  p.ldb = ldb;
  p.M = M;
  p.N = N;
  p.K = K;
  p.epoch = epoch;
  static bool attr_set = false;
  if (!attr_set) {
    TDIST_HIP_TRY(hipFuncSetAttribute((const void*)gemm_ar_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      LDS_BYTES + 16));
    TDIST_HIP_TRY(hipFuncSetAttribute((const void*)gemm_ar_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      LDS_BYTES + 16));
    attr_set = true;
  }
  TDIST_HIP_TRY(hipMemsetAsync(ticket, 0, 16, (hipStream_t)stream));
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  int grid = num_ctas > 0 ? num_ctas : 256;
  const int work = tiles + (tiles + ctx->num_ranks - 1) / ctx->num_ranks;
  if (grid > work) grid = work;
  if (dtype == 0)
    hipLaunchKernelGGL(gemm_ar_kernel<true>, dim3(grid), dim3(THREADS), LDS_BYTES + 16, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(gemm_ar_kernel<false>, dim3(grid), dim3(THREADS), LDS_BYTES + 16, (hipStream_t)stream, p);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
