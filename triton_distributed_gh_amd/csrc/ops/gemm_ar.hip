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
#include "tdist/call_state.hpp"
#include "tdist/common.hpp"
#include "tdist/gemm_gfx950.hpp"
#include "tdist/gemm_mid_gfx950.hpp"
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

// ---------------------------------------------------------------------------------------------------------
// Low-latency form for the shapes GEMM + AllReduce is used at in decode and short prefill (M <= 256 rows
// against a row-parallel weight: attention o_proj, MLP down_proj).  The 256x256 tile leaves most of the
// chip idle there (M = 128, N = 5120 is 20 tiles), so this form runs the mid-M tile body
// (tdist/gemm_mid_gfx950.hpp: blocks of 16*MT rows x 64*WN columns, weights streamed through LDS, split-K
// across workgroups) and reduces two-shot at block granularity:
//   phase 1  every rank computes every block's partial (split-K joined locally by the last split to
//            arrive) and stores it, 16-bit in register layout (one coalesced 512-byte run per wave and
//            accumulator tile), into slot [source rank][b / w] of the block's OWNER b % w, then raises
//            flag1[b / w][source] there.  Nobody waits in this phase.
//   phase 2  a rank's workgroups share out the blocks it owns: wait for the w partials, sum them in fp32 in
//            rank order, write the finished block into EVERY rank's output and raise flag2[b] everywhere.
//   phase 3  before leaving, every workgroup waits for all flag2 words: kernel completion = output complete.
// Workgroups are persistent over a static item list (item i of workgroup g = g + i * grid), the same on
// every rank, so every wait of phase 2 is for data whose producers never wait before producing it.
// Calls are numbered on the device (tdist/call_state.hpp): buffers alternate by call parity, flags carry the
// call's epoch, and the launch is identical every time - it can be captured into a HIP graph and replayed.
//
// Role in the reference: `low_latency_gemm_allreduce_op` / the 32x64x256 low-latency configuration
// (kernels/nvidia/gemm_allreduce.py:509-543, layers/nvidia/gemm_allreduce_layer.py:76-93).
struct GemmArLLParams {
  SymmCtx ctx;
  const uint16_t* A;     // [M, K_local]
  const uint16_t* B;     // [N, K_local]
  long long lda, ldb;
  int M, N, K, splits;
  uint16_t *slots0, *slots1;   // symmetric: [w][max_slots][block elements], register layout
  uint16_t *out0, *out1;       // symmetric: [max_M, N]
  int *flags0, *flags1;        // symmetric (signal arena): flag1 [max_slots][w], then flag2 [max_blocks]
  int max_slots;
  float* partial;              // local split-K workspace
  int* counters;               // local split-K arrival counters (zero at launch, left zero)
  int* err;
  int* state;                  // [0] call number, [1] finished-workgroup counter
  uint16_t* c_local;           // optional [M, N] plain tensor that receives a copy of the result (its address
                               // does not alternate with the call parity: what a captured graph's consumers read)
};

template <bool IS_BF16>
__device__ __forceinline__ void add16x4(f32x4& acc, const uint2& v) {
  if (IS_BF16) {
    acc[0] += __builtin_bit_cast(float, v.x << 16);
    acc[1] += __builtin_bit_cast(float, v.x & 0xffff0000u);
    acc[2] += __builtin_bit_cast(float, v.y << 16);
    acc[3] += __builtin_bit_cast(float, v.y & 0xffff0000u);
  } else {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
    const f16x2 a = __builtin_bit_cast(f16x2, v.x), b = __builtin_bit_cast(f16x2, v.y);
    acc[0] += (float)a[0];
    acc[1] += (float)a[1];
    acc[2] += (float)b[0];
    acc[3] += (float)b[1];
  }
}

template <bool IS_BF16, int MT, int WN>
__global__ __launch_bounds__(MID_THREADS, 2) void gemm_ar_ll_kernel(GemmArLLParams p) {
  using F = Frag<IS_BF16>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_arrived, s_call;
  constexpr int BLOCK = 16 * MT * 64 * WN;               // elements of a block
  constexpr int PER_SPLIT = 4 * MT * WN * 64;            // f32x4 of one split-K partial
  const int w = p.ctx.num_ranks, rank = p.ctx.rank;
  const int tid = threadIdx.x, wv = tid >> 6, l = tid & 63;
  const int ncb = (p.N + 64 * WN - 1) / (64 * WN);
  const int passes = (p.M + 16 * MT - 1) / (16 * MT);
  const int nblocks = ncb * passes, S = p.splits;
  const int nst_total = p.K / 64;
  const int call = call_begin(p.state, &s_call);
  const int epoch = call_epoch(call);
  uint16_t* slots = call_parity(call) ? p.slots1 : p.slots0;
  uint16_t* out = call_parity(call) ? p.out1 : p.out0;
  int* flags = call_parity(call) ? p.flags1 : p.flags0;
  int* flag2 = flags + p.max_slots * w;
  // position of this lane's four values of accumulator tile (m, j) inside a block in register layout
  const int reg_base = (wv * MT * WN) * 64 + l;

  // ---- phase 1: partial blocks to their owners ------------------------------------------------------
  for (int item = blockIdx.x; item < nblocks * S; item += gridDim.x) {
    const int split = item / nblocks, b = item - split * nblocks;
    const int pass = b / ncb, cb = b - pass * ncb;
    const int st0 = (int)((long long)nst_total * split / S), st1 = (int)((long long)nst_total * (split + 1) / S);
    f32x4 acc[MT][WN];
    mid_mainloop<IS_BF16, MT, WN>(p.A, p.lda, p.B, p.ldb, p.M, p.N, pass * 16 * MT, cb, st0, st1 - st0, smem, acc);
    if (S > 1) {
      if (!mid_splitk_join<MT, WN>(acc, (f32x4*)p.partial + (long long)b * S * PER_SPLIT, p.counters + b, split, S,
                                   &s_arrived))
        continue;                       // (the join ends on a barrier: LDS may be restaged)
    }
    const int owner = b % w, slot = b / w;
    uint16_t* dst = (owner == rank ? slots : symm_at(p.ctx, slots, owner)) +
                    ((long long)rank * p.max_slots + slot) * BLOCK;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        uint2 v;
        v.x = F::pack2(acc[m][j][0], acc[m][j][1]);
        v.y = F::pack2(acc[m][j][2], acc[m][j][3]);
        st_wt64_sys((uint2*)dst + reg_base + (m * WN + j) * 64, v);
      }
    drain_stores();
    __syncthreads();
    if (tid == 0) {
      int* f1 = (owner == rank ? flags : symm_at(p.ctx, flags, owner)) + slot * w + rank;
      store_relaxed_system(f1, epoch);
    }
  }
  // ---- phase 2: owned blocks: sum and broadcast -------------------------------------------------------
  const int n_owned = nblocks > rank ? (nblocks - rank + w - 1) / w : 0;
  for (int o = blockIdx.x; o < n_owned; o += gridDim.x) {
    const int b = rank + o * w;
    {
      int failed = 0;
      if (tid < w) {
        const bool dead = load_relaxed_agent(p.err) != 0;
        if (dead || !spin_until_eq(flags + o * w + tid, epoch, SCOPE_SYS)) failed = 1;
        if (failed && !dead) atomicOr(p.err, 1);
      }
      if (tid < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
    f32x4 acc[MT][WN];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < WN; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < w; ++s) {
      const uint2* src = (const uint2*)(slots + ((long long)s * p.max_slots + o) * BLOCK) + reg_base;
      uint2 v[MT][WN];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < WN; ++j) v[m][j] = src[(m * WN + j) * 64];
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < WN; ++j) add16x4<IS_BF16>(acc[m][j], v[m][j]);
    }
    const int pass = b / ncb, cb = b - pass * ncb;
    for (int i = 0; i < w; ++i) {
      const int peer = (rank + i) % w;
      uint16_t* dst = i == 0 ? out : symm_at(p.ctx, out, peer);
      mid_store<IS_BF16, MT, WN>(acc, pass * 16 * MT, cb, [&](int row, int col, const uint2& v) {
        if (row < p.M && col < p.N) st_wt64_sys(dst + (long long)row * p.N + col, v);
      });
    }
    drain_stores();
    __syncthreads();
    if (tid < w) {
      int* f2 = (tid == rank ? flag2 : symm_at(p.ctx, flag2, tid)) + b;
      store_relaxed_system(f2, epoch);
    }
  }
  // ---- phase 3: the whole output has landed here ---------------------------------------------------------
  {
    int failed = 0;
    bool dead = false;
    for (int b = tid; b < nblocks; b += MID_THREADS) {
      if (load_relaxed_system(flag2 + b) == epoch) continue;
      dead = dead || load_relaxed_agent(p.err) != 0;
      if (dead || !spin_until_eq(flag2 + b, epoch, SCOPE_SYS)) failed = 1;
    }
    if (failed && !dead) atomicOr(p.err, 1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (p.c_local) {
    const long long n16 = (long long)p.M * p.N / 8;
    const uint4* src = (const uint4*)out;
    uint4* dst = (uint4*)p.c_local;
    for (long long i = (long long)blockIdx.x * MID_THREADS + tid; i < n16; i += (long long)gridDim.x * MID_THREADS)
      dst[i] = src[i];
  }
  call_end(p.state, call);
}

namespace {

template <bool IS_BF16, int MT, int WN>
int launch_ar_ll(const GemmArLLParams& p, int grid, hipStream_t s) {
  constexpr int LDS = mid_lds_bytes(MT, WN);
  static bool attr_set = false;
  if (!attr_set) {
    TDIST_HIP_TRY(hipFuncSetAttribute((const void*)gemm_ar_ll_kernel<IS_BF16, MT, WN>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_ar_ll_kernel<IS_BF16, MT, WN>), dim3(grid), dim3(MID_THREADS), LDS, s, p);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

template <bool IS_BF16>
int dispatch_ar_ll(const GemmArLLParams& p, int mt, int wn, int grid, hipStream_t s) {
  if (mt == 2) return wn == 1 ? launch_ar_ll<IS_BF16, 2, 1>(p, grid, s) : launch_ar_ll<IS_BF16, 2, 2>(p, grid, s);
  if (mt == 4) return wn == 1 ? launch_ar_ll<IS_BF16, 4, 1>(p, grid, s) : launch_ar_ll<IS_BF16, 4, 2>(p, grid, s);
  return wn == 1 ? launch_ar_ll<IS_BF16, 8, 1>(p, grid, s) : launch_ar_ll<IS_BF16, 8, 2>(p, grid, s);
}

}  // namespace

// Sizes a context needs for problems up to [max_M, N]: elements per slot block (the largest block of any
// plan), slots per source rank, blocks (the narrowest plan gives the most), split-K workspace bytes and
// arrival counters.  Split-K only cuts K while splits * blocks fits the grid, so the workspace is bounded by
// (workgroups) x (fp32 bytes of the largest block).
TDIST_API int tdist_gemm_ar_ll_sizes(int max_M, int N, int world, int* block_elems, int* max_slots, int* max_blocks,
                                     long long* partial_bytes, int* n_counters) {
  TDIST_REQUIRE(max_M > 0 && N > 0 && world >= 1, "bad shape");
  const int blocks = ((max_M + 31) / 32) * ((N + 63) / 64);
  *block_elems = 128 * 128;
  *max_blocks = blocks;
  *max_slots = (blocks + world - 1) / world;
  *partial_bytes = 512ll * 128 * 128 * 4;
  *n_counters = blocks;
  return 0;
}

// dtype: 0 = bf16, 1 = fp16.  `grid_budget`: workgroups this rank may keep resident (0 = two per CU).
// Every symmetric pointer is this rank's address; parity 0 and 1 objects must both be given.
TDIST_API int tdist_gemm_ar_ll(const tdist::SymmCtx* ctx, const void* A, const void* B, void* slots0, void* slots1,
                               void* out0, void* out1, int* flags0, int* flags1, int max_slots, int max_blocks,
                               long long slot_block_elems, void* partial, long long partial_bytes, int* counters,
                               int n_counters, int* err, int* state, void* c_local, long long lda, long long ldb, int M,
                               int N, int K, int grid_budget, int force_wn, int force_splits, int dtype, void* stream) {
  TDIST_REQUIRE(K >= 128 && K % 64 == 0, "K must be a multiple of 64 and >= 128");
  TDIST_REQUIRE(N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "N/lda/ldb must be multiples of 8");
  TDIST_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)out0 | (uintptr_t)out1 | (uintptr_t)slots0 |
                  (uintptr_t)slots1) & 15) == 0, "operands must be 16-byte aligned");
  TDIST_REQUIRE(M > 0 && N > 0 && state && err, "bad arguments");
  static int num_cus = 0;
  if (!num_cus) {
    int dev = 0;
    TDIST_HIP_TRY(hipGetDevice(&dev));
    TDIST_HIP_TRY(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev));
  }
  const int budget = grid_budget > 0 ? grid_budget : num_cus;
  const MidPlan pl = mid_plan(M, N, K, budget, force_wn, force_splits);
  const int nblocks = pl.ncb * pl.passes;
  TDIST_REQUIRE(nblocks <= max_blocks && (nblocks + ctx->num_ranks - 1) / ctx->num_ranks <= max_slots,
                "problem exceeds the context's block capacity");
  TDIST_REQUIRE(16 * pl.mt * 64 * pl.wn <= slot_block_elems, "slot blocks too small for this plan");
  TDIST_REQUIRE(pl.splits == 1 || (partial && pl.partial_bytes <= partial_bytes && counters && nblocks <= n_counters),
                "split-K workspace too small");
  GemmArLLParams p;
  p.ctx = *ctx;
  p.A = (const uint16_t*)A;
  p.B = (const uint16_t*)B;
  p.lda = lda;
  p.ldb = ldb;
  p.M = M;
  p.N = N;
  p.K = K;
  p.splits = pl.splits;
  p.slots0 = (uint16_t*)slots0;
  p.slots1 = (uint16_t*)slots1;
  p.out0 = (uint16_t*)out0;
  p.out1 = (uint16_t*)out1;
  p.flags0 = flags0;
  p.flags1 = flags1;
  p.max_slots = max_slots;
  p.partial = (float*)partial;
  p.counters = counters;
  p.err = err;
  p.state = state;
  p.c_local = (uint16_t*)c_local;
  TDIST_REQUIRE((((uintptr_t)c_local) & 15) == 0, "c_local must be 16-byte aligned");
  // the whole grid must be resident (phase 2 spins): at most `budget` workgroups, never more than the items
  int grid = nblocks * pl.splits;
  if (grid > budget) grid = budget;
  return dtype == 0 ? dispatch_ar_ll<true>(p, pl.mt, pl.wn, grid, (hipStream_t)stream)
                    : dispatch_ar_ll<false>(p, pl.mt, pl.wn, grid, (hipStream_t)stream);
}
