// GEMM + AllReduce in one persistent kernel:  out[M, N] = sum over ranks of A_r[M, K_local] * B_r[N, K_local]^T.
//
// Two-shot over 256x256 output tiles: tile t is OWNED by rank t % w.
//   GEMM(t)      : every rank computes its partial tile and stores it as one contiguous 128 KiB block into the
//                  owner's slot [src_rank][t / w] (remote store over xGMI, or local), then raises
//                  flag1[t / w][src_rank] = epoch on the owner.
//   REDUCE(t, s) : the owner waits for the w partial blocks, sums 32-row slab s of them in fp32 (rank order, so
//                  the result is bitwise identical everywhere), writes the slab into EVERY rank's symmetric
//                  output [M, N] and raises flag2[t][s] = epoch on every rank.  Eight slabs per tile: a round of
//                  reductions is as many short items as the grid has workgroups, not w-times fewer long ones.
//   DRAIN        : before leaving, each workgroup waits until every flag2 has arrived, so kernel completion means
//                  the full output is present locally - and that every owner has finished reading its slots and
//                  writing this call's output everywhere.  The output buffer therefore needs no parity: a peer's
//                  NEXT call can only write into it after that peer has received this rank's next partials, which
//                  this rank's stream issues behind its consumers of the current output.  The result's address is
//                  the same every call (what a captured graph's consumers need); a private copy is optional.
// Tile: 256 x 256, or 256 x 128 when the 256 x 256 tiles would leave half of the workgroups without a GEMM item.
// Work order (the GEMM + RS order): tiles are cut into groups of about one grid round; every one of the eight
// per-XCD queues (tdist/work_queue.hpp) runs  GEMM(g0) GEMM(g1) REDUCE(g0) GEMM(g2) REDUCE(g1) ... REDUCE(last),
// so the reduction and broadcast of a group ride behind the GEMM of the next one instead of all reductions
// waiting at the end, and a queue owns a contiguous run of a group's tiles (shared A row blocks in that XCD's L2).
// Every rank runs the same order, and an item waits only for items earlier in it: no deadlock at any grid size
// that is resident at once.
// Calls are numbered on the device (tdist/call_state.hpp): slots / outputs / flags alternate by call parity, flags
// carry the call's epoch, the last workgroup out re-arms the queue tickets - the launch is ONE kernel node, the
// same every time, and can be captured into a HIP graph and replayed.
//
// Role in the reference: kernels/nvidia/gemm_allreduce.py - `kernel_fused_gemm_allreduce` :232-315
// with NUM_COMM_SMS communication CTAs doing `multimem.ld_reduce` (:123-207) as tiles' flags flip, and the LL
// double-buffered context (:73-91).  xGMI has no in-fabric reduction, hence owner-reduces-then-broadcasts, and
// instead of dedicated communication CTAs every workgroup alternates between the two kinds of item.
#include "tdist/cache_policy.hpp"
#include "tdist/call_state.hpp"
#include "tdist/common.hpp"
#include "tdist/gemm_gfx950.hpp"
#include "tdist/gemm_mid_gfx950.hpp"
#include "tdist/primitives.hpp"
#include "tdist/work_queue.hpp"

using namespace tdist;
using namespace tdist::gemm;

struct GemmArParams {
  SymmCtx ctx;
  const uint16_t* A;     // [M, K_local]
  const uint16_t* B;     // [N, K_local]
  uint16_t *slots0, *slots1;   // symmetric, by call parity: [w][tiles_per_owner][256*256]
  uint16_t* out;               // symmetric [M, N]: ONE buffer for both parities (see the DRAIN note above)
  int *flags0, *flags1;        // symmetric (signal arena), by call parity: flag1 [tiles_per_owner][w], then flag2 [tiles][8]
  int* counters;         // local: [0, 8) queue tickets, [8] finished-workgroup counter; zero at launch, left zero
  int* err;
  int* state;            // [0] call number (tdist/call_state.hpp)
  uint16_t* c_local;     // optional plain [M, N] copy of the result
  long long lda, ldb;
  int M, N, K;
  int group_tiles;       // tiles per group (multiple of 8)
};

constexpr int AR_SLABS = 8;            // reduce items per tile
constexpr int AR_SLAB_ROWS = BM / AR_SLABS;

struct ArCall {
  uint16_t *slots, *out;
  int* flags;
  int epoch;
};

// Slab `slab` of owned tile t: sum the w partials, broadcast, flag.  `acquired`: the tile whose partials this
// workgroup has already waited for (they do not change within the call).
template <bool IS_BF16, int NT>
__device__ __forceinline__ void reduce_slab(const GemmArParams& p, const ArCall& cv, int t, int slab, int tiles_per_owner,
                                            int& acquired) {
  constexpr int BN = 64 * NT, TILE_ELEMS = BM * BN;
  constexpr int TPR = BN / 8, RPP = THREADS / TPR;          // threads per row, rows per pass
  const int w = p.ctx.num_ranks, rank = p.ctx.rank;
  const int num_pid_n = (p.N + BN - 1) / BN;
  const int local = t / w;
  if (acquired != t) {
    const bool dead = load_relaxed_agent(p.err) != 0;
    int failed = 0;
    if ((int)threadIdx.x < w) {
      if (dead || !spin_until_eq(cv.flags + local * w + threadIdx.x, cv.epoch, SCOPE_SYS)) failed = 1;
    }
    if (failed && !dead) atomicOr(p.err, 1);
    if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (__syncthreads_or(failed) == 0) acquired = t;
  }
  const int pid_m = t / num_pid_n, pid_n = t - pid_m * num_pid_n;
  const int rows = min(BM, p.M - pid_m * BM), cols = min(BN, p.N - pid_n * BN);
  const long long slot_stride = (long long)tiles_per_owner * TILE_ELEMS;
  const uint16_t* base = cv.slots + (long long)local * TILE_ELEMS;
  // BN / 8 threads per row (8 elements each)
  const int c8 = (threadIdx.x % TPR) * 8;
#pragma unroll
  for (int pass = 0; pass < (AR_SLAB_ROWS + RPP - 1) / RPP; ++pass) {
    const int r = slab * AR_SLAB_ROWS + pass * RPP + (threadIdx.x / TPR);
    if (pass * RPP + (int)(threadIdx.x / TPR) >= AR_SLAB_ROWS) continue;
    if (r >= rows || c8 >= cols) continue;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < w; s0 += 4) {
      uint4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (s0 + u < w) v[u] = *(const uint4*)(base + (s0 + u) * slot_stride + r * BN + c8);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (s0 + u < w) {
          const uint32_t x[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
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
    }
    uint4 o;
    o.x = Frag<IS_BF16>::pack2(acc[0], acc[1]);
    o.y = Frag<IS_BF16>::pack2(acc[2], acc[3]);
    o.z = Frag<IS_BF16>::pack2(acc[4], acc[5]);
    o.w = Frag<IS_BF16>::pack2(acc[6], acc[7]);
    const long long off = (long long)(pid_m * BM + r) * p.N + pid_n * BN + c8;
    for (int i = 0; i < w; ++i) {
      const int peer = (rank + i) % w;
      uint16_t* dst = (i == 0) ? cv.out : symm_at(p.ctx, cv.out, peer);
      st_wt128_sys(dst + off, o);
    }
  }
  // write-through stores: drain + barrier publish them, no release fence
  drain_stores();
  __syncthreads();
  if ((int)threadIdx.x < w) {
    int* f2 = ((int)threadIdx.x == rank ? cv.flags : symm_at(p.ctx, cv.flags, (int)threadIdx.x)) + tiles_per_owner * w +
              t * AR_SLABS + slab;
    store_relaxed_system(f2, cv.epoch);
  }
}

template <bool IS_BF16, int NT>
__global__ __launch_bounds__(THREADS, 2) void gemm_ar_kernel(GemmArParams p) {
  using TS = TileShape<8, NT>;
  constexpr int BN = TS::BN, TILE_ELEMS = BM * BN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  volatile int* s_ticket = (volatile int*)(smem + staged_lds_bytes<8, NT>());
  const int w = p.ctx.num_ranks, rank = p.ctx.rank;
  const int num_pid_m = (p.M + BM - 1) / BM, num_pid_n = (p.N + BN - 1) / BN;
  const int tiles = num_pid_m * num_pid_n;
  const int tiles_per_owner = (tiles + w - 1) / w;
  const int G = p.group_tiles, g8 = G / 8;
  const int groups = (tiles + G - 1) / G;
  // tiles t < x owned by this rank
  auto owned_below = [=](int x) { return x > rank ? (x - rank + w - 1) / w : 0; };
  auto group_end = [=](int c) { return min((c + 1) * G, tiles); };
  // queue q of group c: GEMM items = tiles [c G + q g8, c G + (q + 1) g8) clipped to the group; REDUCE items =
  // every eighth (owned tile, slab) pair of the group
  auto gq = [=](int c, int q) { return min(max(group_end(c) - c * G - q * g8, 0), g8); };
  auto rq = [=](int c, int q) {
    const int n = (owned_below(group_end(c)) - owned_below(c * G)) * AR_SLABS;
    return n > q ? (n - q + 7) >> 3 : 0;
  };
  auto queue_size = [=](int q) {
    int n = 0;
    for (int c = 0; c < groups; ++c) n += gq(c, q) + rq(c, q);
    return n;
  };
  WorkQueues<decltype(queue_size)> wq(p.counters, s_ticket, 0, queue_size);
  const int call = call_begin(p.state, s_ticket + 2);
  ArCall cv;
  cv.slots = call_parity(call) ? p.slots1 : p.slots0;
  cv.out = p.out;
  cv.flags = call_parity(call) ? p.flags1 : p.flags0;
  cv.epoch = call_epoch(call);

  int item = 0, acquired = -1;
  bool have = wq.first(item);
  if (!have) have = wq.next(item);
  while (have) {
    const int q = wq.q;
    // decode: stage s runs GEMM(group s) then REDUCE(group s - 1)
    int rem = item, grp = 0, idx = 0;
    bool is_gemm = true;
    for (int s = 0; s <= groups; ++s) {
      if (s < groups) {
        const int g = gq(s, q);
        if (rem < g) { is_gemm = true; grp = s; idx = rem; break; }
        rem -= g;
      }
      if (s >= 1) {
        const int r = rq(s - 1, q);
        if (rem < r) { is_gemm = false; grp = s - 1; idx = rem; break; }
        rem -= r;
      }
    }
    if (!is_gemm) {
      wq.prefetch();
      const int j = idx * 8 + q;                         // (owned tile, slab) pair of the group
      const int t0 = grp * G;
      const int first = t0 + ((rank - t0 % w) + w) % w;  // first tile of the group this rank owns
      reduce_slab<IS_BF16, NT>(p, cv, first + (j / AR_SLABS) * w, j % AR_SLABS, tiles_per_owner, acquired);
      have = wq.next(item);
      continue;
    }
    const int t = grp * G + q * g8 + idx;
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
    AccT<8, NT> acc;
    mainloop_2ph<IS_BF16 ? DT_BF16 : DT_F16, 8, NT>(ta, smem, acc);
    // GEMM and reduce items differ 10x in length: the next item is drawn only now, under the epilogue
    wq.prefetch();
    uint16_t* slot_base = (owner == rank) ? cv.slots : symm_at(p.ctx, cv.slots, owner);
    uint16_t* dst = slot_base + ((long long)rank * tiles_per_owner + local) * TILE_ELEMS;
    epilogue_staged<IS_BF16 ? DT_BF16 : DT_F16, 8, NT>(acc, smem, [&](int r, int c, const uint4& v) {
      if (r < rows) st_wt128_sys(dst + r * BN + c, v);
    });
    drain_stores();
    __syncthreads();
    if (threadIdx.x == 0) {
      int* f1 = ((owner == rank) ? cv.flags : symm_at(p.ctx, cv.flags, owner)) + local * w + rank;
      store_relaxed_system(f1, cv.epoch);
    }
    have = wq.next(item);
  }
  // DRAIN: all slabs' results must have landed locally before the kernel completes
  {
    const int* f2 = cv.flags + tiles_per_owner * w;
    const bool dead = load_relaxed_agent(p.err) != 0;
    int failed = 0;
    for (int i = threadIdx.x; i < tiles * AR_SLABS; i += blockDim.x) {
      const int t = i / AR_SLABS, slab = i - t * AR_SLABS;
      const int pm = t / num_pid_n;
      if (slab * AR_SLAB_ROWS >= min(BM, p.M - pm * BM)) continue;       // (slab below the tile's last row: nothing to wait for)
      if (dead || !spin_until_eq(f2 + i, cv.epoch, SCOPE_SYS)) failed = 1;
    }
    if (failed && !dead) atomicOr(p.err, 1);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if (p.c_local) {
    const long long n16 = (long long)p.M * p.N / 8;
    const uint4* src = (const uint4*)cv.out;
    uint4* dst = (uint4*)p.c_local;
    for (long long i = (long long)blockIdx.x * THREADS + threadIdx.x; i < n16; i += (long long)gridDim.x * THREADS)
      dst[i] = src[i];
  }
  call_end(p.state, call, p.counters + 8, p.counters, 8);
}

// Flag words (per parity) and slot elements (per source rank) a context needs for problems up to [max_M, N]: the
// 256 x 128 tiling has the most tiles, the paddings of both tilings are covered.
TDIST_API int tdist_gemm_ar_sizes(int max_M, int N, int world, int* flag_ints, long long* slot_elems_per_rank) {
  const int pm = (max_M + BM - 1) / BM;
  const int t128 = pm * ((N + 127) / 128), t256 = pm * ((N + 255) / 256);
  const int tpo128 = (t128 + world - 1) / world, tpo256 = (t256 + world - 1) / world;
  *flag_ints = tpo128 * world + t128 * AR_SLABS;
  const long long e128 = (long long)tpo128 * BM * 128, e256 = (long long)tpo256 * BM * 256;
  *slot_elems_per_rank = e128 > e256 ? e128 : e256;
  return 0;
}

// `counters`: 12 ints, zero at the first launch (the kernel leaves them zero).  `state`: call number word.  Both
// parities of slots / out / flags are this rank's symmetric addresses.  `c_local`: optional plain [M, N] tensor that
// receives a copy of the result.
TDIST_API int tdist_gemm_ar(const tdist::SymmCtx* ctx, const void* A, const void* B, void* slots0, void* slots1, void* out,
                            int* flags0, int* flags1, int* counters, int* err, int* state, void* c_local,
                            long long lda, long long ldb, int M, int N, int K, int num_ctas, int bn, int dtype,
                            void* stream) {
  TDIST_REQUIRE(K >= 128 && K % 64 == 0, "K must be a multiple of 64 and >= 128");
  TDIST_REQUIRE(N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "N/lda/ldb must be multiples of 8");
  TDIST_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)out | (uintptr_t)slots0 | (uintptr_t)slots1 |
                  (uintptr_t)c_local) & 15) == 0, "operands must be 16-byte aligned");
  TDIST_REQUIRE(bn == 0 || bn == 128 || bn == 256, "bn must be 0 (auto), 128 or 256");
  TDIST_REQUIRE(M > 0 && N > 0 && state && counters && err, "bad arguments");
  GemmArParams p;
  p.ctx = *ctx;
  p.A = (const uint16_t*)A;
  p.B = (const uint16_t*)B;
  p.slots0 = (uint16_t*)slots0;
  p.slots1 = (uint16_t*)slots1;
  p.out = (uint16_t*)out;
  p.flags0 = flags0;
  p.flags1 = flags1;
  p.counters = counters;
  p.err = err;
  p.state = state;
  p.c_local = (uint16_t*)c_local;
  p.lda = lda;
  p.ldb = ldb;
  p.M = M;
  p.N = N;
  p.K = K;
  typedef void (*ar_kern_t)(GemmArParams);
  static const ar_kern_t kerns[2][2] = {{gemm_ar_kernel<true, 4>, gemm_ar_kernel<true, 2>},
                                        {gemm_ar_kernel<false, 4>, gemm_ar_kernel<false, 2>}};
  static const int lds_of[2] = {staged_lds_bytes<8, 4>() + 16, staged_lds_bytes<8, 2>() + 16};
  static bool attr_set = false;
  if (!attr_set) {
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j)
        TDIST_HIP_TRY(hipFuncSetAttribute((const void*)kerns[i][j], hipFuncAttributeMaxDynamicSharedMemorySize, lds_of[j]));
    attr_set = true;
  }
  static int num_cus = 0;
  if (!num_cus) {
    int dev = 0;
    TDIST_HIP_TRY(hipGetDevice(&dev));
    TDIST_HIP_TRY(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev));
  }
  // the whole grid must be resident (reduce items and the drain spin)
  int grid = num_ctas > 0 ? num_ctas : num_cus;
  if (grid > num_cus) grid = num_cus;
  const int pm = (M + BM - 1) / BM;
  const int t256 = pm * ((N + 255) / 256);
  // at most half as many 256-wide tiles as workgroups: 128-wide tiles fill the idle half of the chip (measured,
  // one rank of 8 alone on the GPU, N 5120, K 3200: 1024 rows 126 -> 110 us; from 2048 rows on the 256-wide tile wins)
  const int shape = bn ? (bn == 128 ? 1 : 0) : (2 * t256 <= grid ? 1 : 0);
  const int tiles = shape ? pm * ((N + 127) / 128) : t256;
  const int work = tiles + ((tiles + ctx->num_ranks - 1) / ctx->num_ranks) * AR_SLABS;
  if (grid > work) grid = work;
  // a group = one round of the grid (multiple of 8, at least 8 tiles)
  p.group_tiles = grid >= 8 ? (grid & ~7) : 8;
  hipLaunchKernelGGL(kerns[dtype == 0 ? 0 : 1][shape], dim3(grid), dim3(THREADS), lds_of[shape], (hipStream_t)stream, p);
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
