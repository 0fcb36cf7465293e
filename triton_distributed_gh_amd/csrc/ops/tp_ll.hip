// Low-latency forms of AllGather + GEMM and GEMM + ReduceScatter for decode-sized problems (M <= 256 rows in
// total): the shapes a tensor-parallel MLP / attention block sees in a decode step, where the 256x256 tile
// of ag_gemm.hip / gemm_rs.hip leaves the chip idle and tiles straddle ranks.  Both run the mid-M tile body
// (tdist/gemm_mid_gfx950.hpp: weights streamed through LDS once, split-K across workgroups); the collective
// part is small (M x K or M x N 16-bit values) and is handled whole:
//
//   AG + GEMM   phase A: the workgroups share out this rank's [M/w, K] shard and store it into every rank's
//               gathered workspace [M, K] (write-through, system scope); the last workgroup to finish raises
//               flag[rank] on every rank.  phase B: every workgroup waits for the w flags, then the GEMM runs
//               from the workspace.
//   GEMM + RS   phase 1: every workgroup computes its blocks of the partial [M, N] and stores the rows into
//               their owners' scatter buffers [source][M/w][N]; the last block raises flag[rank] on every
//               rank.  phase 2: wait for the w flags and sum this rank's rows over the sources (fp32).
//
// Workgroups are persistent over a static item list and the whole grid is resident (the host clamps it), so
// every wait is for stores whose producers never wait.  Calls are numbered on the device
// (tdist/call_state.hpp): workspaces / scatter buffers / flags alternate by call parity, flags carry the
// call's epoch, and the launch is identical every time - it can be captured into a HIP graph and replayed,
// which is what lets the `triton_dist` back-end decode from a graph at world size > 1.
//
// Role in the reference: the decode-shape use of ag_gemm / gemm_rs under CUDA-graph capture
// (python/triton_dist/models/engine.py:75-105, kernels/amd/all_gather_gemm.py:310-312).
#include "tdist/cache_policy.hpp"
#include "tdist/call_state.hpp"
#include "tdist/common.hpp"
#include "tdist/gemm_mid_gfx950.hpp"
#include "tdist/primitives.hpp"

using namespace tdist;
using namespace tdist::gemm;

namespace {

// Threads [0, w) of the workgroup wait for flags[0 .. w) == epoch, then everybody may read what the flags
// cover (one system-scope acquire).  Returns after a workgroup barrier.
__device__ __forceinline__ void wait_all_sources(const int* flags, int w, int epoch, int* err) {
  const int tid = threadIdx.x;
  if (tid < w) {
    if (load_relaxed_system(flags + tid) != epoch) {
      const bool dead = load_relaxed_agent(err) != 0;
      if (dead || !spin_until_eq(flags + tid, epoch, SCOPE_SYS)) {
        if (!dead) atomicOr(err, 1);
      }
    }
  }
  if (tid < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

struct AgLLParams {
  SymmCtx ctx;
  const uint16_t* a_local;     // [M_per_rank, K], row stride lda
  const uint16_t* B;           // [N_local, K]
  uint16_t* C;                 // [M, N_local]
  uint16_t *ws0, *ws1;         // symmetric: [max_M, K] dense
  int *flags0, *flags1;        // symmetric (signal arena): [w]
  float* partial;
  int* counters;               // split-K arrival counters; the word after them counts phase-A workgroups
  int n_counters;
  int* err;
  int* state;
  long long lda, ldb, ldc;
  int M_per_rank, N_local, K, splits;
};

template <bool IS_BF16, int MT, int WN>
__global__ __launch_bounds__(MID_THREADS, 2) void ag_gemm_ll_kernel(AgLLParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_arrived, s_call;
  const int w = p.ctx.num_ranks, rank = p.ctx.rank, tid = threadIdx.x;
  const int M = p.M_per_rank * w;
  const int call = call_begin(p.state, &s_call);
  const int epoch = call_epoch(call);
  uint16_t* ws = call_parity(call) ? p.ws1 : p.ws0;
  int* flags = call_parity(call) ? p.flags1 : p.flags0;

  // ---- phase A: this rank's shard into every rank's workspace (own copy included) ---------------------
  {
    const int pieces_per_row = p.K >> 3;
    const long long pieces = (long long)p.M_per_rank * pieces_per_row;
    const long long dst_off = (long long)rank * p.M_per_rank * p.K;
    for (long long i = (long long)blockIdx.x * MID_THREADS + tid; i < pieces; i += (long long)gridDim.x * MID_THREADS) {
      const int r = (int)(i / pieces_per_row), c = (int)(i - (long long)r * pieces_per_row);
      const uint4 v = *(const uint4*)(p.a_local + (long long)r * p.lda + c * 8);
      for (int j = 0; j < w; ++j) {
        const int peer = (rank + j) % w;
        uint16_t* dst = (j == 0 ? ws : symm_at(p.ctx, ws, peer)) + dst_off + (long long)r * p.K + c * 8;
        st_wt128_sys(dst, v);
      }
    }
    drain_stores();
    __syncthreads();
    if (tid == 0) {
      int* done = p.counters + p.n_counters;
      const int prev = atom_add_relaxed_agent(done, 1);
      if (prev == (int)gridDim.x - 1) {
        store_relaxed_agent(done, 0);                    // re-armed for the next launch
        for (int j = 0; j < w; ++j) {
          const int peer = (rank + j) % w;
          store_relaxed_system((j == 0 ? flags : symm_at(p.ctx, flags, peer)) + rank, epoch);
        }
      }
    }
  }
  // ---- phase B: GEMM from the gathered workspace ----------------------------------------------------------
  wait_all_sources(flags, w, epoch, p.err);
  const int ncb = (p.N_local + 64 * WN - 1) / (64 * WN);
  const int passes = (M + 16 * MT - 1) / (16 * MT);
  const int nblocks = ncb * passes, S = p.splits;
  const int nst_total = p.K / 64;
  constexpr int PER_SPLIT = 4 * MT * WN * 64;
  for (int item = blockIdx.x; item < nblocks * S; item += gridDim.x) {
    const int split = item / nblocks, b = item - split * nblocks;
    const int pass = b / ncb, cb = b - pass * ncb;
    const int st0 = (int)((long long)nst_total * split / S), st1 = (int)((long long)nst_total * (split + 1) / S);
    f32x4 acc[MT][WN];
    mid_mainloop<IS_BF16, MT, WN>(ws, p.K, p.B, p.ldb, M, p.N_local, pass * 16 * MT, cb, st0, st1 - st0, smem, acc);
    if (S > 1) {
      if (!mid_splitk_join<MT, WN>(acc, (f32x4*)p.partial + (long long)b * S * PER_SPLIT, p.counters + b, split, S,
                                   &s_arrived))
        continue;
    }
    mid_store<IS_BF16, MT, WN>(acc, pass * 16 * MT, cb, [&](int row, int col, const uint2& v) {
      if (row < M && col < p.N_local) *(uint2*)(p.C + (long long)row * p.ldc + col) = v;
    });
    __syncthreads();                                       // LDS is restaged by the next item
  }
  call_end(p.state, call);
}

struct RsLLParams {
  SymmCtx ctx;
  const uint16_t* A;           // [M, K_local]
  const uint16_t* B;           // [N, K_local]
  uint16_t* out;               // [M_per_rank, N], row stride ldo
  uint16_t *sc0, *sc1;         // symmetric: [w][M_per_rank][N] (sources x this rank's rows)
  int *flags0, *flags1;        // symmetric (signal arena): [w]
  float* partial;
  int* counters;               // split-K arrival counters; the word after them counts pushed blocks
  int n_counters;
  int* err;
  int* state;
  long long lda, ldb, ldo;
  int M_per_rank, N, K, splits;
};

template <bool IS_BF16, int MT, int WN>
__global__ __launch_bounds__(MID_THREADS, 2) void gemm_rs_ll_kernel(RsLLParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_arrived, s_call;
  const int w = p.ctx.num_ranks, rank = p.ctx.rank, tid = threadIdx.x;
  const int Mr = p.M_per_rank, M = Mr * w;
  const int call = call_begin(p.state, &s_call);
  const int epoch = call_epoch(call);
  uint16_t* sc = call_parity(call) ? p.sc1 : p.sc0;
  int* flags = call_parity(call) ? p.flags1 : p.flags0;
  const int ncb = (p.N + 64 * WN - 1) / (64 * WN);
  const int passes = (M + 16 * MT - 1) / (16 * MT);
  const int nblocks = ncb * passes, S = p.splits;
  const int nst_total = p.K / 64;
  constexpr int PER_SPLIT = 4 * MT * WN * 64;

  // ---- phase 1: partial rows to their owners -------------------------------------------------------------
  for (int item = blockIdx.x; item < nblocks * S; item += gridDim.x) {
    const int split = item / nblocks, b = item - split * nblocks;
    const int pass = b / ncb, cb = b - pass * ncb;
    const int st0 = (int)((long long)nst_total * split / S), st1 = (int)((long long)nst_total * (split + 1) / S);
    f32x4 acc[MT][WN];
    mid_mainloop<IS_BF16, MT, WN>(p.A, p.lda, p.B, p.ldb, M, p.N, pass * 16 * MT, cb, st0, st1 - st0, smem, acc);
    if (S > 1) {
      if (!mid_splitk_join<MT, WN>(acc, (f32x4*)p.partial + (long long)b * S * PER_SPLIT, p.counters + b, split, S,
                                   &s_arrived))
        continue;
    }
    mid_store<IS_BF16, MT, WN>(acc, pass * 16 * MT, cb, [&](int row, int col, const uint2& v) {
      if (row < M && col < p.N) {
        const int owner = row / Mr, rl = row - owner * Mr;
        uint16_t* dst = (owner == rank ? sc : symm_at(p.ctx, sc, owner)) + ((long long)rank * Mr + rl) * p.N + col;
        st_wt64_sys(dst, v);
      }
    });
    drain_stores();
    __syncthreads();
    if (tid == 0) {
      int* done = p.counters + p.n_counters;
      const int prev = atom_add_relaxed_agent(done, 1);
      if (prev == nblocks - 1) {                           // every block of this rank's partial is out
        store_relaxed_agent(done, 0);
        for (int j = 0; j < w; ++j) {
          const int peer = (rank + j) % w;
          store_relaxed_system((j == 0 ? flags : symm_at(p.ctx, flags, peer)) + rank, epoch);
        }
      }
    }
  }
  // ---- phase 2: this rank's rows, summed over the sources -------------------------------------------------
  wait_all_sources(flags, w, epoch, p.err);
  {
    const int pieces_per_row = p.N >> 3;
    const long long pieces = (long long)Mr * pieces_per_row;
    const long long src_stride = (long long)Mr * p.N;
    for (long long i = (long long)blockIdx.x * MID_THREADS + tid; i < pieces; i += (long long)gridDim.x * MID_THREADS) {
      const int r = (int)(i / pieces_per_row), c = (int)(i - (long long)r * pieces_per_row);
      float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      for (int s = 0; s < w; ++s) {
        const uint4 v = *(const uint4*)(sc + s * src_stride + (long long)r * p.N + c * 8);
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
      *(uint4*)(p.out + (long long)r * p.ldo + c * 8) = o;
    }
  }
  call_end(p.state, call);
}

int device_cus() {
  static int num_cus = 0;
  if (!num_cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    if (hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) num_cus = 256;
  }
  return num_cus;
}

template <class Kern, class Params>
int launch_ll(Kern kern, const Params& p, int lds, int grid, hipStream_t s, bool& attr_set) {
  if (!attr_set) {
    TDIST_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(MID_THREADS), lds, s, p);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

#define TDIST_LL_CASE(KERNEL, BF, MT_, WN_)                                                            \
  {                                                                                                    \
    static bool attr = false;                                                                          \
    return launch_ll(KERNEL<BF, MT_, WN_>, p, mid_lds_bytes(MT_, WN_), grid, s, attr);                  \
  }

#define TDIST_LL_DISPATCH(KERNEL)                                    \
  if (bf) {                                                          \
    if (mt == 2 && wn == 1) TDIST_LL_CASE(KERNEL, true, 2, 1)        \
    if (mt == 2 && wn == 2) TDIST_LL_CASE(KERNEL, true, 2, 2)        \
    if (mt == 4 && wn == 1) TDIST_LL_CASE(KERNEL, true, 4, 1)        \
    if (mt == 4 && wn == 2) TDIST_LL_CASE(KERNEL, true, 4, 2)        \
    if (mt == 8 && wn == 1) TDIST_LL_CASE(KERNEL, true, 8, 1)        \
    TDIST_LL_CASE(KERNEL, true, 8, 2)                                \
  } else {                                                           \
    if (mt == 2 && wn == 1) TDIST_LL_CASE(KERNEL, false, 2, 1)       \
    if (mt == 2 && wn == 2) TDIST_LL_CASE(KERNEL, false, 2, 2)       \
    if (mt == 4 && wn == 1) TDIST_LL_CASE(KERNEL, false, 4, 1)       \
    if (mt == 4 && wn == 2) TDIST_LL_CASE(KERNEL, false, 4, 2)       \
    if (mt == 8 && wn == 1) TDIST_LL_CASE(KERNEL, false, 8, 1)       \
    TDIST_LL_CASE(KERNEL, false, 8, 2)                               \
  }

int dispatch_ag(const AgLLParams& p, int mt, int wn, int grid, bool bf, hipStream_t s) { TDIST_LL_DISPATCH(ag_gemm_ll_kernel) }
int dispatch_rs(const RsLLParams& p, int mt, int wn, int grid, bool bf, hipStream_t s) { TDIST_LL_DISPATCH(gemm_rs_ll_kernel) }

}  // namespace

// Split-K workspace bytes and counters (one extra word behind them) that a context for problems up to
// [max_M, N] needs: split-K only cuts K while splits * blocks fits the grid, so the workspace is bounded
// by (workgroups) x (fp32 bytes of the largest block).
TDIST_API int tdist_tp_ll_sizes(int max_M, int N, long long* partial_bytes, int* n_counters) {
  TDIST_REQUIRE(max_M > 0 && N > 0, "bad shape");
  *partial_bytes = 512ll * 128 * 128 * 4;
  *n_counters = ((max_M + 31) / 32) * ((N + 63) / 64);
  return 0;
}

// AG + GEMM, low-latency form.  `counters` holds n_counters + 1 zero-initialised ints, `state` 2.
TDIST_API int tdist_ag_gemm_ll(const tdist::SymmCtx* ctx, const void* a_local, const void* B, void* C, void* ws0,
                               void* ws1, int* flags0, int* flags1, void* partial, long long partial_bytes,
                               int* counters, int n_counters, int* err, int* state, long long lda, long long ldb,
                               long long ldc, int M_per_rank, int N_local, int K, long long ws_rows, int grid_budget,
                               int dtype, void* stream) {
  TDIST_REQUIRE(K >= 128 && K % 64 == 0, "K must be a multiple of 64 and >= 128");
  TDIST_REQUIRE(N_local % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0, "strides must keep the vector alignment");
  TDIST_REQUIRE((((uintptr_t)a_local | (uintptr_t)B | (uintptr_t)ws0 | (uintptr_t)ws1) & 15) == 0 &&
                    (((uintptr_t)C) & 7) == 0, "operands must be aligned");
  const int w = ctx->num_ranks;
  TDIST_REQUIRE(M_per_rank > 0 && (long long)M_per_rank * w <= ws_rows, "M exceeds the workspace");
  TDIST_REQUIRE(state && err && counters, "bad arguments");
  const int budget = grid_budget > 0 ? grid_budget : device_cus();
  const MidPlan pl = mid_plan(M_per_rank * w, N_local, K, budget, 0, 0);
  const int nblocks = pl.ncb * pl.passes;
  TDIST_REQUIRE(nblocks <= n_counters && (pl.splits == 1 || (partial && pl.partial_bytes <= partial_bytes)),
                "split-K workspace too small");
  AgLLParams p;
  p.ctx = *ctx;
  p.a_local = (const uint16_t*)a_local;
  p.B = (const uint16_t*)B;
  p.C = (uint16_t*)C;
  p.ws0 = (uint16_t*)ws0;
  p.ws1 = (uint16_t*)ws1;
  p.flags0 = flags0;
  p.flags1 = flags1;
  p.partial = (float*)partial;
  p.counters = counters;
  p.n_counters = n_counters;
  p.err = err;
  p.state = state;
  p.lda = lda;
  p.ldb = ldb;
  p.ldc = ldc;
  p.M_per_rank = M_per_rank;
  p.N_local = N_local;
  p.K = K;
  p.splits = pl.splits;
  int grid = nblocks * pl.splits;
  if (grid > budget) grid = budget;        // the whole grid must be resident (phase B spins)
  hipStream_t s = (hipStream_t)stream;
  const int mt = pl.mt, wn = pl.wn;
  return dispatch_ag(p, mt, wn, grid, dtype == 0, s);
}

// GEMM + RS, low-latency form.  `counters` holds n_counters + 1 zero-initialised ints, `state` 2;
// the scatter buffers hold w * sc_rows_per_source * N values with sc_rows_per_source >= M_per_rank.
TDIST_API int tdist_gemm_rs_ll(const tdist::SymmCtx* ctx, const void* A, const void* B, void* out, void* sc0, void* sc1,
                               int* flags0, int* flags1, void* partial, long long partial_bytes, int* counters,
                               int n_counters, int* err, int* state, long long lda, long long ldb, long long ldo,
                               int M_per_rank, int N, int K, long long sc_elems, int grid_budget, int dtype,
                               void* stream) {
  TDIST_REQUIRE(K >= 128 && K % 64 == 0, "K must be a multiple of 64 and >= 128");
  TDIST_REQUIRE(N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldo % 8 == 0, "strides must keep the vector alignment");
  TDIST_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)out | (uintptr_t)sc0 | (uintptr_t)sc1) & 15) == 0,
                "operands must be 16-byte aligned");
  const int w = ctx->num_ranks;
  TDIST_REQUIRE(M_per_rank > 0 && (long long)w * M_per_rank * N <= sc_elems, "M exceeds the scatter buffers");
  TDIST_REQUIRE(state && err && counters, "bad arguments");
  const int budget = grid_budget > 0 ? grid_budget : device_cus();
  const MidPlan pl = mid_plan(M_per_rank * w, N, K, budget, 0, 0);
  const int nblocks = pl.ncb * pl.passes;
  TDIST_REQUIRE(nblocks <= n_counters && (pl.splits == 1 || (partial && pl.partial_bytes <= partial_bytes)),
                "split-K workspace too small");
  RsLLParams p;
  p.ctx = *ctx;
  p.A = (const uint16_t*)A;
  p.B = (const uint16_t*)B;
  p.out = (uint16_t*)out;
  p.sc0 = (uint16_t*)sc0;
  p.sc1 = (uint16_t*)sc1;
  p.flags0 = flags0;
  p.flags1 = flags1;
  p.partial = (float*)partial;
  p.counters = counters;
  p.n_counters = n_counters;
  p.err = err;
  p.state = state;
  p.lda = lda;
  p.ldb = ldb;
  p.ldo = ldo;
  p.M_per_rank = M_per_rank;
  p.N = N;
  p.K = K;
  p.splits = pl.splits;
  int grid = nblocks * pl.splits;
  if (grid > budget) grid = budget;
  hipStream_t s = (hipStream_t)stream;
  const int mt = pl.mt, wn = pl.wn;
  return dispatch_rs(p, mt, wn, grid, dtype == 0, s);
}
