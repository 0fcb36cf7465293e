// Plain (single-GPU) 16-bit GEMM on the gfx950 tile engine:  C[M,N] = A[M,K] * B[N,K]^T.
//
// This is TP=1 of the headline metric and the building block of every fused op.  The launch is
// one workgroup per 256x256 tile; block ids are remapped so that the ids that share an XCD's L2
// own a compact group_m x (64/group_m) patch of tiles.
//
// Role in the reference: `tl.dot` matmul kernels of python/triton_dist/test/amd/test_matmul_amd.py
// and my_lab/benchmark/layer.py:20-64 (y = x W^T).
#include <stdlib.h>

#include <type_traits>

#include "tdist/atomics.hpp"
#include "tdist/cache_policy.hpp"
#include "tdist/common.hpp"
#include "tdist/gemm_gfx950.hpp"
#include "tdist/tile_plan.hpp"
#include "tdist/wg_trace.hpp"

using namespace tdist::gemm;

template <bool IS_BF16, int VARIANT>
__global__ __launch_bounds__(THREADS, 2) void gemm_nt_256_kernel(const uint16_t* __restrict__ A,
                                                                 const uint16_t* __restrict__ B,
                                                                 uint16_t* __restrict__ C, int M, int N, int K,
                                                                 long long lda, long long ldb, long long ldc,
                                                                 int group_m, int dbg_flags) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int num_pid_m = (M + BM - 1) / BM;
  const int num_pid_n = (N + BN - 1) / BN;
  const int nwg = num_pid_m * num_pid_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  int pid_m, pid_n;
  tile_coords(id, num_pid_m, num_pid_n, group_m, pid_m, pid_n);
  const int row0 = pid_m * BM, col0 = pid_n * BN;

  TileArgs t;
  t.A = A + (long long)row0 * lda;
  t.B = B + (long long)col0 * ldb;
  t.lda = lda;
  t.ldb = ldb;
  t.a_rows = M - row0;
  t.b_rows = N - col0;
  t.K = K;
  Acc<IS_BF16> acc;
  if constexpr (VARIANT == 2)
    mainloop_256_2ph<IS_BF16>(t, smem, acc);
  else if constexpr (VARIANT == 5)
    mainloop_256_2ph<IS_BF16, false, 0, 1>(t, smem, acc);
  else if constexpr (VARIANT == 6)
    mainloop_256_2ph<IS_BF16, false, 4, 1>(t, smem, acc);
  else if constexpr (VARIANT == 7)
    mainloop_256_2ph<IS_BF16, false, 4, 0>(t, smem, acc);
  else if constexpr (VARIANT == 8)
    mainloop_256_2ph<IS_BF16, false, 3, 1>(t, smem, acc);
  else if constexpr (VARIANT == 9)
    mainloop_256_2ph<IS_BF16, false, 0, 3>(t, smem, acc);
  else if constexpr (VARIANT == 10)
    mainloop_256_2ph<IS_BF16, false, 2, 3>(t, smem, acc);
  else
    mainloop_256<IS_BF16>(t, smem, acc);
  const int mrem = M - row0, nrem = N - col0;
  uint16_t* cbase = C + (long long)row0 * ldc + col0;
  epilogue_256<IS_BF16>(acc, smem, [&](int r, int c, const uint4& v) {
    if (r < mrem && c < nrem && !(dbg_flags & 1)) *(uint4*)(cbase + (long long)r * ldc + c) = v;
  });
}

// Persistent form: one workgroup per CU walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... (the
// grid is a multiple of 8, so a workgroup stays on its XCD and keeps the same compact patch mapping).
// The epilogue stores straight from registers (permuted accumulator columns), so the first loads of
// the next tile are issued right behind the stores of the previous one: with two tiles per CU at the
// headline shape, the 32 MiB that every CU of the chip writes at the same moment drains under the
// next main loop instead of idling the matrix cores.
template <bool IS_BF16, int OPT>
__global__ __launch_bounds__(THREADS, 2) void gemm_nt_256_persistent_kernel(const uint16_t* __restrict__ A,
                                                                            const uint16_t* __restrict__ B,
                                                                            uint16_t* __restrict__ C, int M, int N,
                                                                            int K, long long lda, long long ldb,
                                                                            long long ldc, int group_m,
                                                                            tdist::WgTrace trace) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int num_pid_m = (M + BM - 1) / BM;
  const int num_pid_n = (N + BN - 1) / BN;
  const int nwg = num_pid_m * num_pid_n;
  tdist::WgTraceCursor tr = tdist::wg_trace_begin(trace);
  tdist::wg_trace(tr, 1, 0);
  for (int vb = blockIdx.x; vb < nwg; vb += gridDim.x) {
    const int id = xcd_remap(vb, nwg);
    int pid_m, pid_n;
    tile_coords(id, num_pid_m, num_pid_n, group_m, pid_m, pid_n);
    const int row0 = pid_m * BM, col0 = pid_n * BN;
    tdist::wg_trace(tr, 2, id);
    TileArgs t;
    t.A = A + (long long)row0 * lda;
    t.B = B + (long long)col0 * ldb;
    t.lda = lda;
    t.ldb = ldb;
    t.a_rows = M - row0;
    t.b_rows = N - col0;
    t.K = K;
    Acc<IS_BF16> acc;
    mainloop_256_2ph<IS_BF16, true, OPT>(t, smem, acc);
    tdist::wg_trace(tr, 4, id);
    const int mrem = M - row0, nrem = N - col0;
    uint16_t* cbase = C + (long long)row0 * ldc + col0;
    epilogue_256_direct<IS_BF16>(acc, [&](int r, int c, const uint4& v) {
      if (r < mrem && c < nrem) *(uint4*)(cbase + (long long)r * ldc + c) = v;
    });
    tdist::wg_trace(tr, 5, id);
  }
  tdist::wg_trace(tr, 7, 0);
}

// Four-wave persistent form (variant 4): see mainloop_256_4w.
template <bool IS_BF16, int MODE>
__global__ __launch_bounds__(W4_THREADS) void gemm_nt_256_4w_kernel(const uint16_t* __restrict__ A,
                                                                    const uint16_t* __restrict__ B,
                                                                    uint16_t* __restrict__ C, int M, int N, int K,
                                                                    long long lda, long long ldb, long long ldc,
                                                                    int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int num_pid_m = (M + BM - 1) / BM;
  const int num_pid_n = (N + BN - 1) / BN;
  const int nwg = num_pid_m * num_pid_n;
  for (int vb = blockIdx.x; vb < nwg; vb += gridDim.x) {
    const int id = xcd_remap(vb, nwg);
    int pid_m, pid_n;
    tile_coords(id, num_pid_m, num_pid_n, group_m, pid_m, pid_n);
    const int row0 = pid_m * BM, col0 = pid_n * BN;
    TileArgs t;
    t.A = A + (long long)row0 * lda;
    t.B = B + (long long)col0 * ldb;
    t.lda = lda;
    t.ldb = ldb;
    t.a_rows = M - row0;
    t.b_rows = N - col0;
    t.K = K;
    Acc4<IS_BF16> acc;
    if constexpr (MODE == 2)
      mainloop_256_4w_rs<IS_BF16>(t, smem, acc);
    else
      mainloop_256_4w<IS_BF16, MODE == 1>(t, smem, acc);
    const int mrem = M - row0, nrem = N - col0;
    uint16_t* cbase = C + (long long)row0 * ldc + col0;
    epilogue_256_4w_direct<IS_BF16>(acc, [&](int r, int c, const uint4& v) {
      if (r < mrem && c < nrem) *(uint4*)(cbase + (long long)r * ldc + c) = v;
    });
    // the next tile's prologue restages all four slabs: every wave must be done reading first
    __syncthreads();
  }
}

// Four-wave persistent form on 32x32x16 MFMAs (variant 13): see mainloop_256_4w32.
template <bool IS_BF16, int EXP = 0>
__global__ __launch_bounds__(W4_THREADS) void gemm_nt_256_4w32_kernel(const uint16_t* __restrict__ A,
                                                                      const uint16_t* __restrict__ B,
                                                                      uint16_t* __restrict__ C, int M, int N, int K,
                                                                      long long lda, long long ldb, long long ldc,
                                                                      int group_m) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int num_pid_m = (M + BM - 1) / BM;
  const int num_pid_n = (N + BN - 1) / BN;
  const int nwg = num_pid_m * num_pid_n;
  for (int vb = blockIdx.x; vb < nwg; vb += gridDim.x) {
    const int id = xcd_remap(vb, nwg);
    int pid_m, pid_n;
    tile_coords(id, num_pid_m, num_pid_n, group_m, pid_m, pid_n);
    const int row0 = pid_m * BM, col0 = pid_n * BN;
    TileArgs t;
    t.A = A + (long long)row0 * lda;
    t.B = B + (long long)col0 * ldb;
    t.lda = lda;
    t.ldb = ldb;
    t.a_rows = M - row0;
    t.b_rows = N - col0;
    t.K = K;
    Acc32<IS_BF16> acc;
    mainloop_256_4w32<IS_BF16, EXP>(t, smem, acc);
    const int mrem = M - row0, nrem = N - col0;
    uint16_t* cbase = C + (long long)row0 * ldc + col0;
    epilogue_256_4w32_direct<IS_BF16>(acc, [&](int r, int c, const uint4& v) {
      if (r < mrem && c < nrem) *(uint4*)(cbase + (long long)r * ldc + c) = v;
    });
    // the next tile's prologue restages all four slabs: every wave must be done reading first
    __syncthreads();
  }
}

// The tile family (gemm_gfx950.hpp: mainloop_2ph<DT, MT, NT>) as a persistent plain GEMM: BM = 32 MT, BN = 64 NT,
// 16-bit or 8-bit operands.  OUT32: the raw int32 accumulators of an int8 GEMM are stored instead of 16-bit
// results.  sa / sb: optional per-row (A) / per-column (B) fp32 scales applied to the 16-bit output, bias: optional
// per-column addend of the output type (SCALED instances only).
// A, B, lda, ldb, K are in 16-bit units (8-bit operands: two elements per unit).
//
// split_k > 1: K is cut into split_k ranges and the work items are (K range, tile) pairs, K range slowest, so the
// workgroups running at any moment cover all tiles of one K range.  Wait-free join: an item parks its fp32
// accumulators (register layout, write-through stores), drains, and counts itself on the tile's arrival word; the
// item that counts last adds the others' partials to its live accumulators, re-arms the word and stores the tile.
// `partial`: tiles * split_k slots of BM * BN * 4 bytes; `counters`: one int per tile, zero at launch.
template <int DT, int MT, int NT, bool OUT32, bool SCALED = false>
__global__ __launch_bounds__(THREADS, (2 * TileShape<MT, NT>::WGS_PER_CU)) void gemm_nt_tile_kernel(
    const uint16_t* __restrict__ A, const uint16_t* __restrict__ B, void* __restrict__ Cv, int M, int N, int K,
    long long lda, long long ldb, long long ldc, int group_m, const float* __restrict__ sa,
    const float* __restrict__ sb, int split_k, float* __restrict__ partial, int* __restrict__ counters, int n_full,
    const uint16_t* __restrict__ bias) {
  using S = TileShape<MT, NT>;
  using F = Frag<DT>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int num_pid_m = (M + S::BM - 1) / S::BM;
  const int num_pid_n = (N + S::BN - 1) / S::BN;
  const int nwg = num_pid_m * num_pid_n;
  const int ksteps = K / BK;
  // tiles [0, n_full) are one item each; the rest (the last, partial round) split_k items each, K range slowest
  const int n_tail = nwg - n_full;
  for (int item = blockIdx.x; item < n_full + n_tail * split_k; item += gridDim.x) {
    const bool whole = item < n_full;
    const int ks = whole ? 0 : (item - n_full) / n_tail;
    const int vb = whole ? item : n_full + (item - n_full) - ks * n_tail;
    const int S_item = whole ? 1 : split_k;
    const int id = xcd_remap(vb, nwg);
    int pid_m, pid_n;
    tile_coords(id, num_pid_m, num_pid_n, group_m, pid_m, pid_n);
    const int row0 = pid_m * S::BM, col0 = pid_n * S::BN;
    const int k0 = (int)((long long)ksteps * ks / S_item) * BK;
    const int k1 = (int)((long long)ksteps * (ks + 1) / S_item) * BK;
    TileArgs t;
    t.A = A + (long long)row0 * lda + k0;
    t.B = B + (long long)col0 * ldb + k0;
    t.lda = lda;
    t.ldb = ldb;
    t.a_rows = M - row0;
    t.b_rows = N - col0;
    t.K = k1 - k0;
    AccT<MT, NT> acc;
    mainloop_2ph<DT, MT, NT>(t, smem, acc);
    if (S_item > 1) {
      constexpr int SLOT = MT * NT * THREADS;                 // f32x4 per partial tile
      volatile int* s_arrived = (volatile int*)(smem + S::LDS);
      const int tix = vb - n_full;                            // index among the split tiles
      f32x4* tbase = (f32x4*)partial + (long long)tix * split_k * SLOT + threadIdx.x;
      {
        f32x4* mine = tbase + (long long)ks * SLOT;
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) tdist::st_wt128_agent(mine + (m * NT + n) * THREADS, acc.v[m][n]);
      }
      tdist::drain_stores();
      __syncthreads();
      if (threadIdx.x == 0) *s_arrived = tdist::atom_add_relaxed_agent(counters + tix, 1);
      __syncthreads();
      const int arrived = __builtin_amdgcn_readfirstlane(*s_arrived);
      __syncthreads();                                        // (the word is rewritten by the next item)
      if (arrived != split_k - 1) continue;
      if (threadIdx.x < 64) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      if (threadIdx.x == 0) tdist::store_relaxed_agent(counters + tix, 0);    // re-armed for the next launch
      for (int q = 0; q < split_k; ++q) {
        if (q == ks) continue;
        const f32x4* src = tbase + (long long)q * SLOT;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          f32x4 x[NT];
#pragma unroll
          for (int n = 0; n < NT; ++n) x[n] = src[(m * NT + n) * THREADS];
#pragma unroll
          for (int n = 0; n < NT; ++n) acc.v[m][n] = F::add(acc.v[m][n], x[n]);
        }
      }
    }
    const int mrem = M - row0, nrem = N - col0;
    if constexpr (OUT32) {
      uint32_t* cbase = (uint32_t*)Cv + (long long)row0 * ldc + col0;
      epilogue_direct_raw32<MT, NT>(acc, [&](int r, int c, const uint4& lo, const uint4& hi) {
        if (r < mrem && c < nrem) {
          *(uint4*)(cbase + (long long)r * ldc + c) = lo;
          *(uint4*)(cbase + (long long)r * ldc + c + 4) = hi;
        }
      });
    } else {
      uint16_t* cbase = (uint16_t*)Cv + (long long)row0 * ldc + col0;
      epilogue_direct<DT, MT, NT>(acc, SCALED ? sa : nullptr, SCALED ? sb : nullptr, row0, col0, mrem, nrem, [&](int r, int c, const uint4& v) {
        if (r < mrem && c < nrem) *(uint4*)(cbase + (long long)r * ldc + c) = v;
      }, SCALED ? bias : nullptr);
    }
  }
}

// Decode-shaped GEMM (M <= 16): the weights are streamed exactly once with non-temporal 16-byte
// loads and feed 16x16x32 MFMAs whose M dimension is the zero-padded batch.  A workgroup of four
// (or eight, NW) waves covers BN = 16 * WN output columns (WN = 1 / 2 / 4); wave k owns the k-th slice
// of K for ALL of the workgroup's columns, so one activation fragment feeds WN MFMAs and the activations
// (at most 16 x K, L2 resident) are read once per workgroup instead of once per column group.  Each
// lane keeps sixteen 16-byte weight loads in flight.  The K slices are summed through LDS, one
// column group per wave.  Eight waves are used when the grid alone cannot fill the chip (short N)
// and K is long enough to give every wave a few batches of loads.
//
// SILU (WN == 2): W is a gate_up weight [2 I, K]; the workgroup's two column groups are gate columns
// n0 .. n0+15 and the matching up columns I + n0 .. I + n0+15, and the epilogue stores
// silu(gate) * up into C [M, I] - the activation kernel of a gated MLP folded into its first projection
// (reference: mega_triton_kernel/kernels/mlp_fc1.py + activation.py as two tasks).
//
// NORM (bf16): the RMSNorm in front of the projection rides along.  X is the un-normalised input, R an optional
// residual added to it first (h = bf16(X + R); workgroup 0 writes h to Hout, which must not alias X or R) and G the
// norm weight: the MFMAs consume bf16(h * G) and the epilogue multiplies every row by rsqrt(mean(h^2) + eps),
// whose sum of squares each wave collects over its own K slice from the fragments it loads anyway.  Every
// workgroup repeats that arithmetic (a few VALU operations per 16 bytes of weights: the kernel is HBM-bound), and
// a decode step loses the two norm launches per layer with their dependent round trips.
template <bool IS_BF16, int WN, int NW, bool SILU = false, bool NORM = false>
__global__ __launch_bounds__(64 * NW) void gemm_nt_skinny_kernel(const uint16_t* __restrict__ X,
                                                             const uint16_t* __restrict__ W,
                                                             uint16_t* __restrict__ C, int M, int N, int K,
                                                             long long ldx, long long ldw, long long ldc,
                                                             const uint16_t* __restrict__ R = nullptr,
                                                             const uint16_t* __restrict__ G = nullptr,
                                                             uint16_t* __restrict__ Hout = nullptr, float eps = 0.f,
                                                             const int* __restrict__ gid = nullptr, int gdiv = 1,
                                                             long long gstride = 0, int gcount = 1) {
  static_assert(!NORM || IS_BF16, "the fused norm is built for bf16");
  // Grouped form (MoE decode): blockIdx.y = (token, expert) pair p; the pair multiplies row p / gdiv of X with the
  // weights of expert gid[p] (gstride elements apart) into row p of C.
  if (gid) {
    const int p = blockIdx.y;
    W += (long long)min(max(gid[p], 0), gcount - 1) * gstride;      // (an id out of range must not become a fault)
    X += (long long)(p / gdiv) * ldx;
    C += (long long)p * ldc;
  }
  using F = Frag<IS_BF16>;
  using VT = typename F::T;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  static_assert(!SILU || WN == 2, "the fused activation pairs one gate group with one up group");
  constexpr int BN_ = SILU ? 16 : 16 * WN;               // output columns per workgroup
  constexpr int U = 16 / WN;                             // K-steps per batch of loads (16 weight loads in flight)
  const int up_off = SILU ? N : 0;                       // (SILU: N is I, the number of OUTPUT columns)
  __shared__ float red[NW * WN * 64 * 4];                // [K slice][column group][lane][4]
  const int wk = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 15, lg = lane >> 4;
  const int n0 = blockIdx.x * BN_;
  const int ksteps = K / 32;
  const int per = (ksteps + NW - 1) / NW;
  const int ks0 = min(wk * per, ksteps), ks1 = min(ks0 + per, ksteps);
  const uint16_t* xr = X + (long long)min(li, M - 1) * ldx + lg * 8;
  const uint16_t* wr[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j)                            // clamped: out-of-range columns are not stored
    wr[j] = W + (long long)(SILU ? j * up_off + min(n0 + li, N - 1) : min(n0 + j * 16 + li, N - 1)) * ldw + lg * 8;
  const bool xvalid = li < M;
  f32x4 acc[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // ---- fused norm (see the comment above the kernel): normalised-but-unscaled activations go to LDS ----
  extern __shared__ __attribute__((aligned(16))) char tbuf[];     // NORM: [M][K * 2 + 64] bytes (rows padded: banks)
  __shared__ float s_ss[16];
  const long long trow = (long long)K * 2 + 64;
  auto norm_prepass = [&]() {
    if constexpr (NORM) {
      if (threadIdx.x < 16) s_ss[threadIdx.x] = 0.f;
      __syncthreads();
      const int nvec = K / 8;
      for (int row = 0; row < M; ++row) {
        float ss = 0.f;
        for (int c = threadIdx.x; c < nvec; c += 64 * NW) {
          const u32x4 xv = *(const u32x4*)(X + (long long)row * ldx + c * 8);
          const u32x4 gv = *(const u32x4*)(G + c * 8);
          u32x4 rv = {0, 0, 0, 0}, hb, t;
          if (R) rv = *(const u32x4*)(R + (long long)row * K + c * 8);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float lo = __builtin_bit_cast(float, xv[i] << 16), hi = __builtin_bit_cast(float, xv[i] & 0xffff0000u);
            hb[i] = xv[i];
            if (R) {
              lo += __builtin_bit_cast(float, rv[i] << 16);
              hi += __builtin_bit_cast(float, rv[i] & 0xffff0000u);
              hb[i] = F::pack2(lo, hi);      // the sum rounded to 16 bits is what the norm sees (as the separate kernel)
              lo = __builtin_bit_cast(float, hb[i] << 16), hi = __builtin_bit_cast(float, hb[i] & 0xffff0000u);
            }
            ss += lo * lo + hi * hi;
            t[i] = F::pack2(lo * __builtin_bit_cast(float, gv[i] << 16), hi * __builtin_bit_cast(float, gv[i] & 0xffff0000u));
          }
          *(u32x4*)(tbuf + row * trow + c * 16) = t;
          if (R && blockIdx.x == 0) *(u32x4*)(Hout + (long long)row * K + c * 8) = hb;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        if (lane == 0) atomicAdd(&s_ss[row], ss);
      }
      __syncthreads();
    }
  };
  auto load_x = [&](int ks) -> u32x4 {
    if constexpr (NORM) {
      return xvalid ? *(const u32x4*)(tbuf + li * trow + (ks * 32 + lg * 8) * 2) : u32x4{0, 0, 0, 0};
    } else {
      // (unconditional: rows past M read row M - 1 again and are zeroed - a lane-dependent branch around the load
      // would split every batch of weight loads)
      const u32x4 v = *(const u32x4*)(xr + ks * 32);
      return xvalid ? v : u32x4{0, 0, 0, 0};
    }
  };
  // (Walking K from a per-workgroup rotated start, against power-of-two row strides all starting on
  // the same HBM channel, was measured: no gain at N x K = 12800 x 8192, 3-6 % slower at 32768 x 4096.)
  const int nfull = (ks1 - ks0) / U;
  // Full batches of U K-steps run as a rolling window of two halves: when a half has been multiplied, its slots
  // are reloaded for the next batch at once, so U / 2 .. U K-steps of weight loads stay in flight from the first
  // batch to the last (issuing a whole batch only after the previous one was consumed drained the memory pipe
  // between batches).  The norm pre-pass holds workgroup barriers: every wave runs it exactly once - behind its
  // first batch of weight loads when it has a full batch, in front of the tail loop otherwise.
  constexpr int H = U / 2;
  u32x4 wv[U][WN], xv[U];
  auto issue = [&](int ks, auto u0c) {
    constexpr int u0 = decltype(u0c)::value;
#pragma unroll
    for (int u = u0; u < u0 + H; ++u)
#pragma unroll
      for (int j = 0; j < WN; ++j) wv[u][j] = __builtin_nontemporal_load((const u32x4*)(wr[j] + (ks + u) * 32));
    if constexpr (!NORM) {
#pragma unroll
      for (int u = u0; u < u0 + H; ++u) xv[u] = load_x(ks + u);
    }
  };
  auto consume = [&](int ks, auto u0c) {
    constexpr int u0 = decltype(u0c)::value;
    if constexpr (NORM) {
#pragma unroll
      for (int u = u0; u < u0 + H; ++u) xv[u] = load_x(ks + u);
    }
#pragma unroll
    for (int u = u0; u < u0 + H; ++u)
#pragma unroll
      for (int j = 0; j < WN; ++j)
        acc[j] = F::mfma(__builtin_bit_cast(VT, xv[u]), __builtin_bit_cast(VT, wv[u][j]), acc[j]);
  };
  using C0 = std::integral_constant<int, 0>;
  using CH = std::integral_constant<int, H>;
  if (nfull == 0) {
    norm_prepass();
  } else {
    issue(ks0, C0{});
    issue(ks0, CH{});
    norm_prepass();
    for (int bi = 0; bi < nfull; ++bi) {
      const int ks = ks0 + bi * U;
      const bool more = bi + 1 < nfull;
      consume(ks, C0{});
      if (more) issue(ks + U, C0{});
      consume(ks, CH{});
      if (more) issue(ks + U, CH{});
    }
  }
  for (int ks = ks0 + nfull * U; ks < ks1; ++ks) {
    const u32x4 xv = load_x(ks);
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const u32x4 wv = __builtin_nontemporal_load((const u32x4*)(wr[j] + ks * 32));
      acc[j] = F::mfma(__builtin_bit_cast(VT, xv), __builtin_bit_cast(VT, wv), acc[j]);
    }
  }
  // accumulator: row (token) = 4*lg + r, column = li of group j.  Wave g sums column group g.
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    float* slot = red + ((wk * WN + j) * 64 + lane) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) slot[r] = acc[j][r];
  }
  __syncthreads();
  float inv[4] = {1.f, 1.f, 1.f, 1.f};                  // per-row scale of the fused norm (rows 4 lg .. 4 lg + 3)
  if constexpr (NORM) {
#pragma unroll
    for (int r = 0; r < 4; ++r) inv[r] = rsqrtf(s_ss[4 * lg + r] / (float)K + eps);
  }
  if constexpr (SILU) {
    if (wk != 0) return;
    f32x4 g = {0.f, 0.f, 0.f, 0.f}, up = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const float* og = red + ((q * WN + 0) * 64 + lane) * 4;
      const float* ou = red + ((q * WN + 1) * 64 + lane) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) g[r] += og[r], up[r] += ou[r];
    }
    const int col = n0 + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * lg + r;
      // the separate kernels round gate and up to 16 bits first: do the same, then activate
      const float gr = IS_BF16 ? (float)(__bf16)(g[r] * inv[r]) : (float)(_Float16)g[r];
      const float ur = IS_BF16 ? (float)(__bf16)(up[r] * inv[r]) : (float)(_Float16)up[r];
      const float mine = gr / (1.f + __expf(-gr)) * ur;
      const float other = __shfl_xor(mine, 1);
      if (row < M && col < N && (li & 1) == 0) *(uint32_t*)(C + (long long)row * ldc + col) = F::pack2(mine, other);
    }
    return;
  }
  if (wk >= WN) return;
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < NW; ++q) {
    const float* o = red + ((q * WN + wk) * 64 + lane) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) sum[r] += o[r];
  }
  // lanes li and li^1 hold neighbouring columns: the even lane stores both (N, ldc are even)
  const int col = n0 + wk * 16 + li;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * lg + r;
    const float mine = sum[r] * inv[r];
    const float other = __shfl_xor(mine, 1);
    if (row < M && col < N && (li & 1) == 0) *(uint32_t*)(C + (long long)row * ldc + col) = F::pack2(mine, other);
  }
}

// Decode GEMM for ONE row on the vector ALUs (bf16): C[1, N] = X[1, K] W[N, K]^T, or with SILU
// C[1, I] = silu(X Wgate^T) * (X Wup^T) for a gate_up weight [2 I, K].  At one row an MFMA tile is 15/16 padding and
// its operand layout dictates 64-byte runs per weight row and load instruction; here a wave streams eight weight rows
// with eight lanes per row, so every load instruction reads eight contiguous 128-byte runs, in batches of eight
// pieces per lane with the next batch requested before the current one is multiplied (v_dot2_f32_bf16: two products
// per instruction, no unpacking).  Wave k of NW owns the k-th slice of K of the workgroup's eight (SILU: four gate +
// four up) rows; the slices are summed through LDS.  Without NORM the activations are read with the weights they
// meet (L1 / L2 hits).  NORM: the RMSNorm in front of the projection rides along exactly as in
// gemm_nt_skinny_kernel (h = bf16(X + R), workgroup 0 writes h to Hout, the products use bf16(h * G) staged in LDS and
// the result is scaled by rsqrt(mean(h^2) + eps)).  Without NORM and SILU, G may point to a bias [N] that the
// epilogue adds.  Cold weights, one row, MFMA form -> this form: 50 MB 15.7 -> 10.6 us,
// 101 MB 25.2 -> 18.0, 201 MB with SiLU 39.9 -> 34.0, 1245 MB 224.5 -> 182.0.  (MR = 2 exists for measurements: with
// two rows the activation loads double and the MFMA form wins on most shapes; a form that stages two to four rows in
// LDS and reads them from there was built and measured too - 50 MB 15.1 vs 16.5 us and 101 MB 19.6 vs 23.1 at two
// rows, but 201 MB with SiLU 45.1 vs 40.4, 1245 MB 237 vs 226, and slower everywhere at four rows - and dropped.)
template <int NW, int MR, bool SILU, bool NORM = false>
__global__ __launch_bounds__(64 * NW) void gemm_nt_dot_kernel(const uint16_t* __restrict__ X,
                                                              const uint16_t* __restrict__ W,
                                                              uint16_t* __restrict__ C, int M, int N, int K,
                                                              long long ldx, long long ldw, long long ldc,
                                                              const uint16_t* __restrict__ R = nullptr,
                                                              const uint16_t* __restrict__ G = nullptr,
                                                              uint16_t* __restrict__ Hout = nullptr, float eps = 0.f) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  using F = Frag<true>;
  constexpr int U = 8;
  __shared__ float red[NW * MR * 8];
  __shared__ float s_ss[MR];
  extern __shared__ __attribute__((aligned(16))) char tbuf[];     // NORM: [MR][K * 2 + 64] bytes
  const long long trow = (long long)K * 2 + 64;
  const int wk = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane >> 3, kl = lane & 7;
  const int col = SILU ? (int)blockIdx.x * 4 + (r & 3) : (int)blockIdx.x * 8 + r;
  const uint16_t* wr = W + ((SILU ? (long long)(r >> 2) * N : 0ll) + min(col, N - 1)) * ldw;
  const uint16_t* xr[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) xr[m] = X + (long long)min(m, M - 1) * ldx;
  const int steps = (K + 63) / 64;
  const int per = (steps + NW - 1) / NW;
  const int sb = min(wk * per, steps), se = min(sb + per, steps);
  u32x4 wa[U], wb[U], xa[MR][U], xb[MR][U];
  // K is a multiple of 32, so a 16-byte piece is in or out as a whole; a piece past the end of the slice reads the
  // slice's last piece instead and meets zeros
  auto ld = [&](u32x4(&w)[U], u32x4(&xv)[MR][U], int s0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = (s0 + u) * 64 + kl * 8;
      const int kc = min(k, K - 8);
      w[u] = __builtin_nontemporal_load((const u32x4*)(wr + kc));
      if constexpr (!NORM) {
        const bool in = s0 + u < se && k < K;
#pragma unroll
        for (int m = 0; m < MR; ++m) {
          const u32x4 v = *(const u32x4*)(xr[m] + kc);
          xv[m][u] = in ? v : u32x4{0, 0, 0, 0};
        }
      }
    }
  };
  auto ld_x_lds = [&](u32x4(&xv)[MR][U], int s0) {      // NORM: the activations of a batch, from LDS, at multiply time
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = (s0 + u) * 64 + kl * 8;
      const bool in = s0 + u < se && k < K;
      const int kc = min(k, K - 8);
#pragma unroll
      for (int m = 0; m < MR; ++m) {
        const u32x4 v = *(const u32x4*)(tbuf + m * trow + kc * 2);
        xv[m][u] = in ? v : u32x4{0, 0, 0, 0};
      }
    }
  };
  auto norm_prepass = [&]() {
    if constexpr (NORM) {
      if (threadIdx.x < MR) s_ss[threadIdx.x] = 0.f;
      __syncthreads();
      const int nvec = K / 8;
      for (int row = 0; row < min(M, MR); ++row) {
        float ss = 0.f;
        for (int c = threadIdx.x; c < nvec; c += 64 * NW) {
          const u32x4 xv = *(const u32x4*)(X + (long long)row * ldx + c * 8);
          const u32x4 gv = *(const u32x4*)(G + c * 8);
          u32x4 rv = {0, 0, 0, 0}, hb, t;
          if (R) rv = *(const u32x4*)(R + (long long)row * K + c * 8);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float lo = __builtin_bit_cast(float, xv[i] << 16), hi = __builtin_bit_cast(float, xv[i] & 0xffff0000u);
            hb[i] = xv[i];
            if (R) {
              lo += __builtin_bit_cast(float, rv[i] << 16);
              hi += __builtin_bit_cast(float, rv[i] & 0xffff0000u);
              hb[i] = F::pack2(lo, hi);      // the sum rounded to 16 bits is what the norm sees (as the separate kernel)
              lo = __builtin_bit_cast(float, hb[i] << 16), hi = __builtin_bit_cast(float, hb[i] & 0xffff0000u);
            }
            ss += lo * lo + hi * hi;
            t[i] = F::pack2(lo * __builtin_bit_cast(float, gv[i] << 16), hi * __builtin_bit_cast(float, gv[i] & 0xffff0000u));
          }
          *(u32x4*)(tbuf + row * trow + c * 16) = t;
          if (R && blockIdx.x == 0) *(u32x4*)(Hout + (long long)row * K + c * 8) = hb;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        if (lane == 0) atomicAdd(&s_ss[row], ss);
      }
      __syncthreads();
    }
  };
  float acc[MR];
#pragma unroll
  for (int m = 0; m < MR; ++m) acc[m] = 0.f;
  auto mm = [&](const u32x4(&w)[U], const u32x4(&xv)[MR][U]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // (through scalars: __builtin_bit_cast applied to a vector ELEMENT expression read element 0 every time)
        const unsigned wi = w[u][i];
#pragma unroll
        for (int m = 0; m < MR; ++m) {
          const unsigned xi = xv[m][u][i];
          acc[m] = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, wi), __builtin_bit_cast(bf16x2, xi), acc[m], false);
        }
      }
  };
  // The norm pre-pass holds workgroup barriers: every wave runs it exactly once, behind its first batch of weight loads.
  // (One batch ahead, not two: with both batches of a slice requested before the first multiply the kernel ran a
  // third slower - 1245 MB in 272 instead of 182 us.)
  if (sb < se) ld(wa, xa, sb);
  norm_prepass();
  for (int s0 = sb; s0 < se; s0 += 2 * U) {
    if (s0 + U < se) ld(wb, xb, s0 + U);
    if constexpr (NORM) ld_x_lds(xa, s0);
    mm(wa, xa);
    if (s0 + 2 * U < se) ld(wa, xa, s0 + 2 * U);
    if (s0 + U < se) {
      if constexpr (NORM) ld_x_lds(xb, s0 + U);
      mm(wb, xb);
    }
  }
#pragma unroll
  for (int m = 0; m < MR; ++m) {
    acc[m] += __shfl_xor(acc[m], 1);
    acc[m] += __shfl_xor(acc[m], 2);
    acc[m] += __shfl_xor(acc[m], 4);
    if (kl == 0) red[(wk * MR + m) * 8 + r] = acc[m];
  }
  __syncthreads();
  // thread (m, r) of the first MR * 8 sums the K slices of row r for activation row m
  if ((int)threadIdx.x < MR * 8) {
    const int m = threadIdx.x >> 3, rr = threadIdx.x & 7;
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < NW; ++q) sum += red[(q * MR + m) * 8 + rr];
    if constexpr (NORM) sum *= rsqrtf(s_ss[m] / (float)K + eps);
    if constexpr (SILU) {
      // rows 0..3 are gate sums, rows 4..7 the up sums of the same columns (four lanes further on); both are
      // rounded to bf16 first, as the separate kernels do
      const float up = __shfl_down(sum, 4);
      const int c = (int)blockIdx.x * 4 + rr;
      if (rr < 4 && c < N && m < M) {
        const float gr = (float)(__bf16)sum, ur = (float)(__bf16)up;
        C[(long long)m * ldc + c] = __builtin_bit_cast(uint16_t, (__bf16)(gr / (1.f + __expf(-gr)) * ur));
      }
    } else {
      const int c = (int)blockIdx.x * 8 + rr;
      if constexpr (!NORM) {                  // (without the fused norm G is an optional bias [N])
        if (G && c < N) sum += __builtin_bit_cast(float, (uint32_t)G[c] << 16);
      }
      if (c < N && m < M) C[(long long)m * ldc + c] = __builtin_bit_cast(uint16_t, (__bf16)sum);
    }
  }
}

// Decode GEMM for 2 .. 16 rows on MFMAs with 128-byte runs per weight row and load instruction.
//
// The B operand of a 16x16x32 MFMA wants lane (li, lg) to hold k = 8 lg .. 8 lg + 7 of weight row li: four lanes, 64
// contiguous bytes per row and load instruction (gemm_nt_skinny_kernel).  The same one-row kernel on the vector ALUs
// runs 13 - 20 % faster with eight lanes per row than with four (50 MB 11.1 vs 13.7 us, 1245 MB 183 vs 212 us; sixteen
// lanes are slower again), so here a load instruction covers EIGHT rows x 128 bytes: lane (li, lg) loads k-group
// 4 (li >> 3) + lg of row li & 7 of a 64-deep K step.  Lanes li < 8 then hold k-groups 0 .. 3 of the eight rows and
// lanes li >= 8 k-groups 4 .. 7 of the SAME rows, and two MFMAs per step sort it out: the first multiplies with the
// activations' k-groups 0 .. 3 (its columns 0 .. 7 are valid partial sums, 8 .. 15 meaningless), the second with
// k-groups 4 .. 7 (columns 8 .. 15 valid); column j of the result is column j of the first accumulator plus column
// j + 8 of the second.  Twice the MFMAs per weight byte - the matrix pipe idles either way at these row counts.
// Everything else (K slices over the waves, WN column groups sharing the activation fragments, SILU, NORM, the
// rolling window of loads) is gemm_nt_skinny_kernel's.
template <bool IS_BF16, int WN, int NW, bool SILU = false, bool NORM = false>
__global__ __launch_bounds__(64 * NW) void gemm_nt_skinny128_kernel(const uint16_t* __restrict__ X,
                                                                const uint16_t* __restrict__ W,
                                                                uint16_t* __restrict__ C, int M, int N, int K,
                                                                long long ldx, long long ldw, long long ldc,
                                                                const uint16_t* __restrict__ R,
                                                                const uint16_t* __restrict__ G,
                                                                uint16_t* __restrict__ Hout, float eps) {
  static_assert(!NORM || IS_BF16, "the fused norm is built for bf16");
  static_assert(!SILU || WN % 2 == 0, "the fused activation pairs gate group g with up group g + WN / 2");
  static_assert(WN <= NW, "one wave per column group sums the K slices");
  using F = Frag<IS_BF16>;
  using VT = typename F::T;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  constexpr int GN = SILU ? WN / 2 : WN;                 // groups of eight OUTPUT columns per workgroup
  constexpr int BN_ = 8 * GN;
  constexpr int U = 16 / WN;                             // 64-deep K steps per batch (16 weight loads in flight per lane)
  const int up_off = SILU ? N : 0;                       // (SILU: N is I, the number of OUTPUT columns)
  __shared__ float red[NW * WN * 64 * 4];                // [K slice][column group][lane][4]
  const int wk = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 15, lg = lane >> 4;
  const int lrow = li & 7, kg = (li >> 3) * 4 + lg;      // weight row inside the group, k-group inside the step
  const int n0 = blockIdx.x * BN_;
  const int ksteps = (K + 63) / 64;
  const int per = (ksteps + NW - 1) / NW;
  const int ks0 = min(wk * per, ksteps), ks1 = min(ks0 + per, ksteps);
  const uint16_t* xr = X + (long long)min(li, M - 1) * ldx + lg * 8;
  const uint16_t* wr[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j)                            // clamped: out-of-range columns are not stored
    wr[j] = W + (long long)((j / GN) * up_off + min(n0 + (j % GN) * 8 + lrow, N - 1)) * ldw;
  const bool xvalid = li < M;
  f32x4 acc1[WN], acc2[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j) acc1[j] = acc2[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  extern __shared__ __attribute__((aligned(16))) char tbuf[];     // NORM: [M][K * 2 + 64] bytes (rows padded: banks)
  __shared__ float s_ss[16];
  const long long trow = (long long)K * 2 + 64;
  auto norm_prepass = [&]() {
    if constexpr (NORM) {
      if (threadIdx.x < 16) s_ss[threadIdx.x] = 0.f;
      __syncthreads();
      const int nvec = K / 8;
      for (int row = 0; row < M; ++row) {
        float ss = 0.f;
        for (int c = threadIdx.x; c < nvec; c += 64 * NW) {
          const u32x4 xv = *(const u32x4*)(X + (long long)row * ldx + c * 8);
          const u32x4 gv = *(const u32x4*)(G + c * 8);
          u32x4 rv = {0, 0, 0, 0}, hb, t;
          if (R) rv = *(const u32x4*)(R + (long long)row * K + c * 8);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            float lo = __builtin_bit_cast(float, xv[i] << 16), hi = __builtin_bit_cast(float, xv[i] & 0xffff0000u);
            hb[i] = xv[i];
            if (R) {
              lo += __builtin_bit_cast(float, rv[i] << 16);
              hi += __builtin_bit_cast(float, rv[i] & 0xffff0000u);
              hb[i] = F::pack2(lo, hi);      // the sum rounded to 16 bits is what the norm sees (as the separate kernel)
              lo = __builtin_bit_cast(float, hb[i] << 16), hi = __builtin_bit_cast(float, hb[i] & 0xffff0000u);
            }
            ss += lo * lo + hi * hi;
            t[i] = F::pack2(lo * __builtin_bit_cast(float, gv[i] << 16), hi * __builtin_bit_cast(float, gv[i] & 0xffff0000u));
          }
          *(u32x4*)(tbuf + row * trow + c * 16) = t;
          if (R && blockIdx.x == 0) *(u32x4*)(Hout + (long long)row * K + c * 8) = hb;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o);
        if (lane == 0) atomicAdd(&s_ss[row], ss);
      }
      __syncthreads();
    }
  };
  // activation fragment of K step ks, half h (k-groups 4 h .. 4 h + 3); zero past the end of K and for rows past M
  auto load_x = [&](int ks, int h) -> u32x4 {
    const int k = ks * 64 + h * 32 + lg * 8;
    const bool ok = xvalid && k < K;
    if constexpr (NORM) {
      const u32x4 v = *(const u32x4*)(tbuf + min(li, M - 1) * trow + min(k, K - 8) * 2);
      return ok ? v : u32x4{0, 0, 0, 0};
    } else {
      const u32x4 v = *(const u32x4*)(xr + min(k, K - 8) - lg * 8);
      return ok ? v : u32x4{0, 0, 0, 0};
    }
  };
  const int nfull = (ks1 - ks0) / U;
  constexpr int H = U / 2;
  u32x4 wv[U][WN], xv1[U], xv2[U];
  auto issue = [&](int ks, auto u0c) {
    constexpr int u0 = decltype(u0c)::value;
#pragma unroll
    for (int u = u0; u < u0 + H; ++u)
#pragma unroll
      for (int j = 0; j < WN; ++j)
        wv[u][j] = __builtin_nontemporal_load((const u32x4*)(wr[j] + min((ks + u) * 64 + kg * 8, K - 8)));
    if constexpr (!NORM) {
#pragma unroll
      for (int u = u0; u < u0 + H; ++u) xv1[u] = load_x(ks + u, 0), xv2[u] = load_x(ks + u, 1);
    }
  };
  auto consume = [&](int ks, auto u0c) {
    constexpr int u0 = decltype(u0c)::value;
    if constexpr (NORM) {
#pragma unroll
      for (int u = u0; u < u0 + H; ++u) xv1[u] = load_x(ks + u, 0), xv2[u] = load_x(ks + u, 1);
    }
#pragma unroll
    for (int u = u0; u < u0 + H; ++u)
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        acc1[j] = F::mfma(__builtin_bit_cast(VT, xv1[u]), __builtin_bit_cast(VT, wv[u][j]), acc1[j]);
        acc2[j] = F::mfma(__builtin_bit_cast(VT, xv2[u]), __builtin_bit_cast(VT, wv[u][j]), acc2[j]);
      }
  };
  using C0 = std::integral_constant<int, 0>;
  using CH = std::integral_constant<int, H>;
  if (nfull == 0) {
    norm_prepass();
  } else {
    issue(ks0, C0{});
    issue(ks0, CH{});
    norm_prepass();
    for (int bi = 0; bi < nfull; ++bi) {
      const int ks = ks0 + bi * U;
      const bool more = bi + 1 < nfull;
      consume(ks, C0{});
      if (more) issue(ks + U, C0{});
      consume(ks, CH{});
      if (more) issue(ks + U, CH{});
    }
  }
  for (int ks = ks0 + nfull * U; ks < ks1; ++ks) {
    const u32x4 x1 = load_x(ks, 0), x2 = load_x(ks, 1);
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const u32x4 w = __builtin_nontemporal_load((const u32x4*)(wr[j] + min(ks * 64 + kg * 8, K - 8)));
      acc1[j] = F::mfma(__builtin_bit_cast(VT, x1), __builtin_bit_cast(VT, w), acc1[j]);
      acc2[j] = F::mfma(__builtin_bit_cast(VT, x2), __builtin_bit_cast(VT, w), acc2[j]);
    }
  }
  // accumulator: row (token) = 4*lg + r, column = li.  Column j < 8 of the result = acc1 column j + acc2 column j + 8
  // (eight lanes further on); wave g then sums column group g over the K slices.
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    float* slot = red + ((wk * WN + j) * 64 + lane) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) slot[r] = acc1[j][r] + __shfl_down(acc2[j][r], 8);
  }
  __syncthreads();
  float inv[4] = {1.f, 1.f, 1.f, 1.f};                  // per-row scale of the fused norm (rows 4 lg .. 4 lg + 3)
  if constexpr (NORM) {
#pragma unroll
    for (int r = 0; r < 4; ++r) inv[r] = rsqrtf(s_ss[4 * lg + r] / (float)K + eps);
  }
  if constexpr (SILU) {
    if (wk >= GN) return;                      // wave g: gate group g and up group g + GN
    f32x4 g = {0.f, 0.f, 0.f, 0.f}, up = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const float* og = red + ((q * WN + wk) * 64 + lane) * 4;
      const float* ou = red + ((q * WN + wk + GN) * 64 + lane) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) g[r] += og[r], up[r] += ou[r];
    }
    const int col = n0 + wk * 8 + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 4 * lg + r;
      // the separate kernels round gate and up to 16 bits first: do the same, then activate
      const float gr = IS_BF16 ? (float)(__bf16)(g[r] * inv[r]) : (float)(_Float16)g[r];
      const float ur = IS_BF16 ? (float)(__bf16)(up[r] * inv[r]) : (float)(_Float16)up[r];
      const float mine = gr / (1.f + __expf(-gr)) * ur;
      const float other = __shfl_xor(mine, 1);
      if (row < M && li < 8 && col < N && (li & 1) == 0) *(uint32_t*)(C + (long long)row * ldc + col) = F::pack2(mine, other);
    }
    return;
  }
  if (wk >= WN) return;
  f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int q = 0; q < NW; ++q) {
    const float* o = red + ((q * WN + wk) * 64 + lane) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) sum[r] += o[r];
  }
  // lanes li and li^1 hold neighbouring columns: the even lane stores both (N, ldc are even)
  const int col = n0 + wk * 8 + li;
  float bias = 0.f;                            // (without the fused norm G is an optional bias [N])
  if constexpr (!NORM) {
    if (G && li < 8 && col < N) bias = IS_BF16 ? __builtin_bit_cast(float, (uint32_t)G[col] << 16) : (float)__builtin_bit_cast(_Float16, G[col]);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * lg + r;
    const float mine = sum[r] * inv[r] + bias;
    const float other = __shfl_xor(mine, 1);
    if (row < M && li < 8 && col < N && (li & 1) == 0) *(uint32_t*)(C + (long long)row * ldc + col) = F::pack2(mine, other);
  }
}

// Shape of gemm_nt_skinny128_kernel for a call: WN column groups of eight, NW waves (K slices).  Measured on cold
// weights at 2 / 4 / 8 / 16 rows against gemm_nt_skinny_kernel (us): 6144 x 4096 12.6 / 14.4 / 15.2 / 16.4 with WN 4
// (64-byte-run kernel 15.7 / 16.7 / 17.0 / 17.9), 4096 x 4096 9.0 / 9.1 / 9.4 / 10.2 with WN 2 (9.2 / 9.5 / 10.0 / 11.2),
// 4096 x 12288 20.6 / 21.3 / 23.5 / 28.8 with WN 2 (25.7 / 27.1 / 29.2 / 32.8), 151936 x 4096 189 / 190 / 197 / 231 with
// WN 4 (226 / 233 / 248 / 268); WN 1 and eight waves lose on every shape.
// TDIST_SKINNY128: 0 = keep gemm_nt_skinny_kernel, "<WN><NW>" (e.g. 24) = force a shape.
// Long K with a few hundred 64-column blocks (12800 x 8192: 200 workgroups of eight waves, one per compute unit, all
// done at the same time) runs 36 - 42 us at 2 .. 16 rows with WN 8 / NW 8 against 44 - 52 us with WN 4 / NW 4, whose 400
// workgroups sit one or two to a compute unit.
// Cold weights at 4 / 8 / 16 rows over 88 weight shapes (scripts/tune_gemm_table.py): up to 2048 columns eight-column
// blocks are 1.17 - 1.26 x faster (more workgroups than compute units again); between 4096 and 6144 columns 16-column
// blocks make 256 .. 384 workgroups - some compute units stream two, the others one - and 32-column blocks of eight
// waves run 1.05 - 1.24 x faster (5120 columns, K 1024 .. 4096); with K below 4096 a wave of four has less than two
// batches of loads to overlap and eight waves per workgroup are 1.05 - 1.27 x ahead up to 16384 columns
// (2048 x 2048, 3072 x 3072, 3584 x 3584, 8192 x 1024 / 3584).
static int skinny128_choice(int N, int K = 0, int M = 16) {
  static const int v = getenv("TDIST_SKINNY128") ? atoi(getenv("TDIST_SKINNY128")) : -1;
  if (v >= 0) return v;
  (void)M;
  if (K >= 8192 && N >= 12288) return 88;
  if (N <= 2048) return K > 0 && K <= 2048 ? 28 : 14;
  if (N > 4096 && N < 6144) return 48;
  if (N > 8192 && N < 16384) return 88;      // (32-column blocks: 256 .. 512 workgroups, the same bad fit; 1.05 - 1.14 x)
  if (K > 0 && K < 4096 && N < 16384) return N >= 6144 ? 48 : 28;
  return N >= 6144 ? 44 : 24;
}

// TDIST_DECODE_DOT=0 keeps one-row decode GEMMs on the MFMA kernel (A/B measurements)
static bool decode_dot_enabled() {
  static const bool on = !(getenv("TDIST_DECODE_DOT") && atoi(getenv("TDIST_DECODE_DOT")) == 0);
  return on;
}

// Launch of gemm_nt_dot_kernel (bf16).  nw: K slices per workgroup, 0 = four (the best of 1 / 2 / 4 / 8 on every
// decode shape of a dense model with K >= 4096).  A wave wants at least TWO batches of eight 64-deep steps - the second
// is requested while the first is multiplied - so shorter K uses fewer slices: LM heads of 151936 x 2048 and
// 256000 x 3072 run 87.5 and 226 us with two slices against 114 and 290 us with four (one and one and a half batches
// per wave; scripts/dev/sweep_linear_kernels.py).
template <int MR, bool SILU, bool NORM>
static void launch_dot(const uint16_t* X, const uint16_t* W, uint16_t* C, int M, int N, int K, long long ldx,
                       long long ldw, long long ldc, int nw, hipStream_t s, const uint16_t* R = nullptr,
                       const uint16_t* G = nullptr, uint16_t* Hout = nullptr, float eps = 0.f) {
  const int cb = SILU ? 4 : 8;
  const dim3 grid((N + cb - 1) / cb);
  const bool chosen = nw > 0;
  if (nw <= 0) nw = 4;
  while (nw > 1 && K < 64 * 8 * 2 * nw) nw >>= 1;
  // Whole batches per wave: K = 5120 over four slices is two and a half batches each, and the half batch costs a
  // whole one (cold weights, 5120 .. 155136 columns: two slices run 1.11 - 1.17 x faster than four; 24 layers of
  // Qwen3-32B 4.77 -> 4.41 ms).  Only from four slices to two: ONE slice per workgroup loses again (K = 2560: 36 layers
  // of Qwen3-4B 2.58 ms with one slice against 2.39 ms with two; 5120 x 13824 30.6 against 27.0 us).
  // TDIST_DOT_WHOLE_BATCHES=0 keeps four slices.
  static const bool whole = !(getenv("TDIST_DOT_WHOLE_BATCHES") && atoi(getenv("TDIST_DOT_WHOLE_BATCHES")) == 0);
  if (whole && !chosen && nw == 4 && K % (64 * 8 * 4) == 64 * 8 * 2) nw = 2;
  const size_t lds = NORM ? (size_t)MR * ((size_t)K * 2 + 64) : 0;
  if (nw == 1)
    hipLaunchKernelGGL((gemm_nt_dot_kernel<1, MR, SILU, NORM>), grid, dim3(64), lds, s, X, W, C, M, N, K, ldx, ldw, ldc, R, G, Hout, eps);
  else if (nw == 2)
    hipLaunchKernelGGL((gemm_nt_dot_kernel<2, MR, SILU, NORM>), grid, dim3(128), lds, s, X, W, C, M, N, K, ldx, ldw, ldc, R, G, Hout, eps);
  else if (nw == 4)
    hipLaunchKernelGGL((gemm_nt_dot_kernel<4, MR, SILU, NORM>), grid, dim3(256), lds, s, X, W, C, M, N, K, ldx, ldw, ldc, R, G, Hout, eps);
  else
    hipLaunchKernelGGL((gemm_nt_dot_kernel<8, MR, SILU, NORM>), grid, dim3(512), lds, s, X, W, C, M, N, K, ldx, ldw, ldc, R, G, Hout, eps);
}

// C[M, N] = X W^T (silu == 0) or C[M, N] = silu(X W[0:N]^T) * (X W[N:2N]^T) (silu != 0, W [2 N, K]) for M <= 2, bf16,
// on the vector ALUs (gemm_nt_dot_kernel; the entry the probes time - the library's own decode entry points choose
// this kernel at one row).  nw: K slices per workgroup (1 / 2 / 4 / 8; 0 = default).
TDIST_API int tdist_gemm_nt_dot(const void* X, const void* W, void* C, int M, int N, int K, long long ldx, long long ldw,
                                long long ldc, int silu, int nw, void* stream) {
  TDIST_REQUIRE(M > 0 && M <= 2 && N > 0, "M must be 1 or 2");
  TDIST_REQUIRE(K >= 128 && K % 32 == 0, "K must be a multiple of 32 and >= 128");
  TDIST_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0, "row strides must keep the vector alignment");
  TDIST_REQUIRE((((uintptr_t)X | (uintptr_t)W) & 15) == 0 && C, "operands must be aligned");
  TDIST_REQUIRE(nw == 0 || nw == 1 || nw == 2 || nw == 4 || nw == 8, "nw must be 0, 1, 2, 4 or 8");
  const uint16_t *x = (const uint16_t*)X, *w = (const uint16_t*)W;
  uint16_t* c = (uint16_t*)C;
  hipStream_t s = (hipStream_t)stream;
  if (silu) {
    if (M == 1) launch_dot<1, true, false>(x, w, c, M, N, K, ldx, ldw, ldc, nw, s);
    else launch_dot<2, true, false>(x, w, c, M, N, K, ldx, ldw, ldc, nw, s);
  } else {
    if (M == 1) launch_dot<1, false, false>(x, w, c, M, N, K, ldx, ldw, ldc, nw, s);
    else launch_dot<2, false, false>(x, w, c, M, N, K, ldx, ldw, ldc, nw, s);
  }
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

template <bool IS_BF16>
static void launch_skinny(const void* A, const void* B, void* C, int M, int N, int K, long long lda, long long ldb,
                          long long ldc, hipStream_t s, const uint16_t* bias = nullptr) {
  // Column-block size (measured on MI355X, K 2048..12288): with 1-4 tokens the activation loads
  // are negligible and small blocks (more workgroups) win; with more tokens sharing one activation
  // fragment between 2-4 column groups pays as soon as the grid still has a few hundred workgroups.
  if constexpr (IS_BF16) {
    // one row: the vector-ALU form (gemm_nt_dot_kernel) streams the weights 18 - 32 % faster
    if (M == 1 && decode_dot_enabled() && ((uintptr_t)C & 1) == 0) {
      launch_dot<1, false, false>((const uint16_t*)A, (const uint16_t*)B, (uint16_t*)C, M, N, K, lda, ldb, ldc, 0, s, nullptr, bias);
      return;
    }
  }
  if (const int ch = skinny128_choice(N, K, M); ch > 0 && M >= 2) {
    const uint16_t *a = (const uint16_t*)A, *b = (const uint16_t*)B;
    uint16_t* c = (uint16_t*)C;
    const int wn = ch / 10, nw = ch % 10;
#define TDIST_S128(WN_, NW_)                                                                                          \
  if (wn == WN_ && nw == NW_) {                                                                                       \
    hipLaunchKernelGGL((gemm_nt_skinny128_kernel<IS_BF16, WN_, NW_>), dim3((N + 8 * WN_ - 1) / (8 * WN_)), dim3(64 * NW_), \
                       0, s, a, b, c, M, N, K, lda, ldb, ldc, (const uint16_t*)nullptr, bias,                         \
                       (uint16_t*)nullptr, 0.f);                                                                      \
    return;                                                                                                           \
  }
    TDIST_S128(1, 4) TDIST_S128(2, 4) TDIST_S128(4, 4) TDIST_S128(1, 8) TDIST_S128(2, 8) TDIST_S128(4, 8) TDIST_S128(8, 8)
#undef TDIST_S128
  }
  int bn;
  if (M <= 4)
    bn = N >= 16384 ? 32 : 16;
  else
    bn = N >= 32768 ? 64 : (N >= 6144 ? 32 : 16);
  static const int forced = getenv("TDIST_SKINNY_BN") ? atoi(getenv("TDIST_SKINNY_BN")) : 0;   // tuning knob
  if (forced == 16 || forced == 32 || forced == 64) bn = forced;
  const dim3 grid((N + bn - 1) / bn);
  const bool wide = bn == 16 && grid.x < 1024 && K >= 4096;      // 8 waves: twice the loads in flight per workgroup
  const uint16_t *a = (const uint16_t*)A, *b = (const uint16_t*)B;
  uint16_t* c = (uint16_t*)C;
  if (bn == 64)
    hipLaunchKernelGGL((gemm_nt_skinny_kernel<IS_BF16, 4, 4>), grid, dim3(256), 0, s, a, b, c, M, N, K, lda, ldb, ldc);
  else if (bn == 32)
    hipLaunchKernelGGL((gemm_nt_skinny_kernel<IS_BF16, 2, 4>), grid, dim3(256), 0, s, a, b, c, M, N, K, lda, ldb, ldc);
  else if (wide)
    hipLaunchKernelGGL((gemm_nt_skinny_kernel<IS_BF16, 1, 8>), grid, dim3(512), 0, s, a, b, c, M, N, K, lda, ldb, ldc);
  else
    hipLaunchKernelGGL((gemm_nt_skinny_kernel<IS_BF16, 1, 4>), grid, dim3(256), 0, s, a, b, c, M, N, K, lda, ldb, ldc);
}

// gemm_nt_skinny128_kernel with a given shape (the entry the probes time; bf16, 2 .. 16 rows): wn column groups of
// eight per workgroup (1 / 2 / 4 / 8), nw K slices (4 / 8; wn <= nw).
TDIST_API int tdist_gemm_nt_skinny128(const void* A, const void* B, void* C, int M, int N, int K, long long lda,
                                      long long ldb, long long ldc, int wn, int nw, void* stream) {
  TDIST_REQUIRE(M > 0 && M <= 16 && N > 0 && N % 2 == 0, "M must be in [1, 16], N even");
  TDIST_REQUIRE(K >= 128 && K % 32 == 0, "K must be a multiple of 32 and >= 128");
  TDIST_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldc % 2 == 0, "row strides must keep the vector alignment");
  TDIST_REQUIRE((((uintptr_t)A | (uintptr_t)B) & 15) == 0 && (((uintptr_t)C) & 3) == 0, "operands must be aligned");
  const uint16_t *a = (const uint16_t*)A, *b = (const uint16_t*)B;
  uint16_t* c = (uint16_t*)C;
  hipStream_t s = (hipStream_t)stream;
#define TDIST_S128(WN_, NW_)                                                                                          \
  if (wn == WN_ && nw == NW_) {                                                                                       \
    hipLaunchKernelGGL((gemm_nt_skinny128_kernel<true, WN_, NW_>), dim3((N + 8 * WN_ - 1) / (8 * WN_)), dim3(64 * NW_), 0, \
                       s, a, b, c, M, N, K, lda, ldb, ldc, (const uint16_t*)nullptr, (const uint16_t*)nullptr,        \
                       (uint16_t*)nullptr, 0.f);                                                                      \
    TDIST_HIP_TRY(hipGetLastError());                                                                                 \
    return 0;                                                                                                         \
  }
  TDIST_S128(1, 4) TDIST_S128(2, 4) TDIST_S128(4, 4) TDIST_S128(1, 8) TDIST_S128(2, 8) TDIST_S128(4, 8) TDIST_S128(8, 8)
#undef TDIST_S128
  TDIST_REQUIRE(false, "no such shape of the kernel");
  return 0;
}

// C[M, N] = rmsnorm(X [+ R]) * G @ W^T for M <= 16, bf16 (see the NORM comment on the kernel); `silu` != 0: W is a
// gate_up weight [2 N, K] and C = silu(gate) * up.  Hout [M, K] receives X + R when R is given.
TDIST_API int tdist_gemm_nt_norm(const void* X, const void* R, const void* G, float eps, const void* W, void* C,
                                 void* Hout, int M, int N, int K, long long ldx, long long ldw, long long ldc, int silu,
                                 void* stream) {
  TDIST_REQUIRE(M > 0 && M <= 16 && N > 0 && N % 2 == 0, "M must be in [1, 16], N even");
  TDIST_REQUIRE(K >= 128 && K % 32 == 0, "K must be a multiple of 32 and >= 128");
  TDIST_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && ldc % 2 == 0, "row strides must keep the vector alignment");
  TDIST_REQUIRE((((uintptr_t)X | (uintptr_t)W | (uintptr_t)G | (uintptr_t)R | (uintptr_t)Hout) & 15) == 0 &&
                    (((uintptr_t)C) & 3) == 0, "operands must be aligned");
  TDIST_REQUIRE(G != nullptr && (R == nullptr || (Hout != nullptr && Hout != X && Hout != R)),
                "norm weight required; the residual output must be a buffer of its own");
  const uint16_t *x = (const uint16_t*)X, *w = (const uint16_t*)W, *r = (const uint16_t*)R, *g = (const uint16_t*)G;
  uint16_t *c = (uint16_t*)C, *h = (uint16_t*)Hout;
  hipStream_t s = (hipStream_t)stream;
  const long long lds = (long long)M * ((long long)K * 2 + 64);       // normalised activations of the M rows
  TDIST_REQUIRE(lds <= 48 * 1024, "fused norm: M x K too large for the activation buffer in LDS");
  if (M == 1 && decode_dot_enabled()) {
    if (silu)
      launch_dot<1, true, true>(x, w, c, M, N, K, ldx, ldw, ldc, 0, s, r, g, h, eps);
    else
      launch_dot<1, false, true>(x, w, c, M, N, K, ldx, ldw, ldc, 0, s, r, g, h, eps);
    TDIST_HIP_TRY(hipGetLastError());
    return 0;
  }
  if (const int ch = skinny128_choice(N); ch > 0) {
    if (silu && M >= 8)
      hipLaunchKernelGGL((gemm_nt_skinny128_kernel<true, 4, 4, true, true>), dim3((N + 15) / 16), dim3(256), (size_t)lds, s, x, w,
                         c, M, N, K, ldx, ldw, ldc, r, g, h, eps);
    else if (silu)
      hipLaunchKernelGGL((gemm_nt_skinny128_kernel<true, 2, 4, true, true>), dim3((N + 7) / 8), dim3(256), (size_t)lds, s, x, w,
                         c, M, N, K, ldx, ldw, ldc, r, g, h, eps);
    else if (ch / 10 >= 4 && N >= 6144)     // (the eight-wave shapes of the plain kernel are not built with the norm)
      hipLaunchKernelGGL((gemm_nt_skinny128_kernel<true, 4, 4, false, true>), dim3((N + 31) / 32), dim3(256), (size_t)lds, s, x,
                         w, c, M, N, K, ldx, ldw, ldc, r, g, h, eps);
    else
      hipLaunchKernelGGL((gemm_nt_skinny128_kernel<true, 2, 4, false, true>), dim3((N + 15) / 16), dim3(256), (size_t)lds, s, x,
                         w, c, M, N, K, ldx, ldw, ldc, r, g, h, eps);
    TDIST_HIP_TRY(hipGetLastError());
    return 0;
  }
  if (silu) {
    hipLaunchKernelGGL((gemm_nt_skinny_kernel<true, 2, 4, true, true>), dim3((N + 15) / 16), dim3(256), (size_t)lds, s, x, w,
                       c, M, N, K, ldx, ldw, ldc, r, g, h, eps);
  } else {
    // same column-block rule as launch_skinny
    int bn = M <= 4 ? (N >= 16384 ? 32 : 16) : (N >= 32768 ? 64 : (N >= 6144 ? 32 : 16));
    const dim3 grid((N + bn - 1) / bn);
    const bool wide = bn == 16 && grid.x < 1024 && K >= 4096;
    if (bn == 64)
      hipLaunchKernelGGL((gemm_nt_skinny_kernel<true, 4, 4, false, true>), grid, dim3(256), (size_t)lds, s, x, w, c, M, N, K, ldx,
                         ldw, ldc, r, g, h, eps);
    else if (bn == 32)
      hipLaunchKernelGGL((gemm_nt_skinny_kernel<true, 2, 4, false, true>), grid, dim3(256), (size_t)lds, s, x, w, c, M, N, K, ldx,
                         ldw, ldc, r, g, h, eps);
    else if (wide)
      hipLaunchKernelGGL((gemm_nt_skinny_kernel<true, 1, 8, false, true>), grid, dim3(512), (size_t)lds, s, x, w, c, M, N, K, ldx,
                         ldw, ldc, r, g, h, eps);
    else
      hipLaunchKernelGGL((gemm_nt_skinny_kernel<true, 1, 4, false, true>), grid, dim3(256), (size_t)lds, s, x, w, c, M, N, K, ldx,
                         ldw, ldc, r, g, h, eps);
  }
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

// MoE decode, second half: out[t, :] = sum_k wts[t, k] * down[ids[t, k]] @ h[t * topk + k, :]  (bf16).
// One workgroup per (16 output columns, token); wave k streams the 16 rows of expert ids[t, k] (Ip long) through
// 16x16x32 MFMAs whose M dimension holds the single activation row, scales its partial sums by the routing weight
// and the waves are summed through LDS: the top-k reduction costs nothing extra and no [pairs, H] intermediate
// exists.  Loads run as a rolling window of three 8-K-step buffers.
template <int NW>
__global__ __launch_bounds__(64 * NW) void moe_decode_down_kernel(const uint16_t* __restrict__ hbuf,
                                                                  const int* __restrict__ ids,
                                                                  const float* __restrict__ wts,
                                                                  const uint16_t* __restrict__ down,
                                                                  uint16_t* __restrict__ out, int topk, int H, int Ip,
                                                                  int E) {
  using F = Frag<true>;
  using VT = typename F::T;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  __shared__ float red[NW * 16];
  const int t = blockIdx.y, n0 = blockIdx.x * 16;
  const int wk = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 15, lg = lane >> 4;
  const int ksteps = Ip / 32;
  float part = 0.f;
  for (int kk = wk; kk < topk; kk += NW) {
    const int p = t * topk + kk;
    const int e = min(max(ids[p], 0), E - 1);
    const float wgt = wts[p];
    const uint16_t* wr = down + ((long long)e * H + min(n0 + li, H - 1)) * Ip + lg * 8;
    const uint16_t* xr = hbuf + (long long)p * Ip + lg * 8;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // (the activation fragments are loaded WITH the weights they meet: fetched at multiply time, each group of
    // eight paid an L2 round trip after its weights had already arrived)
    u32x4 wa[8], wb[8], wc[8], xa[8], xb[8], xc[8];
    auto ld = [&](u32x4(&w)[8], u32x4(&x)[8], int ks0) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        w[u] = u32x4{0, 0, 0, 0};
        x[u] = u32x4{0, 0, 0, 0};
        if (ks0 + u < ksteps) {
          w[u] = __builtin_nontemporal_load((const u32x4*)(wr + (ks0 + u) * 32));
          if (li == 0) x[u] = *(const u32x4*)(xr + (ks0 + u) * 32);
        }
      }
    };
    auto mm = [&](const u32x4(&w)[8], const u32x4(&x)[8], int ks0) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (ks0 + u < ksteps) acc = F::mfma(__builtin_bit_cast(VT, x[u]), __builtin_bit_cast(VT, w[u]), acc);
    };
    // three buffers of eight K-steps: an expert slice of up to 768 columns is requested in one go
    ld(wa, xa, 0);
    ld(wb, xb, 8);
    ld(wc, xc, 16);
    for (int ks0 = 0; ks0 < ksteps; ks0 += 24) {
      mm(wa, xa, ks0);
      ld(wa, xa, ks0 + 24);
      mm(wb, xb, ks0 + 8);
      ld(wb, xb, ks0 + 32);
      mm(wc, xc, ks0 + 16);
      ld(wc, xc, ks0 + 40);
    }
    part += wgt * acc[0];          // accumulator row 0 (the token) lives in lanes lg == 0, column li
  }
  if (lg == 0) red[wk * 16 + li] = part;
  __syncthreads();
  if (threadIdx.x < 16 && n0 + (int)threadIdx.x < H) {
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < NW; ++q) sum += red[q * 16 + threadIdx.x];
    out[(long long)t * H + n0 + threadIdx.x] = __builtin_bit_cast(uint16_t, (__bf16)sum);
  }
}

// The same reduction with one workgroup per EIGHT output columns and token: twice as many workgroups pull the weights
// (H / 16 per token are 128 for H = 2048, half the CUs at one token).  An MFMA tile is 16 columns wide, so this form
// multiplies on the vector ALUs instead - the work is 2 flops per weight byte, nowhere near a limit.  Wave k streams the eight rows of expert ids[t, k]: eight lanes per row, each lane 16 bytes of
// every 128-byte run of the row (a wave instruction reads eight contiguous 128-byte runs), all of a row's loads in
// flight in batches of eight per lane with the next batch requested before the current one is multiplied.
template <int NW>
__global__ __launch_bounds__(64 * NW) void moe_decode_down_narrow_kernel(const uint16_t* __restrict__ hbuf,
                                                                         const int* __restrict__ ids,
                                                                         const float* __restrict__ wts,
                                                                         const uint16_t* __restrict__ down,
                                                                         uint16_t* __restrict__ out, int topk, int H,
                                                                         int Ip, int E) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  constexpr int CB = 8, U = 6;
  __shared__ float red[NW * CB];
  const int t = blockIdx.y, n0 = blockIdx.x * CB;
  const int wk = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane >> 3, kl = lane & 7;
  const int steps = (Ip + 63) / 64;                  // 64 elements of a row per step (8 lanes x 8)
  float part = 0.f;
  for (int kk = wk; kk < topk; kk += NW) {
    const int p = t * topk + kk;
    const int e = min(max(ids[p], 0), E - 1);
    const float wgt = wts[p];
    const uint16_t* wr = down + ((long long)e * H + min(n0 + r, H - 1)) * Ip;
    const uint16_t* xr = hbuf + (long long)p * Ip;
    u32x4 wa[U], wb[U], xa[U], xb[U];
    // Branch-free batches: Ip is a multiple of 32, so a 16-byte piece is in or out as a whole; a piece past the end
    // of the row reads the row's last piece instead and meets zeros.
    auto ld = [&](u32x4(&w)[U], u32x4(&x)[U], int s0) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int k = (s0 + u) * 64 + kl * 8;
        const int kc = min(k, Ip - 8);
        w[u] = __builtin_nontemporal_load((const u32x4*)(wr + kc));
        const u32x4 xv = *(const u32x4*)(xr + kc);
        x[u] = k < Ip ? xv : u32x4{0, 0, 0, 0};
      }
    };
    float acc = 0.f;
    auto mm = [&](const u32x4(&w)[U], const u32x4(&x)[U]) {
#pragma unroll
      for (int u = 0; u < U; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const unsigned wi = w[u][i], xi = x[u][i];      // (scalars first: see gemm_nt_dot_kernel)
          acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, wi), __builtin_bit_cast(bf16x2, xi), acc, false);
        }
    };
    ld(wa, xa, 0);
    for (int s0 = 0; s0 < steps; s0 += 2 * U) {
      if (s0 + U < steps) ld(wb, xb, s0 + U);
      mm(wa, xa);
      if (s0 + 2 * U < steps) ld(wa, xa, s0 + 2 * U);
      if (s0 + U < steps) mm(wb, xb);
    }
    acc += __shfl_xor(acc, 1);
    acc += __shfl_xor(acc, 2);
    acc += __shfl_xor(acc, 4);
    part += wgt * acc;
  }
  if (kl == 0) red[wk * CB + r] = part;
  __syncthreads();
  if (threadIdx.x < CB && n0 + (int)threadIdx.x < H) {
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < NW; ++q) sum += red[q * CB + threadIdx.x];
    out[(long long)t * H + n0 + threadIdx.x] = __builtin_bit_cast(uint16_t, (__bf16)sum);
  }
}

// MoE decode, first half on the vector ALUs (the counterpart of moe_decode_down_narrow_kernel): one workgroup per
// (4 NW output columns, pair); a wave streams the four gate rows and the four up rows of four columns, eight lanes
// per row, in batches of eight 16-byte pieces per lane with the next batch requested before the current one is
// multiplied; the activation row x[t] is read with the weights it meets (L1 / L2 hits).
template <int NW>
__global__ __launch_bounds__(64 * NW) void moe_decode_gate_up_dot_kernel(const uint16_t* __restrict__ x,
                                                                         const int* __restrict__ ids,
                                                                         const uint16_t* __restrict__ gate_up,
                                                                         uint16_t* __restrict__ hbuf, int topk, int Ip,
                                                                         int H, int E) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
  constexpr int U = 8;
  const int p = blockIdx.y;
  const int wk = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int r = lane >> 3, kl = lane & 7;
  const int col = blockIdx.x * (4 * NW) + wk * 4 + (r & 3);          // output column of this lane's row
  const int e = min(max(ids[p], 0), E - 1);
  const uint16_t* wr = gate_up + ((long long)e * 2 * Ip + (r >> 2) * Ip + min(col, Ip - 1)) * H;
  const uint16_t* xr = x + (long long)(p / topk) * H;
  const int steps = (H + 63) / 64;
  u32x4 wa[U], wb[U], xa[U], xb[U];
  auto ld = [&](u32x4(&w)[U], u32x4(&xv)[U], int s0) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int k = (s0 + u) * 64 + kl * 8;          // (H is a multiple of 32: a piece is in or out as a whole)
      const int kc = min(k, H - 8);
      w[u] = __builtin_nontemporal_load((const u32x4*)(wr + kc));
      const u32x4 v = *(const u32x4*)(xr + kc);
      xv[u] = k < H ? v : u32x4{0, 0, 0, 0};
    }
  };
  float acc = 0.f;
  auto mm = [&](const u32x4(&w)[U], const u32x4(&xv)[U]) {
#pragma unroll
    for (int u = 0; u < U; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned wi = w[u][i], xi = xv[u][i];      // (scalars first: see gemm_nt_dot_kernel)
        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, wi), __builtin_bit_cast(bf16x2, xi), acc, false);
      }
  };
  ld(wa, xa, 0);
  for (int s0 = 0; s0 < steps; s0 += 2 * U) {
    if (s0 + U < steps) ld(wb, xb, s0 + U);
    mm(wa, xa);
    if (s0 + 2 * U < steps) ld(wa, xa, s0 + 2 * U);
    if (s0 + U < steps) mm(wb, xb);
  }
  acc += __shfl_xor(acc, 1);
  acc += __shfl_xor(acc, 2);
  acc += __shfl_xor(acc, 4);
  // rows 0..3 hold the gate sums, rows 4..7 the up sums of the same four columns (32 lanes further on)
  const float up = __shfl_down(acc, 32);
  if (kl == 0 && r < 4 && col < Ip) {
    const float gr = (float)(__bf16)acc, ur = (float)(__bf16)up;      // both rounded to bf16 first, as the separate kernels do
    hbuf[(long long)p * Ip + col] = __builtin_bit_cast(uint16_t, (__bf16)(gr / (1.f + __expf(-gr)) * ur));
  }
}

// MoE decode, first half: hbuf[p, :] = silu(x[t] @ gate[e]^T) * (x[t] @ up[e]^T) for every (token t, expert
// e = ids[p]) pair p = t * topk + k.  gate_up [E, 2 Ip, H] (gate rows first), x [tokens, H], hbuf [pairs, Ip]; bf16.
// The decode GEMM with its fused activation, one grid row per pair.
TDIST_API int tdist_moe_decode_gate_up(const void* x, const int* ids, const void* gate_up, void* hbuf, int tokens,
                                       int topk, int Ip, int H, int E, int form, void* stream) {
  TDIST_REQUIRE(tokens > 0 && topk > 0 && (long long)tokens * topk <= 65535 && E > 0, "bad token / top-k / expert count");
  TDIST_REQUIRE(H >= 128 && H % 32 == 0 && Ip > 0 && Ip % 16 == 0, "H must be a multiple of 32 (>= 128), Ip of 16");
  TDIST_REQUIRE((((uintptr_t)x | (uintptr_t)gate_up) & 15) == 0 && (((uintptr_t)hbuf) & 3) == 0 && ids, "operands must be aligned");
  // form: 0 = the decode GEMM (MFMA) with its fused activation, one grid row per pair; anything else = the vector-ALU
  // kernel with two waves per workgroup (1 / 2 / 4 / 8 waves measured at 1 token: 12.1 / 11.7 / 13.6 / 16.3 us,
  // MFMA form 14.6; at 16 tokens 116 / 118 / 124 / 172, MFMA 137)
  if (form != 0)
    hipLaunchKernelGGL((moe_decode_gate_up_dot_kernel<2>), dim3((Ip + 7) / 8, tokens * topk), dim3(128), 0,
                       (hipStream_t)stream, (const uint16_t*)x, ids, (const uint16_t*)gate_up, (uint16_t*)hbuf, topk, Ip, H, E);
  else
    hipLaunchKernelGGL((gemm_nt_skinny_kernel<true, 2, 4, true>), dim3(Ip / 16, tokens * topk), dim3(256), 0,
                       (hipStream_t)stream, (const uint16_t*)x, (const uint16_t*)gate_up, (uint16_t*)hbuf, 1, Ip, H,
                       (long long)H, (long long)H, (long long)Ip, (const uint16_t*)nullptr, (const uint16_t*)nullptr,
                       (uint16_t*)nullptr, 0.f, ids, topk, (long long)2 * Ip * H, E);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

// out [tokens, H] = sum over the top-k experts of wts * down[e] @ hbuf[pair].  form: 1 = eight columns per workgroup on
// the vector ALUs (moe_decode_down_narrow_kernel), 0 = sixteen columns on MFMAs (moe_decode_down_kernel), -1 = the
// default.  Measured at 128 experts, top-8, H 2048, Ip 768 with cold weights, 1 / 2 / 4 / 8 / 16 tokens:
// narrow 7.0 / 12.3 / 22.3 / 40.6 / 76.7 us, wide 12.2 / 14.4 / 25.1 / 46.4 / 90.2 us - the narrow form is the default.
TDIST_API int tdist_moe_decode_down(const void* hbuf, const int* ids, const float* wts, const void* down, void* out,
                                    int tokens, int topk, int H, int Ip, int E, int form, void* stream) {
  TDIST_REQUIRE(tokens > 0 && tokens <= 65535 && topk > 0 && E > 0, "bad token / top-k / expert count");
  TDIST_REQUIRE(Ip >= 32 && Ip % 32 == 0 && H > 0, "Ip must be a multiple of 32");
  TDIST_REQUIRE((((uintptr_t)hbuf | (uintptr_t)down) & 15) == 0 && ids && wts && out, "operands must be aligned");
  if (form != 0)
    hipLaunchKernelGGL((moe_decode_down_narrow_kernel<8>), dim3((H + 7) / 8, tokens), dim3(512), 0, (hipStream_t)stream,
                       (const uint16_t*)hbuf, ids, wts, (const uint16_t*)down, (uint16_t*)out, topk, H, Ip, E);
  else
    hipLaunchKernelGGL((moe_decode_down_kernel<8>), dim3((H + 15) / 16, tokens), dim3(512), 0, (hipStream_t)stream,
                       (const uint16_t*)hbuf, ids, wts, (const uint16_t*)down, (uint16_t*)out, topk, H, Ip, E);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

// ---- MoE block for a batch of 17 .. a few hundred tokens: rows grouped by expert in blocks of sixteen ---------------
// Above sixteen tokens several tokens share an expert, so streaming an expert's weights once per (token, expert) pair
// (the two kernels above) reads them several times, while the grouped GEMM of the prefill path pads every expert to a
// 256-row tile (about 500 us per Qwen3-30B-A3B layer whatever the batch).  Here the pairs are sorted by expert into
// blocks of sixteen rows - one MFMA row block - and a workgroup streams its columns of the block's expert once for
// all sixteen rows, with the 128-byte runs and the two-MFMA step of gemm_nt_skinny128_kernel.
//
// moe_align16_kernel (one workgroup): sorted_ids[16 b + i] = pair index (token * topk + k) or -1, expert_of_block[b],
// meta[0] = number of blocks in use.  Blocks: sum over experts of ceil(count / 16) <= pairs / 16 + E.  The order of
// the rows inside an expert is whatever the atomics give; every row is computed on its own, so results do not depend
// on it.
__global__ __launch_bounds__(1024) void moe_align16_kernel(const int* __restrict__ ids, int n, int E,
                                                           int* __restrict__ sorted_ids, int* __restrict__ expert_of_block,
                                                           int* __restrict__ meta, int nb_max, int bs,
                                                           int* __restrict__ pos = nullptr) {
  // bs: rows per block; pos (optional): pos[pair] = its row in sorted order, -1 for a pair with an id out of range
  __shared__ int cnt[1024], off[1024], pos_e[1024];
  const int tid = threadIdx.x;
  cnt[tid] = 0, pos_e[tid] = 0;
  __syncthreads();
  for (int i = tid; i < n; i += 1024) {
    const int e = ids[i];
    if ((unsigned)e < (unsigned)E) atomicAdd(&cnt[e], 1);
  }
  __syncthreads();
  const int blocks = tid < E ? (cnt[tid] + bs - 1) / bs : 0;
  off[tid] = blocks;
  __syncthreads();
  for (int d = 1; d < 1024; d <<= 1) {           // inclusive scan over the experts
    const int v = tid >= d ? off[tid - d] : 0;
    __syncthreads();
    off[tid] += v;
    __syncthreads();
  }
  const int excl = off[tid] - blocks;
  const int total = off[1023];
  __syncthreads();
  off[tid] = excl;
  __syncthreads();
  if (tid == 0) meta[0] = min(total, nb_max), meta[1] = n;
  for (int j = 0; j < blocks; ++j)
    if (excl + j < nb_max) expert_of_block[excl + j] = tid;
  for (int i = tid; i < n; i += 1024) {
    const int e = ids[i];
    if ((unsigned)e < (unsigned)E) {
      const int slot = off[e] * bs + atomicAdd(&pos_e[e], 1);
      if (slot < nb_max * bs) sorted_ids[slot] = i;
      if (pos) pos[i] = slot < nb_max * bs ? slot : -1;
    } else if (pos) {
      pos[i] = -1;
    }
  }
  if (tid < E)
    for (int q = cnt[tid]; q < blocks * bs; ++q)             // padding rows of the expert's last block
      if (excl * bs + q < nb_max * bs) sorted_ids[excl * bs + q] = -1;
}

// One block of 16 MB rows x 8 GN output columns per workgroup (GN = WN, or WN / 2 with SILU: gate group g pairs with up
// group g + GN), NW waves = K slices, batches of U 64-deep steps; with MB = 2 every weight fragment feeds two row
// blocks.  x_div > 0: activation row of a pair = pair / x_div (gate_up: the token's row of X); x_div == 0: row
// 16 MB b + i of X (down: the rows the gate_up launch wrote, in sorted order).  SILU: C = bf16 [16 MB nb, N] in sorted
// order; otherwise C = fp32 [pairs, N], row = pair index, times scale[pair].
template <int WN, int NW, int U, bool SILU, int MB = 1>
__global__ __launch_bounds__(64 * NW) void moe_group16_kernel(const uint16_t* __restrict__ X, long long ldx, int x_div,
                                                              const uint16_t* __restrict__ W, long long w_estride,
                                                              long long ldw, void* __restrict__ Cv, long long ldc,
                                                              const int* __restrict__ sorted_ids,
                                                              const int* __restrict__ expert_of_block,
                                                              const int* __restrict__ meta,
                                                              const float* __restrict__ scale, int N, int K) {
  static_assert(!SILU || WN % 2 == 0, "the fused activation pairs gate group g with up group g + WN / 2");
  static_assert(U >= 2 && U % 2 == 0, "a batch is issued in two halves");
  using F = Frag<true>;
  using VT = typename F::T;
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  constexpr int GN = SILU ? WN / 2 : WN;
  constexpr int BN_ = 8 * GN;
  constexpr int BS = 16 * MB;
  const int b = blockIdx.y;
  if (b >= meta[0]) return;
  const int e = expert_of_block[b];
  const int up_off = SILU ? N : 0;
  __shared__ float red[NW * MB * WN * 64 * 4];           // [K slice][row block][column group][lane][4]
  const int wk = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 15, lg = lane >> 4;
  const int lrow = li & 7, kg = (li >> 3) * 4 + lg;      // weight row inside the group, k-group inside the step
  const int n0 = blockIdx.x * BN_;
  const int ksteps = (K + 63) / 64;
  const int per = (ksteps + NW - 1) / NW;
  const int ks0 = min(wk * per, ksteps), ks1 = min(ks0 + per, ksteps);
  bool xvalid[MB];
  const uint16_t* xr[MB];
#pragma unroll
  for (int m = 0; m < MB; ++m) {
    const int sid = sorted_ids[b * BS + m * 16 + li];
    xvalid[m] = sid >= 0;
    const long long xrow = x_div > 0 ? (long long)(xvalid[m] ? sid / x_div : 0) : (long long)b * BS + m * 16 + li;
    xr[m] = X + xrow * ldx;
  }
  const uint16_t* We = W + (long long)e * w_estride;
  const uint16_t* wr[WN];
#pragma unroll
  for (int j = 0; j < WN; ++j)                            // clamped: out-of-range columns are not stored
    wr[j] = We + (long long)((j / GN) * up_off + min(n0 + (j % GN) * 8 + lrow, N - 1)) * ldw;
  f32x4 acc1[MB][WN], acc2[MB][WN];
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc1[m][j] = acc2[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // activation fragment of K step ks, half h (k-groups 4 h .. 4 h + 3); zero past the end of K and for padding rows
  auto load_x = [&](int m, int ks, int h) -> u32x4 {
    const int k = ks * 64 + h * 32 + lg * 8;
    const u32x4 v = *(const u32x4*)(xr[m] + min(k, K - 8));
    return (xvalid[m] && k < K) ? v : u32x4{0, 0, 0, 0};
  };
  const int nfull = (ks1 - ks0) / U;
  constexpr int H = U / 2;
  u32x4 wv[U][WN], xv1[U][MB], xv2[U][MB];
  auto issue = [&](int ks, auto u0c) {
    constexpr int u0 = decltype(u0c)::value;
#pragma unroll
    for (int u = u0; u < u0 + H; ++u)
#pragma unroll
      for (int j = 0; j < WN; ++j)
        wv[u][j] = __builtin_nontemporal_load((const u32x4*)(wr[j] + min((ks + u) * 64 + kg * 8, K - 8)));
#pragma unroll
    for (int u = u0; u < u0 + H; ++u)
#pragma unroll
      for (int m = 0; m < MB; ++m) xv1[u][m] = load_x(m, ks + u, 0), xv2[u][m] = load_x(m, ks + u, 1);
  };
  auto consume = [&](auto u0c) {
    constexpr int u0 = decltype(u0c)::value;
#pragma unroll
    for (int u = u0; u < u0 + H; ++u)
#pragma unroll
      for (int j = 0; j < WN; ++j)
#pragma unroll
        for (int m = 0; m < MB; ++m) {
          acc1[m][j] = F::mfma(__builtin_bit_cast(VT, xv1[u][m]), __builtin_bit_cast(VT, wv[u][j]), acc1[m][j]);
          acc2[m][j] = F::mfma(__builtin_bit_cast(VT, xv2[u][m]), __builtin_bit_cast(VT, wv[u][j]), acc2[m][j]);
        }
  };
  using C0 = std::integral_constant<int, 0>;
  using CH = std::integral_constant<int, H>;
  if (nfull > 0) {
    issue(ks0, C0{});
    issue(ks0, CH{});
    for (int bi = 0; bi < nfull; ++bi) {
      const int ks = ks0 + bi * U;
      const bool more = bi + 1 < nfull;
      consume(C0{});
      if (more) issue(ks + U, C0{});
      consume(CH{});
      if (more) issue(ks + U, CH{});
    }
  }
  for (int ks = ks0 + nfull * U; ks < ks1; ++ks) {
    u32x4 x1[MB], x2[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) x1[m] = load_x(m, ks, 0), x2[m] = load_x(m, ks, 1);
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      const u32x4 w = __builtin_nontemporal_load((const u32x4*)(wr[j] + min(ks * 64 + kg * 8, K - 8)));
#pragma unroll
      for (int m = 0; m < MB; ++m) {
        acc1[m][j] = F::mfma(__builtin_bit_cast(VT, x1[m]), __builtin_bit_cast(VT, w), acc1[m][j]);
        acc2[m][j] = F::mfma(__builtin_bit_cast(VT, x2[m]), __builtin_bit_cast(VT, w), acc2[m][j]);
      }
    }
  }
  // accumulator: row = 4*lg + r, column = li.  Column j < 8 of the result = acc1 column j + acc2 column j + 8 (eight
  // lanes further on); wave g then sums (row block, column group) pairs g, g + NW, .. over the K slices.
#pragma unroll
  for (int m = 0; m < MB; ++m)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      float* slot = red + (((wk * MB + m) * WN + j) * 64 + lane) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) slot[r] = acc1[m][j][r] + __shfl_down(acc2[m][j][r], 8);
    }
  __syncthreads();
  for (int item = wk; item < MB * GN; item += NW) {
    const int m = item / GN, g = item - m * GN;
    f32x4 sum = {0.f, 0.f, 0.f, 0.f}, up = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < NW; ++q) {
      const float* o = red + (((q * MB + m) * WN + g) * 64 + lane) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) sum[r] += o[r];
      if constexpr (SILU) {
        const float* ou = red + (((q * MB + m) * WN + g + GN) * 64 + lane) * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) up[r] += ou[r];
      }
    }
    // lanes li and li^1 hold neighbouring columns: the even lane stores both (N, ldc are even)
    const int col = n0 + g * 8 + li;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = m * 16 + 4 * lg + r;
      if constexpr (SILU) {
        // gate and up rounded to 16 bits first, as the separate kernels do, then activated
        const float gr = (float)(__bf16)sum[r], ur = (float)(__bf16)up[r];
        const float mine = gr / (1.f + __expf(-gr)) * ur;
        const float other = __shfl_xor(mine, 1);
        if (li < 8 && col < N && (li & 1) == 0)
          *(uint32_t*)((uint16_t*)Cv + ((long long)b * BS + row) * ldc + col) = F::pack2(mine, other);
      } else {
        const int pr = sorted_ids[b * BS + row];
        const float mine = pr >= 0 ? sum[r] * scale[pr] : 0.f;
        const float other = __shfl_xor(mine, 1);
        if (pr >= 0 && li < 8 && col < N && (li & 1) == 0)
          *(float2*)((float*)Cv + (long long)pr * ldc + col) = float2{mine, other};
      }
    }
  }
}

// out[t, :] = sum over k of y[t * topk + k, :] (fp32 rows, already weighted) -> bf16.  Rows of pairs whose expert id
// is out of range were never written and are left out.
__global__ __launch_bounds__(256) void moe_topk_sum_kernel(const float* __restrict__ y, uint16_t* __restrict__ out,
                                                           const int* __restrict__ ids, int E, int topk, int H) {
  const int t = blockIdx.x;
  for (int c = threadIdx.x * 4; c < H; c += 256 * 4) {
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < topk; ++k) {
      if ((unsigned)ids[t * topk + k] >= (unsigned)E) continue;
      const float4 v = *(const float4*)(y + ((long long)t * topk + k) * H + c);
      acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
    }
    *(uint2*)(out + (long long)t * H + c) = uint2{Frag<true>::pack2(acc.x, acc.y), Frag<true>::pack2(acc.z, acc.w)};
  }
}

// The sort by expert on its own (the first launch of tdist_moe_group16_fwd) with any block height and, optionally, the
// inverse map pos[pair] -> sorted row: what the 64- / 128-row grouped form (csrc/ops/moe_group_mid.hip) starts from.
TDIST_API int tdist_moe_group_align(const int* ids, int pairs, int E, int* sorted_ids, int* expert_of_block, int* pos,
                                    int* meta, int nb_max, int block_rows, void* stream) {
  TDIST_REQUIRE(pairs > 0 && E > 0 && E <= 1024 && block_rows >= 16 && block_rows % 16 == 0, "bad pair / expert count or block height");
  TDIST_REQUIRE(nb_max >= pairs / block_rows + E, "nb_max must be at least pairs / block_rows + E");
  TDIST_REQUIRE(ids && sorted_ids && expert_of_block && meta, "null operand");
  hipLaunchKernelGGL(moe_align16_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, ids, pairs, E, sorted_ids, expert_of_block,
                     meta, nb_max, block_rows, pos);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

// out[t, :] = sum over k of x[t * topk + k, :] for bf16 rows (fp32 accumulation): the top-k reduce of the grouped-GEMM
// path on one rank (on several ranks it is part of the reduce-scatter kernel).  H a multiple of 8, 16-byte aligned rows.
__global__ __launch_bounds__(256) void moe_topk_sum_bf16_kernel(const uint16_t* __restrict__ x, uint16_t* __restrict__ out,
                                                                int topk, int H) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const int t = blockIdx.x;
  for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < topk; ++k) {
      const u32x4 v = __builtin_nontemporal_load((const u32x4*)(x + ((long long)t * topk + k) * H + c));
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[2 * i] += __builtin_bit_cast(float, v[i] << 16);
        acc[2 * i + 1] += __builtin_bit_cast(float, v[i] & 0xffff0000u);
      }
    }
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = Frag<true>::pack2(acc[2 * i], acc[2 * i + 1]);
    *(u32x4*)(out + (long long)t * H + c) = o;
  }
}

TDIST_API int tdist_moe_topk_sum_bf16(const void* x, void* out, int tokens, int topk, int H, void* stream) {
  TDIST_REQUIRE(tokens > 0 && topk > 0 && H > 0 && H % 8 == 0, "H must be a multiple of 8");
  TDIST_REQUIRE((((uintptr_t)x | (uintptr_t)out) & 15) == 0, "operands must be 16-byte aligned");
  hipLaunchKernelGGL(moe_topk_sum_bf16_kernel, dim3(tokens), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
                     (uint16_t*)out, topk, H);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

// The MoE block for `tokens` rows in four launches: sort the pairs by expert into blocks of sixteen, gate_up + SiLU per
// block, down per block (weighted, fp32, one row per pair), sum of every token's top-k rows.  x [tokens, H],
// ids / wts [tokens * topk], gate_up [E, 2 Ip, H], down [E, H, Ip], out [tokens, H]; bf16.  block_rows: 16 or 32 rows
// per block.  Work buffers: hbuf bf16 [block_rows nb_max, Ip], y fp32 [tokens * topk, H], sorted_ids int [block_rows
// nb_max], expert_of_block int [nb_max], meta int [2], with nb_max >= tokens * topk / block_rows + E.  A pair whose id is outside [0, E) contributes nothing.
TDIST_API int tdist_moe_group16_fwd(const void* x, const int* ids, const float* wts, const void* gate_up, const void* down,
                                    void* hbuf, float* y, void* out, int* sorted_ids, int* expert_of_block, int* meta,
                                    int tokens, int topk, int H, int Ip, int E, int nb_max, int block_rows, void* stream) {
  const long long pairs = (long long)tokens * topk;
  TDIST_REQUIRE(tokens > 0 && topk > 0 && E > 0 && E <= 1024 && pairs <= (1 << 20), "bad token / top-k / expert count");
  TDIST_REQUIRE(block_rows == 16 || block_rows == 32, "rows per block must be 16 or 32");
  TDIST_REQUIRE(nb_max >= pairs / block_rows + E && nb_max <= 65535,
                "nb_max must be at least pairs / block_rows + E (and fit a grid dimension)");
  TDIST_REQUIRE(H >= 128 && H % 32 == 0 && Ip >= 64 && Ip % 32 == 0, "H must be a multiple of 32 (>= 128), Ip of 32 (>= 64)");
  TDIST_REQUIRE((((uintptr_t)x | (uintptr_t)gate_up | (uintptr_t)down | (uintptr_t)hbuf | (uintptr_t)y) & 15) == 0 &&
                    (((uintptr_t)out) & 7) == 0 && ids && wts && sorted_ids && expert_of_block && meta,
                "operands must be aligned");
  hipStream_t s = (hipStream_t)stream;
  const uint16_t *xp = (const uint16_t*)x, *gu = (const uint16_t*)gate_up, *dn = (const uint16_t*)down;
  hipLaunchKernelGGL(moe_align16_kernel, dim3(1), dim3(1024), 0, s, ids, (int)pairs, E, sorted_ids, expert_of_block, meta, nb_max,
                     block_rows);
#define TDIST_G16(KERN, GRIDX, THREADS_, XP, LDX, XDIV, WP, WES, LDW, CP, LDC, SC, N_, K_)                                  \
  hipLaunchKernelGGL(KERN, dim3(GRIDX, nb_max), dim3(THREADS_), 0, s, XP, (long long)(LDX), XDIV, WP, (long long)(WES),       \
                     (long long)(LDW), CP, (long long)(LDC), (const int*)sorted_ids, (const int*)expert_of_block,           \
                     (const int*)meta, SC, N_, K_)
  // down: K = Ip.  Four K slices want at least two batches of four steps each (Ip >= 2048); shorter rows run two slices
  // in batches of two steps.  Blocks of 32 rows: two row blocks share every weight fragment (batches of two steps).
  const float* none = nullptr;
  if (block_rows == 16) {
    TDIST_G16((moe_group16_kernel<4, 4, 4, true, 1>), (Ip + 15) / 16, 256, xp, H, topk, gu, 2LL * Ip * H, H, hbuf, Ip, none, Ip, H);
    if (Ip >= 2048)
      TDIST_G16((moe_group16_kernel<4, 4, 4, false, 1>), (H + 31) / 32, 256, (const uint16_t*)hbuf, Ip, 0, dn, (long long)H * Ip, Ip,
                (void*)y, H, wts, H, Ip);
    else
      TDIST_G16((moe_group16_kernel<4, 2, 2, false, 1>), (H + 31) / 32, 128, (const uint16_t*)hbuf, Ip, 0, dn, (long long)H * Ip, Ip,
                (void*)y, H, wts, H, Ip);
  } else {
    TDIST_G16((moe_group16_kernel<4, 4, 2, true, 2>), (Ip + 15) / 16, 256, xp, H, topk, gu, 2LL * Ip * H, H, hbuf, Ip, none, Ip, H);
    if (Ip >= 2048)
      TDIST_G16((moe_group16_kernel<4, 4, 2, false, 2>), (H + 31) / 32, 256, (const uint16_t*)hbuf, Ip, 0, dn, (long long)H * Ip, Ip,
                (void*)y, H, wts, H, Ip);
    else
      TDIST_G16((moe_group16_kernel<4, 2, 2, false, 2>), (H + 31) / 32, 128, (const uint16_t*)hbuf, Ip, 0, dn, (long long)H * Ip, Ip,
                (void*)y, H, wts, H, Ip);
  }
#undef TDIST_G16
  hipLaunchKernelGGL(moe_topk_sum_kernel, dim3(tokens), dim3(256), 0, s, (const float*)y, (uint16_t*)out, ids, E, topk, H);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

// C[M, I] = silu(X W[0:I]^T) * (X W[I:2I]^T) for M <= 16 (W = gate_up weight [2 I, K], bf16 only).
TDIST_API int tdist_gemm_nt_silu_mul(const void* X, const void* W, void* C, int M, int I, int K, long long ldx,
                                     long long ldw, long long ldc, void* stream) {
  TDIST_REQUIRE(M > 0 && M <= 16 && I > 0 && I % 2 == 0, "M must be in [1, 16], I even");
  TDIST_REQUIRE(K >= 128 && K % 32 == 0, "K must be a multiple of 32 and >= 128");
  TDIST_REQUIRE(ldx % 8 == 0 && ldw % 8 == 0 && ldc % 2 == 0, "row strides must keep the vector alignment");
  TDIST_REQUIRE((((uintptr_t)X | (uintptr_t)W) & 15) == 0 && (((uintptr_t)C) & 3) == 0, "operands must be aligned");
  if (M == 1 && decode_dot_enabled()) {
    launch_dot<1, true, false>((const uint16_t*)X, (const uint16_t*)W, (uint16_t*)C, M, I, K, ldx, ldw, ldc, 0, (hipStream_t)stream);
    TDIST_HIP_TRY(hipGetLastError());
    return 0;
  }
  if (skinny128_choice(I) > 0) {
    // two gate + two up groups per workgroup from eight rows on: the activation fragments are shared by more weights
    if (M >= 8)
      hipLaunchKernelGGL((gemm_nt_skinny128_kernel<true, 4, 4, true, false>), dim3((I + 15) / 16), dim3(256), 0, (hipStream_t)stream,
                         (const uint16_t*)X, (const uint16_t*)W, (uint16_t*)C, M, I, K, ldx, ldw, ldc, (const uint16_t*)nullptr,
                         (const uint16_t*)nullptr, (uint16_t*)nullptr, 0.f);
    else
      hipLaunchKernelGGL((gemm_nt_skinny128_kernel<true, 2, 4, true, false>), dim3((I + 7) / 8), dim3(256), 0, (hipStream_t)stream,
                         (const uint16_t*)X, (const uint16_t*)W, (uint16_t*)C, M, I, K, ldx, ldw, ldc, (const uint16_t*)nullptr,
                         (const uint16_t*)nullptr, (uint16_t*)nullptr, 0.f);
    TDIST_HIP_TRY(hipGetLastError());
    return 0;
  }
  const dim3 grid((I + 15) / 16);
  hipLaunchKernelGGL((gemm_nt_skinny_kernel<true, 2, 4, true>), grid, dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)X, (const uint16_t*)W, (uint16_t*)C, M, I, K, ldx, ldw, ldc);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

// C[M, N] = A W^T + bias for M <= 16, bf16: the weight-streaming decode kernels with the bias [N] added in their
// epilogue (needs the 128-byte-run kernels: TDIST_SKINNY128=0 and TDIST_DECODE_DOT=0 are refused here).
TDIST_API int tdist_gemm_nt_decode_bias(const void* A, const void* B, void* C, const void* bias, int M, int N, int K,
                                        long long lda, long long ldb, long long ldc, void* stream) {
  TDIST_REQUIRE(M > 0 && M <= 16 && N > 0 && bias, "M must be in [1, 16], bias given");
  TDIST_REQUIRE(K >= 128 && K % 64 == 0, "K must be a multiple of 64 and >= 128");
  TDIST_REQUIRE(N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0, "N/lda/ldb/ldc must be multiples of 8");
  TDIST_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0 && (((uintptr_t)bias) & 1) == 0, "operands must be aligned");
  TDIST_REQUIRE(skinny128_choice(N, K) > 0 && decode_dot_enabled(), "the kernels with a bias epilogue are switched off");
  launch_skinny<true>(A, B, C, M, N, K, lda, ldb, ldc, (hipStream_t)stream, (const uint16_t*)bias);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

// dtype: 0 = bf16, 1 = fp16.  Requires K % 64 == 0, K >= 128, N % 8 == 0, 16-byte aligned rows.
TDIST_API int tdist_gemm_nt(const void* A, const void* B, void* C, int M, int N, int K, long long lda,
                            long long ldb, long long ldc, int dtype, int group_m, void* stream, int variant) {
  TDIST_REQUIRE(M > 0 && N > 0, "empty GEMM");
  TDIST_REQUIRE(K >= 128 && K % 64 == 0, "K must be a multiple of 64 and >= 128");
  TDIST_REQUIRE(N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0, "N/lda/ldb/ldc must be multiples of 8");
  TDIST_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0, "operands must be 16-byte aligned");
  if (M <= 16 && !((variant >> 8) & 2)) {   // (debug bit 1 of the variant's second byte forces the tile kernel)
    if (dtype == 0)
      launch_skinny<true>(A, B, C, M, N, K, lda, ldb, ldc, (hipStream_t)stream);
    else
      launch_skinny<false>(A, B, C, M, N, K, lda, ldb, ldc, (hipStream_t)stream);
    TDIST_HIP_TRY(hipGetLastError());
    return 0;
  }
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  if (group_m <= 0) group_m = 8;
  typedef void (*kern_t)(const uint16_t*, const uint16_t*, uint16_t*, int, int, int, long long, long long, long long, int, int);
  // columns: schedule variants 1 (four phases), 2 (two phases), 5..8 (two phases, staging / issue-order knobs;
  // bf16 only - see mainloop_256_2ph)
  kern_t table[2][8] = {{gemm_nt_256_kernel<true, 1>, gemm_nt_256_kernel<true, 2>, gemm_nt_256_kernel<true, 5>,
                         gemm_nt_256_kernel<true, 6>, gemm_nt_256_kernel<true, 7>, gemm_nt_256_kernel<true, 8>,
                         gemm_nt_256_kernel<true, 9>, gemm_nt_256_kernel<true, 10>},
                        {gemm_nt_256_kernel<false, 1>, gemm_nt_256_kernel<false, 2>, gemm_nt_256_kernel<false, 2>,
                         gemm_nt_256_kernel<false, 2>, gemm_nt_256_kernel<false, 2>, gemm_nt_256_kernel<false, 2>,
                         gemm_nt_256_kernel<false, 2>, gemm_nt_256_kernel<false, 2>}};
  static bool attr_set = false;
  if (!attr_set) {
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 8; ++j)
        TDIST_HIP_TRY(hipFuncSetAttribute((const void*)table[i][j], hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_set = true;
  }
  if ((variant & 255) == 13) {
    typedef void (*k4_t)(const uint16_t*, const uint16_t*, uint16_t*, int, int, int, long long, long long, long long, int);
    static const k4_t k32[2] = {gemm_nt_256_4w32_kernel<true>, gemm_nt_256_4w32_kernel<false>};
    static int num_cus32 = 0;
    if (!num_cus32) {
      int dev = 0;
      TDIST_HIP_TRY(hipGetDevice(&dev));
      TDIST_HIP_TRY(hipDeviceGetAttribute(&num_cus32, hipDeviceAttributeMultiprocessorCount, dev));
      for (int i = 0; i < 2; ++i)
        TDIST_HIP_TRY(hipFuncSetAttribute((const void*)k32[i], hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS_BYTES));
    }
    int grid = num_cus32 & ~7;
    if (grid < 8) grid = 8;
    if (tiles < grid) grid = tiles;
    hipLaunchKernelGGL(k32[dtype == 0 ? 0 : 1], dim3(grid), dim3(W4_THREADS), W4_LDS_BYTES, (hipStream_t)stream,
                       (const uint16_t*)A, (const uint16_t*)B, (uint16_t*)C, M, N, K, lda, ldb, ldc, group_m);
    TDIST_HIP_TRY(hipGetLastError());
    return 0;
  }
  if ((variant & 255) == 4 || (variant & 255) == 11 || (variant & 255) == 12) {
    // four-wave forms: 4 = LDS-DMA in groups, 11 = LDS-DMA spread between the MFMAs, 12 = register staging
    typedef void (*k4_t)(const uint16_t*, const uint16_t*, uint16_t*, int, int, int, long long, long long, long long, int);
    static const k4_t k4[2][3] = {{gemm_nt_256_4w_kernel<true, 0>, gemm_nt_256_4w_kernel<true, 1>, gemm_nt_256_4w_kernel<true, 2>},
                                  {gemm_nt_256_4w_kernel<false, 0>, gemm_nt_256_4w_kernel<false, 0>,
                                   gemm_nt_256_4w_kernel<false, 2>}};
    static int num_cus4 = 0;
    if (!num_cus4) {
      int dev = 0;
      TDIST_HIP_TRY(hipGetDevice(&dev));
      TDIST_HIP_TRY(hipDeviceGetAttribute(&num_cus4, hipDeviceAttributeMultiprocessorCount, dev));
      for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 3; ++j)
          TDIST_HIP_TRY(hipFuncSetAttribute((const void*)k4[i][j], hipFuncAttributeMaxDynamicSharedMemorySize, W4_LDS_BYTES));
    }
    int grid = num_cus4 & ~7;
    if (grid < 8) grid = 8;
    if (tiles < grid) grid = tiles;
    const int mode = (variant & 255) == 4 ? 0 : ((variant & 255) == 11 ? 1 : 2);
    hipLaunchKernelGGL(k4[dtype == 0 ? 0 : 1][mode], dim3(grid), dim3(W4_THREADS), W4_LDS_BYTES, (hipStream_t)stream,
                       (const uint16_t*)A, (const uint16_t*)B, (uint16_t*)C, M, N, K, lda, ldb, ldc, group_m);
    TDIST_HIP_TRY(hipGetLastError());
    return 0;
  }
  if ((variant & 255) == 3) {
    typedef void (*pkern_t)(const uint16_t*, const uint16_t*, uint16_t*, int, int, int, long long, long long, long long, int,
                            tdist::WgTrace);
    // bits 16..17 of `variant` pick the schedule knobs of mainloop_256_2ph (fp16: defaults only)
    static const pkern_t ptable[5] = {
        gemm_nt_256_persistent_kernel<true, 0>, gemm_nt_256_persistent_kernel<true, 1>,
        gemm_nt_256_persistent_kernel<true, 2>, gemm_nt_256_persistent_kernel<true, 3>,
        gemm_nt_256_persistent_kernel<false, 3>};
    static int num_cus = 0;
    static bool pattr_set = false;
    if (!pattr_set) {
      int dev = 0;
      TDIST_HIP_TRY(hipGetDevice(&dev));
      TDIST_HIP_TRY(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev));
      for (int i = 0; i < 5; ++i)
        TDIST_HIP_TRY(hipFuncSetAttribute((const void*)ptable[i], hipFuncAttributeMaxDynamicSharedMemorySize, 2 * BUF_BYTES));
      pattr_set = true;
    }
    int grid = num_cus & ~7;                      // multiple of 8: a workgroup's tiles stay on its XCD
    if (grid < 8) grid = 8;
    if (tiles < grid) grid = tiles;               // (fewer tiles than CUs: one tile each, no remap constraint)
    const pkern_t pk = dtype == 0 ? ptable[(variant >> 16) & 3] : ptable[4];
    hipLaunchKernelGGL(pk, dim3(grid), dim3(THREADS), 2 * BUF_BYTES, (hipStream_t)stream, (const uint16_t*)A,
                       (const uint16_t*)B, (uint16_t*)C, M, N, K, lda, ldb, ldc, group_m, tdist::wg_trace_sink());
    TDIST_HIP_TRY(hipGetLastError());
    return 0;
  }
  const int v = variant & 255;
  kern_t k = table[dtype == 0 ? 0 : 1][v == 2 ? 1 : (v >= 5 && v <= 10 ? v - 3 : 0)];
  hipLaunchKernelGGL(k, dim3(tiles), dim3(THREADS), LDS_BYTES, (hipStream_t)stream, (const uint16_t*)A,
                     (const uint16_t*)B, (uint16_t*)C, M, N, K, lda, ldb, ldc, group_m, variant >> 8);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}


static int tile_family_cus() {
  static int num_cus = 0;
  if (!num_cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      num_cus = 256;
  }
  return num_cus;
}

// The plan tdist_gemm_nt_ex will follow for this problem (K in elements of the input type): tile, K split and
// the workspace it needs (partial_bytes of fp32 partial tiles, n_counters zero-initialised ints) when it splits.
// bm / bn / split_k: 0 = free, otherwise forced.
TDIST_API int tdist_gemm_nt_ex_plan(int M, int N, int K, int dtype, int bm, int bn, int split_k, int* out_bm,
                                    int* out_bn, int* out_split_k, long long* partial_bytes, int* n_counters,
                                    int* out_n_full) {
  TDIST_REQUIRE(dtype >= 0 && dtype <= 4 && M > 0 && N > 0 && K > 0, "bad problem");
  const int per16 = dtype >= 2 ? 2 : 1;
  const TilePlan pl = plan_tiles(M, N, K / per16, tile_family_cus(), bm, bn, split_k, true);
  *out_bm = 32 * pl.mt;
  *out_bn = 64 * pl.nt;
  *out_split_k = pl.split_k;
  *partial_bytes = pl.partial_bytes;
  *n_counters = pl.n_counters;
  *out_n_full = pl.n_full;
  return 0;
}

// dtype: 0 = bf16, 1 = fp16, 2 = fp8 e4m3 (OCP), 3 = fp8 e5m2, 4 = int8.  out_kind: 0 = 16-bit output (the input type
// for 16-bit inputs, bf16 for 8-bit inputs), 1 = int32 (int8 inputs only).  K, lda, ldb in ELEMENTS of the input
// type; ldc in elements of the output type.  scale_a [M] / scale_b [N]: optional fp32 scales of the 16-bit output.
// bm / bn: 128 or 256, split_k 1..4; 0 = chosen by the cost model (a split is only chosen when `partial` /
// `counters` are large enough for it: see tdist_gemm_nt_ex_plan).
// bias [N]: optional addend of the 16-bit output's type (out_kind 0).
TDIST_API int tdist_gemm_nt_ex_bias(const void* A, const void* B, void* C, int M, int N, int K, long long lda, long long ldb,
                                    long long ldc, int dtype, int out_kind, const float* scale_a, const float* scale_b,
                                    const void* bias, int bm, int bn, int split_k, void* partial, long long partial_bytes,
                                    int* counters, int n_counters, int group_m, void* stream);

TDIST_API int tdist_gemm_nt_ex(const void* A, const void* B, void* C, int M, int N, int K, long long lda, long long ldb,
                               long long ldc, int dtype, int out_kind, const float* scale_a, const float* scale_b, int bm,
                               int bn, int split_k, void* partial, long long partial_bytes, int* counters,
                               int n_counters, int group_m, void* stream) {
  return tdist_gemm_nt_ex_bias(A, B, C, M, N, K, lda, ldb, ldc, dtype, out_kind, scale_a, scale_b, nullptr, bm, bn, split_k,
                               partial, partial_bytes, counters, n_counters, group_m, stream);
}

TDIST_API int tdist_gemm_nt_ex_bias(const void* A, const void* B, void* C, int M, int N, int K, long long lda, long long ldb,
                                    long long ldc, int dtype, int out_kind, const float* scale_a, const float* scale_b,
                                    const void* bias, int bm, int bn, int split_k, void* partial, long long partial_bytes,
                                    int* counters, int n_counters, int group_m, void* stream) {
  TDIST_REQUIRE(M > 0 && N > 0, "empty GEMM");
  TDIST_REQUIRE(!bias || (out_kind == 0 && ((uintptr_t)bias & 1) == 0), "a bias needs the 16-bit output");
  TDIST_REQUIRE(dtype >= 0 && dtype <= 4, "dtype must be 0..4");
  const int eb = dtype >= 2 ? 1 : 2;                       // bytes per input element
  const int per16 = 2 / eb;
  TDIST_REQUIRE(K % (64 * per16) == 0 && K >= 128 * per16, "K must be a multiple of 64 (16-bit) / 128 (8-bit) and two steps deep");
  TDIST_REQUIRE(lda % (8 * per16) == 0 && ldb % (8 * per16) == 0, "operand rows must be 16-byte aligned");
  TDIST_REQUIRE(out_kind == 0 || (out_kind == 1 && dtype == 4), "int32 output needs int8 inputs");
  TDIST_REQUIRE(N % 8 == 0 && ldc % (out_kind == 1 ? 4 : 8) == 0, "N must be a multiple of 8, output rows 16-byte aligned");
  TDIST_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0, "operands must be 16-byte aligned");
  TDIST_REQUIRE((bm == 0 || bm == 128 || bm == 256) && (bn == 0 || bn == 128 || bn == 256), "bm / bn must be 0, 128 or 256");
  TDIST_REQUIRE(split_k >= 0 && split_k <= 4, "split_k must be 0..4");
  const int num_cus = tile_family_cus();
  const int K16 = K / per16;
  TilePlan pk = plan_tiles(M, N, K16, num_cus, bm, bn, split_k, true);
  if (pk.split_k > 1 && (!partial || !counters || partial_bytes < pk.partial_bytes || n_counters < pk.n_counters)) {
    TDIST_REQUIRE(split_k <= 1, "split-K workspace too small");
    pk = plan_tiles(M, N, K16, num_cus, bm, bn, 1, true);
  }
  if (group_m <= 0) group_m = 8;
  typedef void (*kern_t)(const uint16_t*, const uint16_t*, void*, int, int, int, long long, long long, long long, int,
                         const float*, const float*, int, float*, int*, int, const uint16_t*);
  // [dtype][shape: 84, 82, 44, 42]
  static const kern_t table[5][4] = {
      {gemm_nt_tile_kernel<DT_BF16, 8, 4, false>, gemm_nt_tile_kernel<DT_BF16, 8, 2, false>,
       gemm_nt_tile_kernel<DT_BF16, 4, 4, false>, gemm_nt_tile_kernel<DT_BF16, 4, 2, false>},
      {gemm_nt_tile_kernel<DT_F16, 8, 4, false>, gemm_nt_tile_kernel<DT_F16, 8, 2, false>,
       gemm_nt_tile_kernel<DT_F16, 4, 4, false>, gemm_nt_tile_kernel<DT_F16, 4, 2, false>},
      {gemm_nt_tile_kernel<DT_FP8, 8, 4, false>, gemm_nt_tile_kernel<DT_FP8, 8, 2, false>,
       gemm_nt_tile_kernel<DT_FP8, 4, 4, false>, gemm_nt_tile_kernel<DT_FP8, 4, 2, false>},
      {gemm_nt_tile_kernel<DT_BF8, 8, 4, false>, gemm_nt_tile_kernel<DT_BF8, 8, 2, false>,
       gemm_nt_tile_kernel<DT_BF8, 4, 4, false>, gemm_nt_tile_kernel<DT_BF8, 4, 2, false>},
      {gemm_nt_tile_kernel<DT_I8, 8, 4, false>, gemm_nt_tile_kernel<DT_I8, 8, 2, false>,
       gemm_nt_tile_kernel<DT_I8, 4, 4, false>, gemm_nt_tile_kernel<DT_I8, 4, 2, false>}};
  static const kern_t table32[4] = {gemm_nt_tile_kernel<DT_I8, 8, 4, true>, gemm_nt_tile_kernel<DT_I8, 8, 2, true>,
                                    gemm_nt_tile_kernel<DT_I8, 4, 4, true>, gemm_nt_tile_kernel<DT_I8, 4, 2, true>};
  // 16-bit output with scales (8-bit inputs; the 16-bit input types take scales too, through the same instances)
  static const kern_t table_sc[5][4] = {
      {gemm_nt_tile_kernel<DT_BF16, 8, 4, false, true>, gemm_nt_tile_kernel<DT_BF16, 8, 2, false, true>,
       gemm_nt_tile_kernel<DT_BF16, 4, 4, false, true>, gemm_nt_tile_kernel<DT_BF16, 4, 2, false, true>},
      {gemm_nt_tile_kernel<DT_F16, 8, 4, false, true>, gemm_nt_tile_kernel<DT_F16, 8, 2, false, true>,
       gemm_nt_tile_kernel<DT_F16, 4, 4, false, true>, gemm_nt_tile_kernel<DT_F16, 4, 2, false, true>},
      {gemm_nt_tile_kernel<DT_FP8, 8, 4, false, true>, gemm_nt_tile_kernel<DT_FP8, 8, 2, false, true>,
       gemm_nt_tile_kernel<DT_FP8, 4, 4, false, true>, gemm_nt_tile_kernel<DT_FP8, 4, 2, false, true>},
      {gemm_nt_tile_kernel<DT_BF8, 8, 4, false, true>, gemm_nt_tile_kernel<DT_BF8, 8, 2, false, true>,
       gemm_nt_tile_kernel<DT_BF8, 4, 4, false, true>, gemm_nt_tile_kernel<DT_BF8, 4, 2, false, true>},
      {gemm_nt_tile_kernel<DT_I8, 8, 4, false, true>, gemm_nt_tile_kernel<DT_I8, 8, 2, false, true>,
       gemm_nt_tile_kernel<DT_I8, 4, 4, false, true>, gemm_nt_tile_kernel<DT_I8, 4, 2, false, true>}};
  static const int lds_of[4] = {TileShape<8, 4>::LDS, TileShape<8, 2>::LDS, TileShape<4, 4>::LDS, TileShape<4, 2>::LDS};
  static bool attr_set = false;
  if (!attr_set) {
    for (int s = 0; s < 4; ++s) {
      for (int d = 0; d < 5; ++d)
      {
        TDIST_HIP_TRY(hipFuncSetAttribute((const void*)table[d][s], hipFuncAttributeMaxDynamicSharedMemorySize, lds_of[s] + 16));
        TDIST_HIP_TRY(hipFuncSetAttribute((const void*)table_sc[d][s], hipFuncAttributeMaxDynamicSharedMemorySize, lds_of[s] + 16));
      }
      TDIST_HIP_TRY(hipFuncSetAttribute((const void*)table32[s], hipFuncAttributeMaxDynamicSharedMemorySize, lds_of[s] + 16));
    }
    attr_set = true;
  }
  const int si = pk.mt == 8 ? (pk.nt == 4 ? 0 : 1) : (pk.nt == 4 ? 2 : 3);
  const int bmv = 32 * pk.mt, bnv = 64 * pk.nt;
  const long long tiles_all = (long long)((M + bmv - 1) / bmv) * ((N + bnv - 1) / bnv);
  const long long items = pk.n_full + (tiles_all - pk.n_full) * pk.split_k;
  const int per_cu = si == 3 ? 2 : 1;
  long long grid = (long long)(num_cus & ~7) * per_cu;
  if (grid < 8) grid = 8;
  if (items < grid) grid = items;
  const kern_t k = out_kind == 1 ? table32[si] : ((scale_a || scale_b || bias) ? table_sc[dtype][si] : table[dtype][si]);
  hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(THREADS), lds_of[si] + 16, (hipStream_t)stream, (const uint16_t*)A,
                     (const uint16_t*)B, C, M, N, K16, lda / per16, ldb / per16, ldc, group_m, scale_a, scale_b,
                     pk.split_k, (float*)partial, counters, pk.split_k > 1 ? pk.n_full : (int)tiles_all,
                     (const uint16_t*)bias);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
