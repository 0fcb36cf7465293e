// Mid-M GEMM (17 <= M <= a few hundred rows against a large weight):  C[M, N] = X[M, K] * W[N, K]^T.
//
// This is the band between the decode kernel (M <= 16, gemm.hip) and shapes that fill the chip with
// 256x256 tiles: prefill of short prompts, batched decode, the row-parallel projections of a TP layer at
// small batch.  The weights dominate the traffic (every element is used M <= 128 times, the machine
// balance is ~300 FLOP per byte), so the kernel is shaped to stream them: a workgroup = 4 waves side by
// side along N, each 16*WN columns wide and all 16*MT rows deep (MT = 2 / 4 / 8), walks its K range in
// 64-deep stages through a four-slot LDS ring that direct-to-LDS loads keep three stages ahead; a slot holds
// the X slab of the stage (16*MT rows, staged once per workgroup, read by all four waves) and every wave's
// own strip of W, both in the tile engine's swizzled [16 rows][32 k] sub-tile format (gemm_gfx950.hpp).
// Nothing is loaded into registers from global memory inside the loop: every vector-memory operation is an
// LDS-DMA piece, so the waits are the kernel's own counted `s_waitcnt vmcnt(2 groups)` - with register
// loads in the loop the compiler places its own waits, and across the loop's back edge those drain the
// pipeline (vmcnt(0) twice per four stages, measured 1.3 us per stage instead of 0.2).
//
// The grid is (column blocks x K splits, row passes of 16*MT).  K is cut across workgroups (split-K) until
// the chip is full: every split parks its fp32 accumulators (register layout, coalesced) and counts itself,
// the last one to arrive adds the others to its live accumulators and stores - nobody waits, and the counter
// is re-armed by that last workgroup, so a launch is a single kernel.
//
// One barrier per stage: iteration s waits for this wave's pieces of stage s (everything but the two
// younger groups), then all waves meet - every piece of stage s is in LDS, and everybody is done reading
// stage s - 1, whose slot the pieces of stage s + 3 issued right after the barrier overwrite.
//
// Role in the reference: the autotuned tile space of kernels/amd/all_gather_gemm.py:102-124 (BM 128/256 x
// BN 128/256 x BK 64/128) and the small-M configurations of layers/nvidia/gemm_allreduce_layer.py:76-93.
#include <stdlib.h>

#include "tdist/atomics.hpp"
#include "tdist/cache_policy.hpp"
#include "tdist/common.hpp"
#include "tdist/gemm_gfx950.hpp"

using namespace tdist;
using namespace tdist::gemm;

struct MidParams {
  const uint16_t* X;
  const uint16_t* W;
  uint16_t* C;
  float* partial;     // [row pass][column block][split][4 waves][MT][WN][64 lanes] f32x4 (splits > 1)
  int* counters;      // [row pass][column block] arrival counters, zero at launch, left zero
  long long ldx, ldw, ldc;
  int M, N, K;
  int splits;
};

template <bool IS_BF16, int MT, int WN>
__global__ __launch_bounds__(256, 2) void gemm_nt_mid_kernel(MidParams p) {
  using F = Frag<IS_BF16>;
  using V = typename F::T;
  constexpr int NS = 4;                          // ring slots; loads run NS - 1 stages ahead
  constexpr int X_BYTES = 2 * MT * 1024;         // 16*MT rows x 64 k x 2 bytes, as 2*MT sub-tiles of [16][32]
  constexpr int WS_BYTES = 2 * WN * 1024;        // one wave's strip of W: 16*WN rows x 64 k
  constexpr int STAGE_BYTES = X_BYTES + 4 * WS_BYTES;
  constexpr int NDX = MT / 2;                    // X sub-tiles staged per wave and stage
  constexpr int ND = NDX + 2 * WN;               // LDS-DMA pieces per wave and stage
  constexpr int BNW = 16 * WN, BN_ = 4 * BNW;
  static_assert(MT == 2 || MT == 4 || MT == 8, "X pieces must divide evenly over the four waves");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_arrived;
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63, li = l & 15, lg = l >> 4;
  const int ncb = (p.N + BN_ - 1) / BN_;
  const int cb = blockIdx.x % ncb, split = blockIdx.x / ncb, S = p.splits;
  const int m0 = blockIdx.y * 16 * MT;
  const int nst_total = p.K / 64;
  const int st0 = (int)((long long)nst_total * split / S), st1 = (int)((long long)nst_total * (split + 1) / S);
  const int nst = st1 - st0;

  // Staging: a piece is one sub-tile [16 rows][32 k]; lane l moves row l >> 2, 16-byte chunk (l & 3) with the
  // image's XOR applied to the source.  Rows past the end are clamped (their results are masked at the store).
  const int kchunk = (l & 3) ^ (((l >> 5) & 1) << 1);
  const uint16_t* xsrc[NDX];
  int xdst[NDX];
#pragma unroll
  for (int i = 0; i < NDX; ++i) {
    const int q = w + 4 * i;                      // sub-tile (m tile q >> 1, k half q & 1)
    const int row = min(m0 + (q >> 1) * 16 + (l >> 2), p.M - 1);
    xsrc[i] = p.X + (long long)row * p.ldx + (long long)st0 * 64 + (q & 1) * 32 + kchunk * 8;
    xdst[i] = q * 1024;
  }
  const uint16_t* wsrc[WN];                       // this wave's strip: column group j, both k halves
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int row = min(cb * BN_ + w * BNW + j * 16 + (l >> 2), p.N - 1);
    wsrc[j] = p.W + (long long)row * p.ldw + (long long)st0 * 64 + kchunk * 8;
  }
  const int w_area = X_BYTES + w * WS_BYTES;
  const int lane_off = ((li * 64) + (lg * 16)) ^ (((li >> 3) & 1) << 5);

  f32x4 acc[MT][WN];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < WN; ++j) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto issue = [&](int st, int slot) {            // ND pieces, weights last
    char* base = smem + slot * STAGE_BYTES;
#pragma unroll
    for (int i = 0; i < NDX; ++i)
      __builtin_amdgcn_global_load_lds(TDIST_GLB_PTR(xsrc[i] + (long long)st * 64), TDIST_LDS_PTR(base + xdst[i]), 16,
                                       0, 0);
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      __builtin_amdgcn_global_load_lds(TDIST_GLB_PTR(wsrc[j] + (long long)st * 64),
                                       TDIST_LDS_PTR(base + w_area + (j * 2) * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(TDIST_GLB_PTR(wsrc[j] + (long long)st * 64 + 32),
                                       TDIST_LDS_PTR(base + w_area + (j * 2 + 1) * 1024), 16, 0, 0);
    }
  };
  // `ahead`: groups younger than stage st that are in flight when it is waited for (2 in steady state)
  auto stage = [&](int st, auto u_c, auto ahead_c, bool issue_next) {
    constexpr int u = decltype(u_c)::value;
    constexpr int ahead = decltype(ahead_c)::value;
    if constexpr (ahead == 2) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * ND) : "memory");
    } else if constexpr (ahead == 1) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (issue_next) issue(st + NS - 1, (u + NS - 1) % NS);
    __builtin_amdgcn_sched_barrier(0);
    const char* xs = smem + u * STAGE_BYTES + lane_off;
    const char* ws = xs + w_area;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      V xf[MT], wf[WN];
#pragma unroll
      for (int j = 0; j < WN; ++j) wf[j] = *(const V*)(ws + (j * 2 + ks) * 1024);
#pragma unroll
      for (int m = 0; m < MT; ++m) xf[m] = *(const V*)(xs + (m * 2 + ks) * 1024);
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int j = 0; j < WN; ++j) acc[m][j] = F::mfma(wf[j], xf[m], acc[m][j]);
    }
  };
  using U0 = std::integral_constant<int, 0>;
  using U1 = std::integral_constant<int, 1>;
  using U2 = std::integral_constant<int, 2>;
  using U3 = std::integral_constant<int, 3>;
  static_assert(NS == 4, "the stage schedule below is written out for four slots");
  // prologue: up to three groups in flight
  if (nst > 0) issue(0, 0);
  if (nst > 1) issue(1, 1);
  if (nst > 2) issue(2, 2);
  int base = 0;
  for (; base + 2 * NS - 2 < nst; base += NS) {     // steady state: every stage of the group issues stage + 3
    stage(base, U0{}, U2{}, true);
    stage(base + 1, U1{}, U2{}, true);
    stage(base + 2, U2{}, U2{}, true);
    stage(base + 3, U3{}, U2{}, true);
  }
  // tail (at most 2 * NS - 2 stages): the number of younger groups is no longer constant - drain instead
  for (; base < nst; base += NS) {
    stage(base, U0{}, U0{}, base + 3 < nst);
    if (base + 1 < nst) stage(base + 1, U1{}, U0{}, base + 4 < nst);
    if (base + 2 < nst) stage(base + 2, U2{}, U0{}, base + 5 < nst);
    if (base + 3 < nst) stage(base + 3, U3{}, U0{}, base + 6 < nst);
  }

  // accumulator: lane (li, lg) holds row m*16 + li, columns j*16 + 4 lg .. + 4 of the wave's strip
  const int col0 = cb * BN_ + w * BNW + 4 * lg;
  auto store_all = [&]() {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const int row = m0 + m * 16 + li;
#pragma unroll
      for (int j = 0; j < WN; ++j) {
        const int col = col0 + j * 16;
        if (row < p.M && col < p.N) {
          uint2 v;
          v.x = F::pack2(acc[m][j][0], acc[m][j][1]);
          v.y = F::pack2(acc[m][j][2], acc[m][j][3]);
          *(uint2*)(p.C + (long long)row * p.ldc + col) = v;
        }
      }
    }
  };
  if (S == 1) {
    store_all();
    return;
  }
  const int tile = blockIdx.y * ncb + cb;
  constexpr int PER_SPLIT = 4 * MT * WN * 64;                 // f32x4 per split of one tile
  f32x4* tbase = (f32x4*)p.partial + (long long)tile * S * PER_SPLIT;
  {
    f32x4* mine = tbase + (long long)split * PER_SPLIT + (w * MT * WN) * 64 + l;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < WN; ++j) st_wt128_agent(mine + (m * WN + j) * 64, acc[m][j]);
  }
  drain_stores();
  __syncthreads();
  if (tid == 0) s_arrived = atom_add_relaxed_agent(p.counters + tile, 1);
  __syncthreads();
  if (s_arrived != S - 1) return;
  if (tid < 64) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (tid == 0) store_relaxed_agent(p.counters + tile, 0);      // re-armed for the next launch
  for (int q = 0; q < S; ++q) {
    if (q == split) continue;
    const f32x4* src = tbase + (long long)q * PER_SPLIT + (w * MT * WN) * 64 + l;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < WN; ++j) acc[m][j] += src[(m * WN + j) * 64];
  }
  store_all();
}

namespace {

template <bool IS_BF16, int MT, int WN>
int launch_mid(const MidParams& p, int passes, hipStream_t s) {
  constexpr int LDS = 4 * (2 * MT + 8 * WN) * 1024;
  const int ncb = (p.N + 64 * WN - 1) / (64 * WN);
  static bool attr_set = false;
  if (!attr_set) {
    // (up to 128 KiB of ring: above the default dynamic limit)
    TDIST_HIP_TRY(hipFuncSetAttribute((const void*)gemm_nt_mid_kernel<IS_BF16, MT, WN>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_nt_mid_kernel<IS_BF16, MT, WN>), dim3(ncb * p.splits, passes), dim3(256), LDS, s, p);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

template <bool IS_BF16>
int dispatch_mid(const MidParams& p, int mt, int wn, int passes, hipStream_t s) {
  if (mt == 2) return wn == 1 ? launch_mid<IS_BF16, 2, 1>(p, passes, s) : launch_mid<IS_BF16, 2, 2>(p, passes, s);
  if (mt == 4) return wn == 1 ? launch_mid<IS_BF16, 4, 1>(p, passes, s) : launch_mid<IS_BF16, 4, 2>(p, passes, s);
  return wn == 1 ? launch_mid<IS_BF16, 8, 1>(p, passes, s) : launch_mid<IS_BF16, 8, 2>(p, passes, s);
}

}  // namespace

// Plan for a shape: rows per pass (16 * mt), strip width per wave (16 * wn), K splits; returns the number
// of fp32 bytes of split-K workspace and the number of arrival counters the launch needs.
TDIST_API int tdist_gemm_nt_mid_plan(int M, int N, int K, int num_cus, int force_wn, int force_splits, int* mt_out,
                                     int* wn_out, int* splits_out, long long* partial_bytes, int* n_counters) {
  TDIST_REQUIRE(M > 0 && N > 0 && K >= 128 && K % 64 == 0, "bad shape");
  const int mt = M <= 32 ? 2 : (M <= 64 ? 4 : 8);
  const int passes = (M + 16 * mt - 1) / (16 * mt);
  const int nst = K / 64;
  // Measured on MI355X (scripts/dev/probe_mid_gemm.py scan, N 4096-6400, K 1024-5120, M 32 / 128): 64-column
  // workgroups beat 128-column ones everywhere (the wide form needs 80-128 KiB of LDS ring), and the best split
  // count is the one that gives about ONE workgroup per CU - more splits only add partial-sum traffic, and
  // at 16*MT = 128 rows the ring (96 KiB) admits one workgroup per CU anyway, so a grid beyond the CU count
  // runs in two rounds.  Wide strips win once the narrow form no longer fits one round (N >= 16384:
  // 43.9 vs 54.7 us at 24576 x 4096, M = 128).
  int wn = ((N + 63) / 64) * passes > num_cus ? 2 : 1;
  if (force_wn == 1 || force_wn == 2) wn = force_wn;
  const int ncb = (N + 64 * wn - 1) / (64 * wn);
  int splits = num_cus / (ncb * passes);
  if (splits > nst / 4) splits = nst / 4;
  if (splits > 16) splits = 16;
  if (splits < 1) splits = 1;
  if (force_splits > 0) splits = force_splits < nst / 2 ? force_splits : (nst / 2 > 0 ? nst / 2 : 1);
  *mt_out = mt;
  *wn_out = wn;
  *splits_out = splits;
  *partial_bytes = splits > 1 ? (long long)passes * ncb * splits * 4 * mt * wn * 64 * 16 : 0;
  *n_counters = passes * ncb;
  return 0;
}

// dtype: 0 = bf16, 1 = fp16.  `partial` / `counters` as sized by tdist_gemm_nt_mid_plan for the same
// (mt, wn, splits); the counters must be zero before the first launch (every launch leaves them zero).
TDIST_API int tdist_gemm_nt_mid(const void* X, const void* W, void* C, int M, int N, int K, long long ldx, long long ldw,
                                long long ldc, int dtype, int mt, int wn, int splits, void* partial, int* counters,
                                void* stream) {
  TDIST_REQUIRE(M > 0 && N > 0 && K >= 128 && K % 64 == 0, "K must be a multiple of 64 and >= 128");
  TDIST_REQUIRE(N % 8 == 0 && ldx % 8 == 0 && ldw % 8 == 0 && ldc % 4 == 0, "N/ldx/ldw must be multiples of 8, ldc of 4");
  TDIST_REQUIRE((((uintptr_t)X | (uintptr_t)W) & 15) == 0 && (((uintptr_t)C) & 7) == 0, "operands must be aligned");
  TDIST_REQUIRE((mt == 2 || mt == 4 || mt == 8) && (wn == 1 || wn == 2), "bad tile plan");
  TDIST_REQUIRE(splits >= 1 && splits <= K / 128 && (splits == 1 || (partial && counters)), "bad split plan");
  MidParams p;
  p.X = (const uint16_t*)X;
  p.W = (const uint16_t*)W;
  p.C = (uint16_t*)C;
  p.partial = (float*)partial;
  p.counters = counters;
  p.ldx = ldx;
  p.ldw = ldw;
  p.ldc = ldc;
  p.M = M;
  p.N = N;
  p.K = K;
  p.splits = splits;
  const int passes = (M + 16 * mt - 1) / (16 * mt);
  return dtype == 0 ? dispatch_mid<true>(p, mt, wn, passes, (hipStream_t)stream)
                    : dispatch_mid<false>(p, mt, wn, passes, (hipStream_t)stream);
}
