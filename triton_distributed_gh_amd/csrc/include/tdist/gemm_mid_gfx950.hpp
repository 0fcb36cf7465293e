// Mid-M tile body shared by the plain mid-M GEMM (ops/gemm_mid.hip) and the low-latency GEMM + AllReduce
// (ops/gemm_ar.hip): main loop, accumulator layout helpers and the wait-free split-K join.  The design is
// described at the top of ops/gemm_mid.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <type_traits>

#include "tdist/atomics.hpp"
#include "tdist/cache_policy.hpp"
#include "tdist/gemm_gfx950.hpp"

namespace tdist {
namespace gemm {

constexpr int MID_THREADS = 256;
constexpr int mid_lds_bytes(int mt, int wn) { return 4 * (2 * mt + 8 * wn) * 1024; }

// A planned launch: rows per pass = 16 * mt, columns per workgroup = 64 * wn.
struct MidPlan {
  int mt, wn, splits, passes, ncb;
  long long partial_bytes;     // fp32 split-K workspace (0 when splits == 1)
  int n_counters;              // arrival counters (passes * ncb)
};

// `num_cus`: workgroups that may run at once (the CU count, or a rank's share of it).
inline MidPlan mid_plan(int M, int N, int K, int num_cus, int force_wn, int force_splits) {
  MidPlan pl;
  pl.mt = M <= 32 ? 2 : (M <= 64 ? 4 : 8);
  pl.passes = (M + 16 * pl.mt - 1) / (16 * pl.mt);
  const int nst = K / 64;
  // Measured on MI355X (scripts/dev/probe_mid_gemm.py scan, N 4096-6400, K 1024-5120, M 32 / 128): 64-column
  // workgroups beat 128-column ones everywhere (the wide form needs 80-128 KiB of LDS ring), and the best split
  // count is the one that gives about ONE workgroup per CU - more splits only add partial-sum traffic, and
  // at 16*MT = 128 rows the ring (96 KiB) admits one workgroup per CU anyway, so a grid beyond the CU count
  // runs in two rounds.  Wide strips win once the narrow form no longer fits one round (N >= 16384:
  // 43.9 vs 54.7 us at 24576 x 4096, M = 128).
  // Two refinements from cold-weight sweeps of LM-head sized layers (scripts/dev/sweep_linear_kernels.py, MID_FORMS=1):
  //  * up to 64 rows two narrow workgroups share a CU (64 KiB of ring each), so a narrow grid of exactly two per CU
  //    keeps twice the bytes in flight of one wide workgroup per CU: 32768 x 4096 at 32 / 64 rows 55 / 61 us narrow
  //    against 66 / 66 us wide (a grid of one and a half per CU does not gain: 24576 x 4096 48 / 54 against 48 / 50);
  //  * at 65 - 128 rows a narrow workgroup re-reads two bytes of activations per byte of weights, so when the narrow
  //    grid would fill more than half of the CUs without any K split, the wide form with K cut in two (same number
  //    of workgroups, half the activation traffic) is ahead: 12800 x 8192 at 128 rows 65 us against 72.
  const int narrow = ((N + 63) / 64) * pl.passes;
  pl.wn = narrow > num_cus ? 2 : 1;
  if (pl.mt <= 4 && narrow == 2 * num_cus) pl.wn = 1;
  if (pl.mt == 8 && narrow <= num_cus && 2 * narrow > num_cus && nst >= 8) pl.wn = 2;
  // Two more from timing every form on 74 weight shapes with cold weights (scripts/tune_gemm_table.py):
  //  * when the narrow grid with its K split fills less than 80 % of the CUs and the wide one a tenth of them more
  //    (9216 columns at 64 rows: 144 workgroups narrow, 72 x 3 wide), the wide form is 1.1 - 1.3 x faster;
  //  * at 128-row passes a narrow workgroup that is left with 2048 or more of K after the split re-reads so much of
  //    the activations that the wide form with twice the splits wins (5120 x 10240 .. 27648 at 128 rows 1.11 - 1.22 x,
  //    4096 x 14336 at 256 rows 1.08 x, 2304 x 16384 at 256 rows 1.66 x).
  if (pl.wn == 1 && narrow <= num_cus && nst >= 8) {
    const int cap = nst / 4 < 16 ? nst / 4 : 16;
    auto grid_of = [&](int wn, int* split) {
      const int blocks = ((N + 64 * wn - 1) / (64 * wn)) * pl.passes;
      int sp = num_cus / blocks;
      sp = sp > cap ? cap : sp;
      sp = sp < 1 ? 1 : sp;
      if (split) *split = sp;
      return blocks * sp;
    };
    int s1 = 1;
    const int g1 = grid_of(1, &s1), g2 = grid_of(2, nullptr);
    if (10 * g1 < 8 * num_cus && 10 * (g2 - g1) >= num_cus) pl.wn = 2;
    if (pl.mt == 8 && K / s1 >= 2048) pl.wn = 2;
  }
  if (force_wn == 1 || force_wn == 2) pl.wn = force_wn;
  pl.ncb = (N + 64 * pl.wn - 1) / (64 * pl.wn);
  int splits = num_cus / (pl.ncb * pl.passes);
  if (splits > nst / 4) splits = nst / 4;
  if (splits > 16) splits = 16;
  if (splits < 1) splits = 1;
  if (force_splits > 0) splits = force_splits < nst / 2 ? force_splits : (nst / 2 > 0 ? nst / 2 : 1);
  pl.splits = splits;
  pl.partial_bytes = splits > 1 ? (long long)pl.passes * pl.ncb * splits * 4 * pl.mt * pl.wn * 64 * 16 : 0;
  pl.n_counters = pl.passes * pl.ncb;
  return pl;
}

// Main loop of one workgroup: acc += X[m0 .. m0 + 16 MT, k range] * W[cb * 64 WN .. + 64 WN, k range]^T, the k range
// being stages [st0, st0 + nst) of 64.  `smem`: mid_lds_bytes(MT, WN) of dynamic LDS.  Accumulator of lane
// (li = lane & 15, lg = lane >> 4) of wave w: acc[m][j] = row m0 + 16 m + li, columns
// cb * 64 WN + w * 16 WN + 16 j + 4 lg .. + 4.
template <bool IS_BF16, int MT, int WN>
__device__ __forceinline__ void mid_mainloop(const uint16_t* X, long long ldx, const uint16_t* W, long long ldw, int M,
                                             int N, int m0, int cb, int st0, int nst, char* smem,
                                             f32x4 (&acc)[MT][WN]) {
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
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63, li = l & 15, lg = l >> 4;

  // Staging: a piece is one sub-tile [16 rows][32 k]; lane l moves row l >> 2, 16-byte chunk (l & 3) with the
  // image's XOR applied to the source.  Rows past the end are clamped (their results are masked at the store).
  const int kchunk = (l & 3) ^ (((l >> 5) & 1) << 1);
  const uint16_t* xsrc[NDX];
  int xdst[NDX];
#pragma unroll
  for (int i = 0; i < NDX; ++i) {
    const int q = w + 4 * i;                      // sub-tile (m tile q >> 1, k half q & 1)
    const int row = min(m0 + (q >> 1) * 16 + (l >> 2), M - 1);
    xsrc[i] = X + (long long)row * ldx + (long long)st0 * 64 + (q & 1) * 32 + kchunk * 8;
    xdst[i] = q * 1024;
  }
  const uint16_t* wsrc[WN];                       // this wave's strip: column group j, both k halves
#pragma unroll
  for (int j = 0; j < WN; ++j) {
    const int row = min(cb * BN_ + w * BNW + j * 16 + (l >> 2), N - 1);
    wsrc[j] = W + (long long)row * ldw + (long long)st0 * 64 + kchunk * 8;
  }
  const int w_area = X_BYTES + w * WS_BYTES;
  const int lane_off = ((li * 64) + (lg * 16)) ^ (((li >> 3) & 1) << 5);

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

}

// Store the accumulators as 16-bit values: store(row, col, uint2) gets four neighbouring columns of one row.
template <bool IS_BF16, int MT, int WN, class StoreFn>
__device__ __forceinline__ void mid_store(const f32x4 (&acc)[MT][WN], int m0, int cb, StoreFn&& store) {
  using F = Frag<IS_BF16>;
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, li = l & 15, lg = l >> 4;
  const int col0 = cb * 64 * WN + w * 16 * WN + 4 * lg;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int j = 0; j < WN; ++j) {
      uint2 v;
      v.x = F::pack2(acc[m][j][0], acc[m][j][1]);
      v.y = F::pack2(acc[m][j][2], acc[m][j][3]);
      store(m0 + m * 16 + li, col0 + j * 16, v);
    }
}

// Wait-free split-K join: park the accumulators (register layout, coalesced), count; the last split to
// arrive adds the others to its live accumulators, re-arms the counter and returns true - everybody else
// returns false and is done with the tile.  `tile_partial`: splits * 4 * MT * WN * 64 f32x4 of this tile.
template <int MT, int WN>
__device__ __forceinline__ bool mid_splitk_join(f32x4 (&acc)[MT][WN], f32x4* tile_partial, int* counter, int split,
                                                int S, volatile int* s_arrived) {
  constexpr int PER_SPLIT = 4 * MT * WN * 64;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  {
    f32x4* mine = tile_partial + (long long)split * PER_SPLIT + (w * MT * WN) * 64 + l;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < WN; ++j) st_wt128_agent(mine + (m * WN + j) * 64, acc[m][j]);
  }
  drain_stores();
  __syncthreads();
  if (tid == 0) *s_arrived = atom_add_relaxed_agent(counter, 1);
  __syncthreads();
  if (*s_arrived != S - 1) return false;
  if (tid < 64) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  if (tid == 0) store_relaxed_agent(counter, 0);      // re-armed for the next launch
  for (int q = 0; q < S; ++q) {
    if (q == split) continue;
    const f32x4* src = tile_partial + (long long)q * PER_SPLIT + (w * MT * WN) * 64 + l;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int j = 0; j < WN; ++j) acc[m][j] += src[(m * WN + j) * 64];
  }
  return true;
}

}  // namespace gemm
}  // namespace tdist
