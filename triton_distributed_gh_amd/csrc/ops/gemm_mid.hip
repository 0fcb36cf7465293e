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

#include "tdist/common.hpp"
#include "tdist/gemm_mid_gfx950.hpp"

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
__global__ __launch_bounds__(MID_THREADS, 2) void gemm_nt_mid_kernel(MidParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_arrived;
  const int ncb = (p.N + 64 * WN - 1) / (64 * WN);
  const int cb = blockIdx.x % ncb, split = blockIdx.x / ncb, S = p.splits;
  const int m0 = blockIdx.y * 16 * MT;
  const int nst_total = p.K / 64;
  const int st0 = (int)((long long)nst_total * split / S), st1 = (int)((long long)nst_total * (split + 1) / S);
  f32x4 acc[MT][WN];
  mid_mainloop<IS_BF16, MT, WN>(p.X, p.ldx, p.W, p.ldw, p.M, p.N, m0, cb, st0, st1 - st0, smem, acc);
  if (S > 1) {
    const int tile = blockIdx.y * ncb + cb;
    f32x4* tile_partial = (f32x4*)p.partial + (long long)tile * S * (4 * MT * WN * 64);
    if (!mid_splitk_join<MT, WN>(acc, tile_partial, p.counters + tile, split, S, &s_arrived)) return;
  }
  mid_store<IS_BF16, MT, WN>(acc, m0, cb, [&](int row, int col, const uint2& v) {
    if (row < p.M && col < p.N) *(uint2*)(p.C + (long long)row * p.ldc + col) = v;
  });
}

namespace {

template <bool IS_BF16, int MT, int WN>
int launch_mid(const MidParams& p, int passes, hipStream_t s) {
  constexpr int LDS = mid_lds_bytes(MT, WN);
  const int ncb = (p.N + 64 * WN - 1) / (64 * WN);
  static bool attr_set = false;
  if (!attr_set) {
    // (up to 128 KiB of ring: above the default dynamic limit)
    TDIST_HIP_TRY(hipFuncSetAttribute((const void*)gemm_nt_mid_kernel<IS_BF16, MT, WN>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  hipLaunchKernelGGL((gemm_nt_mid_kernel<IS_BF16, MT, WN>), dim3(ncb * p.splits, passes), dim3(MID_THREADS), LDS, s, p);
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
  const MidPlan pl = mid_plan(M, N, K, num_cus, force_wn, force_splits);
  *mt_out = pl.mt;
  *wn_out = pl.wn;
  *splits_out = pl.splits;
  *partial_bytes = pl.partial_bytes;
  *n_counters = pl.n_counters;
  return 0;
}

// Identity of the capture `stream` is recording into (0 = not capturing).  Unique per capture, so a caller
// can keep one workspace per graph.
TDIST_API int tdist_stream_capture_id(void* stream, unsigned long long* id_out) {
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  TDIST_HIP_TRY(hipStreamGetCaptureInfo((hipStream_t)stream, &status, &id));
  *id_out = status == hipStreamCaptureStatusActive ? id : 0;
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
