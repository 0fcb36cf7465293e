// MoE block for a few hundred to a few thousand tokens on one rank: rows grouped by expert in blocks of 64 or 128.
//
// Between the sixteen-row blocks of the decode form (csrc/ops/gemm.hip: tdist_moe_group16_fwd - weights from the L2 for
// every block, fine while an expert has a block or two) and the 256-row tiles of the prefill path (csrc/ops/moe.hip:
// every expert padded to 256 rows, about 500 us per Qwen3-30B-A3B layer however few rows an expert has) sits the band
// where an expert has 30 .. 250 rows.  Here the pairs are sorted by expert into blocks of 16 MT rows (MT = 4 / 8), the
// activations are gathered into that order once, and every (block, column strip) is one workgroup of the mid-M GEMM
// body (gemm_mid_gfx950.hpp: activations and weights through a four-slot LDS ring kept three stages ahead by LDS-DMA,
// four waves side by side along N) against the block's expert.  Launch sequence (ops/moe.py: moe_group_mid_fwd): sort
// (tdist_moe_group_align), gather, grouped GEMM gate_up, SiLU * up, grouped GEMM down, weighted sum of every token's
// top-k rows through the inverse map of the sort.  Unused blocks of the static grid exit at once: no host
// synchronisation, so the sequence records into a HIP graph.
#include <stdlib.h>

#include "tdist/common.hpp"
#include "tdist/gemm_mid_gfx950.hpp"

using namespace tdist;
using namespace tdist::gemm;

namespace {

// xs[row, :] = x[sorted_ids[row] / topk, :] (zeros for padding rows); one workgroup per row of the blocks in use
__global__ __launch_bounds__(256) void moe_gather_rows_kernel(const uint16_t* __restrict__ x, const int* __restrict__ sorted_ids,
                                                              const int* __restrict__ meta, uint16_t* __restrict__ xs, int topk,
                                                              int H, int block_rows) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const int row = blockIdx.x;
  if (row >= meta[0] * block_rows) return;
  const int sid = sorted_ids[row];
  const uint16_t* src = x + (long long)(sid >= 0 ? sid / topk : 0) * H;
  for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
    u32x4 v = {0, 0, 0, 0};
    if (sid >= 0) v = *(const u32x4*)(src + c);
    *(u32x4*)(xs + (long long)row * H + c) = v;
  }
}

// C[16 MT b .. + 16 MT, :] = X[16 MT b .. + 16 MT, :] @ W[expert_of_block[b]]^T for every block b in use.  X, C in sorted
// row order ([16 MT nb, K] / [16 MT nb, N]); W [E, N, K].
template <int MT, int WN>
__global__ __launch_bounds__(MID_THREADS, 2) void moe_group_mid_kernel(const uint16_t* __restrict__ X, long long ldx,
                                                                       const uint16_t* __restrict__ W, long long w_estride,
                                                                       long long ldw, uint16_t* __restrict__ C, long long ldc,
                                                                       const int* __restrict__ expert_of_block,
                                                                       const int* __restrict__ meta, int rows_total, int N,
                                                                       int K) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int b = blockIdx.y;
  if (b >= meta[0]) return;
  const int e = expert_of_block[b];
  const int cb = blockIdx.x, m0 = b * 16 * MT;
  f32x4 acc[MT][WN];
  mid_mainloop<true, MT, WN>(X, ldx, W + (long long)e * w_estride, ldw, rows_total, N, m0, cb, 0, K / 64, smem, acc);
  mid_store<true, MT, WN>(acc, m0, cb, [&](int row, int col, const uint2& v) {
    if (col < N) *(uint2*)(C + (long long)row * ldc + col) = v;
  });
}

// out[t, :] = sum over k of wts[t, k] * y[pos[t * topk + k], :] (y bf16 in sorted row order, fp32 accumulation)
__global__ __launch_bounds__(256) void moe_topk_sum_gather_kernel(const uint16_t* __restrict__ y, const int* __restrict__ pos,
                                                                  const float* __restrict__ wts, uint16_t* __restrict__ out,
                                                                  int topk, int H) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const int t = blockIdx.x;
  for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < topk; ++k) {
      const int row = pos[t * topk + k];
      if (row < 0) continue;
      const float wk = wts[t * topk + k];
      const u32x4 v = *(const u32x4*)(y + (long long)row * H + c);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc[2 * i] += wk * __builtin_bit_cast(float, v[i] << 16);
        acc[2 * i + 1] += wk * __builtin_bit_cast(float, v[i] & 0xffff0000u);
      }
    }
    u32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = Frag<true>::pack2(acc[2 * i], acc[2 * i + 1]);
    *(u32x4*)(out + (long long)t * H + c) = o;
  }
}

template <int MT, int WN>
int launch_group_mid(const uint16_t* X, long long ldx, const uint16_t* W, long long w_estride, long long ldw, uint16_t* C,
                     long long ldc, const int* eob, const int* meta, int nb_max, int N, int K, hipStream_t s) {
  constexpr int LDS = mid_lds_bytes(MT, WN);
  static bool attr_set = false;
  if (!attr_set) {
    TDIST_HIP_TRY(hipFuncSetAttribute((const void*)moe_group_mid_kernel<MT, WN>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    attr_set = true;
  }
  const int ncb = (N + 64 * WN - 1) / (64 * WN);
  hipLaunchKernelGGL((moe_group_mid_kernel<MT, WN>), dim3(ncb, nb_max), dim3(MID_THREADS), LDS, s, X, ldx, W, w_estride, ldw, C,
                     ldc, eob, meta, nb_max * 16 * MT, N, K);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

}  // namespace

// xs [block_rows * nb_max, H] <- rows of x [tokens, H] in sorted order (zeros for padding rows); bf16.
TDIST_API int tdist_moe_gather_rows(const void* x, const int* sorted_ids, const int* meta, void* xs, int topk, int H,
                                    int block_rows, int nb_max, void* stream) {
  TDIST_REQUIRE(H > 0 && H % 8 == 0 && topk > 0 && nb_max > 0 && block_rows > 0, "H must be a multiple of 8");
  TDIST_REQUIRE((((uintptr_t)x | (uintptr_t)xs) & 15) == 0 && sorted_ids && meta, "operands must be 16-byte aligned");
  hipLaunchKernelGGL(moe_gather_rows_kernel, dim3(nb_max * block_rows), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
                     sorted_ids, meta, (uint16_t*)xs, topk, H, block_rows);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

// C [block_rows * nb_max, N] = per block: X rows @ W[expert]^T.  X [block_rows * nb_max, K], W [E, N, K]; bf16;
// block_rows 32, 64 or 128, wn 1 or 2 (64 / 128 columns per workgroup).
TDIST_API int tdist_moe_group_mid_gemm(const void* X, const void* W, void* C, const int* expert_of_block, const int* meta,
                                       int nb_max, int block_rows, int wn, int N, int K, void* stream) {
  TDIST_REQUIRE(nb_max > 0 && nb_max <= 65535 && (block_rows == 32 || block_rows == 64 || block_rows == 128) && (wn == 1 || wn == 2),
                "bad block plan");
  TDIST_REQUIRE(N > 0 && N % 8 == 0 && K >= 128 && K % 64 == 0, "N must be a multiple of 8, K of 64 (>= 128)");
  TDIST_REQUIRE((((uintptr_t)X | (uintptr_t)W) & 15) == 0 && (((uintptr_t)C) & 7) == 0 && expert_of_block && meta,
                "operands must be aligned");
  const uint16_t *x = (const uint16_t*)X, *w = (const uint16_t*)W;
  uint16_t* c = (uint16_t*)C;
  hipStream_t s = (hipStream_t)stream;
  const long long es = (long long)N * K;
  if (block_rows == 32)
    return wn == 1 ? launch_group_mid<2, 1>(x, K, w, es, K, c, N, expert_of_block, meta, nb_max, N, K, s)
                   : launch_group_mid<2, 2>(x, K, w, es, K, c, N, expert_of_block, meta, nb_max, N, K, s);
  if (block_rows == 64)
    return wn == 1 ? launch_group_mid<4, 1>(x, K, w, es, K, c, N, expert_of_block, meta, nb_max, N, K, s)
                   : launch_group_mid<4, 2>(x, K, w, es, K, c, N, expert_of_block, meta, nb_max, N, K, s);
  return wn == 1 ? launch_group_mid<8, 1>(x, K, w, es, K, c, N, expert_of_block, meta, nb_max, N, K, s)
                 : launch_group_mid<8, 2>(x, K, w, es, K, c, N, expert_of_block, meta, nb_max, N, K, s);
}

// out [tokens, H] = sum over k of wts[t, k] * y[pos[t, k], :]; y bf16 [sorted rows, H].
TDIST_API int tdist_moe_topk_sum_gather(const void* y, const int* pos, const float* wts, void* out, int tokens, int topk, int H,
                                        void* stream) {
  TDIST_REQUIRE(tokens > 0 && topk > 0 && H > 0 && H % 8 == 0, "H must be a multiple of 8");
  TDIST_REQUIRE((((uintptr_t)y | (uintptr_t)out) & 15) == 0 && pos && wts, "operands must be 16-byte aligned");
  hipLaunchKernelGGL(moe_topk_sum_gather_kernel, dim3(tokens), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)y, pos, wts,
                     (uint16_t*)out, topk, H);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
