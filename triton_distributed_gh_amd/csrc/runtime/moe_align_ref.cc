// CPU reference of the MoE align / tile-order computation (test oracle of the HIP kernel
// tdist_moe_align in csrc/ops/moe.hip).
//
// Role in the reference: kernels/nvidia/threadblock_swizzle_ag_moe.cc:109-229 - a CPU twin of the
// AG-MoE tile swizzle that the CUDA program checks itself against - and csrc/lib/moe_utils.cu:60-192
// (align-to-block-size).  The algorithm here is this framework's own formulation:
//   1. every flat index i = token * topk + k has an expert e(i) and a source rank
//      s(i) = min(token / tokens_per_rank, w - 1);
//   2. rows are ordered by (expert, source rank, i) and every expert's row range is padded to a
//      multiple of BLOCK (256) with -1;
//   3. per 256-row block: its expert and the [min, max] source rank of its rows;
//   4. blocks are issued in "arrival stage" order: stage(b) = max over the block's source ranks of
//      (s - rank + w) % w, i.e. how far round the push order the last needed shard is; ties keep
//      block order.  Own-rank data is stage 0 and is computed first.
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

#define TDH_API extern "C" __attribute__((visibility("default")))

TDH_API int tdh_moe_align_reference(const int32_t* topk_ids, int n, int num_experts, int world, int topk,
                                    int tokens_per_rank, int rank, int cap, int block, int32_t* sorted_ids,
                                    int32_t* expert_of_block, int32_t* seg_range, int32_t* order, int32_t* meta) {
  if (n < 0 || num_experts <= 0 || world <= 0 || topk <= 0 || tokens_per_rank <= 0 || block <= 0 || cap % block) return -1;
  auto src_of = [&](int i) { return std::min((i / topk) / tokens_per_rank, world - 1); };
  std::vector<int64_t> counts(num_experts, 0);
  for (int i = 0; i < n; ++i) {
    if (topk_ids[i] < 0 || topk_ids[i] >= num_experts) return -2;
    ++counts[topk_ids[i]];
  }
  std::vector<int64_t> pad_off(num_experts + 1, 0);
  for (int e = 0; e < num_experts; ++e) pad_off[e + 1] = pad_off[e] + (counts[e] + block - 1) / block * block;
  if (pad_off[num_experts] > cap) return -3;
  std::vector<int> idx(n);
  std::iota(idx.begin(), idx.end(), 0);
  std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) {
    const int64_t ka = (int64_t)topk_ids[a] * world + src_of(a), kb = (int64_t)topk_ids[b] * world + src_of(b);
    return ka < kb;
  });
  std::fill(sorted_ids, sorted_ids + cap, -1);
  std::vector<int64_t> cursor(pad_off.begin(), pad_off.end() - 1);
  for (int i : idx) sorted_ids[cursor[topk_ids[i]]++] = i;
  const int nblocks = (int)(pad_off[num_experts] / block);
  const int nb_cap = cap / block;
  std::fill(expert_of_block, expert_of_block + nb_cap, 0);
  std::fill(seg_range, seg_range + 2 * nb_cap, 0);
  std::fill(order, order + nb_cap, 0);
  std::vector<int> stage(nblocks, 0);
  for (int b = 0; b < nblocks; ++b) {
    int lo = world, hi = -1, e = 0;
    for (int r = 0; r < block; ++r) {
      const int i = sorted_ids[(int64_t)b * block + r];
      if (i < 0) continue;
      e = topk_ids[i];
      lo = std::min(lo, src_of(i));
      hi = std::max(hi, src_of(i));
    }
    expert_of_block[b] = e;
    seg_range[2 * b] = lo;
    seg_range[2 * b + 1] = hi;
    int st = 0;
    for (int s = lo; s <= hi; ++s) st = std::max(st, (s - rank + world) % world);
    stage[b] = st;
  }
  // eight contiguous ranges of blocks (one per XCD of the grouped GEMM), each in arrival order
  std::vector<int> ord(nblocks);
  std::iota(ord.begin(), ord.end(), 0);
  for (int q = 0; q < 8; ++q) {
    const int64_t lo = (int64_t)q * nblocks / 8, hi = (int64_t)(q + 1) * nblocks / 8;
    std::stable_sort(ord.begin() + lo, ord.begin() + hi, [&](int a, int b) { return stage[a] < stage[b]; });
  }
  for (int b = 0; b < nblocks; ++b) order[b] = ord[b];
  meta[0] = nblocks;
  meta[1] = (int32_t)pad_off[num_experts];
  for (int e = 0; e < num_experts; ++e) meta[2 + e] = (int32_t)counts[e];
  return 0;
}
