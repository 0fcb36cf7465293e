// Device-side pieces of GQA decode attention shared by the stand-alone kernels
// (csrc/ops/flash_decode.hip) and the task interpreter of the persistent mega kernel
// (csrc/ops/mega_kernel.hip): argument block, paged-slot lookup, the split-KV body and the
// LSE-weighted row merge.  See flash_decode.hip for the algorithm description.
#pragma once
#include "tdist/atomics.hpp"
#include "tdist/common.hpp"
#include "tdist/gemm_gfx950.hpp"

namespace tdist {
namespace fdec {


using tdist::gemm::f32x4;
using tdist::gemm::Frag;

struct DecodeArgs {
  const uint16_t* q;         // [batch, q_heads, D]
  const uint16_t* k_cache;   // [slots, kv_heads, D] (slot = page * page_size + offset)
  const uint16_t* v_cache;
  const int* block_table;    // [batch, max_pages]
  const int* kv_len;         // [batch]
  void* partial;             // [batch, q_heads, splits, D + 1], fp32 or (p16) the 16-bit type of q
  void* out;                 // fused: [batch, q_heads, D] (16-bit) / [batch, q_heads, D + 1] (fp32 or 16-bit, LSE last)
  int* tickets;              // fused: [batch * gridDim.y], zero before the first launch (self-resetting)
  long long q_stride_b, q_stride_h;
  long long k_stride_slot, k_stride_h;
  long long v_stride_slot, v_stride_h;
  long long p_stride_b, p_stride_h, p_stride_s;
  long long o_stride_b, o_stride_h;
  int table_stride;
  int page_size, page_shift;  // page_shift >= 0 when page_size is a power of two
  int q_heads, kv_heads, splits;
  int fused;                  // 0: partials only, 1: merged 16-bit out, 2: merged fp32 out + LSE column,
                              // 3: merged 16-bit out + LSE column
  int p16;                    // 1: partial rows in the 16-bit type of q (the reference's *_fp16_fp16_fp16 kernels)
  int deep;                   // 1: two register sets (loads one block ahead): throughput launches, many workgroups per CU
  float sm_scale, soft_cap;
  // PREP (decode_split_body<..., PREP>): the step's Q / K / V come straight from the QKV projection
  const uint16_t* qkv = nullptr;   // [batch, qkv_stride]: q heads, then k heads, then v heads of the new token
  long long qkv_stride = 0;
  const uint16_t* q_w = nullptr;   // per-head RMSNorm weights [D] (null: no norm)
  const uint16_t* k_w = nullptr;
  float q_eps = 0.f, k_eps = 0.f;
  const long long* pos = nullptr;  // [batch] rotary position of the new token
  const float* cos_t = nullptr;    // [positions, D / 2]
  const float* sin_t = nullptr;
  const int* kv_offset = nullptr;  // [batch] tokens in the cache before this step
  int* total_out = nullptr;        // [batch] <- kv_offset + 1
  long long max_slots = 0;         // slots per sequence (a full cache drops the new token instead of writing outside)
  // 1 (the grid was sized from the cache CAPACITY): a sequence uses as many of the `splits` as its own length calls
  // for - fused launches min(splits, ~sqrt(kv_len / 32)), the optimum of streaming ~ L / s against merging ~ s in one
  // workgroup, and the workgroups of later splits leave at once; launches with a separate merge kernel
  // min(splits, ~sqrt(kv_len / 16), kv_len / 128), and later splits write empty partial rows
  int adaptive = 0;
};

__device__ __forceinline__ long long slot_of(const DecodeArgs& a, const int* table, int tok) {
  if (a.page_shift >= 0) {
    const int page = table[tok >> a.page_shift];
    return ((long long)page << a.page_shift) + (tok & (a.page_size - 1));
  }
  const int pi = tok / a.page_size;
  return (long long)table[pi] * a.page_size + (tok - pi * a.page_size);
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// K / V rows are read exactly once per call.  Non-temporal loads were measured (TDIST_FDEC_NT_LOADS at build
// time): 301 us against 289 us for plain loads at batch 64 x 4096 tokens - the L1 miss queue, not cache
// pollution, is the limit - so plain loads stay.
__device__ __forceinline__ uint4 ld_stream16(const uint16_t* p) {
#ifndef TDIST_FDEC_NT_LOADS
  return *(const uint4*)p;
#else
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const u32x4 v = __builtin_nontemporal_load((const u32x4*)p);
  return uint4{v.x, v.y, v.z, v.w};
#endif
}

template <bool IS_BF16>
__device__ __forceinline__ uint16_t to16(float f) {
  if constexpr (IS_BF16) {
    return __builtin_bit_cast(uint16_t, (__bf16)f);
  } else {
    return __builtin_bit_cast(uint16_t, (_Float16)f);
  }
}

// Merge `n` partial rows (stride `stride` floats, LSE at column D) of one head, cooperatively by a
// group of `nthreads` (16 or 64) consecutive lanes of one wave, `tid` = lane inside the group.
// The LSE maximum and the denominator are computed with the splits spread over the lanes and
// folded by cross-lane shuffles; the weighted sum keeps four columns per lane in flight per split
// so the loads of one pass overlap instead of queueing behind each other.  `lens` (optional)
// skips empty items, whose rows may never have been written.
template <bool IS_BF16>
__device__ __forceinline__ float from16(uint16_t v) {
  if constexpr (IS_BF16) {
    return __builtin_bit_cast(float, (uint32_t)v << 16);
  } else {
    return (float)__builtin_bit_cast(_Float16, v);
  }
}

// Merge of fp32 partial rows of 128 values + LSE by ONE wave (the head-parallel merge launch).  Lane (cg = lane & 15,
// rg = lane >> 4) owns columns 8 cg .. 8 cg + 7 of rows rg, rg + 4, ...: for a chunk of 32 rows every lane issues its
// eight rows' loads - two 16-byte loads and the LSE each - before it uses any of them, so a merge of up to 32 splits
// is ONE round trip to memory (the rows were written by the previous launch and sit in HBM or the memory-side
// cache; the loop that kept eight rows in flight per lane took 11.7 us for 32 heads x 32 splits, most of it
// waiting).  Lane groups keep running (max, sum, row) states and are merged by two butterfly steps at the end.
// out_mode as for merge_rows.
typedef float f32x4_u __attribute__((ext_vector_type(4), aligned(4)));

template <bool IS_BF16>
__device__ __forceinline__ void merge_rows_wave_d128(const float* part, long long stride, int n, const int* lens,
                                                     int lens_stride, void* out, int out_mode, int lane) {
  constexpr int D = 128;
  // "no tokens" is a large negative FINITE number in here (an empty split stores -inf, clamped on load): the file is
  // built with finite-math assumptions, under which a test against infinity may be folded away and inf - inf is a NaN
  constexpr float NEG = -1.0e30f;
  const int cg = lane & 15, rg = lane >> 4;
  float m_run = NEG, l_run = 0.f;
  float acc[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) acc[k] = 0.f;
  for (int s0 = 0; s0 < n; s0 += 32) {
    f32x4_u v[8][2];
    float l[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int s = s0 + rg + 4 * j;
      const bool valid = s < n && !(lens && lens[(long long)s * lens_stride] <= 0);
      const float* row = part + (long long)min(s, n - 1) * stride;
      v[j][0] = *(const f32x4_u*)(row + cg * 8);
      v[j][1] = *(const f32x4_u*)(row + cg * 8 + 4);
      l[j] = valid ? fmaxf(row[D], NEG) : NEG;
    }
    float m_new = m_run;
#pragma unroll
    for (int j = 0; j < 8; ++j) m_new = fmaxf(m_new, l[j]);
    const float alpha = __expf(m_run - m_new);
    l_run *= alpha;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] *= alpha;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float w = l[j] <= NEG ? 0.f : __expf(l[j] - m_new);
      l_run += w;
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += w != 0.f ? w * v[j][k >> 2][k & 3] : 0.f;   // (an unused row may hold anything)
    }
    m_run = m_new;
  }
#pragma unroll
  for (int off = 16; off <= 32; off <<= 1) {
    const float m_o = __shfl_xor(m_run, off), l_o = __shfl_xor(l_run, off);
    const float m_new = fmaxf(m_run, m_o);
    const float a = __expf(m_run - m_new), b = __expf(m_o - m_new);
    l_run = l_run * a + l_o * b;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = acc[k] * a + __shfl_xor(acc[k], off) * b;
    m_run = m_new;
  }
  const float inv = l_run > 0.f ? 1.f / l_run : 0.f;
  if (rg == 0) {
    if (out_mode == 2) {
      float* o = (float*)out + cg * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = acc[k] * inv;
    } else {
      uint16_t* o = (uint16_t*)out + cg * 8;
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = to16<IS_BF16>(acc[k] * inv);
    }
  }
  if (out_mode >= 2 && lane == 0) {
    const float lse = l_run > 0.f ? m_run + __logf(l_run) : -INFINITY;
    if (out_mode == 2) {
      ((float*)out)[D] = lse;
    } else {
      ((uint16_t*)out)[D] = to16<IS_BF16>(lse);
    }
  }
}

// out_mode: 1 = 16-bit row, 2 = fp32 row + LSE at column D, 3 = 16-bit row + LSE at column D.
// P16: the partial rows are in the 16-bit type instead of fp32.
template <bool IS_BF16, bool P16 = false>
__device__ __forceinline__ void merge_rows(const void* part_, long long stride, int n, int D, const int* lens,
                                           int lens_stride, void* out, int out_mode, int tid, int nthreads) {
  struct Rows {
    const void* base;
    __device__ __forceinline__ float operator[](long long i) const {
      if constexpr (P16) {
        return from16<IS_BF16>(((const uint16_t*)base)[i]);
      } else {
        return ((const float*)base)[i];
      }
    }
    __device__ __forceinline__ Rows operator+(long long off) const {
      return Rows{(const char*)base + off * (P16 ? 2 : 4)};
    }
  };
  const Rows part{part_};
  // Up to 4 * nthreads rows (every launch of this package: at most 64 splits, 16 or 64 threads per head): the LSE
  // column is read ONCE - thread t keeps the weights of rows t, t + nthreads, ... and hands them out by shuffle -
  // and the value loads of eight rows are in flight together.  (Before, the accumulation re-read the LSE of every
  // row and kept two rows in flight: ~0.5 us per split for the one workgroup that merges a head block.)
  if (n <= 4 * nthreads && D >= nthreads) {
    float l[4], w[4];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int s = tid + j * nthreads;
      l[j] = -INFINITY;
      if (s < n && !(lens && lens[(long long)s * lens_stride] <= 0)) l[j] = part[s * stride + D];
      mx = fmaxf(mx, l[j]);
    }
    for (int off = nthreads >> 1; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      w[j] = (l[j] == -INFINITY) ? 0.f : __expf(l[j] - mx);
      den += w[j];
    }
    for (int off = nthreads >> 1; off > 0; off >>= 1) den += __shfl_xor(den, off);
    const float inv = den > 0.f ? 1.f / den : 0.f;
    for (int c0 = tid; c0 < D; c0 += 4 * nthreads) {
      float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int sn = min(nthreads, n - j * nthreads);
#pragma unroll 8
        for (int si = 0; si < sn; ++si) {
          const float wgt = __shfl(w[j], si, nthreads);
          const Rows row = part + (long long)(j * nthreads + si) * stride;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int c = c0 + k * nthreads;
            const float v = (c < D && wgt != 0.f) ? row[c] : 0.f;
            acc[k] += wgt * v;
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = c0 + k * nthreads;
        if (c >= D) continue;
        const float r = acc[k] * inv;
        if (out_mode == 2) {
          ((float*)out)[c] = r;
        } else {
          ((uint16_t*)out)[c] = to16<IS_BF16>(r);
        }
      }
    }
    if (out_mode >= 2 && tid == 0) {
      const float lse = den > 0.f ? mx + __logf(den) : -INFINITY;
      if (out_mode == 2) {
        ((float*)out)[D] = lse;
      } else {
        ((uint16_t*)out)[D] = to16<IS_BF16>(lse);
      }
    }
    return;
  }
  float mx = -INFINITY;
  for (int s = tid; s < n; s += nthreads) {
    if (lens && lens[(long long)s * lens_stride] <= 0) continue;
    mx = fmaxf(mx, part[s * stride + D]);
  }
  for (int off = nthreads >> 1; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  float den = 0.f;
  for (int s = tid; s < n; s += nthreads) {
    if (lens && lens[(long long)s * lens_stride] <= 0) continue;
    const float l = part[s * stride + D];
    den += (l == -INFINITY) ? 0.f : __expf(l - mx);
  }
  for (int off = nthreads >> 1; off > 0; off >>= 1) den += __shfl_xor(den, off);
  const float inv = den > 0.f ? 1.f / den : 0.f;
  for (int c0 = tid; c0 < D; c0 += 4 * nthreads) {
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int s = 0; s < n; ++s) {
      const bool skip = lens && lens[(long long)s * lens_stride] <= 0;
      const Rows row = part + s * stride;
      const float l = skip ? -INFINITY : row[D];
      const float wgt = (l == -INFINITY) ? 0.f : __expf(l - mx);
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = c0 + k * nthreads;
        v[k] = (c < D && wgt != 0.f) ? row[c] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] += wgt * v[k];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int c = c0 + k * nthreads;
      if (c >= D) continue;
      const float r = acc[k] * inv;
      if (out_mode == 2) {
        ((float*)out)[c] = r;
      } else {
        ((uint16_t*)out)[c] = to16<IS_BF16>(r);
      }
    }
  }
  if (out_mode >= 2 && tid == 0) {
    const float lse = den > 0.f ? mx + __logf(den) : -INFINITY;
    if (out_mode == 2) {
      ((float*)out)[D] = lse;
    } else {
      ((uint16_t*)out)[D] = to16<IS_BF16>(lse);
    }
  }
}

// One head of the new token, D = 128, as the lane (k-group lg) of an MFMA operand holds it: elements 32 s + 8 lg + j,
// s = 0..3.  Optional RMSNorm over the head (the four lanes lg of one row hold all of it: two shuffles), rounded to
// bf16 like the stand-alone kernel, then the neox rotation, whose pairs (e, e + 64) are chunks s and s + 2 of the
// SAME lane.  `src` / `w` already point at element 8 lg; `cs` / `sn` at [position][8 lg].
__device__ __forceinline__ void prep_head_d128(const uint16_t* src, bool valid, const uint16_t* w, float eps, const float* cs,
                                               const float* sn, uint4 (&out)[4]) {
  float x[4][8];
  float ss = 0.f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    uint4 v = uint4{0, 0, 0, 0};
    if (valid) v = *(const uint4*)(src + 32 * s);
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      x[s][2 * i] = __builtin_bit_cast(float, u[i] << 16);
      x[s][2 * i + 1] = __builtin_bit_cast(float, u[i] & 0xffff0000u);
      ss += x[s][2 * i] * x[s][2 * i] + x[s][2 * i + 1] * x[s][2 * i + 1];
    }
  }
  if (w) {
    ss += __shfl_xor(ss, 16);
    ss += __shfl_xor(ss, 32);
    const float inv = rsqrtf(ss / 128.f + eps);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const uint4 wv = *(const uint4*)(w + 32 * s);
      const uint32_t u[4] = {wv.x, wv.y, wv.z, wv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        x[s][2 * i] = (float)(__bf16)(x[s][2 * i] * inv * __builtin_bit_cast(float, u[i] << 16));
        x[s][2 * i + 1] = (float)(__bf16)(x[s][2 * i + 1] * inv * __builtin_bit_cast(float, u[i] & 0xffff0000u));
      }
    }
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    float co[8], si[8];
    *(float4*)co = *(const float4*)(cs + 32 * s);
    *(float4*)(co + 4) = *(const float4*)(cs + 32 * s + 4);
    *(float4*)si = *(const float4*)(sn + 32 * s);
    *(float4*)(si + 4) = *(const float4*)(sn + 32 * s + 4);
    uint32_t lo[4], hi[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float a0 = x[s][2 * i], a1 = x[s][2 * i + 1], c0 = x[s + 2][2 * i], c1 = x[s + 2][2 * i + 1];
      lo[i] = Frag<true>::pack2(a0 * co[2 * i] - c0 * si[2 * i], a1 * co[2 * i + 1] - c1 * si[2 * i + 1]);
      hi[i] = Frag<true>::pack2(c0 * co[2 * i] + a0 * si[2 * i], c1 * co[2 * i + 1] + a1 * si[2 * i + 1]);
    }
    out[s] = uint4{lo[0], lo[1], lo[2], lo[3]};
    out[s + 2] = uint4{hi[0], hi[1], hi[2], hi[3]};
  }
}

// Split-KV body for work item (batch b, head block by, split z); writes the normalised partial row
// (+ LSE) of every head of the block.  `smem` needs (128 + 64 * D) floats; 256 threads.
// ALLOW_DEEP = false compiles the single-register-set loop only (the mega kernel: its task interpreter must
// keep two workgroups per CU resident, and the second set costs 56 VGPRs).
//
// PREP (bf16, D = 128, one new token per sequence, whole-sequence pages): the kernel replaces the launch between
// the QKV projection and the attention - every workgroup norms and rotates its own query heads out of `qkv`, the
// one wave whose 32-token block holds the new position norms and rotates the new K row, takes K / V of that token
// from registers instead of the cache and (head block 0) appends them to the cache for the steps to come.
// Returns the number of splits in use for this sequence (a.splits unless a.adaptive).
template <bool IS_BF16, int D, bool ALLOW_DEEP = true, bool PREP = false>
__device__ __forceinline__ int decode_split_body(const DecodeArgs& a, const int b, const int by, const int z, char* smem) {
  static_assert(!PREP || (IS_BF16 && D == 128), "the fused projection epilogue is built for bf16, head_dim 128");
  using F = Frag<IS_BF16>;
  using VT = typename F::T;
  constexpr int KS = D / 32;    // MFMA k-steps over the head dim for S
  constexpr int NDV = D / 16;   // dv blocks == 16-bit elements of V per lane per token
  constexpr int VW = NDV / 2;   // dwords of V per lane per token
  constexpr float LOG2E = 1.4426950408889634f;

  float* s_m = (float*)smem;                    // [4][16]
  float* s_l = s_m + 64;                        // [4][16]
  float* s_acc = s_l + 64;                      // [4][16][D]

  const int group = a.q_heads / a.kv_heads;
  const int hblocks = (group + 15) / 16;
  const int kvh = by / hblocks;
  const int hblk = by - kvh * hblocks;
  const int h0 = kvh * group + hblk * 16;
  const int valid_h = min(16, group - hblk * 16);

  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int li = lane & 15;   // head (S / O column), token row of K operand, dv column of V operand
  const int lg = lane >> 4;   // k-group

  int kv_len;
  if constexpr (PREP) {
    kv_len = a.kv_offset[b] + 1;
    if (by == 0 && z == 0 && threadIdx.x == 0) a.total_out[b] = kv_len;
  } else {
    kv_len = a.kv_len[b];
  }
  int nsplit = a.splits;
  if (a.adaptive) {
    nsplit = min(nsplit, max(1, (int)(__builtin_sqrtf((float)kv_len * (1.f / 16.f)) + 0.5f)));
    if (!a.fused) nsplit = min(nsplit, max(1, kv_len >> 7));
    if (a.fused && z >= nsplit) return nsplit;
  }
  const int per = (((kv_len + nsplit - 1) / nsplit) + 31) & ~31;
  const int start = z * per;
  const int end = min(start + per, kv_len);
  const int* table = a.block_table + (long long)b * a.table_stride;

  // Q operand: lane (head li, k-group lg) holds Q[h0 + li][32 s + 8 lg .. +8]
  VT qreg[KS];
  if constexpr (PREP) {
    const long long p = a.pos[b];
    uint4 qv[4];
    prep_head_d128(a.qkv + b * a.qkv_stride + (long long)(h0 + li) * D + lg * 8, li < valid_h, a.q_w ? a.q_w + lg * 8 : nullptr,
                   a.q_eps, a.cos_t + p * (D / 2) + lg * 8, a.sin_t + p * (D / 2) + lg * 8, qv);
#pragma unroll
    for (int s = 0; s < KS; ++s) qreg[s] = __builtin_bit_cast(VT, qv[s]);
  } else {
    const uint16_t* qp = a.q + b * a.q_stride_b + (long long)(h0 + li) * a.q_stride_h + lg * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      uint4 v = uint4{0, 0, 0, 0};
      if (li < valid_h) v = *(const uint4*)(qp + 32 * s);
      qreg[s] = __builtin_bit_cast(VT, v);
    }
  }

  float m_run = -INFINITY, l_run = 0.f;
  f32x4 acc[NDV];
#pragma unroll
  for (int j = 0; j < NDV; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float qk_scale = a.soft_cap > 0.f ? a.sm_scale / a.soft_cap : a.sm_scale * LOG2E;
  const uint16_t* kbase = a.k_cache + (long long)kvh * a.k_stride_h + lg * 8;
  const uint16_t* vbase = a.v_cache + (long long)kvh * a.v_stride_h + li * NDV;

  // Pages of a power-of-two size >= 32 tokens: a wave's 32-token block (32-aligned) lies in ONE page, so the
  // page index is a single wave-uniform (scalar) look-up per block, fetched one block ahead, instead of ten
  // per-lane look-ups through the vector L1 in front of the sixteen K / V loads that depend on them.
  // The same holds for any page size that is a multiple of 32, and for a sequence that fits its first page
  // whatever the page size (a cache of one contiguous run of max_len slots per sequence: max_len is arbitrary).
  // PREP with a full cache (kv_offset == max_slots: the new token is attended from registers and dropped): the
  // attended range is one token longer than the cache, so everything that ADDRESSES the cache is clamped to its
  // last slot / last 32-token block - the block that holds only the new token would otherwise look up a page
  // beyond the sequence's table.
  const int tok_cap = PREP ? min(kv_len, a.max_slots) - 1 : kv_len - 1;
  const int n0_cap = PREP ? (tok_cap & ~31) : 0x7fffffff;
  const bool single = tok_cap < a.page_size;
  const bool one_page = single || a.page_shift >= 5 || (a.page_shift < 0 && (a.page_size & 31) == 0);
  auto block_slot0 = [&](int n0) -> long long {     // slot of token n0 (wave-uniform)
    const int n0u = __builtin_amdgcn_readfirstlane(min(n0, n0_cap));
    if (single) return (long long)table[0] * a.page_size + n0u;
    if (a.page_shift >= 0) {
      const int page = table[n0u >> a.page_shift];
      return ((long long)page << a.page_shift) + (n0u & (a.page_size - 1));
    }
    const int pi = n0u / a.page_size;
    return (long long)table[pi] * a.page_size + (n0u - pi * a.page_size);
  };
  // Pages shorter than a block: the lanes look their pages up one by one.  The split's share of the page table
  // ((end - start) / page_size entries) is staged into LDS once - the accumulator area, unused until the final merge
  // - so the look-up in front of every K / V load is an LDS read instead of a dependent trip through the vector L1.
  int* s_pages = (int*)s_acc;
  const int pi0 = a.page_shift >= 0 ? (start >> a.page_shift) : start / a.page_size;
  const int pi1 = end > start ? (a.page_shift >= 0 ? ((end - 1) >> a.page_shift) : (end - 1) / a.page_size) : pi0 - 1;
  const bool lds_pages = !one_page && pi1 >= pi0 && pi1 - pi0 + 1 <= 64 * D;
  if (lds_pages) {
    for (int i = threadIdx.x; i <= pi1 - pi0; i += blockDim.x) s_pages[i] = table[pi0 + i];
    __syncthreads();
  }
  auto slot_lookup = [&](int tok) -> long long {
    if (!lds_pages) return slot_of(a, table, tok);
    const int pi = a.page_shift >= 0 ? (tok >> a.page_shift) : tok / a.page_size;
    return (long long)s_pages[pi - pi0] * a.page_size + (tok - pi * a.page_size);
  };
  // One 32-token block of K and V in registers (16 x 16 bytes per lane for D = 128).
  struct KVBlock {
    uint4 k[2][KS];
    uint32_t v[8][VW];
  };
  auto load_block = [&](int n0, long long slot0, KVBlock& r) {
    const int n0c = min(n0, n0_cap);
    auto slot_in_block = [&](int tok) -> long long {      // tok in [n0c, n0c + 32), already clamped to tok_cap
      return one_page ? slot0 + (tok - n0c) : slot_lookup(tok);
    };
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      const int tok = min(n0 + 16 * blk + li, tok_cap);
      const uint16_t* kp = kbase + slot_in_block(tok) * a.k_stride_slot;
#pragma unroll
      for (int s = 0; s < KS; ++s) r.k[blk][s] = ld_stream16(kp + 32 * s);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int tok = min(n0 + 16 * (t >> 2) + 4 * lg + (t & 3), tok_cap);
      const uint16_t* vp = vbase + slot_in_block(tok) * a.v_stride_slot;
      if constexpr (VW >= 4) {
#pragma unroll
        for (int c = 0; c < VW / 4; ++c) {
          const uint4 v = ld_stream16(vp + 8 * c);
          r.v[t][4 * c] = v.x, r.v[t][4 * c + 1] = v.y, r.v[t][4 * c + 2] = v.z, r.v[t][4 * c + 3] = v.w;
        }
      } else {
        const uint2 v = *(const uint2*)vp;
        r.v[t][0] = v.x, r.v[t][1] = v.y;
      }
    }
    if constexpr (PREP) {
      const int last = kv_len - 1;
      if (n0 <= last && last < n0 + 32) {                  // wave-uniform: this block holds the new token
        const int rel = last - n0;
        const long long p = a.pos[b];
        const uint16_t* tok_qkv = a.qkv + b * a.qkv_stride;
        uint4 kn[4];
        prep_head_d128(tok_qkv + (long long)(a.q_heads + kvh) * D + lg * 8, true, a.k_w ? a.k_w + lg * 8 : nullptr, a.k_eps,
                       a.cos_t + p * (D / 2) + lg * 8, a.sin_t + p * (D / 2) + lg * 8, kn);
        const uint4 vn = *(const uint4*)(tok_qkv + (long long)(a.q_heads + a.kv_heads + kvh) * D + li * NDV);
        // every row clamped to the new token takes the register copy: its cache slot is not written yet, and
        // whatever it holds (a NaN would survive the zero probability of a masked row) must not reach the MFMAs
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
          if (n0 + 16 * blk + li >= last) {
#pragma unroll
            for (int s = 0; s < KS; ++s) r.k[blk][s] = kn[s];
          }
#pragma unroll
        for (int t = 0; t < 8; ++t)
          if (n0 + 16 * (t >> 2) + 4 * lg + (t & 3) >= last) {
            r.v[t][0] = vn.x, r.v[t][1] = vn.y, r.v[t][2] = vn.z, r.v[t][3] = vn.w;
          }
        // append to the cache for the steps to come (one head block does it; a full cache drops the token)
        if (hblk == 0 && last < a.max_slots) {
          const long long slot = one_page ? slot0 + rel : slot_lookup(last);
          if (li == 0) {
            uint16_t* kd = const_cast<uint16_t*>(a.k_cache) + slot * a.k_stride_slot + (long long)kvh * a.k_stride_h + lg * 8;
#pragma unroll
            for (int s = 0; s < KS; ++s) *(uint4*)(kd + 32 * s) = kn[s];
          }
          if (lg == 0)
            *(uint4*)(const_cast<uint16_t*>(a.v_cache) + slot * a.v_stride_slot + (long long)kvh * a.v_stride_h + li * NDV) = vn;
        }
      }
    }
  };
  auto compute_block = [&](int n0, const KVBlock& r) {
    // ---- S^T = K . Q^T  (lane: head li, tokens 16 blk + 4 lg + i)
    f32x4 st[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
      st[blk] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) st[blk] = F::mfma(__builtin_bit_cast(VT, r.k[blk][s]), qreg[s], st[blk]);
    }
    float sc[8];
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float s = st[t >> 2][t & 3] * qk_scale;
      if (a.soft_cap > 0.f) s = a.soft_cap * LOG2E * tanhf(s);
      const int tok = n0 + 16 * (t >> 2) + 4 * lg + (t & 3);
      s = tok < end ? s : -INFINITY;
      sc[t] = s;
      mx = fmaxf(mx, s);
    }
    mx = xor_max<16>(mx);
    mx = xor_max<32>(mx);
    const float m_new = fmaxf(m_run, mx);   // finite: the block holds at least one valid token
    const float alpha = fast_exp2(m_run - m_new);
    m_run = m_new;
    float psum = 0.f;
    uint32_t pw[4];
#pragma unroll
    for (int t = 0; t < 8; t += 2) {
      const float p0 = fast_exp2(sc[t] - m_new), p1 = fast_exp2(sc[t + 1] - m_new);
      psum += p0 + p1;
      pw[t >> 1] = F::pack2(p0, p1);
    }
    l_run = l_run * alpha + psum;
    const VT pfrag = __builtin_bit_cast(VT, uint4{pw[0], pw[1], pw[2], pw[3]});

    // ---- O^T += V^T . P^T, one dv block per MFMA; V operand by in-register transpose
#pragma unroll
    for (int j = 0; j < NDV; ++j) {
      const uint32_t sel = (j & 1) ? 0x07060302u : 0x05040100u;
      uint32_t vw[4];
#pragma unroll
      for (int p = 0; p < 4; ++p) vw[p] = __builtin_amdgcn_perm(r.v[2 * p + 1][j >> 1], r.v[2 * p][j >> 1], sel);
      f32x4 c = acc[j];
#pragma unroll
      for (int i = 0; i < 4; ++i) c[i] *= alpha;
      acc[j] = F::mfma(__builtin_bit_cast(VT, uint4{vw[0], vw[1], vw[2], vw[3]}), pfrag, c);
    }
  };
  // Software pipeline over two register sets: the loads of block i + 1 are issued before block i is
  // multiplied (D <= 128; at D = 256 a second set would not fit two waves per SIMD).  Batch 64 x 4096 tokens:
  // 190-198 us against 204-211 us without it.  Latency launches (about one workgroup per CU: long-context
  // decode of one sequence cut into many splits) keep the single set - 33.8 against 37.0 us at 1 x 32768.
  constexpr bool DOUBLE = ALLOW_DEEP && D <= 128;
  int n0 = start + wave * 32;
  if (DOUBLE && a.deep) {
    KVBlock ra, rb;
    if (n0 < end) load_block(n0, one_page ? block_slot0(n0) : 0, ra);
    while (n0 < end) {
      if (n0 + 128 < end) load_block(n0 + 128, one_page ? block_slot0(n0 + 128) : 0, rb);
      compute_block(n0, ra);
      n0 += 128;
      if (n0 >= end) break;
      if (n0 + 128 < end) load_block(n0 + 128, one_page ? block_slot0(n0 + 128) : 0, ra);
      compute_block(n0, rb);
      n0 += 128;
    }
  } else {
    for (; n0 < end; n0 += 128) {
      KVBlock r;
      load_block(n0, one_page ? block_slot0(n0) : 0, r);
      compute_block(n0, r);
    }
  }

  // ---- merge the four waves: lane (head li, group lg) holds dv = NDV * (4 lg + i) + j
  if (lds_pages) __syncthreads();        // (the page list shares the accumulator area: every wave is done with it)
  l_run += __shfl_xor(l_run, 16);
  l_run += __shfl_xor(l_run, 32);
  if (lg == 0) {
    s_m[wave * 16 + li] = m_run;
    s_l[wave * 16 + li] = l_run;
  }
#pragma unroll
  for (int j = 0; j < NDV; ++j)
#pragma unroll
    for (int i = 0; i < 4; ++i) s_acc[(wave * 16 + li) * D + NDV * (4 * lg + i) + j] = acc[j][i];
  __syncthreads();

  const long long prow = b * a.p_stride_b + z * a.p_stride_s;     // element offset of this split's rows
  for (int idx = threadIdx.x; idx < 16 * D; idx += 256) {
    const int h = idx / D, c = idx - h * D;
    if (h >= valid_h) continue;
    float M = -INFINITY;
#pragma unroll
    for (int w = 0; w < 4; ++w) M = fmaxf(M, s_m[w * 16 + h]);
    float L = 0.f, o = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float mw = s_m[w * 16 + h];
      const float f = mw == -INFINITY ? 0.f : fast_exp2(mw - M);
      L += s_l[w * 16 + h] * f;
      o += s_acc[(w * 16 + h) * D + c] * f;
    }
    const long long doff = prow + (long long)(h0 + h) * a.p_stride_h;
    float* dst = (float*)a.partial + doff;
    const float val = L > 0.f ? o / L : 0.f;
    const float lse = L > 0.f ? (M + __log2f(L)) * 0.6931471805599453f : -INFINITY;
    if (a.fused == 1 && nsplit == 1) {
      // a single split is the final result: no partial row, no ticket, no second pass
      ((uint16_t*)a.out)[b * a.o_stride_b + (long long)(h0 + h) * a.o_stride_h + c] = to16<IS_BF16>(val);
    } else if (a.p16) {
      // 16-bit partial rows (plain stores; a fused launch publishes them with a release fence before its ticket)
      uint16_t* d16 = (uint16_t*)a.partial + doff;
      d16[c] = to16<IS_BF16>(val);
      if (c == 0) d16[D] = to16<IS_BF16>(lse);
    } else if (a.fused) {
      // read by another workgroup of this launch: write-through stores (they leave this XCD's L2
      // at once), so the hand-off needs no release fence, only a drain before the ticket
      __hip_atomic_store(dst + c, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (c == 0) __hip_atomic_store(dst + D, lse, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      dst[c] = val;
      if (c == 0) dst[D] = lse;
    }
  }
  return nsplit;
}

}  // namespace fdec
}  // namespace tdist
