// GQA decode attention over a paged KV cache for gfx950, split-KV with LSE-weighted merges.
//
// Role in the reference: kernels/nvidia/flash_decode.py - kernel_gqa_fwd_batch_decode_split_kv
// (:112-239), the three combine kernels (:288-532) and the persistent fused kernel (:577-760).
//
// Design (decode is bound by streaming K/V once from HBM, so nothing is staged through LDS: a wave-private
// direct-to-LDS ring of two 16 KiB slots per wave, one block ahead, was measured at 472 us against 290 us for
// register loads at batch 64 x 4096 tokens - it halves the occupancy and every lane only reads back what it
// requested):
//   * One workgroup (4 waves) per (batch, kv head x block of 16 query heads, KV split).  The 16
//     query heads that share a KV head are the 16 columns of a 16x16x32 MFMA.
//   * S^T[token, head] = K . Q^T : the K operand of the MFMA wants, per lane, 8 consecutive
//     head-dim elements of ONE token - which is a plain 16-byte global load of the cache row.
//   * O^T[dv, head]   = V^T . P^T : the contraction runs over tokens.  The order of the 32
//     tokens inside the contraction is free as long as both operands agree, so the P operand
//     is taken exactly as the S^T accumulators fall (no cross-lane movement) and the V operand
//     is built from eight 16-byte row loads followed by an in-register 8x8 transpose of 16-bit
//     values (v_perm_b32).  MFMA column c of dv-block j is dv = (D/16)*c + j.
//   * Each wave walks its own 32-token blocks with private online-softmax state; the four
//     states are merged through LDS once at the end.
//   * "fused" launches merge the splits in the last workgroup to finish a (batch, head block)
//     (atomic ticket), which replaces the reference's grid barrier + second phase.
#include <cstdlib>

#include "tdist/common.hpp"
#include "tdist/flash_decode_device.hpp"
#include "tdist/gemm_gfx950.hpp"

namespace {

using namespace tdist::fdec;

// LAT: the latency form for launches of at most two workgroups per CU (long-context decode of a few sequences cut
// into many splits) - the single-register-set loop under a 256-register cap, so that two workgroups share a CU and
// twice as many loads are in flight; the throughput form keeps the second register set and a CU to itself.
template <bool IS_BF16, int D, bool PREP = false, bool LAT = false>
__global__ __launch_bounds__(256, (LAT ? 2 : 1)) void decode_split_kernel(DecodeArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __shared__ int s_last;
  const int b = blockIdx.x;
  const int wave = threadIdx.x >> 6;
  const int group = a.q_heads / a.kv_heads;
  const int hblocks = (group + 15) / 16;
  const int kvh = blockIdx.y / hblocks;
  const int hblk = blockIdx.y - kvh * hblocks;
  const int h0 = kvh * group + hblk * 16;
  const int valid_h = min(16, group - hblk * 16);
  const int nsplit = decode_split_body<IS_BF16, D, !LAT, PREP>(a, b, (int)blockIdx.y, (int)blockIdx.z, smem);

  if (!a.fused || (a.fused == 1 && nsplit == 1) || (int)blockIdx.z >= nsplit) return;
  if (a.p16) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");   // 16-bit rows went out as plain stores
  // ---- last workgroup of this (batch, head block) merges the splits.  The partial rows were
  // stored write-through: every wave drains its stores, the workgroup barriers, one lane takes a
  // ticket; the workgroup that draws the last ticket drops its stale L1 lines with ONE acquire
  // and reads the rows with plain loads.
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    int* t = a.tickets + b * gridDim.y + blockIdx.y;
    const int prev = __hip_atomic_fetch_add(t, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    s_last = prev == nsplit - 1;
    if (s_last) __hip_atomic_store(t, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed for the next launch
  }
  __syncthreads();
  if (!s_last) return;
  if (wave == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
  // head_dim 128, fp32 rows: one wave per head, every row of the head loaded in one round trip
  if constexpr (D == 128) {
    if (!a.p16) {
      for (int h = wave; h < valid_h; h += 4) {
        const long long poff = b * a.p_stride_b + (long long)(h0 + h) * a.p_stride_h;
        char* o = (char*)a.out + (b * a.o_stride_b + (long long)(h0 + h) * a.o_stride_h) * (a.fused == 2 ? 4 : 2);
        merge_rows_wave_d128<IS_BF16>((const float*)a.partial + poff, a.p_stride_s, nsplit, nullptr, 0, o, a.fused,
                                      (int)(threadIdx.x & 63));
      }
      return;
    }
  }
  // 16 threads per head, all heads of the block at once
  {
    const int h = threadIdx.x >> 4, sub = threadIdx.x & 15;
    if (h < valid_h) {
      const long long poff = b * a.p_stride_b + (long long)(h0 + h) * a.p_stride_h;
      char* o = (char*)a.out + (b * a.o_stride_b + (long long)(h0 + h) * a.o_stride_h) * (a.fused == 2 ? 4 : 2);
      if (a.p16)
        merge_rows<IS_BF16, true>((const uint16_t*)a.partial + poff, a.p_stride_s, nsplit, D, nullptr, 0, o, a.fused, sub, 16);
      else
        merge_rows<IS_BF16, false>((const float*)a.partial + poff, a.p_stride_s, nsplit, D, nullptr, 0, o, a.fused, sub, 16);
    }
  }
}

// Generic LSE-weighted merge: partial rows [n, batch, heads, D + 1] (given by strides) ->
// out [batch, heads, D] (16-bit) or [batch, heads, D + 1] (fp32 with LSE).  With `lens`
// ([n, batch] int32) items of length 0 are skipped (ranks without KV for a sequence).  P16: 16-bit partials.
template <bool IS_BF16, bool P16>
__global__ __launch_bounds__(64) void decode_combine_kernel(const void* part, long long p_stride_n, long long p_stride_b,
                                                            long long p_stride_h, int n, int D, const int* lens,
                                                            int lens_stride, void* out, long long o_stride_b,
                                                            long long o_stride_h, int out_mode) {
  const int b = blockIdx.x, h = blockIdx.y;
  const char* p = (const char*)part + (b * p_stride_b + h * p_stride_h) * (P16 ? 2 : 4);
  char* o = (char*)out + (b * o_stride_b + h * o_stride_h) * (out_mode == 2 ? 4 : 2);
  if constexpr (!P16) {
    if (D == 128) {
      merge_rows_wave_d128<IS_BF16>((const float*)p, p_stride_n, n, lens ? lens + b : nullptr, lens_stride, o, out_mode,
                                    (int)threadIdx.x);
      return;
    }
  }
  merge_rows<IS_BF16, P16>(p, p_stride_n, n, D, lens ? lens + b : nullptr, lens_stride, o, out_mode, threadIdx.x, 64);
}

template <bool IS_BF16, int D, bool PREP = false>
hipError_t launch_split(const DecodeArgs& a, int batch, hipStream_t st) {
  const int group = a.q_heads / a.kv_heads;
  const int hblocks = (group + 15) / 16;
  const size_t lds = (128 + 64 * D) * sizeof(float);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute((const void*)decode_split_kernel<IS_BF16, D, PREP, false>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    if constexpr (D <= 128) {
      e = hipFuncSetAttribute((const void*)decode_split_kernel<IS_BF16, D, PREP, true>,
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return e;
    }
    attr_done = true;
  }
  const dim3 grid(batch, a.kv_heads * hblocks, a.splits);
  static const int lat_knob = getenv("TDIST_DECODE_LAT") ? atoi(getenv("TDIST_DECODE_LAT")) : -1;    // tuning knob
  bool lat = !a.deep;
  if (lat_knob == 0 || lat_knob == 1) lat = lat_knob == 1 && !a.deep;
  if constexpr (D <= 128) {
    if (lat) {
      hipLaunchKernelGGL((decode_split_kernel<IS_BF16, D, PREP, true>), grid, dim3(256), lds, st, a);
      return hipGetLastError();
    }
  }
  hipLaunchKernelGGL((decode_split_kernel<IS_BF16, D, PREP, false>), grid, dim3(256), lds, st, a);
  return hipGetLastError();
}

// Two register sets (the loads of block i + 1 issued before block i is multiplied) pay once a CU holds more than one
// round of workgroups; TDIST_DECODE_DEEP = 0 / 1 forces the choice (tuning knob).
int deep_choice(long long workgroups, int num_cus) {
  static const int forced = getenv("TDIST_DECODE_DEEP") ? atoi(getenv("TDIST_DECODE_DEEP")) : -1;
  if (forced == 0 || forced == 1) return forced;
  return workgroups > 2ll * num_cus ? 1 : 0;
}

int num_cus_cached() {
  static int num_cus = 0;
  if (!num_cus) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      num_cus = 256;
  }
  return num_cus;
}

}  // namespace

// dtype: 0 = bf16, 1 = fp16; + 16: the partial rows are stored in that 16-bit type instead of fp32.
// fused: 0 partials only / 1 merged 16-bit / 2 merged fp32 + LSE / 3 merged 16-bit + LSE; + 4: `splits` was sized
// from the cache capacity and every sequence uses as many of them as its own length calls for.
TDIST_API int tdist_flash_decode_split(const void* q, const void* k_cache, const void* v_cache, const int* block_table,
                                       const int* kv_len, void* partial, void* out, int* tickets, int batch,
                                       int q_heads, int kv_heads, int head_dim, int splits, int page_size,
                                       int table_stride, long long q_stride_b, long long q_stride_h,
                                       long long k_stride_slot, long long k_stride_h, long long v_stride_slot,
                                       long long v_stride_h, long long o_stride_b, long long o_stride_h, float sm_scale,
                                       float soft_cap, int fused, int dtype, void* stream) {
  TDIST_REQUIRE(batch > 0 && q_heads > 0 && kv_heads > 0 && q_heads % kv_heads == 0, "bad head counts");
  TDIST_REQUIRE(head_dim == 64 || head_dim == 128 || head_dim == 256, "head_dim must be 64, 128 or 256");
  TDIST_REQUIRE(splits >= 1 && page_size >= 1, "bad splits / page size");
  TDIST_REQUIRE(q_stride_h % 8 == 0 && q_stride_b % 8 == 0 && k_stride_slot % 8 == 0 && k_stride_h % 8 == 0 &&
                    v_stride_slot % 8 == 0 && v_stride_h % 8 == 0,
                "q/k/v strides must keep 16-byte alignment");
  const int adaptive = (fused & 4) ? 1 : 0;
  fused &= ~4;
  TDIST_REQUIRE(fused == 0 || (out && tickets), "fused merge needs out and tickets");
  DecodeArgs a;
  a.adaptive = adaptive;
  a.q = (const uint16_t*)q, a.k_cache = (const uint16_t*)k_cache, a.v_cache = (const uint16_t*)v_cache;
  a.block_table = block_table, a.kv_len = kv_len, a.partial = partial, a.out = out, a.tickets = tickets;
  a.q_stride_b = q_stride_b, a.q_stride_h = q_stride_h;
  a.k_stride_slot = k_stride_slot, a.k_stride_h = k_stride_h;
  a.v_stride_slot = v_stride_slot, a.v_stride_h = v_stride_h;
  a.p_stride_s = head_dim + 1, a.p_stride_h = (long long)splits * (head_dim + 1),
  a.p_stride_b = (long long)q_heads * splits * (head_dim + 1);
  a.o_stride_b = o_stride_b, a.o_stride_h = o_stride_h;
  a.table_stride = table_stride, a.page_size = page_size;
  a.page_shift = (page_size & (page_size - 1)) == 0 ? __builtin_ctz(page_size) : -1;
  a.q_heads = q_heads, a.kv_heads = kv_heads, a.splits = splits, a.fused = fused;
  a.p16 = (dtype & 16) ? 1 : 0;
  dtype &= 15;
  TDIST_REQUIRE(dtype == 0 || dtype == 1, "dtype must be 0 (bf16) or 1 (fp16), + 16 for 16-bit partial rows");
  TDIST_REQUIRE(fused >= 0 && fused <= 3, "fused must be 0..3");
  a.sm_scale = sm_scale, a.soft_cap = soft_cap;
  {
    const int hblocks = (q_heads / kv_heads + 15) / 16;
    a.deep = deep_choice((long long)batch * kv_heads * hblocks * splits, num_cus_cached());
  }
  hipStream_t st = (hipStream_t)stream;
  const bool bf = dtype == 0;
  hipError_t e;
  if (head_dim == 64) e = bf ? launch_split<true, 64>(a, batch, st) : launch_split<false, 64>(a, batch, st);
  else if (head_dim == 128) e = bf ? launch_split<true, 128>(a, batch, st) : launch_split<false, 128>(a, batch, st);
  else e = bf ? launch_split<true, 256>(a, batch, st) : launch_split<false, 256>(a, batch, st);
  TDIST_HIP_TRY(e);
  return 0;
}

// A decode step straight from the QKV projection (bf16, head_dim 128, one new token per sequence; the cache of
// sequence b is the contiguous run of `max_len` slots starting at slot b * max_len).  One launch does what
// tdist_qkv_prep + tdist_flash_decode_split(fused = 1) do in two: per-head RMSNorm (q_w / k_w, may be null) and
// neox rotation of the q and k heads at pos[b], K / V of the new token appended at slot kv_offset[b],
// total_out[b] = kv_offset[b] + 1, attention over those total_out[b] tokens; `fused` 1: splits merged by the last
// workgroup into `out`, 0: fp32 partial rows [batch, q_heads, splits, 129] for tdist_flash_decode_combine.
TDIST_API int tdist_flash_decode_prep(const void* qkv, long long qkv_stride, const void* q_w, const void* k_w,
                                      float q_eps, float k_eps, const long long* pos, const float* cos_t,
                                      const float* sin_t, void* k_cache, void* v_cache, const int* kv_offset,
                                      int* total_out, const int* seq_table, void* partial, void* out, int* tickets,
                                      int batch, int q_heads, int kv_heads, long long max_len, int splits,
                                      long long o_stride_b, long long o_stride_h, float sm_scale, int adaptive,
                                      int fused, void* stream) {
  constexpr int D = 128;
  TDIST_REQUIRE(batch > 0 && q_heads > 0 && kv_heads > 0 && q_heads % kv_heads == 0, "bad head counts");
  TDIST_REQUIRE(splits >= 1 && max_len >= 1 && max_len < (1ll << 30), "bad splits / cache length");
  TDIST_REQUIRE(qkv && pos && cos_t && sin_t && k_cache && v_cache && kv_offset && total_out && seq_table && partial,
                "null argument");
  TDIST_REQUIRE(fused == 0 || (fused == 1 && out && tickets), "fused must be 0 or 1; the fused merge needs out and tickets");
  TDIST_REQUIRE(qkv_stride % 8 == 0 && qkv_stride >= (long long)(q_heads + 2 * kv_heads) * D &&
                    ((uintptr_t)qkv & 15) == 0 && ((uintptr_t)k_cache & 15) == 0 && ((uintptr_t)v_cache & 15) == 0 &&
                    ((uintptr_t)cos_t & 15) == 0 && ((uintptr_t)sin_t & 15) == 0 &&
                    (!q_w || ((uintptr_t)q_w & 15) == 0) && (!k_w || ((uintptr_t)k_w & 15) == 0),
                "qkv rows, caches, rotation tables and norm weights must keep 16-byte alignment");
  DecodeArgs a;
  a.q = nullptr, a.k_cache = (const uint16_t*)k_cache, a.v_cache = (const uint16_t*)v_cache;
  a.block_table = seq_table, a.kv_len = nullptr, a.partial = partial, a.out = out, a.tickets = tickets;
  a.q_stride_b = 0, a.q_stride_h = 0;
  a.k_stride_slot = a.v_stride_slot = (long long)kv_heads * D;
  a.k_stride_h = a.v_stride_h = D;
  a.p_stride_s = D + 1, a.p_stride_h = (long long)splits * (D + 1), a.p_stride_b = (long long)q_heads * splits * (D + 1);
  a.o_stride_b = o_stride_b, a.o_stride_h = o_stride_h;
  // one page per sequence; the page arithmetic only needs a power of two that covers it
  a.table_stride = 1;
  a.page_size = (int)max_len;
  a.page_shift = (max_len & (max_len - 1)) == 0 ? __builtin_ctzll(max_len) : -1;
  a.q_heads = q_heads, a.kv_heads = kv_heads, a.splits = splits, a.fused = fused, a.p16 = 0;
  a.sm_scale = sm_scale, a.soft_cap = 0.f;
  a.deep = deep_choice((long long)batch * kv_heads * ((q_heads / kv_heads + 15) / 16) * splits, num_cus_cached());
  a.qkv = (const uint16_t*)qkv, a.qkv_stride = qkv_stride;
  a.q_w = (const uint16_t*)q_w, a.k_w = (const uint16_t*)k_w, a.q_eps = q_eps, a.k_eps = k_eps;
  a.pos = pos, a.cos_t = cos_t, a.sin_t = sin_t, a.kv_offset = kv_offset, a.total_out = total_out, a.max_slots = max_len;
  a.adaptive = adaptive ? 1 : 0;
  TDIST_HIP_TRY((launch_split<true, D, true>(a, batch, (hipStream_t)stream)));
  return 0;
}

// dtype as for tdist_flash_decode_split (+ 16: 16-bit partial rows); out_mode 1 / 2 / 3 as `fused` there.
TDIST_API int tdist_flash_decode_combine(const void* partial, long long p_stride_n, long long p_stride_b,
                                         long long p_stride_h, int n, int batch, int heads, int head_dim,
                                         const int* lens, int lens_stride, void* out, long long o_stride_b,
                                         long long o_stride_h, int out_mode, int dtype, void* stream) {
  TDIST_REQUIRE(n >= 1 && batch > 0 && heads > 0 && head_dim > 0, "bad combine shape");
  TDIST_REQUIRE(out_mode >= 1 && out_mode <= 3, "out_mode must be 1 (16-bit), 2 (fp32 + LSE) or 3 (16-bit + LSE)");
  hipStream_t st = (hipStream_t)stream;
  const bool p16 = (dtype & 16) != 0;
  dtype &= 15;
  TDIST_REQUIRE(dtype == 0 || dtype == 1, "dtype must be 0 (bf16) or 1 (fp16), + 16 for 16-bit partial rows");
  auto launch = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(batch, heads), dim3(64), 0, st, partial, p_stride_n, p_stride_b, p_stride_h, n,
                       head_dim, lens, lens_stride, out, o_stride_b, o_stride_h, out_mode);
  };
  if (dtype == 0) {
    if (p16) launch(decode_combine_kernel<true, true>); else launch(decode_combine_kernel<true, false>);
  } else {
    if (p16) launch(decode_combine_kernel<false, true>); else launch(decode_combine_kernel<false, false>);
  }
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
