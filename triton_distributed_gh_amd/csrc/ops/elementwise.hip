// Memory-bound helper kernels, all with 16-byte vector accesses on bf16 data:
//   rms_norm, rope (neox style, cos/sin table), silu(gate) * up, residual add,
//   fp8 (OCP e4m3fn) per-128-group quantisation, paged / contiguous KV-cache append.
//
// Role in the reference: layers/amd/tp_attn.py:52-168 (layer_norm, rope kernel),
// tutorials/04-deepseek-infer-all2all.py:334-427 (fp8 quant), mega_triton_kernel/kernels
// norm.py / activation.py / elementwise.py (the same math as mega-kernel tasks).
#include <hip/hip_fp8.h>

#include "tdist/common.hpp"

namespace {

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    f[2 * e] = __builtin_bit_cast(float, x[e] << 16);
    f[2 * e + 1] = __builtin_bit_cast(float, x[e] & 0xffff0000u);
  }
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  uint32_t x[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    b2 v = {(__bf16)f[2 * e], (__bf16)f[2 * e + 1]};
    x[e] = __builtin_bit_cast(uint32_t, v);
  }
  return uint4{x[0], x[1], x[2], x[3]};
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// one workgroup (256 threads) per row; hidden % 8 == 0; optional residual add fused in front:
// if res != null: x = x + res (written back to res_out), y = rmsnorm(x) * w
__global__ __launch_bounds__(256) void rms_norm_kernel(const uint16_t* __restrict__ x, const uint16_t* __restrict__ res,
                                                       const uint16_t* __restrict__ w, uint16_t* __restrict__ y,
                                                       uint16_t* __restrict__ res_out, int hidden, float eps,
                                                       long long x_stride, long long y_stride) {
  __shared__ float red[4];
  const long long row = blockIdx.x;
  const uint16_t* xr = x + row * x_stride;
  float ss = 0.f;
  for (int c = threadIdx.x * 8; c < hidden; c += 256 * 8) {
    float f[8];
    unpack8(*(const uint4*)(xr + c), f);
    if (res) {
      float g[8];
      unpack8(*(const uint4*)(res + row * hidden + c), g);
#pragma unroll
      for (int i = 0; i < 8; ++i) f[i] += g[i];
      *(uint4*)(res_out + row * hidden + c) = pack8(f);
      // use the rounded value so norm(x) matches a two-kernel implementation bit for bit
      unpack8(*(const uint4*)(res_out + row * hidden + c), f);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) ss += f[i] * f[i];
  }
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float tot = red[0] + red[1] + red[2] + red[3];
  const float inv = rsqrtf(tot / hidden + eps);
  const uint16_t* src = res ? res_out + row * hidden : xr;
  for (int c = threadIdx.x * 8; c < hidden; c += 256 * 8) {
    float f[8], g[8];
    unpack8(*(const uint4*)(src + c), f);
    unpack8(*(const uint4*)(w + c), g);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = f[i] * inv * g[i];
    *(uint4*)(y + row * y_stride + c) = pack8(f);
  }
}

// rotary embedding, neox layout: out[..., :D/2] = x1*cos - x2*sin ; out[..., D/2:] = x2*cos + x1*sin
// x: [tokens, heads, D] (row stride given), positions: [tokens] int32/int64, cos/sin: [max_pos, D/2] f32
__global__ __launch_bounds__(256) void rope_kernel(uint16_t* __restrict__ x, const long long* __restrict__ pos,
                                                   const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                   int heads, int D, long long token_stride, long long head_stride) {
  const long long tok = blockIdx.x;
  const int half = D / 2;
  const long long p = pos[tok];
  const int per_head = half / 8;   // 8-element vectors of the first half
  for (int i = threadIdx.x; i < heads * per_head; i += blockDim.x) {
    const int h = i / per_head, c = (i % per_head) * 8;
    uint16_t* base = x + tok * token_stride + h * head_stride;
    float a[8], b[8], co[8], si[8];
    unpack8(*(const uint4*)(base + c), a);
    unpack8(*(const uint4*)(base + half + c), b);
    *(float4*)co = *(const float4*)(cos_t + p * half + c);
    *(float4*)(co + 4) = *(const float4*)(cos_t + p * half + c + 4);
    *(float4*)si = *(const float4*)(sin_t + p * half + c);
    *(float4*)(si + 4) = *(const float4*)(sin_t + p * half + c + 4);
    float o1[8], o2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      o1[k] = a[k] * co[k] - b[k] * si[k];
      o2[k] = b[k] * co[k] + a[k] * si[k];
    }
    *(uint4*)(base + c) = pack8(o1);
    *(uint4*)(base + half + c) = pack8(o2);
  }
}

// Everything between the QKV projection and the attention kernel in ONE launch (decode runs it once
// per layer, where each separate launch costs more than the work): per token and head
//   q heads : [rmsnorm(head) * q_w] -> rope -> q_out [tokens, Hq, D]
//   k heads : [rmsnorm(head) * k_w] -> rope -> k_cache[b, kv_offset[b] + s, head]
//   v heads :                                 v_cache[b, kv_offset[b] + s, head]
// and total_out[b] = kv_offset[b] + S (the length the attention kernel has to cover).  One wave per
// head, four heads per workgroup; lane i owns elements i and i + D/2 of the head (the neox rotation
// pair).  The norm result is rounded to 16 bits before the rotation, like the separate kernels do.
// Role in the reference: layers/amd/tp_attn.py:52-168 (norm + rope) and the mega kernel's
// qk_norm_rope_update_kvcache task (mega_triton_kernel/kernels/norm.py:31-190).
__global__ __launch_bounds__(256) void qkv_prep_kernel(const uint16_t* __restrict__ qkv, long long qkv_stride,
                                                       const uint16_t* __restrict__ q_w, const uint16_t* __restrict__ k_w,
                                                       float q_eps, float k_eps, const long long* __restrict__ pos,
                                                       const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                       uint16_t* __restrict__ q_out, uint16_t* __restrict__ k_cache,
                                                       uint16_t* __restrict__ v_cache, const int* __restrict__ kv_offset,
                                                       int* __restrict__ total_out, int S, int Hq, int Hkv, int D,
                                                       long long max_len) {
  const long long tok = blockIdx.x;
  const int b = (int)(tok / S), sidx = (int)(tok % S);
  const int head = blockIdx.y * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int off = kv_offset[b];
  if (blockIdx.y == 0 && threadIdx.x == 0 && sidx == 0) total_out[b] = off + S;
  if (head >= Hq + 2 * Hkv) return;
  const long long slot = off + sidx;
  const bool slot_ok = slot >= 0 && slot < max_len;        // a full cache drops the token instead of writing outside
  const uint16_t* src = qkv + tok * qkv_stride + (long long)head * D;
  if (head >= Hq + Hkv) {
    if (!slot_ok) return;
    uint16_t* dst = v_cache + (((long long)b * max_len + slot) * Hkv + (head - Hq - Hkv)) * D;
    for (int c = lane * 8; c < D; c += 512) *(uint4*)(dst + c) = *(const uint4*)(src + c);
    return;
  }
  const bool is_q = head < Hq;
  const int half = D >> 1;
  const uint16_t* w = is_q ? q_w : k_w;
  float a[2] = {0.f, 0.f}, c[2] = {0.f, 0.f};
  float ss = 0.f;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = lane + 64 * j;
    if (i < half) {
      a[j] = __builtin_bit_cast(float, (uint32_t)src[i] << 16);
      c[j] = __builtin_bit_cast(float, (uint32_t)src[i + half] << 16);
      ss += a[j] * a[j] + c[j] * c[j];
    }
  }
  if (w) {
    ss = wave_sum(ss);
    const float inv = rsqrtf(ss / D + (is_q ? q_eps : k_eps));
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = lane + 64 * j;
      if (i < half) {
        const float w1 = __builtin_bit_cast(float, (uint32_t)w[i] << 16);
        const float w2 = __builtin_bit_cast(float, (uint32_t)w[i + half] << 16);
        a[j] = (float)(__bf16)(a[j] * inv * w1);
        c[j] = (float)(__bf16)(c[j] * inv * w2);
      }
    }
  }
  if (!is_q && !slot_ok) return;
  uint16_t* dst = is_q ? q_out + (tok * Hq + head) * D
                       : k_cache + (((long long)b * max_len + slot) * Hkv + (head - Hq)) * D;
  const long long p = pos[tok];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = lane + 64 * j;
    if (i < half) {
      const float co = cos_t[p * half + i], si = sin_t[p * half + i];
      const __bf16 o1 = (__bf16)(a[j] * co - c[j] * si), o2 = (__bf16)(c[j] * co + a[j] * si);
      dst[i] = __builtin_bit_cast(uint16_t, o1);
      dst[i + half] = __builtin_bit_cast(uint16_t, o2);
    }
  }
}

// y[row, c] = silu(gu[row, c]) * gu[row, inter + c]   (gate and up halves of one [*, 2*inter] tensor)
__global__ __launch_bounds__(256) void silu_mul_kernel(const uint16_t* __restrict__ gu, uint16_t* __restrict__ y,
                                                       long long rows, int inter) {
  const long long nvec = rows * (inter / 8);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    const long long row = i / (inter / 8);
    const int c = (int)(i % (inter / 8)) * 8;
    float g[8], u[8];
    unpack8(*(const uint4*)(gu + row * 2 * inter + c), g);
    unpack8(*(const uint4*)(gu + row * 2 * inter + inter + c), u);
#pragma unroll
    for (int k = 0; k < 8; ++k) g[k] = g[k] / (1.f + __expf(-g[k])) * u[k];
    *(uint4*)(y + row * inter + c) = pack8(g);
  }
}

__global__ __launch_bounds__(256) void add_kernel(const uint16_t* __restrict__ a, const uint16_t* __restrict__ b,
                                                  uint16_t* __restrict__ y, long long nvec) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (long long)gridDim.x * blockDim.x) {
    float f[8], g[8];
    unpack8(*(const uint4*)(a + i * 8), f);
    unpack8(*(const uint4*)(b + i * 8), g);
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] += g[k];
    *(uint4*)(y + i * 8) = pack8(f);
  }
}

// one wave per 128-element group: q = x / scale, scale = max|x| / 448 (OCP e4m3fn)
__global__ __launch_bounds__(256) void fp8_group_quant_kernel(const uint16_t* __restrict__ x, uint8_t* __restrict__ q,
                                                              float* __restrict__ scale, long long groups) {
  const long long g = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= groups) return;
  const int lane = threadIdx.x & 63;
  const uint32_t pair = *(const uint32_t*)(x + g * 128 + lane * 2);
  const float a = __builtin_bit_cast(float, pair << 16), b = __builtin_bit_cast(float, pair & 0xffff0000u);
  const float m = wave_max(fmaxf(fabsf(a), fabsf(b)));
  const float s = fmaxf(m, 1e-12f) / 448.f;
  if (lane == 0) scale[g] = s;
  const float inv = 1.f / s;
  __hip_fp8_e4m3 qa(a * inv), qb(b * inv);
  uint16_t packed = (uint16_t)qa.__x | ((uint16_t)qb.__x << 8);
  *(uint16_t*)(q + g * 128 + lane * 2) = packed;
}

// KV-cache append: cache[slot[t], head, :] = src[t, head, :]; row = kv_heads * D elements
__global__ __launch_bounds__(256) void kv_append_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ cache,
                                                        const long long* __restrict__ slots, int row_elems,
                                                        long long src_stride) {
  const long long t = blockIdx.x;
  const long long s = slots[t];
  for (int c = threadIdx.x * 8; c < row_elems; c += blockDim.x * 8)
    *(uint4*)(cache + s * row_elems + c) = *(const uint4*)(src + t * src_stride + c);
}

}  // namespace

TDIST_API int tdist_rms_norm(const void* x, const void* res, const void* w, void* y, void* res_out, long long rows,
                             int hidden, float eps, long long x_stride, long long y_stride, void* stream) {
  TDIST_REQUIRE(hidden % 8 == 0, "hidden must be a multiple of 8");
  if (rows == 0) return 0;
  hipLaunchKernelGGL(rms_norm_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)x,
                     (const uint16_t*)res, (const uint16_t*)w, (uint16_t*)y, (uint16_t*)res_out, hidden, eps, x_stride,
                     y_stride);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
TDIST_API int tdist_rope(void* x, const long long* pos, const float* cos_t, const float* sin_t, long long tokens,
                         int heads, int D, long long token_stride, long long head_stride, void* stream) {
  TDIST_REQUIRE(D % 16 == 0, "head_dim must be a multiple of 16");
  if (tokens == 0) return 0;
  hipLaunchKernelGGL(rope_kernel, dim3((unsigned)tokens), dim3(256), 0, (hipStream_t)stream, (uint16_t*)x, pos, cos_t,
                     sin_t, heads, D, token_stride, head_stride);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
TDIST_API int tdist_qkv_prep(const void* qkv, long long qkv_stride, const void* q_w, const void* k_w, float q_eps,
                             float k_eps, const long long* pos, const float* cos_t, const float* sin_t, void* q_out,
                             void* k_cache, void* v_cache, const int* kv_offset, int* total_out, long long tokens, int S,
                             int Hq, int Hkv, int D, long long max_len, void* stream) {
  TDIST_REQUIRE(D == 64 || D == 128 || D == 256, "head_dim must be 64, 128 or 256");
  TDIST_REQUIRE(S > 0 && Hq > 0 && Hkv > 0 && max_len > 0 && qkv_stride % 8 == 0, "bad shape");
  TDIST_REQUIRE((((uintptr_t)qkv | (uintptr_t)q_out | (uintptr_t)k_cache | (uintptr_t)v_cache) & 15) == 0,
                "tensors must be 16-byte aligned");
  if (tokens == 0) return 0;
  const int heads = Hq + 2 * Hkv;
  hipLaunchKernelGGL(qkv_prep_kernel, dim3((unsigned)tokens, (unsigned)((heads + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, (const uint16_t*)qkv, qkv_stride, (const uint16_t*)q_w, (const uint16_t*)k_w,
                     q_eps, k_eps, pos, cos_t, sin_t, (uint16_t*)q_out, (uint16_t*)k_cache, (uint16_t*)v_cache,
                     kv_offset, total_out, S, Hq, Hkv, D, max_len);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
TDIST_API int tdist_silu_mul(const void* gu, void* y, long long rows, int inter, void* stream) {
  TDIST_REQUIRE(inter % 8 == 0, "intermediate size must be a multiple of 8");
  if (rows == 0) return 0;
  const long long nvec = rows * (inter / 8);
  const int grid = (int)((nvec + 255) / 256 < 2048 ? (nvec + 255) / 256 : 2048);
  hipLaunchKernelGGL(silu_mul_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)gu, (uint16_t*)y,
                     rows, inter);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
TDIST_API int tdist_add(const void* a, const void* b, void* y, long long n, void* stream) {
  TDIST_REQUIRE(n % 8 == 0, "element count must be a multiple of 8");
  if (n == 0) return 0;
  const long long nvec = n / 8;
  const int grid = (int)((nvec + 255) / 256 < 2048 ? (nvec + 255) / 256 : 2048);
  hipLaunchKernelGGL(add_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)a, (const uint16_t*)b,
                     (uint16_t*)y, nvec);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
TDIST_API int tdist_fp8_group_quant(const void* x, void* q, float* scale, long long groups, void* stream) {
  if (groups == 0) return 0;
  hipLaunchKernelGGL(fp8_group_quant_kernel, dim3((unsigned)((groups + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)x, (uint8_t*)q, scale, groups);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
TDIST_API int tdist_kv_append(const void* src, void* cache, const long long* slots, long long tokens, int row_elems,
                              long long src_stride, void* stream) {
  TDIST_REQUIRE(row_elems % 8 == 0, "row must be a multiple of 8 elements");
  if (tokens == 0) return 0;
  hipLaunchKernelGGL(kv_append_kernel, dim3((unsigned)tokens), dim3(256), 0, (hipStream_t)stream, (const uint16_t*)src,
                     (uint16_t*)cache, slots, row_elems, src_stride);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
