// Stand-alone reduce-scatter over xGMI:  out[M/w, N] = sum_r x_r[rank*M/w : (rank+1)*M/w, :].
//
// One MI355X node is a full mesh, so the reduce-scatter is ONE hop: every rank publishes its whole
// input in a symmetric buffer, raises a per-call epoch flag on every peer, and then reads "its" row
// segment from all w buffers concurrently (w-1 of them over w-1 different xGMI links), adding in
// fp32.  The reference's ring / 2-D (intra-node scatter + ring reduce + inter-node p2p) pipeline
// (kernels/nvidia/reduce_scatter.py:285-742) exists because NVLink-switch / IB topologies differ;
// its entry points are kept by the Python wrapper (ops/reduce_scatter.py).
//
// Buffers are double-buffered by call parity on the host side: a rank can run at most one call
// ahead of its slowest peer, because finishing call n needs every peer's call-n flag.
#include "tdist/common.hpp"
#include "tdist/call_state.hpp"
#include "tdist/primitives.hpp"
#include "tdist/shmem_device.hpp"

using namespace tdist;

struct RsParams {
  SymmCtx ctx;
  const uint16_t* buf;   // this rank's symmetric input copy [w][seg_elems] (symmetric address)
  int* flags;            // symmetric int32 [w]: flags[s] == epoch  <=>  rank s published its input
  void* out;             // [seg_elems] 16-bit, or fp32 when out_fp32
  int* err;
  long long seg_elems;   // elements per rank segment (multiple of 8)
  int epoch;
  int out_fp32;
  // Device-side call numbering (tdist/call_state.hpp).  With `state` set: `flags` is the base of both halves (half h
  // at flags + h * 2 w: [0, w) "input published", [w, 2 w) "segment read"), half and flag value come from the call
  // number on the device, and the input copy is ONE buffer for every call: the kernel does not end before every peer
  // has read this rank's copy (the last workgroup to finish tells the peers and waits for theirs), so the copy that
  // stages the next call's input - the next thing on the stream - cannot overtake a reader.  The launch is then one
  // kernel node, the same every time.
  int* state;            // {call number, finished workgroups, workgroups done reading}
};

template <bool IS_BF16>
__device__ __forceinline__ float to_f32(uint16_t v) {
  if constexpr (IS_BF16) {
    return __builtin_bit_cast(float, (uint32_t)v << 16);
  } else {
    return (float)__builtin_bit_cast(_Float16, v);
  }
}
template <bool IS_BF16>
__device__ __forceinline__ uint16_t from_f32(float f) {
  if constexpr (IS_BF16) {
    return __builtin_bit_cast(uint16_t, (__bf16)f);
  } else {
    return __builtin_bit_cast(uint16_t, (_Float16)f);
  }
}

template <bool IS_BF16>
__global__ __launch_bounds__(512) void reduce_scatter_kernel(RsParams p) {
  __shared__ int s_call;
  const int w = p.ctx.num_ranks, rank = p.ctx.rank;
  int call = 0;
  if (p.state) {
    call = call_begin(p.state, &s_call);
    p.flags += call_parity(call) * 2 * w;
    p.epoch = call_epoch(call);
  }
  // Publish: the input copy was written by an earlier kernel of this stream, so it is complete;
  // one system-scope release makes it visible to the peers before the flag.
  if (blockIdx.x == 0 && threadIdx.x < w) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_relaxed_system(symm_at(p.ctx, p.flags, (int)threadIdx.x) + rank, p.epoch);
  }
  wait(p.flags, w, SCOPE_SYS, SEM_ACQUIRE, p.epoch, p.err);

  const uint16_t* src[TDIST_MAX_RANKS];
#pragma unroll
  for (int i = 0; i < TDIST_MAX_RANKS; ++i) {
    const int peer = (rank + i) % w;   // own buffer first, then the peers in rotated order
    src[i] = i < w ? symm_at(p.ctx, const_cast<uint16_t*>(p.buf), peer) + (long long)rank * p.seg_elems : nullptr;
  }
  const long long nvec = p.seg_elems / 8;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (long long)gridDim.x * blockDim.x) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < w; ++i) {
      const uint4 q = *(const uint4*)(src[i] + v * 8);
      const uint32_t words[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        acc[2 * k] += to_f32<IS_BF16>((uint16_t)(words[k] & 0xffffu));
        acc[2 * k + 1] += to_f32<IS_BF16>((uint16_t)(words[k] >> 16));
      }
    }
    if (p.out_fp32) {
      float4* o = (float4*)((float*)p.out + v * 8);
      o[0] = float4{acc[0], acc[1], acc[2], acc[3]};
      o[1] = float4{acc[4], acc[5], acc[6], acc[7]};
    } else {
      uint4 r;
      r.x = from_f32<IS_BF16>(acc[0]) | ((uint32_t)from_f32<IS_BF16>(acc[1]) << 16);
      r.y = from_f32<IS_BF16>(acc[2]) | ((uint32_t)from_f32<IS_BF16>(acc[3]) << 16);
      r.z = from_f32<IS_BF16>(acc[4]) | ((uint32_t)from_f32<IS_BF16>(acc[5]) << 16);
      r.w = from_f32<IS_BF16>(acc[6]) | ((uint32_t)from_f32<IS_BF16>(acc[7]) << 16);
      *(uint4*)((uint16_t*)p.out + v * 8) = r;
    }
  }
  if (p.state) {
    // loads of the peers' copies have returned (their values were consumed above); the last workgroup to get here
    // reports "segment read" to every peer and waits until every peer has read this rank's copy
    __syncthreads();
    if (threadIdx.x == 0) s_call = atom_add_relaxed_agent(p.state + 2, 1);
    __syncthreads();
    if (s_call == (int)gridDim.x - 1) {
      if ((int)threadIdx.x < w) store_relaxed_system(symm_at(p.ctx, p.flags, (int)threadIdx.x) + w + rank, p.epoch);
      wait(p.flags + w, w, SCOPE_SYS, SEM_RELAXED, p.epoch, p.err);
      if (threadIdx.x == 0) store_relaxed_agent(p.state + 2, 0);
    }
    call_end(p.state, call, p.state + 1, nullptr, 0);
  }
}

// dtype: 0 = bf16, 1 = fp16.  `buf` must hold this rank's full input ([w * seg_elems]) before the
// launch (written earlier on the same stream).
// `state` != null: int32 [3], zero when created (see RsParams); `flags` is then int32 [4 w] and `epoch` is ignored.
TDIST_API int tdist_reduce_scatter(const SymmCtx* ctx, const void* buf, int* flags, void* out, int* err,
                                   long long seg_elems, int epoch, int out_fp32, int dtype, int num_ctas, void* stream,
                                   int* state) {
  TDIST_REQUIRE(seg_elems > 0 && seg_elems % 8 == 0, "segment must be a positive multiple of 8 elements");
  TDIST_REQUIRE(ctx->num_ranks >= 1 && ctx->num_ranks <= TDIST_MAX_RANKS, "bad world size");
  TDIST_REQUIRE((((uintptr_t)buf | (uintptr_t)out) & 15) == 0, "buffers must be 16-byte aligned");
  RsParams p;
  p.ctx = *ctx;
  p.buf = (const uint16_t*)buf;
  p.flags = flags;
  p.out = out;
  p.err = err;
  p.seg_elems = seg_elems;
  p.epoch = epoch;
  p.out_fp32 = out_fp32;
  p.state = state;
  long long want = (seg_elems / 8 + 511) / 512;
  int grid = (int)(want < 1 ? 1 : (want > 1024 ? 1024 : want));
  if (num_ctas > 0 && grid > num_ctas) grid = num_ctas;
  if (dtype == 0)
    hipLaunchKernelGGL(reduce_scatter_kernel<true>, dim3(grid), dim3(512), 0, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(reduce_scatter_kernel<false>, dim3(grid), dim3(512), 0, (hipStream_t)stream, p);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
