// Persistent "mega kernel": one launch interprets a whole decode step as a queue of tile tasks.
//
// Role in the reference: python/triton_dist/mega_triton_kernel (generated MEGA_TRITON_KERNEL,
// core/code_generator.py:31-89; device context kernels/task_context.py:31-138; task kernels
// kernels/*.py).  The reference GENERATES Triton source with an if/elif chain per task type; here the
// interpreter is one hand-written HIP kernel with a switch over task types.
//
// Execution model
//   * grid = number of work queues (one workgroup of 256 threads per queue, all co-resident);
//   * a queue is an array of 16-word task records [type, op, tile, ndeps, 3 x (dep op, lo, hi)];
//     the operands of a task live in an op descriptor table (pointers, dims) indexed by `op`;
//   * scoreboard[op][tile] holds the run epoch once that tile is complete: consumers spin
//     (thread-strided, agent scope, relaxed poll + one acquire fence) on the contiguous tile range
//     of each dependency, producers publish with an agent-scope release after a workgroup barrier;
//     the epoch changes every launch, so the scoreboard is never reset;
//   * tasks are queued in topological order round-robin over the queues, so the globally earliest
//     unfinished task is always at the head of its queue and has all its dependencies satisfied:
//     the kernel cannot deadlock as long as every workgroup is resident (grid <= k x 256 CUs).
//
// Task bodies are decode-shaped (at most 16 tokens): the linear layers stream the weights exactly
// once with 16-byte loads and feed 16x16x32 MFMAs whose M dimension is the (zero-padded) batch.
#include "tdist/common.hpp"
#include "tdist/flash_decode_device.hpp"
#include "tdist/gemm_gfx950.hpp"
#include "tdist/primitives.hpp"
#include "tdist/shmem_device.hpp"

using namespace tdist;
using tdist::gemm::f32x4;

namespace {

enum TaskType : int {
  TASK_LINEAR = 1,
  TASK_RMS_NORM = 2,
  TASK_ADD = 3,
  TASK_QK_NORM_ROPE_KV = 4,
  TASK_ATTN_SPLIT = 5,
  TASK_ATTN_COMBINE = 6,
  TASK_ALLREDUCE = 7,
  TASK_BARRIER = 8,
  TASK_PREFETCH = 9,
  TASK_SILU_MUL = 10,
};

constexpr int TASK_WORDS = 16;
constexpr int MAX_DEPS = 3;

// Generic op descriptor; the meaning of the slots depends on the task type (see the bodies).
struct MegaOp {
  void* p[10];
  int i[16];
  float f[4];
};

struct MegaParams {
  SymmCtx ctx;
  const int* queues;       // [num_queues][max_tasks][TASK_WORDS]
  const int* queue_len;    // [num_queues]
  const MegaOp* ops;
  int* scoreboard;         // [num_ops][max_tiles]
  int* err;
  int max_tasks, max_tiles, epoch;
};

__device__ __forceinline__ float bf2f(uint16_t v) { return __builtin_bit_cast(float, (uint32_t)v << 16); }
__device__ __forceinline__ uint16_t f2bf(float f) { return __builtin_bit_cast(uint16_t, (__bf16)f); }

__device__ __forceinline__ void unpack8(const uint4& q, float* o) {
  const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    o[2 * k] = bf2f((uint16_t)(w[k] & 0xffffu));
    o[2 * k + 1] = bf2f((uint16_t)(w[k] >> 16));
  }
}
__device__ __forceinline__ uint4 pack8(const float* v) {
  uint4 r;
  r.x = f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
  r.y = f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
  r.z = f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16);
  r.w = f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
  return r;
}

// Operand pointers come out of the op table in memory, so the compiler sees generic pointers and would
// emit FLAT loads / stores: those also count on the LDS counter, return out of order (a counted
// s_waitcnt vmcnt(N) means nothing for them) and cost an aperture check each.  Every hot access goes
// through these casts to the global address space instead.
#define TDIST_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ const TDIST_GLOBAL T* as_global(const void* p) {
  return (const TDIST_GLOBAL T*)(uintptr_t)p;
}
template <class T>
__device__ __forceinline__ TDIST_GLOBAL T* as_global_mut(void* p) {
  return (TDIST_GLOBAL T*)(uintptr_t)p;
}

// Write-through (sc1) stores: the bytes leave this XCD's L2 at once, so a hand-off needs no release
// fence - only a drain (s_waitcnt vmcnt(0)) in every storing wave before the flag.  4 / 8 bytes per
// store (agent-scope relaxed atomic stores lower to global_store ... sc1).
__device__ __forceinline__ void st_wt32(void* p, uint32_t v) {
  __hip_atomic_store(as_global_mut<uint32_t>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_wt128(void* p, const uint4& v) {
  __hip_atomic_store(as_global_mut<uint64_t>(p), ((uint64_t)v.y << 32) | v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __hip_atomic_store(as_global_mut<uint64_t>(p) + 1, ((uint64_t)v.w << 32) | v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// L1-bypassing (sc1) 16-byte load of handed-off bytes: with write-through producer stores and every
// consumer load of this form, a hand-off needs no acquire fence (no L1 invalidate) either.
__device__ __forceinline__ uint4 ld_sc1_128(const void* p) {
  const uint64_t lo = __hip_atomic_load(as_global<uint64_t>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const uint64_t hi = __hip_atomic_load(as_global<uint64_t>(p) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return uint4{(uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32)};
}
__device__ __forceinline__ float ld_sc1_bf16(const uint16_t* p) {
  // 4-byte aligned container load, then pick the half
  const uint32_t w = __hip_atomic_load(as_global<uint32_t>((const void*)((uintptr_t)p & ~(uintptr_t)3)), __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_AGENT);
  return bf2f((uint16_t)(((uintptr_t)p & 2) ? (w >> 16) : (w & 0xffffu)));
}

// Streamed-once weights.  Plain loads, not non-temporal ones: a lane reads 16 bytes of a 64-byte run per weight row
// (the MFMA operand layout), the next K step reads the other half of the same 128-byte line, and only a cached load
// lets the L1 serve that second half - with non-temporal loads every line came up from the L2 twice (Qwen3-8B,
// batch 1: 6.42 -> 6.06 ms per step).
__device__ __forceinline__ uint4 ld_nt128(const void* p) {
  typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
  const u32x4 v = *as_global<u32x4>(p);
  return uint4{v[0], v[1], v[2], v[3]};
}

// Activation fragment through a bounds-checked buffer descriptor, L1-bypassing (sc1, aux bit 4): an offset
// past the descriptor's size reads as zero, so rows beyond the batch and a window that starts before the
// row need no branch.  `off` = byte offset of the lane, `imm` a compile-time addend (instruction offset).
typedef __attribute__((__vector_size__(4 * sizeof(unsigned int)))) unsigned int u32x4_t;
__device__ __forceinline__ uint4 ld_x_frag(__amdgpu_buffer_rsrc_t rsrc, int off, int imm) {
  const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, off + imm, 0, 16);
  return uint4{v[0], v[1], v[2], v[3]};
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// ---- LINEAR: out[M, N] (+ residual) = x[M, K] @ W[N, K]^T, M <= 16, bf16 ---------------------------
// p0 x, p1 W, p2 out, p3 residual (or null); i0 M, i1 N, i2 K, i3 ldx, i4 ldo, i5 ldr, i6 BN (16/32/64),
// i7 mode: 0 plain, 1 gate/up pair (W rows [0,N) = gate, [N,2N) = up; out = silu(gate) * up).
// Tile t covers columns [t*BN, (t+1)*BN).  Waves: WN = BN/16 column groups x WK = 4/WN K-slices.
//
// A task runs in two steps.  linear_issue() - called BEFORE the task's dependencies are waited for -
// puts the first PF weight loads of every wave in flight (weights depend on nobody), so the memory
// round trip of a task's first bytes, and with 16-column tiles of a 4096-deep layer ALL its bytes, overlaps
// the wait for the producer and the write-out of the previous task.  It issues exactly PF loads per wave
// whatever the shape (addresses past the wave's K range are clamped and their products multiplied by a
// zero activation), because the deferred publish of the previous task counts them (s_waitcnt vmcnt(PF)).
// task_linear() then loads the activations, consumes the prefetched fragments and streams the rest.
constexpr int LIN_PF = 8;
struct LinearPrefetch {
  uint4 w[LIN_PF];
};

struct LinearGeom {
  int WN, WK, wn, wk, li, lg, n0, ks0, ks1, ksteps;
  int kb;                     // K-step held by prefetched fragment 0 (fragment u holds K-step kb + u)
  const uint16_t *wr, *ur;    // this lane's gate / plain row and up row, at its 16-byte column of K-step 0
};

// The prefetch window of a wave is LIN_PF (mode 1: LIN_PF / 2 per matrix half) consecutive K-steps read
// at constant offsets from ONE base address per lane (32 per-load addresses would cost 64 registers).  It
// starts at the wave's first K-step; where fewer K-steps than the window are left before the END OF THE
// WEIGHT MATRIX (last rows of a short-K layer) it starts earlier, by the same amount in every lane.  K-steps
// of the window outside [ks0, ks1) - the neighbouring wave's, the next row's - are loaded and discarded.
__device__ __forceinline__ LinearGeom linear_geom(const MegaOp& op, int tile) {
  LinearGeom g;
  const int N = op.i[1], K = op.i[2], BN = op.i[6], mode = op.i[7];
  g.WN = BN >> 4;
  g.WK = 4 / g.WN;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  g.wn = wave % g.WN;
  g.wk = wave / g.WN;
  g.li = lane & 15;
  g.lg = lane >> 4;
  g.n0 = tile * BN + g.wn * 16;
  const int ncol = min(g.n0 + g.li, N - 1);
  g.ksteps = K / 32;
  const int per = (g.ksteps + g.WK - 1) / g.WK;
  g.ks0 = min(g.wk * per, g.ksteps);
  g.ks1 = min(g.ks0 + per, g.ksteps);
  const uint16_t* W = (const uint16_t*)op.p[1];
  g.wr = W + (long long)ncol * K + g.lg * 8;
  g.ur = W + (long long)(ncol + N) * K + g.lg * 8;
  // K-steps from (last row any lane of this wave reads, ks0) to the end of the matrix; the gate half of a
  // gate/up matrix is followed by the N up rows, so only the second half can come up short
  const int rows = mode == 1 ? 2 * N : N;
  const int last_row = min(g.n0 + 15, N - 1) + (mode == 1 ? N : 0);
  const long long left = (long long)(rows - 1 - last_row) * g.ksteps + (g.ksteps - g.ks0);
  const int PFm = mode == 1 ? LIN_PF / 2 : LIN_PF;
  g.kb = g.ks0 - (left < PFm ? PFm - (int)left : 0);
  return g;
}

__device__ __forceinline__ void linear_issue(const MegaOp& op, int tile, LinearPrefetch& pre) {
  const LinearGeom g = linear_geom(op, tile);
  constexpr int H = LIN_PF / 2;
  // two bases, constant offsets: fragments [0, H) are K-steps kb .. kb + H of the (gate) row, fragments
  // [H, 2H) the same K-steps of the up row (mode 1) or K-steps kb + H .. kb + 2H of the same row (mode 0)
  const uint16_t* a = g.wr + (long long)g.kb * 32;
  const uint16_t* b = op.i[7] == 1 ? g.ur + (long long)g.kb * 32 : a + H * 32;
#pragma unroll
  for (int u = 0; u < H; ++u) pre.w[u] = ld_nt128(a + u * 32);
#pragma unroll
  for (int u = 0; u < H; ++u) pre.w[H + u] = ld_nt128(b + u * 32);
}

__device__ __forceinline__ void task_linear(const MegaOp& op, int tile, float* smem, const LinearPrefetch& pre) {
  using VT = tdist::gemm::bf16x8;
  const int M = op.i[0], N = op.i[1];
  const long long ldx = op.i[3], ldo = op.i[4], ldr = op.i[5];
  const int mode = op.i[7];
  const LinearGeom g = linear_geom(op, tile);
  const int WN = g.WN, WK = g.WK, wn = g.wn, wk = g.wk, li = g.li, lg = g.lg, n0 = g.n0, ks1 = g.ks1;
  const int lane = threadIdx.x & 63;
  // activations [M, K] behind a descriptor of exactly their bytes; lanes of rows >= M point past its end
  const int K = op.i[2];
  const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)op.p[0], 0, (int)(((M - 1) * ldx + K) * 2), 0x00020000);
  const int xoff = li < M ? (int)((li * ldx + lg * 8) * 2) : (int)0x40000000;
  const uint16_t* wr = g.wr;
  const uint16_t* ur = g.ur;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f};
  constexpr int U = 8;
  // ---- the prefetched fragments: K-steps kb .. kb + PFm, of which [ks0, ks1) are this wave's ----
  const int PFm = mode == 1 ? LIN_PF / 2 : LIN_PF;
  const int xw = xoff + g.kb * 64;                          // (kb < 0: far past the end as an unsigned offset)
  const int v0 = g.ks0 - g.kb, v1 = g.ks1 - g.kb;          // valid fragments: v0 <= u < v1
  const uint4 zero4 = {0, 0, 0, 0};
  constexpr int H = LIN_PF / 2;
  constexpr int UB = H < U ? H : U;        // activation fragments loaded per batch
#pragma unroll
  for (int b = 0; b < H; b += UB) {
    uint4 xv[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) xv[u] = ld_x_frag(xrs, xw, (b + u) * 64);
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const VT a = __builtin_bit_cast(VT, (b + u >= v0 && b + u < v1) ? xv[u] : zero4);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(VT, pre.w[b + u]), acc, 0, 0, 0);
      if (mode == 1) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(VT, pre.w[H + b + u]), acc2, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);   // one batch of activation loads at a time (registers)
  }
  if (mode != 1) {
#pragma unroll
    for (int b = H; b < LIN_PF; b += UB) {
      uint4 xv[UB];
#pragma unroll
      for (int u = 0; u < UB; ++u) xv[u] = ld_x_frag(xrs, xw, (b + u) * 64);
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const VT a = __builtin_bit_cast(VT, (b + u >= v0 && b + u < v1) ? xv[u] : zero4);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(VT, pre.w[b + u]), acc, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- the rest of the K range, streamed in batches of U ----
  int ks = g.kb + PFm;                 // (>= ks1 whenever the window was moved back)
  for (; ks + U <= ks1; ks += U) {
    uint4 wv[U], xv[U], uv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) wv[u] = ld_nt128(wr + (ks + u) * 32);
    if (mode == 1) {
#pragma unroll
      for (int u = 0; u < U; ++u) uv[u] = ld_nt128(ur + (ks + u) * 32);
    }
    const int xk = xoff + ks * 64;
#pragma unroll
    for (int u = 0; u < U; ++u) xv[u] = ld_x_frag(xrs, xk, u * 64);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const VT a = __builtin_bit_cast(VT, xv[u]);
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(VT, wv[u]), acc, 0, 0, 0);
      if (mode == 1) acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(VT, uv[u]), acc2, 0, 0, 0);
    }
  }
  for (; ks < ks1; ++ks) {
    const uint4 wv = ld_nt128(wr + ks * 32);
    const uint4 xv = ld_x_frag(xrs, xoff + ks * 64, 0);
    const VT a = __builtin_bit_cast(VT, xv);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(VT, wv), acc, 0, 0, 0);
    if (mode == 1) {
      const uint4 uv = ld_nt128(ur + ks * 32);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(VT, uv), acc2, 0, 0, 0);
    }
  }
  // accumulator: row (token) = 4*lg + r, column = li.  Reduce the K-slices through LDS.
  if (WK > 1) {
    // smem layout [wk][wn][2][64 lanes][4]
    float* slot = smem + (((wk * WN + wn) * 2) * 64 + lane) * 4;
#pragma unroll
    for (int r = 0; r < 4; ++r) slot[r] = acc[r];
    if (mode == 1) {
#pragma unroll
      for (int r = 0; r < 4; ++r) slot[256 + r] = acc2[r];
    }
    __syncthreads();
    if (wk != 0) return;
    for (int k2 = 1; k2 < WK; ++k2) {
      const float* o = smem + (((k2 * WN + wn) * 2) * 64 + lane) * 4;
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[r] += o[r];
      if (mode == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc2[r] += o[256 + r];
      }
    }
  }
  // Epilogue: lanes li and li^1 hold neighbouring columns; the even lane stores both as one 4-byte
  // write-through store (N and the row strides are even).
  const int col = n0 + li;
  uint16_t* out = (uint16_t*)op.p[2];
  const uint16_t* res = (const uint16_t*)op.p[3];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * lg + r;
    float v = acc[r];
    if (mode == 1) {
      const float g = v;
      v = g / (1.f + __expf(-g)) * acc2[r];
    }
    const bool ok = row < M && col < N;
    if (res && ok) v += ld_sc1_bf16(res + (long long)row * ldr + col);
    const uint32_t mine = f2bf(v);
    const uint32_t other = __shfl_xor((int)mine, 1);
    if (ok && (li & 1) == 0) st_wt32(out + (long long)row * ldo + col, mine | (other << 16));
  }
}

// ---- RMS_NORM: tile = row.  p0 x, p1 weight, p2 out, p3 residual-in (or null), p4 residual-out (or
// null: x + residual written there); i0 M, i1 H; f0 eps.
__device__ __forceinline__ void task_rms_norm(const MegaOp& op, int row, float* smem) {
  const int H = op.i[1];
  const uint16_t* x = (const uint16_t*)op.p[0] + (long long)row * H;
  const uint16_t* w = (const uint16_t*)op.p[1];
  uint16_t* out = (uint16_t*)op.p[2] + (long long)row * H;
  const uint16_t* rin = op.p[3] ? (const uint16_t*)op.p[3] + (long long)row * H : nullptr;
  uint16_t* rout = op.p[4] ? (uint16_t*)op.p[4] + (long long)row * H : nullptr;
  float ss = 0.f;
  for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
    float v[8], r[8];
    unpack8(ld_sc1_128(x + c), v);
    if (rin) {
      unpack8(ld_sc1_128(rin + c), r);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = bf2f(f2bf(v[k] + r[k]));
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) ss += v[k] * v[k];
  }
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) smem[threadIdx.x >> 6] = ss;
  __syncthreads();
  const float tot = smem[0] + smem[1] + smem[2] + smem[3];
  const float inv = rsqrtf(tot / (float)H + op.f[0]);
  for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
    float v[8], r[8], g[8];
    unpack8(ld_sc1_128(x + c), v);
    if (rin) {
      // recomputed rather than re-read: rout may alias rin (in place), and it is written here, after
      // this thread's own read of the same 16 bytes
      unpack8(ld_sc1_128(rin + c), r);
#pragma unroll
      for (int k = 0; k < 8; ++k) v[k] = bf2f(f2bf(v[k] + r[k]));
      if (rout) st_wt128(rout + c, pack8(v));
    }
    unpack8(*(const uint4*)(w + c), g);
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = v[k] * inv * g[k];
    st_wt128(out + c, pack8(v));
  }
}

// ---- ADD / SILU_MUL: tile = row.  ADD: p0 a, p1 b, p2 out; i1 H.  SILU_MUL: p0 gate_up [M, 2I], p2 out [M, I]; i1 I.
__device__ __forceinline__ void task_add(const MegaOp& op, int row) {
  const int H = op.i[1];
  const uint16_t* a = (const uint16_t*)op.p[0] + (long long)row * H;
  const uint16_t* b = (const uint16_t*)op.p[1] + (long long)row * H;
  uint16_t* o = (uint16_t*)op.p[2] + (long long)row * H;
  for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
    float x[8], y[8];
    unpack8(ld_sc1_128(a + c), x);
    unpack8(ld_sc1_128(b + c), y);
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] += y[k];
    st_wt128(o + c, pack8(x));
  }
}

__device__ __forceinline__ void task_silu_mul(const MegaOp& op, int row) {
  const int I = op.i[1];
  const uint16_t* g = (const uint16_t*)op.p[0] + (long long)row * 2 * I;
  uint16_t* o = (uint16_t*)op.p[2] + (long long)row * I;
  for (int c = threadIdx.x * 8; c < I; c += 256 * 8) {
    float x[8], y[8];
    unpack8(ld_sc1_128(g + c), x);
    unpack8(ld_sc1_128(g + I + c), y);
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = x[k] / (1.f + __expf(-x[k])) * y[k];
    st_wt128(o + c, pack8(x));
  }
}

// ---- QK_NORM_ROPE_KV: tile = token b.  One wave per head (heads strided over the 4 waves).
// p0 qkv [M, (Hq + 2 Hkv) D], p1 q_norm w (or null), p2 k_norm w (or null), p3 cos, p4 sin [pos, D/2] f32,
// p5 kv_len int32 [M] (position of the new token), p6 k_cache, p7 v_cache [slots, Hkv, D], p8 block_table,
// p9 q_out [M, Hq, D]; i0 M, i1 Hq, i2 Hkv, i3 D (64 | 128), i4 page, i5 table_stride; f0 eps.
__device__ __forceinline__ void task_qk_norm_rope_kv(const MegaOp& op, int b) {
  const int Hq = op.i[1], Hkv = op.i[2], D = op.i[3], page = op.i[4], tstride = op.i[5];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint16_t* qkv = (const uint16_t*)op.p[0] + (long long)b * (Hq + 2 * Hkv) * D;
  const float* cosb = (const float*)op.p[3];
  const float* sinb = (const float*)op.p[4];
  const int pos = ((const int*)op.p[5])[b];
  const int* table = (const int*)op.p[8] + (long long)b * tstride;
  const long long slot = (long long)table[pos / page] * page + pos % page;
  const int half = D / 2;
  const int per = D / 64;          // elements per lane in each half: D=128 -> 1, handled as pairs (i, i+half)
  for (int h = wave; h < Hq + 2 * Hkv; h += 4) {
    const uint16_t* src = qkv + (long long)h * D;
    const bool is_q = h < Hq, is_k = !is_q && h < Hq + Hkv;
    uint16_t* dst;
    if (is_q)
      dst = (uint16_t*)op.p[9] + ((long long)b * Hq + h) * D;
    else if (is_k)
      dst = (uint16_t*)op.p[6] + (slot * Hkv + (h - Hq)) * D;
    else
      dst = (uint16_t*)op.p[7] + (slot * Hkv + (h - Hq - Hkv)) * D;
    if (!is_q && !is_k) {   // V: plain copy
      for (int c = lane; c < D; c += 64) dst[c] = src[c];
      continue;
    }
    const uint16_t* nw = (const uint16_t*)(is_q ? op.p[1] : op.p[2]);
    // lane handles pairs (c, c + half) for c = lane, lane + 64, ... < half
    float x1[2], x2[2];
    float ss = 0.f;
    int cnt = 0;
    for (int c = lane; c < half; c += 64, ++cnt) {
      x1[cnt] = bf2f(src[c]);
      x2[cnt] = bf2f(src[c + half]);
      ss += x1[cnt] * x1[cnt] + x2[cnt] * x2[cnt];
    }
    (void)per;
    if (nw) {
      ss = wave_sum(ss);
      const float inv = rsqrtf(ss / (float)D + op.f[0]);
      cnt = 0;
      for (int c = lane; c < half; c += 64, ++cnt) {
        x1[cnt] = bf2f(f2bf(x1[cnt] * inv * bf2f(nw[c])));
        x2[cnt] = bf2f(f2bf(x2[cnt] * inv * bf2f(nw[c + half])));
      }
    }
    cnt = 0;
    for (int c = lane; c < half; c += 64, ++cnt) {
      const float cs = cosb[(long long)pos * half + c], sn = sinb[(long long)pos * half + c];
      dst[c] = f2bf(x1[cnt] * cs - x2[cnt] * sn);
      dst[c + half] = f2bf(x2[cnt] * cs + x1[cnt] * sn);
    }
  }
}

// ---- ATTN_SPLIT / ATTN_COMBINE.  p0 q [B, Hq, D], p1 k_cache, p2 v_cache, p3 block_table, p4 kv_len,
// p5 partial [B, Hq, splits, D + 1] f32, p6 out [B, Hq, D] bf16; i0 B, i1 Hq, i2 Hkv, i3 D, i4 splits, i5 page,
// i6 table_stride, i7 kv_len_is_position (1: the cache already holds the new token, length = kv_len + 1);
// f0 scale, f1 soft_cap.  ATTN_SPLIT tile = (b * Y + y) * splits + z; ATTN_COMBINE tile = b.
__device__ __forceinline__ void fill_decode_args(const MegaOp& op, fdec::DecodeArgs& a, const int* lens) {
  const int D = op.i[3];
  a.q = (const uint16_t*)op.p[0];
  a.k_cache = (const uint16_t*)op.p[1];
  a.v_cache = (const uint16_t*)op.p[2];
  a.block_table = (const int*)op.p[3];
  a.kv_len = lens;
  a.partial = (float*)op.p[5];
  a.out = nullptr;
  a.tickets = nullptr;
  a.q_heads = op.i[1], a.kv_heads = op.i[2], a.splits = op.i[4];
  a.q_stride_b = (long long)a.q_heads * D, a.q_stride_h = D;
  a.k_stride_slot = (long long)a.kv_heads * D, a.k_stride_h = D;
  a.v_stride_slot = a.k_stride_slot, a.v_stride_h = D;
  a.p_stride_s = D + 1, a.p_stride_h = (long long)a.splits * (D + 1), a.p_stride_b = a.p_stride_h * a.q_heads;
  a.o_stride_b = 0, a.o_stride_h = 0;
  a.table_stride = op.i[6];
  a.page_size = op.i[5];
  a.page_shift = (a.page_size & (a.page_size - 1)) == 0 ? __builtin_ctz(a.page_size) : -1;
  a.fused = 0;
  a.deep = 0;
  a.p16 = 0;
  a.sm_scale = op.f[0], a.soft_cap = op.f[1];
}

__device__ __forceinline__ void task_attn_split(const MegaOp& op, int tile, char* smem, int* s_lens) {
  const int Hq = op.i[1], Hkv = op.i[2], D = op.i[3], splits = op.i[4], B = op.i[0];
  const int group = Hq / Hkv;
  const int Y = Hkv * ((group + 15) / 16);
  const int z = tile % splits;
  const int by = (tile / splits) % Y;
  const int b = tile / (splits * Y);
  // effective lengths (position + 1 when the step's token was just appended) staged in LDS
  if ((int)threadIdx.x < B) s_lens[threadIdx.x] = ((const int*)op.p[4])[threadIdx.x] + (op.i[7] ? 1 : 0);
  __syncthreads();
  fdec::DecodeArgs a;
  fill_decode_args(op, a, s_lens);
  if (D == 128)
    fdec::decode_split_body<true, 128, false>(a, b, by, z, smem);
  else
    fdec::decode_split_body<true, 64, false>(a, b, by, z, smem);
}

__device__ __forceinline__ void task_attn_combine(const MegaOp& op, int b) {
  const int Hq = op.i[1], D = op.i[3], splits = op.i[4];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const float* part = (const float*)op.p[5] + (long long)b * Hq * splits * (D + 1);
  uint16_t* out = (uint16_t*)op.p[6] + (long long)b * Hq * D;
  for (int h = wave; h < Hq; h += 4)
    fdec::merge_rows<true>(part + (long long)h * splits * (D + 1), D + 1, splits, D, nullptr, 0, out + (long long)h * D, 1,
                           lane, 64);
}

// ---- ALLREDUCE (one-shot, pull): tile = row.  p0 symmetric input [M, H] (this rank's address), p1 out [M, H]
// local, p2 symmetric flags int32 [M][w]; i0 M, i1 H.  Every rank publishes row `tile` (flag := epoch on all
// peers), waits for the w flags of the row and sums the w copies in fp32.
__device__ __forceinline__ void task_allreduce(const MegaParams& P, const MegaOp& op, int row) {
  const int H = op.i[1];
  const int w = P.ctx.num_ranks, rank = P.ctx.rank;
  uint16_t* in = (uint16_t*)op.p[0];
  int* flags = (int*)op.p[2];
  if ((int)threadIdx.x < w) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    store_relaxed_system(symm_at(P.ctx, flags, (int)threadIdx.x) + row * w + rank, P.epoch);
  }
  wait(flags + row * w, w, SCOPE_SYS, SEM_ACQUIRE, P.epoch, P.err);
  uint16_t* out = (uint16_t*)op.p[1] + (long long)row * H;
  for (int c = threadIdx.x * 8; c < H; c += 256 * 8) {
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < w; ++i) {
      const int peer = (rank + i) % w;
      float v[8];
      unpack8(*(const uint4*)(symm_at(P.ctx, in, peer) + (long long)row * H + c), v);
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] += v[k];
    }
    st_wt128(out + c, pack8(acc));
  }
}

// ---- PREFETCH: pull [tile * chunk, +chunk) bytes of p0 towards L2 / Infinity Cache.  i0 = chunk bytes,
// i1.. total bytes (two words).  The loaded values feed a store that never executes.
__device__ __forceinline__ void task_prefetch(const MegaOp& op, int tile, int* sink) {
  const long long chunk = op.i[0];
  const long long total = ((long long)(unsigned)op.i[2] << 32) | (unsigned)op.i[1];
  const long long b0 = (long long)tile * chunk, b1 = min(b0 + chunk, total);
  const char* base = (const char*)op.p[0];
  unsigned acc = 0;
  for (long long o = b0 + (long long)threadIdx.x * 128; o < b1; o += 256 * 128) acc += *(const unsigned*)(base + o);
  if (acc == 0x9e3779b9u && b1 < 0) *sink = (int)acc;
}

// The interpreter pipelines three things across task boundaries of a queue:
//   * the record of task t+1 is fetched while task t runs (two LDS slots);
//   * a linear task's first weight loads go out before its dependencies are waited for (linear_issue);
//   * task t is PUBLISHED after task t+1's weight loads were issued: the wait that covers t's stores is a
//     counted one (everything but the LIN_PF youngest memory operations of each wave), so the write-out
//     of one task drains under the loads of the next instead of idling the queue.
__global__ __launch_bounds__(256, 2) void mega_kernel(MegaParams P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_task = (int*)smem;                  // [2][TASK_WORDS], then [16] effective kv lengths (task_attn_split)
  int* s_lens = s_task + 2 * TASK_WORDS;     // [16] effective kv lengths
  char* work = smem + 256;                   // task scratch
  const int q = blockIdx.x;
  const int ntask = P.queue_len[q];
  const int* myq = P.queues + (long long)q * P.max_tasks * TASK_WORDS;
  int seen_op[MAX_DEPS] = {-1, -1, -1}, seen_lo[MAX_DEPS] = {0, 0, 0}, seen_hi[MAX_DEPS] = {0, 0, 0};
  int seen_next = 0;
  if (ntask > 0 && threadIdx.x < TASK_WORDS) s_task[threadIdx.x] = myq[threadIdx.x];
  __syncthreads();
  int pend_type = 0, pend_op = 0, pend_tile = 0;      // task whose tile is computed but not yet published
  auto publish = [&](bool loads_behind) {
    // Tasks whose outputs are all write-through stores need only the drain + barrier; the attention
    // and rope/kv-cache tasks store plain and pay one agent-scope release (L2 write-back).
    if (loads_behind) {
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LIN_PF) : "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (threadIdx.x == 0) {
      if (pend_type == TASK_ATTN_SPLIT || pend_type == TASK_ATTN_COMBINE || pend_type == TASK_QK_NORM_ROPE_KV) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      store_relaxed_agent(P.scoreboard + (long long)pend_op * P.max_tiles + pend_tile, P.epoch);
    }
  };
  for (int t = 0; t < ntask; ++t) {
    const int* rec = s_task + (t & 1) * TASK_WORDS;
    // (workgroup-uniform words: kept in scalar registers, the vector file is needed for the weight prefetch)
    const int type = __builtin_amdgcn_readfirstlane(rec[0]), opi = __builtin_amdgcn_readfirstlane(rec[1]);
    const int tile = __builtin_amdgcn_readfirstlane(rec[2]), ndeps = __builtin_amdgcn_readfirstlane(rec[3]);
    // The op table is read-only for the kernel: read it through the constant address space, so that the
    // descriptor (pointers, dims) arrives by scalar loads and lives in scalar registers.  Through a generic
    // pointer the compiler must assume the kernel's own stores may change it and uses vector loads - two
    // vector registers per pointer, per lane, in every task body.
    MegaOp op;
    {
      typedef const __attribute__((address_space(4))) MegaOp ConstOp;
      ConstOp* cop = (ConstOp*)(uintptr_t)(P.ops + opi);
#pragma unroll
      for (int k = 0; k < 10; ++k) op.p[k] = cop->p[k];
#pragma unroll
      for (int k = 0; k < 16; ++k) op.i[k] = cop->i[k];
#pragma unroll
      for (int k = 0; k < 4; ++k) op.f[k] = cop->f[k];
    }
    // record of the next task (lands while this one runs; stored to the other slot at the bottom)
    int next_word = 0;
    if (t + 1 < ntask && threadIdx.x < TASK_WORDS) next_word = myq[(t + 1) * TASK_WORDS + threadIdx.x];
    LinearPrefetch pre;
    const bool linear = type == TASK_LINEAR;
    if (linear) linear_issue(op, tile, pre);
    if (t > 0) publish(linear);
    // ---- wait for the dependencies (contiguous tile ranges of producer ops) ----
    // A range this workgroup has already waited for (typical: consecutive column tiles of one
    // linear op) is complete for good and its data was acquired then: skip the poll and the fence.
    // Tasks that read handed-off bytes only through sc1 loads skip the L1 invalidate.
    const bool acquire = !(type == TASK_LINEAR || type == TASK_RMS_NORM || type == TASK_ADD || type == TASK_SILU_MUL);
    for (int d = 0; d < ndeps && d < MAX_DEPS; ++d) {
      const int dop = __builtin_amdgcn_readfirstlane(rec[4 + 3 * d]), lo = __builtin_amdgcn_readfirstlane(rec[5 + 3 * d]);
      const int hi = __builtin_amdgcn_readfirstlane(rec[6 + 3 * d]);
      bool seen = false;
#pragma unroll
      for (int k = 0; k < MAX_DEPS; ++k) seen |= (dop == seen_op[k] && lo >= seen_lo[k] && hi <= seen_hi[k]);
      if (seen) continue;
      wait(P.scoreboard + (long long)dop * P.max_tiles + lo, hi - lo, SCOPE_GPU, acquire ? SEM_ACQUIRE : SEM_RELAXED,
           P.epoch, P.err);
      seen_op[seen_next] = dop, seen_lo[seen_next] = lo, seen_hi[seen_next] = hi;
      seen_next = (seen_next + 1) % MAX_DEPS;
    }
    switch (type) {
      case TASK_LINEAR: task_linear(op, tile, (float*)work, pre); break;
      case TASK_RMS_NORM: task_rms_norm(op, tile, (float*)work); break;
      case TASK_ADD: task_add(op, tile); break;
      case TASK_SILU_MUL: task_silu_mul(op, tile); break;
      case TASK_QK_NORM_ROPE_KV: task_qk_norm_rope_kv(op, tile); break;
      case TASK_ATTN_SPLIT: task_attn_split(op, tile, work, s_lens); break;
      case TASK_ATTN_COMBINE: task_attn_combine(op, tile); break;
      case TASK_ALLREDUCE: task_allreduce(P, op, tile); break;
      case TASK_BARRIER:
        if (!shmem::barrier_all_block(P.ctx, (int*)op.p[0])) atomicOr(P.err, 2);
        break;
      case TASK_PREFETCH: task_prefetch(op, tile, P.err + 1); break;
      default: atomicOr(P.err, 4); break;
    }
    pend_type = type, pend_op = opi, pend_tile = tile;
    // hand the next record over: every wave has read this task's record long ago, the other slot is free
    if (threadIdx.x < TASK_WORDS) s_task[((t + 1) & 1) * TASK_WORDS + threadIdx.x] = next_word;
    // (a bare barrier: this task's global stores stay in flight - publish() counts them down later)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (ntask > 0) publish(false);
}

}  // namespace

// LDS: 256 B header + max(attention scratch (128 + 64 * 128) floats, linear reduce 4 * 2 * 64 * 4 floats).
static constexpr int MEGA_LDS = 256 + (128 + 64 * 128) * 4;

TDIST_API int tdist_mega_op_size() { return (int)sizeof(MegaOp); }
TDIST_API int tdist_mega_task_words() { return TASK_WORDS; }

TDIST_API int tdist_mega_launch(const SymmCtx* ctx, const int* queues, const int* queue_len, const void* ops,
                                int* scoreboard, int* err, int num_queues, int max_tasks, int max_tiles, int epoch,
                                void* stream) {
  TDIST_REQUIRE(num_queues > 0 && num_queues <= 1024, "bad queue count");
  TDIST_REQUIRE(max_tasks > 0 && max_tiles > 0 && epoch != 0, "bad launch parameters");
  MegaParams P;
  P.ctx = *ctx;
  P.queues = queues;
  P.queue_len = queue_len;
  P.ops = (const MegaOp*)ops;
  P.scoreboard = scoreboard;
  P.err = err;
  P.max_tasks = max_tasks;
  P.max_tiles = max_tiles;
  P.epoch = epoch;
  static int resident = 0;    // workgroups of this kernel the device keeps resident at once
  if (resident == 0) {
    TDIST_HIP_TRY(hipFuncSetAttribute((const void*)mega_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, MEGA_LDS));
    int dev = 0, cus = 0, per_cu = 0;
    TDIST_HIP_TRY(hipGetDevice(&dev));
    TDIST_HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    TDIST_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)mega_kernel, 256, MEGA_LDS));
    resident = cus * per_cu;
  }
  // Tasks wait for tasks of other queues: a queue whose workgroup is not resident would never run and the
  // waiters would time out, so a grid larger than the resident set is refused up front.
  TDIST_REQUIRE(num_queues <= resident, "more queues than workgroups the device keeps resident");
  hipLaunchKernelGGL(mega_kernel, dim3(num_queues), dim3(256), MEGA_LDS, (hipStream_t)stream, P);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
