// AllReduce over the xGMI full mesh: one-shot (latency-bound sizes) and two-shot (bandwidth-bound).
//
// Both run as ONE kernel of G independent workgroups; workgroup c owns slice c of the data on every
// rank and only ever waits for workgroup c of the other ranks (flag index [src_rank][c]), so there
// is no grid barrier and no cross-rank barrier.  Flags carry the call's epoch and the workspaces /
// flags are double-buffered by call parity, so nothing is reset between calls.
//
// The call number lives ON THE DEVICE (`state[0]`, one word per context and rank): every workgroup
// reads it on entry and derives buffer parity and flag value from it; the workgroup that finishes
// last advances it.  A launch therefore carries no per-call argument, and a HIP graph that captured
// it picks a fresh epoch on every replay - all ranks issue the same sequence of launches, so their
// counters stay equal.
//
//   one-shot : push own slice into every peer's slot [rank]; flag; wait w-1 flags; sum the w
//              contributions in rank order (bitwise identical result on every rank).
//   two-shot : rank r owns chunk r.  Phase 1: push sub-slice of chunk p to peer p (reduce-scatter);
//              flag; wait; sum -> reduced sub-slice, stored into every rank's gather buffer
//              (all-gather); flag; wait; copy the gather buffer to the output.
//              Per link: n/w bytes per phase instead of n for one-shot.
//
// Role in the reference: kernels/nvidia/allreduce.py one_shot_push :332-383, two_shot_push :446-524;
// the NVLS multimem / TMA / double-tree variants (:214-329, :386-443, :527-682) have no xGMI
// counterpart and are served by these two kernels (see ops/allreduce.py for the mapping).
#include "tdist/cache_policy.hpp"
#include "tdist/call_state.hpp"
#include "tdist/common.hpp"
#include "tdist/primitives.hpp"
#include "tdist/shmem_device.hpp"

using namespace tdist;

struct ArParams {
  SymmCtx ctx;
  const void* in;
  void* out;
  void* ws;        // symmetric, this call's parity (resolved on the device, see resolve_call)
  int* flags;      // symmetric (signal arena), this call's parity: [2][w][G]
  void *ws0, *ws1;     // both parities, as given by the host (separate fields selected with ?: - indexing an
  int *flags0, *flags1;   // array member with the runtime parity would push the argument block into scratch)
  int* state;      // [0] call number in [0, 2 * EPOCH_MOD), [1] finished-workgroup count
  int* err;
  long long n;     // elements
  long long cap;   // one-shot: slot stride in elements; two-shot: chunk stride (ce) in elements
  int G;
  int epoch;
};

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static constexpr int VEC = 4;
  static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
    f[0] = __builtin_bit_cast(float, v.x);
    f[1] = __builtin_bit_cast(float, v.y);
    f[2] = __builtin_bit_cast(float, v.z);
    f[3] = __builtin_bit_cast(float, v.w);
  }
  static __device__ __forceinline__ uint4 pack(const float* f) {
    return uint4{__builtin_bit_cast(uint32_t, f[0]), __builtin_bit_cast(uint32_t, f[1]),
                 __builtin_bit_cast(uint32_t, f[2]), __builtin_bit_cast(uint32_t, f[3])};
  }
  static __device__ __forceinline__ float to_f(float x) { return x; }
  static __device__ __forceinline__ float from_f(float x) { return x; }
};
template <>
struct Elem<__bf16> {
  static constexpr int VEC = 8;
  static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
    const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      f[2 * e] = __builtin_bit_cast(float, x[e] << 16);
      f[2 * e + 1] = __builtin_bit_cast(float, x[e] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ uint4 pack(const float* f) {
    typedef __attribute__((ext_vector_type(2))) __bf16 b2;
    uint32_t x[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      b2 v = {(__bf16)f[2 * e], (__bf16)f[2 * e + 1]};
      x[e] = __builtin_bit_cast(uint32_t, v);
    }
    return uint4{x[0], x[1], x[2], x[3]};
  }
  static __device__ __forceinline__ float to_f(__bf16 x) { return (float)x; }
  static __device__ __forceinline__ __bf16 from_f(float x) { return (__bf16)x; }
};
template <>
struct Elem<_Float16> {
  static constexpr int VEC = 8;
  static __device__ __forceinline__ void unpack(const uint4& v, float* f) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const h2 h = __builtin_bit_cast(h2, x[e]);
      f[2 * e] = (float)h[0];
      f[2 * e + 1] = (float)h[1];
    }
  }
  static __device__ __forceinline__ uint4 pack(const float* f) {
    typedef __attribute__((ext_vector_type(2))) _Float16 h2;
    uint32_t x[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      h2 v = {(_Float16)f[2 * e], (_Float16)f[2 * e + 1]};
      x[e] = __builtin_bit_cast(uint32_t, v);
    }
    return uint4{x[0], x[1], x[2], x[3]};
  }
  static __device__ __forceinline__ float to_f(_Float16 x) { return (float)x; }
  static __device__ __forceinline__ _Float16 from_f(float x) { return (_Float16)x; }
};

// Per-call view of the double-buffered objects, resolved on entry of every workgroup from the call
// number (tdist/call_state.hpp).  It is a separate small value on purpose: writing the selected
// pointers back into the large, dynamically indexed argument block would force a private copy of the
// whole block into scratch memory.
struct ArCall {
  void* ws;
  int* flags;
  int epoch;
  int call;
};

__device__ __forceinline__ ArCall resolve_call(const ArParams& p) {
  __shared__ int s_call;
  const int call = call_begin(p.state, &s_call);
  return ArCall{call_parity(call) ? p.ws1 : p.ws0, call_parity(call) ? p.flags1 : p.flags0, call_epoch(call), call};
}

__device__ __forceinline__ void finish_call(const ArParams& p, int call) { call_end(p.state, call); }

// Raise flag[phase][rank][c] on every peer, then wait for flag[phase][s][c] of every peer s.
// The payload went out with write-through stores, so publishing is drain + barrier; only a push
// that had to fall back to plain stores (`need_release`, workgroup-uniform) pays a release fence.
__device__ __forceinline__ void exchange_flags(const ArParams& p, const ArCall& cv, int phase, int c, bool need_release) {
  const int w = p.ctx.num_ranks, rank = p.ctx.rank;
  if (need_release) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  drain_stores();
  __syncthreads();
  const int t = threadIdx.x;
  if (t < w && t != rank) {
    int* remote = symm_at(p.ctx, cv.flags, t) + (phase * w + rank) * p.G + c;
    store_relaxed_system(remote, cv.epoch);
    const bool dead = load_relaxed_agent(p.err) != 0;
    if (dead || !spin_until_eq(cv.flags + (phase * w + t) * p.G + c, cv.epoch, SCOPE_SYS)) {
      if (!dead) atomicOr(p.err, 1);
    }
  }
  if (t < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// out[e] = sum over ranks s (ascending) of (s == rank ? own[e] : slot_s[e]) for e in [e0, e1)
template <typename T>
__device__ __forceinline__ void sum_slices(const ArParams& p, const T* own, const T* slots, long long slot_stride,
                                           T* dst, long long e0, long long e1) {
  constexpr int VEC = Elem<T>::VEC;
  const int w = p.ctx.num_ranks, rank = p.ctx.rank;
  const long long nvec = (e1 - e0) / VEC;
  for (long long i = threadIdx.x; i < nvec; i += blockDim.x) {
    const long long e = e0 + i * VEC;
    float acc[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) acc[k] = 0.f;
    for (int s = 0; s < w; ++s) {
      const T* src = (s == rank) ? own + e : slots + s * slot_stride + e;
      float f[VEC];
      Elem<T>::unpack(*(const uint4*)src, f);
#pragma unroll
      for (int k = 0; k < VEC; ++k) acc[k] += f[k];
    }
    *(uint4*)(dst + e) = Elem<T>::pack(acc);
  }
  for (long long e = e0 + nvec * VEC + threadIdx.x; e < e1; e += blockDim.x) {
    float acc = 0.f;
    for (int s = 0; s < w; ++s) acc += Elem<T>::to_f((s == rank) ? own[e] : slots[s * slot_stride + e]);
    dst[e] = Elem<T>::from_f(acc);
  }
}

template <typename T>
__global__ __launch_bounds__(512) void allreduce_one_shot_kernel(ArParams p) {
  constexpr int VEC = Elem<T>::VEC;
  const ArCall cv = resolve_call(p);
  const int call = cv.call;
  const int w = p.ctx.num_ranks, rank = p.ctx.rank, c = blockIdx.x;
  const long long per = ((p.n + p.G - 1) / p.G + VEC - 1) / VEC * VEC;
  const long long e0 = min((long long)c * per, p.n), e1 = min(e0 + per, p.n);
  const T* in = (const T*)p.in;
  T* ws = (T*)cv.ws;
  bool plain = false;
  for (int i = 1; i < w; ++i) {
    const int peer = (rank + i) % w;
    T* dst = symm_at(p.ctx, ws, peer) + (long long)rank * p.cap + e0;
    plain |= shmem::copy_bytes_wt_sys(dst, in + e0, (size_t)(e1 - e0) * sizeof(T), threadIdx.x, blockDim.x);
  }
  exchange_flags(p, cv, 0, c, plain);
  sum_slices<T>(p, in, ws, p.cap, (T*)p.out, e0, e1);
  finish_call(p, call);
}

template <typename T>
__global__ __launch_bounds__(512) void allreduce_two_shot_kernel(ArParams p) {
  constexpr int VEC = Elem<T>::VEC;
  const ArCall cv = resolve_call(p);
  const int call = cv.call;
  const int w = p.ctx.num_ranks, rank = p.ctx.rank, c = blockIdx.x;
  const long long ce = p.cap;                                   // chunk stride (elements, multiple of VEC)
  const long long sub = ((ce + p.G - 1) / p.G + VEC - 1) / VEC * VEC;
  const long long s0 = min((long long)c * sub, ce), s1 = min(s0 + sub, ce);
  const T* in = (const T*)p.in;
  T* recv = (T*)cv.ws;                         // [w][ce]
  T* gather = recv + (long long)w * ce;       // [w][ce]
  // phase 1: reduce-scatter pushes
  bool plain = false;
  for (int i = 1; i < w; ++i) {
    const int peer = (rank + i) % w;
    const long long g0 = min(peer * ce + s0, p.n), g1 = min(peer * ce + s1, p.n);
    if (g1 > g0) {
      T* dst = symm_at(p.ctx, recv, peer) + (long long)rank * ce + s0;
      plain |= shmem::copy_bytes_wt_sys(dst, in + g0, (size_t)(g1 - g0) * sizeof(T), threadIdx.x, blockDim.x);
    }
  }
  exchange_flags(p, cv, 0, c, plain);
  plain = false;
  // reduce own chunk's sub-slice into the local gather buffer, then broadcast it
  {
    const long long g0 = min(rank * ce + s0, p.n), g1 = min(rank * ce + s1, p.n);
    if (g1 > g0) {
      // sum_slices indexes own[e] / slots[s*stride + e]: shift bases so e runs over [s0, s0 + len)
      sum_slices<T>(p, in + rank * ce, recv, ce, gather + (long long)rank * ce, s0, s0 + (g1 - g0));
      __syncthreads();
      for (int i = 1; i < w; ++i) {
        const int peer = (rank + i) % w;
        T* dst = symm_at(p.ctx, gather, peer) + (long long)rank * ce + s0;
        plain |= shmem::copy_bytes_wt_sys(dst, gather + (long long)rank * ce + s0, (size_t)(g1 - g0) * sizeof(T),
                                          threadIdx.x, blockDim.x);
      }
    }
  }
  exchange_flags(p, cv, 1, c, plain);
  // phase 2: every chunk's sub-slice is now in the local gather buffer
  for (int s = 0; s < w; ++s) {
    const long long g0 = min(s * ce + s0, p.n), g1 = min(s * ce + s1, p.n);
    if (g1 > g0)
      shmem::copy_bytes((T*)p.out + g0, gather + (long long)s * ce + s0, (size_t)(g1 - g0) * sizeof(T), threadIdx.x,
                        blockDim.x);
  }
  finish_call(p, call);
}

// dtype: 0 bf16, 1 fp16, 2 fp32.  method: 1 one-shot, 2 two-shot.
// ws0/ws1, flags0/flags1: the two parity buffers (this rank's symmetric addresses).  state: two
// int32 in device memory, zeroed once when the context is created, never touched by the host again.
TDIST_API int tdist_allreduce(const tdist::SymmCtx* ctx, const void* in, void* out, void* ws0, void* ws1, int* flags0,
                              int* flags1, int* state, int* err, long long n, long long cap, int G, int dtype,
                              int method, void* stream) {
  TDIST_REQUIRE(n > 0 && G > 0 && G <= 1024, "bad n / G");
  TDIST_REQUIRE((((uintptr_t)in | (uintptr_t)out | (uintptr_t)ws0 | (uintptr_t)ws1) & 15) == 0,
                "buffers must be 16-byte aligned");
  TDIST_REQUIRE(state && flags0 && flags1, "missing call state / flags");
  ArParams p;
  p.ctx = *ctx;
  p.in = in;
  p.out = out;
  p.ws = nullptr;
  p.flags = nullptr;
  p.ws0 = ws0;
  p.ws1 = ws1;
  p.flags0 = flags0;
  p.flags1 = flags1;
  p.state = state;
  p.err = err;
  p.n = n;
  p.cap = cap;
  p.G = G;
  p.epoch = 0;
  hipStream_t s = (hipStream_t)stream;
#define LAUNCH(T)                                                                              \
  do {                                                                                         \
    if (method == 1)                                                                           \
      hipLaunchKernelGGL(allreduce_one_shot_kernel<T>, dim3(G), dim3(512), 0, s, p);           \
    else                                                                                       \
      hipLaunchKernelGGL(allreduce_two_shot_kernel<T>, dim3(G), dim3(512), 0, s, p);           \
  } while (0)
  if (dtype == 0)
    LAUNCH(__bf16);
  else if (dtype == 1)
    LAUNCH(_Float16);
  else
    LAUNCH(float);
#undef LAUNCH
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
