// SHMEM-style device API over xGMI peer mappings (no rocSHMEM dependency).
//
// Mirrors the device surface the reference exposes to kernels through `libshmem_device`
// (reference: python/triton_dist/language/extra/libshmem_device.py:28-340; NVSHMEM wrappers
// shmem/nvshmem_bind/runtime/nvshmem_wrapper.cu; rocSHMEM wrappers
// shmem/rocshmem_bind/runtime/rocshmem_wrapper.cc:27-31).  Inside one xGMI node every peer's
// arena is mapped into this process, so:
//   remote_ptr      = base translation (symm_at)
//   putmem/getmem   = vectorised 16-byte copies by a thread / wave / workgroup
//   *_nbi           = identical data path; completion is forced by quiet()/fence()
//   signal_op       = system-scope atomic store / add on the peer's word
//   signal_wait_until = bounded spin with a comparison operator
//   quiet / fence   = system-scope release fence + drain of outstanding stores
//   barrier_all*    = sense-free slot-matrix barrier (one slot per (dst, src) pair)
#pragma once
#include "tdist/cache_policy.hpp"
#include "tdist/primitives.hpp"

namespace tdist {
namespace shmem {

enum Cmp : int { CMP_EQ = 0, CMP_NE = 1, CMP_GT = 2, CMP_LE = 3, CMP_LT = 4, CMP_GE = 5 };
// Numeric values follow NVSHMEM's (libshmem_device.py:310-311).
enum SigOp : int { SIGNAL_SET_OP = 9, SIGNAL_ADD_OP = 10 };

__device__ __forceinline__ int my_pe(const SymmCtx& c) { return c.rank; }
__device__ __forceinline__ int n_pes(const SymmCtx& c) { return c.num_ranks; }
__device__ __forceinline__ int team_my_pe(const SymmCtx& c, int /*team*/) { return c.rank; }
__device__ __forceinline__ int team_n_pes(const SymmCtx& c, int /*team*/) { return c.num_ranks; }
__device__ __forceinline__ int team_translate_pe(const SymmCtx&, int, int pe, int) { return pe; }

template <typename T>
__device__ __forceinline__ T* remote_ptr(const SymmCtx& c, T* local, int pe) {
  return symm_at(c, local, pe);
}
// No multicast object exists on xGMI; callers must loop over peers.
template <typename T>
__device__ __forceinline__ T* remote_mc_ptr(const SymmCtx&, int, T*) { return nullptr; }

__device__ __forceinline__ void int_p(const SymmCtx& c, int* dest, int value, int pe) {
  store_relaxed_system(symm_at(c, dest, pe), value);
}

__device__ __forceinline__ void fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void quiet() { fence(); }

// ---------------------------------------------------------------------------------------------
// Cooperative byte copy: `nthreads` threads (index `tid`) move `bytes` from src to dst.
// 16-byte path when both pointers and the size allow it, tail handled bytewise.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void copy_bytes(void* dst, const void* src, size_t bytes, int tid, int nthreads) {
  const uintptr_t d = reinterpret_cast<uintptr_t>(dst), s = reinterpret_cast<uintptr_t>(src);
  if (((d | s) & 15) == 0) {
    const size_t n16 = bytes >> 4;
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    size_t i = tid;
    // 4 independent 16-byte loads in flight per thread before the stores.
    for (; i + 3 * (size_t)nthreads < n16; i += 4 * (size_t)nthreads) {
      uint4 a = s4[i], b = s4[i + nthreads], c = s4[i + 2 * (size_t)nthreads], e = s4[i + 3 * (size_t)nthreads];
      d4[i] = a;
      d4[i + nthreads] = b;
      d4[i + 2 * (size_t)nthreads] = c;
      d4[i + 3 * (size_t)nthreads] = e;
    }
    for (; i < n16; i += nthreads) d4[i] = s4[i];
    const size_t done = n16 << 4;
    for (size_t j = done + tid; j < bytes; j += nthreads)
      reinterpret_cast<unsigned char*>(dst)[j] = reinterpret_cast<const unsigned char*>(src)[j];
  } else if (((d | s) & 3) == 0 && (bytes & 3) == 0) {
    const size_t n4 = bytes >> 2;
    for (size_t i = tid; i < n4; i += nthreads)
      reinterpret_cast<uint32_t*>(dst)[i] = reinterpret_cast<const uint32_t*>(src)[i];
  } else {
    for (size_t j = tid; j < bytes; j += nthreads)
      reinterpret_cast<unsigned char*>(dst)[j] = reinterpret_cast<const unsigned char*>(src)[j];
  }
}

// Same copy with write-through system-scope stores for the 16-byte body (see cache_policy.hpp):
// the payload leaves the L2 at once, so the publisher needs no release fence (a system-scope
// release walks the whole L2), only `drain_stores()` in every storing wave + a barrier before the
// flag.  Returns true when some bytes had to go out as plain stores (unaligned operands or a tail
// shorter than 16 bytes) - then the caller still owes a release fence before it signals.  The
// result depends only on the arguments, so it is uniform over the cooperating threads.
__device__ __forceinline__ bool copy_bytes_wt_sys(void* dst, const void* src, size_t bytes, int tid, int nthreads) {
  const uintptr_t d = reinterpret_cast<uintptr_t>(dst), s = reinterpret_cast<uintptr_t>(src);
  if (((d | s) & 15) != 0) {
    copy_bytes(dst, src, bytes, tid, nthreads);
    return bytes != 0;
  }
  const size_t n16 = bytes >> 4;
  const uint4* s4 = reinterpret_cast<const uint4*>(src);
  uint4* d4 = reinterpret_cast<uint4*>(dst);
  size_t i = tid;
  for (; i + 3 * (size_t)nthreads < n16; i += 4 * (size_t)nthreads) {
    const uint4 a = s4[i], b = s4[i + nthreads], c = s4[i + 2 * (size_t)nthreads], e = s4[i + 3 * (size_t)nthreads];
    st_wt128_sys(d4 + i, a);
    st_wt128_sys(d4 + i + nthreads, b);
    st_wt128_sys(d4 + i + 2 * (size_t)nthreads, c);
    st_wt128_sys(d4 + i + 3 * (size_t)nthreads, e);
  }
  for (; i < n16; i += nthreads) st_wt128_sys(d4 + i, s4[i]);
  const size_t done = n16 << 4;
  for (size_t j = done + tid; j < bytes; j += nthreads)
    reinterpret_cast<unsigned char*>(dst)[j] = reinterpret_cast<const unsigned char*>(src)[j];
  return done != bytes;
}

// put: dest is a symmetric address (translated to `pe`), source is local.
__device__ __forceinline__ void putmem_nbi(const SymmCtx& c, void* dest, const void* src, size_t bytes, int pe) {
  copy_bytes(symm_at(c, reinterpret_cast<char*>(dest), pe), src, bytes, 0, 1);
}
__device__ __forceinline__ void putmem(const SymmCtx& c, void* dest, const void* src, size_t bytes, int pe) {
  putmem_nbi(c, dest, src, bytes, pe);
  quiet();
}
__device__ __forceinline__ void putmem_nbi_warp(const SymmCtx& c, void* dest, const void* src, size_t bytes, int pe) {
  copy_bytes(symm_at(c, reinterpret_cast<char*>(dest), pe), src, bytes, lane_id(), 64);
}
__device__ __forceinline__ void putmem_warp(const SymmCtx& c, void* dest, const void* src, size_t bytes, int pe) {
  putmem_nbi_warp(c, dest, src, bytes, pe);
  quiet();
}
__device__ __forceinline__ void putmem_nbi_block(const SymmCtx& c, void* dest, const void* src, size_t bytes, int pe) {
  copy_bytes(symm_at(c, reinterpret_cast<char*>(dest), pe), src, bytes, threadIdx.x, blockDim.x);
}
__device__ __forceinline__ void putmem_block(const SymmCtx& c, void* dest, const void* src, size_t bytes, int pe) {
  putmem_nbi_block(c, dest, src, bytes, pe);
  quiet();
  __syncthreads();
}

// get: source is a symmetric address on `pe`, dest is local.
__device__ __forceinline__ void getmem_nbi(const SymmCtx& c, void* dest, const void* src, size_t bytes, int pe) {
  copy_bytes(dest, symm_at(c, reinterpret_cast<char*>(const_cast<void*>(src)), pe), bytes, 0, 1);
}
__device__ __forceinline__ void getmem(const SymmCtx& c, void* dest, const void* src, size_t bytes, int pe) {
  getmem_nbi(c, dest, src, bytes, pe);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void getmem_nbi_warp(const SymmCtx& c, void* dest, const void* src, size_t bytes, int pe) {
  copy_bytes(dest, symm_at(c, reinterpret_cast<char*>(const_cast<void*>(src)), pe), bytes, lane_id(), 64);
}
__device__ __forceinline__ void getmem_warp(const SymmCtx& c, void* dest, const void* src, size_t bytes, int pe) {
  getmem_nbi_warp(c, dest, src, bytes, pe);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void getmem_nbi_block(const SymmCtx& c, void* dest, const void* src, size_t bytes, int pe) {
  copy_bytes(dest, symm_at(c, reinterpret_cast<char*>(const_cast<void*>(src)), pe), bytes, threadIdx.x, blockDim.x);
}
__device__ __forceinline__ void getmem_block(const SymmCtx& c, void* dest, const void* src, size_t bytes, int pe) {
  getmem_nbi_block(c, dest, src, bytes, pe);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
}

// signal_op: single thread.  Data written before it by THIS thread's wave is made visible first.
__device__ __forceinline__ void signal_op(const SymmCtx& c, uint64_t* sig_addr, uint64_t signal, int sig_op, int pe) {
  uint64_t* dst = symm_at(c, sig_addr, pe);
  fence();
  if (sig_op == SIGNAL_ADD_OP)
    (void)atom_add_relaxed_system(dst, signal);
  else
    store_relaxed_system(dst, signal);
}
__device__ __forceinline__ void signal_op(const SymmCtx& c, int* sig_addr, int signal, int sig_op, int pe) {
  int* dst = symm_at(c, sig_addr, pe);
  fence();
  if (sig_op == SIGNAL_ADD_OP)
    (void)atom_add_relaxed_system(dst, signal);
  else
    store_relaxed_system(dst, signal);
}

template <typename T>
__device__ __forceinline__ bool cmp_ok(T v, int cmp, T ref) {
  switch (cmp) {
    case CMP_EQ: return v == ref;
    case CMP_NE: return v != ref;
    case CMP_GT: return v > ref;
    case CMP_LE: return v <= ref;
    case CMP_LT: return v < ref;
    default: return v >= ref;
  }
}

// Spin on a LOCAL signal word.  Returns the value that satisfied the comparison
// (or the last value seen when the spin budget ran out).
template <typename T>
__device__ __forceinline__ T signal_wait_until(const T* sig_addr, int cmp, T cmp_val,
                                               uint64_t budget = TDIST_SPIN_TIMEOUT_TICKS) {
  const uint64_t t0 = wall_ticks();
  int backoff = 0;
  T v;
  while (true) {
    v = load_relaxed_system(sig_addr);
    if (cmp_ok(v, cmp, cmp_val)) break;
    if (++backoff > 16) {
      __builtin_amdgcn_s_sleep(2);
      if ((backoff & 255) == 0 && wall_ticks() - t0 > budget) break;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  return v;
}

// put + signal.  The block/warp variants copy cooperatively, drain, and one thread signals.
template <typename S>
__device__ __forceinline__ void putmem_signal_nbi(const SymmCtx& c, void* dest, const void* src, size_t bytes,
                                                  S* sig_addr, S signal, int sig_op, int pe) {
  putmem_nbi(c, dest, src, bytes, pe);
  signal_op(c, sig_addr, signal, sig_op, pe);
}
template <typename S>
__device__ __forceinline__ void putmem_signal(const SymmCtx& c, void* dest, const void* src, size_t bytes,
                                              S* sig_addr, S signal, int sig_op, int pe) {
  putmem_signal_nbi(c, dest, src, bytes, sig_addr, signal, sig_op, pe);
}
template <typename S>
__device__ __forceinline__ void putmem_signal_nbi_warp(const SymmCtx& c, void* dest, const void* src, size_t bytes,
                                                       S* sig_addr, S signal, int sig_op, int pe) {
  putmem_nbi_warp(c, dest, src, bytes, pe);
  // every lane drains its own stores; lane 0 then publishes
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane_id() == 0) signal_op(c, sig_addr, signal, sig_op, pe);
}
template <typename S>
__device__ __forceinline__ void putmem_signal_warp(const SymmCtx& c, void* dest, const void* src, size_t bytes,
                                                   S* sig_addr, S signal, int sig_op, int pe) {
  putmem_signal_nbi_warp(c, dest, src, bytes, sig_addr, signal, sig_op, pe);
}
template <typename S>
__device__ __forceinline__ void putmem_signal_nbi_block(const SymmCtx& c, void* dest, const void* src, size_t bytes,
                                                        S* sig_addr, S signal, int sig_op, int pe) {
  putmem_nbi_block(c, dest, src, bytes, pe);
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) signal_op(c, sig_addr, signal, sig_op, pe);
}
template <typename S>
__device__ __forceinline__ void putmem_signal_block(const SymmCtx& c, void* dest, const void* src, size_t bytes,
                                                    S* sig_addr, S signal, int sig_op, int pe) {
  putmem_signal_nbi_block(c, dest, src, bytes, sig_addr, signal, sig_op, pe);
}

// ---------------------------------------------------------------------------------------------
// Barrier over all ranks of the node.  `slots` is a symmetric int32[num_ranks] array in the signal
// arena, zero-initialised.  Rank r raises slot[r] on every peer (0 -> 1, release/system), then
// lowers its own slots (1 -> 0, acquire) as each peer arrives.  Self-resetting, so it may be
// called back-to-back (protocol of the reference's `barrier_all_ipc`,
// python/triton_dist/kernels/amd/common_ops.py:45-59).  Lane i of ONE wave serves peer i.
// Returns false on timeout.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ bool barrier_all_wave(const SymmCtx& c, int* slots) {
  const int lane = lane_id();
  bool ok = true;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane < c.num_ranks) {
    int* remote = symm_at(c, slots, lane) + c.rank;
    const uint64_t t0 = wall_ticks();
    int spins = 0;
    while (atom_cas_release_relaxed_system(remote, 0, 1) != 0) {
      __builtin_amdgcn_s_sleep(1);
      if (((++spins) & 255) == 0 && wall_ticks() - t0 > TDIST_SPIN_TIMEOUT_TICKS) { ok = false; break; }
    }
    int* mine = slots + lane;
    spins = 0;
    while (ok && atom_cas_acquire_relaxed_system(mine, 1, 0) != 1) {
      __builtin_amdgcn_s_sleep(1);
      if (((++spins) & 255) == 0 && wall_ticks() - t0 > TDIST_SPIN_TIMEOUT_TICKS) { ok = false; break; }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
  return __all(ok);
}

// Workgroup form: wave 0 runs the protocol, everyone leaves together.
__device__ __forceinline__ bool barrier_all_block(const SymmCtx& c, int* slots) {
  __shared__ int ok_s;
  __syncthreads();
  if (threadIdx.x < 64) {
    bool ok = barrier_all_wave(c, slots);
    if (threadIdx.x == 0) ok_s = ok ? 1 : 0;
  }
  __syncthreads();
  return ok_s != 0;
}

}  // namespace shmem
}  // namespace tdist
