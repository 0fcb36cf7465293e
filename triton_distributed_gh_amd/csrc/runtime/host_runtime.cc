// CPU-only runtime: POSIX shared-memory symmetric heap + signal primitives.
//
// Lets the whole Python stack (symmetric tensors, notify / wait, barriers, op contexts) run with
// world_size >= 1 on a machine without a GPU: every rank maps every other rank's arena from
// /dev/shm, and signals are std::atomic words inside those arenas.  This is the "world_size=1
// notify/wait + reference GEMM correctness on CPU" configuration; the reference has no such mode
// (its tests need real GPUs, reference: python/triton_dist/test/amd/test_distributed-notify-wait.py).
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <string>
#include <thread>

#define TDH_API extern "C" __attribute__((visibility("default")))

static thread_local std::string g_err;
TDH_API const char* tdh_last_error() { return g_err.c_str(); }

static int fail(const char* what) {
  g_err = std::string(what) + ": " + strerror(errno);
  return -1;
}

// Create (create != 0) or open a named shared-memory arena of `bytes` and map it.
TDH_API int tdh_shm_map(const char* name, uint64_t bytes, int create, void** out) {
  int fd = shm_open(name, create ? (O_CREAT | O_RDWR) : O_RDWR, 0600);
  if (fd < 0) return fail("shm_open");
  if (create && ftruncate(fd, (off_t)bytes) != 0) {
    close(fd);
    return fail("ftruncate");
  }
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) return fail("mmap");
  *out = p;
  return 0;
}

TDH_API int tdh_shm_unmap(void* p, uint64_t bytes) { return munmap(p, bytes) == 0 ? 0 : fail("munmap"); }
TDH_API int tdh_shm_unlink(const char* name) { return shm_unlink(name) == 0 ? 0 : fail("shm_unlink"); }

// ---- signals -----------------------------------------------------------------------------------
// sig_op: 1 = set, 2 = add (same numeric values as the device-side SignalOp enum).
TDH_API void tdh_notify32(int32_t* p, int32_t v, int sig_op) {
  auto* a = reinterpret_cast<std::atomic<int32_t>*>(p);
  if (sig_op == 2)
    a->fetch_add(v, std::memory_order_release);
  else
    a->store(v, std::memory_order_release);
}
TDH_API void tdh_notify64(int64_t* p, int64_t v, int sig_op) {
  auto* a = reinterpret_cast<std::atomic<int64_t>*>(p);
  if (sig_op == 2)
    a->fetch_add(v, std::memory_order_release);
  else
    a->store(v, std::memory_order_release);
}
TDH_API int32_t tdh_load32(const int32_t* p) {
  return reinterpret_cast<const std::atomic<int32_t>*>(p)->load(std::memory_order_acquire);
}
TDH_API int64_t tdh_load64(const int64_t* p) {
  return reinterpret_cast<const std::atomic<int64_t>*>(p)->load(std::memory_order_acquire);
}

// cmp: 0 EQ, 1 NE, 2 GT, 3 LE, 4 LT, 5 GE.  Returns 0 when satisfied, 1 on timeout.
template <typename T>
static bool cmp_ok(T v, int cmp, T ref) {
  switch (cmp) {
    case 0: return v == ref;
    case 1: return v != ref;
    case 2: return v > ref;
    case 3: return v <= ref;
    case 4: return v < ref;
    default: return v >= ref;
  }
}

template <typename T>
static int wait_impl(const T* p, int cmp, T ref, int n, int64_t timeout_ms) {
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) {
    auto* a = reinterpret_cast<const std::atomic<T>*>(p + i);
    int spins = 0;
    while (!cmp_ok(a->load(std::memory_order_acquire), cmp, ref)) {
      if (++spins > 64) {
        std::this_thread::yield();
        if ((spins & 1023) == 0) {
          auto dt = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0);
          if (timeout_ms >= 0 && dt.count() > timeout_ms) return 1;
        }
      }
    }
  }
  return 0;
}

TDH_API int tdh_wait32(const int32_t* p, int n, int cmp, int32_t ref, int64_t timeout_ms) {
  return wait_impl<int32_t>(p, cmp, ref, n, timeout_ms);
}
TDH_API int tdh_wait64(const int64_t* p, int n, int cmp, int64_t ref, int64_t timeout_ms) {
  return wait_impl<int64_t>(p, cmp, ref, n, timeout_ms);
}

// Slot-matrix barrier over shared memory: slots[r] is rank r's int32[num_ranks] array.
// Same protocol as the device barrier: raise slot[rank] at every peer, lower own slots.
TDH_API int tdh_barrier_all(int32_t** slots, int rank, int num_ranks, int64_t timeout_ms) {
  auto t0 = std::chrono::steady_clock::now();
  auto expired = [&]() {
    auto dt = std::chrono::duration_cast<std::chrono::milliseconds>(std::chrono::steady_clock::now() - t0);
    return timeout_ms >= 0 && dt.count() > timeout_ms;
  };
  for (int p = 0; p < num_ranks; ++p) {
    auto* a = reinterpret_cast<std::atomic<int32_t>*>(slots[p] + rank);
    int32_t exp = 0;
    while (!a->compare_exchange_weak(exp, 1, std::memory_order_release, std::memory_order_relaxed)) {
      exp = 0;
      std::this_thread::yield();
      if (expired()) return 1;
    }
  }
  for (int p = 0; p < num_ranks; ++p) {
    auto* a = reinterpret_cast<std::atomic<int32_t>*>(slots[rank] + p);
    int32_t exp = 1;
    while (!a->compare_exchange_weak(exp, 0, std::memory_order_acquire, std::memory_order_relaxed)) {
      exp = 1;
      std::this_thread::yield();
      if (expired()) return 1;
    }
  }
  return 0;
}
