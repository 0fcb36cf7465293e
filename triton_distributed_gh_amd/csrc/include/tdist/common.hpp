// Shared host/device definitions for the MI355X (gfx950) op library.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TDIST_MAX_RANKS 16
#define TDIST_MAX_EXTRA_SEGMENTS 7
#define TDIST_WAVE 64

#define TDIST_API extern "C" __attribute__((visibility("default")))

// Every host entry point returns 0 on success or a negative / hipError code.
#define TDIST_HIP_TRY(expr)                                                      \
  do {                                                                           \
    hipError_t _e = (expr);                                                      \
    if (_e != hipSuccess) {                                                      \
      (void)hipGetLastError(); /* reported through the return code, not left for the next caller */ \
      tdist_set_last_error(#expr, (int)_e, __FILE__, __LINE__);                  \
      return (int)_e;                                                            \
    }                                                                            \
  } while (0)

#define TDIST_REQUIRE(cond, msg)                                                 \
  do {                                                                           \
    if (!(cond)) {                                                               \
      tdist_set_last_error(msg, -1, __FILE__, __LINE__);                         \
      return -1;                                                                 \
    }                                                                            \
  } while (0)

extern "C" void tdist_set_last_error(const char* what, int code, const char* file, int line);

namespace tdist {

// Device-visible description of the symmetric heap: the same offset inside every
// rank's arena names "the same" symmetric object.  Passed to kernels by value.
struct SymmCtx {
  int rank;
  int num_ranks;
  // heap_base[r] is rank r's arena mapped into THIS process (own arena for r == rank).
  uint64_t heap_base[TDIST_MAX_RANKS];
  // sig_base[r]: small fine-grained arena used for flags / signals.
  uint64_t sig_base[TDIST_MAX_RANKS];
  uint64_t heap_bytes;
  uint64_t sig_bytes;
  // Growth segments of the data arena (each heap_bytes long): one hipIpc mapping cannot reach
  // 2 GiB, so a larger heap is a list of equal segments added on demand.
  int num_extra;
  int pad_;
  uint64_t extra_base[TDIST_MAX_EXTRA_SEGMENTS][TDIST_MAX_RANKS];
  // Virtual-memory-management arena (runtime_hip.cc: tdist_vmm_*): one contiguous range of vmm_bytes
  // per rank, any size, holds the tensors that do not fit a hipIpc segment.  0 = not created.
  uint64_t vmm_bytes;
  uint64_t vmm_base[TDIST_MAX_RANKS];
};

// Spin budget: every device-side wait gives up after this many 100 MHz ticks
// (s_memrealtime) and raises the error word so a lost peer can never hang the GPU.
#ifndef TDIST_SPIN_TIMEOUT_TICKS
#define TDIST_SPIN_TIMEOUT_TICKS (1500ull * 1000ull * 1000ull)  // 15 s
#endif

}  // namespace tdist
