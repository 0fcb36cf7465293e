// Per-workgroup event trace for the persistent kernels (a development tool: where does a workgroup's
// time go between tickets, waits, main loops and epilogues?).
//
// The host arms a process-wide sink with tdist_debug_set_wg_trace(buffer, events_per_workgroup); every
// launch of a traced kernel after that records, per workgroup, up to that many (tag << 32 | argument,
// 100 MHz timestamp) pairs into buffer[blockIdx.x][event][2] (int64).  With no sink armed the cost is one
// uniform branch per event.  Events are written by thread 0 with ordinary vector stores.
#pragma once
#include <hip/hip_runtime.h>

namespace tdist {

struct WgTrace {
  long long* buf;   // [grid][cap][2], zero-filled by the host; nullptr = tracing off
  int cap;
};

// the process-wide sink (runtime/wg_trace_hip.cc)
WgTrace wg_trace_sink();

struct WgTraceCursor {
  long long* p;
  long long* end;
};

__device__ __forceinline__ WgTraceCursor wg_trace_begin(const WgTrace& t) {
  if (!t.buf) return WgTraceCursor{nullptr, nullptr};
  long long* p = t.buf + (long long)blockIdx.x * t.cap * 2;
  return WgTraceCursor{p, p + (long long)t.cap * 2};
}

__device__ __forceinline__ void wg_trace(WgTraceCursor& c, int tag, int arg) {
  if (c.p && c.p < c.end) {
    if (threadIdx.x == 0) {
      c.p[0] = ((long long)tag << 32) | (unsigned)arg;
      c.p[1] = (long long)wall_clock64();
    }
    c.p += 2;
  }
}

}  // namespace tdist
