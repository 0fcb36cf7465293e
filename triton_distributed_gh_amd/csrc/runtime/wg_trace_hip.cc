// Process-wide sink of the per-workgroup event trace (tdist/wg_trace.hpp).
#include "tdist/common.hpp"
#include "tdist/wg_trace.hpp"

namespace {
tdist::WgTrace g_sink{nullptr, 0};
}

namespace tdist {
WgTrace wg_trace_sink() { return g_sink; }
}  // namespace tdist

// buf: device memory of at least grid * cap * 2 int64, zero-filled; nullptr switches tracing off.
TDIST_API int tdist_debug_set_wg_trace(void* buf, int cap) {
  TDIST_REQUIRE(buf == nullptr || cap > 0, "trace capacity must be positive");
  g_sink.buf = (long long*)buf;
  g_sink.cap = buf ? cap : 0;
  return 0;
}
