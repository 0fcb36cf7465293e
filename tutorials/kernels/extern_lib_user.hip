// A user-side DEVICE library: functions a kernel file declares `extern "C" __device__` and calls.
// tools.compile_hip(kernel, extern_libs=[this file]) links it into the kernel's code object.
#include <hip/hip_runtime.h>

extern "C" __device__ float user_silu(float a) { return a / (1.f + expf(-a)); }

extern "C" __device__ int user_hash(int a, int b) { return a * 31 + b; }
