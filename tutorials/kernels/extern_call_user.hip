// User kernel that calls an EXTERNAL device library (extern_lib_user.hip) and the package's own
// primitives: out[i] = user_silu(x[i]) on every rank's symmetric buffer, then one notify per rank.
#include <hip/hip_runtime.h>

#include "tdist/primitives.hpp"

extern "C" __device__ float user_silu(float a);
extern "C" __device__ int user_hash(int a, int b);

extern "C" __global__ void silu_and_tag(tdist::SymmCtx ctx, const float* x, float* out, int* tags, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) {
    out[i] = user_silu(x[i]);
    tags[i] = user_hash(i, tdist::rank(ctx));
  }
}
