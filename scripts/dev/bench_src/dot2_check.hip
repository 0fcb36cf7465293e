// v_dot2_f32_bf16 semantics check: c = a.x * b.x + a.y * b.y + 1
#include <hip/hip_runtime.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
extern "C" __global__ void dot2_check(const unsigned* a, const unsigned* b, float* c) {
  float acc = 1.0f;
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a[threadIdx.x]), __builtin_bit_cast(bf16x2, b[threadIdx.x]), acc, false);
  c[threadIdx.x] = acc;
}
