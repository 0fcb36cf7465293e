// Tutorial 05 as a user kernel: reduce-scatter = scatter (push) + local reduce.
//
//   partial : local [w][n] fp32    - this rank's contribution to every rank's output slice
//   scatter : symmetric [w][n]     - slot s of rank d receives rank s's contribution to d's slice
//   flags   : symmetric [w] ints   - flags[s] == epoch  <=>  slot s has landed
//   out     : local [n]            - sum over the w slots
//
// Grid = w + R workgroups.  Workgroup d < w pushes partial[d] into rank d's slot [rank] and notifies it;
// workgroups w .. w+R-1 wait for all w flags and reduce a share of the n elements each.  The pushing workgroups
// never wait, so the waiting ones can only be waiting for other ranks (reference:
// tutorials/05-intra-node-reduce-scatter.py:63-222, scatter over copy engines + ring reduce).
#include "tdist/primitives.hpp"

using namespace tdist;

extern "C" __global__ void reduce_scatter_user(SymmCtx ctx, const float* partial, float* scatter, int* flags, float* out,
                                               int n, int epoch, int* err) {
  const int w = num_ranks(ctx), me = rank(ctx);
  if ((int)blockIdx.x < w) {
    const int d = (me + 1 + blockIdx.x) % w;            // start with the right neighbour: spreads the links
    const float4* src = (const float4*)(partial + (long long)d * n);
    float4* dst = (float4*)(symm_at(ctx, scatter, d) + (long long)me * n);
    for (int i = threadIdx.x; i < n / 4; i += blockDim.x) dst[i] = src[i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) notify(ctx, flags + me, d, epoch, SIGNAL_SET, COMM_INTRA_NODE);
    return;
  }
  const int token = wait(flags, w, SCOPE_SYS, SEM_ACQUIRE, epoch, err);
  const float* slots = consume_token(scatter, token);
  const int R = gridDim.x - w, r = blockIdx.x - w;
  for (int i = r * blockDim.x + threadIdx.x; i < n; i += R * blockDim.x) {
    float acc = 0.f;
    for (int s = 0; s < w; ++s) acc += slots[(long long)s * n + i];
    out[i] = acc;
  }
}
