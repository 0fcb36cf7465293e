// Issue cost of memory instructions placed between back-to-back v_mfma_f32_16x16x32_bf16 at ONE wave per SIMD
// (256-thread workgroups, one per CU, 160 KiB... no: 64 KiB of LDS so that only one workgroup fits a CU with the
// register budget below).  Every wave runs ITER blocks of 64 MFMAs on 16 independent accumulators with NF fillers
// of one kind spread evenly through the block; cycles per block = (s_memtime delta) / ITER.  The difference to the
// NF = 0 build is what the fillers cost the wave that issues them.
//   kind 1  global_load_lds_dwordx4 (per-lane 64-bit address)      kind 2  buffer_load_dwordx4 ... lds
//   kind 3  global_load_dwordx4 -> VGPR                            kind 4  buffer_load_dwordx4 -> VGPR
//   kind 5  ds_write_b128                                          kind 6  ds_read_b128
//   kind 7  s_barrier (all four waves of the workgroup)
// Sources sit in a 1 MiB buffer (L2 resident after the first block).  Every load with a register destination is
// an ordinary C++ load whose value is consumed at the end of its block, so the compiler keeps the waits and never
// reuses a register that still has data in flight (an inline-asm load left in flight across the loop edge had its
// destination reused as an address: memory fault); LDS-DMA pieces are drained with a counted vmcnt per block.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int KIND, int NF>
__global__ __launch_bounds__(256) void bench(const char* __restrict__ src, float* out, long long* cycles, int iters,
                                              unsigned src_bytes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.001f * (lane + i)); b[i] = (__bf16)(0.002f * (lane - i)); }
  // per-lane source: this wave's 1 KiB piece inside a 64 KiB window that moves with the iteration
  const unsigned lane_off = (unsigned)(wave * 16384 + lane * 16);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)src_bytes, 0x00020000);
  char* lds_dst = smem + wave * 16384;
  u32x4 sink[NF > 0 ? NF : 1];
#pragma unroll
  for (int i = 0; i < (NF > 0 ? NF : 1); ++i) sink[i] = u32x4{0, 0, 0, 0};
  u32x4 wdata = {1u, 2u, 3u, (unsigned)lane};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  // two blocks per trip: the register loads of one block are consumed at the end of the NEXT block, so the
  // compiler's wait for them never exposes their latency (it only has to keep NF loads in flight)
  u32x4 setA[NF > 0 ? NF : 1], setB[NF > 0 ? NF : 1];
#pragma unroll
  for (int i = 0; i < (NF > 0 ? NF : 1); ++i) setA[i] = setB[i] = u32x4{0, 0, 0, 0};
  auto block = [&](int it, u32x4 (&cur)[NF > 0 ? NF : 1]) {
    const unsigned win = (unsigned)(it & 15) * 65536u;     // 16 windows x 64 KiB = 1 MiB
    int f = 0;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i & 15]) : "v"(a), "v"(b));
      if constexpr (NF > 0) {
        if (i % (64 / NF) == 0 && f < NF) {
          const unsigned piece = (unsigned)f * 1024u;
          if constexpr (KIND == 1) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + win + lane_off + piece),
                                             (__attribute__((address_space(3))) void*)(lds_dst + piece), 16, 0, 0);
          } else if constexpr (KIND == 2) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(lds_dst + piece), 16,
                                                     lane_off + piece, win, 0, 0);
          } else if constexpr (KIND == 3) {
            cur[f] = *(const u32x4*)(src + win + lane_off + piece);
          } else if constexpr (KIND == 4) {
            cur[f] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(lane_off + piece), (int)win, 0);
          } else if constexpr (KIND == 5) {
            *(u32x4*)(lds_dst + piece + lane * 16) = wdata;
          } else if constexpr (KIND == 6) {
            cur[f] = *(const u32x4*)(lds_dst + piece + lane * 16);
          } else if constexpr (KIND == 7) {
            __builtin_amdgcn_s_barrier();
          }
          ++f;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (KIND == 1 || KIND == 2) {               // one block of LDS-DMA pieces stays in flight
      if constexpr (NF == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      else if constexpr (NF == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if constexpr (NF == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if constexpr (NF == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    }
  };
  auto consume = [&](u32x4 (&set)[NF > 0 ? NF : 1]) {
    if constexpr (KIND == 3 || KIND == 4 || KIND == 6) {
#pragma unroll
      for (int i = 0; i < (NF > 0 ? NF : 1); ++i) sink[i] ^= set[i];
    }
  };
  for (int it = 0; it < iters; it += 2) {
    block(it, setA);
    consume(setB);
    block(it + 1, setB);
    consume(setA);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) r += acc[i][0] + acc[i][3];
  unsigned sk = 0;
#pragma unroll
  for (int i = 0; i < (NF > 0 ? NF : 1); ++i) sk ^= sink[i][0] ^ sink[i][3];
  if (KIND == 1 || KIND == 2) sk ^= *(const unsigned*)(smem + wave * 16384 + lane * 4);
  out[blockIdx.x * 256 + tid] = r + (float)(sk & 1);
  if (lane == 0) cycles[blockIdx.x * 4 + wave] = t1 - t0;
}

template <int KIND, int NF>
static double run(const char* src, float* out, long long* cyc, long long* hcyc, int grid, int iters, unsigned bytes) {
  CHECK(hipFuncSetAttribute((const void*)bench<KIND, NF>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304));
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL((bench<KIND, NF>), dim3(grid), dim3(256), 98304, 0, src, out, cyc, iters, bytes);
    CHECK(hipDeviceSynchronize());
  }
  CHECK(hipMemcpy(hcyc, cyc, sizeof(long long) * grid * 4, hipMemcpyDeviceToHost));
  double sum = 0;
  for (int i = 0; i < grid * 4; ++i) sum += (double)hcyc[i];
  return sum / (grid * 4) / iters;
}

int main() {
  const int grid = 256, iters = 2000;
  const unsigned bytes = 1u << 20;
  char* src; float* out; long long* cyc;
  CHECK(hipMalloc(&src, bytes + 65536));
  CHECK(hipMemset(src, 1, bytes + 65536));
  CHECK(hipMalloc(&out, sizeof(float) * grid * 256));
  CHECK(hipMalloc(&cyc, sizeof(long long) * grid * 4));
  long long* hcyc = (long long*)malloc(sizeof(long long) * grid * 4);
  const double base = run<0, 0>(src, out, cyc, hcyc, grid, iters, bytes);
  printf("baseline: %.1f s_memtime ticks per 64 MFMAs (%.2f per MFMA)\n", base, base / 64);
#define ROW(K, name) do { \
    const double c2 = run<K, 2>(src, out, cyc, hcyc, grid, iters, bytes); \
    const double c4 = run<K, 4>(src, out, cyc, hcyc, grid, iters, bytes); \
    const double c8 = run<K, 8>(src, out, cyc, hcyc, grid, iters, bytes); \
    const double c16 = run<K, 16>(src, out, cyc, hcyc, grid, iters, bytes); \
    printf("%-28s extra ticks per filler: NF=2 %6.1f  NF=4 %6.1f  NF=8 %6.1f  NF=16 %6.1f   (block: %.0f %.0f %.0f %.0f)\n", name, \
           (c2 - base) / 2, (c4 - base) / 4, (c8 - base) / 8, (c16 - base) / 16, c2, c4, c8, c16); fflush(stdout); } while (0)
  ROW(1, "global_load_lds_dwordx4");
  ROW(2, "buffer_load_dwordx4 lds");
  ROW(3, "global_load_dwordx4 -> vgpr");
  ROW(4, "buffer_load_dwordx4 -> vgpr");
  ROW(5, "ds_write_b128");
  ROW(6, "ds_read_b128");
  ROW(7, "s_barrier");
  return 0;
}
