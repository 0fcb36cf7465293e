// gfx950 bf16/fp16 MFMA tile engine:  C[M,N] = A[M,K] * B[N,K]^T, fp32 accumulate.
//
// One workgroup (512 threads = 8 waves, 2 along M x 4 along N) computes one 256x256 output tile
// with v_mfma_f32_16x16x32_{bf16,f16}.  K is walked in 64-deep steps through a 2 x 64 KiB LDS
// ring filled by direct-to-LDS loads (global_load_lds_dwordx4) that stay in flight across raw
// s_barriers (counted s_waitcnt vmcnt, never 0 in steady state).  Each K-step is cut into four
// phases = four 64x32 accumulator quadrants per wave (16 MFMAs each); each phase also stages one
// 16 KiB "half tile" seven half-tiles ahead.  Waves 4-7 run one barrier behind waves 0-3 so that on
// every SIMD one wave is in its MFMA segment while its partner is in its LDS/DMA segment.
//
// LDS image of a half tile (128 rows x 64 k, 16-bit elements): sixteen 1-KiB sub-tiles of
// [16 rows][32 k]; inside a sub-tile byte = row*64 + kbyte, with bit 5 XOR-ed by bit 9 (rows 8-15
// swap their 32-byte halves) so one ds_read_b128 wave-instruction touches all 16 slots of the
// 256-B bank row once.  The DMA destination is lane-linear, so the same XOR is applied to the
// per-lane SOURCE address instead.
//
// Half-tile order inside a K-step buffer is B0, A0, B1, A1 where A0/A1 hold, for both wave rows,
// the first/second 64 rows of that wave row's 128, and B0/B1 hold, for all four wave columns,
// the first/second 32 columns of that wave column's 64.  The order is what makes the write-after-
// read distances legal: a slot is restaged two phases after its last ds_read, or one phase after
// when those reads were retired by an lgkmcnt wait before the reading phase's first barrier (B0).
//
// Role in the reference: the Triton `tl.dot` GEMM bodies of
// python/triton_dist/kernels/amd/all_gather_gemm.py:127-245 and gemm_reduce_scatter.py:125-229
// (tile 128x256x64, 3 stages, autotuned) - redesigned here for 64-wide waves / 160 KiB LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace tdist {
namespace gemm {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

constexpr int BM = 256;
constexpr int BN = 256;
constexpr int BK = 64;
constexpr int THREADS = 512;
constexpr int HALF_BYTES = 16384;
constexpr int BUF_BYTES = 65536;
constexpr int OFF_B0 = 0;
constexpr int OFF_A0 = 16384;
constexpr int OFF_B1 = 32768;
constexpr int OFF_A1 = 49152;
constexpr int C_STRIDE = 528;                 // bytes per staged output row (512 + 16 pad)
constexpr int LDS_BYTES = BM * C_STRIDE;      // 135168 >= 2 * BUF_BYTES

#define TDIST_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define TDIST_GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// Operand types of the tile engine.  The two 16-bit types keep the values the `IS_BF16` template
// parameters of the kernels convert to (false = f16, true = bf16); the 8-bit types are new:
// OCP fp8 e4m3 / e5m2 on v_mfma_scale_f32_16x16x128_f8f6f4 (unit scales: twice the bf16 rate per clock)
// and int8 on v_mfma_i32_16x16x64_i8 (int32 accumulate, twice the bf16 rate).  An 8-bit operand matrix is
// handed to the engine as a 16-bit matrix with half as many columns - K, lda, ldb in TileArgs count
// 16-bit units - so staging, LDS image and fragment reads are byte for byte those of the 16-bit types;
// only the MFMA that consumes the fragments differs (reference: the dtype-generic `tl.dot` of
// python/triton_dist/kernels/amd/all_gather_gemm.py:173,211-221 with int32 accumulators for int8).
enum : int { DT_F16 = 0, DT_BF16 = 1, DT_FP8 = 2, DT_BF8 = 3, DT_I8 = 4 };
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(8))) int i32x8;

template <int DT>
struct Frag;
template <>
struct Frag<DT_BF16> {
  using T = bf16x8;
  static constexpr bool K128 = false;
  static __device__ __forceinline__ float val(float raw) { return raw; }
  static __device__ __forceinline__ f32x4 add(f32x4 a, f32x4 b) { return a + b; }
  static __device__ __forceinline__ f32x4 mfma(T a, T b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
  // accumulator pinned to the accumulation registers (the register allocator otherwise moves
  // 256-register accumulators back and forth between the two halves of the register file)
  static __device__ __forceinline__ void mfma_acc(T a, T b, f32x4& c) {
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(uint32_t, v);
  }
};
template <>
struct Frag<DT_F16> {
  using T = f16x8;
  static constexpr bool K128 = false;
  static __device__ __forceinline__ float val(float raw) { return raw; }
  static __device__ __forceinline__ f32x4 add(f32x4 a, f32x4 b) { return a + b; }
  static __device__ __forceinline__ f32x4 mfma(T a, T b, f32x4 c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ void mfma_acc(T a, T b, f32x4& c) {
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) {
    typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
    f16x2 v = {(_Float16)lo, (_Float16)hi};
    return __builtin_bit_cast(uint32_t, v);
  }
};

// fp8 (FMT 0 = e4m3, 1 = e5m2): a lane's operand of the 128-deep MFMA is 32 bytes = the 16-byte fragments
// of BOTH 64-byte k halves of a K-step; which 32 of the step's 128 k a lane group holds differs from the
// instruction's natural order, but identically for both operands, so the dot products are the same.
template <int FMT>
struct FragF8 {
  using T = i32x4;
  static constexpr bool K128 = true;
  static __device__ __forceinline__ f32x4 mfma2(T a0, T a1, T b0, T b1, f32x4 c) {
    const i32x8 a = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
    const i32x8 b = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
    // scale bytes 127 = 2^0 (E8M0) for both operands
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, FMT, FMT, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
  }
  static __device__ __forceinline__ float val(float raw) { return raw; }
  static __device__ __forceinline__ f32x4 add(f32x4 a, f32x4 b) { return a + b; }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return Frag<DT_BF16>::pack2(lo, hi); }
};
template <>
struct Frag<DT_FP8> : FragF8<0> {};
template <>
struct Frag<DT_BF8> : FragF8<1> {};

// int8: the accumulator registers hold int32 bit patterns (f32x4 is only their register type).
template <>
struct Frag<DT_I8> {
  using T = i32x4;
  static constexpr bool K128 = false;
  static __device__ __forceinline__ f32x4 mfma(T a, T b, f32x4 c) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, __builtin_bit_cast(i32x4, c), 0, 0, 0));
  }
  static __device__ __forceinline__ float val(float raw) { return (float)__builtin_bit_cast(int, raw); }
  static __device__ __forceinline__ f32x4 add(f32x4 a, f32x4 b) {
    return __builtin_bit_cast(f32x4, __builtin_bit_cast(i32x4, a) + __builtin_bit_cast(i32x4, b));
  }
  static __device__ __forceinline__ uint32_t pack2(float lo, float hi) { return Frag<DT_BF16>::pack2(val(lo), val(hi)); }
};

// Description of one 256x256 tile job.  A and B are 16-bit row-major with K contiguous.
struct TileArgs {
  const uint16_t* A;   // first row of the A operand this tile reads (row index 0 == tile row 0 .. clamp below)
  const uint16_t* B;
  long long lda;       // elements
  long long ldb;
  int a_rows;          // valid rows in A starting at A (rows >= a_rows are clamped, results masked)
  int b_rows;          // valid rows in B starting at B
  int K;               // multiple of 64, >= 128
  // Optional row gather for A (grouped / MoE GEMMs): tile row r reads A row a_gather[r]
  // (negative = padding: reads row 0, the caller discards that output row).  Because the
  // direct-to-LDS load takes a per-lane SOURCE address, the gather costs nothing extra.
  const int* a_gather = nullptr;
};

// Accumulators of one wave: 8 (m) x 4 (n) tiles of 16x16, stored transposed (lane: m = lane&15,
// n = 4*(lane>>4) + reg) because the MFMA is issued as B*A^T.
template <int DT>
struct Acc {
  f32x4 v[8][4];
};

template <bool IS_BF16>
__device__ __forceinline__ void mainloop_256(const TileArgs& t, char* smem, Acc<IS_BF16>& acc) {
  using F = Frag<IS_BF16>;
  using V = typename F::T;
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int wr = w >> 2, wc = w & 3;
  const bool late_group = w >= 4;

  // ---- staging source addresses (per lane) ------------------------------------------------
  const int hr = w * 16 + (l >> 2);                       // row inside a half tile, 0..127
  const int kchunk = (l & 3) ^ (((l >> 5) & 1) << 1);     // 16-byte chunk inside the 64-byte row segment
  int arow0 = (hr >> 6) * 128 + (hr & 63);                // A0 ; A1 = +64
  int arow1 = arow0 + 64;
  int brow0 = (hr >> 5) * 64 + (hr & 31);                 // B0 ; B1 = +32
  int brow1 = brow0 + 32;
  arow0 = arow0 < t.a_rows ? arow0 : t.a_rows - 1;
  arow1 = arow1 < t.a_rows ? arow1 : t.a_rows - 1;
  brow0 = brow0 < t.b_rows ? brow0 : t.b_rows - 1;
  brow1 = brow1 < t.b_rows ? brow1 : t.b_rows - 1;
  const uint16_t* pa0 = t.A + (long long)arow0 * t.lda + kchunk * 8;
  const uint16_t* pa1 = t.A + (long long)arow1 * t.lda + kchunk * 8;
  const uint16_t* pb0 = t.B + (long long)brow0 * t.ldb + kchunk * 8;
  const uint16_t* pb1 = t.B + (long long)brow1 * t.ldb + kchunk * 8;

  const int nt = t.K / BK;
  const int stage_sub = w * 2048;  // this wave's two sub-tiles inside a half tile

  // Stage half tile `kind` (0=B0 1=A0 2=B1 3=A1) of K-step `kt` into ring buffer kt&1.
  auto stage = [&](int kt, auto kind_c) {
    constexpr int kind = decltype(kind_c)::value;
    if (kt < nt) {
      const uint16_t* src = kind == 0 ? pb0 : (kind == 1 ? pa0 : (kind == 2 ? pb1 : pa1));
      src += (long long)kt * BK;
      char* dst = smem + (kt & 1) * BUF_BYTES + kind * HALF_BYTES + stage_sub;
      __builtin_amdgcn_global_load_lds(TDIST_GLB_PTR(src), TDIST_LDS_PTR(dst), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(TDIST_GLB_PTR(src + 32), TDIST_LDS_PTR(dst + 1024), 16, 0, 0);
    }
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;

  // ---- fragment read addresses (per lane) -------------------------------------------------
  const int lane_off = (((l & 15) * 64) + ((l >> 4) * 16)) ^ ((((l & 15) >> 3) & 1) << 5);
  const char* a_rd = smem + wr * 8192 + lane_off;   // + OFF_A{0,1} + (mt*2+ks)*1024
  const char* b_rd = smem + wc * 4096 + lane_off;   // + OFF_B{0,1} + (nt*2+ks)*1024

#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc.v[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prologue: 7 half tiles in flight, first K-step landed ------------------------------
  stage(0, K0{}); stage(0, K1{}); stage(0, K2{}); stage(0, K3{});
  stage(1, K0{}); stage(1, K1{}); stage(1, K2{});
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if (late_group) __builtin_amdgcn_s_barrier();

  V a[4][2], b0[2][2], b1[2][2];

  for (int kt = 0; kt < nt; ++kt) {
    const int boff = (kt & 1) * BUF_BYTES;
    // ---------------- phase 0: quadrant (m0, n0) ----------------
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) b0[n][ks] = *(const V*)(b_rd + boff + OFF_B0 + (n * 2 + ks) * 1024);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) a[m][ks] = *(const V*)(a_rd + boff + OFF_A0 + (m * 2 + ks) * 1024);
    stage(kt + 1, K3{});
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc.v[m][n] = F::mfma(b0[n][ks], a[m][ks], acc.v[m][n]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---------------- phase 1: quadrant (m0, n1) ----------------
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) b1[n][ks] = *(const V*)(b_rd + boff + OFF_B1 + (n * 2 + ks) * 1024);
    stage(kt + 2, K0{});
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc.v[m][2 + n] = F::mfma(b1[n][ks], a[m][ks], acc.v[m][2 + n]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---------------- phase 2: quadrant (m1, n1) ----------------
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) a[m][ks] = *(const V*)(a_rd + boff + OFF_A1 + (m * 2 + ks) * 1024);
    stage(kt + 2, K1{});
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc.v[4 + m][2 + n] = F::mfma(b1[n][ks], a[m][ks], acc.v[4 + m][2 + n]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---------------- phase 3: quadrant (m1, n0) ----------------
    stage(kt + 2, K2{});
    // Retire the next K-step: everything but the three newest half tiles (6 loads).  In the last
    // two K-steps nothing new was issued, so drain completely / not at all.
    if (kt + 2 < nt) {
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if (kt + 2 == nt) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc.v[4 + m][n] = F::mfma(b0[n][ks], a[m][ks], acc.v[4 + m][n]);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
  if (!late_group) __builtin_amdgcn_s_barrier();
}

// Variant with TWO phases per K-step (32 MFMAs per compute segment instead of 16): half as many
// barrier hand-overs between the two wave groups of a SIMD, at the price of a shorter prefetch
// distance.  Phase 0 reads B0, B1, A0 and computes the upper 64 rows of the wave tile, phase 1
// reads A1 and computes the lower 64 rows.  Staging: phase 0 of step t issues A0, A1 of step t+1,
// phase 1 issues B0, B1 of step t+2 (their slots were read in phase 0 and retired by the counted
// lgkmcnt before that phase's barrier; A1's reads are retired before phase 1's barrier).
//
// PERM_B permutes which B row (output column) each accumulator register stands for, so that the
// epilogue can store straight from registers: register r of accumulator tile n in lane group
// g = lane >> 4 then holds column 32*(n>>1) + 8*g + 4*(n&1) + r of the wave's 64 columns, i.e. tiles
// (0,1) and (2,3) each give a lane 8 neighbouring columns = one 16-byte store (epilogue_256_direct).
// The permutation is applied to the per-lane DMA source row and costs nothing.
//
// OPT (schedule knobs, bit mask): 1 = in phase 0 the A fragments are waited for one 16-row block at
// a time inside the MFMA segment (they are consumed in the order they were read) instead of all at
// once after the barrier, 2 = no s_setprio around the MFMA segment.  Measured at 4096x8192x8192 bf16
// (MI355X, interleaved with the default): OPT 3 is 0.5-0.8 % faster, each bit alone is neutral.
// Issuing the LDS-DMA loads between the MFMAs of the compute segment instead of in the read segment
// was measured 10 % slower and is not kept.  Timing experiments with this loop (results discarded):
// without any staging in the loop the GEMM takes 306-311 us instead of 353 us; with the staging kept
// but its vmcnt waits removed, or its sources pinned to K-step 0 (cache hits), it stays within 3 % of
// the full kernel - the cost is the issue of the eight LDS-DMA pieces per wave and K-step itself,
// not HBM latency or bandwidth.
//
// DS (how the eight LDS-DMA pieces a wave issues per K-step are split over the two read segments;
// the read segment of phase 0 carries 16 fragment reads, that of phase 1 only 8):
//   0 = 4 + 4: phase 0 of step t stages A0, A1 of step t+1, phase 1 stages B0, B1 of step t+2;
//   1 = 2 + 6: phase 0 stages A0 of step t+1, phase 1 stages A1 of step t+1 and B0, B1 of step t+2.  A1(t+1) is
//       then waited for in phase 0 of step t+1 (counted vmcnt before that phase's first barrier) instead of in
//       phase 1 of step t; its slot (other ring buffer) was last read in phase 1 of step t-1.
//   3 = 2 + 6, everything one whole K-step ahead: phase 0 of step t stages A1 of step t+1 (its slot was read in
//       phase 1 of step t-1 ... no: see below), phase 1 stages A0, B0, B1 of step t+2.  Precisely: phase 1 of step t
//       refills the slots phase 0 of step t has just read (A0, B0, B1 of ring buffer t&1) with step t+2, and phase
//       0 of step t+1 refills the slot phase 1 of step t has read (A1 of buffer t&1) with step t+2.  Every piece is
//       in flight for a full K-step (two MFMA segments) before the counted wait that retires it, against one MFMA
//       segment for the A pieces of DS 0.  Phase 0 then retires ALL its fragment reads before its first barrier
//       (the other wave group restages the A0 slot right after that barrier).
// OPT bit 4: issue a segment's LDS-DMA pieces before its fragment reads instead of after them.
template <int IS_BF16, bool PERM_B = false, int OPT = 0, int DS = 0>
__device__ __forceinline__ void mainloop_256_2ph(const TileArgs& t, char* smem, Acc<IS_BF16>& acc) {
  using F = Frag<IS_BF16>;
  using V = typename F::T;
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int wr = w >> 2, wc = w & 3;
  const bool late_group = w >= 4;
  const int hr = w * 16 + (l >> 2);
  const int kchunk = (l & 3) ^ (((l >> 5) & 1) << 1);
  int arow0 = (hr >> 6) * 128 + (hr & 63);
  int arow1 = arow0 + 64;
  // half-tile row hr = 32*wave column + 16*(accumulator tile & 1) + i, i = 4*g + r
  int brow0 = PERM_B ? (hr >> 5) * 64 + ((hr & 15) >> 2) * 8 + ((hr >> 4) & 1) * 4 + (hr & 3)
                     : (hr >> 5) * 64 + (hr & 31);
  int brow1 = brow0 + 32;
  arow0 = arow0 < t.a_rows ? arow0 : t.a_rows - 1;
  arow1 = arow1 < t.a_rows ? arow1 : t.a_rows - 1;
  brow0 = brow0 < t.b_rows ? brow0 : t.b_rows - 1;
  brow1 = brow1 < t.b_rows ? brow1 : t.b_rows - 1;
  if (t.a_gather) {
    const int g0 = t.a_gather[arow0], g1 = t.a_gather[arow1];
    arow0 = g0 < 0 ? 0 : g0;
    arow1 = g1 < 0 ? 0 : g1;
  }
  const uint16_t* pa0 = t.A + (long long)arow0 * t.lda + kchunk * 8;
  const uint16_t* pa1 = t.A + (long long)arow1 * t.lda + kchunk * 8;
  const uint16_t* pb0 = t.B + (long long)brow0 * t.ldb + kchunk * 8;
  const uint16_t* pb1 = t.B + (long long)brow1 * t.ldb + kchunk * 8;
  const int nt = t.K / BK;
  const int stage_sub = w * 2048;
  // (Staging through the buffer form - `buffer_load_dwordx4 ... offen lds` with the tile bases in two
  // scalar descriptors and the K walk in the scalar offset, no vector address arithmetic per piece -
  // was measured too: 350.7 vs 354.4 us and 352.0 vs 350.1 us against the global form, i.e. noise.
  // Its immediate offset also shifts the LDS destination, so it cannot address the second k half.)
  auto stage = [&](int kt, auto kind_c) {
    constexpr int kind = decltype(kind_c)::value;
    if (kt < nt) {
      const uint16_t* src = kind == 0 ? pb0 : (kind == 1 ? pa0 : (kind == 2 ? pb1 : pa1));
      src += (long long)kt * BK;
      char* dst = smem + (kt & 1) * BUF_BYTES + kind * HALF_BYTES + stage_sub;
      __builtin_amdgcn_global_load_lds(TDIST_GLB_PTR(src), TDIST_LDS_PTR(dst), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(TDIST_GLB_PTR(src + 32), TDIST_LDS_PTR(dst + 1024), 16, 0, 0);
    }
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;
  const int lane_off = (((l & 15) * 64) + ((l >> 4) * 16)) ^ ((((l & 15) >> 3) & 1) << 5);
  const char* a_rd = smem + wr * 8192 + lane_off;
  const char* b_rd = smem + wc * 4096 + lane_off;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc.v[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage(0, K0{}); stage(0, K1{}); stage(0, K2{}); stage(0, K3{});
  if constexpr (DS == 3) {
    stage(1, K1{}); stage(1, K0{}); stage(1, K2{});
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  } else {
    stage(1, K0{}); stage(1, K2{});
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();
  if (late_group) __builtin_amdgcn_s_barrier();

  V a[4][2], b0[2][2], b1[2][2];
  constexpr bool PROG = (OPT & 1) != 0, NOPRIO = (OPT & 2) != 0, DMA_FIRST = (OPT & 4) != 0;
  // MFMAs of one 16-row block m of the current 64-row half (both k halves, all four column tiles)
  // (the IS_BF16 parameter is the operand type DT_*: 8-bit types pair the two k halves in one 128-deep MFMA)
  auto mfma_row = [&](int m, int mo) {
    if constexpr (F::K128) {
#pragma unroll
      for (int n = 0; n < 2; ++n) acc.v[mo + m][n] = F::mfma2(b0[n][0], b0[n][1], a[m][0], a[m][1], acc.v[mo + m][n]);
#pragma unroll
      for (int n = 0; n < 2; ++n) acc.v[mo + m][2 + n] = F::mfma2(b1[n][0], b1[n][1], a[m][0], a[m][1], acc.v[mo + m][2 + n]);
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int n = 0; n < 2; ++n) acc.v[mo + m][n] = F::mfma(b0[n][ks], a[m][ks], acc.v[mo + m][n]);
#pragma unroll
        for (int n = 0; n < 2; ++n) acc.v[mo + m][2 + n] = F::mfma(b1[n][ks], a[m][ks], acc.v[mo + m][2 + n]);
      }
    }
  };
  for (int kt = 0; kt < nt; ++kt) {
    const int boff = (kt & 1) * BUF_BYTES;
    // ---------------- phase 0: rows 0..63 of the wave tile ----------------
    if constexpr (DMA_FIRST) {
      stage(kt + 1, K1{});
      if constexpr (DS == 0) stage(kt + 1, K3{});
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) b0[n][ks] = *(const V*)(b_rd + boff + OFF_B0 + (n * 2 + ks) * 1024);
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) b1[n][ks] = *(const V*)(b_rd + boff + OFF_B1 + (n * 2 + ks) * 1024);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) a[m][ks] = *(const V*)(a_rd + boff + OFF_A0 + (m * 2 + ks) * 1024);
    if constexpr (DS == 3) {
      // A1 of step kt+1 into the slot phase 1 of step kt-1 read; then retire A1 of THIS step (younger: A0, B0,
      // B1 of step kt+1 and the piece just issued = 8 pieces, if step kt+1 exists)
      stage(kt + 1, K3{});
      if (kt + 1 < nt) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    } else if constexpr (!DMA_FIRST) {
      stage(kt + 1, K1{});
      if constexpr (DS == 0) stage(kt + 1, K3{});
    }
    if constexpr (DS == 1) {
      // A1 of THIS step (issued in phase 1 of the previous step, or in the prologue) must have landed
      // before the first barrier of this phase; younger: B0, B1, A0 of step kt+1 (6 pieces), if they exist
      if (kt + 1 < nt) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (DS == 3) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if constexpr (!PROG) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!NOPRIO) __builtin_amdgcn_s_setprio(1);
    if constexpr (PROG) {
      // reads were issued B0 x4, B1 x4, then A0 block by block: two more retire per 16-row block
      asm volatile("s_waitcnt lgkmcnt(6)" ::: "memory");
      mfma_row(0, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
      mfma_row(1, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
      mfma_row(2, 0);
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      mfma_row(3, 0);
    } else if constexpr (F::K128) {
#pragma unroll
      for (int m = 0; m < 4; ++m) mfma_row(m, 0);
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
          for (int n = 0; n < 2; ++n) acc.v[m][n] = F::mfma(b0[n][ks], a[m][ks], acc.v[m][n]);
#pragma unroll
          for (int n = 0; n < 2; ++n) acc.v[m][2 + n] = F::mfma(b1[n][ks], a[m][ks], acc.v[m][2 + n]);
        }
    }
    if constexpr (!NOPRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    // ---------------- phase 1: rows 64..127 of the wave tile ----------------
    if constexpr (DMA_FIRST) {
      if constexpr (DS == 1) stage(kt + 1, K3{});
      stage(kt + 2, K0{});
      stage(kt + 2, K2{});
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) a[m][ks] = *(const V*)(a_rd + boff + OFF_A1 + (m * 2 + ks) * 1024);
    if constexpr (DS == 3) {
      stage(kt + 2, K1{});
      stage(kt + 2, K0{});
      stage(kt + 2, K2{});
    } else if constexpr (!DMA_FIRST) {
      if constexpr (DS == 1) stage(kt + 1, K3{});
      stage(kt + 2, K0{});
      stage(kt + 2, K2{});
    }
    // retire what phase 0 of the next step reads (A0, B0, B1 of step kt+1): everything but the younger pieces -
    // DS 0: B0, B1 of step kt+2; DS 1: those and A1 of step kt+1; DS 3: A1 of step kt+1 and A0, B0, B1 of step kt+2
    if constexpr (DS == 3) {
      if (kt + 2 < nt) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else if (kt + 1 < nt) {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      }
    } else if (kt + 2 < nt) {
      if constexpr (DS == 1) {
        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      }
    } else if (kt + 1 < nt) {
      if constexpr (DS == 1) {
        asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    // (always complete here: the other wave group restages this buffer's A1 slot one barrier later)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    if constexpr (!NOPRIO) __builtin_amdgcn_s_setprio(1);
    if constexpr (F::K128) {
#pragma unroll
      for (int m = 0; m < 4; ++m) mfma_row(m, 4);
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
#pragma unroll
          for (int n = 0; n < 2; ++n) acc.v[4 + m][n] = F::mfma(b0[n][ks], a[m][ks], acc.v[4 + m][n]);
#pragma unroll
          for (int n = 0; n < 2; ++n) acc.v[4 + m][2 + n] = F::mfma(b1[n][ks], a[m][ks], acc.v[4 + m][2 + n]);
        }
    }
    if constexpr (!NOPRIO) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
  if (!late_group) __builtin_amdgcn_s_barrier();
}

// Convert accumulators to 16-bit, stage the 256x256 tile in LDS and hand 16-byte row pieces to
// `store(row_in_tile, col_in_tile, uint4)`; every thread calls store 16 times with rows
// (pass*16 + tid/32) and columns (tid%32)*8.  `get(m, v)` fills v[0..3] with the calling thread's
// fp32 values of accumulator tiles (m, 0..3) in the Acc layout.  Ends with a workgroup barrier so
// LDS may be reused.
template <int IS_BF16, class GetFn, class StoreFn>
__device__ __forceinline__ void epilogue_256_from(GetFn&& get, char* smem, StoreFn&& store) {
  using F = Frag<IS_BF16>;
  const int tid = threadIdx.x;
  const int w = tid >> 6, l = tid & 63;
  const int wr = w >> 2, wc = w & 3;
  // all waves are past their last fragment read (mainloop ends on a barrier)
  char* base = smem + (wr * 128 + (l & 15)) * C_STRIDE + (wc * 64 + (l >> 4) * 4) * 2;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    f32x4 v[4];
    get(m, v);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      uint2 p;
      p.x = F::pack2(v[n][0], v[n][1]);
      p.y = F::pack2(v[n][2], v[n][3]);
      *(uint2*)(base + m * 16 * C_STRIDE + n * 32) = p;
    }
  }
  __syncthreads();
  const int r0 = tid >> 5, cc = tid & 31;
#pragma unroll 4
  for (int pass = 0; pass < 16; ++pass) {
    const int row = pass * 16 + r0;
    const uint4 v = *(const uint4*)(smem + row * C_STRIDE + cc * 16);
    store(row, cc * 8, v);
  }
  __syncthreads();
}

// Epilogue for accumulators produced with PERM_B: no LDS, no barrier.  Every lane converts its
// values and hands sixteen 16-byte pieces to `store(row_in_tile, col_in_tile, uint4)`; one wave
// instruction covers 16 rows x 64 contiguous bytes.  Because LDS is not touched, a persistent kernel
// can issue the next tile's first loads right behind these stores and the write-out drains under
// the next main loop.  `get(m, v)` fills v[0..3] with the calling thread's fp32 values of accumulator tiles
// (m, 0..3) in the Acc layout (epilogue_256_direct: the accumulators themselves).
template <bool IS_BF16, class GetFn, class StoreFn>
__device__ __forceinline__ void epilogue_256_direct_from(GetFn&& get, StoreFn&& store) {
  using F = Frag<IS_BF16>;
  const int tid = threadIdx.x;
  const int w = tid >> 6, l = tid & 63;
  const int row = (w >> 2) * 128 + (l & 15);
  const int col = (w & 3) * 64 + (l >> 4) * 8;
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    f32x4 x[4];
    get(m, x);
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      uint4 v;
      v.x = F::pack2(x[2 * p][0], x[2 * p][1]);
      v.y = F::pack2(x[2 * p][2], x[2 * p][3]);
      v.z = F::pack2(x[2 * p + 1][0], x[2 * p + 1][1]);
      v.w = F::pack2(x[2 * p + 1][2], x[2 * p + 1][3]);
      store(row + m * 16, col + 32 * p, v);
    }
  }
}

template <bool IS_BF16, class StoreFn>
__device__ __forceinline__ void epilogue_256_direct(const Acc<IS_BF16>& acc, StoreFn&& store) {
  epilogue_256_direct_from<IS_BF16>(
      [&](int m, f32x4(&v)[4]) {
#pragma unroll
        for (int n = 0; n < 4; ++n) v[n] = acc.v[m][n];
      },
      store);
}

template <int IS_BF16, class StoreFn>
__device__ __forceinline__ void epilogue_256(const Acc<IS_BF16>& acc, char* smem, StoreFn&& store) {
  epilogue_256_from<IS_BF16>(
      [&](int m, f32x4(&v)[4]) {
#pragma unroll
        for (int n = 0; n < 4; ++n) v[n] = acc.v[m][n];
      },
      smem, store);
}

// ---------------------------------------------------------------------------------------------
// Four-wave form of the 256x256 tile: 256 threads, waves 2 (M) x 2 (N), each wave owns 128x128 of C
// (64 accumulator tiles = 256 registers, which with one wave per SIMD live in the accumulation
// half of the register file).  Per 32-deep K slab a wave reads 8 A + 8 B fragments for 64 MFMAs -
// 128 KiB of LDS reads per 64 of K for the workgroup instead of 192 KiB with eight 128x64 waves.
// There is no partner wave to hide a wave's LDS / DMA segment, so reads and DMA issue are
// interleaved with the MFMAs of the same wave:
//   * LDS is a ring of four 32 KiB slabs ([A: 16 sub-tiles of 16 rows x 32 k][B: 16 sub-tiles],
//     sub-tile image and XOR swizzle as above); wave w stages A and B sub-tiles 4w .. 4w+3 of every slab;
//   * fragments are double-buffered in registers: during the MFMAs of slab s the wave reads the
//     fragments of slab s+1 (first four MFMA rows) and issues the DMA of slab s+4 (last four rows)
//     into the buffer slab s has just left;
//   * one barrier per slab: before it every wave has retired its reads of slab s (lgkmcnt) and its
//     own DMA pieces of slab s+1 (counted vmcnt: slabs s+2, s+3 stay in flight), after it slab s+1
//     may be read and slab s's buffer restaged.
// Accumulator columns are permuted as with PERM_B (epilogue_256_4w_direct stores from registers).
constexpr int W4_THREADS = 256;
constexpr int W4_SLAB = 32768;
constexpr int W4_NS = 4;
constexpr int W4_LDS_BYTES = W4_NS * W4_SLAB;

template <bool IS_BF16>
struct Acc4 {
  f32x4 v[8][8];
};

template <bool IS_BF16, bool SPREAD = false>
__device__ __forceinline__ void mainloop_256_4w(const TileArgs& t, char* smem, Acc4<IS_BF16>& acc) {
  using F = Frag<IS_BF16>;
  using V = typename F::T;
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int wr = w >> 1, wc = w & 1;
  const int i = l >> 2;                                     // row inside a staged sub-tile
  const int kchunk = (l & 3) ^ (((l >> 5) & 1) << 1);
  const uint16_t* pa[4];
  const uint16_t* pb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ts = 4 * w + j;                               // sub-tile index inside the slab's A / B half
    int ar = ts * 16 + i;
    ar = ar < t.a_rows ? ar : t.a_rows - 1;
    // B sub-tile ts = 8 * wave column + n; its row i = 4g + r stands for output column
    // 32*(n>>1) + 8*g + 4*(n&1) + r of that wave column's 128
    int br = (ts >> 3) * 128 + ((ts & 7) >> 1) * 32 + (i >> 2) * 8 + (ts & 1) * 4 + (i & 3);
    br = br < t.b_rows ? br : t.b_rows - 1;
    pa[j] = t.A + (long long)ar * t.lda + kchunk * 8;
    pb[j] = t.B + (long long)br * t.ldb + kchunk * 8;
  }
  const int ns = t.K / 32;                                  // even, >= 4
  char* stage_dst = smem + w * 4096;
  auto stage_piece = [&](int s, auto jc) {
    constexpr int j = decltype(jc)::value;                  // 0..3 A, 4..7 B
    const uint16_t* src = (j < 4 ? pa[j & 3] : pb[j & 3]) + (long long)s * 32;
    char* dst = stage_dst + (s & 3) * W4_SLAB + (j < 4 ? 0 : 16384) + (j & 3) * 1024;
    __builtin_amdgcn_global_load_lds(TDIST_GLB_PTR(src), TDIST_LDS_PTR(dst), 16, 0, 0);
  };
  auto stage_slab = [&](int s) {
    stage_piece(s, std::integral_constant<int, 0>{}); stage_piece(s, std::integral_constant<int, 1>{});
    stage_piece(s, std::integral_constant<int, 2>{}); stage_piece(s, std::integral_constant<int, 3>{});
    stage_piece(s, std::integral_constant<int, 4>{}); stage_piece(s, std::integral_constant<int, 5>{});
    stage_piece(s, std::integral_constant<int, 6>{}); stage_piece(s, std::integral_constant<int, 7>{});
  };
  const int lane_off = (((l & 15) * 64) + ((l >> 4) * 16)) ^ ((((l & 15) >> 3) & 1) << 5);
  const char* a_rd = smem + wr * 8192 + lane_off;           // + slab buffer + m * 1024
  const char* b_rd = smem + 16384 + wc * 8192 + lane_off;   // + slab buffer + n * 1024
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 8; ++n) acc.v[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  V fa[2][8], fb[2][8];
  stage_slab(0); stage_slab(1); stage_slab(2); stage_slab(3);
  asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    fa[0][m] = *(const V*)(a_rd + m * 1024);
    fb[0][m] = *(const V*)(b_rd + m * 1024);
  }

  // STEADY: slab s+4 exists, so every read and every DMA piece is issued unconditionally - the
  // instruction stream has no branches and the compiler's own lgkmcnt bookkeeping stays exact
  // (fragments of slab s: 16 reads issued one slab ago; it waits for them with the new reads in flight).
  auto slab_body = [&](int s, auto pc, auto steady_c) {
    constexpr int P = decltype(pc)::value;
    constexpr bool STEADY = decltype(steady_c)::value;
    const bool more = STEADY || s + 1 < ns;
    if constexpr (STEADY) {
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else if (more) {
      if (s + 3 < ns) {
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      } else if (s + 2 < ns) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    // all of this wave's reads of slab s (issued during slab s-1) must be retired before the
    // barrier, because after it another wave may restage that buffer
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    const int nb = ((s + 1) & 3) * W4_SLAB;
    const bool restage = STEADY || s + 4 < ns;
    if constexpr (SPREAD) {
      // One memory instruction behind every second (reads) or fourth (LDS-DMA pieces) MFMA instead of
      // groups of four behind every eight: a 16x16x32 MFMA occupies the matrix pipe for 16 cycles but
      // the wave's issue port for 8, so a single filler per gap issues for free while a group of
      // them lets the pipe run dry.  Reads come B first (all eight are needed by row 0), then A in row order.
#pragma unroll
      for (int i = 0; i < 64; ++i) {
        const int m = i >> 3, n = i & 7;
        F::mfma_acc(fb[P][n], fa[P][m], acc.v[m][n]);
        if (i < 32 && (i & 1) == 0) {
          const int j = i >> 1;
          if (more) {
            if (j < 8)
              fb[P ^ 1][j] = *(const V*)(b_rd + nb + j * 1024);
            else
              fa[P ^ 1][j - 8] = *(const V*)(a_rd + nb + (j - 8) * 1024);
          }
        } else if (i >= 32 && (i & 3) == 0 && restage) {
          switch ((i - 32) >> 2) {
            case 0: stage_piece(s + 4, std::integral_constant<int, 0>{}); break;
            case 1: stage_piece(s + 4, std::integral_constant<int, 1>{}); break;
            case 2: stage_piece(s + 4, std::integral_constant<int, 2>{}); break;
            case 3: stage_piece(s + 4, std::integral_constant<int, 3>{}); break;
            case 4: stage_piece(s + 4, std::integral_constant<int, 4>{}); break;
            case 5: stage_piece(s + 4, std::integral_constant<int, 5>{}); break;
            case 6: stage_piece(s + 4, std::integral_constant<int, 6>{}); break;
            default: stage_piece(s + 4, std::integral_constant<int, 7>{}); break;
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    } else
#pragma unroll
    for (int m = 0; m < 8; ++m) {
      // row m of the wave tile first, then this row's share of the memory work: the compiler
      // (which cannot see the waits above) puts its own full lgkmcnt wait in front of the first
      // MFMA, so nothing new may be in flight at that point
#pragma unroll
      for (int n = 0; n < 8; ++n) F::mfma_acc(fb[P][n], fa[P][m], acc.v[m][n]);
      __builtin_amdgcn_sched_barrier(0);
      if (m < 4) {
        if (more) {
          fa[P ^ 1][2 * m] = *(const V*)(a_rd + nb + (2 * m) * 1024);
          fa[P ^ 1][2 * m + 1] = *(const V*)(a_rd + nb + (2 * m + 1) * 1024);
          fb[P ^ 1][2 * m] = *(const V*)(b_rd + nb + (2 * m) * 1024);
          fb[P ^ 1][2 * m + 1] = *(const V*)(b_rd + nb + (2 * m + 1) * 1024);
        }
      } else if (restage) {
        if (m == 4) { stage_piece(s + 4, std::integral_constant<int, 0>{}); stage_piece(s + 4, std::integral_constant<int, 1>{}); }
        if (m == 5) { stage_piece(s + 4, std::integral_constant<int, 2>{}); stage_piece(s + 4, std::integral_constant<int, 3>{}); }
        if (m == 6) { stage_piece(s + 4, std::integral_constant<int, 4>{}); stage_piece(s + 4, std::integral_constant<int, 5>{}); }
        if (m == 7) { stage_piece(s + 4, std::integral_constant<int, 6>{}); stage_piece(s + 4, std::integral_constant<int, 7>{}); }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  int s = 0;
  for (; s + 5 < ns; s += 2) {
    slab_body(s, P0{}, std::true_type{});
    slab_body(s + 1, P1{}, std::true_type{});
  }
  for (; s < ns; s += 2) {
    slab_body(s, P0{}, std::false_type{});
    slab_body(s + 1, P1{}, std::false_type{});
  }
  // every wave's last fragment reads were retired before the last slab's barrier: LDS is free.
  // The MFMAs are inline assembly, so the compiler does not know that the accumulators are still
  // in the matrix pipeline: cover the read-after-MFMA distance (18 wait states) by hand.
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
}

// Four-wave form with REGISTER staging (variant 12).  Same tile, wave layout, LDS image and fragment
// double-buffering as mainloop_256_4w, but a slab travels global -> registers -> LDS: an LDS-DMA piece
// holds the issuing wave for about 60 cycles, and with one wave per SIMD nothing else feeds the matrix
// pipe meanwhile (8 pieces per 1024 MFMA cycles: the 40 % the LDS-DMA form loses), whereas a
// `global_load_dwordx4` and a `ds_write_b128` each fit the 8 spare issue cycles behind a 16-cycle MFMA.
//   * two register sets of 8 x 16 bytes per lane: during slab s, set s&1 (slab s+2, loaded during slab s-2)
//     is written to LDS buffer (s+2)&3 piece by piece and each piece's registers are reloaded with slab s+4
//     at once - a load has two slabs (~2000 cycles) to land;
//   * first half of the slab's MFMAs: the 8 write + load pairs; second half: the 16 fragment reads of slab
//     s+1, in the order the next slab's MFMAs consume them, so none is waited for at the slab boundary;
//   * one barrier per slab: it publishes the writes of slab s+2 to the other waves' reads a slab later.
template <bool IS_BF16>
__device__ __forceinline__ void mainloop_256_4w_rs(const TileArgs& t, char* smem, Acc4<IS_BF16>& acc) {
  using F = Frag<IS_BF16>;
  using V = typename F::T;
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int wr = w >> 1, wc = w & 1;
  const int i = l >> 2;
  const int kchunk = (l & 3) ^ (((l >> 5) & 1) << 1);
  const uint16_t* pa[4];
  const uint16_t* pb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int ts = 4 * w + j;
    int ar = ts * 16 + i;
    ar = ar < t.a_rows ? ar : t.a_rows - 1;
    int br = (ts >> 3) * 128 + ((ts & 7) >> 1) * 32 + (i >> 2) * 8 + (ts & 1) * 4 + (i & 3);
    br = br < t.b_rows ? br : t.b_rows - 1;
    pa[j] = t.A + (long long)ar * t.lda + kchunk * 8;
    pb[j] = t.B + (long long)br * t.ldb + kchunk * 8;
  }
  const int ns = t.K / 32;                                  // even, >= 4
  char* st_dst = smem + w * 4096 + l * 16;                  // lane-linear image of a staged piece
  typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
  u32x4 st[2][8];
  auto ld_piece = [&](int s, int set, auto jc) {
    constexpr int j = decltype(jc)::value;                  // 0..3 A, 4..7 B
    const uint16_t* src = (j < 4 ? pa[j & 3] : pb[j & 3]) + (long long)s * 32;
    st[set][j] = *(const __attribute__((address_space(1))) u32x4*)src;
  };
  auto wr_piece = [&](int s, int set, auto jc) {
    constexpr int j = decltype(jc)::value;
    *(__attribute__((address_space(3))) u32x4*)(st_dst + (s & 3) * W4_SLAB + (j < 4 ? 0 : 16384) + (j & 3) * 1024) = st[set][j];
  };
  const int lane_off = (((l & 15) * 64) + ((l >> 4) * 16)) ^ ((((l & 15) >> 3) & 1) << 5);
  const char* a_rd = smem + wr * 8192 + lane_off;
  const char* b_rd = smem + 16384 + wc * 8192 + lane_off;
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int n = 0; n < 8; ++n) acc.v[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  auto for8 = [&](auto&& f) {
    f(std::integral_constant<int, 0>{}); f(std::integral_constant<int, 1>{}); f(std::integral_constant<int, 2>{});
    f(std::integral_constant<int, 3>{}); f(std::integral_constant<int, 4>{}); f(std::integral_constant<int, 5>{});
    f(std::integral_constant<int, 6>{}); f(std::integral_constant<int, 7>{});
  };
  // prologue: slabs 0, 1 into LDS, slabs 2, 3 into the two register sets
  for8([&](auto j) { ld_piece(0, 0, j); });
  for8([&](auto j) { ld_piece(1, 1, j); });
  for8([&](auto j) { wr_piece(0, 0, j); });
  for8([&](auto j) { ld_piece(2, 0, j); });
  for8([&](auto j) { wr_piece(1, 1, j); });
  for8([&](auto j) { ld_piece(3, 1, j); });
  __syncthreads();
  V fa[2][8], fb[2][8];
#pragma unroll
  for (int m = 0; m < 8; ++m) {
    fa[0][m] = *(const V*)(a_rd + m * 1024);
    fb[0][m] = *(const V*)(b_rd + m * 1024);
  }

  auto slab_body = [&](int s, auto pc, auto steady_c) {
    constexpr int P = decltype(pc)::value;
    constexpr bool STEADY = decltype(steady_c)::value;
    const bool more = STEADY || s + 1 < ns;                 // slab s+1 exists: read its fragments
    const bool wr_next = STEADY || s + 2 < ns;              // slab s+2 exists: it sits in set P, write it
    const bool ld_next = STEADY || s + 4 < ns;              // slab s+4 exists: load it into set P
    // The wave's writes of slab s+1 (first half of the previous slab) must be complete before the barrier that
    // publishes them; they are older than the 16 fragment reads issued behind them, so "all but the 15
    // youngest" covers them.  The reads themselves need no retiring here: with four buffers the one they read
    // is not rewritten before slab s+3.
    if constexpr (STEADY) {
      asm volatile("s_waitcnt lgkmcnt(15)" ::: "memory");
    } else {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    const int nb = ((s + 1) & 3) * W4_SLAB;
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      const int m = i >> 3, n = i & 7;
      F::mfma_acc(fb[P][n], fa[P][m], acc.v[m][n]);
      if (i >= 32) {
        if ((i & 1) == 0 && more) {
          const int j = (i - 32) >> 1;
          if (j < 8)
            fb[P ^ 1][j] = *(const V*)(b_rd + nb + j * 1024);
          else
            fa[P ^ 1][j - 8] = *(const V*)(a_rd + nb + (j - 8) * 1024);
        }
      } else if ((i & 3) == 0) {
        if (wr_next) {
          switch (i >> 2) {
            case 0: wr_piece(s + 2, P, std::integral_constant<int, 0>{}); break;
            case 1: wr_piece(s + 2, P, std::integral_constant<int, 1>{}); break;
            case 2: wr_piece(s + 2, P, std::integral_constant<int, 2>{}); break;
            case 3: wr_piece(s + 2, P, std::integral_constant<int, 3>{}); break;
            case 4: wr_piece(s + 2, P, std::integral_constant<int, 4>{}); break;
            case 5: wr_piece(s + 2, P, std::integral_constant<int, 5>{}); break;
            case 6: wr_piece(s + 2, P, std::integral_constant<int, 6>{}); break;
            default: wr_piece(s + 2, P, std::integral_constant<int, 7>{}); break;
          }
        }
      } else if ((i & 3) == 2) {
        if (ld_next) {
          switch (i >> 2) {
            case 0: ld_piece(s + 4, P, std::integral_constant<int, 0>{}); break;
            case 1: ld_piece(s + 4, P, std::integral_constant<int, 1>{}); break;
            case 2: ld_piece(s + 4, P, std::integral_constant<int, 2>{}); break;
            case 3: ld_piece(s + 4, P, std::integral_constant<int, 3>{}); break;
            case 4: ld_piece(s + 4, P, std::integral_constant<int, 4>{}); break;
            case 5: ld_piece(s + 4, P, std::integral_constant<int, 5>{}); break;
            case 6: ld_piece(s + 4, P, std::integral_constant<int, 6>{}); break;
            default: ld_piece(s + 4, P, std::integral_constant<int, 7>{}); break;
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  int s = 0;
  for (; s + 5 < ns; s += 2) {
    slab_body(s, P0{}, std::true_type{});
    slab_body(s + 1, P1{}, std::true_type{});
  }
  for (; s < ns; s += 2) {
    slab_body(s, P0{}, std::false_type{});
    slab_body(s + 1, P1{}, std::false_type{});
  }
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
}

// ---------------------------------------------------------------------------------------------
// Four-wave form on 32x32x16 MFMAs (variant 13).  One wave per SIMD has nobody to hide its memory
// instructions behind, and behind a 16-cycle 16x16x32 MFMA a wave has only ~8 spare issue cycles - less
// than a `ds_read_b128` takes (the 40 % the other four-wave forms lose).  A 32x32x16 MFMA keeps the
// matrix pipe busy for 32 cycles per issue, which leaves room for one memory instruction per MFMA.
//   * wave tile 128 x 128 = 4 x 4 accumulator blocks of 32 x 32 (16 registers each, 256 accumulation
//     registers); per 32-deep slab two k16 steps x 16 MFMAs, 16 fragment reads and 8 LDS-DMA pieces;
//   * a staged piece is [16 rows][4 chunks of 8 k] (adjacent lanes load adjacent bytes of a row), stored
//     lane-linear with chunk' = chunk ^ (row bits 2..3); a fragment of 32 rows x 16 k is then read from two
//     pieces with every 16-lane service group of `ds_read_b128` on 16 distinct bank slots;
//   * B rows are permuted at staging (fragment row f holds column 16*bit2(f) + 4*(f >> 3) + (f & 3) of its
//     32) so that a lane's 16 accumulator values of a block are 16 CONSECUTIVE columns of one output row;
//   * ring of four slab buffers, pieces of three slabs in flight, one barrier per slab (as mainloop_256_4w).
// EXP (timing experiments only, wrong results): bit 0 drops the barrier, bit 1 the staging inside the loop, bit 2
// the fragment reads.  Measured at 4096 x 8192 x 8192: MFMAs alone 296 us, + reads 334, + staging 381, all 413
// (barrier: 4 us) - an LDS-DMA piece still idles the matrix pipe for ~37 cycles and a read for ~9, so one wave
// per SIMD does not reach the eight-wave kernel (344 us) with this MFMA shape either.
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <bool IS_BF16>
struct Acc32 {
  f32x16 v[4][4];
};

template <bool IS_BF16>
__device__ __forceinline__ void mfma32_acc(typename Frag<IS_BF16>::T a, typename Frag<IS_BF16>::T b, f32x16& c) {
  if constexpr (IS_BF16) {
    asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  } else {
    asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  }
}

template <bool IS_BF16, int EXP = 0>
__device__ __forceinline__ void mainloop_256_4w32(const TileArgs& t, char* smem, Acc32<IS_BF16>& acc) {
  using F = Frag<IS_BF16>;
  using V = typename F::T;
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int wr = w >> 1, wc = w & 1;
  const int rl = l >> 2;                                    // row inside a staged piece
  const int kchunk = (l & 3) ^ ((rl >> 2) & 3);             // source chunk of the slot this lane fills
  const uint16_t* pa[4];
  const uint16_t* pb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int piece = 4 * w + j;                            // 0..15 inside the slab's A / B half
    int ar = piece * 16 + rl;
    ar = ar < t.a_rows ? ar : t.a_rows - 1;
    // B piece = half (piece & 1) of fragment (piece >> 1); its row rl is fragment row f = 16 * half + rl
    int br = (piece >> 1) * 32 + 16 * ((rl >> 2) & 1) + 4 * (2 * (piece & 1) + (rl >> 3)) + (rl & 3);
    br = br < t.b_rows ? br : t.b_rows - 1;
    pa[j] = t.A + (long long)ar * t.lda + kchunk * 8;
    pb[j] = t.B + (long long)br * t.ldb + kchunk * 8;
  }
  const int ns = t.K / 32;                                  // even, >= 4
  char* stage_dst = smem + w * 4096;
  auto stage_piece = [&](int s, auto jc) {
    constexpr int j = decltype(jc)::value;                  // 0..3 A, 4..7 B
    const uint16_t* src = (j < 4 ? pa[j & 3] : pb[j & 3]) + (long long)s * 32;
    char* dst = stage_dst + (s & 3) * W4_SLAB + (j < 4 ? 0 : 16384) + (j & 3) * 1024;
    __builtin_amdgcn_global_load_lds(TDIST_GLB_PTR(src), TDIST_LDS_PTR(dst), 16, 0, 0);
  };
  auto stage_slab = [&](int s) {
    stage_piece(s, std::integral_constant<int, 0>{}); stage_piece(s, std::integral_constant<int, 1>{});
    stage_piece(s, std::integral_constant<int, 2>{}); stage_piece(s, std::integral_constant<int, 3>{});
    stage_piece(s, std::integral_constant<int, 4>{}); stage_piece(s, std::integral_constant<int, 5>{});
    stage_piece(s, std::integral_constant<int, 6>{}); stage_piece(s, std::integral_constant<int, 7>{});
  };
  // fragment read offsets: lane = (row l & 31 of the fragment, k half l >> 5); k16 step c reads chunk 2c + h
  const int frl = l & 15, fh = l >> 5, fsw = (frl >> 2) & 3;
  const int lane_off0 = ((l >> 4) & 1) * 1024 + frl * 64 + ((fh ^ fsw) << 4);
  const int lane_off1 = ((l >> 4) & 1) * 1024 + frl * 64 + (((2 + fh) ^ fsw) << 4);
  const char* a_rd0 = smem + wr * 8192 + lane_off0;         // + slab buffer + m * 2048
  const char* a_rd1 = smem + wr * 8192 + lane_off1;
  const char* b_rd0 = smem + 16384 + wc * 8192 + lane_off0; // + slab buffer + n * 2048
  const char* b_rd1 = smem + 16384 + wc * 8192 + lane_off1;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc.v[m][n][r] = 0.f;

  V fa[2][2][4], fb[2][2][4];                               // [slab parity][k16 step][block]
  stage_slab(0); stage_slab(1); stage_slab(2); stage_slab(3);
  asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  __builtin_amdgcn_s_barrier();
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    fb[0][0][m] = *(const V*)(b_rd0 + m * 2048);
    fa[0][0][m] = *(const V*)(a_rd0 + m * 2048);
    fb[0][1][m] = *(const V*)(b_rd1 + m * 2048);
    fa[0][1][m] = *(const V*)(a_rd1 + m * 2048);
  }

  auto slab_body = [&](int s, auto pc, auto steady_c) {
    constexpr int P = decltype(pc)::value;
    constexpr bool STEADY = decltype(steady_c)::value;
    const bool more = (EXP & 4) ? false : (STEADY || s + 1 < ns);
    if constexpr (STEADY) {
      asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    } else if (more) {
      if (s + 3 < ns) {
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      } else if (s + 2 < ns) {
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
    }
    // this wave's reads of slab s (issued during slab s-1) are retired before the barrier: after it another
    // wave restages that buffer
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(EXP & 1)) __builtin_amdgcn_s_barrier();
    const int nb = ((s + 1) & 3) * W4_SLAB;
    const bool restage = (EXP & 2) ? false : (STEADY || s + 4 < ns);
#pragma unroll
    for (int i = 0; i < 32; ++i) {
      const int c = i >> 4, m = (i >> 2) & 3, n = i & 3;
      mfma32_acc<IS_BF16>(fb[P][c][n], fa[P][c][m], acc.v[m][n]);
      if (i < 16) {
        // next slab's fragments in the order its MFMAs need them: k16 step 0 (B0..3, A0..3), then step 1
        if (more) {
          const int j = i & 3;
          switch (i >> 2) {
            case 0: fb[P ^ 1][0][j] = *(const V*)(b_rd0 + nb + j * 2048); break;
            case 1: fa[P ^ 1][0][j] = *(const V*)(a_rd0 + nb + j * 2048); break;
            case 2: fb[P ^ 1][1][j] = *(const V*)(b_rd1 + nb + j * 2048); break;
            default: fa[P ^ 1][1][j] = *(const V*)(a_rd1 + nb + j * 2048); break;
          }
        }
      } else if ((i & 1) == 0 && restage) {
        switch ((i - 16) >> 1) {
          case 0: stage_piece(s + 4, std::integral_constant<int, 0>{}); break;
          case 1: stage_piece(s + 4, std::integral_constant<int, 1>{}); break;
          case 2: stage_piece(s + 4, std::integral_constant<int, 2>{}); break;
          case 3: stage_piece(s + 4, std::integral_constant<int, 3>{}); break;
          case 4: stage_piece(s + 4, std::integral_constant<int, 4>{}); break;
          case 5: stage_piece(s + 4, std::integral_constant<int, 5>{}); break;
          case 6: stage_piece(s + 4, std::integral_constant<int, 6>{}); break;
          default: stage_piece(s + 4, std::integral_constant<int, 7>{}); break;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  using P0 = std::integral_constant<int, 0>;
  using P1 = std::integral_constant<int, 1>;
  int s = 0;
  for (; s + 5 < ns; s += 2) {
    slab_body(s, P0{}, std::true_type{});
    slab_body(s + 1, P1{}, std::true_type{});
  }
  for (; s < ns; s += 2) {
    slab_body(s, P0{}, std::false_type{});
    slab_body(s + 1, P1{}, std::false_type{});
  }
  // the MFMAs are inline assembly: cover the read-after-MFMA distance of a 16-pass MFMA by hand
  asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
}

// Stores of the 32x32-block accumulators: per block a lane holds 16 consecutive columns of one row.
template <bool IS_BF16, class StoreFn>
__device__ __forceinline__ void epilogue_256_4w32_direct(const Acc32<IS_BF16>& acc, StoreFn&& store) {
  using F = Frag<IS_BF16>;
  const int tid = threadIdx.x;
  const int w = tid >> 6, l = tid & 63;
  const int row = (w >> 1) * 128 + (l & 31);
  const int col = (w & 1) * 128 + (l >> 5) * 16;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const f32x16& a = acc.v[m][n];
        uint4 v;
        v.x = F::pack2(a[8 * hh + 0], a[8 * hh + 1]);
        v.y = F::pack2(a[8 * hh + 2], a[8 * hh + 3]);
        v.z = F::pack2(a[8 * hh + 4], a[8 * hh + 5]);
        v.w = F::pack2(a[8 * hh + 6], a[8 * hh + 7]);
        store(row + 32 * m, col + 32 * n + 8 * hh, v);
      }
}

// Stores of the four-wave accumulators straight from registers: 32 pieces of 16 bytes per lane,
// `store(row_in_tile, col_in_tile, uint4)`.
template <bool IS_BF16, class StoreFn>
__device__ __forceinline__ void epilogue_256_4w_direct(const Acc4<IS_BF16>& acc, StoreFn&& store) {
  using F = Frag<IS_BF16>;
  const int tid = threadIdx.x;
  const int w = tid >> 6, l = tid & 63;
  const int row = (w >> 1) * 128 + (l & 15);
  const int col = (w & 1) * 128 + (l >> 4) * 8;
#pragma unroll
  for (int m = 0; m < 8; ++m)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      uint4 v;
      v.x = F::pack2(acc.v[m][2 * p][0], acc.v[m][2 * p][1]);
      v.y = F::pack2(acc.v[m][2 * p][2], acc.v[m][2 * p][3]);
      v.z = F::pack2(acc.v[m][2 * p + 1][0], acc.v[m][2 * p + 1][1]);
      v.w = F::pack2(acc.v[m][2 * p + 1][2], acc.v[m][2 * p + 1][3]);
      store(row + m * 16, col + 32 * p, v);
    }
}

// ---------------------------------------------------------------------------------------------
// The tile family: BM = 32 * MT rows x BN = 64 * NT columns, MT in {4, 8}, NT in {2, 4}, every operand type.
//
// Same eight waves (2 along M x 4 along N, wave tile 16 MT x 16 NT), same two-phase K-step, same ring of two
// K-step buffers, same LDS sub-tile image and the same barrier / counted-wait skeleton as mainloop_256_2ph
// (PERM_B columns, A fragments waited for block by block, no s_setprio) - only the sizes are parameters:
//   half tiles      A0 / A1 = MT/2 sixteen-row blocks per wave row (MT * 2 KiB), B0 / B1 = NT/2 blocks per wave
//                   column (NT * 4 KiB); a 16 KiB half is staged as two LDS-DMA pieces per wave (both k halves of
//                   sub-tile row w), an 8 KiB half as one (sub-tile row w / 2, k half w % 2);
//   phase 0         reads B0, B1, A0 (2 NT + MT fragments) and runs MT/2 * NT MFMA pairs, phase 1 reads A1 (MT);
//   LDS             2 * (MT * 4 + NT * 8) KiB: 128 KiB at 256 x 256, 96 at 256 x 128 and 128 x 256, 64 at 128 x 128 -
//                   two 128 x 128 workgroups are co-resident on a CU (64 accumulator + fragment registers per lane),
//                   one hides its LDS-DMA issue under the other's MFMAs.
// What the smaller tiles are for: a GEMM with fewer 256 x 256 tiles than the chip has CUs (a TP shard of a
// projection, a short prompt) leaves CUs idle; per K-step a 256 x 128 tile streams 3/4 and a 128 x 128 tile 1/2 of
// the bytes of a 256 x 256 tile for 1/2 and 1/4 of its MFMAs, so they fill the chip at the price of more
// L2 -> LDS traffic per FLOP (reference: the BM {128, 256} x BN {128, 256} autotune space of
// python/triton_dist/kernels/amd/all_gather_gemm.py:102-124 and gemm_reduce_scatter.py:39-122).
// 8-bit operand types: see the DT_* comment at the top; fp8 consumes both k halves of a fragment pair in one
// 128-deep MFMA.
template <int MT, int NT>
struct TileShape {
  static_assert((MT == 4 || MT == 8) && (NT == 2 || NT == 4), "tile family: BM in {128, 256}, BN in {128, 256}");
  static constexpr int BM = 32 * MT, BN = 64 * NT;
  static constexpr int HA = MT * 2048, HB = NT * 4096;        // bytes of an A / B half tile
  static constexpr int PA = HA / 8192, PB = HB / 8192;        // LDS-DMA pieces per wave and half tile
  static constexpr int BUF = 2 * (HA + HB);                   // one K-step
  static constexpr int OFF_B0 = 0, OFF_A0 = HB, OFF_B1 = HB + HA, OFF_A1 = 2 * HB + HA;
  static constexpr int LDS = 2 * BUF;
  static constexpr int WGS_PER_CU = LDS <= 65536 ? 2 : 1;
};

template <int MT, int NT>
struct AccT {
  f32x4 v[MT][NT];     // [16-row block][column tile: B0's NT/2 tiles, then B1's]
};

template <int N>
__device__ __forceinline__ void wait_lgkm() {
  asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vm() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Column (inside the wave's 16 NT columns) that accumulator register r of lane group g stands for in the
// column tile j of half h: a lane's registers of tiles (h, 0), (h, 1) [NT 4] or (0, 0), (1, 0) [NT 2] are eight
// neighbouring columns - one 16-byte store of 16-bit results.
template <int NT>
__device__ __forceinline__ int perm_col(int h, int j, int g, int r) {
  return NT == 4 ? 32 * h + 8 * g + 4 * j + r : 8 * g + 4 * h + r;
}

template <int DT, int MT, int NT>
__device__ __forceinline__ void mainloop_2ph(const TileArgs& t, char* smem, AccT<MT, NT>& acc) {
  using F = Frag<DT>;
  using V = typename F::T;
  using S = TileShape<MT, NT>;
  constexpr int MH = MT / 2, NH = NT / 2;
  const int tid = threadIdx.x;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l = tid & 63;
  const int wr = w >> 2, wc = w & 3;
  const bool late_group = w >= 4;
  const int i = l >> 2;                                        // row inside a staged sub-tile
  const int kchunk = (l & 3) ^ (((l >> 5) & 1) << 1);
  // ---- staging sources: sub-tile row `sa` / `sb` of the A / B half tiles, k half `ka` / `kb` when one piece ----
  const int sa = S::PA == 2 ? w : (w >> 1), ka = S::PA == 2 ? 0 : (w & 1);
  const int sb = S::PB == 2 ? w : (w >> 1), kb = S::PB == 2 ? 0 : (w & 1);
  int arow0 = (sa / MH) * (16 * MT) + (sa % MH) * 16 + i;      // A0; A1 = + 8 MT
  int arow1 = arow0 + 8 * MT;
  const int bcol = (sb / NH) * (16 * NT);                      // first column of the wave column this sub-tile feeds
  int brow0 = bcol + perm_col<NT>(0, sb % NH, i >> 2, i & 3);
  int brow1 = bcol + perm_col<NT>(1, sb % NH, i >> 2, i & 3);
  arow0 = arow0 < t.a_rows ? arow0 : t.a_rows - 1;
  arow1 = arow1 < t.a_rows ? arow1 : t.a_rows - 1;
  brow0 = brow0 < t.b_rows ? brow0 : t.b_rows - 1;
  brow1 = brow1 < t.b_rows ? brow1 : t.b_rows - 1;
  if (t.a_gather) {
    const int g0 = t.a_gather[arow0], g1 = t.a_gather[arow1];
    arow0 = g0 < 0 ? 0 : g0;
    arow1 = g1 < 0 ? 0 : g1;
  }
  const uint16_t* pa0 = t.A + (long long)arow0 * t.lda + kchunk * 8 + ka * 32;
  const uint16_t* pa1 = t.A + (long long)arow1 * t.lda + kchunk * 8 + ka * 32;
  const uint16_t* pb0 = t.B + (long long)brow0 * t.ldb + kchunk * 8 + kb * 32;
  const uint16_t* pb1 = t.B + (long long)brow1 * t.ldb + kchunk * 8 + kb * 32;
  const int nt = t.K / BK;
  const int dst_a = S::PA == 2 ? w * 2048 : w * 1024;
  const int dst_b = S::PB == 2 ? w * 2048 : w * 1024;
  auto stage = [&](int kt, auto kind_c) {
    constexpr int kind = decltype(kind_c)::value;              // 0 = B0, 1 = A0, 2 = B1, 3 = A1
    constexpr bool is_a = (kind & 1) != 0;
    constexpr int off = kind == 0 ? S::OFF_B0 : (kind == 1 ? S::OFF_A0 : (kind == 2 ? S::OFF_B1 : S::OFF_A1));
    if (kt < nt) {
      const uint16_t* src = kind == 0 ? pb0 : (kind == 1 ? pa0 : (kind == 2 ? pb1 : pa1));
      src += (long long)kt * BK;
      char* dst = smem + (kt & 1) * S::BUF + off + (is_a ? dst_a : dst_b);
      __builtin_amdgcn_global_load_lds(TDIST_GLB_PTR(src), TDIST_LDS_PTR(dst), 16, 0, 0);
      if constexpr ((is_a ? S::PA : S::PB) == 2)
        __builtin_amdgcn_global_load_lds(TDIST_GLB_PTR(src + 32), TDIST_LDS_PTR(dst + 1024), 16, 0, 0);
    }
  };
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, 1>;
  using K2 = std::integral_constant<int, 2>;
  using K3 = std::integral_constant<int, 3>;
  const int lane_off = (((l & 15) * 64) + ((l >> 4) * 16)) ^ ((((l & 15) >> 3) & 1) << 5);
  const char* a_rd = smem + wr * (S::HA / 2) + lane_off;       // + OFF_A{0,1} + (m * 2 + ks) * 1024
  const char* b_rd = smem + wc * (S::HB / 4) + lane_off;       // + OFF_B{0,1} + (j * 2 + ks) * 1024
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int n = 0; n < NT; ++n) acc.v[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};

  stage(0, K0{}); stage(0, K1{}); stage(0, K2{}); stage(0, K3{});
  stage(1, K0{}); stage(1, K2{});
  wait_vm<2 * S::PB>();
  __builtin_amdgcn_s_barrier();
  if (late_group) __builtin_amdgcn_s_barrier();

  V a[MH][2], b0[NH][2], b1[NH][2];
  // MFMAs of the 16-row block m of the current half against all NT column tiles, both k halves
  auto mfma_row = [&](int m, int mo) {
    if constexpr (F::K128) {
#pragma unroll
      for (int n = 0; n < NH; ++n) acc.v[mo + m][n] = F::mfma2(b0[n][0], b0[n][1], a[m][0], a[m][1], acc.v[mo + m][n]);
#pragma unroll
      for (int n = 0; n < NH; ++n)
        acc.v[mo + m][NH + n] = F::mfma2(b1[n][0], b1[n][1], a[m][0], a[m][1], acc.v[mo + m][NH + n]);
    } else {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int n = 0; n < NH; ++n) acc.v[mo + m][n] = F::mfma(b0[n][ks], a[m][ks], acc.v[mo + m][n]);
#pragma unroll
        for (int n = 0; n < NH; ++n) acc.v[mo + m][NH + n] = F::mfma(b1[n][ks], a[m][ks], acc.v[mo + m][NH + n]);
      }
    }
  };
  for (int kt = 0; kt < nt; ++kt) {
    const int boff = (kt & 1) * S::BUF;
    // ---------------- phase 0: upper half of the wave tile's rows ----------------
#pragma unroll
    for (int n = 0; n < NH; ++n)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) b0[n][ks] = *(const V*)(b_rd + boff + S::OFF_B0 + (n * 2 + ks) * 1024);
#pragma unroll
    for (int n = 0; n < NH; ++n)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) b1[n][ks] = *(const V*)(b_rd + boff + S::OFF_B1 + (n * 2 + ks) * 1024);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int m = 0; m < MH; ++m)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) a[m][ks] = *(const V*)(a_rd + boff + S::OFF_A0 + (m * 2 + ks) * 1024);
    stage(kt + 1, K1{});
    stage(kt + 1, K3{});
    __builtin_amdgcn_sched_barrier(0);
    // the B reads are retired before the barrier (phase 1 of the OTHER wave group restages their slots right
    // behind it); the MT A0 reads are younger and are waited for block by block inside the MFMA segment
    wait_lgkm<MT>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    wait_lgkm<MT - 2>();
    mfma_row(0, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (MH > 1) {
      wait_lgkm<MT - 4>();
      mfma_row(1, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (MH > 2) {
      wait_lgkm<MT - 6>();
      mfma_row(2, 0);
      __builtin_amdgcn_sched_barrier(0);
      wait_lgkm<MT - 8>();
      mfma_row(3, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_s_barrier();
    // ---------------- phase 1: lower half ----------------
#pragma unroll
    for (int m = 0; m < MH; ++m)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) a[m][ks] = *(const V*)(a_rd + boff + S::OFF_A1 + (m * 2 + ks) * 1024);
    stage(kt + 2, K0{});
    stage(kt + 2, K2{});
    // retire A0, A1 of step kt+1 (issued in phase 0) and its B halves (issued one step ago): everything but the
    // two B halves just issued, if they were
    if (kt + 2 < nt) {
      wait_vm<2 * S::PB>();
    } else if (kt + 1 < nt) {
      wait_vm<0>();
    }
    // (complete: the other wave group restages this buffer's A1 slot one barrier later)
    wait_lgkm<0>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int m = 0; m < MH; ++m) mfma_row(m, MH);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
  }
  if (!late_group) __builtin_amdgcn_s_barrier();
}

// Register-direct stores of the tile family's accumulators.
//   store16(row, col, uint4): eight 16-bit results of one row (columns col .. col + 7), scaled by
//     sa[row] * sb[col ..] when the scale pointers are given (per-row scale of A, per-column scale of B: the
//     de-quantisation of 8-bit operands), after Frag::val (int32 -> float for int8), plus bias[col ..] (16-bit, the
//     output type) when that pointer is given;
//   store32(row, col, uint4 lo, uint4 hi): the same eight results as raw 32-bit accumulator words (int32 output
//     of an int8 GEMM).
template <int DT, int MT, int NT, class StoreFn>
__device__ __forceinline__ void epilogue_direct(const AccT<MT, NT>& acc, const float* sa, const float* sb, int row0,
                                                int col0, int mrem, int nrem, StoreFn&& store16,
                                                const uint16_t* bias = nullptr) {
  using F = Frag<DT>;
  const int tid = threadIdx.x;
  const int w = tid >> 6, l = tid & 63;
  const int row = (w >> 2) * (16 * MT) + (l & 15);
  const int col = (w & 3) * (16 * NT) + (l >> 4) * 8;
  const bool scaled = sa != nullptr || sb != nullptr || bias != nullptr;
  float cs[NT / 2][8], cb[NT / 2][8];
  if (scaled) {
#pragma unroll
    for (int p = 0; p < NT / 2; ++p)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = col + 32 * p + e;
        cs[p][e] = (sb && c < nrem) ? sb[col0 + c] : 1.f;
        float bv = 0.f;
        if (bias && c < nrem) {
          const uint16_t raw = bias[col0 + c];
          bv = DT == DT_F16 ? (float)__builtin_bit_cast(_Float16, raw) : __builtin_bit_cast(float, (uint32_t)raw << 16);
        }
        cb[p][e] = bv;
      }
  }
#pragma unroll
  for (int m = 0; m < MT; ++m) {
    const int r = row + m * 16;
    float rs = 1.f;
    if (scaled && sa && r < mrem) rs = sa[row0 + r];
#pragma unroll
    for (int p = 0; p < NT / 2; ++p) {
      const f32x4& x0 = NT == 4 ? acc.v[m][2 * p] : acc.v[m][0];
      const f32x4& x1 = NT == 4 ? acc.v[m][2 * p + 1] : acc.v[m][NT - 1];
      uint4 v;
      if (scaled) {
        const float *c8 = cs[p], *b8 = cb[p];
        using FO = Frag<DT == DT_F16 ? DT_F16 : DT_BF16>;     // output type: f16 for f16 operands, bf16 otherwise
        v.x = FO::pack2(F::val(x0[0]) * rs * c8[0] + b8[0], F::val(x0[1]) * rs * c8[1] + b8[1]);
        v.y = FO::pack2(F::val(x0[2]) * rs * c8[2] + b8[2], F::val(x0[3]) * rs * c8[3] + b8[3]);
        v.z = FO::pack2(F::val(x1[0]) * rs * c8[4] + b8[4], F::val(x1[1]) * rs * c8[5] + b8[5]);
        v.w = FO::pack2(F::val(x1[2]) * rs * c8[6] + b8[6], F::val(x1[3]) * rs * c8[7] + b8[7]);
      } else {
        v.x = F::pack2(x0[0], x0[1]);
        v.y = F::pack2(x0[2], x0[3]);
        v.z = F::pack2(x1[0], x1[1]);
        v.w = F::pack2(x1[2], x1[3]);
      }
      store16(r, col + 32 * p, v);
    }
  }
}

// LDS-staged stores of the tile family's accumulators: the tile is laid out row-major in LDS (BM rows of BN 16-bit
// values + 16 bytes of padding) and handed to `store16(row, col, uint4)` in pieces such that BN / 8 neighbouring
// threads cover one whole row - every row leaves as ONE contiguous run of 2 BN bytes.  For write-through (sc1 /
// sc0 sc1) stores, which are not merged in L2, that is 1.3 - 1.5x faster than the 64-byte pieces of epilogue_direct.
// Needs staged_lds_bytes<MT, NT>() of LDS (the 256 x 256 image is 4 KiB larger than the main loop's ring); begins and
// ends with a workgroup barrier.
template <int MT, int NT>
constexpr int staged_lds_bytes() {
  return 32 * MT * (2 * 64 * NT + 16) > TileShape<MT, NT>::LDS ? 32 * MT * (2 * 64 * NT + 16) : TileShape<MT, NT>::LDS;
}
template <int DT, int MT, int NT, class StoreFn>
__device__ __forceinline__ void epilogue_staged(const AccT<MT, NT>& acc, char* smem, StoreFn&& store16) {
  using F = Frag<DT>;
  constexpr int BM_ = 32 * MT, BN_ = 64 * NT, STRIDE = 2 * BN_ + 16;
  const int tid = threadIdx.x;
  const int w = tid >> 6, l = tid & 63;
  char* base = smem + ((w >> 2) * (16 * MT) + (l & 15)) * STRIDE + ((w & 3) * (16 * NT) + (l >> 4) * 8) * 2;
  __syncthreads();                       // (every wave is past its last fragment read)
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int p = 0; p < NT / 2; ++p) {
      const f32x4& x0 = acc.v[m][2 * p];
      const f32x4& x1 = acc.v[m][2 * p + 1];
      uint4 v;
      v.x = F::pack2(x0[0], x0[1]);
      v.y = F::pack2(x0[2], x0[3]);
      v.z = F::pack2(x1[0], x1[1]);
      v.w = F::pack2(x1[2], x1[3]);
      *(uint4*)(base + m * 16 * STRIDE + p * 64) = v;
    }
  __syncthreads();
  constexpr int TPR = BN_ / 8;           // threads per row
  constexpr int RPP = THREADS / TPR;     // rows per pass
  const int r0 = tid / TPR, cc = tid % TPR;
#pragma unroll 4
  for (int pass = 0; pass < BM_ / RPP; ++pass) {
    const int row = pass * RPP + r0;
    store16(row, cc * 8, *(const uint4*)(smem + row * STRIDE + cc * 16));
  }
  __syncthreads();
}

template <int MT, int NT, class StoreFn>
__device__ __forceinline__ void epilogue_direct_raw32(const AccT<MT, NT>& acc, StoreFn&& store32) {
  const int tid = threadIdx.x;
  const int w = tid >> 6, l = tid & 63;
  const int row = (w >> 2) * (16 * MT) + (l & 15);
  const int col = (w & 3) * (16 * NT) + (l >> 4) * 8;
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int p = 0; p < NT / 2; ++p) {
      const f32x4& x0 = NT == 4 ? acc.v[m][2 * p] : acc.v[m][0];
      const f32x4& x1 = NT == 4 ? acc.v[m][2 * p + 1] : acc.v[m][NT - 1];
      store32(row + m * 16, col + 32 * p, __builtin_bit_cast(uint4, x0), __builtin_bit_cast(uint4, x1));
    }
}

// Bijective remap of a linear workgroup id so that each group of ids that shares an XCD
// (ids equal mod 8 under round-robin dispatch) owns one contiguous range of work.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + idx;
}

// Grouped (GROUP_M rows at a time) mapping of a linear tile id to (pid_m, pid_n).
__device__ __forceinline__ void tile_coords(int id, int num_pid_m, int num_pid_n, int group_m, int& pid_m, int& pid_n) {
  const int per_group = group_m * num_pid_n;
  const int group = id / per_group;
  const int first_m = group * group_m;
  const int gsz = min(num_pid_m - first_m, group_m);
  const int in = id - group * per_group;
  pid_m = first_m + in % gsz;
  pid_n = in / gsz;
}

}  // namespace gemm
}  // namespace tdist
