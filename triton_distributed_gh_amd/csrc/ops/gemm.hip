// Plain (single-GPU) 16-bit GEMM on the gfx950 tile engine:  C[M,N] = A[M,K] * B[N,K]^T.
//
// This is TP=1 of the headline metric and the building block of every fused op.  The launch is
// one workgroup per 256x256 tile; block ids are remapped so that the ids that share an XCD's L2
// own a compact group_m x (64/group_m) patch of tiles.
//
// Role in the reference: `tl.dot` matmul kernels of python/triton_dist/test/amd/test_matmul_amd.py
// and my_lab/benchmark/layer.py:20-64 (y = x W^T).
#include "tdist/common.hpp"
#include "tdist/gemm_gfx950.hpp"

using namespace tdist::gemm;

template <bool IS_BF16, int VARIANT>
__global__ __launch_bounds__(THREADS, 2) void gemm_nt_256_kernel(const uint16_t* __restrict__ A,
                                                                 const uint16_t* __restrict__ B,
                                                                 uint16_t* __restrict__ C, int M, int N, int K,
                                                                 long long lda, long long ldb, long long ldc,
                                                                 int group_m, int dbg_flags) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int num_pid_m = (M + BM - 1) / BM;
  const int num_pid_n = (N + BN - 1) / BN;
  const int nwg = num_pid_m * num_pid_n;
  const int id = xcd_remap(blockIdx.x, nwg);
  int pid_m, pid_n;
  tile_coords(id, num_pid_m, num_pid_n, group_m, pid_m, pid_n);
  const int row0 = pid_m * BM, col0 = pid_n * BN;

  TileArgs t;
  t.A = A + (long long)row0 * lda;
  t.B = B + (long long)col0 * ldb;
  t.lda = lda;
  t.ldb = ldb;
  t.a_rows = M - row0;
  t.b_rows = N - col0;
  t.K = K;
  Acc<IS_BF16> acc;
  if constexpr (VARIANT == 2)
    mainloop_256_2ph<IS_BF16>(t, smem, acc);
  else
    mainloop_256<IS_BF16>(t, smem, acc);
  const int mrem = M - row0, nrem = N - col0;
  uint16_t* cbase = C + (long long)row0 * ldc + col0;
  epilogue_256<IS_BF16>(acc, smem, [&](int r, int c, const uint4& v) {
    if (r < mrem && c < nrem && !(dbg_flags & 1)) *(uint4*)(cbase + (long long)r * ldc + c) = v;
  });
}

// dtype: 0 = bf16, 1 = fp16.  Requires K % 64 == 0, K >= 128, N % 8 == 0, 16-byte aligned rows.
TDIST_API int tdist_gemm_nt(const void* A, const void* B, void* C, int M, int N, int K, long long lda,
                            long long ldb, long long ldc, int dtype, int group_m, void* stream, int variant) {
  TDIST_REQUIRE(M > 0 && N > 0, "empty GEMM");
  TDIST_REQUIRE(K >= 128 && K % 64 == 0, "K must be a multiple of 64 and >= 128");
  TDIST_REQUIRE(N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0, "N/lda/ldb/ldc must be multiples of 8");
  TDIST_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)C) & 15) == 0, "operands must be 16-byte aligned");
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  if (group_m <= 0) group_m = 8;
  typedef void (*kern_t)(const uint16_t*, const uint16_t*, uint16_t*, int, int, int, long long, long long, long long, int, int);
  kern_t table[2][2] = {{gemm_nt_256_kernel<true, 1>, gemm_nt_256_kernel<true, 2>},
                        {gemm_nt_256_kernel<false, 1>, gemm_nt_256_kernel<false, 2>}};
  static bool attr_set = false;
  if (!attr_set) {
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j)
        TDIST_HIP_TRY(hipFuncSetAttribute((const void*)table[i][j], hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    attr_set = true;
  }
  kern_t k = table[dtype == 0 ? 0 : 1][(variant & 255) == 2 ? 1 : 0];
  hipLaunchKernelGGL(k, dim3(tiles), dim3(THREADS), LDS_BYTES, (hipStream_t)stream, (const uint16_t*)A,
                     (const uint16_t*)B, (uint16_t*)C, M, N, K, lda, ldb, ldc, group_m, variant >> 8);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
