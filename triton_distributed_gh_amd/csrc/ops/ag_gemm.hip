// AllGather + GEMM consumer:  C[M, N_local] = AllGather_M(A[M/w, K]) * B[N_local, K]^T.
//
// The producer side (copy-engine pushes over xGMI, or the push kernel below) writes every peer's
// shard of A into this rank's symmetric workspace [M, K] chunk by chunk and raises one int32 flag
// per (source rank, chunk) to the call's epoch value.  This persistent kernel pulls 256x256 tile
// jobs from a device-side ticket counter in expected-arrival order (own rows first, then chunk 0
// of every peer, chunk 1 of every peer, ...), waits - workgroup-collectively, system scope,
// acquire - for the flags covering the tile's rows, and runs the MFMA main loop.
//
// Differences from the reference's consumer
// (reference: python/triton_dist/kernels/amd/all_gather_gemm.py:127-245, 401-433):
//   * flags carry a per-call epoch and workspaces / flags are double-buffered by call parity, so
//     no flag reset and no cross-rank barrier is needed after the GEMM;
//   * tiles are handed out dynamically, so a workgroup stalled on a late chunk never holds back
//     tiles whose data already arrived;
//   * every spin is bounded and reports through an error word instead of hanging the GPU.
#include "tdist/common.hpp"
#include "tdist/gemm_gfx950.hpp"
#include "tdist/primitives.hpp"
#include "tdist/shmem_device.hpp"

using namespace tdist;
using namespace tdist::gemm;

struct AgGemmParams {
  const uint16_t* a_local;   // [M_per_rank, K] own shard (never copied unless straddle)
  const uint16_t* ws;        // [M, K] symmetric workspace of this call's parity
  const uint16_t* B;         // [N_local, K]
  uint16_t* C;               // [M, N_local]
  const int* flags;          // [w * chunks_per_rank] of this call's parity
  int* ticket;               // zeroed before launch
  int* err;                  // sticky error word (bit 0 = wait timeout)
  long long lda_local, ld_ws, ldb, ldc;
  int M_per_rank, N_local, K;
  int rank, world;
  int rows_per_chunk;
  int epoch;
  int straddle;              // 1: tiles walk the whole [M, K] workspace (own rows copied in too)
};

constexpr int TICKET_SLOT = LDS_BYTES;  // 16 bytes after the tile staging area

template <bool IS_BF16>
__global__ __launch_bounds__(THREADS, 2) void ag_gemm_consumer_kernel(AgGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  volatile int* s_ticket = (volatile int*)(smem + TICKET_SLOT);
  const int w = p.world;
  const int M = p.M_per_rank * w;
  const int num_pid_n = (p.N_local + BN - 1) / BN;
  const int cpr = p.M_per_rank / p.rows_per_chunk;  // chunks per rank
  int mb_per_rank, total_mb;
  if (p.straddle) {
    mb_per_rank = 0;
    total_mb = (M + BM - 1) / BM;
  } else {
    mb_per_rank = (p.M_per_rank + BM - 1) / BM;
    total_mb = mb_per_rank * w;
  }
  const int total = total_mb * num_pid_n;

  while (true) {
    const int item = fetch_ticket_block(p.ticket, s_ticket);
    if (item >= total) break;
    const int o = item / num_pid_n;
    const int pid_n = item - o * num_pid_n;

    // ---- which rows does this tile cover -----------------------------------------------
    int g0, rows;          // first global row, valid rows (<= 256)
    const uint16_t* a_ptr;
    long long lda;
    bool need_wait;
    if (p.straddle) {
      // start with the row block that contains this rank's rows, then walk upwards cyclically
      const int first = (p.rank * p.M_per_rank) / BM;
      const int mb = (first + o) % total_mb;
      g0 = mb * BM;
      rows = min(BM, M - g0);
      a_ptr = p.ws + (long long)g0 * p.ld_ws;
      lda = p.ld_ws;
      need_wait = true;
    } else {
      int s, j;
      if (o < mb_per_rank) {
        s = p.rank;
        j = o;
      } else {
        const int o2 = o - mb_per_rank;
        j = o2 / (w - 1);
        s = (p.rank + 1 + o2 % (w - 1)) % w;
      }
      const int r_in = j * BM;
      rows = min(BM, p.M_per_rank - r_in);
      g0 = s * p.M_per_rank + r_in;
      if (s == p.rank) {
        a_ptr = p.a_local + (long long)r_in * p.lda_local;
        lda = p.lda_local;
        need_wait = false;
      } else {
        a_ptr = p.ws + (long long)g0 * p.ld_ws;
        lda = p.ld_ws;
        need_wait = true;
      }
    }
    if (need_wait) {
      const int g1 = g0 + rows - 1;
      const int f0 = (g0 / p.M_per_rank) * cpr + (g0 % p.M_per_rank) / p.rows_per_chunk;
      const int f1 = (g1 / p.M_per_rank) * cpr + (g1 % p.M_per_rank) / p.rows_per_chunk;
      const int token = wait(p.flags + f0, f1 - f0 + 1, SCOPE_SYS, SEM_ACQUIRE, p.epoch, p.err);
      a_ptr = consume_token(a_ptr, token);
    }

    TileArgs t;
    t.A = a_ptr;
    t.B = p.B + (long long)pid_n * BN * p.ldb;
    t.lda = lda;
    t.ldb = p.ldb;
    t.a_rows = rows;
    t.b_rows = p.N_local - pid_n * BN;
    t.K = p.K;
    Acc<IS_BF16> acc;
    mainloop_256_2ph<IS_BF16>(t, smem, acc);
    const int nrem = p.N_local - pid_n * BN;
    uint16_t* cbase = p.C + (long long)g0 * p.ldc + pid_n * BN;
    const long long ldc = p.ldc;
    epilogue_256<IS_BF16>(acc, smem, [&](int r, int c, const uint4& v) {
      if (r < rows && c < nrem) *(uint4*)(cbase + (long long)r * ldc + c) = v;
    });
  }
}

// In-kernel producer (alternative to the copy engines): workgroup b pushes slice (b % slices) of
// chunk (b / slices) of the local shard to every peer, then the last slice of a chunk to finish
// raises that chunk's flag on every peer.
struct AgPushParams {
  SymmCtx ctx;
  const uint16_t* a_local;
  uint16_t* ws;            // this rank's workspace (symmetric address)
  int* flags;              // this rank's flag array (symmetric address, signal arena)
  int* done_count;         // [chunks] local counters, zeroed before launch
  long long row_bytes;
  int M_per_rank, rows_per_chunk, slices, epoch;
  int include_self;        // also copy into own workspace (straddle mode)
};

__global__ __launch_bounds__(512) void ag_push_kernel(AgPushParams p) {
  const int chunk = blockIdx.x / p.slices;
  const int slice = blockIdx.x % p.slices;
  const int w = p.ctx.num_ranks, rank = p.ctx.rank;
  const long long chunk_bytes = (long long)p.rows_per_chunk * p.row_bytes;
  const long long per = ((chunk_bytes / p.slices) + 15) & ~15ll;
  const long long b0 = min((long long)slice * per, chunk_bytes);
  const long long b1 = min(b0 + per, chunk_bytes);
  const char* src = (const char*)p.a_local + (long long)chunk * chunk_bytes + b0;
  const long long dst_off = ((long long)rank * p.M_per_rank + (long long)chunk * p.rows_per_chunk) * p.row_bytes + b0;
  // Rotate the peer order per workgroup so that at any moment the workgroups of this kernel are
  // spread over all w-1 xGMI links instead of all writing to the same peer.
  if (p.include_self) {
    char* dst = (char*)p.ws + dst_off;
    shmem::copy_bytes(dst, src, (size_t)(b1 - b0), threadIdx.x, blockDim.x);
  }
  for (int i = 0; i < w - 1; ++i) {
    const int peer = (rank + 1 + (i + blockIdx.x) % (w - 1)) % w;
    char* dst = (char*)symm_at(p.ctx, p.ws, peer) + dst_off;
    shmem::copy_bytes(dst, src, (size_t)(b1 - b0), threadIdx.x, blockDim.x);
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const int prev = atom_add_acqrel_agent(p.done_count + chunk, 1);
    if (prev == p.slices - 1) {
      const int cpr = p.M_per_rank / p.rows_per_chunk;
      for (int i = p.include_self ? 0 : 1; i < w; ++i) {
        const int peer = (rank + i) % w;
        store_release_system(symm_at(p.ctx, p.flags, peer) + rank * cpr + chunk, p.epoch);
      }
    }
  }
}

static int check_common(const void* B, const void* C, int N_local, int K, long long ldb, long long ldc) {
  TDIST_REQUIRE(K >= 128 && K % 64 == 0, "K must be a multiple of 64 and >= 128");
  TDIST_REQUIRE(N_local % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0, "N_local/ldb/ldc must be multiples of 8");
  TDIST_REQUIRE((((uintptr_t)B | (uintptr_t)C) & 15) == 0, "B and C must be 16-byte aligned");
  return 0;
}

TDIST_API int tdist_ag_gemm_consumer(const void* a_local, const void* ws, const void* B, void* C, const int* flags,
                                     int* ticket, int* err, long long lda_local, long long ld_ws, long long ldb,
                                     long long ldc, int M_per_rank, int N_local, int K, int rank, int world,
                                     int rows_per_chunk, int epoch, int straddle, int num_ctas, int dtype,
                                     void* stream) {
  if (check_common(B, C, N_local, K, ldb, ldc)) return -1;
  TDIST_REQUIRE(world >= 1 && world <= TDIST_MAX_RANKS && rank >= 0 && rank < world, "bad rank/world");
  TDIST_REQUIRE(M_per_rank > 0 && rows_per_chunk > 0 && M_per_rank % rows_per_chunk == 0,
                "M_per_rank must be a positive multiple of rows_per_chunk");
  TDIST_REQUIRE(lda_local % 8 == 0 && ld_ws % 8 == 0, "lda must be a multiple of 8");
  TDIST_REQUIRE((((uintptr_t)a_local | (uintptr_t)ws) & 15) == 0, "A operands must be 16-byte aligned");
  TDIST_REQUIRE(straddle || world == 1 || true, "");
  AgGemmParams p;
  p.a_local = (const uint16_t*)a_local;
  p.ws = (const uint16_t*)ws;
  p.B = (const uint16_t*)B;
  p.C = (uint16_t*)C;
  p.flags = flags;
  p.ticket = ticket;
  p.err = err;
  p.lda_local = lda_local;
  p.ld_ws = ld_ws;
  p.ldb = ldb;
  p.ldc = ldc;
  p.M_per_rank = M_per_rank;
  p.N_local = N_local;
  p.K = K;
  p.rank = rank;
  p.world = world;
  p.rows_per_chunk = rows_per_chunk;
  p.epoch = epoch;
  p.straddle = straddle;
  const int M = M_per_rank * world;
  const int total_mb = straddle ? (M + BM - 1) / BM : ((M_per_rank + BM - 1) / BM) * world;
  const int total = total_mb * ((N_local + BN - 1) / BN);
  int grid = num_ctas > 0 ? num_ctas : 256;
  if (grid > total) grid = total;
  static bool attr_set = false;
  if (!attr_set) {
    TDIST_HIP_TRY(hipFuncSetAttribute((const void*)ag_gemm_consumer_kernel<true>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 16));
    TDIST_HIP_TRY(hipFuncSetAttribute((const void*)ag_gemm_consumer_kernel<false>,
                                      hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 16));
    attr_set = true;
  }
  TDIST_HIP_TRY(hipMemsetAsync(ticket, 0, 16, (hipStream_t)stream));
  if (dtype == 0)
    hipLaunchKernelGGL(ag_gemm_consumer_kernel<true>, dim3(grid), dim3(THREADS), LDS_BYTES + 16, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(ag_gemm_consumer_kernel<false>, dim3(grid), dim3(THREADS), LDS_BYTES + 16, (hipStream_t)stream, p);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

TDIST_API int tdist_ag_push_kernel(const tdist::SymmCtx* ctx, const void* a_local, void* ws, int* flags, int* done_count,
                                   long long row_bytes, int M_per_rank, int rows_per_chunk, int slices, int epoch,
                                   int include_self, void* stream) {
  TDIST_REQUIRE(rows_per_chunk > 0 && M_per_rank % rows_per_chunk == 0, "M_per_rank % rows_per_chunk != 0");
  TDIST_REQUIRE(slices > 0, "slices must be positive");
  const int chunks = M_per_rank / rows_per_chunk;
  AgPushParams p;
  p.ctx = *ctx;
  p.a_local = (const uint16_t*)a_local;
  p.ws = (uint16_t*)ws;
  p.flags = flags;
  p.done_count = done_count;
  p.row_bytes = row_bytes;
  p.M_per_rank = M_per_rank;
  p.rows_per_chunk = rows_per_chunk;
  p.slices = slices;
  p.epoch = epoch;
  p.include_self = include_self;
  TDIST_HIP_TRY(hipMemsetAsync(done_count, 0, (size_t)chunks * sizeof(int), (hipStream_t)stream));
  hipLaunchKernelGGL(ag_push_kernel, dim3(chunks * slices), dim3(512), 0, (hipStream_t)stream, p);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
