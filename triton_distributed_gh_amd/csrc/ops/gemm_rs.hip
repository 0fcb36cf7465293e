// GEMM + ReduceScatter, tile-pipelined, one persistent kernel per rank:
//     out[M/w, N] = ReduceScatter_M( A[M, K_local] * B[N, K_local]^T ).
//
// Work items live in eight queues, one per XCD (tdist/work_queue.hpp: static first item, tickets drawn ahead
// of the main loop, a workgroup whose queue is empty moves on to one that is not).  A queue owns a
// band of column tiles (and, with fewer than eight column tiles, every qm-th row block) plus every eighth
// reduction item, so the workgroups of one XCD share A row blocks and B panels in that XCD's L2 the way the
// plain GEMM's block-id remap arranges it.  Every queue runs in this order (cpr = chunks per rank):
//     GEMM(chunk 0), GEMM(chunk 1), REDUCE(chunk 0), GEMM(chunk 2), REDUCE(chunk 1), ... REDUCE(cpr-1)
// GEMM(c) covers chunk c of EVERY destination rank, destinations visited as rank+1, rank+2, ...,
// rank (own rows last), so in each step every xGMI link of the full mesh carries one stream.  The
// epilogue of a tile stores its 256x256 block straight into the destination's symmetric scatter
// buffer (segment = source rank).  A per-(destination, chunk) counter counts finished tiles; the
// workgroup that completes a chunk raises flag[src_rank, chunk] = epoch on the destination
// (system-scope release).  REDUCE(c) items wait until all w partials of chunk c have landed
// locally and sum them in fp32.  Because REDUCE(c) is queued behind GEMM(c+1), the reduction of a
// chunk overlaps the GEMM of the next one and only the last chunk's reduction is exposed.
//
// The reference's AMD version is barrier - GEMM(fuse scatter) - barrier - reduce with bf16
// accumulation (reference: python/triton_dist/kernels/amd/gemm_reduce_scatter.py:125-365); its
// NVIDIA version pipelines per segment with a separate reduction stream
// (kernels/nvidia/gemm_reduce_scatter.py:121-250, reduce_scatter.py:600-697).  Here scatter
// buffers and flags are double-buffered by call parity and flags carry the call's epoch, so no
// cross-rank barrier is needed at all.
#include <stdio.h>
#include <stdlib.h>

#include "tdist/cache_policy.hpp"
#include "tdist/call_state.hpp"
#include "tdist/common.hpp"
#include "tdist/gemm_gfx950.hpp"
#include "tdist/primitives.hpp"
#include "tdist/race_debug.hpp"
#include "tdist/wg_trace.hpp"
#include "tdist/work_queue.hpp"

using namespace tdist;
using namespace tdist::gemm;

struct GemmRsParams {
  SymmCtx ctx;
  const uint16_t* A;     // [M, K_local]
  const uint16_t* B;     // [N, K_local]
  uint16_t* scatter;     // symmetric, this parity: [w][M_per_rank][N]
  int* flags;            // symmetric (signal arena), this parity: [w * cpr]
  int* counters;         // local [w * cpr + 12]: [0, w*cpr) tiles finished per (destination, chunk), then the eight
                         // queue tickets, then the finished-workgroup counter.  Zero at launch; the last
                         // workgroup of every launch re-arms them (call_state.hpp).
  int qn;                // queues: qn column bands x (8 / qn) row-block residues
  int red_rows;          // rows per reduce item (multiple of 4)
  int race_debug;        // tdist/race_debug.hpp
  WgTrace trace;         // tdist/wg_trace.hpp (off unless the host armed a sink)
  int* err;
  uint16_t* out;         // [M_per_rank, N]
  long long lda, ldb, ldo;
  int M_per_rank, N, K;      // (8-bit operands: A, B, lda, ldb, K in 16-bit units = two elements)
  int rows_per_chunk;
  int epoch;
  const float* scale_a;      // optional fp32 scales of the partial sums (8-bit operands): per row of A [M] ...
  const float* scale_b;      // ... and per row of B [N]; both ranks' partials are de-quantised before the reduction
  int* dbg;               // optional progress trace: [grid][4] ints (last item, stage, items done, spare)
  // Device-side call numbering (tdist/call_state.hpp): when `state` is set, scatter / flags / epoch are
  // resolved on the device from the call number and the launch can be replayed from a HIP graph.
  int* state;
  uint16_t *scatter0, *scatter1;   // (separate fields, selected with ?: - indexing an array member with the
  int *flags0, *flags1;            //  runtime parity would push the whole argument block into scratch memory)
};

// Per-call view of the double-buffered objects (kept out of the argument block: writing into that large,
// dynamically indexed block would force a private copy of all of it into scratch memory).
struct RsCall {
  uint16_t* scatter;
  int* flags;
  int epoch;
};

constexpr int RS_TICKET_SLOT = LDS_BYTES;
constexpr int RED_COLS = THREADS * 8;  // 4096 columns per item

// Reduce item i of chunk c: p.red_rows rows x 4096 columns of this rank's output.  Four rows are in
// flight at a time (every thread: 4 rows x up to 4 sources x 16 bytes of loads before the first add).
// `acquired` is the last chunk this workgroup has already waited for: its partials landed before that
// wait's acquire and do not change within the call, so the flags are not polled again.
// `wq`: the draw for the next item was issued before the call; it is parked in LDS once the last loads have
// been consumed, i.e. before the item's last stores (collecting it afterwards would wait for those stores).
template <bool IS_BF16, class WQ>
__device__ __forceinline__ void reduce_item(const GemmRsParams& p, const RsCall& cv, int c, int i, int cpr, int ncb,
                                            int& acquired, WQ& wq) {
  const int w = p.ctx.num_ranks;
  if (acquired != c) {
    int failed = 0;
    const bool dead = load_relaxed_agent(p.err) != 0;
    if ((int)threadIdx.x < w) {
      if (dead || !spin_until_eq(cv.flags + threadIdx.x * cpr + c, cv.epoch, SCOPE_SYS)) failed = 1;
    }
    if (failed && !dead) atomicOr(p.err, 1);
    if (threadIdx.x < 64) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (__syncthreads_or(failed) == 0) acquired = c;
  }
  const int rb = i / ncb, cb = i - rb * ncb;
  const int col = cb * RED_COLS + threadIdx.x * 8;
  const int row0 = c * p.rows_per_chunk + rb * p.red_rows;
  const int row_end = min(min(row0 + p.red_rows, (c + 1) * p.rows_per_chunk), p.M_per_rank);
  const long long seg_stride = (long long)p.M_per_rank * p.N;
  // Threads beyond the last column run the same (uniform) control flow on column 0 and only skip the store:
  // stash() must stay workgroup-uniform, and per-lane conditions around the loads would serialise them.
  const bool active = col < p.N;
  const int colc = active ? col : 0;
  for (int row = row0; row < row_end; row += 4) {
    float acc[4][8];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
      for (int e = 0; e < 8; ++e) acc[rr][e] = 0.f;
    const uint16_t* src = cv.scatter + (long long)row * p.N + colc;
    for (int s0 = 0; s0 < w; s0 += 4) {
      uint4 v[4][4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (row + rr < row_end && s0 + u < w)
            v[rr][u] = *(const uint4*)(src + (long long)rr * p.N + (long long)(s0 + u) * seg_stride);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (row + rr < row_end && s0 + u < w) {
            const uint32_t x[4] = {v[rr][u].x, v[rr][u].y, v[rr][u].z, v[rr][u].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              if (IS_BF16) {
                acc[rr][2 * e] += __builtin_bit_cast(float, x[e] << 16);
                acc[rr][2 * e + 1] += __builtin_bit_cast(float, x[e] & 0xffff0000u);
              } else {
                typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
                const f16x2 h = __builtin_bit_cast(f16x2, x[e]);
                acc[rr][2 * e] += (float)h[0];
                acc[rr][2 * e + 1] += (float)h[1];
              }
            }
          }
    }
    if (row + 4 >= row_end) wq.stash();
#pragma unroll
    for (int rr = 0; rr < 4; ++rr)
      if (active && row + rr < row_end) {
        uint4 o;
        o.x = Frag<IS_BF16>::pack2(acc[rr][0], acc[rr][1]);
        o.y = Frag<IS_BF16>::pack2(acc[rr][2], acc[rr][3]);
        o.z = Frag<IS_BF16>::pack2(acc[rr][4], acc[rr][5]);
        o.w = Frag<IS_BF16>::pack2(acc[rr][6], acc[rr][7]);
        *(uint4*)(p.out + (long long)(row + rr) * p.ldo + col) = o;
      }
  }
}

// DT: operand type (gemm_gfx950.hpp DT_*).  Partials and output are 16-bit: f16 for f16 operands, bf16 otherwise.
template <int DT>
__global__ __launch_bounds__(THREADS, 2) void gemm_rs_kernel(GemmRsParams p) {
  constexpr bool IS_BF16 = DT != DT_F16;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  volatile int* s_ticket = (volatile int*)(smem + RS_TICKET_SLOT);
  const int w = p.ctx.num_ranks, rank = p.ctx.rank;
  const int cpr = p.M_per_rank / p.rows_per_chunk;
  const int mbpc = (p.rows_per_chunk + BM - 1) / BM;
  const int num_pid_n = (p.N + BN - 1) / BN;
  const int tiles_per_dc = mbpc * num_pid_n;          // tiles per (destination, chunk)
  const int ncb = (p.N + RED_COLS - 1) / RED_COLS;
  const int Rc = ((p.rows_per_chunk + p.red_rows - 1) / p.red_rows) * ncb;  // reduce items per chunk
  int* tickets = p.counters + w * cpr;
  const int qn = p.qn, qm = 8 / qn;
  const int cols_per = (num_pid_n + qn - 1) / qn;
  const int n_rb_all = w * mbpc;                      // row blocks of one chunk over all destinations
  // queue q = (row-block residue q / qn, column band q % qn) plus every eighth reduce item
  auto queue_cols = [=](int q) { return min(max(num_pid_n - (q % qn) * cols_per, 0), cols_per); };
  auto queue_gemm = [=](int q) {                      // GEMM items of queue q per chunk
    const int qmi = q / qn;
    const int n_rb = n_rb_all > qmi ? (n_rb_all - qmi + qm - 1) / qm : 0;
    return n_rb * queue_cols(q);
  };
  auto queue_red = [=](int q) { return Rc > q ? (Rc - q + 7) >> 3 : 0; };   // reduce items of queue q per chunk
  auto queue_size = [=](int q) { return cpr * (queue_gemm(q) + queue_red(q)); };
  WorkQueues<decltype(queue_size)> wq(tickets, s_ticket, 0, queue_size);
  int call = 0;
  RsCall cv{p.scatter, p.flags, p.epoch};
  WgTraceCursor tr = wg_trace_begin(p.trace);
  wg_trace(tr, 1, 0);
  if (p.state) {
    call = call_begin(p.state, s_ticket + 2);
    cv.scatter = call_parity(call) ? p.scatter1 : p.scatter0;
    cv.flags = call_parity(call) ? p.flags1 : p.flags0;
    cv.epoch = call_epoch(call);
  }
  race_debug_check_flags(p.race_debug, cv.flags, w * cpr, cv.epoch, p.err);

  int item = 0;
  int acquired = -1;
  bool have = wq.first(item);
  if (!have) have = wq.next(item);
  while (have) {
    const int q = wq.q;
    const int gq = queue_gemm(q), rq = queue_red(q), n_cols = queue_cols(q);
    if (p.dbg && threadIdx.x == 0) { p.dbg[blockIdx.x * 4 + 0] = item; p.dbg[blockIdx.x * 4 + 1] = 1; }
    wg_trace(tr, 2, item);

    bool is_gemm;
    int c, i;
    if (item < gq) {
      is_gemm = true;
      c = 0;
      i = item;
    } else {
      const int S = gq + rq;
      const int i2 = item - gq;
      const int blk = i2 / S, rem = i2 - blk * S;
      if (blk < cpr - 1) {
        if (rem < gq) {
          is_gemm = true;
          c = blk + 1;
          i = rem;
        } else {
          is_gemm = false;
          c = blk;
          i = rem - gq;
        }
      } else {
        is_gemm = false;
        c = cpr - 1;
        i = rem;
      }
    }

    if (!is_gemm) {
      wq.prefetch();
      reduce_item<IS_BF16>(p, cv, c, i * 8 + q, cpr, ncb, acquired, wq);
      if (p.dbg && threadIdx.x == 0) { p.dbg[blockIdx.x * 4 + 1] = 5; p.dbg[blockIdx.x * 4 + 2] += 1; }
      wg_trace(tr, 6, c);
      have = wq.next(item);
      continue;
    }

    const int rb = (i / n_cols) * qm + q / qn;        // row block of this chunk over all destinations
    const int pid_n = (q % qn) * cols_per + i % n_cols;
    const int dd = rb / mbpc, b = rb - dd * mbpc;
    const int d = (rank + 1 + dd) % w;
    const int r_in = c * p.rows_per_chunk + b * BM;    // row inside the destination's shard
    const int rows = min(BM, min((c + 1) * p.rows_per_chunk, p.M_per_rank) - r_in);
    const int g0 = d * p.M_per_rank + r_in;

    TileArgs t;
    t.A = p.A + (long long)g0 * p.lda;
    t.B = p.B + (long long)pid_n * BN * p.ldb;
    t.lda = p.lda;
    t.ldb = p.ldb;
    t.a_rows = rows;
    t.b_rows = p.N - pid_n * BN;
    t.K = p.K;
    Acc<DT> acc;
    mainloop_256_2ph<DT, false, 3>(t, smem, acc);
    // GEMM and reduce items differ in length by 10-40x: the next item is drawn only now, under the epilogue,
    // so a workgroup does not commit to another tile while idle workgroups could take it
    wq.prefetch();
    if (p.dbg && threadIdx.x == 0) p.dbg[blockIdx.x * 4 + 1] = 2;
    wg_trace(tr, 4, c);

    uint16_t* dst_buf = (d == rank) ? cv.scatter : symm_at(p.ctx, cv.scatter, d);
    uint16_t* cbase = dst_buf + ((long long)rank * p.M_per_rank + r_in) * p.N + pid_n * BN;
    const int nrem = p.N - pid_n * BN;
    const long long ldn = p.N;
    race_debug_delay(p.race_debug, (unsigned)i);
    // Tile stores are write-through (they leave this XCD's L2 at once, bound for a peer over xGMI
    // or for the local reduction on another XCD), so publishing needs no release fence: drain,
    // barrier, count.  The tile is staged through LDS so that every row leaves as one contiguous 512-byte
    // run: write-through stores are not merged in L2, and 64-byte pieces straight from the accumulator
    // layout cost 1.5x the time in HBM (and would halve the xGMI packet size).
    auto store_tile = [&](int r, int cc, const uint4& v) {
      if (r < rows && cc < nrem) st_wt128_sys(cbase + (long long)r * ldn + cc, v);
    };
    if (DT >= DT_FP8 && (p.scale_a || p.scale_b)) {
      // de-quantise: accumulator tile (m, n) of this lane is row 16 m + (lane & 15) of the wave's 128 rows,
      // columns 16 n + 4 (lane >> 4) .. + 3 of its 64 columns
      const int wv = threadIdx.x >> 6, ln = threadIdx.x & 63;
      const int rloc = (wv >> 2) * 128 + (ln & 15), cloc = (wv & 3) * 64 + (ln >> 4) * 4;
      float cs[4][4];
#pragma unroll
      for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int cc = cloc + n * 16 + e;
          cs[n][e] = (p.scale_b && cc < nrem) ? p.scale_b[pid_n * BN + cc] : 1.f;
        }
      epilogue_256_from<DT_BF16>(
          [&](int m, f32x4(&v)[4]) {
            const int rr = rloc + m * 16;
            const float rs = (p.scale_a && rr < rows) ? p.scale_a[g0 + rr] : 1.f;
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
              for (int e = 0; e < 4; ++e) v[n][e] = Frag<DT>::val(acc.v[m][n][e]) * rs * cs[n][e];
          },
          smem, store_tile);
    } else {
      epilogue_256<DT>(acc, smem, store_tile);
    }
    if (p.dbg && threadIdx.x == 0) p.dbg[blockIdx.x * 4 + 1] = 3;
    drain_stores();
    race_debug_delay(p.race_debug, (unsigned)i + 7u);
    __syncthreads();
    if (threadIdx.x == 0) {
      // Every tile's stores completed before its own add; the add that completes the chunk is
      // ordered after all earlier adds on the same word, so the flag follows all of the chunk's data.
      const int prev = atom_add_relaxed_agent(p.counters + d * cpr + c, 1);
      if (prev == tiles_per_dc - 1) {
        int* flag = ((d == rank) ? cv.flags : symm_at(p.ctx, cv.flags, d)) + rank * cpr + c;
        store_relaxed_system(flag, cv.epoch);
      }
    }
    wg_trace(tr, 5, c);
    have = wq.next(item);
  }
  wg_trace(tr, 7, 0);
  call_end(p.state, call, tickets + 8, p.counters, w * cpr + 8);
}

TDIST_API int tdist_gemm_rs(const tdist::SymmCtx* ctx, const void* A, const void* B, void* scatter, int* flags,
                            int* counters, int* err, void* out, long long lda, long long ldb, long long ldo,
                            int M_per_rank, int N, int K, int rows_per_chunk, int epoch, int num_ctas, int dtype,
                            void* stream, int* dbg, int* state, void* scatter1, int* flags1, const float* scale_a,
                            const float* scale_b) {
  // dtype: 0 = bf16, 1 = fp16, 2 = fp8 e4m3, 3 = fp8 e5m2, 4 = int8 (8-bit operands: bf16 partials and output, scaled by
  // scale_a [M] x scale_b [N] when given); K, lda, ldb in ELEMENTS of the operand type
  TDIST_REQUIRE(dtype >= 0 && dtype <= 4, "dtype must be 0..4");
  const int per16 = dtype >= 2 ? 2 : 1;
  TDIST_REQUIRE(K % per16 == 0 && lda % per16 == 0 && ldb % per16 == 0, "odd 8-bit strides");
  K /= per16;
  lda /= per16;
  ldb /= per16;
  TDIST_REQUIRE(K >= 128 && K % 64 == 0, "K must be a multiple of 64 (16-bit) / 128 (8-bit) and at least two such steps");
  TDIST_REQUIRE(N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldo % 8 == 0, "N/lda/ldb/ldo must be multiples of 8");
  TDIST_REQUIRE((((uintptr_t)A | (uintptr_t)B | (uintptr_t)out | (uintptr_t)scatter) & 15) == 0,
                "operands must be 16-byte aligned");
  TDIST_REQUIRE(M_per_rank > 0 && rows_per_chunk > 0 && M_per_rank % rows_per_chunk == 0,
                "M_per_rank must be a positive multiple of rows_per_chunk");
  const int w = ctx->num_ranks;
  TDIST_REQUIRE(w >= 1 && w <= TDIST_MAX_RANKS, "bad world size");
  GemmRsParams p;
  p.ctx = *ctx;
  p.A = (const uint16_t*)A;
  p.B = (const uint16_t*)B;
  p.scatter = (uint16_t*)scatter;
  p.flags = flags;
  p.counters = counters;
  p.err = err;
  p.out = (uint16_t*)out;
  p.lda = lda;
  p.ldb = ldb;
  p.ldo = ldo;
  p.M_per_rank = M_per_rank;
  p.N = N;
  p.K = K;
  p.rows_per_chunk = rows_per_chunk;
  p.epoch = epoch;
  p.dbg = dbg;
  p.scale_a = scale_a;
  p.scale_b = scale_b;
  p.race_debug = race_debug_enabled();
  p.trace = wg_trace_sink();
  {
    const int npn = (N + BN - 1) / BN;
    int qn = 1;
    while (qn < 8 && qn * 2 <= npn) qn *= 2;
    p.qn = qn;
  }
  TDIST_REQUIRE(!state || (scatter1 && flags1 && (((uintptr_t)scatter1) & 15) == 0),
                "device-side call numbering needs both parities");
  p.state = state;
  p.scatter0 = (uint16_t*)scatter;
  p.scatter1 = (uint16_t*)scatter1;
  p.flags0 = flags;
  p.flags1 = flags1;
  const int cpr = M_per_rank / rows_per_chunk;
  const size_t counter_bytes = (((size_t)(w * cpr + 12) * sizeof(int)) + 15) & ~(size_t)15;
  typedef void (*rs_kern_t)(GemmRsParams);
  static const rs_kern_t kerns[5] = {gemm_rs_kernel<DT_BF16>, gemm_rs_kernel<DT_F16>, gemm_rs_kernel<DT_FP8>,
                                     gemm_rs_kernel<DT_BF8>, gemm_rs_kernel<DT_I8>};
  static bool attr_set = false;
  if (!attr_set) {
    for (int i = 0; i < 5; ++i)
      TDIST_HIP_TRY(hipFuncSetAttribute((const void*)kerns[i], hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES + 16));
    attr_set = true;
  }
  // The whole grid must be resident at once (reduce items spin on flags raised by this rank's and the
  // peers' GEMM items).
  static int num_cus = 0;
  if (!num_cus) {
    int dev = 0;
    TDIST_HIP_TRY(hipGetDevice(&dev));
    TDIST_HIP_TRY(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev));
  }
  int grid = num_ctas > 0 ? num_ctas : num_cus;
  if (grid > num_cus) grid = num_cus;
  {
    // Reduce items: as coarse as still gives every workgroup one item per chunk (the last chunk's
    // reduction is the exposed tail and should run on the whole chip).
    const int ncb = (N + RED_COLS - 1) / RED_COLS;
    int rr = 16;
    while (rr > 4 && ((rows_per_chunk + rr - 1) / rr) * ncb < grid) rr >>= 1;
    p.red_rows = rr;
  }
  static const bool trace = getenv("TDIST_TRACE_LAUNCH") != nullptr;
  if (trace)
    fprintf(stderr,
            "[tdist] gemm_rs rank %d/%d A=%p B=%p scatter0=%p scatter1=%p flags0=%p flags1=%p counters=%p(%zu B) "
            "err=%p out=%p state=%p Mr=%d N=%d K=%d rpc=%d grid=%d\n",
            ctx->rank, w, A, B, scatter, scatter1, (void*)flags, (void*)flags1, (void*)counters, counter_bytes,
            (void*)err, out, (void*)state, M_per_rank, N, K, rows_per_chunk, grid);
  hipLaunchKernelGGL(kerns[dtype], dim3(grid), dim3(THREADS), LDS_BYTES + 16, (hipStream_t)stream, p);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
