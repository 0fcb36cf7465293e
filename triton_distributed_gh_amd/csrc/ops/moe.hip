// Mixture-of-experts kernels for tensor-parallel MoE:
//   moe_align        - sort (token, k) pairs by (expert, source rank), pad each expert to the 256-row
//                      tile, emit per-tile expert id and the range of source ranks the tile touches
//   moe_group_gemm   - persistent grouped GEMM on the 256x256 MFMA engine with gathered A rows
//                      (AllGather + MoE: waits per tile on the arrival flags of the ranks it needs;
//                      MoE + ReduceScatter: scales rows by the routing weight in the epilogue)
//   moe_topk_reduce_rs - sum the topk expert outputs of every token, push the result to the rank
//                      that owns the token, wait for all ranks, reduce the w partials in fp32
//
// Role in the reference: csrc/lib/moe_utils.cu:60-353 (align, single-CTA and per-rank variants),
// kernels/nvidia/allgather_group_gemm.py:84-165 (sorted gather index) and :534-608 (grouped GEMM
// consumer), kernels/nvidia/moe_reduce_rs.py:166-244 (grouped GEMM with weights) and :292-376
// (topk reduce + reduce-scatter), kernels/nvidia/threadblock_swizzle_ag_moe.cu (tile order by
// arrival stage: here tiles are simply SORTED by the last source rank they need, in arrival
// order, on the device side of `moe_align`).
#include "tdist/ag_push.hpp"
#include "tdist/cache_policy.hpp"
#include "tdist/call_state.hpp"
#include "tdist/common.hpp"
#include "tdist/gemm_gfx950.hpp"
#include "tdist/primitives.hpp"
#include "tdist/wg_trace.hpp"
#include "tdist/work_queue.hpp"

using namespace tdist;
using namespace tdist::gemm;

// ------------------------------------------------------------------------------------------------
// align, two launches over G workgroups of 1024 threads (G = ceil(n / ALIGN_PER_WG)):
//   count   - workgroup g histograms its slice of the (token, k) pairs by (expert, source rank) in LDS, writes
//             the E*w counts to part[g][.] and fills its share of sorted_ids with -1 (padding rows);
//   scatter - every workgroup sums the G partial histograms (its own start inside each bin = the counts of the
//             workgroups before it), thread 0 walks the E*w bins once for the padded expert offsets, and the
//             workgroup scatters its pairs through LDS cursors; workgroup 0 also emits the per-block tables
//             (expert, source-rank range, arrival order) and the meta words.
// One workgroup did all of this in 32-40 us at 32k pairs (every pass is a chain of LDS atomics with results on
// ONE compute unit); spread over 16 compute units the two launches take a few microseconds each.
// LDS per workgroup: hist[E*w], base[E*w], pad_off[E+1], stage_of[cap/256].
// ------------------------------------------------------------------------------------------------
struct AlignParams {
  const int* topk_ids;    // [n] expert of flat pair i = token*topk + k
  int* sorted_ids;        // [cap] flat pair index or -1 (cap = n + E*256 rounded)
  int* expert_of_block;   // [cap/256]
  int* seg_range;         // [cap/256][2] first / last source rank present in the block
  int* order;             // [cap/256] block ids, each of the eight contiguous ranges sorted by arrival (rotated last source rank)
  int* meta;              // [0] num_blocks, [1] padded rows, [2..2+E) tokens per expert
  int* part;              // [G][E*w] per-workgroup histograms (written by count, read by scatter)
  int n, E, w, topk, tokens_per_rank, rank, cap;
};

constexpr int ALIGN_PER_WG = 2048;     // pairs per workgroup (two per thread)

__device__ __forceinline__ int align_bin(const AlignParams& p, int i, int e) {
  return e * p.w + min((i / p.topk) / p.tokens_per_rank, p.w - 1);
}

__global__ __launch_bounds__(1024) void moe_align_count_kernel(AlignParams p) {
  extern __shared__ int lds[];
  int* hist = lds;
  const int EW = p.E * p.w;
  const int G = gridDim.x, g = blockIdx.x;
  for (int i = threadIdx.x; i < EW; i += blockDim.x) hist[i] = 0;
  // this workgroup's share of the padding fill
  const int per = (p.cap + G - 1) / G;
  for (int i = g * per + threadIdx.x; i < min(p.cap, (g + 1) * per); i += blockDim.x) p.sorted_ids[i] = -1;
  __syncthreads();
  const int i0 = g * ALIGN_PER_WG + threadIdx.x, i1 = i0 + blockDim.x;
  const int e0 = i0 < p.n ? p.topk_ids[i0] : -1;
  const int e1 = i1 < p.n ? p.topk_ids[i1] : -1;
  if (e0 >= 0) atomicAdd(&hist[align_bin(p, i0, e0)], 1);
  if (e1 >= 0) atomicAdd(&hist[align_bin(p, i1, e1)], 1);
  __syncthreads();
  for (int i = threadIdx.x; i < EW; i += blockDim.x) p.part[g * EW + i] = hist[i];
}

__global__ __launch_bounds__(1024) void moe_align_scatter_kernel(AlignParams p) {
  extern __shared__ int lds[];
  int* hist = lds;                       // [E*w] totals, later this workgroup's cursors
  int* base = lds + p.E * p.w;           // [E*w] start row of (e, s)
  int* pad_off = base + p.E * p.w;       // [E+1]
  int* stage_of = pad_off + p.E + 1;     // [cap / 256] arrival stage of every block (workgroup 0)
  const int EW = p.E * p.w;
  const int G = gridDim.x, g = blockIdx.x;
  // the pairs of this workgroup (loads in flight while the histograms are summed)
  const int i0 = g * ALIGN_PER_WG + threadIdx.x, i1 = i0 + blockDim.x;
  const int e0 = i0 < p.n ? p.topk_ids[i0] : -1;
  const int e1 = i1 < p.n ? p.topk_ids[i1] : -1;
  for (int b = threadIdx.x; b < EW; b += blockDim.x) {
    int tot = 0, before = 0;
    for (int k = 0; k < G; ++k) {
      const int c = p.part[k * EW + b];
      tot += c;
      before += k < g ? c : 0;
    }
    hist[b] = tot;
    base[b] = before;                    // (start inside the bin; the bin's own start is added below)
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int off = 0;
    for (int e = 0; e < p.E; ++e) {
      pad_off[e] = off;
      int tot = 0;
      for (int s = 0; s < p.w; ++s) {
        const int c = hist[e * p.w + s];
        const int mine = base[e * p.w + s];
        base[e * p.w + s] = off + tot;   // global start row of (e, s)
        hist[e * p.w + s] = off + tot + mine;      // this workgroup's cursor
        tot += c;
        // keep the count for the per-block tables below (workgroup 0 needs start + count of every bin)
        lds[2 * EW + p.E + 1 + (p.cap / BM) + e * p.w + s] = c;
      }
      if (g == 0) p.meta[2 + e] = tot;
      off += (tot + BM - 1) / BM * BM;
    }
    pad_off[p.E] = off;
    if (g == 0) {
      p.meta[0] = off / BM;
      p.meta[1] = off;
    }
  }
  __syncthreads();
  if (e0 >= 0) p.sorted_ids[atomicAdd(&hist[align_bin(p, i0, e0)], 1)] = i0;
  if (e1 >= 0) p.sorted_ids[atomicAdd(&hist[align_bin(p, i1, e1)], 1)] = i1;
  if (g != 0) return;
  // ---- workgroup 0: per-block expert id, source-rank range, arrival order ----
  const int* count = lds + 2 * EW + p.E + 1 + (p.cap / BM);
  const int nblocks = pad_off[p.E] / BM;
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
    const int r0 = b * BM, r1 = r0 + BM;
    int e = 0;
    while (pad_off[e + 1] <= r0) ++e;
    int smin = p.w, smax = -1;
    for (int s = 0; s < p.w; ++s) {
      const int a = base[e * p.w + s], c = a + count[e * p.w + s];
      if (c > a && a < r1 && c > r0) {
        smin = min(smin, s);
        smax = max(smax, s);
      }
    }
    const int lo = smin > smax ? 0 : smin, hi = smin > smax ? 0 : smax;
    p.expert_of_block[b] = e;
    p.seg_range[2 * b] = lo;
    p.seg_range[2 * b + 1] = hi;
    // arrival order: a block is ready once its LAST needed source (in push-arrival order: own rank first,
    // then rank+1, rank+2, ... as pushed by the producers) has landed
    int stage = 0;
    for (int s = lo; s <= hi; ++s) stage = max(stage, (s - p.rank + p.w) % p.w);
    stage_of[b] = stage;
  }
  __syncthreads();
  // The grouped GEMM cuts the blocks into eight contiguous ranges, one per XCD (range q = blocks
  // [q * nblocks / 8, (q + 1) * nblocks / 8): neighbouring blocks share an expert, i.e. its B panels), and every XCD
  // walks its range in arrival order.  Stable counting sort on the stage INSIDE each range, one thread per block:
  // position = first block of the range + its blocks of an earlier stage + lower-numbered ones of the same stage.
  for (int b = threadIdx.x; b < nblocks; b += blockDim.x) {
    const int mine = stage_of[b];
    int q = 0;
    while (q < 7 && b >= (int)((long long)(q + 1) * nblocks / 8)) ++q;
    const int lo = (int)((long long)q * nblocks / 8), hi = (int)((long long)(q + 1) * nblocks / 8);
    int pos = lo;
    for (int o = lo; o < hi; ++o) {
      const int st = stage_of[o];
      pos += (st < mine || (st == mine && o < b)) ? 1 : 0;
    }
    p.order[pos] = b;
  }
}

// ------------------------------------------------------------------------------------------------
// grouped GEMM
// ------------------------------------------------------------------------------------------------
struct GroupGemmParams {
  const uint16_t* A;          // rows gathered through a_rows
  const uint16_t* B;          // [E][N][K]
  uint16_t* out;              // rows scattered through out_rows
  const int* sorted_ids;      // [cap] flat pair index or -1
  const int* expert_of_block;
  const int* seg_range;       // may be null (no waiting)
  const int* order;           // may be null (natural order)
  const int* meta;            // meta[0] = number of 256-row blocks
  const float* row_scale;     // [n] routing weight of flat pair i, or null
  const int* flags;           // arrival flags [w * cpr] (epoch valued) or null
  int* ticket;                // [8] queue tickets, then [cpr] producer counters; zero at launch
  int* err;
  long long lda, ldb, ldo, b_expert_stride;
  int N, K;
  int topk;                   // A row of flat pair i: a_mode 0 -> i / topk (token row), 1 -> i
  int a_mode;
  int cpr, epoch;
  // In-kernel all-gather producer (AG + MoE, tdist/ag_push.hpp): the first push_wgs workgroups store this
  // rank's shard into every rank's workspace (the own one included - tiles read every row from there) before
  // they join the consumers.  push_wgs == 0: the data comes from elsewhere.
  SymmCtx ctx;
  const uint16_t* a_local;
  long long lda_local;
  uint16_t* ws;               // symmetric [M, K] workspace of this call = A
  int* push_flags;            // = flags, as the symmetric pointer the producers store through
  int* push_done;
  int M_per_rank, rows_per_chunk, rank, world, push_wgs, straddle, race_debug;
  int push_store = 0;       // see ag_push.hpp
  WgTrace trace;              // tdist/wg_trace.hpp (off unless the host armed a sink)
  // Device-side call numbering (tdist/call_state.hpp).  With `state` set, A / ws / flags / epoch above are
  // ignored: parity and flag value come from the call number on the device (ws, flags = parity 0; ws1, flags1 =
  // parity 1), so the launch is the same every time and replays from a HIP graph.  The last workgroup of every
  // launch zeroes the n_counters ints from `ticket` on (with or without `state`): no memset in front.
  int* state;
  uint16_t* ws1;
  int* flags1;
  int* fin;
  int n_counters;
};

struct GroupCall {
  uint16_t* ws;
  int* flags;
  int epoch;
};

constexpr int GG_TICKET = LDS_BYTES;        // 16 bytes: work-queue mailboxes
constexpr int GG_ROWS = LDS_BYTES + 16;     // two sets of: int[256] gather rows, int[256] out rows, float[256] scales

template <bool IS_BF16>
__global__ __launch_bounds__(THREADS, 2) void moe_group_gemm_kernel(GroupGemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  volatile int* s_ticket = (volatile int*)(smem + GG_TICKET);
  const int num_pid_n = (p.N + BN - 1) / BN;
  const int nblocks = p.meta[0];
  // Eight work queues, one per XCD (tdist/work_queue.hpp).  Queue q holds the contiguous range of row blocks
  // [q * nblocks / 8, (q + 1) * nblocks / 8) - neighbouring blocks belong to one expert, so an XCD's L2 sees the B
  // panels of one or two experts instead of an eighth of every expert's rows (with blocks dealt round-robin each XCD
  // held ~2 row blocks per expert and streamed all of B for them: 41 - 50 % L2 misses against 24 % for a dense GEMM
  // of the same shape; 29 - 32 % now).  Inside a queue the items run in super-tiles of 4 row blocks x 8 column tiles
  // (row-group major, then column chunk, then row, then column): the 32 workgroups of an XCD work on 4 + 8 operand
  // panels at a time instead of 2 + 16.  The first item of every workgroup is static, later ones are drawn ahead of
  // the main loop, and a workgroup whose queue has run dry moves on to one that has not.
  constexpr int SR = 4, SC = 8;
  auto q_lo = [=](int q) { return (int)((long long)q * nblocks / 8); };
  auto queue_size = [=](int q) { return (q_lo(q + 1) - q_lo(q)) * num_pid_n; };
  // item `local` of queue q -> (index into the queue's arrival order, column tile)
  auto decode = [=](int q, int local, int& bidx, int& pid_n) {
    const int nbq = q_lo(q + 1) - q_lo(q);
    const int groups = (nbq + SR - 1) / SR;
    const int g = min(local / (SR * num_pid_n), groups - 1);
    const int rows = min(SR, nbq - g * SR);
    int rem = local - g * SR * num_pid_n;
    const int chunks = (num_pid_n + SC - 1) / SC;
    const int k = min(rem / (rows * SC), chunks - 1);
    rem -= k * rows * SC;
    const int ck = min(SC, num_pid_n - k * SC);
    const int r = rem / ck;
    bidx = q_lo(q) + g * SR + r;
    pid_n = k * SC + (rem - r * ck);
  };
  WorkQueues<decltype(queue_size)> wq(p.ticket, s_ticket, p.push_wgs, queue_size);
  int local = 0;
  bool have;
  WgTraceCursor tr = wg_trace_begin(p.trace);
  wg_trace(tr, 1, 0);
  int call = 0;
  GroupCall cv{p.ws, p.push_flags, p.epoch};
  const uint16_t* A = p.A;
  const int* flags = p.flags;
  if (p.state) {
    call = call_begin(p.state, s_ticket + 2);       // (mailbox ints 1 and 2 are not used by the work queues)
    if (call_parity(call)) {
      cv.ws = p.ws1;
      cv.flags = p.flags1;
    }
    cv.epoch = call_epoch(call);
    A = cv.ws;
    flags = cv.flags;
  }
  if ((int)blockIdx.x < p.push_wgs) {
    push_item(p, cv, (int)blockIdx.x);
    wg_trace(tr, 6, 0);
    have = wq.next(local);
  } else {
    have = wq.first(local);
    if (!have) have = wq.next(local);
  }
  // Row tables of an item (gather row, output row, routing weight per tile row) are fetched into registers
  // as soon as the item is known - for all but the first that is before the PREVIOUS item's epilogue, whose
  // stores then cover the two dependent look-ups (sorted id, then weight) - and written to LDS at the top.
  int it = 0;
  int t_flat = -1;
  float t_scale = 1.f;
  auto fetch_tables = [&](int item_local, int item_q) {
    int bi, pn;
    decode(item_q, item_local, bi, pn);
    const int b = p.order ? p.order[bi] : bi;
    if (threadIdx.x < BM) {
      t_flat = p.sorted_ids[b * BM + threadIdx.x];
      t_scale = (t_flat >= 0 && p.row_scale) ? p.row_scale[t_flat] : 1.f;
    }
  };
  if (have) fetch_tables(local, wq.q);
  while (have) {
    int bi, pid_n;
    decode(wq.q, local, bi, pid_n);
    const int b = p.order ? p.order[bi] : bi;
    const int e = p.expert_of_block[b];
    // Two sets in turn: the epilogue of item i still reads its set while a faster wave already fills the
    // other one for item i + 1 (set i is rewritten for item i + 2, behind the main-loop barriers of i + 1).
    int* s_arow = (int*)(smem + GG_ROWS) + (it & 1) * 3 * BM;
    int* s_orow = s_arow + BM;
    float* s_scale = (float*)(s_orow + BM);
    ++it;
    if (threadIdx.x < BM) {
      s_arow[threadIdx.x] = t_flat < 0 ? -1 : (p.a_mode == 0 ? t_flat / p.topk : t_flat);
      s_orow[threadIdx.x] = t_flat;
      s_scale[threadIdx.x] = t_scale;
    }
    wg_trace(tr, 2, b);
    wq.prefetch();
    if (flags && p.seg_range) {
      const int f0 = p.seg_range[2 * b] * p.cpr, f1 = (p.seg_range[2 * b + 1] + 1) * p.cpr;
      wait(flags + f0, f1 - f0, SCOPE_SYS, SEM_ACQUIRE, cv.epoch, p.err);
    } else {
      __syncthreads();
    }
    wg_trace(tr, 3, b);
    TileArgs t;
    t.A = A;
    t.B = p.B + (long long)e * p.b_expert_stride + (long long)pid_n * BN * p.ldb;
    t.lda = p.lda;
    t.ldb = p.ldb;
    t.a_rows = BM;
    t.b_rows = p.N - pid_n * BN;
    t.K = p.K;
    t.a_gather = s_arow;
    Acc<IS_BF16> acc;
    // accumulator columns permuted (PERM_B): the tile is stored straight from registers, and the next item
    // is looked up before the stores so that no barrier follows them (see ops/ag_gemm.hip)
    mainloop_256_2ph<IS_BF16, true, 3>(t, smem, acc);
    wg_trace(tr, 4, b);
    wq.stash();
    int next_local = 0;
    const bool have_next = wq.next(next_local);
    if (have_next) fetch_tables(next_local, wq.q);
    if (p.row_scale) {
      // accumulator layout: lane holds tile-row (l & 15) of each 16-row block
      const int w_ = threadIdx.x >> 6, l = threadIdx.x & 63;
      const int wr = w_ >> 2;
#pragma unroll
      for (int m = 0; m < 8; ++m) {
        const float sc = s_scale[wr * 128 + m * 16 + (l & 15)];
#pragma unroll
        for (int n = 0; n < 4; ++n) acc.v[m][n] *= sc;
      }
    }
    const int nrem = p.N - pid_n * BN;
    uint16_t* obase = p.out + pid_n * BN;
    const long long ldo = p.ldo;
    epilogue_256_direct<IS_BF16>(acc, [&](int r, int c, const uint4& v) {
      const int orow = s_orow[r];
      if (orow >= 0 && c < nrem) *(uint4*)(obase + (long long)orow * ldo + c) = v;
    });
    wg_trace(tr, 5, b);
    have = have_next;
    local = next_local;
  }
  wg_trace(tr, 7, 0);
  call_end(p.state, call, p.fin, p.ticket, p.n_counters);
}

// ------------------------------------------------------------------------------------------------
// topk reduce + reduce-scatter.  gemm_out [ntokens*topk, N] (already weight-scaled).  A launch may cover a
// band of N columns only (row strides `ld`): the caller cuts N into chunks and runs the reduce of chunk c next
// to the grouped GEMM of chunk c + 1 on a second stream, each chunk with flags and counters of its own.
// Work items hold RS_TG consecutive tokens of one owner (per-item costs - ticket, drain, barrier,
// counter - are paid once per group, and a thread keeps 4 x topk 16-byte loads in flight):
//   push items   [0, w*gpo)      : item -> owner d = (rank+1 + item%w) % w, group item/w of that
//                                  owner's tpr tokens: sum the topk rows of each token, store to
//                                  the owner's scatter slot [rank]; the group that completes an
//                                  owner raises flags[rank] := epoch there
//   reduce items [w*gpo, w*gpo+gpo): wait for all w sources, out[t_local] = sum over the w slots
// (gpo = groups per owner).  Push items come first in ticket order, so a workgroup that waits in
// a reduce item can only be waiting for other ranks, never for unclaimed local work.
// ------------------------------------------------------------------------------------------------
struct TopkRsParams {
  SymmCtx ctx;
  const uint16_t* gemm_out;
  uint16_t* scatter;     // symmetric [w][tpr][ld]
  int* flags;            // symmetric [w]
  int* counters;         // local [w] + ticket at [w]; zero at launch, re-armed by the last workgroup;
                         // [TDIST_MAX_RANKS + 1] = finished-workgroup counter
  int* err;
  uint16_t* out;         // [tpr, ld]
  long long ld;          // row stride of gemm_out / scatter / out in elements (N of the whole problem)
  int ntokens, topk, N, tpr, epoch;
  // device-side call numbering (tdist/call_state.hpp): scatter / flags = parity 0, scatter1 / flags1 = parity 1
  int* state;
  uint16_t* scatter1;
  int* flags1;
};

constexpr int RS_TG = 8;

__device__ __forceinline__ void acc_bf16x8(float (&acc)[8], const uint4& v) {
  const uint32_t x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    acc[2 * i] += __builtin_bit_cast(float, x[i] << 16);
    acc[2 * i + 1] += __builtin_bit_cast(float, x[i] & 0xffff0000u);
  }
}

__device__ __forceinline__ uint4 pack_bf16x8(const float (&acc)[8]) {
  uint4 o;
  o.x = Frag<true>::pack2(acc[0], acc[1]);
  o.y = Frag<true>::pack2(acc[2], acc[3]);
  o.z = Frag<true>::pack2(acc[4], acc[5]);
  o.w = Frag<true>::pack2(acc[6], acc[7]);
  return o;
}

__global__ __launch_bounds__(256) void moe_topk_reduce_rs_kernel(TopkRsParams p) {
  __shared__ int s_item;
  __shared__ int s_call;
  const int w = p.ctx.num_ranks, rank = p.ctx.rank;
  int call = 0;
  uint16_t* scatter = p.scatter;
  int* flags = p.flags;
  int epoch = p.epoch;
  if (p.state) {
    call = call_begin(p.state, &s_call);
    if (call_parity(call)) {
      scatter = p.scatter1;
      flags = p.flags1;
    }
    epoch = call_epoch(call);
  }
  const int gpo = (p.tpr + RS_TG - 1) / RS_TG;
  const int n_push = w * gpo;
  const int total = n_push + gpo;
  const int cpr = p.N >> 3;                     // 16-byte pieces per row
  while (true) {
    __syncthreads();
    if (threadIdx.x == 0) s_item = atomicAdd(p.counters + w, 1);
    __syncthreads();
    const int item = __builtin_amdgcn_readfirstlane(s_item);
    if (item >= total) break;
    if (item < n_push) {
      // consecutive items go to different owners, so all links carry traffic at any moment
      const int d = (rank + 1 + item % w) % w;
      const int tl0 = (item / w) * RS_TG;
      const int ntl = min(RS_TG, p.tpr - tl0);  // >= 1
      const uint16_t* src = p.gemm_out + ((long long)d * p.tpr + tl0) * p.topk * p.ld;
      uint16_t* dst = ((d == rank) ? scatter : symm_at(p.ctx, scatter, d)) + ((long long)rank * p.tpr + tl0) * p.ld;
      const int units = ntl * cpr;              // one unit = one 16-byte output piece
      for (int u0 = threadIdx.x; u0 < units; u0 += 4 * blockDim.x) {
        float acc[4][8];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[j][i] = 0.f;
          const int u = u0 + j * blockDim.x;
          if (u < units) {
            const int tl = u / cpr, c = (u - tl * cpr) << 3;
            for (int k = 0; k < p.topk; ++k)
              acc_bf16x8(acc[j], *(const uint4*)(src + ((long long)tl * p.topk + k) * p.ld + c));
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int u = u0 + j * blockDim.x;
          if (u < units) {
            const int tl = u / cpr, c = (u - tl * cpr) << 3;
            st_wt128_sys(dst + (long long)tl * p.ld + c, pack_bf16x8(acc[j]));
          }
        }
      }
      // write-through stores: drain + barrier publish them, no release fence; the add that
      // completes an owner's groups is ordered after every earlier add on that word
      drain_stores();
      __syncthreads();
      if (threadIdx.x == 0) {
        const int prev = atom_add_relaxed_agent(p.counters + d, 1);
        if (prev == gpo - 1) {
          store_relaxed_system(((d == rank) ? flags : symm_at(p.ctx, flags, d)) + rank, epoch);
        }
      }
    } else {
      const int tl0 = (item - n_push) * RS_TG;
      const int ntl = min(RS_TG, p.tpr - tl0);
      wait(flags, w, SCOPE_SYS, SEM_ACQUIRE, epoch, p.err);
      const int units = ntl * cpr;
      for (int u0 = threadIdx.x; u0 < units; u0 += 2 * blockDim.x) {
        float acc[2][8];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int i = 0; i < 8; ++i) acc[j][i] = 0.f;
          const int u = u0 + j * blockDim.x;
          if (u < units) {
            const int tl = u / cpr, c = (u - tl * cpr) << 3;
            for (int s = 0; s < w; ++s)
              acc_bf16x8(acc[j], *(const uint4*)(scatter + ((long long)s * p.tpr + tl0 + tl) * p.ld + c));
          }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int u = u0 + j * blockDim.x;
          if (u < units) {
            const int tl = u / cpr, c = (u - tl * cpr) << 3;
            *(uint4*)(p.out + (long long)(tl0 + tl) * p.ld + c) = pack_bf16x8(acc[j]);
          }
        }
      }
    }
  }
  call_end(p.state, call, p.counters + TDIST_MAX_RANKS + 1, p.counters, w + 1);
}

// ------------------------------------------------------------------------------------------------
// `part`: int32 scratch of at least tdist_moe_align_part_ints(n, E, w) words (any contents).
TDIST_API int tdist_moe_align_part_ints(int n, int E, int w) { return ((n + ALIGN_PER_WG - 1) / ALIGN_PER_WG) * E * w; }

TDIST_API int tdist_moe_align(const int* topk_ids, int* sorted_ids, int* expert_of_block, int* seg_range, int* order,
                              int* meta, int n, int E, int w, int topk, int tokens_per_rank, int rank, int cap,
                              void* stream, int* part) {
  TDIST_REQUIRE(E > 0 && w > 0 && cap % BM == 0 && cap >= n && n > 0 && part, "bad align arguments");
  const size_t lds = (size_t)(3 * E * w + E + 1 + cap / BM) * sizeof(int);
  TDIST_REQUIRE(lds <= 160 * 1024, "too many experts x ranks for one workgroup");
  AlignParams p{topk_ids, sorted_ids, expert_of_block, seg_range, order, meta, part, n, E, w, topk, tokens_per_rank, rank, cap};
  static bool attr_set = false;
  if (!attr_set) {
    TDIST_HIP_TRY(hipFuncSetAttribute((const void*)moe_align_count_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    TDIST_HIP_TRY(hipFuncSetAttribute((const void*)moe_align_scatter_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  const int G = (n + ALIGN_PER_WG - 1) / ALIGN_PER_WG;
  hipLaunchKernelGGL(moe_align_count_kernel, dim3(G), dim3(1024), (size_t)E * w * sizeof(int), (hipStream_t)stream, p);
  hipLaunchKernelGGL(moe_align_scatter_kernel, dim3(G), dim3(1024), lds, (hipStream_t)stream, p);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

// `ticket`: 8 + cpr + 1 ints, zero at the first launch (queue tickets, the producer's per-chunk counters, the
// finished-workgroup counter); every launch leaves them zero again.  `state` != null: device-side call
// numbering - `ws` / `flags` are then parity 0, `ws1` / `flags1` parity 1, `A` and `epoch` are ignored.
// Producer (AG + MoE): ctx / a_local / ws / M_per_rank / rows_per_chunk / push_wgs describe the in-kernel
// all-gather (push_wgs == 0 or ctx == null: no producer); `flags` is then also the symmetric flag array the
// producers raise on every rank.
TDIST_API int tdist_moe_group_gemm(const void* A, const void* B, void* out, const int* sorted_ids,
                                   const int* expert_of_block, const int* seg_range, const int* order, const int* meta,
                                   const float* row_scale, const int* flags, int* ticket, int* err, long long lda,
                                   long long ldb, long long ldo, long long b_expert_stride, int N, int K, int topk,
                                   int a_mode, int cpr, int epoch, int num_ctas, int dtype, void* stream,
                                   const tdist::SymmCtx* ctx, const void* a_local, long long lda_local, void* ws,
                                   int M_per_rank, int rows_per_chunk, int push_wgs, int* state, void* ws1,
                                   int* flags1) {
  TDIST_REQUIRE(K >= 128 && K % 64 == 0, "K must be a multiple of 64 and >= 128");
  TDIST_REQUIRE(N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldo % 8 == 0, "N/lda/ldb/ldo must be multiples of 8");
  GroupGemmParams p;
  p.A = (const uint16_t*)A;
  p.B = (const uint16_t*)B;
  p.out = (uint16_t*)out;
  p.sorted_ids = sorted_ids;
  p.expert_of_block = expert_of_block;
  p.seg_range = seg_range;
  p.order = order;
  p.meta = meta;
  p.row_scale = row_scale;
  p.flags = flags;
  p.ticket = ticket;
  p.err = err;
  p.lda = lda;
  p.ldb = ldb;
  p.ldo = ldo;
  p.b_expert_stride = b_expert_stride;
  p.N = N;
  p.K = K;
  p.topk = topk;
  p.a_mode = a_mode;
  p.cpr = cpr;
  p.epoch = epoch;
  static int num_cus = 0;
  if (!num_cus) {
    int dev = 0;
    TDIST_HIP_TRY(hipGetDevice(&dev));
    TDIST_HIP_TRY(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev));
  }
  // the whole grid must be resident: consumers spin on flags that the producers of this launch raise
  int grid = num_ctas > 0 ? num_ctas : num_cus;
  if (grid > num_cus) grid = num_cus;
  p.push_wgs = 0;
  p.a_local = nullptr;
  p.ws = nullptr;
  p.push_flags = nullptr;
  p.push_done = ticket + 8;
  p.lda_local = 0;
  p.M_per_rank = p.rows_per_chunk = 0;
  p.rank = p.world = 0;
  p.straddle = 1;
  p.race_debug = 0;
  p.trace = wg_trace_sink();
  size_t counter_ints = 8;
  if (ctx && push_wgs > 0 && ctx->num_ranks > 1) {
    TDIST_REQUIRE(a_local && ws && flags && M_per_rank > 0 && rows_per_chunk > 0 && M_per_rank % rows_per_chunk == 0,
                  "bad producer arguments");
    TDIST_REQUIRE(lda == K && lda_local % 8 == 0, "the in-kernel producer needs a dense [M, K] workspace");
    TDIST_REQUIRE((((uintptr_t)a_local | (uintptr_t)ws) & 15) == 0, "producer operands must be 16-byte aligned");
    p.ctx = *ctx;
    p.a_local = (const uint16_t*)a_local;
    p.lda_local = lda_local;
    p.ws = (uint16_t*)ws;
    p.push_flags = (int*)flags;
    p.M_per_rank = M_per_rank;
    p.rows_per_chunk = rows_per_chunk;
    p.rank = ctx->rank;
    p.world = ctx->num_ranks;
    p.push_wgs = push_wgs < 64 ? push_wgs : 64;
    if (p.push_wgs >= grid) p.push_wgs = grid > 1 ? grid - 1 : 1;
    counter_ints += (size_t)(M_per_rank / rows_per_chunk);
  } else {
    p.ctx.rank = 0;
    p.ctx.num_ranks = 1;
  }
  const int lds = LDS_BYTES + 16 + 2 * 3 * BM * 4;
  static bool attr_set = false;
  if (!attr_set) {
    TDIST_HIP_TRY(hipFuncSetAttribute((const void*)moe_group_gemm_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    TDIST_HIP_TRY(hipFuncSetAttribute((const void*)moe_group_gemm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    attr_set = true;
  }
  TDIST_REQUIRE(!state || (p.push_wgs > 0 && ws1 && flags1 && (((uintptr_t)ws1) & 15) == 0),
                "device-side call numbering is for the in-kernel all-gather and needs both parities");
  p.state = state;
  p.ws1 = (uint16_t*)ws1;
  p.flags1 = flags1;
  p.n_counters = (int)counter_ints;
  p.fin = ticket + counter_ints;
  if (dtype == 0)
    hipLaunchKernelGGL(moe_group_gemm_kernel<true>, dim3(grid), dim3(THREADS), lds, (hipStream_t)stream, p);
  else
    hipLaunchKernelGGL(moe_group_gemm_kernel<false>, dim3(grid), dim3(THREADS), lds, (hipStream_t)stream, p);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

TDIST_API int tdist_moe_topk_reduce_rs(const SymmCtx* ctx, const void* gemm_out, void* scatter, int* flags,
                                       int* counters, int* err, void* out, int ntokens, int topk, int N, long long ld,
                                       int epoch, int num_ctas, void* stream, int* state, void* scatter1, int* flags1) {
  const int w = ctx->num_ranks;
  TDIST_REQUIRE(ntokens % w == 0 && N % 8 == 0 && ld % 8 == 0 && ld >= N,
                "ntokens must divide by world size, N and the row stride by 8");
  TDIST_REQUIRE((((uintptr_t)gemm_out | (uintptr_t)scatter | (uintptr_t)out) & 15) == 0, "buffers must be 16-byte aligned");
  TopkRsParams p;
  p.ctx = *ctx;
  p.gemm_out = (const uint16_t*)gemm_out;
  p.scatter = (uint16_t*)scatter;
  p.flags = flags;
  p.counters = counters;
  p.err = err;
  p.out = (uint16_t*)out;
  p.ntokens = ntokens;
  p.topk = topk;
  p.N = N;
  p.ld = ld;
  p.tpr = ntokens / w;
  p.epoch = epoch;
  TDIST_REQUIRE(!state || (scatter1 && flags1 && (((uintptr_t)scatter1) & 15) == 0),
                "device-side call numbering needs both parities");
  p.state = state;
  p.scatter1 = (uint16_t*)scatter1;
  p.flags1 = flags1;
  // (`counters`: TDIST_MAX_RANKS + 4 ints, zero at the first launch; the kernel re-arms them)
  // memory-bound streaming work: four workgroups per CU unless the caller budgets fewer
  hipLaunchKernelGGL(moe_topk_reduce_rs_kernel, dim3(num_ctas > 0 ? num_ctas : 1024), dim3(256), 0, (hipStream_t)stream, p);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}

// ---- router epilogue: softmax over the experts, top-k, optional renormalisation ------------------------------------
// One wave per token.  logits bf16 [tokens, E] (row stride ld) -> ids int32 [tokens, k], weights fp32 [tokens, k],
// in descending order of probability.  Ties go to the lower expert index (the set torch.topk returns: everything
// above the k-th value, then the k-th value's duplicates in index order).  Replaces five library launches
// (upcast, softmax, top-k select, sort, sum, divide) of ~23 us per MoE layer at decode sizes with one of ~3 us.
template <int VPT>                              // values per lane: E <= 64 * VPT
__global__ __launch_bounds__(64) void moe_route_kernel(const uint16_t* __restrict__ logits, long long ld, int E, int k,
                                                       int renorm, int* __restrict__ ids, float* __restrict__ wts) {
  const int t = blockIdx.x, lane = threadIdx.x;
  const uint16_t* row = logits + t * ld;
  float v[VPT];
  float mx = -3.0e38f;
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    const int e = lane + 64 * j;
    v[j] = e < E ? __builtin_bit_cast(float, (uint32_t)row[e] << 16) : -3.0e38f;
    mx = fmaxf(mx, v[j]);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
  float den = 0.f;
#pragma unroll
  for (int j = 0; j < VPT; ++j) {
    float pj = expf(v[j] - mx);
    // a NaN logit (or +inf: inf - inf) must stay a candidate, or no lane would win a round and the sentinel index
    // would be written as an expert id; tested on the bits because the file is built with finite-math assumptions
    if ((__builtin_bit_cast(uint32_t, pj) & 0x7fffffffu) > 0x7f800000u) pj = 0.f;
    v[j] = lane + 64 * j < E ? pj : -1.f;                    // probabilities are >= 0: -1 marks "not a candidate"
    den += fmaxf(v[j], 0.f);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) den += __shfl_xor(den, o);
  const float inv = 1.f / den;
  int my_id = 0;
  float my_w = 0.f, sum_w = 0.f;
  for (int i = 0; i < k; ++i) {
    float best = -2.f;
    int bidx = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < VPT; ++j)
      if (v[j] > best) best = v[j], bidx = lane + 64 * j;          // ascending j: the lowest index of a lane's ties
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ob = __shfl_xor(best, o);
      const int oi = __shfl_xor(bidx, o);
      if (ob > best || (ob == best && oi < bidx)) best = ob, bidx = oi;
    }
    if ((bidx & 63) == lane) {
#pragma unroll
      for (int j = 0; j < VPT; ++j)
        if (j == (bidx >> 6)) v[j] = -1.f;
    }
    const float p = best * inv;
    sum_w += p;
    if (lane == i) my_id = min(bidx, E - 1), my_w = p;      // (in range whatever the inputs were)
  }
  if (lane < k) {
    ids[(long long)t * k + lane] = my_id;
    wts[(long long)t * k + lane] = renorm ? my_w / sum_w : my_w;
  }
}

TDIST_API int tdist_moe_route(const void* logits, long long ld, int tokens, int E, int k, int renorm, int* ids, float* wts,
                              void* stream) {
  TDIST_REQUIRE(tokens > 0 && E > 0 && E <= 1024 && k > 0 && k <= 64 && k <= E, "router: E <= 1024, k <= min(64, E)");
  TDIST_REQUIRE(logits && ids && wts && ld >= E, "router: null argument / bad row stride");
  auto launch = [&](auto kernel) {
    hipLaunchKernelGGL(kernel, dim3(tokens), dim3(64), 0, (hipStream_t)stream, (const uint16_t*)logits, ld, E, k, renorm,
                       ids, wts);
  };
  if (E <= 128) launch(moe_route_kernel<2>);
  else if (E <= 256) launch(moe_route_kernel<4>);
  else if (E <= 512) launch(moe_route_kernel<8>);
  else launch(moe_route_kernel<16>);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
