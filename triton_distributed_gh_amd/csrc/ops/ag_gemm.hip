// AllGather + GEMM, one persistent kernel per rank:  C[M, N_local] = AllGather_M(A[M/w, K]) * B[N_local, K]^T.
//
// The first `push_wgs` workgroups of the grid are PRODUCERS: they store this rank's shard of A into every
// peer's symmetric workspace [M, K] over xGMI, chunk by chunk, and raise one int32 flag per (source rank,
// chunk) to the call's epoch value on every peer; when they are done they join the consumers.  Tile jobs
// (256x256) live in EIGHT queues, one per XCD (tdist/work_queue.hpp: static first item, tickets drawn ahead
// of the main loop): a queue owns a band of column tiles (and, when there are fewer than
// eight column tiles, every qm-th row block), each in expected-arrival order (own rows first, then chunk
// 0 of every peer, chunk 1 of every peer, ...).  The workgroups of one XCD therefore work on a compact
// rows x columns patch whose A row blocks and B panels they share in that XCD's L2 - the mapping the plain
// GEMM gets from its block-id remap - instead of a random eighth of everything; a workgroup whose queue
// has run dry moves on to the next XCD's queue.  A tile job waits - workgroup-collectively, system scope,
// acquire - for the flags covering its rows and runs the MFMA main loop.  Producer and consumer share one launch because the GEMM workgroups fill the
// register file of every CU: a separate push kernel that reaches the GPU after the GEMM could
// not be scheduled until GEMM workgroups exit, and those are waiting for the peers' pushes.
// Producers never wait and the whole grid is resident at once (the host clamps it to the number of CUs),
// so progress does not depend on dispatch order.  (With push_wgs == 0 the data is expected from the SDMA
// copy engines instead.)
//
// When the problem has fewer tiles than the chip has CUs (large TP degree), K is cut into
// `split_k` ranges, one work item each: every item parks its fp32 accumulators in a scratch
// buffer (register layout, coalesced) and whichever item of a tile finishes last sums them and
// runs the epilogue - nobody waits.  This also divides the exposed tail after the last chunk
// lands by split_k.
//
// Differences from the reference's consumer
// (reference: python/triton_dist/kernels/amd/all_gather_gemm.py:127-245, 401-433):
//   * flags carry a per-call epoch and workspaces / flags are double-buffered by call parity, so
//     no flag reset and no cross-rank barrier is needed after the GEMM;
//   * tiles are handed out dynamically, so a workgroup stalled on a late chunk never holds back
//     tiles whose data already arrived;
//   * every spin is bounded and reports through an error word instead of hanging the GPU.
#include <stdio.h>
#include <stdlib.h>

#include "tdist/cache_policy.hpp"
#include "tdist/call_state.hpp"
#include "tdist/common.hpp"
#include "tdist/gemm_gfx950.hpp"
#include "tdist/primitives.hpp"
#include "tdist/race_debug.hpp"
#include "tdist/shmem_device.hpp"
#include "tdist/tile_plan.hpp"
#include "tdist/ag_push.hpp"
#include "tdist/wg_trace.hpp"
#include "tdist/work_queue.hpp"

using namespace tdist;
using namespace tdist::gemm;

struct AgGemmParams {
  SymmCtx ctx;               // only read when push_wgs > 0
  const uint16_t* a_local;   // [M_per_rank, K] own shard (never copied unless straddle)
  uint16_t* ws;              // [M, K] symmetric workspace of this call's parity
  const uint16_t* B;         // [N_local, K]
  uint16_t* C;               // [M, N_local]
  int* flags;                // [w * chunks_per_rank] of this call's parity (symmetric, signal arena)
  int* ticket;               // work counters: zero at launch (re-armed by the last workgroup of every launch):
                             // [0] push items, [4 .. 12) the eight tile queues
  int* err;                  // sticky error word (bit 0 = wait timeout)
  int* fin;                  // finished-workgroup counter of this launch
  int n_counters;            // ints from `ticket` on that the last workgroup zeroes
  int* push_done;            // [chunks_per_rank] (push_wgs > 0)
  int* tile_flags;           // [tiles] arrival counters (split_k > 1)
  float* partial;            // [tiles * split_k][32][512] f32x4 (split_k > 1)
  long long lda_local, ld_ws, ldb, ldc;
  int M_per_rank, N_local, K;
  int rank, world;
  int rows_per_chunk;
  int epoch;
  int straddle;              // 1: tiles walk the whole [M, K] workspace (own rows copied in too)
  int push_wgs;              // first push_wgs tickets push the local shard to every peer
  int split_k;               // K is cut into split_k ranges, one work item each
  int qn;                    // tile queues: qn column bands x (8 / qn) row-block residues
  int race_debug;            // tdist/race_debug.hpp
  int push_store;            // producer stores: 0 = write-through system scope, 1 = write-through agent scope, 2 = plain + release fence
  int acquire_mode;          // consumer's fence behind a flag: 0 = system-scope acquire, 1 = agent scope, 2 = none (measurement knob)
  WgTrace trace;             // tdist/wg_trace.hpp (off unless the host armed a sink)
  // Device-side call numbering (tdist/call_state.hpp).  When `state` is set, ws / flags / epoch above
  // are ignored by the host and resolved on the device from the call number: the launch is then the
  // same every time and can be replayed from a HIP graph.
  const float* scale_a;      // optional fp32 scale per gathered row [M] / per output column [N_local] (8-bit operands)
  const float* scale_b;
  int* state;
  uint16_t *ws0, *ws1;        // (separate fields selected with ?: - indexing an array member with the runtime
  int *flags0, *flags1;       //  parity would push the whole argument block into scratch memory)
};

// The per-call view of the double-buffered objects.  It is a separate small value on purpose: writing the
// selected pointers back into the (large, dynamically indexed) argument block would force a private
// copy of the whole block into scratch memory.
struct AgCall {
  uint16_t* ws;
  int* flags;
  int epoch;
};

// 16 bytes of LDS behind the tile staging area (TileShape::LDS): [0] ticket, [1] split-K arrival, [2] call number, [3] ticket

// SPLIT: false = every tile is one item (split_k == 1; no fix-up code in the kernel at all), true = split-K.
// DT: operand type, MT x NT: macro-tile of the tile family (gemm_gfx950.hpp; BM = 32 MT, BN = 64 NT) - chosen on the
// host by plan_ag() so that the consumers' share of the chip is filled in one round where possible.
template <int DT, int MT, int NT, bool SPLIT>
__global__ __launch_bounds__(THREADS, 2) void ag_gemm_consumer_kernel(AgGemmParams p) {
  using TS = TileShape<MT, NT>;
  using F = Frag<DT>;
  constexpr int BM = TS::BM, BN = TS::BN;
  constexpr int SLOT_BYTES = MT * NT * THREADS * 16;          // one parked fp32 tile
  extern __shared__ __attribute__((aligned(16))) char smem[];
  volatile int* s_ticket = (volatile int*)(smem + TS::LDS);
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
  const int S = SPLIT ? p.split_k : 1;
  const int ksteps = p.K / BK;
  const int qn = p.qn, qm = 8 / qn;
  const int cols_per = (num_pid_n + qn - 1) / qn;
  // queue q = (row-block residue q / qn, column band q % qn); items ordered row block, column, K range
  auto queue_cols = [=](int q) { return min(max(num_pid_n - (q % qn) * cols_per, 0), cols_per); };
  auto queue_size = [=](int q) {
    const int qmi = q / qn;
    const int n_o = total_mb > qmi ? (total_mb - qmi + qm - 1) / qm : 0;
    return n_o * queue_cols(q) * S;
  };
  WorkQueues<decltype(queue_size)> wq(p.ticket + 4, s_ticket, p.push_wgs, queue_size);
  int call = 0;
  AgCall cv{p.ws, p.flags, p.epoch};
  WgTraceCursor tr = wg_trace_begin(p.trace);
  wg_trace(tr, 1, 0);
  if (p.state) {
    call = call_begin(p.state, s_ticket + 2);
    cv.ws = call_parity(call) ? p.ws1 : p.ws0;
    cv.flags = call_parity(call) ? p.flags1 : p.flags0;
    cv.epoch = call_epoch(call);
  }
  race_debug_check_flags(p.race_debug, cv.flags, w * cpr, cv.epoch, p.err);

  int local = 0;
  bool have;
  if ((int)blockIdx.x < p.push_wgs) {
    // dedicated producers first push (they never wait, so progress does not depend on anybody else),
    // then help with whatever tiles are left
    push_item(p, cv, (int)blockIdx.x);
    wg_trace(tr, 6, (int)blockIdx.x);
    have = wq.next(local);
  } else {
    have = wq.first(local);
    if (!have) have = wq.next(local);
  }
  int seen_f0 = -1, seen_f1 = -1;       // flag range this workgroup has already acquired
  while (have) {
    const int q = wq.q;
    const int per_o = queue_cols(q) * S;
    const int o_idx = local / per_o, rem = local - o_idx * per_o;
    const int o = o_idx * qm + q / qn;
    const int pid_n = (q % qn) * cols_per + rem / S;
    const int ks = rem % S;                   // K range
    const int tile = o * num_pid_n + pid_n;
    wg_trace(tr, 2, tile);
    wq.prefetch();

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
      a_ptr = cv.ws + (long long)g0 * p.ld_ws;
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
        a_ptr = cv.ws + (long long)g0 * p.ld_ws;
        lda = p.ld_ws;
        need_wait = true;
      }
    }
    if (need_wait) {
      const int g1 = g0 + rows - 1;
      const int f0 = (g0 / p.M_per_rank) * cpr + (g0 % p.M_per_rank) / p.rows_per_chunk;
      const int f1 = (g1 / p.M_per_rank) * cpr + (g1 % p.M_per_rank) / p.rows_per_chunk;
      // Consecutive items of a queue mostly share their row block.  Rows this workgroup has waited for
      // before landed before that wait's acquire and do not change within the call: no second wait.
      if (f0 != seen_f0 || f1 != seen_f1) {
        int token;
        if (tr.p) {
          // traced launch: the three parts of the wait timed separately (flag load, acquire, barrier); the deltas
          // travel as event arguments (tags 8, 9, 10) so that no trace store sits between the parts
          const long long ta = (long long)wall_clock64();
          int bad = 0;
          if ((int)threadIdx.x < f1 - f0 + 1) {
            const int* fp = (const int*)cv.flags + f0 + (int)threadIdx.x;
            if (load_relaxed_system(fp) != cv.epoch && !spin_until_eq(fp, cv.epoch, SCOPE_SYS)) bad = 1;
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          const long long tb = (long long)wall_clock64();
          if (threadIdx.x < 64) acquire_fence(SCOPE_SYS);
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          const long long tc = (long long)wall_clock64();
          __syncthreads();
          const long long td = (long long)wall_clock64();
          token = bad;
          wg_trace(tr, 8, (int)(tb - ta));
          wg_trace(tr, 9, (int)(tc - tb));
          wg_trace(tr, 10, (int)(td - tc));
        } else if (p.acquire_mode == 0) {
          token = wait((const int*)cv.flags + f0, f1 - f0 + 1, SCOPE_SYS, SEM_ACQUIRE, cv.epoch, p.err);
        } else {
          token = wait((const int*)cv.flags + f0, f1 - f0 + 1, SCOPE_SYS, SEM_RELAXED, cv.epoch, p.err);
          if (p.acquire_mode == 1) {
            if (threadIdx.x < 64) {
              acquire_fence(SCOPE_GPU);
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();
          }
        }
        a_ptr = consume_token(a_ptr, token);
        if (token == 0) {
          seen_f0 = f0;
          seen_f1 = f1;
        }
        wg_trace(tr, 3, tile);
      }
    }

    const int k0 = (int)((long long)ksteps * ks / S) * BK;
    const int k1 = (int)((long long)ksteps * (ks + 1) / S) * BK;
    TileArgs t;
    t.A = a_ptr + k0;
    t.B = p.B + (long long)pid_n * BN * p.ldb + k0;
    t.lda = lda;
    t.ldb = p.ldb;
    t.a_rows = rows;
    t.b_rows = p.N_local - pid_n * BN;
    t.K = k1 - k0;
    AccT<MT, NT> acc;
    // accumulator columns permuted so the epilogue stores 16-byte pieces straight from registers:
    // no LDS staging, no barrier, and the next item's first loads go out right behind the stores
    mainloop_2ph<DT, MT, NT>(t, smem, acc);
    wg_trace(tr, 4, tile);
    // The next item is looked up HERE, before this tile is written out: the main loop ends with every load
    // drained and all waves in step, so collecting the ticket and the barrier that broadcasts it cost
    // nothing now - after the epilogue the same barrier would wait for the slowest wave to get its stores
    // into the memory pipeline (2-3 us when every CU of the chip writes its tile at the same moment).
    // With it out of the way each wave goes from its last store straight to the next tile's first loads.
    wq.stash();
    int next_local = 0;
    const bool have_next = wq.next(next_local);

    const int nrem = p.N_local - pid_n * BN;
    uint16_t* cbase = p.C + (long long)g0 * p.ldc + pid_n * BN;
    const long long ldc = p.ldc;
    auto store_c = [&](int r, int c, const uint4& v) {
      if (r < rows && c < nrem) *(uint4*)(cbase + (long long)r * ldc + c) = v;
    };
    if constexpr (!SPLIT) {
      epilogue_direct<DT, MT, NT>(acc, DT >= DT_FP8 ? p.scale_a : nullptr, DT >= DT_FP8 ? p.scale_b : nullptr, g0, pid_n * BN, rows, nrem, store_c);
      wg_trace(tr, 5, tile);
    } else {
    // Split-K fix-up, wait-free: every item parks its fp32 accumulators (register layout, so both
    // directions are fully coalesced) and counts itself; whichever item of the tile counts last adds the
    // other partials to its own accumulators and stores the tile.  The per-thread offset is made opaque so
    // it is not hoisted out of the persistent loop (it would stay live across the main loop).
    unsigned voff = threadIdx.x * 16u;
    asm volatile("" : "+v"(voff));
    char* tbase = (char*)p.partial + (long long)tile * S * SLOT_BYTES;
    {
      char* mine = tbase + (long long)ks * SLOT_BYTES;
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) st_wt128_agent(mine + (voff + (unsigned)((m * NT + n) * THREADS * 16)), acc.v[m][n]);
    }
    // write-through stores: drain, barrier, count - no release fence
    drain_stores();
    __syncthreads();
    if (threadIdx.x == 0) s_ticket[1] = atom_add_relaxed_agent(p.tile_flags + tile, 1);
    __syncthreads();
    const int arrived = __builtin_amdgcn_readfirstlane(s_ticket[1]);
    if (arrived == S - 1) {
      // last arriver: one acquire drops this CU's stale L1 lines, then plain loads
      if (threadIdx.x < 64) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
      // This item's own partial is still in its accumulator registers: add the other split_k - 1 partials
      // to it (loads only, so they pipeline freely) and store the tile straight from registers.
      for (int qq = 0; qq < S; ++qq) {
        if (qq == ks) continue;
        const char* src = tbase + (long long)qq * SLOT_BYTES;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          f32x4 x[NT];
#pragma unroll
          for (int n = 0; n < NT; ++n) x[n] = *(const f32x4*)(src + (voff + (unsigned)((m * NT + n) * THREADS * 16)));
#pragma unroll
          for (int n = 0; n < NT; ++n) acc.v[m][n] = F::add(acc.v[m][n], x[n]);
        }
      }
      epilogue_direct<DT, MT, NT>(acc, DT >= DT_FP8 ? p.scale_a : nullptr, DT >= DT_FP8 ? p.scale_b : nullptr, g0, pid_n * BN, rows, nrem, store_c);
    }
    wg_trace(tr, 5, tile);
    }
    have = have_next;
    local = next_local;
  }
  wg_trace(tr, 7, 0);
  call_end(p.state, call, p.fin, p.ticket, p.n_counters);
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
  // write-through stores (payload leaves the L2 at once): publishing is drain + barrier + counter;
  // a release fence is only needed if an unaligned shard forced plain stores
  bool plain = false;
  if (p.include_self) {
    char* dst = (char*)p.ws + dst_off;
    plain |= shmem::copy_bytes_wt_sys(dst, src, (size_t)(b1 - b0), threadIdx.x, blockDim.x);
  }
  for (int i = 0; i < w - 1; ++i) {
    const int peer = (rank + 1 + (i + blockIdx.x) % (w - 1)) % w;
    char* dst = (char*)symm_at(p.ctx, p.ws, peer) + dst_off;
    plain |= shmem::copy_bytes_wt_sys(dst, src, (size_t)(b1 - b0), threadIdx.x, blockDim.x);
  }
  if (plain) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  drain_stores();
  __syncthreads();
  if (threadIdx.x == 0) {
    const int prev = atom_add_relaxed_agent(p.done_count + chunk, 1);
    if (prev == p.slices - 1) {
      const int cpr = p.M_per_rank / p.rows_per_chunk;
      for (int i = p.include_self ? 0 : 1; i < w; ++i) {
        const int peer = (rank + i) % w;
        store_relaxed_system(symm_at(p.ctx, p.flags, peer) + rank * cpr + chunk, p.epoch);
      }
    }
  }
}

static int check_common(const void* B, const void* C, int N_local, int K16, long long ldb16, long long ldc) {
  TDIST_REQUIRE(K16 >= 128 && K16 % 64 == 0, "K must be a multiple of 64 (16-bit) / 128 (8-bit) and >= two such steps");
  TDIST_REQUIRE(N_local % 8 == 0 && ldb16 % 8 == 0 && ldc % 8 == 0, "N_local/ldb/ldc must keep 16-byte row alignment");
  TDIST_REQUIRE((((uintptr_t)B | (uintptr_t)C) & 15) == 0, "B and C must be 16-byte aligned");
  return 0;
}

// Tile shape and K split of the consumer for one call.  `consumers`: workgroups that compute from the start (the
// grid minus the dedicated producers).  The cost model is the plain GEMM's (tdist/tile_plan.hpp) over the rank's
// row blocks: a macro-tile never spans two ranks' rows unless M_per_rank is not a multiple of BM (straddle mode,
// tiles walk the whole gathered operand).  A split is only taken while all items fit one round of consumers.
struct AgPlan {
  int mt, nt, split_k, straddle, tiles;
};
static AgPlan plan_ag(int M_per_rank, int world, int N_local, int K16, int consumers, int bm, int bn, int split_k) {
  const int cand[3][2] = {{8, 4}, {8, 2}, {4, 4}};
  const int ksteps = K16 / 64;
  double best = 1e30;
  AgPlan pick{8, 4, 1, 0, 0};
  bool any = false;
  for (int ci = 0; ci < 3; ++ci) {
    const int mt = cand[ci][0], nt = cand[ci][1];
    if ((bm && bm != 32 * mt) || (bn && bn != 64 * nt)) continue;
    const int tbm = 32 * mt, tbn = 64 * nt;
    const int straddle = (world > 1 && M_per_rank % tbm != 0) ? 1 : 0;
    const int mb = straddle ? (M_per_rank * world + tbm - 1) / tbm : ((M_per_rank + tbm - 1) / tbm) * world;
    const long long tiles = (long long)mb * ((N_local + tbn - 1) / tbn);
    for (int s = 1; s <= 4; ++s) {
      if (split_k && s != split_k) continue;
      if (s > 1 && (ksteps < 2 * s || (!split_k && (ksteps / s < 8 || tiles * s > consumers)))) continue;
      const long long items = tiles * s;
      const long long rounds = (items + consumers - 1) / consumers;
      const double tile_kb = tbm * tbn * 4 / 1024.0;
      // (parking and re-reading fp32 tiles while every CU does the same: 19 - 25 us measured for 256 x 256 tiles cut in
      // two, twice what the plain kernel's model charges)
      const double fix = s > 1 ? 2500.0 + 40.0 * tile_kb + (s - 1) * 60.0 * tile_kb : 0.0;
      // a straddling walk costs the own rows a pass through the workspace; prefer aligned shapes on a near tie
      const double total = rounds * ((ksteps / s) * gemm::tile_step_ns(mt, nt, items > consumers / 2) + 7000.0) + fix +
                           (straddle ? 3000.0 : 0.0);
      if (total < best) {
        best = total;
        pick = AgPlan{mt, nt, s, straddle, (int)tiles};
        any = true;
      }
    }
  }
  if (!any) pick = AgPlan{8, 4, 1, (world > 1 && M_per_rank % 256 != 0) ? 1 : 0, 0};
  return pick;
}

// What tdist_ag_gemm_consumer will do for this call (K in ELEMENTS of the operand type; dtype as tdist_gemm_nt_ex;
// bm / bn / split_k: 0 = free): tile, split, whether tiles straddle ranks, number of tiles, bytes of fp32 partial
// tiles the split needs, and the ints of scratch ([16 + chunks_per_rank + tiles], see below).
TDIST_API int tdist_ag_gemm_plan(int M_per_rank, int world, int N_local, int K, int dtype, int consumers, int bm, int bn,
                                 int split_k, int* out_bm, int* out_bn, int* out_split_k, int* out_straddle,
                                 int* out_tiles, long long* partial_bytes) {
  TDIST_REQUIRE(dtype >= 0 && dtype <= 4 && M_per_rank > 0 && world >= 1 && N_local > 0 && K > 0 && consumers > 0, "bad problem");
  const int per16 = dtype >= 2 ? 2 : 1;
  const AgPlan pl = plan_ag(M_per_rank, world, N_local, K / per16, consumers, bm, bn, split_k);
  const int tbm = 32 * pl.mt, tbn = 64 * pl.nt;
  const int mb = pl.straddle ? (M_per_rank * world + tbm - 1) / tbm : ((M_per_rank + tbm - 1) / tbm) * world;
  *out_bm = tbm;
  *out_bn = tbn;
  *out_split_k = pl.split_k;
  *out_straddle = pl.straddle;
  *out_tiles = mb * ((N_local + tbn - 1) / tbn);
  *partial_bytes = pl.split_k > 1 ? (long long)(*out_tiles) * pl.split_k * tbm * tbn * 4 : 0;
  return 0;
}

typedef void (*ag_kern_t)(AgGemmParams);
template <int DT>
static ag_kern_t ag_kernel_of(int shape, bool split) {
  switch (shape * 2 + (split ? 1 : 0)) {
    case 0: return ag_gemm_consumer_kernel<DT, 8, 4, false>;
    case 1: return ag_gemm_consumer_kernel<DT, 8, 4, true>;
    case 2: return ag_gemm_consumer_kernel<DT, 8, 2, false>;
    case 3: return ag_gemm_consumer_kernel<DT, 8, 2, true>;
    case 4: return ag_gemm_consumer_kernel<DT, 4, 4, false>;
    default: return ag_gemm_consumer_kernel<DT, 4, 4, true>;
  }
}

// scratch layout (int32): [0] error word (sticky, never cleared here), [1] finished-workgroup counter,
// [4] push ticket, [8, 16) tile-queue tickets, [16, 16 + cpr) push counters, [16 + cpr, 16 + cpr + tiles)
// split-K arrival counters.  Everything from [1] on must be zero at the first launch; every launch leaves
// it zero again (call_state.hpp).
// `state` != nullptr: device-side call numbering; ws / flags are then parity 0 and ws1 / flags1 parity 1,
// and `epoch` is ignored.
// dtype: 0 = bf16, 1 = fp16, 2 = fp8 e4m3, 3 = fp8 e5m2, 4 = int8 (8-bit operands: bf16 output, optional fp32
// scale_a [M] per gathered row and scale_b [N_local] per column).  K, lda_local, ld_ws, ldb in ELEMENTS of the
// operand type.  bm / bn / split_k / straddle: as returned by tdist_ag_gemm_plan for this call.
TDIST_API int tdist_ag_gemm_consumer(const tdist::SymmCtx* ctx, const void* a_local, void* ws, const void* B, void* C,
                                     int* flags, int* scratch, long long scratch_ints, void* partial,
                                     long long partial_bytes, long long lda_local, long long ld_ws, long long ldb,
                                     long long ldc, int M_per_rank, int N_local, int K, int rows_per_chunk, int epoch,
                                     int straddle, int push_wgs, int split_k, int num_ctas, int dtype, void* stream,
                                     int* state, void* ws1, int* flags1, int bm, int bn, const float* scale_a,
                                     const float* scale_b) {
  TDIST_REQUIRE(dtype >= 0 && dtype <= 4, "dtype must be 0..4");
  const int per16 = dtype >= 2 ? 2 : 1;
  TDIST_REQUIRE(K % per16 == 0 && lda_local % per16 == 0 && ld_ws % per16 == 0 && ldb % per16 == 0, "odd 8-bit strides");
  K /= per16;                 // from here on in 16-bit units
  lda_local /= per16;
  ld_ws /= per16;
  ldb /= per16;
  if (check_common(B, C, N_local, K, ldb, ldc)) return -1;
  TDIST_REQUIRE(!state || (ws1 && flags1 && (((uintptr_t)ws1) & 15) == 0), "device-side call numbering needs both parities");
  const int world = ctx->num_ranks, rank = ctx->rank;
  TDIST_REQUIRE(world >= 1 && world <= TDIST_MAX_RANKS && rank >= 0 && rank < world, "bad rank/world");
  TDIST_REQUIRE(M_per_rank > 0 && rows_per_chunk > 0 && M_per_rank % rows_per_chunk == 0,
                "M_per_rank must be a positive multiple of rows_per_chunk");
  TDIST_REQUIRE(lda_local % 8 == 0 && ld_ws % 8 == 0, "lda must keep 16-byte row alignment");
  TDIST_REQUIRE((((uintptr_t)a_local | (uintptr_t)ws) & 15) == 0, "A operands must be 16-byte aligned");
  TDIST_REQUIRE(push_wgs >= 0 && push_wgs <= 128, "push_wgs must be in [0, 128]");
  TDIST_REQUIRE(push_wgs == 0 || ld_ws == K, "the in-kernel producer needs a dense [M, K] workspace");
  if (bm == 0) bm = 256;
  if (bn == 0) bn = 256;
  TDIST_REQUIRE((bm == 256 && (bn == 256 || bn == 128)) || (bm == 128 && bn == 256), "tile must be 256x256, 256x128 or 128x256");
  TDIST_REQUIRE(split_k >= 1 && split_k <= 4 && K / BK >= 2 * split_k, "split_k must be in [1, 4] with >= 128 of K each");
  TDIST_REQUIRE(straddle || world == 1 || M_per_rank % bm == 0, "tiles that cross ranks need straddle mode");
  const int M = M_per_rank * world;
  const int cpr = M_per_rank / rows_per_chunk;
  const int total_mb = straddle ? (M + bm - 1) / bm : ((M_per_rank + bm - 1) / bm) * world;
  const int tiles = total_mb * ((N_local + bn - 1) / bn);
  const long long need_ints = 16 + cpr + (split_k > 1 ? tiles : 0);
  TDIST_REQUIRE(scratch_ints >= ((need_ints + 3) & ~3ll), "scratch too small");
  TDIST_REQUIRE(split_k == 1 || (partial && partial_bytes >= (long long)tiles * split_k * bm * bn * 4),
                "split-K partial buffer too small");
  AgGemmParams p;
  p.ctx = *ctx;
  p.a_local = (const uint16_t*)a_local;
  p.ws = (uint16_t*)ws;
  p.B = (const uint16_t*)B;
  p.C = (uint16_t*)C;
  p.flags = flags;
  p.err = scratch;
  p.fin = scratch + 1;
  p.n_counters = (int)((need_ints - 4 + 3) & ~3ll);
  p.ticket = scratch + 4;
  p.push_done = scratch + 16;
  p.tile_flags = scratch + 16 + cpr;
  p.partial = (float*)partial;
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
  p.push_wgs = world > 1 ? push_wgs : 0;
  p.split_k = split_k;
  {
    // column bands first (B panels are the larger operand to share), row-block residues for the rest.  (Choosing the
    // split that moves the fewest bytes per XCD - rows + columns - was measured at 4096 x 1024 x 8192: one band of
    // four column tiles per XCD ran the main loops 10 % slower than four bands.)
    const int npn = (N_local + bn - 1) / bn;
    int qn = 1;
    while (qn < 8 && qn * 2 <= npn) qn *= 2;
    static const int forced = getenv("TDIST_AG_QN") ? atoi(getenv("TDIST_AG_QN")) : 0;      // tuning knob
    if ((forced == 1 || forced == 2 || forced == 4 || forced == 8) && forced <= (npn > 1 ? npn : 1)) qn = forced;
    p.qn = qn;
  }
  p.race_debug = race_debug_enabled();
  {
    static const int ps = getenv("TDIST_PUSH_STORE") ? atoi(getenv("TDIST_PUSH_STORE")) : 0;
    p.push_store = ps;
    static const int am = getenv("TDIST_AG_ACQUIRE") ? atoi(getenv("TDIST_AG_ACQUIRE")) : 0;
    p.acquire_mode = am;
  }
  p.trace = wg_trace_sink();
  p.scale_a = scale_a;
  p.scale_b = scale_b;
  p.state = state;
  p.ws0 = (uint16_t*)ws;
  p.ws1 = (uint16_t*)ws1;
  p.flags0 = flags;
  p.flags1 = flags1;
  const int total = p.push_wgs + tiles * split_k;
  // The whole grid must be resident at once: consumers spin on flags that this rank's producers (and the
  // peers') raise, and a producer that cannot be scheduled behind spinning consumers never would.
  static int num_cus = 0;
  if (!num_cus) {
    int dev = 0;
    TDIST_HIP_TRY(hipGetDevice(&dev));
    TDIST_HIP_TRY(hipDeviceGetAttribute(&num_cus, hipDeviceAttributeMultiprocessorCount, dev));
  }
  int grid = num_ctas > 0 ? num_ctas : num_cus;
  if (grid > num_cus) grid = num_cus;
  if (grid > total) grid = total;
  if (p.push_wgs > grid) p.push_wgs = grid;     // (they all push first and compute afterwards)
  static const bool trace = getenv("TDIST_TRACE_LAUNCH") != nullptr;
  if (trace)
    fprintf(stderr,
            "[tdist] ag_gemm rank %d/%d a=%p ws0=%p ws1=%p B=%p C=%p flags0=%p flags1=%p scratch=%p(%lld ints) "
            "partial=%p(%lld B) state=%p Mr=%d N=%d K16=%d rpc=%d straddle=%d push=%d splitk=%d tile=%dx%d grid=%d total=%d\n",
            rank, world, a_local, ws, ws1, B, C, (void*)flags, (void*)flags1, (void*)scratch, scratch_ints, partial,
            partial_bytes, (void*)state, M_per_rank, N_local, K, rows_per_chunk, straddle, p.push_wgs, split_k, bm, bn,
            grid, total);
  const int shape = bm == 256 ? (bn == 256 ? 0 : 1) : 2;
  const bool split = split_k > 1;
  ag_kern_t kern;
  switch (dtype) {
    case 0: kern = ag_kernel_of<DT_BF16>(shape, split); break;
    case 1: kern = ag_kernel_of<DT_F16>(shape, split); break;
    case 2: kern = ag_kernel_of<DT_FP8>(shape, split); break;
    case 3: kern = ag_kernel_of<DT_BF8>(shape, split); break;
    default: kern = ag_kernel_of<DT_I8>(shape, split); break;
  }
  static const int lds_of[3] = {TileShape<8, 4>::LDS + 16, TileShape<8, 2>::LDS + 16, TileShape<4, 4>::LDS + 16};
  // (per kernel, once: the attribute call is cheap but not free)
  static bool attr_set[5][3][2] = {};
  if (!attr_set[dtype][shape][split]) {
    TDIST_HIP_TRY(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, lds_of[shape]));
    attr_set[dtype][shape][split] = true;
  }
  hipLaunchKernelGGL(kern, dim3(grid), dim3(THREADS), lds_of[shape], (hipStream_t)stream, p);
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
