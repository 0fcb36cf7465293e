// Variable-length causal GQA flash-attention forward for gfx950 and the sequence-parallel
// KV all-gather that feeds it.
//
// Role in the reference: kernels/nvidia/sp_ag_attention_intra_node.py - KV all-gather producer
// (:105-183) and kernel_consumer_flash_attn_forward (:255-425); the same kernel serves the
// prefill attention of the TP attention layer (layers/amd/tp_attn.py:275-308 uses flash_attn).
//
// Attention kernel: one workgroup (4 waves) per 128 (or, for small grids, 64) query rows of one
// (sequence, head); each wave owns 32 rows = two 16-row MFMA column blocks (or one).  K/V blocks of 32
// tokens are loaded once per workgroup straight from global memory into a four-stage LDS ring
// (direct-to-LDS loads, XOR-swizzled 16-byte chunks, three blocks in flight, one raw barrier per
// block); workgroups are mapped to (head, query block) so that each XCD serves a contiguous group
// of heads and all XCDs carry the same causal load.  As in the decode kernel the scores
// are produced transposed (S^T = K.Q^T) so the P operand of O^T = V^T.P^T falls out of the
// accumulators without cross-lane traffic and the V operand is an in-register transpose of
// eight 16-byte LDS row reads.
//
// Sequence parallelism: the query rows of a rank map to global positions either contiguously
// or zig-zag (rank r owns half-blocks r and 2w-1-r), and the kernel can wait on per-source-rank
// arrival flags right before it first touches a rank's slab of the gathered KV, so attention
// over early slabs overlaps the rest of the gather (the reference joins the streams instead).
#include "tdist/cache_policy.hpp"
#include "tdist/common.hpp"
#include "tdist/atomics.hpp"
#include "tdist/gemm_gfx950.hpp"
#include "tdist/primitives.hpp"
#include "tdist/shmem_device.hpp"

namespace {

using namespace tdist;
using tdist::gemm::f32x4;
using tdist::gemm::Frag;

struct AttnArgs {
  const uint16_t *q, *k, *v;
  uint16_t* o;
  float* lse;              // optional [q_heads, total_q]
  const int *cu_q, *cu_k;  // [batch + 1]
  long long q_stride_t, q_stride_h, k_stride_t, k_stride_h, v_stride_t, v_stride_h, o_stride_t, o_stride_h;
  long long lse_stride_h;
  int q_heads, kv_heads;
  int causal;
  int pos_mode;            // 0: q rows are the LAST Lq positions of the sequence; 1: contiguous shard of
                           // rank `rank`; 2: zig-zag shard of rank `rank`
  int rank, world;
  float sm_scale;
  const int* flags;        // optional [world] KV arrival flags (value == epoch when slab landed)
  int epoch;
  int* err;
};

__device__ __forceinline__ int q_position(const AttnArgs& a, int i, int Lq, int Lk) {
  if (a.pos_mode == 0) return i + (Lk - Lq);
  if (a.pos_mode == 1) return a.rank * Lq + i;
  const int half = Lq >> 1;
  return i < half ? a.rank * half + i : (2 * a.world - 1 - a.rank) * half + (i - half);
}

__device__ __forceinline__ int kv_source_rank(const AttnArgs& a, int n, int Lk) {
  if (a.pos_mode == 2) {
    const int half = max(Lk / (2 * a.world), 1);
    const int blk = min(n / half, 2 * a.world - 1);
    return blk < a.world ? blk : 2 * a.world - 1 - blk;
  }
  const int per = max(Lk / a.world, 1);
  return min(n / per, a.world - 1);
}

// K/V ring depth and LDS footprint (the ring is reused for the 4 x 32 padded output rows at the end)
constexpr int attn_stages(int d, int rb) { return d <= 128 ? (rb == 1 ? 3 : 4) : 2; }
constexpr int attn_lds_bytes(int d, int rb) {
  const int ring = attn_stages(d, rb) * 2 * 32 * d * 2, out = 4 * 32 * (d * 2 + 16);
  return ring > out ? ring : out;
}

// RB = 16-row query blocks per wave: a workgroup covers 64 * RB query rows.  RB = 1 halves the work of
// the heaviest causal workgroup and doubles the number of workgroups - chosen by the launcher when
// the grid of 128-row blocks would leave wave slots empty (few heads per rank, short sequences).
template <bool IS_BF16, int D, int RB>
__global__ __launch_bounds__(256, 2) void flash_attn_fwd_kernel(AttnArgs a) {
  constexpr int QROWS = 64 * RB;
  using F = Frag<IS_BF16>;
  using VT = typename F::T;
  constexpr int KS = D / 32;
  constexpr int NDV = D / 16;
  constexpr int VW = NDV / 2;
  constexpr int ROWB = D * 2;                // LDS row of a K/V tile (bytes), unpadded: chunks are XOR-swizzled instead
  constexpr int TILE_B = 32 * ROWB;
  constexpr int CPR = D / 8;                 // 16-byte chunks per row
  constexpr int IPW = CPR / 8;               // direct-to-LDS loads (64 lanes x 16 bytes) per wave and tile
  constexpr int LPS = 2 * IPW;               // ... per wave and stage (K and V)
  constexpr int NST = attn_stages(D, RB);        // stages of the K/V ring; NST - 1 blocks are in flight
  constexpr int ROW_B = D * 2 + 16;          // padded row of the epilogue's output staging
  constexpr int OUT_TILE_B = 32 * ROW_B;
  constexpr float LOG2E = 1.4426950408889634f;

  extern __shared__ char smem[];             // [NST stages][K tile, V tile]; reused for the output staging

  // Workgroups go round-robin over the 8 XCDs in launch order (x fastest).  With the plain mapping an
  // XCD would always get the same query blocks of every head (causal: 52 to 80 KV-block units of
  // work against a mean of 66) and all KV heads.  Remapped: XCD k serves the contiguous head group
  // k (its KV heads stay in that XCD's L2) and every group hands out its query blocks in the same order.
  int qb = blockIdx.x, h = blockIdx.y;
  if ((a.q_heads & 7) == 0) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y, hp = a.q_heads >> 3;
    const int j = lin >> 3;
    h = (lin & 7) * hp + j % hp;
    qb = j / hp;
  } else {
    // head counts that do not divide over the XCDs: heads fastest, so the equally heavy workgroups of
    // one query block are dealt round the XCDs (balance first; KV heads are then shared between L2s)
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    h = lin % a.q_heads;
    qb = lin / a.q_heads;
  }
  const int b = blockIdx.z;
  const int q0 = a.cu_q[b], Lq = a.cu_q[b + 1] - q0;
  const int k0 = a.cu_k[b], Lk = a.cu_k[b + 1] - k0;
  // causal: the last query blocks see the longest KV range - hand them out first so the short ones fill the tail
  const int m0 = (a.causal ? (int)gridDim.x - 1 - qb : qb) * QROWS;
  if (m0 >= Lq) return;
  const int kvh = h / (a.q_heads / a.kv_heads);

  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lg = lane >> 4;
  const int row_base = m0 + wave * 16 * RB;

  // ---- Q operand (column block rb, head-dim step s): lane (row li, k-group lg)
  VT qreg[RB][KS];
  int qpos[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int r = row_base + 16 * rb + li;
    qpos[rb] = r < Lq ? q_position(a, r, Lq, Lk) : -1;
    const uint16_t* qp = a.q + (long long)(q0 + min(r, Lq - 1)) * a.q_stride_t + (long long)h * a.q_stride_h + lg * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      uint4 v = uint4{0, 0, 0, 0};
      if (r < Lq) v = *(const uint4*)(qp + 32 * s);
      qreg[rb][s] = __builtin_bit_cast(VT, v);
    }
  }
  const int last_row = min(m0 + QROWS - 1, Lq - 1);
  const int kv_hi = a.causal ? min(Lk, q_position(a, last_row, Lq, Lk) + 1) : Lk;
  const int wave_last = min(row_base + 16 * RB - 1, Lq - 1);
  const int wave_hi = row_base >= Lq ? 0 : (a.causal ? min(Lk, q_position(a, wave_last, Lq, Lk) + 1) : Lk);
  const int nblk = (kv_hi + 31) / 32;

  float m_run[RB], l_run[RB];
  f32x4 acc[RB][NDV];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) m_run[rb] = -INFINITY, l_run[rb] = 0.f;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < NDV; ++j) acc[rb][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const float scale2 = a.sm_scale * LOG2E;
  const uint16_t* kg = a.k + (long long)k0 * a.k_stride_t + (long long)kvh * a.k_stride_h;
  const uint16_t* vg = a.v + (long long)k0 * a.v_stride_t + (long long)kvh * a.v_stride_h;
  unsigned seen = 0;   // bit r: rank r's KV slab is known to have landed
  // last KV position each of this lane's query rows may see, and the smallest one in the wave: KV blocks
  // that end at or below it need no masking at all
  int lim[RB], lim_min = 0x7fffffff;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    lim[rb] = a.causal ? min(qpos[rb], Lk - 1) : (qpos[rb] < 0 ? -1 : Lk - 1);
    lim_min = min(lim_min, lim[rb]);
  }
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) lim_min = min(lim_min, __shfl_xor(lim_min, off));
  lim_min = __builtin_amdgcn_readfirstlane(lim_min);

  // K/V tiles go from global memory straight into LDS (no registers, no second hop), NST - 1 blocks
  // ahead.  A load instruction fills 1024 consecutive LDS bytes, lane i the 16 bytes at 16 * i, so the
  // bank-conflict padding of a register-staged tile is not available; instead the lane picks WHICH
  // source chunk it fetches: tile row r keeps chunk c at chunk position c ^ swz(r), and the 16 rows a
  // fragment read touches land on 16 distinct bank quads.
  auto swz = [](int r) { return D == 64 ? ((r >> 1) & 7) : (r & 15); };
  // The V tile is read with the transposing LDS load (4 tile rows x 16 columns per 16-lane group, column-
  // major into the lanes), which hands the MFMA its V^T operand without any in-register shuffling.
  // Its image: tile row v = 8 lg + 4 tb + i holds the token that the P operand has at k-group lg,
  // element 4 tb + i (token 16 tb + 4 lg + i of the block); chunk ch of row v sits at byte
  //   VSUB (v >> 3) + 512 (ch >> 2) + 64 (v & 7) + 16 ((ch & 3) ^ ((v >> 2) & 3)),
  // so the two 16-lane groups of a half wave read disjoint bank ranges and all column-block offsets
  // are instruction immediates.
  constexpr int VSUB = 128 * CPR;
  int ld_row[IPW], ld_col[IPW], ldv_tok[IPW], ldv_col[IPW];
#pragma unroll
  for (int i = 0; i < IPW; ++i) {
    const int pos = (wave * IPW + i) * 64 + lane;
    ld_row[i] = pos / CPR;
    ld_col[i] = (((pos % CPR) ^ swz(ld_row[i])) & (CPR - 1)) * 8;
    const int off = pos * 16;
    const int vhi = off / VSUB, rem = off % VSUB;
    const int v = 8 * vhi + (rem % 512) / 64;
    const int ch = 4 * (rem / 512) + (((rem % 64) / 16) ^ ((v >> 2) & 3));
    ldv_tok[i] = 16 * ((v >> 2) & 1) + 4 * (v >> 3) + (v & 3);
    ldv_col[i] = ch * 8;
  }
  auto issue = [&](int blk) {
    const int n0 = blk * 32;
    if (a.flags) {
      const int r_lo = kv_source_rank(a, n0, Lk), r_hi = kv_source_rank(a, min(n0 + 31, Lk - 1), Lk);
      for (int r = r_lo;; r = r_hi) {
        if (!((seen >> r) & 1u)) {
          (void)wait(a.flags + r, 1, SCOPE_SYS, SEM_ACQUIRE, a.epoch, a.err);
          seen |= 1u << r;
        }
        if (r == r_hi) break;
      }
    }
    char* kt = smem + (blk % NST) * 2 * TILE_B + wave * IPW * 1024;
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
      const int tok = min(n0 + ld_row[i], Lk - 1), tokv = min(n0 + ldv_tok[i], Lk - 1);
      __builtin_amdgcn_global_load_lds(TDIST_GLB_PTR(kg + (long long)tok * a.k_stride_t + ld_col[i]),
                                       TDIST_LDS_PTR(kt + i * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds(TDIST_GLB_PTR(vg + (long long)tokv * a.v_stride_t + ldv_col[i]),
                                       TDIST_LDS_PTR(kt + TILE_B + i * 1024), 16, 0, 0);
    }
  };
  // fragment read offsets of this lane inside a tile
  int koff[KS], vbase_off[2][2];   // V: [token half tb][parity of the 16-column block j]
#pragma unroll
  for (int s2 = 0; s2 < KS; ++s2) koff[s2] = li * ROWB + (((4 * s2 + lg) ^ swz(li)) << 4);
#pragma unroll
  for (int tb = 0; tb < 2; ++tb)
#pragma unroll
    for (int jp = 0; jp < 2; ++jp) {
      const int q = li >> 2, pp = li & 3;   // lane 4 q + p of a group addresses row q, columns 4 p .. 4 p + 3
      vbase_off[tb][jp] = VSUB * lg + 64 * (4 * tb + q) + 16 * ((2 * jp + (pp >> 1)) ^ ((2 * lg + tb) & 3)) + 8 * (pp & 1);
    }

#pragma unroll
  for (int s2 = 0; s2 < NST - 1; ++s2)
    if (s2 < nblk) issue(s2);

  for (int blk = 0; blk < nblk; ++blk) {
    // this wave's part of block blk has landed once at most the loads of the younger blocks are pending;
    // after the barrier everybody's part has, and nobody reads the stage that the next issue refills
    const int ahead = min(NST - 2, nblk - 1 - blk);
    if (NST >= 4 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(2 * LPS) : "memory");
    else if (NST >= 3 && ahead == 1) asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(LPS) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    if (blk + NST - 1 < nblk) issue(blk + NST - 1);
    const int n0 = blk * 32;
    const char* kt = smem + (blk % NST) * 2 * TILE_B;
    const char* vt = kt + TILE_B;
    if (n0 < wave_hi) {
      VT kfr[2][KS];
#pragma unroll
      for (int tb = 0; tb < 2; ++tb)
#pragma unroll
        for (int s = 0; s < KS; ++s)
          kfr[tb][s] = __builtin_bit_cast(VT, *(const uint4*)(kt + 16 * tb * ROWB + koff[s]));
      typedef short tr4 __attribute__((ext_vector_type(4)));
      uint2 vtr[NDV][2];
#pragma unroll
      for (int j = 0; j < NDV; ++j)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
          vtr[j][tb] = __builtin_bit_cast(uint2, __builtin_amdgcn_ds_read_tr16_b64_v4i16(
              (__attribute__((address_space(3))) tr4*)(vt + vbase_off[tb][j & 1] + 512 * (j >> 1))));
      VT pfrag[RB];
      float alpha[RB];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        f32x4 st[2];
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
          st[tb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int s = 0; s < KS; ++s) st[tb] = F::mfma(kfr[tb][s], qreg[rb][s], st[tb]);
        }
        float sc[8];
        float mx = -INFINITY;
        // raw scores: the scale (> 0) is applied to the row maximum and inside the exponent's fused multiply-add
        if (n0 + 31 <= lim_min) {
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            sc[t] = st[t >> 2][t & 3];
            mx = fmaxf(mx, sc[t]);
          }
        } else {
#pragma unroll
          for (int t = 0; t < 8; ++t) {
            const int tok = n0 + 16 * (t >> 2) + 4 * lg + (t & 3);
            sc[t] = tok <= lim[rb] ? st[t >> 2][t & 3] : -INFINITY;
            mx = fmaxf(mx, sc[t]);
          }
        }
        mx = xor_max<16>(mx);   // the four k-groups of a query row sit 16 lanes apart
        mx = xor_max<32>(mx);
        const float m_new = fmaxf(m_run[rb], mx * scale2);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;
        alpha[rb] = __builtin_amdgcn_exp2f(m_run[rb] - m_use);
        m_run[rb] = m_new;
        float psum = 0.f;
        uint32_t pw[4];
#pragma unroll
        for (int t = 0; t < 8; t += 2) {
          const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[t], scale2, -m_use));
          const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(sc[t + 1], scale2, -m_use));
          psum += p0 + p1;
          pw[t >> 1] = F::pack2(p0, p1);
        }
        l_run[rb] = l_run[rb] * alpha[rb] + psum;
        pfrag[rb] = __builtin_bit_cast(VT, uint4{pw[0], pw[1], pw[2], pw[3]});
      }
      // once the running maxima have settled every alpha is exactly 1: skip the 16 * NDV multiplies
      // of the accumulator rescale for the whole wave then (wave-uniform branch)
      if (__builtin_amdgcn_ballot_w64(alpha[0] != 1.f || alpha[RB - 1] != 1.f) != 0) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int j = 0; j < NDV; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[rb][j][i] *= alpha[rb];
      }
#pragma unroll
      for (int j = 0; j < NDV; ++j) {
        const VT vfrag = __builtin_bit_cast(VT, uint4{vtr[j][0].x, vtr[j][0].y, vtr[j][1].x, vtr[j][1].y});
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb][j] = F::mfma(vfrag, pfrag[rb], acc[rb][j]);
      }
    }
  }

  // ---- epilogue: normalise, stage this wave's 32 x D tile in LDS, store full rows
  __syncthreads();
  char* stg = smem + wave * OUT_TILE_B;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    float l = l_run[rb];
    l += __shfl_xor(l, 16);
    l += __shfl_xor(l, 32);
    const float inv = l > 0.f ? 1.f / l : 0.f;
    const int r = row_base + 16 * rb + li;
    if (a.lse && lg == 0 && r < Lq)
      a.lse[(long long)h * a.lse_stride_h + q0 + r] = l > 0.f ? (m_run[rb] + __log2f(l)) * 0.6931471805599453f : -INFINITY;
#pragma unroll
    for (int j = 0; j < NDV; ++j) {
      // accumulator j of lane (li, lg): head-dim columns 16 j + 4 lg .. + 3 of query row li
      uint2 pk;
      pk.x = F::pack2(acc[rb][j][0] * inv, acc[rb][j][1] * inv);
      pk.y = F::pack2(acc[rb][j][2] * inv, acc[rb][j][3] * inv);
      *(uint2*)(stg + (16 * rb + li) * ROW_B + (16 * j + 4 * lg) * 2) = pk;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  for (int idx = lane; idx < 16 * RB * CPR; idx += 64) {
    const int row = idx / CPR, ch = idx - row * CPR;
    const int r = row_base + row;
    if (r < Lq)
      *(uint4*)(a.o + (long long)(q0 + r) * a.o_stride_t + (long long)h * a.o_stride_h + ch * 8) =
          *(const uint4*)(stg + row * ROW_B + ch * 16);
  }
}

// ---------------------------------------------------------------------------------------------
// KV all-gather for sequence-parallel attention: every rank pushes the K and V rows of its shard
// to the global-order position in every rank's gathered buffer (own copy included), then raises
// flag[rank] = epoch on each peer once all of its workgroups are done.
struct KvPushArgs {
  SymmCtx ctx;
  const uint16_t *k_shard, *v_shard;   // [total_shard, row_elems]
  uint16_t *k_full, *v_full;           // symmetric [total_full, row_elems]
  int* flags;                          // symmetric [world]
  int* done;                           // local counter, zero at launch
  const int *cu_shard, *cu_full;       // [batch + 1]
  int batch, zigzag, epoch;
  long long row_bytes;
  int rows_per_wg;
};

__global__ __launch_bounds__(256) void sp_kv_push_kernel(KvPushArgs p) {
  const int w = p.ctx.num_ranks, rank = p.ctx.rank;
  const int total = p.cu_shard[p.batch];
  const int r0 = blockIdx.x * p.rows_per_wg, r1 = min(r0 + p.rows_per_wg, total);
  const int chunks = (int)(p.row_bytes / 16);
  for (int i = 0; i < w; ++i) {
    const int peer = (rank + (i + blockIdx.x) % w) % w;   // rotate so the links are loaded evenly
    char* kd = (char*)symm_at(p.ctx, p.k_full, peer);
    char* vd = (char*)symm_at(p.ctx, p.v_full, peer);
    for (int idx = threadIdx.x; idx < (r1 - r0) * chunks; idx += 256) {
      const int row = r0 + idx / chunks, ch = idx % chunks;
      int b = 0;
      while (b + 1 < p.batch && row >= p.cu_shard[b + 1]) ++b;
      const int i_loc = row - p.cu_shard[b], Ls = p.cu_shard[b + 1] - p.cu_shard[b];
      int g;
      if (p.zigzag) {
        const int half = Ls >> 1;
        g = i_loc < half ? rank * half + i_loc : (2 * w - 1 - rank) * half + (i_loc - half);
      } else {
        g = rank * Ls + i_loc;
      }
      const long long dst = ((long long)p.cu_full[b] + g) * p.row_bytes + ch * 16;
      const long long src = (long long)row * p.row_bytes + ch * 16;
      const uint4 kq = *(const uint4*)((const char*)p.k_shard + src);
      const uint4 vq = *(const uint4*)((const char*)p.v_shard + src);
      st_wt128_sys(kd + dst, kq);      // write-through: no release fence before the flags
      st_wt128_sys(vd + dst, vq);
    }
  }
  drain_stores();
  __syncthreads();
  if (threadIdx.x == 0) {
    const int prev = atom_add_relaxed_agent(p.done, 1);
    if (prev == (int)gridDim.x - 1) {
      store_relaxed_agent(p.done, 0);
      for (int i = 0; i < w; ++i) store_relaxed_system(symm_at(p.ctx, p.flags, (rank + i) % w) + rank, p.epoch);
    }
  }
}

template <bool IS_BF16, int D, int RB>
hipError_t launch_attn_rb(const AttnArgs& a, int max_seqlen_q, int batch, hipStream_t st) {
  constexpr int lds = attn_lds_bytes(D, RB);
  static bool attr_done = false;
  if (!attr_done) {
    const hipError_t e = hipFuncSetAttribute((const void*)flash_attn_fwd_kernel<IS_BF16, D, RB>,
                                             hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    attr_done = true;
  }
  hipLaunchKernelGGL((flash_attn_fwd_kernel<IS_BF16, D, RB>), dim3((max_seqlen_q + 64 * RB - 1) / (64 * RB), a.q_heads, batch),
                     dim3(256), lds, st, a);
  return hipGetLastError();
}

// 64-row workgroups (three per CU: 138 registers, three-stage ring) while 128-row ones (two per CU) would
// number at most 768: the heaviest causal workgroup then sets the run time, and halving its rows
// shortens it (S=2048 with 32 heads 74 -> 60 us, S=4096 with 20 heads 152 -> 133 us); from 1024 on the
// larger tile wins (S=4096, 32 heads: 200 vs 220 us).  TDIST_ATTN_QROWS = 64 / 128 forces one form.
template <bool IS_BF16, int D>
hipError_t launch_attn(const AttnArgs& a, int max_seqlen_q, int batch, hipStream_t st) {
  static const int forced = [] {
    const char* e = getenv("TDIST_ATTN_QROWS");
    return e ? atoi(e) : 0;
  }();
  const long long wgs128 = (long long)((max_seqlen_q + 127) / 128) * a.q_heads * batch;
  const bool small = forced == 64 || (forced != 128 && wgs128 <= 768);
  return small ? launch_attn_rb<IS_BF16, D, 1>(a, max_seqlen_q, batch, st)
               : launch_attn_rb<IS_BF16, D, 2>(a, max_seqlen_q, batch, st);
}

}  // namespace

// q [total_q, q_heads, D], k/v [total_k, kv_heads, D] (strides in elements, last dim contiguous),
// o like q.  dtype 0 = bf16, 1 = fp16.  flags/err may be null.
TDIST_API int tdist_flash_attn_varlen_fwd(const void* q, const void* k, const void* v, void* o, float* lse,
                                          const int* cu_q, const int* cu_k, int batch, int max_seqlen_q, int q_heads,
                                          int kv_heads, int head_dim, long long q_stride_t, long long q_stride_h,
                                          long long k_stride_t, long long k_stride_h, long long v_stride_t,
                                          long long v_stride_h, long long o_stride_t, long long o_stride_h,
                                          long long lse_stride_h, float sm_scale, int causal, int pos_mode, int rank,
                                          int world, const int* flags, int epoch, int* err, int dtype, void* stream) {
  TDIST_REQUIRE(sm_scale > 0.f, "softmax scale must be positive");
  TDIST_REQUIRE(batch > 0 && max_seqlen_q > 0 && q_heads > 0 && kv_heads > 0 && q_heads % kv_heads == 0, "bad shape");
  TDIST_REQUIRE(head_dim == 64 || head_dim == 128, "head_dim must be 64 or 128");
  TDIST_REQUIRE((q_stride_t | q_stride_h | k_stride_t | k_stride_h | v_stride_t | v_stride_h | o_stride_t | o_stride_h) % 8 == 0,
                "strides must keep 16-byte alignment");
  TDIST_REQUIRE((((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) == 0, "q/k/v/o must be 16-byte aligned");
  TDIST_REQUIRE(pos_mode >= 0 && pos_mode <= 2 && world >= 1 && world <= 32 && rank >= 0 && rank < world, "bad position map");
  AttnArgs a;
  a.q = (const uint16_t*)q, a.k = (const uint16_t*)k, a.v = (const uint16_t*)v, a.o = (uint16_t*)o, a.lse = lse;
  a.cu_q = cu_q, a.cu_k = cu_k;
  a.q_stride_t = q_stride_t, a.q_stride_h = q_stride_h, a.k_stride_t = k_stride_t, a.k_stride_h = k_stride_h;
  a.v_stride_t = v_stride_t, a.v_stride_h = v_stride_h, a.o_stride_t = o_stride_t, a.o_stride_h = o_stride_h;
  a.lse_stride_h = lse_stride_h;
  a.q_heads = q_heads, a.kv_heads = kv_heads, a.causal = causal, a.pos_mode = pos_mode, a.rank = rank, a.world = world;
  a.sm_scale = sm_scale, a.flags = flags, a.epoch = epoch, a.err = err;
  hipStream_t st = (hipStream_t)stream;
  hipError_t e;
  if (head_dim == 64)
    e = dtype == 0 ? launch_attn<true, 64>(a, max_seqlen_q, batch, st) : launch_attn<false, 64>(a, max_seqlen_q, batch, st);
  else
    e = dtype == 0 ? launch_attn<true, 128>(a, max_seqlen_q, batch, st) : launch_attn<false, 128>(a, max_seqlen_q, batch, st);
  TDIST_HIP_TRY(e);
  return 0;
}

TDIST_API int tdist_sp_kv_push(const tdist::SymmCtx* ctx, const void* k_shard, const void* v_shard, void* k_full,
                               void* v_full, int* flags, int* done, const int* cu_shard, const int* cu_full, int batch,
                               int total_shard_rows, long long row_bytes, int zigzag, int epoch, void* stream) {
  TDIST_REQUIRE(batch > 0 && total_shard_rows > 0 && row_bytes > 0 && row_bytes % 16 == 0, "bad KV push shape");
  KvPushArgs p;
  p.ctx = *ctx;
  p.k_shard = (const uint16_t*)k_shard, p.v_shard = (const uint16_t*)v_shard;
  p.k_full = (uint16_t*)k_full, p.v_full = (uint16_t*)v_full;
  p.flags = flags, p.done = done, p.cu_shard = cu_shard, p.cu_full = cu_full;
  p.batch = batch, p.zigzag = zigzag, p.epoch = epoch, p.row_bytes = row_bytes;
  const int target_wgs = 128;
  p.rows_per_wg = (total_shard_rows + target_wgs - 1) / target_wgs;
  if (p.rows_per_wg < 4) p.rows_per_wg = 4;
  const int grid = (total_shard_rows + p.rows_per_wg - 1) / p.rows_per_wg;
  hipLaunchKernelGGL(sp_kv_push_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
  TDIST_HIP_TRY(hipGetLastError());
  return 0;
}
