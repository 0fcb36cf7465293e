// Host-side planning for the tile family (gemm_gfx950.hpp: mainloop_2ph): which macro-tile and which K split a
// GEMM of a given shape should run with on a given number of workgroup slots.  Shared by the plain GEMM
// (ops/gemm.hip) and the fused tensor-parallel kernels (ops/ag_gemm.hip).
#pragma once
#include <stdlib.h>

namespace tdist {
namespace gemm {

// Plan of a plain GEMM on the tile family: tile shape and K split.  Cost model (all times in ns): one work item
// (a tile, or one K range of it) takes steps * step_ns(shape, co-resident workgroups) + FIXED, split-K adds the
// parking of the accumulators and, for the last arriver, the reading of the other partials; the launch takes
// ceil(items / slots) rounds of that (slots = CUs x co-resident workgroups).  step_ns: measured on MI355X with
// bf16 random operands (scripts/dev/probe_tile_family.py time); the 8-bit types run the same steps over twice the K.
struct TilePlan {
  int mt, nt, split_k;
  long long partial_bytes;
  int n_counters;
  int n_full;        // tiles [0, n_full) run unsplit (whole rounds); only the tiles of the last, partial round are split
};
inline double tile_step_ns(int mt, int nt, bool chip_full) {
  if (mt == 8 && nt == 4) return chip_full ? 1430.0 : 1620.0;
  if (mt == 8) return 900.0;
  if (nt == 4) return 840.0;
  return chip_full ? 950.0 : 580.0;       // 128 x 128: two co-resident workgroups share the CU when the chip is full
}
// `slots_1`: workgroups of a one-per-CU tile shape that run at once (the CU count, or the consumers' share of it);
// `allow_128`: whether the 128 x 128 shape (two co-resident workgroups per slot) may be chosen.
inline TilePlan plan_tiles(int M, int N, int K16, int slots_1, int bm, int bn, int split_k, bool allow_128) {
  const int cus = slots_1;
  static const int forced = getenv("TDIST_GEMM_TILE") ? atoi(getenv("TDIST_GEMM_TILE")) : 0;   // e.g. 842 = MT 8, NT 4, split 2
  const int cand[4][2] = {{8, 4}, {8, 2}, {4, 4}, {4, 2}};
  double best = 1e30;
  TilePlan pick{8, 4, 1, 0, 0, 0};
  const int ksteps = K16 / 64;
  for (int ci = 0; ci < 4; ++ci) {
    const int mt = cand[ci][0], nt = cand[ci][1];
    if ((bm && bm != 32 * mt) || (bn && bn != 64 * nt)) continue;
    if (forced && (forced / 100 != mt || (forced / 10) % 10 != nt)) continue;
    if (!allow_128 && mt == 4 && nt == 2) continue;
    const int tbm = 32 * mt, tbn = 64 * nt;
    const long long tiles = (long long)((M + tbm - 1) / tbm) * ((N + tbn - 1) / tbn);
    const int per_cu = (mt == 4 && nt == 2) ? 2 : 1;
    const long long slots = (long long)cus * per_cu;
    for (int s = 1; s <= 4; ++s) {
      if (split_k && s != split_k) continue;
      if (forced && forced % 10 && s != forced % 10) continue;
      // A split applies to the LAST round only: n_full tiles fill whole rounds unsplit, the remaining `rem` tiles
      // (a partial round that would leave slots idle) are cut into s K ranges each, rem * s <= slots.
      const long long full_rounds = tiles / slots, rem = tiles % slots;
      const long long n_full = s > 1 ? full_rounds * slots : 0;
      if (s > 1 && (rem == 0 || ksteps / s < 8 || rem * s > slots)) {
        if (!(split_k && s == split_k && tiles * s <= slots && ksteps >= 2 * s)) continue;   // (forced small split)
      }
      const bool full = tiles * s > cus / 2;
      const double tile_kb = tbm * tbn * 4 / 1024.0;
      const double step = tile_step_ns(mt, nt, full);
      double total;
      if (s == 1) {
        const long long rounds = (tiles + slots - 1) / slots;
        total = rounds * (ksteps * step + 7000.0);
      } else {
        const double fix = 2500.0 + 20.0 * tile_kb + (s - 1) * 30.0 * tile_kb;
        total = full_rounds * (ksteps * step + 7000.0) + ((ksteps / s) * step + 7000.0) + fix;
      }
      if (total < best) {
        best = total;
        pick = TilePlan{mt, nt, s, 0, 0, (int)n_full};
      }
    }
  }
  if (pick.split_k > 1) {
    const int tbm = 32 * pick.mt, tbn = 64 * pick.nt;
    const long long tiles = (long long)((M + tbm - 1) / tbm) * ((N + tbn - 1) / tbn);
    pick.partial_bytes = (tiles - pick.n_full) * pick.split_k * tbm * tbn * 4;
    pick.n_counters = (int)(tiles - pick.n_full);
  }
  return pick;
}


}  // namespace gemm
}  // namespace tdist
