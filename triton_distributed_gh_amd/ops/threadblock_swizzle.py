"""Host models of the tile orders the fused kernels walk, with their checkers.

Reference: python/triton_dist/kernels/nvidia/threadblock_swizzle_ag_moe.py:38-235 (pure-Python tile order of
AG + grouped GEMM and its self-check), threadblock_swizzle_ag_moe_triton.py (the same on the device) and
gemm_rs_threadblock_swizzle.py:54-243 (row-block order of GEMM + reduce-scatter).

The orders themselves live in the kernels (csrc/ops/moe.hip ``moe_align_*`` computes ``stage`` / ``order`` on the
device, csrc/ops/gemm_rs.hip visits destinations ``rank + 1, rank + 2, ...``); this module states the same rules
in a few lines of Python so that a test - or a user planning a schedule - can ask "which tile runs at position i,
and which sources must have arrived for it".

AG + grouped GEMM: the rows of expert ``e`` are sorted by source rank, so a 256-row tile of that expert covers a
contiguous range of sources [first, last].  It can start when the LAST of them has arrived; arrival order seen
from ``rank`` is rank, rank + 1, ... (own rows first), so the tile's stage is ``max((s - rank) mod w)`` over its
sources and tiles run in stage order (stable in (expert, tile) within a stage).
"""
from __future__ import annotations

import bisect
import dataclasses
import random
from typing import List, Sequence, Tuple

import numpy as np


def cdiv(x: int, y: int) -> int:
    return (x + y - 1) // y


def cumsum(x: Sequence[int]) -> List[int]:
    out, s = [], 0
    for v in x:
        s += int(v)
        out.append(s)
    return out


def transpose2d(x: Sequence[Sequence[int]]) -> List[List[int]]:
    return [list(col) for col in zip(*x)]


reshape_2d = transpose2d          # [rank][expert] -> [expert][rank]


@dataclasses.dataclass
class Tile:
    expert_id: int
    tiled_m: int                  # tile index inside the expert's rows
    segment_start: int            # first source rank with rows in the tile
    segment_end: int              # last source rank with rows in the tile
    stage: int = 0


def _expert_tiles(expert_id: int, rank: int, tp_size: int, block_size_m: int, counts: Sequence[int]) -> List[Tile]:
    acc = cumsum(counts)
    total = acc[-1] if acc else 0
    tiles = []
    for t in range(cdiv(total, block_size_m)):
        lo, hi = t * block_size_m, min((t + 1) * block_size_m, total) - 1
        s0, s1 = bisect.bisect_right(acc, lo), bisect.bisect_right(acc, hi)
        stage = max((s - rank) % tp_size for s in range(s0, s1 + 1))
        tiles.append(Tile(expert_id, t, s0, s1, stage))
    return tiles


def swizzled_tiles(rank: int, tp_size: int, block_size_m: int, token_cnts_per_rank_per_expert) -> List[Tile]:
    """All tiles of all experts in execution order for ``rank``."""
    per_expert = transpose2d(token_cnts_per_rank_per_expert)
    tiles = [t for e, counts in enumerate(per_expert) for t in _expert_tiles(e, rank, tp_size, block_size_m, counts)]
    return sorted(tiles, key=lambda t: (t.stage, t.expert_id, t.tiled_m))


def threadblock_swizzle_ag_moe(tiled_m: int, rank: int, nexperts: int, tp_size: int, block_size_m: int,
                               token_cnts_per_rank_per_expert) -> Tuple[int, int, Tile]:
    """Position ``tiled_m`` of the execution order -> (stage, expert id, tile)."""
    t = swizzled_tiles(rank, tp_size, block_size_m, token_cnts_per_rank_per_expert)[tiled_m]
    return t.stage, t.expert_id, t


threadblock_swizzle_ag_moe_triton = threadblock_swizzle_ag_moe       # the device form is moe_align's ``order``


def check_swizzled(swizzled: Sequence[Tuple[int, int]], token_cnts_per_rank_per_expert, block_size_m: int = 256) -> None:
    """Every tile of every expert appears exactly once in ``swizzled`` = [(expert, tile), ...]."""
    for e, counts in enumerate(transpose2d(token_cnts_per_rank_per_expert)):
        got = sorted(t for (eid, t) in swizzled if eid == e)
        assert got == list(range(cdiv(sum(int(c) for c in counts), block_size_m))), f"expert {e}: tiles {got}"


def check_with_token_cnt_per_rank_per_expert(token_cnts_per_rank_per_expert, verbose: bool = False,
                                             block_size_m: int = 256) -> None:
    """For every rank: the order is a permutation of all tiles, stages never decrease, and a tile's stage is the
    arrival position of its last source."""
    w = len(token_cnts_per_rank_per_expert)
    for rank in range(w):
        tiles = swizzled_tiles(rank, w, block_size_m, token_cnts_per_rank_per_expert)
        if verbose:
            print(rank, [(t.expert_id, t.tiled_m, t.stage) for t in tiles])
        check_swizzled([(t.expert_id, t.tiled_m) for t in tiles], token_cnts_per_rank_per_expert, block_size_m)
        assert all(a.stage <= b.stage for a, b in zip(tiles, tiles[1:]))
        assert all(0 <= t.stage < w and t.segment_start <= t.segment_end for t in tiles)


check_with_ntokens_per_rank_per_expert = check_with_token_cnt_per_rank_per_expert


def generate_token_cnts_per_rank_per_expert_uniform(ntokens_per_rank: int, nexperts: int, tp_size: int):
    assert ntokens_per_rank % nexperts == 0
    return [[ntokens_per_rank // nexperts] * nexperts for _ in range(tp_size)]


def generate_token_cnts_per_rank_per_expert_random(ntokens_per_rank: int, nexperts: int, tp_size: int, rng=None):
    rng = rng or np.random.default_rng()
    return [rng.multinomial(ntokens_per_rank, [1.0 / nexperts] * nexperts).tolist() for _ in range(tp_size)]


def generate_token_cnts_per_rank_per_expert_random_with_many_zeros(ntokens_per_rank: int, nexperts: int, tp_size: int,
                                                                   zero_rate: float = 0.3, rng=None):
    rng = rng or np.random.default_rng()
    weight = np.array([0.0 if random.random() < zero_rate else random.random() for _ in range(nexperts)]) + 1e-5
    weight /= weight.sum()
    return [rng.multinomial(ntokens_per_rank, weight).tolist() for _ in range(tp_size)]


generate_ntokens_per_rank_per_expert_uniform = generate_token_cnts_per_rank_per_expert_uniform
generate_ntokens_per_rank_per_expert_random = generate_token_cnts_per_rank_per_expert_random
generate_ntokens_per_rank_per_expert_random_with_many_zeros = generate_token_cnts_per_rank_per_expert_random_with_many_zeros


# ---- GEMM + reduce-scatter ------------------------------------------------------------------------------------
def threadblock_swizzle_gemm_reduce_scatter(tiled_m: int, M: int, rank: int, WORLD_SIZE: int, NNODES: int = 1,
                                            BLOCK_SIZE_M: int = 256) -> int:
    """Position ``tiled_m`` of the row-block order -> row block of the [M, N] product computed there.
    Row blocks are visited destination by destination, starting with the rank after this one and ending with the
    rank's own rows (which need no transfer); with several nodes the same rotation applies to nodes first.  Rows
    per rank need not be a multiple of the block: a block that straddles two destinations is computed with the
    earlier one."""
    assert WORLD_SIZE % NNODES == 0 and M % WORLD_SIZE == 0
    local = WORLD_SIZE // NNODES
    node, lrank = rank // local, rank % local
    rows = M // WORLD_SIZE
    order = []                                            # destinations in visiting order
    for dn in range(1, NNODES + 1):
        n = (node + dn) % NNODES
        order += [n * local + (lrank + 1 + d) % local for d in range(local)]
    seen, out = set(), []
    for d in order:
        for blk in range((d * rows) // BLOCK_SIZE_M, cdiv((d + 1) * rows, BLOCK_SIZE_M)):
            if blk not in seen:
                seen.add(blk)
                out.append(blk)
    return out[tiled_m]


threadblock_swizzle_gemm_reduce_scatter_triton = threadblock_swizzle_gemm_reduce_scatter


def check_gemm_rs_swizzle(M: int, WORLD_SIZE: int, NNODES: int = 1, BLOCK_SIZE_M: int = 256) -> None:
    n = cdiv(M, BLOCK_SIZE_M)
    for rank in range(WORLD_SIZE):
        got = [threadblock_swizzle_gemm_reduce_scatter(i, M, rank, WORLD_SIZE, NNODES, BLOCK_SIZE_M) for i in range(n)]
        assert sorted(got) == list(range(n)), f"rank {rank}: {got}"


# ---- AG + GEMM ----------------------------------------------------------------------------------------------------
def threadblock_swizzle_allgather_gemm(tiled_m: int, M: int, rank: int, WORLD_SIZE: int, NNODES: int = 1,
                                       BLOCK_SIZE_M: int = 256) -> int:
    """Position ``tiled_m`` of the row-block order of AG + GEMM -> row block of the gathered A computed there
    (reference: kernels/nvidia/ag_gemm_threadblock_swizzle.py:51-136, a device function there).  Row blocks are
    visited in ARRIVAL order: the rank's own rows (already local), then the ranks after it on its node, then the
    same rotation on the following nodes.  A block that straddles two sources runs with the later one."""
    assert WORLD_SIZE % NNODES == 0 and M % WORLD_SIZE == 0
    local = WORLD_SIZE // NNODES
    node, lrank = rank // local, rank % local
    rows = M // WORLD_SIZE
    sources = [((node + dn) % NNODES) * local + (lrank + d) % local for dn in range(NNODES) for d in range(local)]
    arrival = {s: i for i, s in enumerate(sources)}
    nblk = cdiv(M, BLOCK_SIZE_M)

    def last_source(blk):
        lo, hi = blk * BLOCK_SIZE_M, min((blk + 1) * BLOCK_SIZE_M, M) - 1
        return max(arrival[s] for s in range(lo // rows, hi // rows + 1))

    order = sorted(range(nblk), key=lambda b: (last_source(b), arrival[min((b * BLOCK_SIZE_M) // rows, WORLD_SIZE - 1)], b))
    return order[tiled_m]


def check_ag_gemm_swizzle(M: int, WORLD_SIZE: int, NNODES: int = 1, BLOCK_SIZE_M: int = 256) -> None:
    n = cdiv(M, BLOCK_SIZE_M)
    for rank in range(WORLD_SIZE):
        got = [threadblock_swizzle_allgather_gemm(i, M, rank, WORLD_SIZE, NNODES, BLOCK_SIZE_M) for i in range(n)]
        assert sorted(got) == list(range(n)), f"rank {rank}: {got}"
        assert got[0] * BLOCK_SIZE_M // (M // WORLD_SIZE) == rank, "the first block is one of the rank's own"
