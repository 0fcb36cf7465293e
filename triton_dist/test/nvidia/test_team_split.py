"""Teams.  The reference splits the NVSHMEM world team into 2-D sub-teams (node x local rank) and
checks team_my_pe / team_n_pes / team_translate_pe (test/nvidia/test_team_split.py).  One MI355X
node is a single full-mesh team: every team id maps onto the world, translation is the identity,
and 2-D layouts (e.g. pipeline x tensor parallel) are built from torch.distributed sub-groups as
test_pp.py does.  The script checks both.

    bash scripts/launch.sh python/triton_dist/test/nvidia/test_team_split.py
    python triton_dist/test/nvidia/test_team_split.py          # world size 1
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

import torch.distributed as dist  # noqa: E402

from triton_dist.language.extra import libshmem_device as shmem  # noqa: E402
from triton_dist.utils import finalize_distributed, initialize_distributed  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", type=int, default=2, help="number of sub-groups to split the world into")
    args = ap.parse_args()
    initialize_distributed()
    rank, w = shmem.my_pe(), shmem.n_pes()
    for team in (shmem.NVSHMEM_TEAM_WORLD, shmem.NVSHMEM_TEAM_SHARED, shmem.NVSHMEMX_TEAM_NODE):
        assert shmem.team_my_pe(team) == rank and shmem.team_n_pes(team) == w
        for pe in range(w):
            assert shmem.team_translate_pe(team, pe, shmem.NVSHMEM_TEAM_WORLD) == pe
    if w > 1 and w % args.groups == 0 and dist.is_initialized():
        per = w // args.groups
        mine = None
        for gi in range(args.groups):                      # every rank creates every group (collective)
            grp = dist.new_group(list(range(gi * per, (gi + 1) * per)), backend="gloo")
            if rank // per == gi:
                mine = grp
        t = torch.tensor([rank], dtype=torch.int64)
        dist.all_reduce(t, group=mine)
        lo = (rank // per) * per
        assert int(t) == sum(range(lo, lo + per)), "sub-group all-reduce saw ranks of another group"
    print(f"✅ team queries on {w} rank(s): world / shared / node teams agree, sub-group split ok", flush=True)
    finalize_distributed()


if __name__ == "__main__":
    main()
