"""SIMT region helpers.  The reference needs a simt dialect to run per-thread code inside a
tile program (language/simt_ops.py, used by kernels/nvidia/ep_a2a.py:281-288 to permute a small
table); HIP kernels are per-thread code already, so the region is a no-op scope and extract /
insert are scalar accesses.  The script runs the reference's table permutation
[src_rank, ep_rank, e] -> [ep_rank, e, src_rank] written with those helpers and compares it with
permute.

    python triton_dist/test/nvidia/test_simt.py

(Scenario of reference python/triton_dist/test/nvidia/test_simt.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

from triton_dist.language import extract, insert, simt_exec_region  # noqa: E402


def main():
    argparse.ArgumentParser().parse_args()
    w, epr = 4, 3
    src = torch.arange(w * w * epr, dtype=torch.int32).view(w, w, epr)      # [src_rank, ep_rank, e]
    dst = torch.zeros(w, epr, w, dtype=torch.int32)
    with simt_exec_region():
        for s in range(w):
            for r in range(w):
                for e in range(epr):
                    insert(dst, extract(src, (s, r, e)), (r, e, s))
    assert torch.equal(dst, src.permute(1, 2, 0))
    print("✅ simt_exec_region / extract / insert: table permutation matches permute()")


if __name__ == "__main__":
    main()
