"""Scoped atomics (load/store acquire-release, add, CAS at workgroup / agent / system scope) on the
device, checked against the expected value sequence.

    python triton_dist/test/amd/test_libdevice.py

(Scenario of reference python/triton_dist/test/amd/test_libdevice.py: the AMD ``libdevice`` atomics
of backends/amd/language/hip/libdevice.py:518-734.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

from triton_distributed_gh_amd.ops import primitives  # noqa: E402


def main():
    argparse.ArgumentParser().parse_args()
    assert torch.cuda.is_available(), "needs a GPU"
    word, out = primitives.atomics_smoke()
    torch.cuda.synchronize()
    got = out[:6].tolist()
    # store_release 5 | load_acquire | atom_add_acqrel(+3) returns 5 | cas(8->11) returns 8 |
    # failed cas returns 11 ... red_add(+1) -> 12 | store_relaxed 20 / load_acquire
    assert got == [5, 5, 8, 11, 12, 20] and int(word) == 20, got
    print("✅ scoped atomics (acquire/release/relaxed x workgroup/agent/system) passed", flush=True)


if __name__ == "__main__":
    main()
