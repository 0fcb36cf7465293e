"""Low-level device helpers.  The reference's language_extra is a set of PTX snippets (scoped
ld / st / atomics, fences, __syncthreads, tid / laneid, red_release, atomic_add_per_warp, wait_eq:
backends/nvidia/language/cuda/language_extra.py:109-929).  Their gfx950 forms live in
csrc/include/tdist/atomics.hpp and primitives.hpp; this script checks how they are lowered
(instructions and cache-scope bits, no GPU needed) and, on a GPU, runs the atomics kernel.

    python triton_dist/test/nvidia/test_language_extra.py

(Scenarios of reference python/triton_dist/test/nvidia/test_language_extra.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402

from triton_distributed_gh_amd.tools import isa  # noqa: E402


def main():
    argparse.ArgumentParser().parse_args()
    body = isa.kernel_isa("primitives_test.hip", "atomics_kernel")
    facts = {
        "store_release at agent scope writes back the L2 first (buffer_wbl2 sc1)": "buffer_wbl2 sc1" in body,
        "load_acquire at agent scope invalidates after the load (buffer_inv sc1)": "buffer_inv sc1" in body,
        "system-scope accesses carry sc0 sc1": "sc0 sc1" in body,
        "atom_add / red_add are global_atomic_add": "global_atomic_add" in body or "flat_atomic_add" in body,
        "atom_cas is global_atomic_cmpswap": "atomic_cmpswap" in body,
    }
    ext = isa.kernel_isa("primitives_test.hip", "language_extra_kernel")
    facts["multimem_st_v4 is a write-through 16-byte store per rank (global_store_dwordx4 ... sc0 sc1)"] = \
        "global_store_dwordx4" in ext and "sc0 sc1" in ext
    facts["xor_max<32> is v_permlane32_swap (no LDS round trip)"] = "v_permlane32_swap" in ext
    facts["ballot / one-atomic-per-wave use the 64-bit exec mask (s_ff1_i32_b64 / v_readlane)"] = \
        "s_ff1_i32_b64" in ext or "v_readfirstlane" in ext or "v_readlane" in ext
    ring = isa.kernel_isa("primitives_test.hip", "ring_queue_kernel")
    facts["__syncthreads is s_barrier"] = "s_barrier" in ring
    for fact, ok in facts.items():
        print(("  ok   " if ok else "  FAIL ") + fact)
    assert all(facts.values())
    if torch.cuda.is_available():
        from triton_distributed_gh_amd.ops import primitives

        word, out = primitives.atomics_smoke()
        torch.cuda.synchronize()
        assert out[:6].tolist() == [5, 5, 8, 11, 12, 20] and int(word) == 20
        print("  ok   atomics kernel: release / acquire / add / cas sequence returns the expected values")
        from triton_distributed_gh_amd.utils.testing import VirtualCluster

        vc = VirtualCluster(2, heap_bytes=1 << 24)
        try:
            bufs = vc.collective(lambda r, env: env.heap.create_tensor([6 * 512], torch.int32))
            syncs = vc.collective(lambda r, env: env.heap.create_tensor([4], torch.int32, arena="sig"))
            outs = [torch.zeros(4, dtype=torch.int32, device="cuda") for _ in range(2)]
            torch.cuda.synchronize()
            vc.run(lambda r, env: primitives.language_extra_smoke(bufs[r], syncs[r], outs[r], env=env))
            assert all(o[:3].tolist() == [1, 1, 1] for o in outs)
        finally:
            vc.close()
        print("  ok   language_extra kernel on two ranks: vector ld / st, shuffles, multimem stores, arrive / wait")
    print("✅ scoped load / store / atomic / barrier helpers lower to the expected gfx950 instructions")


if __name__ == "__main__":
    main()
