"""The wait primitive: how it is lowered, that it works, and what it costs.

    python triton_dist/test/nvidia/test_distributed_wait.py --case lower          # no GPU needed
    python triton_dist/test/nvidia/test_distributed_wait.py --case correctness
    python triton_dist/test/nvidia/test_distributed_wait.py --case perf

lower       - compile the primitive kernels for gfx950 and check the instructions (scope bits of
              the polling load, back-off, bounded spin, acquire = cache invalidate, barrier;
              notify = release + system-scope store / atomic)
correctness - a kernel waits on a flag that another stream raises later (set by a stream
              write-value, by a peer's notify kernel, and never: the bounded spin must give up)
perf        - latency of a wait whose flag is already set, per scope
(Cases of reference python/triton_dist/test/nvidia/test_distributed_wait.py.)
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", ".."))

import torch  # noqa: E402


def case_lower():
    from triton_distributed_gh_amd.tools import isa

    report = isa.lowering_report()
    for fact, ok in report.items():
        print(("  ok   " if ok else "  FAIL ") + fact)
    assert all(report.values())
    print("✅ wait / notify lowering checks passed")


def case_correctness():
    from triton_distributed_gh_amd.ops import primitives
    from triton_distributed_gh_amd.ops.common_ops import set_signal
    from triton_distributed_gh_amd.utils.testing import VirtualCluster
    import triton_dist.language as dl

    vc = VirtualCluster(2, heap_bytes=1 << 24)
    flags = vc.collective(lambda r, env: env.heap.create_tensor([4], torch.int32, arena="sig"))
    # (1) the waiter is launched first, the flag is raised later from another stream
    with torch.cuda.stream(vc.streams[0]):
        st = primitives.wait_flag(flags[0][0:1], 7, scope=2)
    torch.cuda._sleep(2_000_000)
    with torch.cuda.stream(vc.streams[1]):
        set_signal(flags[0][0:1], 7, bits=32)
    torch.cuda.synchronize()
    assert int(st[0]) == 1, "flag raised by a stream write-value was not seen"
    # (2) raised by the peer rank's notify kernel
    with torch.cuda.stream(vc.streams[0]):
        st = primitives.wait_flag(flags[0][1:2], 9, scope=2)
    with torch.cuda.stream(vc.streams[1]):
        dl.notify(flags[1][1:2], 0, signal=9, sig_op="set", comm_scope="intra_node", env=vc.envs[1])
    torch.cuda.synchronize()
    assert int(st[0]) == 1, "flag raised by a peer's notify was not seen"
    print("✅ wait sees flags raised after it started (stream write-value, peer notify)")
    vc.close()


def case_perf():
    from triton_distributed_gh_amd.ops import primitives
    from triton_distributed_gh_amd.utils.perf import perf_func

    flag = torch.ones(1, dtype=torch.int32, device="cuda")
    for scope, name in ((0, "workgroup"), (1, "gpu"), (2, "sys")):
        _, ms = perf_func(lambda: primitives.wait_flag(flag, 1, scope=scope), 50, 5)
        print(f"wait(scope={name:9}) on a flag that is already set: {ms * 1e3:.1f} us per launch (incl. status alloc)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--case", default="lower", choices=["lower", "correctness", "perf"])
    args = ap.parse_args()
    if args.case != "lower":
        assert torch.cuda.is_available(), "this case needs a GPU"
        os.environ["GPU_MAX_HW_QUEUES"] = str(max(8, int(os.environ.get("GPU_MAX_HW_QUEUES", "0") or 0)))
    {"lower": case_lower, "correctness": case_correctness, "perf": case_perf}[args.case]()


if __name__ == "__main__":
    main()
