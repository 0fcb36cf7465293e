"""Per-workgroup timeline of the fused TP kernels, one rank alone on the GPU (see probe_solo_rank.py).

    python scripts/dev/trace_solo_rank.py 1,2,8

Arms the per-workgroup event trace (csrc/include/tdist/wg_trace.hpp), runs one traced call of AG-GEMM and
one of GEMM-RS after a few warm-ups and prints where the time of the 256 workgroups went."""
import ctypes
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from triton_distributed_gh_amd import _lib  # noqa: E402
from triton_distributed_gh_amd.ops.ag_gemm import ag_gemm_intra_node, create_ag_gemm_intra_node_context  # noqa: E402
from triton_distributed_gh_amd.ops.gemm import gemm_nt  # noqa: E402
from triton_distributed_gh_amd.ops.gemm_rs import create_gemm_rs_intra_node_context, gemm_rs_intra_node  # noqa: E402
from triton_distributed_gh_amd.utils.testing import VirtualCluster  # noqa: E402

M, N, K = 4096, int(os.environ.get("TRACE_N", 8192)), 8192      # TRACE_N: another width (N // w per rank)
EPOCH_MOD = 4093
CAP = 160
NAMES = {1: "start", 2: "ticket", 3: "wait", 4: "mainloop", 5: "epilogue", 6: "other item", 7: "loop exit"}


def analyse(buf, title):
    ev = buf.cpu().view(-1, CAP, 2)
    t0 = None
    rows = []
    for b in range(ev.shape[0]):
        e = [(int(x[0]) >> 32, int(x[0]) & 0xffffffff, int(x[1])) for x in ev[b] if int(x[1]) != 0]
        if e:
            rows.append(e)
            t0 = e[0][2] if t0 is None else min(t0, e[0][2])
    if not rows:
        print(title, ": no events")
        return
    end = max(e[-1][2] for e in rows)
    span = (end - t0) / 100.0
    seg = defaultdict(list)
    n_items = []
    for e in rows:
        n_items.append(sum(1 for x in e if x[0] == 2))
        for a, b in zip(e, e[1:]):
            seg[NAMES.get(b[0], str(b[0]))].append((b[2] - a[2]) / 100.0)
    print(f"{title}: span {span:.1f} us over {len(rows)} workgroups; tickets per workgroup "
          f"min {min(n_items)} max {max(n_items)}")
    starts = sorted((e[0][2] - t0) / 100.0 for e in rows)
    exits = sorted((end - e[-1][2]) / 100.0 for e in rows)
    print(f"   start skew: median {starts[len(starts) // 2]:.1f} us, max {starts[-1]:.1f} us; "
          f"idle before the end: median {exits[len(exits) // 2]:.1f} us, max {exits[-1]:.1f} us")
    args = defaultdict(list)            # tags >= 8 carry a duration (100 MHz ticks) as their argument
    for e in rows:
        for tag, arg, _ in e:
            if tag >= 8:
                args[tag].append(arg / 100.0)
    for tag in sorted(args):
        v = sorted(args[tag])
        print(f"   arg of tag {tag}: n={len(v)} mean {sum(v) / len(v):.2f} us  median {v[len(v) // 2]:.2f}  max {v[-1]:.2f}")
    for k, v in seg.items():
        v.sort()
        print(f"   -> {k:10s} n={len(v):5d} mean {sum(v) / len(v):8.2f} us  median {v[len(v) // 2]:8.2f}  "
              f"max {v[-1]:8.2f}  total/wg {sum(v) / len(rows):8.2f}")


def main():
    ws = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2").split(",")]
    dt = torch.bfloat16
    os.environ["TDIST_FORCE_FUSED"] = "1"
    lib = _lib.hip()
    for w in ws:
        vc = VirtualCluster(w, heap_bytes=1 << 30)
        ag = vc.collective(lambda r, env: create_ag_gemm_intra_node_context(
            M, N // w, K, dt, dt, r, w, num_ctas=256, env=env, producer=os.environ.get("PROBE_PRODUCER", "kernel")))
        rs = vc.collective(lambda r, env: create_gemm_rs_intra_node_context(M, N, dt, r, w, num_ctas=256, env=env))
        a1 = (torch.rand(M // w, K, device="cuda") * 2 - 1).to(dt)
        w1 = (torch.rand(N // w, K, device="cuda") * 2 - 1).to(dt)
        a2 = (torch.rand(M, K // w, device="cuda") * 2 - 1).to(dt)
        w2 = (torch.rand(N, K // w, device="cuda") * 2 - 1).to(dt)
        buf = torch.zeros(256 * CAP * 2, dtype=torch.int64, device="cuda")

        def run_ag():
            c = ag[0]
            c.flags[c.call_count & 1][0].fill_(c.call_count % EPOCH_MOD + 1)
            return ag_gemm_intra_node(a1, w1, False, c)

        def run_rs():
            c = rs[0]
            c.flags[c.call_count & 1][0].fill_(c.call_count % EPOCH_MOD + 1)
            return gemm_rs_intra_node(a2, w2, c)

        a1f = (torch.rand(M, K, device="cuda") * 2 - 1).to(dt)
        o1 = torch.empty(M, N // w, device="cuda", dtype=dt)

        def run_plain():
            return gemm_nt(a1f, w1, out=o1)

        with torch.cuda.stream(vc.streams[0]):
            for fn, name in ((run_plain, "plain gemm (ag shape)"), (run_ag, "ag_gemm"), (run_rs, "gemm_rs")):
                # steady state: the traced call is the last of a back-to-back series (clocks settled)
                buf.zero_()
                for _ in range(12):
                    fn()
                _lib.check(lib.tdist_debug_set_wg_trace(ctypes.c_void_p(buf.data_ptr()), CAP), "set trace")
                fn()
                _lib.check(lib.tdist_debug_set_wg_trace(None, 0), "clear trace")
                torch.cuda.synchronize()
                analyse(buf, f"w={w} {name}")
        ag[0].check_error()
        rs[0].check_error()
        vc.close()


if __name__ == "__main__":
    main()
