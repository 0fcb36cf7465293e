"""Headline benchmark: intra-node AG-GEMM + GEMM-RS latency, M=4096 N=K=8192 bf16, TP = --gpus.

One step = one tensor-parallel MLP-style pair on every rank:
    AG-GEMM : C1[M, N/w] = AllGather_M(A1[M/w, K]) @ W1[N/w, K]^T
    GEMM-RS : C2[M/w, N] = ReduceScatter_M(A2[M, K/w] @ W2[N, K/w]^T)
with M=4096, N=K=8192 (the headline shape; total work is fixed as the GPU count grows -> strong scaling).
`value` is the per-step latency in microseconds (AG-GEMM + GEMM-RS, max over ranks); the two ops
are also timed separately, and the same pair built from torch.matmul + RCCL collectives is timed in
the same process as the yardstick the reference uses (reference:
python/triton_dist/test/amd/test_gemm_rs_intra_node.py:48-64, layers/amd/tp_mlp.py:156-202).

The timed path is ALWAYS this repository's kernels.  Before timing, the fused ops are checked
element by element against an fp32-accumulated gold on fresh random inputs - 50 times at TP > 1 - and
once more after the timed loop.  The tolerance of an element is 2e-2 of the magnitude that was rounded
to bf16 on its way (|gold| for AG-GEMM; the sum of the |per-rank partials| for GEMM-RS, whose partials
travel in bf16 as in the reference and in RCCL: one bf16 ulp, 2^-7 relative at worst, per partial plus the
final rounding) plus 1e-3 - relative like the reference's rtol = 1e-2 but without its atol = 1e-2 floor,
which would hide a wrong row of small values.  If the symmetric
heap cannot be set up, a check fails or a device-side wait times out, the script prints the JSON line
with "value": null (the torch + RCCL times stay in it as extra fields) and exits with status 1: it
never reports a torch + RCCL time as the result.

Besides the K timed steps (the headline `value`), every op is timed against its torch (+ RCCL) counterpart in
interleaved rounds (fused, torch, fused, torch, ... at least five rounds each, barrier + synchronize around every
burst, max over ranks) and the MEDIANS are reported with the per-op and pair speed-ups derived from them - the
order of measurement then cannot favour either side on a power-managed part.  At TP > 1 both AG-GEMM producers
(push work items inside the consumer launch / SDMA copy engines) are timed in those rounds and the headline step
runs the one with the lower median (`producer`, `producer_rule`).  `--shapes baseline` adds the reference's
published MI308X shapes and the six m8192 chart shapes of each op as extra fields.  The JSON also carries the
device topology as seen by the driver and the perf model's link- / MFMA-bound estimate of each op.

Launch:  python bench.py --gpus 1            (single process)
         python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
                --master-port P bench.py --gpus N --steps K --warmup W
Synthetic data, random-init weights.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

M, N, K = 4096, 8192, 8192
RTOL, ATOL = 2e-2, 1e-3


def _sync_all(env):
    torch.cuda.synchronize()
    if env.world_size > 1:
        dist.barrier(group=env.cpu_group)
    torch.cuda.synchronize()


def _time_steps(env, fn, steps, warmup):
    for _ in range(warmup):
        fn()
    _sync_all(env)
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    _sync_all(env)
    dt = (time.perf_counter() - t0) / steps
    if env.world_size > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=env.cpu_group)
        dt = float(t[0])
    return dt * 1e6  # us per step


def _burst_us(env, fn, iters):
    """us per call of `iters` back-to-back calls, max over ranks (barrier + synchronize on both sides)."""
    _sync_all(env)
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    _sync_all(env)
    dt = (time.perf_counter() - t0) / iters
    if env.world_size > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=env.cpu_group)
        dt = float(t[0])
    return dt * 1e6


def _interleaved(env, fns: dict, rounds: int, iters: int) -> dict:
    """Median us per call of every entry of `fns`, measured A B C A B C ... for `rounds` rounds."""
    names = list(fns)
    for n in names:                       # one untimed pass: allocations, attribute calls, clocks
        for _ in range(2):
            fns[n]()
    samples = {n: [] for n in names}
    for _ in range(rounds):
        for n in names:
            samples[n].append(_burst_us(env, fns[n], iters))
    return {n: sorted(v)[len(v) // 2] for n, v in samples.items()}


# The reference's published AMD numbers (8 x MI308X, docs/getting-started/e2e/e2e_dense.md:44-45) and the six
# m8192 chart shapes of docs/amd-perf.md:4,7 - global (M, N, K); AG-GEMM shards N, GEMM-RS shards K.
BASELINE_AG = [(4096, 6400 * 8, 5120), (8192, 11008, 4096), (8192, 14336, 4096), (8192, 28672, 8192),
               (8192, 53248, 16384), (8192, 29568, 8192)]
BASELINE_RS = [(4096, 5120, 3200 * 8), (8192, 4096, 12288), (8192, 4096, 14336), (8192, 3584, 14336),
               (8192, 4608, 36864), (8192, 8192, 28672), (8192, 8192, 30720)]


def _plain_env():
    """torch.distributed only (no symmetric heap): enough to time torch.matmul + RCCL for the report
    of a run whose fused path could not even start."""
    import datetime
    from types import SimpleNamespace

    rank, world = int(os.environ.get("RANK", 0)), int(os.environ.get("WORLD_SIZE", 1))
    ndev = torch.cuda.device_count()
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", rank)) % max(ndev, 1))
    torch.cuda.set_device(dev)
    shared = world > ndev
    cpu_group = tp_group = None
    if world > 1:
        if not dist.is_initialized():
            dist.init_process_group(backend="gloo" if shared else "cpu:gloo,cuda:nccl", world_size=world, rank=rank,
                                    timeout=datetime.timedelta(seconds=600))
        cpu_group = dist.new_group(ranks=list(range(world)), backend="gloo")
        tp_group = dist.group.WORLD
    return SimpleNamespace(rank=rank, world_size=world, device=dev, heap=None, shared_gpu=shared, has_rccl=not shared,
                           cpu_group=cpu_group, tp_group=tp_group, heap_mode="none", selftest=None, selftest_log=())


def _all_agree(env, ok: bool) -> bool:
    if env.world_size > 1:
        flag = torch.tensor([1 if ok else 0])
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=env.cpu_group)
        ok = bool(flag[0])
    return ok


def _mismatch(got: torch.Tensor, gold: torch.Tensor, mag: torch.Tensor) -> str:
    """'' when every element satisfies |got - gold| <= ATOL + RTOL * mag, else a description."""
    g = got.float()
    bad = (g - gold).abs() > ATOL + RTOL * mag
    n = int(bad.sum())
    if n == 0 and bool(torch.isfinite(g).all()):
        return ""
    idx = bad.nonzero()[0].tolist() if n else [-1, -1]
    return (f"{n} of {g.numel()} elements outside {ATOL} + {RTOL} * magnitude (first at {idx}, max abs diff "
            f"{(g - gold).abs().max().item():.4e})")


def _baseline_shapes(env, ctas, producer, baseline_on, args):
    """Fused vs torch (+ RCCL) medians at the reference's published shapes; contexts are created per shape and
    their symmetric buffers returned to the heap afterwards."""
    from triton_distributed_gh_amd.ops.ag_gemm import ag_gemm_intra_node, create_ag_gemm_intra_node_context
    from triton_distributed_gh_amd.ops.gemm_rs import create_gemm_rs_intra_node_context, gemm_rs_intra_node

    w, r, dev, dt = env.world_size, env.rank, env.device, torch.bfloat16
    rows = []
    rounds, iters = 5, 5

    def rand(shape, scale=1.0):
        return ((torch.rand(*shape, device=dev) * 2 - 1) * scale).to(dt)

    def release(ctx, lists):
        for name in lists:
            for parity in getattr(ctx, name):
                env.heap.free_tensor(parity[r])

    for (m, n, k) in BASELINE_AG:
        if m % w or n % w:
            continue
        a, b = rand((m // w, k)), rand((n // w, k), 0.05)
        full = torch.empty(m, k, device=dev, dtype=dt)
        ctx = create_ag_gemm_intra_node_context(m, n // w, k, dt, dt, r, w, env.tp_group, num_ctas=ctas, producer=producer)

        def t_ref():
            if w > 1:
                dist.all_gather_into_tensor(full, a, group=env.tp_group)
                return torch.matmul(full, b.t())
            return torch.matmul(a, b.t())

        fns = {"fused": lambda: ag_gemm_intra_node(a, b, False, ctx)}
        if baseline_on:
            fns["torch"] = t_ref
        med = _interleaved(env, fns, rounds, iters)
        ctx.check_error()
        rows.append({"op": "ag_gemm", "M": m, "N": n, "K": k, "fused_us": round(med["fused"], 1),
                     "torch_us": round(med["torch"], 1) if "torch" in med else None,
                     "speedup": round(med["torch"] / med["fused"], 3) if "torch" in med else None})
        torch.cuda.synchronize()
        release(ctx, ("workspaces", "flags"))
        del ctx, a, b, full
    for (m, n, k) in BASELINE_RS:
        if m % w or k % w or (k // w) % 64:
            continue
        a, b = rand((m, k // w)), rand((n, k // w), 0.05)
        full = torch.empty(m, n, device=dev, dtype=dt)
        out = torch.empty(m // w, n, device=dev, dtype=dt)
        ctx = create_gemm_rs_intra_node_context(m, n, dt, r, w, env.tp_group, num_ctas=ctas)

        def t_ref():
            torch.matmul(a, b.t(), out=full)
            if w > 1:
                dist.reduce_scatter_tensor(out, full, group=env.tp_group)
                return out
            return full

        fns = {"fused": lambda: gemm_rs_intra_node(a, b, ctx)}
        if baseline_on:
            fns["torch"] = t_ref
        med = _interleaved(env, fns, rounds, iters)
        ctx.check_error()
        rows.append({"op": "gemm_rs", "M": m, "N": n, "K": k, "fused_us": round(med["fused"], 1),
                     "torch_us": round(med["torch"], 1) if "torch" in med else None,
                     "speedup": round(med["torch"] / med["fused"], 3) if "torch" in med else None})
        torch.cuda.synchronize()
        release(ctx, ("scatter", "flags"))
        del ctx, a, b, full, out
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-baseline", action="store_true", help="skip the torch+RCCL yardstick")
    ap.add_argument("--producer", choices=("kernel", "copy_engine"), default=None,
                    help="AG-GEMM producer: in-kernel push work items (default) or the SDMA copy engines")
    ap.add_argument("--check-iters", type=int, default=-1,
                    help="fresh-input checks before timing (default: 50 at TP > 1 on real devices, 3 otherwise)")
    ap.add_argument("--rounds", type=int, default=7, help="interleaved fused / torch rounds per op (>= 5)")
    ap.add_argument("--shapes", choices=("headline", "baseline"), default="headline",
                    help="baseline: also time the reference's published shapes (extra JSON fields)")
    args = ap.parse_args()
    args.rounds = max(5, args.rounds)

    from triton_distributed_gh_amd.ops.ag_gemm import ag_gemm_intra_node, create_ag_gemm_intra_node_context
    from triton_distributed_gh_amd.ops.gemm_rs import create_gemm_rs_intra_node_context, gemm_rs_intra_node
    from triton_distributed_gh_amd.utils import finalize_distributed, initialize_distributed

    # The symmetric heap (hipIpc mappings of every peer, proven coherent by the start-up self-test) is
    # what the fused path stands on.  If it cannot be set up there is no result to report.
    problems = []
    notes = []                        # remarks that do not invalidate the result
    try:
        if os.environ.get("TDIST_BENCH_FORCE_PLAIN") == "1":      # exercises the failure path below
            raise RuntimeError("forced by TDIST_BENCH_FORCE_PLAIN")
        env = initialize_distributed(seed=0)
    except Exception as exc:  # noqa: BLE001 - reported in the JSON line, exit status 1
        problems.append(f"symmetric heap unavailable ({type(exc).__name__}: {exc})")
        env = _plain_env()
    w, r, dev = env.world_size, env.rank, env.device
    assert w == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={w}"
    dt = torch.bfloat16
    Mr = M // w
    ctas = max(8, 224 // w) if env.shared_gpu else 0

    def fresh(shape, scale=1.0):
        return ((torch.rand(*shape, device=dev) * 2 - 1) * scale).to(dt)

    a1, w1 = fresh((Mr, K)), fresh((N // w, K), 0.05)
    a2, w2 = fresh((M, K // w)), fresh((N, K // w), 0.05)

    ag_ctx = rs_ctx = None
    ag_ctxs = {}                      # producer -> context (TP > 1: both producers where the copy engines may be used)
    if env.heap is not None:
        want = [args.producer] if args.producer else (["kernel", "copy_engine"] if w > 1 else ["kernel"])
        for prod in want:
            try:
                ag_ctxs[prod] = create_ag_gemm_intra_node_context(M, N // w, K, dt, dt, r, w, env.tp_group, M_PER_CHUNK=256,
                                                                  num_ctas=ctas, producer=prod)
            except RuntimeError as exc:      # (copy engines: refused when several processes share one GPU)
                if prod == want[0]:
                    raise
                notes.append(f"producer {prod} not timed ({exc})")
            # a producer is timed only where every rank has it
            if not _all_agree(env, prod in ag_ctxs) and prod in ag_ctxs:
                del ag_ctxs[prod]
                notes.append(f"producer {prod} not timed (a peer rank could not set it up)")
        ag_ctx = ag_ctxs[want[0]]
        rs_ctx = create_gemm_rs_intra_node_context(M, N, dt, r, w, env.tp_group, M_PER_CHUNK=256, num_ctas=ctas)

    # ---- torch + RCCL implementation of the same pair (yardstick; not available on a shared GPU) ----
    a1_full = torch.empty(M, K, device=dev, dtype=dt)
    c2_full = torch.empty(M, N, device=dev, dtype=dt)
    c2_out = torch.empty(Mr, N, device=dev, dtype=dt)

    def torch_ag_gemm():
        if w > 1:
            dist.all_gather_into_tensor(a1_full, a1, group=env.tp_group)
            return torch.matmul(a1_full, w1.t())
        return torch.matmul(a1, w1.t())

    def torch_gemm_rs():
        torch.matmul(a2, w2.t(), out=c2_full)
        if w > 1:
            dist.reduce_scatter_tensor(c2_out, c2_full, group=env.tp_group)
            return c2_out
        return c2_full

    def fused_ag():
        return ag_gemm_intra_node(a1, w1, False, ag_ctx)

    def fused_ag_with(prod):
        return lambda: ag_gemm_intra_node(a1, w1, False, ag_ctxs[prod])

    def fused_rs():
        return gemm_rs_intra_node(a2, w2, rs_ctx)

    # ---- fp32-accumulated gold of both ops on the current inputs -------------------------------------
    sample = torch.arange(0, Mr, max(1, Mr // 128), device=dev)        # shared GPU: rows checked per shard

    def gold_ag():
        """fp32 [M, N/w] (or the sampled rows of every shard on a shared GPU) and the matching row index."""
        if w == 1:
            return a1.float() @ w1.float().t(), None
        if env.has_rccl:
            dist.all_gather_into_tensor(a1_full, a1, group=env.tp_group)
            return a1_full.float() @ w1.float().t(), None
        mine = a1[sample].contiguous().view(torch.uint8).cpu()          # (bytes: gloo has no 16-bit integer type)
        parts = [torch.empty_like(mine) for _ in range(w)]
        dist.all_gather(parts, mine, group=env.cpu_group)
        rows = torch.cat([p.view(dt) for p in parts]).to(dev)
        idx = torch.cat([sample + j * Mr for j in range(w)])
        return rows.float() @ w1.float().t(), idx

    def gold_rs():
        """(gold, magnitude, rows): fp32 sum over ranks of the bf16-rounded partials of this rank's shard
        [M/w, N] and the sum of their absolute values (sampled rows on a shared GPU)."""
        if w == 1:
            g = a2.float() @ w2.float().t()
            return g, g.abs(), None
        if env.has_rccl:
            part = (a2.float() @ w2.float().t()).to(dt).float()
            out = torch.empty(Mr, N, device=dev, dtype=torch.float32)
            mag = torch.empty(Mr, N, device=dev, dtype=torch.float32)
            dist.reduce_scatter_tensor(out, part, group=env.tp_group)
            dist.reduce_scatter_tensor(mag, part.abs_(), group=env.tp_group)
            return out, mag, None
        idx = torch.cat([sample + j * Mr for j in range(w)])
        t = (a2[idx].float() @ w2.float().t()).to(dt).float().cpu()    # [w * |sample|, N]
        m = t.abs()
        dist.all_reduce(t, group=env.cpu_group)
        dist.all_reduce(m, group=env.cpu_group)
        return t.view(w, -1, N)[r].to(dev), m.view(w, -1, N)[r].to(dev), sample

    def check_once(tag: str, keep=None) -> str:
        """'' when every timed path reproduces the gold.  A producer of AG-GEMM that fails (on any rank) while
        another one passes is dropped with a remark instead of failing the run - unless it is ``keep``, the one the
        headline was timed with - so one broken transport cannot cost the whole measurement."""
        nonlocal ag_ctx
        c1s, broke = {}, {}
        for prod in ag_ctxs:                                         # every producer that will be timed
            try:
                c1s[prod] = fused_ag_with(prod)()
                torch.cuda.synchronize()
                ag_ctxs[prod].check_error()
            except Exception as exc:  # noqa: BLE001
                broke[prod] = f"failed: {type(exc).__name__}: {exc}"
        c2 = fused_rs()
        torch.cuda.synchronize()
        rs_ctx.check_error()
        g1, i1 = gold_ag()
        g2, mag2, i2 = gold_rs()
        verdicts = {}
        for prod in list(ag_ctxs):
            m1 = broke.get(prod) or _mismatch(c1s[prod] if i1 is None else c1s[prod][i1], g1, g1.abs())
            if not _all_agree(env, m1 == ""):
                verdicts[prod] = m1 or "failed on a peer rank"
        bad = []
        for prod, m1 in verdicts.items():
            if len(verdicts) < len(ag_ctxs) and prod != keep:
                notes.append(f"producer {prod} dropped at {tag}: {m1}")
                del ag_ctxs[prod]
            else:
                bad.append(f"AG-GEMM ({prod} producer) " + m1)
        if ag_ctx not in ag_ctxs.values():
            ag_ctx = next(iter(ag_ctxs.values()))
        m2 = _mismatch(c2 if i2 is None else c2[i2], g2, mag2)
        if m2:
            bad.append("GEMM-RS " + m2)
        return f"{tag}: " + "; ".join(bad) if bad else ""

    # ---- checks on fresh inputs: the timed path must produce the gold's numbers ------------------------
    n_checks = args.check_iters if args.check_iters >= 0 else (1 if w == 1 else (50 if env.has_rccl else 3))
    checks_run = 0
    if not problems:
        for it in range(n_checks):
            if it:
                a1.copy_(fresh((Mr, K)))
                a2.copy_(fresh((M, K // w)))
            try:
                msg = check_once(f"check {it}")
            except Exception as exc:  # noqa: BLE001
                msg = f"check {it}: fused path failed: {type(exc).__name__}: {exc}"
            if not _all_agree(env, msg == ""):
                problems.append(msg or "a peer rank failed its check")
                break
            checks_run += 1

    fused_ok = not problems

    def step():
        fused_ag()
        fused_rs()

    def base_step():
        torch_ag_gemm()
        torch_gemm_rs()

    total_us = None
    med = {}
    producer, producer_rule = (args.producer or "kernel"), "only one producer available"
    if ag_ctxs and producer not in ag_ctxs:          # (the checks dropped it)
        producer = next(iter(ag_ctxs))
    baseline_on = (not args.no_baseline) and (env.has_rccl or w == 1)
    if fused_ok:
        # Clock settle (setup, untimed): the first back-to-back full-chip GEMMs after an idle period run
        # through a power-management transient (kernel times swing 360 -> 500 -> 360 us over ~30 ms in
        # rocprofv3 traces); drive the chip to its steady state before anything is timed.
        settle_t0 = time.perf_counter()
        while time.perf_counter() - settle_t0 < 0.25:
            for _ in range(10):
                step()
            torch.cuda.synchronize()
        # ---- per-op medians, fused and torch interleaved -------------------------------------------------
        fns = {f"ag_gemm[{prod}]": fused_ag_with(prod) for prod in ag_ctxs}
        fns["gemm_rs"] = fused_rs
        if baseline_on:
            fns["torch_ag_gemm"] = torch_ag_gemm
            fns["torch_gemm_rs"] = torch_gemm_rs
        med = _interleaved(env, fns, args.rounds, max(5, args.steps // 5))
        if len(ag_ctxs) > 1:
            # every rank computed the same medians (max over ranks), so every rank picks the same producer
            producer = min(ag_ctxs, key=lambda prod: (med[f"ag_gemm[{prod}]"], prod != "kernel"))
            producer_rule = "lower median AG-GEMM time of this run (kernel on a tie)"
            ag_ctx = ag_ctxs[producer]
        # ---- the headline: W untimed steps, then exactly K timed steps of the pair ----------------------------
        total_us = _time_steps(env, step, args.steps, args.warmup)
        # The timed calls reuse the parity buffers many times over: check the LAST state too, so a number
        # is only reported for a path that was still producing the gold's values at the end.
        try:
            msg = check_once("check after the timed loop", keep=producer)
        except Exception as exc:  # noqa: BLE001
            msg = f"fused path failed during / after the timed loop: {type(exc).__name__}: {exc}"
        if not _all_agree(env, msg == ""):
            problems.append(msg or "a peer rank failed its check after the timed loop")
            fused_ok = False
    elif baseline_on:
        for _ in range(50):
            base_step()
        torch.cuda.synchronize()
        med = _interleaved(env, {"torch_ag_gemm": torch_ag_gemm, "torch_gemm_rs": torch_gemm_rs}, args.rounds,
                           max(5, args.steps // 5))
    base_us = _time_steps(env, base_step, max(5, args.steps // 2), 3) if baseline_on else None
    ag_us = med.get(f"ag_gemm[{producer}]") if fused_ok else None
    rs_us = med.get("gemm_rs") if fused_ok else None
    base_ag_us, base_rs_us = med.get("torch_ag_gemm"), med.get("torch_gemm_rs")

    # ---- the reference's published shapes (extra fields) ------------------------------------------------------
    extra_shapes = []
    if fused_ok and args.shapes == "baseline":
        extra_shapes = _baseline_shapes(env, ctas, producer if w > 1 else "kernel", baseline_on, args)

    if r == 0:
        from triton_distributed_gh_amd.utils import perf_model, topo

        flops = 2.0 * (2.0 * M * N * K) / w  # per GPU per step
        rnd = lambda x: None if x is None else round(x, 2)  # noqa: E731
        ratio = lambda a, b: round(a / b, 4) if (a and b) else None  # noqa: E731
        if not fused_ok:
            timed_path = "none (fused path rejected; exit status 1)"
        elif w == 1:
            timed_path = "native gfx950 GEMM kernels (TP=1: no collective, AG-GEMM and GEMM-RS are plain GEMMs)"
        else:
            timed_path = "fused HIP kernels (AG-GEMM and GEMM-RS, one persistent launch each)"
        try:
            ndev = torch.cuda.device_count()
            links = [[("self" if i == j else ("p2p" if topo.can_access_peer(i, j) else "none")) for j in range(ndev)]
                     for i in range(min(ndev, 8))]
            topology = {"devices": ndev, "fullmesh_xgmi": bool(topo.has_fullmesh_xgmi(max(w, 1))) if ndev > 1 else None,
                        "peer_access": links, "xgmi_gbps_per_link_per_dir": topo.XGMI_LINK_GBPS_PER_DIR,
                        "egress_gbps_at_this_tp": topo.get_intranode_max_speed_gbps(w) if w > 1 else 0.0}
        except Exception as exc:  # noqa: BLE001 - informational only
            topology = {"error": f"{type(exc).__name__}: {exc}"}
        us = lambda d: {k.replace("_ms", "_us"): round(v * 1e3, 1) for k, v in d.items()}  # noqa: E731
        model = {
            # parts and total of each op as the perf model sees them (utils/perf_model.py): MFMA-bound GEMM time,
            # link-bound collective time over the xGMI links this TP degree can use, and what cannot overlap
            "ag_gemm": us(perf_model.estimate_ag_gemm_time_ms(M, N // w, K, w)),
            "gemm_rs": us(perf_model.estimate_gemm_rs_time_ms(M, N, K // w, w)),
            "gemm_sol_us_per_op": rnd(perf_model.estimate_gemm_sol_time_ms(M, N // w, K) * 1e3),
        }
        out = {
            "metric": "intranode AG-GEMM + GEMM-RS latency (us), M=4096 N=K=8192 bf16",
            "value": rnd(total_us) if fused_ok else None,
            "unit": "us",
            "n_gpus": w,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(total_us / 1e3, 5) if fused_ok else None,
            "higher_is_better": False,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "bf16",
            "data": "synthetic (uniform random activations, random-init weights)",
            "config": {"model": "TP MLP-style pair (AG-GEMM then GEMM-RS)", "M": M, "N": N, "K": K,
                       "global_batch": M, "seq_len": 1, "parallelism": f"tp{w}"},
            # medians of `rounds` interleaved bursts (fused, torch, fused, torch, ...), max over ranks per burst
            "ag_gemm_us": rnd(ag_us),
            "gemm_rs_us": rnd(rs_us),
            "ag_gemm_us_kernel": rnd(med.get("ag_gemm[kernel]")) if fused_ok else None,
            "ag_gemm_us_sdma": rnd(med.get("ag_gemm[copy_engine]")) if fused_ok else None,
            "torch_ag_gemm_us": rnd(base_ag_us),
            "torch_gemm_rs_us": rnd(base_rs_us),
            "speedup_ag_gemm": ratio(base_ag_us, ag_us),
            "speedup_gemm_rs": ratio(base_rs_us, rs_us),
            # pair speed-up from the per-op medians; the torch pair timed back to back is reported next to it
            "speedup_vs_torch_rccl": ratio((base_ag_us or 0) + (base_rs_us or 0), (ag_us or 0) + (rs_us or 0))
            if (ag_us and rs_us and base_ag_us and base_rs_us) else None,
            "torch_rccl_pair_us": rnd(base_us),
            "interleaved_rounds": args.rounds,
            "tflops_per_gpu": round(flops / total_us / 1e6, 1) if fused_ok else None,
            "timed_path": timed_path,
            "producer": (producer if (ag_ctx is not None and w > 1) else "none"),
            "producer_rule": producer_rule if w > 1 else "TP=1: nothing to gather",
            "perf_model": model,
            "topology": topology,
            "baseline_shapes": extra_shapes,
            "heap_mode": env.heap_mode,
            "safe_publish": bool(getattr(env, "safe_publish", False)),
            "heap_selftest": [f"{m}: {d}" for m, d in getattr(env, "selftest_log", ())],
            "ranks_share_one_gpu": bool(env.shared_gpu),
            "checks_passed_before_timing": checks_run,
            "note": "; ".join(problems),
            "remarks": "; ".join(notes),
        }
        print(json.dumps(out), flush=True)
    if env.heap is not None:
        finalize_distributed()
    elif dist.is_initialized():
        dist.destroy_process_group()
    if not fused_ok:
        sys.exit(1)


if __name__ == "__main__":
    main()
