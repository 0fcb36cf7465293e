"""Shader clock and socket power while one GEMM kernel runs back to back (rocm-smi sampled from a thread)."""
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, __file__.rsplit("/scripts/", 1)[0])
from triton_distributed_gh_amd.ops import gemm as G  # noqa: E402


def sample(stop, out):
    while not stop.is_set():
        try:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5)
            out.append((time.time(), r.stdout.strip()))
        except Exception as e:  # noqa: BLE001
            out.append((time.time(), f"ERR {e}"))
        time.sleep(0.05)


def run(name, fn, seconds=3.0):
    fn()
    torch.cuda.synchronize()
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    th.start()
    t0 = time.time()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    print(f"== {name}: {e0.elapsed_time(e1) * 1000 / n:.1f} us per call over {n} calls")
    import json

    sclk, pw = [], []
    for _, s in out:
        try:
            d = json.loads(s)
            c = d.get("card0", {})
            for k, v in c.items():
                if "sclk" in k.lower() and "level" not in k.lower():
                    sclk.append(v)
                if "power" in k.lower():
                    pw.append(v)
        except Exception:  # noqa: BLE001
            pass
    print("  samples", len(out), "sclk", sclk[:3], "...", sclk[-3:], "power", pw[:3], "...", pw[-3:])
    if out:
        print("  raw last:", out[-1][1][:600])


M, N, K = 4096, 8192, 8192
a = (torch.rand(M, K, device="cuda") * 2 - 1).bfloat16()
b = (torch.rand(N, K, device="cuda") * 2 - 1).bfloat16()
c = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
run("native", lambda: G.gemm_nt(a, b, out=c))
run("library", lambda: torch.matmul(a, b.t(), out=c))
run("native again", lambda: G.gemm_nt(a, b, out=c))
