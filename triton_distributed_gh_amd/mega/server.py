"""Demo chat server / client on top of the mega-kernel decode model.

Reference: python/triton_dist/mega_triton_kernel/test/models/model_server.py (TCP JSON server on
localhost:9999 run by rank 0, prompt token ids broadcast to the other ranks) and chat.py (client).
Protocol: one JSON object per line, ``{"prompt": str, "max_new_tokens": int}`` ->
``{"text": str, "tokens": [...], "ms_per_token": float}``.

    python -m triton_distributed_gh_amd.mega.server serve --model Qwen/Qwen3-8B [--backend mega|torch]
    python -m triton_distributed_gh_amd.mega.server chat "hello"
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import socketserver
import time

import torch
import torch.distributed as dist

HOST, PORT = "127.0.0.1", 9999


class Generator:
    """Prefill token by token through the decode graph (the mega kernel is decode-shaped), then
    greedy decode."""

    def __init__(self, model_name: str, backend: str = "mega", max_length: int = 1024, num_layers=None):
        os.environ.setdefault("RANDOM_PARAMS", "1")
        from ..models import AutoLLM, AutoTokenizer, KV_Cache, ModelConfig
        from ..utils import initialize_distributed
        from .qwen3 import Qwen3Model

        self.env = initialize_distributed(seed=0)
        cfg = ModelConfig(model_name=model_name, max_length=max_length, dtype=torch.bfloat16, rank=self.env.rank,
                          world_size=self.env.world_size, num_layers=num_layers)
        self.model = AutoLLM.from_pretrained(cfg, self.env.tp_group, env=self.env)
        self.tokenizer = AutoTokenizer.from_pretrained(cfg)
        self.backend = backend
        self.max_length = max_length
        if backend == "mega":
            self.mega = Qwen3Model(self.model, 1, max_length, env=self.env)
        else:
            m = self.model
            self.kv = KV_Cache(m.num_layers, 1, max_length, m.num_key_value_heads, m.head_dim, m.dtype, m.world_size)

    def _step(self, tok: torch.Tensor) -> torch.Tensor:
        if self.backend == "mega":
            return self.mega.mega_forward(tok)
        logits = self.model.inference(tok, self.kv.get_kv_len()[:, None].long(), self.kv)
        self.kv.inc_offset()
        return logits

    def _sync_prompt(self, ids):
        """Rank 0 owns the prompt; everyone else receives it (reference: dist.broadcast of token ids)."""
        if self.env.world_size == 1:
            return ids
        n = torch.tensor([len(ids) if ids is not None else 0], dtype=torch.long)
        dist.broadcast(n, src=0, group=self.env.cpu_group)
        buf = torch.tensor(ids, dtype=torch.long) if self.env.rank == 0 else torch.zeros(int(n), dtype=torch.long)
        dist.broadcast(buf, src=0, group=self.env.cpu_group)
        return buf.tolist()

    @torch.inference_mode()
    def generate(self, prompt_ids, max_new_tokens: int = 32):
        ids = self._sync_prompt(prompt_ids)
        vocab = self.model.embed_tokens.shape[0]
        ids = [int(i) % vocab for i in ids][: self.max_length - max_new_tokens - 1] or [0]
        if self.backend == "mega":
            self.mega.set_kv_len(0)
        else:
            self.kv.clear()
        dev = self.model.device
        logits = None
        for t in ids:
            logits = self._step(torch.tensor([[t]], device=dev))
        out = []
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(max_new_tokens):
            nxt = int(logits[0, -1].argmax())
            out.append(nxt)
            logits = self._step(torch.tensor([[nxt]], device=dev))
        torch.cuda.synchronize()
        return out, (time.perf_counter() - t0) * 1e3 / max(max_new_tokens, 1)


def serve(args):
    gen = Generator(args.model, args.backend, args.max_length, args.num_layers)
    if gen.env.rank != 0:
        while True:                      # follower ranks: take part in every generation rank 0 starts
            gen.generate(None, args.max_new_tokens)

    class Handler(socketserver.StreamRequestHandler):

        def handle(self):
            for line in self.rfile:
                req = json.loads(line)
                ids = gen.tokenizer.encode(req["prompt"])
                toks, ms = gen.generate(ids, int(req.get("max_new_tokens", args.max_new_tokens)))
                resp = {"text": gen.tokenizer.decode(toks), "tokens": toks, "ms_per_token": round(ms, 3)}
                self.wfile.write((json.dumps(resp) + "\n").encode())
                self.wfile.flush()

    with socketserver.TCPServer((HOST, args.port), Handler) as srv:
        print(f"serving {args.model} ({args.backend}) on {HOST}:{args.port}", flush=True)
        srv.serve_forever()


def chat(args):
    with socket.create_connection((HOST, args.port)) as s:
        f = s.makefile("rw")
        prompts = [args.prompt] if args.prompt else iter(lambda: input("> "), "")
        for p in prompts:
            f.write(json.dumps({"prompt": p, "max_new_tokens": args.max_new_tokens}) + "\n")
            f.flush()
            r = json.loads(f.readline())
            print(f"{r['text']}   [{r['ms_per_token']} ms/token]")


def main():
    ap = argparse.ArgumentParser()
    sub = ap.add_subparsers(dest="cmd", required=True)
    for name in ("serve", "chat"):
        p = sub.add_parser(name)
        p.add_argument("--port", type=int, default=PORT)
        p.add_argument("--max_new_tokens", type=int, default=32)
    sv = sub.choices["serve"]
    sv.add_argument("--model", default="Qwen/Qwen3-8B")
    sv.add_argument("--backend", default="mega", choices=["mega", "torch"])
    sv.add_argument("--max_length", type=int, default=1024)
    sv.add_argument("--num_layers", type=int, default=None)
    sub.choices["chat"].add_argument("prompt", nargs="?", default=None)
    args = ap.parse_args()
    (serve if args.cmd == "serve" else chat)(args)


if __name__ == "__main__":
    main()
