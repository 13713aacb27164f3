#!/usr/bin/env python3
"""Kernel-level microbench for the HIP attention path (development tool, not the judged bench).

Times nvh_paged_decode / nvh_decode_step / nvh_prefill_varlen with HIP events on torch's current stream,
cycling over `--layers` distinct KV caches so that no launch re-reads what the previous one left in the
256 MiB Infinity Cache, and prints achieved algorithmic GB/s (decode) or TFLOP/s (prefill).

    python tools/microbench.py decode --batch 32 --ctx 1536 --heads 14 --kv-heads 2 --head-dim 64
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-learn_amd"))
from nanovllm_hip import ops  # noqa: E402


def decode_bytes(ctxs, h, kvh, d, bs):
    """Algorithmic bytes of one decode launch."""
    ctxs = np.asarray(ctxs)
    return int((2 * ctxs * kvh * d * 2).sum() + 2 * len(ctxs) * h * d * 2 + 4 * (np.ceil(ctxs / bs).sum() + len(ctxs)))


def make_decode(args, dev):
    b, h, kvh, d, bs = args.batch, args.heads, args.kv_heads, args.head_dim, args.block_size
    rng = np.random.default_rng(0)
    if args.ctx_lo:
        ctxs = rng.integers(args.ctx_lo, args.ctx + 1, size=b)
    else:
        ctxs = np.full(b, args.ctx)
    need = (ctxs + bs - 1) // bs
    width = args.width or int(need.max())
    nb = int(need.sum()) + 1
    n_caches = args.caches or args.layers                   # (--caches n < layers: the calls cycle over n caches — K/V re-read from the Infinity Cache)
    caches = [torch.randn(2, nb, bs, kvh, d, device=dev, dtype=torch.bfloat16) for _ in range(n_caches)]
    caches = [caches[l % n_caches] for l in range(args.layers)]
    bt = np.zeros((b, width), np.int32)
    ids = iter(rng.permutation(nb).tolist())
    for i in range(b):
        for j in range(need[i]):
            bt[i, j] = next(ids)
    q = torch.randn(b, h, d, device=dev, dtype=torch.bfloat16)
    return q, caches, torch.from_numpy(ctxs.astype(np.int32)).to(dev), torch.from_numpy(bt).to(dev), ctxs


def time_loop(fn, n_layers, iters, warmup):
    for _ in range(warmup):
        for l in range(n_layers):
            fn(l)
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        for l in range(n_layers):
            fn(l)
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) * 1e3 / (iters * n_layers)       # us per call


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("mode", choices=["decode", "prefill"])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--ctx", type=int, default=1536)
    ap.add_argument("--ctx-lo", type=int, default=0, help="if set, context lengths uniform in [ctx-lo, ctx]")
    ap.add_argument("--heads", type=int, default=14)
    ap.add_argument("--kv-heads", type=int, default=2)
    ap.add_argument("--head-dim", type=int, default=64)
    ap.add_argument("--block-size", type=int, default=256)
    ap.add_argument("--width", type=int, default=0, help="block-table width (0 = tight; 16 = graph-replay shape)")
    ap.add_argument("--layers", type=int, default=24)
    ap.add_argument("--caches", type=int, default=0, help="decode: distinct K/V caches the `layers` calls cycle over (0 = one per call)")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--graph", action="store_true", help="time a HIP graph of `layers` back-to-back launches")
    ap.add_argument("--fused", action="store_true", help="decode: time nvh_decode_step (store + attend)")
    ap.add_argument("--seq", type=int, default=1024, help="prefill: sequence length")
    ap.add_argument("--paged", action="store_true", help="prefill: K / V through the paged cache and a shuffled block table (prefix-cached / chunked prefill)")
    ap.add_argument("--q-len", type=int, default=0, help="prefill --paged: new tokens per sequence (the last q-len of --seq; 0 = all)")
    ap.add_argument("--pv", default="auto", choices=["auto", "fp16", "exact"], help="prefill: P V form (ops.flash_attn_varlen_func pv_fp16 = None / True / False)")
    ap.add_argument("--variant", default=None, help="decode: chunked | chunked_p64 | chunked_p128 | chunked_p256 | split_mfma | split_valu (nvh_paged_decode_variant); prefill: auto | tiled | short | tiled_f16v")
    ap.add_argument("--waves", type=int, default=0, help="decode (chunked, D=64): 4 or 8 waves; prefill short kernel: 8 or 16")
    ap.add_argument("--chunks", type=int, default=0, help="decode (chunked): workgroups per (sequence, kv head)")
    args = ap.parse_args()
    dev = "cuda"
    torch.manual_seed(0)
    if args.mode == "decode":
        q, caches, cl, bt, ctxs = make_decode(args, dev)
        b, h, kvh, d = args.batch, args.heads, args.kv_heads, args.head_dim
        out = torch.empty(b, h, d, device=dev, dtype=torch.bfloat16)
        ops.reserve_workspace(dev, ops.decode_workspace_bytes(b, h, d, bt.shape[1], args.block_size))
        knew = torch.randn(b, kvh, d, device=dev, dtype=torch.bfloat16)
        slots = torch.tensor([int(bt[i, (c - 1) // args.block_size]) * args.block_size + (c - 1) % args.block_size
                              for i, c in enumerate(ctxs)], dtype=torch.int32, device=dev)

        def call(l):
            if args.fused:
                ops.decode_step(q, knew, knew, caches[l][0], caches[l][1], slots, cl, bt, out=out)
            else:
                ops.flash_attn_with_kvcache(q, caches[l][0], caches[l][1], cl, bt, out=out, variant=args.variant, waves=args.waves, chunks=args.chunks)

        if args.graph:
            for l in range(args.layers):
                call(l)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for l in range(args.layers):
                    call(l)
            us = time_loop(lambda l: g.replay(), 1, args.iters, args.warmup) / args.layers
        else:
            us = time_loop(call, args.layers, args.iters, args.warmup)
        nbytes = decode_bytes(ctxs, h, kvh, d, args.block_size)
        print(json.dumps({"mode": "decode" + ("_fused" if args.fused else ""), "graph": args.graph, "batch": b, "ctx_mean": float(ctxs.mean()),
                          "shape": [h, kvh, d], "width": int(bt.shape[1]), "us_per_call": round(us, 2),
                          "alg_MB": round(nbytes / 1e6, 2), "GBps": round(nbytes / us / 1e3, 1),
                          "frac_of_8TBps": round(nbytes / us / 1e3 / 8000, 3)}))
    else:
        b, s, h, kvh, d = args.batch, args.seq, args.heads, args.kv_heads, args.head_dim
        t = b * s
        qkv = torch.randn(t, (h + 2 * kvh) * d, device=dev, dtype=torch.bfloat16)
        q = qkv[:, :h * d].view(t, h, d)
        k = qkv[:, h * d:(h + kvh) * d].view(t, kvh, d)
        v = qkv[:, (h + kvh) * d:].view(t, kvh, d)
        cu = torch.arange(0, t + 1, s, dtype=torch.int32, device=dev)
        if args.variant == "tiled_f16v":
            v = v.to(torch.float16)                              # (converted once, outside the timed loop: what a producer-side conversion would hand over)
        if args.paged:
            # prefix-cached / chunked form (attention.py:90-96 with block_table): K / V of every sequence live in the paged cache behind a shuffled block
            # table; q holds the LAST --q-len tokens of each sequence (0 = all of them), bottom-right aligned causal mask
            bs = args.block_size
            nblk = (s + bs - 1) // bs
            perm = torch.randperm(b * nblk, device=dev).int().view(b, nblk)
            kc = torch.zeros(b * nblk, bs, kvh, d, device=dev, dtype=torch.bfloat16)
            vc = torch.zeros_like(kc)
            for i in range(b):
                for j in range(nblk):
                    n = min(bs, s - j * bs)
                    kc[perm[i, j], :n] = k[i * s + j * bs: i * s + j * bs + n]
                    vc[perm[i, j], :n] = v[i * s + j * bs: i * s + j * bs + n]
            ql = args.q_len or s
            rows = (torch.arange(b, device=dev)[:, None] * s + torch.arange(s - ql, s, device=dev)[None, :]).reshape(-1)
            qp = qkv[rows][:, :h * d].contiguous().view(b * ql, h, d)
            cuq = torch.arange(0, b * ql + 1, ql, dtype=torch.int32, device=dev)
            us = time_loop(lambda l: ops.flash_attn_varlen_func(qp, kc, vc, ql, cuq, s, cu, block_table=perm), 1, args.iters, args.warmup)
            flops = b * 4 * d * h * (ql * (s - ql) + ql * (ql + 1) / 2)
            print(json.dumps({"mode": "prefill_paged", "batch": b, "q_len": ql, "k_len": s, "shape": [h, kvh, d], "us_per_call": round(us, 1),
                              "TFLOPs": round(flops / us / 1e6, 1), "frac_of_2.5PF": round(flops / us / 1e6 / 2500, 4)}))
            return
        pv = None if args.variant or args.waves else {"auto": None, "fp16": True, "exact": False}[args.pv]
        call = lambda l=0: ops.flash_attn_varlen_func(q, k, v, s, cu, s, cu, kernel=args.variant, short_waves=args.waves, pv_fp16=pv)
        if args.graph:                                           # 8 calls per replayed graph: no host time between the launches
            call(); torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                for _ in range(8):
                    call()
            us = time_loop(lambda l: g.replay(), 1, args.iters, args.warmup) / 8
        else:
            us = time_loop(call, 1, args.iters, args.warmup)
        flops = b * 4 * d * h * s * (s + 1) / 2
        short = d == 64 and 64 < s <= 128 and b * kvh >= 128
        form = args.variant or ("fp16 P V (guarded, conversion inside the call)" if pv is True or (pv is None and not args.waves and (s >= ops.PV16_MIN_KEYS or short)) else "bf16 hi + lo")
        print(json.dumps({"mode": "prefill", "graph": args.graph, "batch": b, "seq": s, "shape": [h, kvh, d], "form": form, "us_per_call": round(us, 1),
                          "TFLOPs": round(flops / us / 1e6, 1), "frac_of_2.5PF": round(flops / us / 1e6 / 2500, 4)}))


if __name__ == "__main__":
    main()
