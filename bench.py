#!/usr/bin/env python3
"""bench.py — decode tok/s of the `hip` attention backend in a bench_my.py-shaped run (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): Qwen2-0.5B shapes, random-init bf16 weights, bs=32 sequences of
1024 random prompt tokens (`random.seed(0)`, `randint(0, 10000)`, bench_my.py:27-40), greedy decode with
ignore_eos.  A "step" is one decode step of the whole batch through all 24 layers: per layer one
store_kvcache + paged-decode attention call through the C ABI (the hot path) plus the PyTorch-ROCm model
body around it, replayed from a HIP graph with device-resident metadata.  Prefill runs before the timed
region (it is reported separately, and the bench_my-style figure that includes it is in `bench_my_tok_s`).
The K timed steps start at context 1025 as bench_my's decode phase does (K = 1024 covers 1025 -> 2048).

Output: ONE JSON line on rank 0 (contract in the task statement) with these extra objects:
  roofline      the decode attention op (nvh_paged_decode: one chunked split-KV launch, combine included) at the mean context of the
                timed window, timed live with HIP events on the launching stream over a graph of per-layer calls
                on distinct caches; achieved = algorithmic bytes / average time per call; traffic = HBM bytes per call from the
                committed rocprofv3 PMC passes (profiles/r03_pmc_decode_traffic.json), taken at the profiled context nearest to
                the measured one and scaled by the ratio of algorithmic bytes (the line says which context it came from).
  prefill       the varlen prefill attention op (nvh_prefill_varlen) timed the same way at BASELINE config 5 (one scheduler
                batch: 128 sequences x 128 tokens) and at S = 1024 (16 sequences: one prefill batch of config 2), with flops,
                bytes, bound = the slower of HBM and MFMA at their peaks, and the fraction of that bound achieved.
  step_floor    the decode step with every attention call at its floor as a launch of its own (fixed 3.75 us + bytes at 6.29 TB/s) and every
                other launch as measured: what SURVEY section 8's rows can move of `decode_step_roofline`.
  full_window   one more replay of bench_my's whole decode window (out = in tokens: contexts in + 1 -> 2 in) after the timed region:
                tok/s, ms per step, the step's roofline fraction, and `bench_my_tok_s` = the figure bench_my.py:27-40 defines (prefill inside).
  attention_sweep  the decode attention call at the other BASELINE configs' shapes (config 3; config 4 per rank and at tp = 1; Qwen3-0.6B).
  cpu_baseline  the CPU port of the reference's sdpa.math attention (oracle/sdpa_math_cpu.py, "port"), timed on this host on
                bounded samples: the bs=32 decode call of this workload, and (`config1`) BASELINE config 1's shape (bs=1,
                in=out=512: one 512-token prefill call and decode calls over contexts 513..1024); rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time
from random import randint, seed

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-learn_amd"))

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
MFMA_PEAK_TFLOPS = 2500.0       # dense bf16 MFMA peak (MI355X_MICROARCH.md; the 5 PF headline includes 2:1 sparsity)


def decode_attn_bytes(ctxs, h, kvh, d, bs=256, with_store=True):
    """Algorithmic bytes of one decode attention call, per rank."""
    ctxs = np.asarray(ctxs, dtype=np.int64)
    b = len(ctxs)
    n = int((2 * ctxs * kvh * d * 2).sum() + 2 * b * h * d * 2 + 4 * (np.ceil(ctxs / bs).sum() + b))
    if with_store:
        n += 4 * b * kvh * d * 2
    return n


@torch.inference_mode()
def attention_leg(cfg, tp, batch, ctx, layers, iters=30, shape=None):
    """Time nvh_decode_step alone: a HIP graph of `layers` calls on distinct KV caches (no Infinity-Cache reuse
    between calls), replayed `iters` times between two HIP events on the launching stream.  shape = (h, kvh, d) overrides the
    model's per-rank head shape (the attention_sweep of the other BASELINE configs)."""
    from nanovllm_hip import ops
    dev = torch.device("cuda", torch.cuda.current_device())
    from nanovllm_hip.models.qwen import tp_partition
    rank = dist.get_rank() if dist.is_initialized() else 0
    _, h, _, kvh = tp_partition(cfg.num_attention_heads, cfg.num_key_value_heads, tp, rank)
    d, bs = cfg.head_dim, cfg.kvcache_block_size
    if shape is not None:
        h, kvh, d = shape
    nblk = (ctx + bs - 1) // bs
    nb = batch * nblk + 1
    gen = torch.Generator(device="cpu").manual_seed(0)
    caches = [torch.randn(2, nb, bs, kvh, d, device=dev, dtype=torch.bfloat16) for _ in range(layers)]
    bt = torch.randperm(nb - 1, generator=gen)[: batch * nblk].view(batch, nblk).int().to(dev)
    cl = torch.full((batch,), ctx, dtype=torch.int32, device=dev)
    slots = (bt[:, (ctx - 1) // bs].long() * bs + (ctx - 1) % bs).int()
    qkv = torch.randn(batch, (h + 2 * kvh) * d, device=dev, dtype=torch.bfloat16)
    q, k, v = qkv[:, :h * d].view(batch, h, d), qkv[:, h * d:(h + kvh) * d].view(batch, kvh, d), qkv[:, (h + kvh) * d:].view(batch, kvh, d)
    out = torch.empty(batch, h, d, device=dev, dtype=torch.bfloat16)
    ops.reserve_workspace(dev, ops.decode_workspace_bytes(batch, h, d, nblk, bs))

    reps = max(1, -(-24 // layers))                # at least 24 calls per graph: the ~7 us between two replays is amortised alike in every leg
    n_calls = reps * layers

    def calls():                                   # exactly what a decoder layer launches for attention at decode
        for _ in range(reps):                      # (the K/V store rides in the qkv projection's epilogue, nvh_linear_small_m_ex)
            for c in caches:                       # (the caches of one cycle exceed the 256 MiB Infinity Cache: no reuse between calls)
                ops.flash_attn_with_kvcache(q, c[0], c[1], cl, bt, out=out)

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        calls()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        calls()
    for _ in range(3):
        graph.replay()
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        graph.replay()
    end.record()
    torch.cuda.synchronize()
    us = start.elapsed_time(end) * 1e3 / (iters * n_calls)
    nbytes = decode_attn_bytes([ctx] * batch, h, kvh, d, bs, with_store=False)
    del graph, caches
    torch.cuda.empty_cache()
    return us, nbytes, (h, kvh, d)


# floor of one decode attention call as its own launch: launch boundary + dispatch ramp + first byte, then the K/V
# stream at the rate the part sustains; the hand-off tail is NOT in it (a free hand-off)
ATTN_FIXED_FLOOR_US = 1.25 + 0.6 + 1.9
HBM_SUSTAINED_GBPS = 6290.0     # float4 copy, MI355X_MICROARCH.md


def attention_sweep(cfg):
    """The decode attention call at the other BASELINE configs (not bench lines of their own; here so that the driver's record holds
    them): config 3 (B = 64, ctx 3072: the window mean of 2049 -> 4096), config 4's per-rank shape at tp = 4 (7 / 1 / 128) and its
    tp = 1 shape (28 / 4 / 128) at B = 32, ctx 1536, and the reference's default model's head shape (Qwen3-0.6B: 16 / 8 / 128)."""
    cases = [("config 3: Qwen2-0.5B bs=64 in=out=2048, window-mean context", 64, 3072, (14, 2, 64), 6),
             ("config 4 per rank at tp=4: Qwen2-7B heads 28/4/128 -> 7/1/128, bs=32", 32, 1536, (7, 1, 128), 12),
             ("config 4 at tp=1: Qwen2-7B heads 28/4/128, bs=32", 32, 1536, (28, 4, 128), 6),
             ("reference default model Qwen3-0.6B heads 16/8/128, bs=32", 32, 1536, (16, 8, 128), 4)]
    out = []
    for name, b, ctx, shape, layers in cases:
        us, nbytes, _ = attention_leg(cfg, 1, b, ctx, layers, iters=12, shape=shape)
        out.append({"workload": name, "batch": b, "ctx": ctx, "shape": list(shape), "us": round(us, 2), "bytes": int(nbytes),
                    "GBps": round(nbytes / us / 1e3, 1), "frac": round(nbytes / us / 1e3 / HBM_PEAK_GBPS, 4)})
    return out


def pmc_traffic(cfg, tp, batch, ctx, attn_bytes):
    """HBM bytes per attention call from the committed rocprofv3 PMC passes (profiles/r03_pmc_decode_traffic.json: one
    --pmc FETCH_SIZE and one --pmc WRITE_SIZE pass of tools/microbench.py per profiled context; FETCH_SIZE doubled per the
    gfx950 correction, + WRITE_SIZE).  The record of the profiled context nearest to `ctx` is scaled by the ratio of
    algorithmic bytes (measured / algorithmic is 1.04-1.06 at every profiled context: the kernel reads each K/V byte once).
    Returns (bytes, note) or (None, reason)."""
    for name in ("r03_pmc_decode_traffic.json", "r02_pmc_decode_traffic.json", "r01_pmc_decode_traffic.json"):
        try:
            p = json.load(open(os.path.join(ROOT, "profiles", name)))
            break
        except OSError:
            p = None
    if p is None:
        return None, "no committed PMC record"
    recs = p["records"] if "records" in p else [p]
    same = [r for r in recs if tp == 1 and r["workload"]["batch"] == batch and r["workload"]["heads"] == cfg.num_attention_heads and
            r["workload"]["kv_heads"] == cfg.num_key_value_heads and r["workload"]["head_dim"] == cfg.head_dim]
    if not same:
        return None, "no PMC record for this shape"
    r = min(same, key=lambda r: abs(r["workload"]["ctx"] - ctx))
    ratio = r["hbm_bytes_per_attention_call"] / r["algorithmic_bytes_per_attention_call"]
    note = (f"rocprofv3 PMC (FETCH_SIZE x2 + WRITE_SIZE) at ctx {r['workload']['ctx']}: {r['hbm_bytes_per_attention_call']} B per call = "
            f"{ratio:.3f} x algorithmic" + ("" if r["workload"]["ctx"] == ctx else f"; scaled to ctx {ctx} by algorithmic bytes"))
    return int(round(ratio * attn_bytes)), note


@torch.inference_mode()
def prefill_leg(cfg, tp, batch, seq, buffers=8, iters=6, pv_fp16=None):
    """Time the prefill attention call (ops.flash_attn_varlen_func as the Attention module calls it; pv_fp16=None is the module's default rule: P V on the
    fp16 pipe from 512 keys on (and in the short-sequence kernel), conversion of V and its range guard inside the timed call; False = P as bf16 hi + lo everywhere) alone on `batch` sequences of `seq` tokens (q / k / v strided views of a fused projection
    output, as the model hands them over), cycling over distinct inputs (more than the 256 MiB Infinity Cache in total) between
    two HIP events on the launching stream; the calls are replayed from a HIP graph (no host time between launches)."""
    from nanovllm_hip import ops
    from nanovllm_hip.models.qwen import tp_partition
    rank = dist.get_rank() if dist.is_initialized() else 0
    _, h, _, kvh = tp_partition(cfg.num_attention_heads, cfg.num_key_value_heads, tp, rank)
    d = cfg.head_dim
    t = batch * seq
    dev = torch.device("cuda", torch.cuda.current_device())
    qkvs = [torch.randn(t, (h + 2 * kvh) * d, device=dev, dtype=torch.bfloat16) for _ in range(buffers)]
    cu = torch.arange(0, t + 1, seq, dtype=torch.int32, device=dev)

    def call(x):
        q, k, v = x[:, :h * d].view(t, h, d), x[:, h * d:(h + kvh) * d].view(t, kvh, d), x[:, (h + kvh) * d:].view(t, kvh, d)
        return ops.flash_attn_varlen_func(q, k, v, seq, cu, seq, cu, pv_fp16=pv_fp16)      # (fp16 form: the conversion of v is inside the timed call)

    for x in qkvs[:2]:
        call(x)
    torch.cuda.synchronize()
    # the calls of one pass over the buffers are captured into ONE HIP graph and replayed: at 20 us per call an eager loop measures the host
    # (ctypes + allocator), not the launch; the events bracket the replays on the launching stream
    # (the output tensor is dropped after each call, so the captured calls reuse one output block of the graph's pool — as the eager allocator does)
    cold, warm = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(cold):
        for x in qkvs:
            call(x)
    with torch.cuda.graph(warm):
        for _ in qkvs:
            call(qkvs[0])
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    cold.replay()
    torch.cuda.synchronize()
    start.record()
    for _ in range(iters):
        cold.replay()
    end.record()
    torch.cuda.synchronize()
    us = start.elapsed_time(end) * 1e3 / (iters * buffers)
    # the same launch on ONE input, re-used: q / k / v resident in the 256 MiB Infinity Cache, as they are right after the qkv
    # projection that produces them in the model (reported beside the cold figure; `frac` stays on the cold one)
    warm.replay()
    torch.cuda.synchronize()
    start.record()
    for _ in range(iters):
        warm.replay()
    end.record()
    torch.cuda.synchronize()
    us_warm = start.elapsed_time(end) * 1e3 / (iters * buffers)
    del cold, warm
    flops = batch * 4 * d * h * seq * (seq + 1) / 2                    # QK^T + PV over the causal triangle incl. the diagonal
    nbytes = t * (2 * h + 2 * kvh) * d * 2                              # q in, o out, k and v in
    t_hbm, t_mfma = nbytes / (HBM_PEAK_GBPS * 1e3), flops / (MFMA_PEAK_TFLOPS * 1e6)     # us at the two peaks
    bound = "hbm" if t_hbm >= t_mfma else "mfma"
    short = d == 64 and 64 < seq <= 128 and batch * kvh >= 128
    fp16_form = (pv_fp16 is None and (seq >= ops.PV16_MIN_KEYS or short)) or pv_fp16 is True
    form = ("P V on the fp16 pipe behind a range guard (nvh_prefill_varlen_pv16: conversion of v + attention, both inside the timed call — inside the one "
            "kernel for the short-sequence shapes; 4.5e-4 abs on the reference goldens)" if fp16_form else "P as bf16 hi + lo (nvh_prefill_varlen; 6e-6 on the reference goldens)")
    return {"workload": f"{batch} sequences x {seq} tokens, H/KVH/D = {h}/{kvh}/{d}", "form": form, "us_per_launch": round(us, 2),
            "us_per_launch_inputs_in_infinity_cache": round(us_warm, 2), "flops_per_launch": int(flops),
            "bytes_per_launch": int(nbytes), "achieved_TFLOPs": round(flops / us / 1e6, 1), "achieved_GBps": round(nbytes / us / 1e3, 1),
            "bound": bound, "us_at_bound": round(max(t_hbm, t_mfma), 2), "frac": round(max(t_hbm, t_mfma) / us, 4),
            "frac_of_mfma_peak": round(flops / us / 1e6 / MFMA_PEAK_TFLOPS, 4), "frac_of_hbm_peak": round(nbytes / us / 1e3 / HBM_PEAK_GBPS, 4)}


def _cpu_decode_case(cfg, batch, ctx, seed=0):
    h, kvh, d, bs = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim, cfg.kvcache_block_size
    nblk = (ctx + bs - 1) // bs
    nb = batch * nblk + 1
    gen = torch.Generator().manual_seed(seed)
    kc = torch.randn(nb, bs, kvh, d, generator=gen).bfloat16()
    vc = torch.randn(nb, bs, kvh, d, generator=gen).bfloat16()
    q = torch.randn(batch, 1, h, d, generator=gen).bfloat16()
    bt = torch.randperm(nb - 1, generator=gen)[: batch * nblk].view(batch, nblk).int()
    cl = torch.full((batch,), ctx, dtype=torch.int32)
    return q, kc, vc, cl, bt


def _time_cpu(fn, budget_s, max_reps):
    fn()                                                                # warm-up
    reps, t0 = 0, time.perf_counter()
    while True:
        fn()
        reps += 1
        el = time.perf_counter() - t0
        if el >= budget_s or reps >= max_reps:
            return el / reps, reps, el


def cpu_baseline_leg(cfg, batch, ctx, budget_s=10.0):
    """Time the CPU port of the reference's sdpa.math attention (bf16 like the reference) on the host: the decode call of this
    workload (one layer, bs = `batch`), and BASELINE config 1's shape (bs = 1, in = out = 512) as a second bounded sample."""
    from oracle.sdpa_math_cpu import flash_attn_varlen_func_cpu, flash_attn_with_kvcache_cpu
    h, kvh, d = cfg.num_attention_heads, cfg.num_key_value_heads, cfg.head_dim
    layers = cfg.num_hidden_layers
    case = _cpu_decode_case(cfg, batch, ctx)
    per_call, reps, el = _time_cpu(lambda: flash_attn_with_kvcache_cpu(*case), budget_s, 96)   # ~10 s: four decode steps' worth of layer calls at most
    # ---- config 1 (bs=1, in=out=512, --attn-backend sdpa.math on CPU): attention work of one generate() = one 512-token prefill call
    # and 512 decode calls at contexts 513..1024, per layer; sampled at three contexts (~2 s each), the rest interpolated linearly
    gen = torch.Generator().manual_seed(1)
    n_in = n_out = 512
    qkv = torch.randn(n_in, (h + 2 * kvh) * d, generator=gen).bfloat16()
    qp, kp, vp = qkv[:, :h * d].view(n_in, h, d), qkv[:, h * d:(h + kvh) * d].view(n_in, kvh, d), qkv[:, (h + kvh) * d:].view(n_in, kvh, d)
    cu = torch.tensor([0, n_in], dtype=torch.int32)
    pre_s, pre_reps, _ = _time_cpu(lambda: flash_attn_varlen_func_cpu(qp, kp, vp, cu, cu), 1.5, 40)
    dec = {}
    for c in (n_in + 1, n_in + n_out // 2, n_in + n_out):
        one = _cpu_decode_case(cfg, 1, c, seed=c)
        dec[c], _, _ = _time_cpu(lambda: flash_attn_with_kvcache_cpu(*one), 1.5, 400)
    cs = sorted(dec)
    mean_dec = (dec[cs[0]] + 2 * dec[cs[1]] + dec[cs[2]]) / 4            # trapezoid over the 512 contexts
    total = layers * (pre_s + n_out * mean_dec)
    return {"value": round(batch / (per_call * layers), 2), "unit": "tok/s (attention path only: one decode step = %d layer calls)" % layers,
            "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} calls of the sdpa.math decode attention (B={batch}, ctx={ctx}, H/KVH/D={h}/{kvh}/{d}, bf16) in {el:.1f} s; "
                      f"{per_call * 1e3:.1f} ms per layer call; host cpu_count={os.cpu_count()}",
            "config1": {"value": round(n_out / total, 2), "unit": "tok/s (attention path only, bs=1 in=512 out=512, %d layers)" % layers,
                        "workload": "BASELINE config 1 shape: Qwen2-0.5B bs=1 in=out=512, sdpa.math attention on CPU",
                        "sample": f"prefill call (512 tokens) {pre_s * 1e3:.2f} ms x {pre_reps} reps; decode calls at ctx "
                                  + ", ".join(f"{c}: {dec[c] * 1e3:.3f} ms" for c in cs) + "; other contexts interpolated",
                        "cores": torch.get_num_threads(), "kind": "port"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1024)
    ap.add_argument("--warmup", type=int, default=16)
    ap.add_argument("--model", default="Qwen2-0.5B")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--input-len", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--eager", action="store_true", help="no HIP graph (debug)")
    ap.add_argument("--graph-steps", type=int, default=0, help="A/B: decode steps captured per replayed graph (0 = the session's default)")
    ap.add_argument("--kv-ahead", default=None, choices=["o_k+gu_v", "o_v+gu_k", "o_k", "gu_k"],
                    help="A/B: the o_proj / gate_up launches of a layer touch the NEXT layer's K / V cache regions (Infinity-Cache prefetch experiment, DESIGN.md 12.1)")
    ap.add_argument("--no-prefetch", action="store_true", help="A/B: the qkv / down launches do not prefetch the next small projection's weights")
    ap.add_argument("--qkv-attend", default=None, choices=["two_launches", "one_launch", "two_launches_kv_prefetch", "auto"],
                    help="A/B: how the qkv projection + decode attention front of a layer runs (nvh_qkv_rope_attend_variant)")
    ap.add_argument("--kv-prefetch-passes", type=int, default=1)
    ap.add_argument("--no-full-window", action="store_true", help="skip the extra replay of bench_my's whole decode window (out = in tokens)")
    ap.add_argument("--no-sweep", action="store_true", help="skip the attention_sweep of the other BASELINE configs")
    ap.add_argument("--prefill-leg", action="store_true", help="time the prefill attention op for models other than the headline one too")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                             "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hip attention backend has no CPU path")
    # NVH_BENCH_REHEARSE=gloo: all ranks share cuda:0 and talk over gloo — a dry run of the N>1 code path (sharding, per-rank
    # kernel shapes, capture fallback) on a one-GPU box; numbers from it mean nothing and the line says so
    rehearse = os.environ.get("NVH_BENCH_REHEARSE") == "gloo"
    if rehearse:
        local_rank = 0
        args.no_full_window = True                 # (ranks time-share one GPU: a dry run of the code path, kept short)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("NVH_ALLREDUCE_MEASURE", "1")          # log one-shot vs RCCL at start-up (the choice does not depend on it)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    from nanovllm_hip.engine.llm_engine import LLMEngine
    from nanovllm_hip.engine.sequence import Sequence
    from nanovllm_hip.models.qwen import model_config

    if args.graph_steps > 0:
        from nanovllm_hip.engine import model_runner as _mr
        _mr.DecodeSession.MULTI = args.graph_steps
    from nanovllm_hip.models import qwen as _qwen
    if args.kv_ahead:
        _qwen.KV_AHEAD = args.kv_ahead
    if args.no_prefetch:
        _qwen.PREFETCH_WEIGHTS = False
    if args.qkv_attend:
        from nanovllm_hip.models import qwen as _qwen
        _qwen.QKV_ATTEND_MODE = args.qkv_attend
        _qwen.KV_PREFETCH_PASSES = args.kv_prefetch_passes
    cfg = model_config(args.model)
    bs = cfg.kvcache_block_size
    window = max(args.steps, args.input_len) if not args.no_full_window else args.steps      # bench_my's decode phase: out = in tokens
    total_len = args.input_len + window + args.warmup + 2
    blocks_per_seq = (total_len + bs - 1) // bs
    def run_decode():
        """Engine start-up, prefill, W untimed + K timed decode steps and the full-window replay.  A one-shot all-reduce that marked a
        failed call (a peer that did not arrive within its bounded wait: never yet seen, but xGMI has never run this kernel either) raises
        on EVERY rank at the first synchronisation behind it."""
        engine = LLMEngine(cfg, num_kvcache_blocks=args.batch * blocks_per_seq + 8, max_model_len=max(4096, total_len),
                           enforce_eager=args.eager, seed=0, warmup=True)     # start-up warmup as the reference (model_runner.py:107-121)

        seed(0)
        prompts = [[randint(0, 10000) for _ in range(args.input_len)] for _ in range(args.batch)]
        seqs = [Sequence(p, max_tokens=window + 1) for p in prompts]

        # ---- prefill (outside the timed decode region; timed on its own)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        engine.prefill(seqs, reserve_tokens=window + args.warmup + 2)
        torch.cuda.synchronize()
        prefill_s = time.perf_counter() - t0
        ctx0 = len(seqs[0])                                               # input_len + 1: first decode step's context

        # ---- decode: W untimed steps, rewind to the same context, then exactly K timed steps
        sess = engine.runner.decode_session(seqs, window + args.warmup + 1, use_graph=not args.eager)
        state0 = sess.state()
        sess.step(args.warmup)
        sess.rewind(state0)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sess.step(args.steps)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        graph_mode = sess.graph is not None
        if world > 1:
            t = torch.tensor([elapsed], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        tok_s = args.batch * args.steps / elapsed
        engine.runner.raise_if_device_failed()

        # ---- the whole bench_my window (contexts in+1 -> 2 in), once more from the same start: what bench_my.py:27-40 reports for in = out
        full = None
        if not args.no_full_window:
            if window == args.steps:
                full_elapsed = elapsed
            else:
                sess.rewind(state0)
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                t0 = time.perf_counter()
                sess.step(window)
                torch.cuda.synchronize()
                if world > 1:
                    dist.barrier()
                full_elapsed = time.perf_counter() - t0
                if world > 1:
                    t = torch.tensor([full_elapsed], device="cuda", dtype=torch.float64)
                    dist.all_reduce(t, op=dist.ReduceOp.MAX)
                    full_elapsed = float(t.item())
                engine.runner.raise_if_device_failed()
            full = (window, full_elapsed)
        return engine, seqs, prefill_s, ctx0, sess, elapsed, graph_mode, full

    oneshot_note = None
    try:
        engine, seqs, prefill_s, ctx0, sess, elapsed, graph_mode, full = run_decode()
    except RuntimeError as e:
        if world == 1 or "one-shot all-reduce failed" not in str(e):
            raise
        # every rank is here (the check is a collective): measure on RCCL instead and say so in the line
        oneshot_note = f"the one-shot all-reduce was abandoned after: {e}"
        os.environ["NVH_ALLREDUCE"] = "rccl"
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        engine, seqs, prefill_s, ctx0, sess, elapsed, graph_mode, full = run_decode()
    tok_s = args.batch * args.steps / elapsed

    # ---- roofline leg: the attention op alone at the mean context of the timed window
    mean_ctx = ctx0 + (args.steps - 1) // 2
    tp = world
    attn_us, attn_bytes, shape_rank = attention_leg(cfg, tp, args.batch, mean_ctx, cfg.num_hidden_layers)
    achieved = attn_bytes / attn_us / 1e3                              # GB/s

    traffic, traffic_note = pmc_traffic(cfg, tp, args.batch, mean_ctx, attn_bytes)
    prefill = None
    if args.model == "Qwen2-0.5B" or args.prefill_leg:
        del sess
        torch.cuda.empty_cache()
        prefill = {"config5_half": prefill_leg(cfg, tp, 128, 128), "s1024": prefill_leg(cfg, tp, 16, 1024, buffers=8, iters=4),
                   "s1024_exact_hi_lo": prefill_leg(cfg, tp, 16, 1024, buffers=8, iters=4, pv_fp16=False)}
    sweep = None
    if args.model == "Qwen2-0.5B" and world == 1 and not args.no_sweep:
        sweep = attention_sweep(cfg)
    from nanovllm_hip import distributed as nvh_dist
    tp_choice = nvh_dist.last_choice
    failed_epoch = engine.runner.comm.failed_epoch() if engine.runner.comm is not None else None
    from nanovllm_hip.models.qwen import tp_partition
    shapes = [list(tp_partition(cfg.num_attention_heads, cfg.num_key_value_heads, tp, r)) for r in range(tp)]

    result = None
    if rank == 0:
        weight_bytes = sum(p.numel() * p.element_size() for p in engine.runner.model.parameters())
        step_bytes = weight_bytes + cfg.num_hidden_layers * attn_bytes
        step_us = elapsed / args.steps * 1e6
        layers = cfg.num_hidden_layers
        attn_floor_us = ATTN_FIXED_FLOOR_US + attn_bytes / (HBM_SUSTAINED_GBPS * 1e3)
        step_floor_us = step_us - layers * max(attn_us - attn_floor_us, 0.0)
        result = {
            "metric": "decode tok/s (bench_my.py) Qwen2-0.5B bs=32 in=out=1024; % HBM roofline" if args.model == "Qwen2-0.5B" and args.batch == 32
                      else f"decode tok/s (bench_my.py-shaped) {args.model} bs={args.batch}",
            "value": round(tok_s, 1), "unit": "tok/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic (random prompts randint(0,10000) seed 0; random-init weights N(0,0.02) seed 0)" + (" REHEARSAL over gloo on one GPU: not a measurement" if rehearse else ""),
            "config": {"workload": f"{args.model} bs={args.batch} in={args.input_len} decode steps={args.steps} --attn-backend hip, "
                                   f"TP={tp}, {'HIP-graph replay' if graph_mode else 'eager steps (graph capture unavailable)'}, device-resident metadata",
                       "global_batch": args.batch, "context_first_step": ctx0, "parallelism": f"tp{tp}",
                       "collective": ((f"{dist.get_backend()} world_size {world} (torch.distributed; nccl = RCCL over xGMI); decode all-reduces: "
                                       + ("one-shot over IPC-mapped peer buffers with the residual add fused in (nvh_allreduce_oneshot), inside the HIP graph"
                                          if engine.runner.comm is not None else "RCCL ring through torch.distributed") + f" [{tp_choice}]") if world > 1 else "none (single GPU)"),
                       "collective_detail": ({"path": "oneshot" if engine.runner.comm is not None else "torch.distributed",
                                              "startup_measurement_us": nvh_dist.last_measurement, "failed_epoch": failed_epoch,
                                              "note": oneshot_note} if world > 1 else None),
                       "heads_per_rank_q0_qn_kv0_kvn": shapes},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_note,
                         "kernel": "one decode attention call (nvh_paged_decode) = one launch of paged_decode_chunked_kernel (split-KV passes + last-arriver combine)",
                         "bytes_per_launch": attn_bytes, "us_per_launch": round(attn_us, 2), "context": mean_ctx,
                         "shape_per_rank": list(shape_rank)},
            "decode_step_roofline": {"bytes_per_step": int(step_bytes), "us_at_8TBps": round(step_bytes / 8e6, 1), "us_measured": round(step_us, 1),
                                     "frac": round(step_bytes / 8e6 / step_us, 4),
                                     "attention_share_of_step": round(cfg.num_hidden_layers * attn_us / step_us, 3)},
            # what the step would take with every attention call at its floor as a launch of its own (boundary + ramp + first byte 3.75 us,
            # then the K/V stream at the 6.29 TB/s the part sustains, a free hand-off) and every other launch as measured: the part of the
            # step SURVEY section 8's rows can move; the rest is the model body's launches (out of scope, hipBLASLt in the reference)
            "step_floor": {"attention_floor_us_per_call": round(attn_floor_us, 2), "attention_measured_us_per_call": round(attn_us, 2),
                           "us_per_step_with_attention_at_floor": round(step_floor_us, 1),
                           "decode_step_roofline_frac_at_floor": round(step_bytes / 8e6 / step_floor_us, 4)},
            "full_window": ({"steps": full[0], "contexts": [ctx0, ctx0 + full[0] - 1], "tok_s": round(args.batch * full[0] / full[1], 1),
                             "ms_per_step": round(full[1] / full[0] * 1e3, 4),
                             "decode_step_roofline_frac": round((weight_bytes + cfg.num_hidden_layers * decode_attn_bytes(
                                 [ctx0 + (full[0] - 1) // 2] * args.batch, *shape_rank, cfg.kvcache_block_size, with_store=False)) / 8e6 / (full[1] / full[0] * 1e6), 4),
                             "bench_my_tok_s": round(args.batch * (full[0] + 1) / (prefill_s + full[1]), 1)} if full else None),
            "attention_sweep": sweep,
            "prefill": prefill,
            "prefill_s": round(prefill_s, 4),
            "bench_my_tok_s": round(args.batch * (args.steps + 1) / (prefill_s + elapsed), 1),
            "attention_only_tok_s": round(args.batch / (attn_us * 1e-6 * cfg.num_hidden_layers), 1),
        }
        if world == 1 and not args.no_cpu_baseline:
            sys.path.insert(0, ROOT)
            result["cpu_baseline"] = cpu_baseline_leg(cfg, args.batch, mean_ctx)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
