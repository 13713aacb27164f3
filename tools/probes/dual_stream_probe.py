#!/usr/bin/env python3
"""Feasibility probe: do two independent decode graphs (half batches) replayed on two streams overlap on this stack?
Timing only — the two sessions share scratch buffers here, so their tokens are not meaningful."""
import os, sys, time
from random import randint, seed
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-learn_amd"))
from nanovllm_hip.engine.llm_engine import LLMEngine
from nanovllm_hip.engine.sequence import Sequence
from nanovllm_hip.models.qwen import model_config

B, IN, STEPS = 32, 1024, 200
CAP = 3 * STEPS              # session capacity: each timing loop advances the same sessions again
cfg = model_config("Qwen2-0.5B")
bs = cfg.kvcache_block_size
total = IN + CAP + 8
eng = LLMEngine(cfg, num_kvcache_blocks=B * ((total + bs - 1) // bs) + 8, max_model_len=4096, seed=0, warmup=True)
seed(0)
seqs = [Sequence([randint(0, 10000) for _ in range(IN)], max_tokens=CAP + 1) for _ in range(B)]
eng.prefill(seqs, reserve_tokens=CAP + 8)

def timed(sessions, streams):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS - 20):
        for s, st in zip(sessions, streams):
            with torch.cuda.stream(st):
                s.step(1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / (STEPS - 20) * 1e3

one = eng.runner.decode_session(seqs, CAP + 2)
one.step(10)
print(f"one session of 32 rows:            {timed([one], [torch.cuda.current_stream()]):.3f} ms per step")
halves = [eng.runner.decode_session(seqs[:16], CAP + 2), eng.runner.decode_session(seqs[16:], CAP + 2)]
for h in halves:
    h.step(5)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
print(f"two sessions of 16 rows, 1 stream: {timed(halves, [torch.cuda.current_stream()] * 2):.3f} ms per step (both halves)")
print(f"two sessions of 16 rows, 2 streams:{timed(halves, [s1, s2]):.3f} ms per step (both halves)")
