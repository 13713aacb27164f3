"""Minimal generate() loop with the reference's scheduling rules that shape attention inputs
(nanovllm/engine/scheduler.py:27-89, llm_engine.py:94-143): prefill first, FCFS, a prefill batch holds at most
max_num_batched_tokens tokens and max_num_seqs sequences; then all running sequences decode together.
Greedy sampling only (bench_my.py uses temperature 0, ignore_eos, fixed max_tokens), so control flow is
value-independent and the decode phase runs as a device-resident graph session."""
from __future__ import annotations

import torch

from ..models.qwen import ModelConfig
from .block_manager import BlockManager
from .model_runner import ModelRunner
from .sequence import Sequence


class LLMEngine:
    def __init__(self, cfg: ModelConfig, num_kvcache_blocks, max_num_batched_tokens=16384, max_num_seqs=512,
                 max_model_len=4096, enforce_eager=False, device=None, seed=0, warmup=False):
        self.cfg = cfg
        self.max_num_batched_tokens = max_num_batched_tokens
        self.max_num_seqs = max_num_seqs
        self.enforce_eager = enforce_eager
        self.runner = ModelRunner(cfg, num_kvcache_blocks, device=device, max_model_len=max_model_len, seed=seed)
        self.block_manager = BlockManager(num_kvcache_blocks, cfg.kvcache_block_size)
        if warmup:                                                   # the reference always warms up (model_runner.py:52)
            self.runner.warmup_model(max_num_batched_tokens, max_num_seqs)

    def prefill(self, seqs, reserve_tokens):
        """Run every waiting sequence through prefill in FCFS batches; appends the first generated token."""
        waiting = list(seqs)
        while waiting:
            batch, ntok = [], 0
            while waiting and len(batch) < self.max_num_seqs and ntok + len(waiting[0]) <= self.max_num_batched_tokens:
                s = waiting.pop(0)
                self.block_manager.allocate(s, reserve_tokens=reserve_tokens)
                batch.append(s)
                ntok += len(s)
            if not batch:
                raise RuntimeError("a prompt exceeds max_num_batched_tokens")
            for s, t in zip(batch, self.runner.run(batch, True)):
                s.append_token(t)

    def generate(self, prompt_token_ids, max_tokens):
        """Greedy-generate `max_tokens` tokens for every prompt; returns the completion token ids."""
        seqs = [Sequence(p, max_tokens=max_tokens) for p in prompt_token_ids]
        self.prefill(seqs, reserve_tokens=max_tokens)
        if max_tokens > 1:
            if self.enforce_eager:
                for _ in range(max_tokens - 1):
                    for s, t in zip(seqs, self.runner.run(seqs, False)):
                        s.append_token(t)
            else:
                sess = self.runner.decode_session(seqs, max_tokens - 1)
                sess.step(max_tokens - 1)
                sess.finish()
        out = [s.completion_token_ids for s in seqs]
        for s in seqs:
            self.block_manager.deallocate(s)
        return out
