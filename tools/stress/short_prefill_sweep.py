#!/usr/bin/env python3
"""Stress: the short-sequence prefill kernel against the tiled kernel on many random varlen batches (no oracle: both are held to
the oracle by tests/test_hip_parity.py; here only agreement, on far more shapes).  usage: short_prefill_sweep.py [cases]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-learn_amd"))
from nanovllm_hip import ops

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(123)
worst = 0.0
for c in range(cases):
    D = int(rng.choice([64, 128])); KVH = int(rng.choice([1, 2, 3, 4, 8])); G = int(rng.choice([1, 2, 3, 4, 7, 8])); H = KVH * G
    nseq = int(rng.integers(1, 40))
    klens = rng.integers(1, 129, size=nseq)
    qlens = np.array([int(rng.integers(1, k + 1)) for k in klens]) if rng.random() < 0.3 else klens.copy()
    Tq, Tk = int(qlens.sum()), int(klens.sum())
    q = torch.randn(Tq, H, D, device="cuda", dtype=torch.bfloat16)
    kv = torch.randn(Tk, 2, KVH, D, device="cuda", dtype=torch.bfloat16)
    cuq = torch.tensor(np.concatenate([[0], np.cumsum(qlens)]), dtype=torch.int32, device="cuda")
    cuk = torch.tensor(np.concatenate([[0], np.cumsum(klens)]), dtype=torch.int32, device="cuda")
    outs = []
    for mode in ("short", "tiled"):
        for waves in ((8, 16) if mode == "short" and D == 64 else (8,)):
            outs.append(ops.flash_attn_varlen_func(q, kv[:, 0], kv[:, 1], int(qlens.max()), cuq, int(klens.max()), cuk, out_dtype=torch.float32,
                                                   kernel=mode, short_waves=waves if mode == "short" else 0))
    torch.cuda.synchronize()
    for o in outs[:-1]:
        assert torch.isfinite(o).all()
        err = (o - outs[-1]).abs().max().item()
        worst = max(worst, err)
        assert err <= 5e-4, (c, D, KVH, G, qlens.tolist(), klens.tolist(), err)
print(f"{cases} cases ok, worst |short - tiled| = {worst:.2e}")
