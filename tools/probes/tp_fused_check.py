#!/usr/bin/env python3
"""Rehearsal of the tensor-parallel fused decode layer on ONE GPU: ranks share cuda:0 and talk over gloo.
Each rank runs prefill + one decode step of a 2-layer model with the fused layer and with the plain layer and compares its
logits; rank 0 also compares them with a single-process (tp=1) run it is given via --ref.
  python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/probes/tp_fused_check.py"""
import os, sys
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-learn_amd"))
world = int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=world)
torch.cuda.set_device(0)
from nanovllm_hip.engine.llm_engine import LLMEngine
from nanovllm_hip.engine.model_runner import build_decode_meta
from nanovllm_hip.engine.sequence import Sequence
from nanovllm_hip.models import qwen
from nanovllm_hip.models.qwen import model_config
from nanovllm_hip.utils.context import reset_context, set_context

cfg = model_config(os.environ.get("MODEL", "Qwen2-0.5B"), num_hidden_layers=2, vocab_size=4096)
g = torch.Generator().manual_seed(7)
prompts = [torch.randint(0, 4096, (n,), generator=g).tolist() for n in (300, 17, 256, 5, 129)]
out = {}
for fused in (True, False):
    qwen.FUSED_DECODE = fused
    eng = LLMEngine(cfg, num_kvcache_blocks=16, enforce_eager=True, seed=3)
    seqs = [Sequence(p, max_tokens=4) for p in prompts]
    eng.prefill(seqs, reserve_tokens=4)
    r = eng.runner
    m = build_decode_meta(seqs, r.block_size)
    with torch.inference_mode():
        set_context(False, slot_mapping=r._dev(m["slot_mapping"]), context_lens=r._dev(m["context_lens"]), block_tables=r._dev(m["block_tables"]))
        hidden = r.model(r._dev(m["input_ids"]), r._dev(m["positions"]))
        used_fused = isinstance(hidden, qwen.PackedResidual)
        out[fused] = r.model.compute_logits(hidden).float().cpu()
        reset_context()
    if fused:
        assert used_fused or not r.model._fused_shapes_ok(), "the fused layer was expected to run"
        took = used_fused
a, b = out[True], out[False]
rank = dist.get_rank() if world > 1 else 0
print(f"[rank {rank}/{world}] fused layer used: {took}; fused vs plain logits: max abs diff {float((a - b).abs().max()):.4f} "
      f"(scale {float(b.abs().max()):.2f}); same arg-max: {bool(torch.equal(a.argmax(-1), b.argmax(-1)))}", flush=True)
assert (a - b).abs().max() <= 0.03 * b.abs().max() and torch.equal(a.argmax(-1), b.argmax(-1))
if world > 1:
    gathered = [torch.zeros_like(a) for _ in range(world)]
    dist.all_gather(gathered, a)
    assert all(torch.equal(gathered[0], t) for t in gathered), "ranks disagree on the logits"
    dist.barrier()
if rank == 0:
    torch.save(a, os.path.join(ROOT, "gpurun_out", f"tp_fused_logits_w{world}.pt"))
    ref = os.path.join(ROOT, "gpurun_out", "tp_fused_logits_w1.pt")
    if world > 1 and os.path.exists(ref):
        r1 = torch.load(ref, weights_only=True)
        print(f"[rank 0] tp={world} vs tp=1 logits: max abs diff {float((a - r1).abs().max()):.4f}; same arg-max: {bool(torch.equal(a.argmax(-1), r1.argmax(-1)))}", flush=True)
        assert (a - r1).abs().max() <= 0.03 * r1.abs().max()
