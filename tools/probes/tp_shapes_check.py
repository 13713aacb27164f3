#!/usr/bin/env python3
"""Every rank's shapes of a tensor-parallel run, exercised on ONE GPU in ONE process: torch.distributed is stubbed (rank r of
tp, all_reduce = identity, backend "nccl" so that the decode step is captured as on a real node), so the numbers mean nothing —
what is checked is that each rank's kernels accept their shard shapes (uneven head splits with replicated kv heads at tp = 4 / 8,
a 608-wide MLP shard at tp = 8 that leaves the fused layer), that the graph captures and replays, and that outputs are finite.
The real N > 1 runs are the driver's; this is the closest a one-GPU box gets to the N = 8 shapes.
  python tools/probes/tp_shapes_check.py [Qwen2-0.5B|Qwen3-0.6B|Qwen2-7B]"""
import os, sys
import torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-learn_amd"))
from nanovllm_hip.engine.llm_engine import LLMEngine
from nanovllm_hip.engine.sequence import Sequence
from nanovllm_hip.models.qwen import model_config, tp_partition

name = sys.argv[1] if len(sys.argv) > 1 else "Qwen2-0.5B"
state = {"rank": 0, "tp": 1}
dist.is_initialized = lambda: True
dist.get_rank = lambda *a, **k: state["rank"]
dist.get_world_size = lambda *a, **k: state["tp"]
dist.get_backend = lambda *a, **k: "nccl"
dist.all_reduce = lambda t, *a, **k: None
dist.barrier = lambda *a, **k: None
dist.broadcast = lambda *a, **k: None

g = torch.Generator().manual_seed(0)
for tp in (2, 4, 8):
    cfg = model_config(name, num_hidden_layers=2)
    if cfg.num_attention_heads % tp and tp % cfg.num_key_value_heads:
        print(f"tp={tp}: not expressible for {name}"); continue
    for rank in range(tp):
        state.update(rank=rank, tp=tp)
        try:
            part = tp_partition(cfg.num_attention_heads, cfg.num_key_value_heads, tp, rank)
        except AssertionError as e:
            print(f"tp={tp}: {e}"); break
        eng = LLMEngine(cfg, num_kvcache_blocks=32 * 2 + 8, max_model_len=4096, seed=0)
        prompts = [torch.randint(0, 10000, (int(n),), generator=g).tolist() for n in torch.randint(200, 300, (32,), generator=g)]
        seqs = [Sequence(p, max_tokens=8) for p in prompts]
        eng.prefill(seqs, reserve_tokens=10)
        sess = eng.runner.decode_session(seqs, 8, use_graph=True)
        sess.step(4)
        torch.cuda.synchronize()
        toks = sess.tokens[:4]
        fused = eng.runner.model._fused_shapes_ok()
        assert sess.graph is not None, "capture failed"
        assert int(toks.min()) >= 0 and int(toks.max()) < cfg.vocab_size
        print(f"tp={tp} rank {rank}: heads (q_start, q_count, kv_start, kv_count) = {part}; fused layer {fused}; graph replay ok", flush=True)
        del sess, eng
        torch.cuda.empty_cache()
# fused vs plain decode layer on the same (stubbed) shard: the zero-padded MLP shard of tp = 8 goes through the streaming GEMMs
from nanovllm_hip.engine.model_runner import build_decode_meta
from nanovllm_hip.models import qwen
from nanovllm_hip.utils.context import reset_context, set_context
for tp, rank in ((8, 0), (8, 3), (4, 1)):
    cfg = model_config(name, num_hidden_layers=2, vocab_size=4096)
    try:
        tp_partition(cfg.num_attention_heads, cfg.num_key_value_heads, tp, rank)
    except AssertionError:
        continue
    state.update(rank=rank, tp=tp)
    out = {}
    for fused in (True, False):
        qwen.FUSED_DECODE = fused
        eng = LLMEngine(cfg, num_kvcache_blocks=16, enforce_eager=True, seed=3)
        gg = torch.Generator().manual_seed(7)
        seqs = [Sequence(torch.randint(0, 4096, (n,), generator=gg).tolist(), max_tokens=4) for n in (300, 17, 256, 5, 129)]
        eng.prefill(seqs, reserve_tokens=4)
        r = eng.runner
        m = build_decode_meta(seqs, r.block_size)
        with torch.inference_mode():
            set_context(False, slot_mapping=r._dev(m["slot_mapping"]), context_lens=r._dev(m["context_lens"]), block_tables=r._dev(m["block_tables"]))
            hidden = r.model(r._dev(m["input_ids"]), r._dev(m["positions"]))
            if fused:
                assert isinstance(hidden, qwen.PackedResidual), "the fused layer was expected to run"
            out[fused] = r.model.compute_logits(hidden).float().cpu()
            reset_context()
    qwen.FUSED_DECODE = True
    a, b = out[True], out[False]
    print(f"tp={tp} rank {rank}: MLP shard {eng.runner.model.layers[0].mlp.inter} -> {eng.runner.model.layers[0].mlp.down_proj.weight.shape[1]} columns; "
          f"fused vs plain logits max abs diff {float((a - b).abs().max()):.4f} (scale {float(b.abs().max()):.2f})", flush=True)
    assert (a - b).abs().max() <= 0.03 * b.abs().max()
print("all rank shapes ok")
