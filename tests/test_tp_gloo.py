"""Multi-process CPU tests of the N>1 path (gloo): tensor-parallel head split + all-reduce of the model body gives the
single-process result, for the reference's even split (tp=2) and for the replicated-kv / uneven-q split (tp=4 with
6/2 heads).  The attention op itself needs a GPU, so these tests plug a TEST-ONLY stand-in (the numpy oracle) into the
same dispatch point the hip backend uses; what is under test is the sharding, the weight slicing and the collectives."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class OracleAttention(torch.nn.Module):
    """Test stand-in with the reference module contract (ctor args, k_cache/v_cache attrs, forward(q, k, v), Context)."""

    def __init__(self, num_heads, head_dim, scale, num_kv_heads, **kw):
        super().__init__()
        self.num_heads, self.head_dim, self.scale, self.num_kv_heads = num_heads, head_dim, scale, num_kv_heads
        self.k_cache = self.v_cache = torch.tensor([])

    def forward(self, q, k, v):
        from nanovllm_hip import get_context
        from oracle import oracle as O
        ctx = get_context()
        assert ctx.is_prefill
        n = q.shape[0]
        o = O.prefill_varlen(q.view(n, self.num_heads, self.head_dim).double().numpy(), k.reshape(n, self.num_kv_heads, self.head_dim).double().numpy(),
                             v.reshape(n, self.num_kv_heads, self.head_dim).double().numpy(), ctx.cu_seqlens_q.numpy(), ctx.cu_seqlens_k.numpy(), scale=self.scale)
        return torch.from_numpy(o).to(q.dtype).view(n, self.num_heads * self.head_dim)


def _model_out(cfg_kwargs, lens):
    from nanovllm_hip import reset_context, set_context
    from nanovllm_hip.models import qwen
    qwen.resolve_attention = lambda backend, block_size=256: (OracleAttention, {})
    cfg = qwen.ModelConfig(name="tiny", **cfg_kwargs)
    model = qwen.QwenForCausalLM(cfg).init_random(seed=3).float()
    t = sum(lens)
    gen = torch.Generator().manual_seed(1)
    ids = torch.randint(0, cfg.vocab_size, (t,), generator=gen)
    pos = torch.cat([torch.arange(n) for n in lens])
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32)
    set_context(True, cu, cu, max(lens), max(lens), None, None, None)
    with torch.no_grad():
        out = model.compute_logits(model(ids, pos))
    reset_context()
    return out


def _worker(rank, world, port, cfg_kwargs, lens, ret):
    for p in (ROOT, os.path.join(ROOT, "nano-vllm-learn_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    try:
        # without a GPU the start-up choice of the all-reduce path must come out as "torch.distributed" on every rank, without
        # touching the HIP library (the one-shot IPC path is GPU-only; tests/test_hip_allreduce.py rehearses it on the GPU box)
        from nanovllm_hip.distributed import init_tensor_parallel_comm, tensor_parallel_comm
        assert init_tensor_parallel_comm(64, cfg_kwargs["hidden_size"]) is None and tensor_parallel_comm() is None
        out = _model_out(cfg_kwargs, lens)
        gathered = [torch.empty_like(out) for _ in range(world)]
        dist.all_gather(gathered, out)
        if rank == 0:
            ret["out"] = out.numpy()
            ret["ranks_agree"] = all(torch.equal(g, out) for g in gathered)      # replicated LM head -> identical logits everywhere
    finally:
        dist.destroy_process_group()


TINY = dict(num_hidden_layers=2, hidden_size=128, head_dim=32, intermediate_size=256, vocab_size=512, tie_word_embeddings=True,
            qkv_bias=True, qk_norm=False, max_position_embeddings=512)


@pytest.mark.parametrize("world,heads,kv_heads,qk_norm,inter", [(2, 4, 2, False, 256), (4, 6, 2, False, 256), (2, 4, 2, True, 256),
                                                                (4, 6, 2, False, 320)])   # 320 / 4 = 80-wide MLP shards, zero-padded to 128
def test_tp_equals_single_process(world, heads, kv_heads, qk_norm, inter):
    cfg_kwargs = dict(TINY, num_attention_heads=heads, num_key_value_heads=kv_heads, qk_norm=qk_norm, intermediate_size=inter)
    lens = [9, 1, 20]
    ref = _model_out(cfg_kwargs, lens).numpy()                      # single process (no process group): tp = 1
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() + world * 7 + heads) % 400
    mp.spawn(_worker, args=(world, port, cfg_kwargs, lens, ret), nprocs=world, join=True)
    assert ret["ranks_agree"]
    assert np.abs(ret["out"] - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())


def test_partition_covers_every_head_exactly_once():
    from nanovllm_hip.models.qwen import tp_partition
    for h, kvh in ((14, 2), (28, 4), (16, 8), (6, 2)):
        g = h // kvh
        for tp in (1, 2, 4, 8):
            if not ((h % tp == 0 and kvh % tp == 0) or (tp % kvh == 0 and g >= tp // kvh)):
                continue
            seen = []
            for r in range(tp):
                q0, qn, kv0, kvn = tp_partition(h, kvh, tp, r)
                assert qn >= 1 and kvn >= 1
                for qh in range(q0, q0 + qn):
                    assert kv0 <= qh // g < kv0 + kvn          # a rank holds the kv head of every q head it owns
                seen += list(range(q0, q0 + qn))
            assert sorted(seen) == list(range(h))
