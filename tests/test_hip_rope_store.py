"""GPU: the fused (q/k norm ->) RoPE -> store_kvcache launch (nvh_rope_store) against the
reference-generated RoPE golden and the oracle restatement.  bf16 outputs: bit-exact against the reference's own
arithmetic (same fp32 operation order), store rows bit-exact."""
import numpy as np
import pytest
import torch

from oracle import oracle as O

pytestmark = pytest.mark.gpu


def dev_bf16(bits):
    return torch.from_numpy(np.ascontiguousarray(bits).view(np.int16)).cuda().view(torch.bfloat16)


def bits(t):
    return t.contiguous().view(torch.int16).cpu().numpy().view(np.uint16)


@pytest.mark.parametrize("name", ["rope_d64.npz", "rope_d128.npz"])
def test_rope_golden_bit_exact(golden, name):
    """All heads treated as q heads (no cache): output must equal the reference's apply_rotary_emb bit for bit, with the
    reference's own cos/sin table rebuilt by torch on the device (same ops as rotary_embedding.py:29-36)."""
    from nanovllm_hip import ops
    from nanovllm_hip.models.qwen import cos_sin_table
    g = golden(name)
    heads, d, max_pos = (int(x) for x in g["shape"])
    table = cos_sin_table(d, max_pos, float(g["base"][0]), "cuda")
    assert np.array_equal(table[:64].cpu().numpy(), g["cos_sin"])
    x = dev_bf16(g["x"])                                    # [N, heads, D]
    n = x.shape[0]
    kvh = 1
    qkv = torch.cat([x.reshape(n, heads * d), torch.zeros(n, 2 * kvh * d, dtype=torch.bfloat16, device="cuda")], dim=1).contiguous()
    ops.rope_store(qkv, torch.from_numpy(g["positions"]).cuda(), table, heads, kvh, d)
    torch.cuda.synchronize()
    assert np.array_equal(bits(qkv[:, :heads * d]).reshape(n, heads, d), g["expected"])


@pytest.mark.parametrize("H,KVH,D,norm", [(14, 2, 64, False), (16, 8, 128, True), (7, 1, 128, False), (4, 2, 64, True)])
def test_rope_store_vs_oracle(H, KVH, D, norm):
    from nanovllm_hip import ops
    from nanovllm_hip.models.qwen import cos_sin_table
    gen = torch.Generator().manual_seed(H * D)
    n, bs, nb = 45, 256, 3
    table = cos_sin_table(D, 4096, 1e6, "cuda")
    qkv = torch.randn(n, (H + 2 * KVH) * D, generator=gen).bfloat16()
    pos = torch.randint(0, 4096, (n,), generator=gen)
    slots = torch.randperm(nb * bs, generator=gen)[:n].int()
    slots[3::5] = -1
    qw = (1 + 0.1 * torch.randn(D, generator=gen)).bfloat16() if norm else None
    kw = (1 + 0.1 * torch.randn(D, generator=gen)).bfloat16() if norm else None
    kc = torch.randn(nb, bs, KVH, D, generator=gen).bfloat16()
    vc = torch.randn(nb, bs, KVH, D, generator=gen).bfloat16()
    # oracle
    x = qkv.float().numpy()
    q = x[:, :H * D].reshape(n, H, D)
    k = x[:, H * D:(H + KVH) * D].reshape(n, KVH, D)
    v = x[:, (H + KVH) * D:].reshape(n, KVH, D)
    if norm:
        q = O.rms_norm_heads(q, qw.float().numpy(), 1e-6)
        k = O.rms_norm_heads(k, kw.float().numpy(), 1e-6)
    tab = table.cpu().numpy()
    q_exp, k_exp = O.rope_neox(q, pos.numpy(), tab), O.rope_neox(k, pos.numpy(), tab)
    kc_exp, vc_exp = kc.float().numpy().copy(), vc.float().numpy().copy()
    O.store_kvcache(k_exp, v, kc_exp, vc_exp, slots.numpy())
    # device
    qkv_d, kc_d, vc_d = qkv.cuda(), kc.cuda(), vc.cuda()
    ops.rope_store(qkv_d, pos.cuda(), table, H, KVH, D, kc_d, vc_d, slots.cuda(), qw.cuda() if norm else None, kw.cuda() if norm else None, 1e-6)
    torch.cuda.synchronize()
    got = qkv_d.float().cpu().numpy()
    tol = 2.0 ** -7 if norm else 0.0              # with the norm, rsqrt differs in the last fp32 bit -> <= one bf16 ulp
    for name, g_, e_ in (("q", got[:, :H * D].reshape(n, H, D), q_exp), ("k", got[:, H * D:(H + KVH) * D].reshape(n, KVH, D), k_exp)):
        assert (np.abs(g_ - e_) <= tol * np.abs(e_)).all(), name
        if norm:
            assert (g_ != e_).mean() < 0.02
    assert np.array_equal(got[:, (H + KVH) * D:], x[:, (H + KVH) * D:])            # v untouched
    # cache rows: whatever k the kernel produced is what it stored; v rows exact
    k_rows = got[:, H * D:(H + KVH) * D].reshape(n, KVH, D)
    kc_chk, vc_chk = kc.float().numpy().copy(), vc.float().numpy().copy()
    O.store_kvcache(k_rows, v, kc_chk, vc_chk, slots.numpy())
    assert np.array_equal(kc_d.float().cpu().numpy(), kc_chk) and np.array_equal(vc_d.float().cpu().numpy(), vc_chk)
    assert np.array_equal(vc_d.float().cpu().numpy(), vc_exp)


def test_model_fused_path_equals_reference_call_sequence():
    """One decoder attention block: rope_store_attend (fused launch) == q/k norm + rotary_emb + Attention.forward (the
    reference's call sequence, qwen3.py:104-117), for a prefill followed by a decode step."""
    from nanovllm_hip import reset_context, set_context
    from nanovllm_hip.models.qwen import QwenAttention, model_config
    cfg = model_config("Qwen3-0.6B", num_hidden_layers=1)
    torch.manual_seed(0)
    blk = QwenAttention(cfg).cuda().bfloat16()
    ref = QwenAttention(cfg).cuda().bfloat16()
    ref.load_state_dict(blk.state_dict())
    del ref.attn.__class__.rope_store_attend                          # force the unfused reference call sequence
    try:
        bs = 256
        kv1 = torch.zeros(2, 4, bs, cfg.num_key_value_heads, cfg.head_dim, dtype=torch.bfloat16, device="cuda")
        kv2 = torch.zeros_like(kv1)
        blk.attn.k_cache, blk.attn.v_cache = kv1[0], kv1[1]
        ref.attn.k_cache, ref.attn.v_cache = kv2[0], kv2[1]
        lens, tables = [300, 40], [[2, 0], [3]]
        seqs = [O.SeqState(n, t) for n, t in zip(lens, tables)]
        m = O.prepare_prefill(seqs)
        i32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).cuda()
        hidden = torch.randn(sum(lens), cfg.hidden_size, device="cuda").bfloat16()
        pos = torch.from_numpy(m["positions"]).cuda()
        outs = []
        for mod in (blk, ref):
            set_context(True, i32(m["cu_seqlens_q"]), i32(m["cu_seqlens_k"]), m["max_seqlen_q"], m["max_seqlen_k"], i32(m["slot_mapping"]), None, None)
            outs.append(mod(pos, hidden))
        assert torch.equal(kv1, kv2)
        assert (outs[0].float() - outs[1].float()).abs().max() <= 2e-2 * outs[1].float().abs().max()
        for s in seqs:
            s.num_tokens += 1
        p, slots, ctx, bt = O.prepare_decode(seqs)
        hd = torch.randn(2, cfg.hidden_size, device="cuda").bfloat16()
        outs = []
        for mod in (blk, ref):
            set_context(False, slot_mapping=i32(slots), context_lens=i32(ctx), block_tables=i32(bt))
            outs.append(mod(torch.from_numpy(p).cuda(), hd))
        reset_context()
        assert torch.equal(kv1, kv2)
        assert (outs[0].float() - outs[1].float()).abs().max() <= 2e-2 * outs[1].float().abs().max()
    finally:
        from nanovllm_hip.layers import attention_hip
        import importlib
        importlib.reload(attention_hip)
