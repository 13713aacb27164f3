"""GPU, model level: the engine's decoder stack with the hip `Attention` against THE SAME stack (same weights, same token
stream) with an attention module backed by the CPU oracle — not hip against hip.

SURVEY section 4 asks for "engine greedy tokens hip == sdpa.math"; the reference's own call sequence is
qkv_proj -> split -> RoPE -> attn(q, k, v) -> o_proj (models/qwen3.py:99-119) with the store and the attention inside
`attn` (layers/attention.py:74-103).  The oracle-backed module below has exactly that contract (ctor, k_cache / v_cache
attributes, forward(q, k, v), global Context) and evaluates store / prefill / decode with oracle/oracle.py in float64 on the
host, so everything the hip side fuses around the attention call (RoPE + store in one launch, the store riding in the qkv
GEMM epilogue at decode, fragment-packed attention output, device metadata) is checked against plain arithmetic.

Two hip-side runs, both against the oracle side:
  * the PLAIN layer body with the hip attention module (rope_store_attend: RoPE + store in one launch, then the attention
    kernels): everything outside the attention call is the same torch / GEMM code on both sides, so the bar is bf16 noise —
    max |diff| <= 2^-6 of the largest logit, mean |diff| <= 2^-7 of the mean magnitude (measured on MI355X: 5.7e-3..6.9e-3 and
    3.2e-3..4.9e-3; the logits themselves are bf16, whose rounding alone is 2^-9 relative per value);
  * the FUSED decode layer (what the engine replays from a graph: norm-folded weights, the store in the qkv GEMM epilogue,
    fragment-packed activations): its folded bf16 weights round differently by design, so the bar is the one
    tests/test_hip_layer_ops.py holds fused-vs-plain to (max <= 3 % of the largest logit; mean <= 2^-6; measured 8.5e-3 / 8.4e-3),
    here against the oracle side.
In both runs the greedy token must agree wherever the oracle side's top-2 margin exceeds that noise, and the K/V rows
the hip side stored must match the oracle module's caches."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


class OracleAttention(torch.nn.Module):
    """TEST-ONLY attention with the reference module contract, computed by the CPU oracle (float64)."""

    def __init__(self, num_heads, head_dim, scale, num_kv_heads, **kw):
        super().__init__()
        self.num_heads, self.head_dim, self.scale, self.num_kv_heads = num_heads, head_dim, scale, num_kv_heads
        self.k_cache = self.v_cache = torch.tensor([])
        self._kc = self._vc = None

    def forward(self, q, k, v):
        from nanovllm_hip import get_context
        from oracle import oracle as O
        ctx = get_context()
        n, h, kvh, d = q.shape[0], self.num_heads, self.num_kv_heads, self.head_dim
        qn = q.reshape(n, h, d).float().cpu().numpy().astype(np.float64)
        kn = k.reshape(n, kvh, d).float().cpu().numpy().astype(np.float64)
        vn = v.reshape(n, kvh, d).float().cpu().numpy().astype(np.float64)
        if self.k_cache.numel() and self._kc is None:
            self._kc = np.zeros(tuple(self.k_cache.shape), np.float64)
            self._vc = np.zeros(tuple(self.v_cache.shape), np.float64)
        if self._kc is not None and ctx.slot_mapping is not None:
            O.store_kvcache(kn, vn, self._kc, self._vc, ctx.slot_mapping.cpu().numpy())
        if ctx.is_prefill:
            assert ctx.block_tables is None
            o = O.prefill_varlen(qn, kn, vn, ctx.cu_seqlens_q.cpu().numpy(), ctx.cu_seqlens_k.cpu().numpy(), scale=self.scale)
        else:
            o = O.paged_decode(qn, self._kc, self._vc, ctx.context_lens.cpu().numpy(), ctx.block_tables.cpu().numpy(), scale=self.scale)
        return torch.from_numpy(np.ascontiguousarray(o)).to(device=q.device, dtype=q.dtype).view(n, h * d)


def _runner(backend, cfg_kwargs, monkeypatch):
    from nanovllm_hip.engine.model_runner import ModelRunner
    from nanovllm_hip.models import qwen
    if backend == "oracle":
        monkeypatch.setattr(qwen, "resolve_attention", lambda name, block_size=256: (OracleAttention, {}))
    cfg = qwen.model_config("Qwen2-0.5B", attn_backend=backend, **cfg_kwargs)
    r = ModelRunner(cfg, num_kvcache_blocks=12, seed=4)
    if backend == "oracle":
        monkeypatch.undo()
    return r


@torch.inference_mode()
def _logits(runner, seqs, is_prefill, fused=True):
    from nanovllm_hip import reset_context, set_context
    from nanovllm_hip.engine.model_runner import build_decode_meta, build_prefill_meta
    from nanovllm_hip.models import qwen
    dev = runner.device
    qwen.FUSED_DECODE = fused
    try:
        return _logits_inner(runner, seqs, is_prefill, dev, set_context, reset_context, build_decode_meta, build_prefill_meta)
    finally:
        qwen.FUSED_DECODE = True


def _logits_inner(runner, seqs, is_prefill, dev, set_context, reset_context, build_decode_meta, build_prefill_meta):
    if is_prefill:
        m = build_prefill_meta(seqs)
        set_context(True, m["cu_seqlens_q"].to(dev), m["cu_seqlens_k"].to(dev), m["max_seqlen_q"], m["max_seqlen_k"], m["slot_mapping"].to(dev), None, None)
    else:
        m = build_decode_meta(seqs)
        set_context(False, slot_mapping=m["slot_mapping"].to(dev), context_lens=m["context_lens"].to(dev), block_tables=m["block_tables"].to(dev))
    hidden = runner.model(m["input_ids"].to(dev), m["positions"].to(dev))
    if is_prefill:
        hidden = hidden[m["cu_seqlens_q"].to(dev)[1:].long() - 1]
    logits = runner.model.compute_logits(hidden).float().cpu()
    reset_context()
    return logits


@pytest.mark.parametrize("fused", [False, True], ids=["plain_layer", "fused_decode_layer"])
@pytest.mark.parametrize("family", ["qwen2", "qwen3"])
def test_decoder_stack_hip_vs_oracle_attention(family, fused, monkeypatch):
    from nanovllm_hip.engine.sequence import Sequence
    kw = dict(num_hidden_layers=2, vocab_size=2048)
    if family == "qwen3":                                       # per-head q/k RMSNorm before RoPE, no bias, D = 128, G = 2
        kw.update(num_attention_heads=4, num_key_value_heads=2, head_dim=128, hidden_size=512, intermediate_size=1024, qkv_bias=False, qk_norm=True)
    hip = _runner("hip", kw, monkeypatch)
    ora = _runner("oracle", kw, monkeypatch)
    for (na, pa), (nb, pb) in zip(hip.model.named_parameters(), ora.model.named_parameters()):
        assert na == nb and torch.equal(pa, pb)                 # same seed -> same weights
    assert type(ora.model.layers[0].self_attn.attn).__name__ == "OracleAttention"
    assert type(hip.model.layers[0].self_attn.attn).__module__.endswith("attention_hip")

    g = torch.Generator().manual_seed(11)
    lens = (300, 17, 256, 1)                                    # crosses a block, ends on a block boundary, single token
    tables = ([3, 7], [1], [5, 2], [9])                         # 256 -> 257 tokens needs a second block: booked up front
    steps = 3

    def mk():
        out = []
        gg = torch.Generator().manual_seed(11)
        for n, t in zip(lens, tables):
            s = Sequence(torch.randint(0, 2048, (n,), generator=gg).tolist())
            s.block_table = list(t)
            out.append(s)
        return out

    sa, sb = mk(), mk()
    worst_max = worst_mean = 0.0
    agree = total = 0
    for step in range(steps + 1):
        la, lb = _logits(hip, sa, step == 0, fused=fused), _logits(ora, sb, step == 0)
        diff = (la - lb).abs()
        rel_max = diff.max().item() / lb.abs().max().item()
        rel_mean = diff.mean().item() / lb.abs().mean().item()
        worst_max, worst_mean = max(worst_max, rel_max), max(worst_mean, rel_mean)
        print(f"[{family} fused={fused}] step {step}: rel max {rel_max:.3e} rel mean {rel_mean:.3e}")
        bar_max, bar_mean = (0.03, 2.0 ** -6) if fused and step > 0 else (2.0 ** -6, 2.0 ** -7)
        assert rel_max <= bar_max and rel_mean <= bar_mean, f"step {step}: logits differ beyond bf16 noise: max {rel_max:.3e} mean {rel_mean:.3e}"
        tok_a, tok_b = la.argmax(-1), lb.argmax(-1)
        top2 = lb.topk(2, dim=-1).values
        clear = (top2[:, 0] - top2[:, 1]) > 4 * diff.max()      # rows whose oracle-side decision is not inside the noise
        assert torch.equal(tok_a[clear], tok_b[clear])
        agree += int((tok_a == tok_b).sum())
        total += len(lens)
        for s_a, s_b, t in zip(sa, sb, tok_b.tolist()):         # both sides continue with the oracle side's token
            s_a.append_token(t)
            s_b.append_token(t)
    # the K/V rows the hip side stored (RoPE + store fused into one launch / the qkv GEMM epilogue) against the oracle module's caches
    for la_, lb_ in zip(hip.model.layers, ora.model.layers):
        kc = la_.self_attn.attn.k_cache.float().cpu().numpy()
        vc = la_.self_attn.attn.v_cache.float().cpu().numpy()
        ok, ov = lb_.self_attn.attn._kc, lb_.self_attn.attn._vc
        live = np.abs(ok).sum(axis=(2, 3)) > 0                   # slots the oracle side wrote
        assert live.sum() == sum(lens) + steps * len(lens)
        assert np.abs(kc - ok)[live].max() <= 2.0 ** -6 * np.abs(ok).max() and np.abs(vc - ov)[live].max() <= 2.0 ** -6 * np.abs(ov).max()
        assert not kc[~live].any() and not vc[~live].any()       # nothing was stored anywhere else
    print(f"[{family} fused={fused}] hip vs oracle attention: worst rel max {worst_max:.3e}, rel mean {worst_mean:.3e}, same token {agree}/{total}")
