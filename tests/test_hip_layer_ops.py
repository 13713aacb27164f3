"""GPU: fused residual-add + RMSNorm and SiLU*mul launches against the reference's arithmetic (layernorm.py:17-41,
activation.py:11-14) restated with torch ops in the same rounding order.  bf16 outputs may differ by one ulp where
rsqrt / exp differ in the last fp32 bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def ref_add_rms(x, w, eps, residual=None):
    x32 = x.float()
    if residual is not None:
        x32 = x32 + residual.float()
        residual = x32.to(x.dtype)
    var = x32.pow(2).mean(dim=-1, keepdim=True)
    y = (x32 * torch.rsqrt(var + eps)).to(x.dtype) * w
    return y, residual


def close_bf16(a, b, frac=0.02):
    a, b = a.float(), b.float()
    diff = (a - b).abs()
    assert (diff <= 2.0 ** -7 * b.abs() + 1e-30).all(), diff.max()
    assert (diff > 0).float().mean() < frac


@pytest.mark.parametrize("rows,hidden", [(32, 896), (1, 1024), (77, 3584), (5, 8192), (3, 64)])
@pytest.mark.parametrize("with_res", [False, True])
def test_add_rmsnorm(rows, hidden, with_res):
    from nanovllm_hip import ops
    g = torch.Generator().manual_seed(rows + hidden)
    x = torch.randn(rows, hidden, generator=g).bfloat16().cuda()
    w = (1 + 0.2 * torch.randn(hidden, generator=g)).bfloat16().cuda()
    res = torch.randn(rows, hidden, generator=g).bfloat16().cuda() if with_res else None
    y_ref, res_ref = ref_add_rms(x, w, 1e-6, res)
    res_in = res.clone() if with_res else None
    y = ops.add_rmsnorm(x, w, 1e-6, res_in)
    torch.cuda.synchronize()
    close_bf16(y, y_ref)
    if with_res:
        assert torch.equal(res_in, res_ref)                    # fp32 add, one rounding: bit-exact


@pytest.mark.parametrize("rows,inter", [(32, 4864), (2, 3072), (19, 18944), (1, 8)])
def test_silu_mul(rows, inter):
    from nanovllm_hip import ops
    g = torch.Generator().manual_seed(inter)
    gu = (2 * torch.randn(rows, 2 * inter, generator=g)).bfloat16().cuda()
    ref = torch.nn.functional.silu(gu[:, :inter]) * gu[:, inter:]
    out = ops.silu_mul(gu)
    torch.cuda.synchronize()
    close_bf16(out, ref, frac=0.05)


@pytest.mark.parametrize("max_tokens", [13, 270])
def test_engine_greedy_tokens_graph_equals_eager(max_tokens):
    """End to end through the engine: prefill + decode steps of a 2-layer Qwen2-shaped model; the device-resident
    HIP-graph session (metadata advanced on the device, next step's embedding looked up by the arg-max launch) must produce
    exactly the tokens of the eager, host-metadata path (model_runner.py:278-303).  270 steps carry every sequence across a
    block boundary (slot arithmetic of the device-side advance) and through hundreds of generation changes of the attention hand-off."""
    from nanovllm_hip.engine.llm_engine import LLMEngine
    from nanovllm_hip.models.qwen import model_config
    cfg = model_config("Qwen2-0.5B", num_hidden_layers=2, vocab_size=2048)
    g = torch.Generator().manual_seed(0)
    prompts = [torch.randint(0, 2048, (n,), generator=g).tolist() for n in (300, 17, 256, 5)]
    outs = []
    for eager in (True, False):
        eng = LLMEngine(cfg, num_kvcache_blocks=16, enforce_eager=eager, seed=1)
        outs.append(eng.generate(prompts, max_tokens=max_tokens))
    assert outs[0] == outs[1]
    assert all(len(o) == max_tokens for o in outs[0])


@pytest.mark.parametrize("m,n,k,bias", [(32, 1152, 896, True), (32, 896, 896, False), (32, 896, 4864, False), (1, 1024, 1024, False),
                                         (64, 576, 3584, True), (17, 151936, 896, False), (48, 32, 64, True)])
def test_linear_small_m(m, n, k, bias):
    """out = x W^T + b against an fp32 reference of the same bf16 operands (fp32 accumulate, one rounding)."""
    from nanovllm_hip import ops
    g = torch.Generator().manual_seed(m + n + k)
    x = torch.randn(m, k, generator=g).bfloat16().cuda()
    w = (torch.randn(n, k, generator=g) * 0.05).bfloat16().cuda()
    b = torch.randn(n, generator=g).bfloat16().cuda() if bias else None
    ref = torch.nn.functional.linear(x.float(), w.float(), b.float() if bias else None)
    out = ops.linear_small_m(x, w, b)
    torch.cuda.synchronize()
    err = (out.float() - ref).abs()
    assert (err <= 2.0 ** -8 * ref.abs() + 2e-3).all(), err.max()          # one bf16 rounding of the fp32 result (+ summation-order slack)


@pytest.mark.parametrize("m,inter,k", [(32, 4864, 896), (5, 608, 896), (64, 1536, 1024), (17, 4864, 896), (64, 4864, 896), (9, 4112, 1024)])
def test_linear_small_m_silu(m, inter, k):
    """gate_up projection + SiluAndMul in one launch == F.linear -> bf16 -> silu*mul (activation.py:11-14)."""
    from nanovllm_hip import ops
    g = torch.Generator().manual_seed(inter + k)
    x = torch.randn(m, k, generator=g).bfloat16().cuda()
    w = (torch.randn(2 * inter, k, generator=g) * 0.05).bfloat16().cuda()
    gu = torch.nn.functional.linear(x.float(), w.float()).bfloat16()
    ref = torch.nn.functional.silu(gu[:, :inter]) * gu[:, inter:]
    out = ops.linear_small_m(x, w, silu_mul=True)
    torch.cuda.synchronize()
    err = (out.float() - ref.float()).abs()
    # the fused kernel rounds the same fp32 sums to bf16; summation order can flip a rounding of gate or up (<= 1 ulp each)
    assert (err <= 2.0 ** -6 * ref.float().abs() + 2e-3).all(), err.max()
    assert (err > 2.0 ** -8 * ref.float().abs() + 1e-4).float().mean() < 0.05


def _ref_rms(x, w, eps):
    x32 = x.float()
    return (x32 * torch.rsqrt(x32.pow(2).mean(dim=-1, keepdim=True) + eps)).to(x.dtype) * w


def _close(a, b, rel=2.0 ** -6, abs_=3e-3, frac=0.05):
    a, b = a.float(), b.float()
    err = (a - b).abs()
    assert (err <= rel * b.abs() + abs_).all(), err.max()
    assert (err > 2.0 ** -8 * b.abs() + abs_ / 10).float().mean() < frac


@pytest.mark.parametrize("m,n,k", [(32, 896, 896), (32, 151936, 896), (7, 576, 3584)])
def test_fused_linear_norm_prologue(m, n, k):
    from nanovllm_hip import ops
    g = torch.Generator().manual_seed(m + n)
    x = torch.randn(m, k, generator=g).bfloat16().cuda()
    w = (torch.randn(n, k, generator=g) * 0.05).bfloat16().cuda()
    gw = (1 + 0.2 * torch.randn(k, generator=g)).bfloat16().cuda()
    ref = torch.nn.functional.linear(_ref_rms(x, gw, 1e-6).float(), w.float())
    out = ops.fused_linear(x, w, norm_weight=gw, norm_eps=1e-6)
    torch.cuda.synchronize()
    _close(out, ref)


@pytest.mark.parametrize("m,n,k", [(32, 896, 896), (32, 896, 4864), (3, 3584, 18944)])
def test_fused_linear_residual_add(m, n, k):
    from nanovllm_hip import ops
    g = torch.Generator().manual_seed(n + k)
    x = torch.randn(m, k, generator=g).bfloat16().cuda()
    w = (torch.randn(n, k, generator=g) * 0.02).bfloat16().cuda()
    res = torch.randn(m, n, generator=g).bfloat16().cuda()
    ref = torch.nn.functional.linear(x.float(), w.float()) + res.float()
    ops.fused_linear(x, w, epilogue="residual_add", out=res)
    torch.cuda.synchronize()
    _close(res, ref)


def test_fused_linear_silu_with_norm():
    from nanovllm_hip import ops
    g = torch.Generator().manual_seed(5)
    m, k, inter = 32, 896, 4864
    x = torch.randn(m, k, generator=g).bfloat16().cuda()
    w = (torch.randn(2 * inter, k, generator=g) * 0.05).bfloat16().cuda()
    gw = (1 + 0.2 * torch.randn(k, generator=g)).bfloat16().cuda()
    gu = torch.nn.functional.linear(_ref_rms(x, gw, 1e-6).float(), w.float()).bfloat16()
    ref = torch.nn.functional.silu(gu[:, :inter]) * gu[:, inter:]
    out = ops.fused_linear(x, w, norm_weight=gw, norm_eps=1e-6, epilogue="silu_mul")
    torch.cuda.synchronize()
    _close(out, ref, rel=2.0 ** -5)


@pytest.mark.parametrize("H,KVH,D", [(14, 2, 64), (7, 1, 128)])
def test_fused_linear_rope_store_equals_linear_then_rope_store(H, KVH, D):
    """qkv projection with the RoPE+store epilogue == skinny GEMM followed by nvh_rope_store (itself bit-exact against the
    reference's RoPE): q identical, cache rows identical, -1 slots skipped."""
    from nanovllm_hip import ops
    from nanovllm_hip.models.qwen import cos_sin_table
    g = torch.Generator().manual_seed(H)
    m, k, bs, nb = 32, 896, 256, 3
    n = (H + 2 * KVH) * D
    x = torch.randn(m, k, generator=g).bfloat16().cuda()
    w = (torch.randn(n, k, generator=g) * 0.05).bfloat16().cuda()
    b = torch.randn(n, generator=g).bfloat16().cuda()
    pos = torch.randint(0, 4096, (m,), generator=g).cuda()
    slots = torch.randperm(nb * bs, generator=g)[:m].int()
    slots[2::9] = -1
    slots = slots.cuda()
    table = cos_sin_table(D, 4096, 1e6, "cuda")
    kc1 = torch.randn(nb, bs, KVH, D, generator=g).bfloat16().cuda()
    vc1 = torch.randn(nb, bs, KVH, D, generator=g).bfloat16().cuda()
    kc2, vc2 = kc1.clone(), vc1.clone()
    qkv = ops.linear_small_m(x, w, b)
    ops.rope_store(qkv, pos, table, H, KVH, D, kc1, vc1, slots)
    q = ops.fused_linear(x, w, bias=b, epilogue="rope_store",
                         rope=dict(positions=pos, cos_sin=table, k_cache=kc2, v_cache=vc2, slot_mapping=slots, num_heads=H, num_kv_heads=KVH, head_dim=D))
    torch.cuda.synchronize()
    assert torch.equal(q, qkv[:, :H * D])
    assert torch.equal(kc1, kc2) and torch.equal(vc1, vc2)


def test_fused_decode_layer_path_close_to_unfused():
    """The 6-launch decode path (norm prologues, residual/SiLU/RoPE epilogues) against the 10-launch path on the same
    weights and cache state: logits agree to bf16 accumulation noise."""
    from nanovllm_hip.engine.llm_engine import LLMEngine
    from nanovllm_hip.engine.model_runner import build_decode_meta
    from nanovllm_hip.engine.sequence import Sequence
    from nanovllm_hip.models import qwen
    from nanovllm_hip import reset_context, set_context
    cfg = qwen.model_config("Qwen2-0.5B", num_hidden_layers=3, vocab_size=4096)
    g = torch.Generator().manual_seed(0)
    prompts = [torch.randint(0, 4096, (n,), generator=g).tolist() for n in (300, 40, 257)]
    logits = []
    for fused in (True, False):
        eng = LLMEngine(cfg, num_kvcache_blocks=12, seed=2)
        seqs = [Sequence(p, max_tokens=4) for p in prompts]
        eng.prefill(seqs, reserve_tokens=4)
        m = build_decode_meta(seqs)
        dev = lambda t: t.cuda()
        orig = qwen._fused_decode_ok
        if not fused:
            qwen._fused_decode_ok = lambda cfg_, x: False
        try:
            with torch.inference_mode():
                set_context(False, slot_mapping=dev(m["slot_mapping"]), context_lens=dev(m["context_lens"]), block_tables=dev(m["block_tables"]))
                h = eng.runner.model(dev(m["input_ids"]), dev(m["positions"]))
                logits.append(eng.runner.model.compute_logits(h).float().cpu())
                reset_context()
        finally:
            qwen._fused_decode_ok = orig
    scale = logits[1].abs().max()
    assert (logits[0] - logits[1]).abs().max() <= 3e-2 * scale
    assert (logits[0].argmax(-1) == logits[1].argmax(-1)).all()


@pytest.mark.parametrize("m,n", [(32, 151936), (1, 1000), (5, 152064), (3, 17)])
def test_argmax_rows(m, n):
    from nanovllm_hip import ops
    g = torch.Generator().manual_seed(n)
    x = torch.randn(m, n, generator=g).bfloat16()
    x[0, min(7, n - 1)] = 9.0
    x[0, min(11, n - 1)] = 9.0                                # a tie: the lowest index must win
    stride = (n + 7) // 8 * 8
    buf = torch.zeros(m, stride, dtype=torch.bfloat16)
    buf[:, :n] = x
    xd = buf.cuda()[:, :n]
    got = ops.argmax_rows(xd).cpu()
    vals = x.float()
    exp = torch.tensor([int((vals[i] == vals[i].max()).nonzero()[0]) for i in range(m)])
    assert torch.equal(got, exp)


@pytest.mark.parametrize("m,n,k,epi", [(32, 1152, 896, "none"), (32, 9728, 896, "silu_mul"), (9, 151936, 896, "none")])
def test_fused_linear_folded_norm(m, n, k, epi):
    """Folded RMSNorm (w := w*diag(g), row scale in the epilogue) against norm-then-project: same algebra, differs only by the
    reference's two bf16 roundings of the normalised activations and the rounding of g*W."""
    from nanovllm_hip import ops
    g = torch.Generator().manual_seed(k + n)
    x = (torch.randn(m, k, generator=g) * 3).bfloat16().cuda()
    w = (torch.randn(n, k, generator=g) * 0.05).bfloat16().cuda()
    gw = (1 + 0.2 * torch.randn(k, generator=g)).bfloat16().cuda()
    wf = (w.float() * gw.float().unsqueeze(0)).bfloat16().contiguous()
    y = torch.nn.functional.linear(_ref_rms(x, gw, 1e-6).float(), w.float())
    if epi == "silu_mul":
        y = y.bfloat16()
        ref = (torch.nn.functional.silu(y[:, :n // 2]) * y[:, n // 2:]).float()
    else:
        ref = y
    out = ops.fused_linear(x, wf, norm_folded=True, norm_eps=1e-6, epilogue=epi).float()
    torch.cuda.synchronize()
    err = (out - ref).abs()
    assert err.max() <= 0.03 * ref.abs().max()                      # bf16-level agreement of two valid roundings of the same math
    assert (err / (ref.abs() + 0.05 * ref.abs().max())).mean() < 0.01


# ---- streaming GEMM (csrc/linear_stream.hip): split-K hand-off, fragment-packed activations, multi-tile workgroups

def _zero_ws(nbytes):
    return torch.zeros(max(nbytes, 16), dtype=torch.uint8, device="cuda")


@pytest.mark.parametrize("m,n,k", [(32, 896, 4864), (3, 3584, 18944), (64, 512, 2048), (17, 64, 1088)])
def test_stream_linear_split_k_residual_add(m, n, k):
    """K > 1024 is split over workgroups; the last arriver sums the partials in split order: result within one bf16 rounding
    of the fp32 reference, bitwise identical across repeats, tickets left at zero."""
    from nanovllm_hip import ops
    g = torch.Generator().manual_seed(n + k)
    x = torch.randn(m, k, generator=g).bfloat16().cuda()
    w = (torch.randn(n, k, generator=g) * 0.02).bfloat16().cuda()
    res0 = torch.randn(m, n, generator=g).bfloat16().cuda()
    need = ops.linear_workspace_bytes(m, n, k, "residual_add")
    assert need > 0
    ws = _zero_ws(need)
    ref = torch.nn.functional.linear(x.float(), w.float()) + res0.float()
    outs = []
    for _ in range(3):
        res = res0.clone()
        ops.fused_linear(x, w, epilogue="residual_add", out=res, workspace=ws)
        outs.append(res)
    torch.cuda.synchronize()
    _close(outs[0], ref)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    tiles = n // 16
    assert int(ws[: tiles * 128].view(torch.int32).abs().sum()) == 0       # one ticket per 128-byte line, all back at zero


def test_stream_linear_split_k_silu_and_rope():
    """The non-linear epilogues run in the last arriver on the complete sums (Qwen2-7B tp=4 shapes: hidden 3584, heads 7/1 x 128)."""
    from nanovllm_hip import ops
    from nanovllm_hip.models.qwen import cos_sin_table
    g = torch.Generator().manual_seed(5)
    m, k, inter = 32, 3584, 1184
    x = torch.randn(m, k, generator=g).bfloat16().cuda()
    w = (torch.randn(2 * inter, k, generator=g) * 0.02).bfloat16().cuda()
    gu = torch.nn.functional.linear(x.float(), w.float()).bfloat16()
    ref = torch.nn.functional.silu(gu[:, :inter]) * gu[:, inter:]
    ws = _zero_ws(ops.linear_workspace_bytes(m, 2 * inter, k, "silu_mul"))
    out = ops.fused_linear(x, w, epilogue="silu_mul", workspace=ws)
    torch.cuda.synchronize()
    err = (out.float() - ref.float()).abs()
    assert (err <= 2.0 ** -6 * ref.float().abs() + 2e-3).all(), err.max()
    # rope_store: the split-K streaming kernel against the loop kernel (no workspace -> fallback), same rounding points
    H, KVH, D, bs = 7, 1, 128, 256
    n = (H + 2 * KVH) * D
    wq = (torch.randn(n, k, generator=g) * 0.02).bfloat16().cuda()
    b = torch.randn(n, generator=g).bfloat16().cuda()
    cs = cos_sin_table(D, 4096, 1e6, "cuda")
    pos = torch.randint(0, 4096, (m,), generator=g).cuda()
    slots = torch.randperm(4 * bs, generator=g)[:m].int().cuda()
    caches = [torch.zeros(2, 4, bs, KVH, D, dtype=torch.bfloat16, device="cuda") for _ in range(2)]
    qs = []
    for i, wsp in enumerate((None, _zero_ws(ops.linear_workspace_bytes(m, n, k, "rope_store")))):
        rope = dict(positions=pos, cos_sin=cs, k_cache=caches[i][0], v_cache=caches[i][1], slot_mapping=slots, num_heads=H, num_kv_heads=KVH, head_dim=D)
        qs.append(ops.fused_linear(x, wq, bias=b, epilogue="rope_store", rope=rope, workspace=wsp))
    torch.cuda.synchronize()
    _close(qs[1], qs[0].float(), rel=2.0 ** -7, abs_=2e-3)
    _close(caches[1], caches[0].float(), rel=2.0 ** -7, abs_=2e-3)


@pytest.mark.parametrize("m,n,k,epi", [(32, 896, 896, "residual_add"), (32, 896, 4864, "residual_add"), (20, 1152, 896, "none"), (32, 9728, 896, "silu_mul")])
def test_stream_linear_prefetch_hint_changes_nothing(m, n, k, epi):
    """nvh_linear_desc.prefetch is a hint: launches that carry it (extra workgroups on the idle CUs read the range) give bitwise the
    results of launches that do not, whatever the range looks like: unaligned start and length, shorter than two lines (ignored),
    a few MB (read), above 4 MiB (ignored).  With split-K the tickets still return to zero."""
    from nanovllm_hip import ops
    g = torch.Generator().manual_seed(n + k + 1)
    x = torch.randn(m, k, generator=g).bfloat16().cuda()
    w = (torch.randn(n, k, generator=g) * 0.02).bfloat16().cuda()
    res0 = torch.randn(m, n, generator=g).bfloat16().cuda()
    need = ops.linear_workspace_bytes(m, n, k, epi)
    ws = _zero_ws(max(need, 16))
    big = torch.zeros((6 << 20) + 3, dtype=torch.uint8, device="cuda")
    hints = [None, big[1:200], big[3: 3 + (1 << 20) + 77], big[64: 64 + (3 << 20)], big]

    def run(hint):
        res = res0.clone()
        kw = dict(epilogue=epi, workspace=ws if need else None, prefetch=hint)
        out = ops.fused_linear(x, w, out=res, **kw) if epi == "residual_add" else ops.fused_linear(x, w, **kw)
        torch.cuda.synchronize()
        return out.clone()
    outs = [run(h) for h in hints]
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    if need:
        assert int(ws[: (n // 16) * 128].view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize("m,n,k,epi", [(32, 9728, 896, "silu_mul"), (32, 896, 4864, "residual_add"), (20, 1152, 896, "none"), (32, 151936, 896, "none")])
def test_stream_linear_packed_activations(m, n, k, epi):
    """Fragment-packed x gives bit-identical results to row-major x (same arithmetic, different fetch), and out_packed is the
    row-major output in fragment order."""
    from nanovllm_hip import ops
    g = torch.Generator().manual_seed(n * 7 + k)
    x = torch.randn(m, k, generator=g).bfloat16().cuda()
    w = (torch.randn(n, k, generator=g) * 0.03).bfloat16().cuda()
    folded = epi != "residual_add"
    cols = n // 2 if epi == "silu_mul" else n
    ws = _zero_ws(ops.linear_workspace_bytes(m, n, k, epi))
    res0 = torch.randn(m, n, generator=g).bfloat16().cuda() if epi == "residual_add" else None
    kw = dict(norm_folded=folded, norm_eps=1e-6, epilogue=epi, workspace=ws)
    out_a = ops.fused_linear(x, w, out=res0.clone() if res0 is not None else None, **kw)
    packed = torch.zeros(((m + 15) // 16) * 16 * cols, dtype=torch.bfloat16, device="cuda")
    out_b = ops.fused_linear(ops.pack_rows(x), w, x_packed_rows=m, out=res0.clone() if res0 is not None else None, out_packed=packed, **kw)
    torch.cuda.synchronize()
    assert torch.equal(out_a, out_b)
    assert torch.equal(ops.unpack_rows(packed, m, cols), out_b)
    if epi in ("none", "silu_mul"):                       # packed-only output
        packed2 = torch.zeros_like(packed)
        assert ops.fused_linear(ops.pack_rows(x), w, x_packed_rows=m, out_packed=packed2, want_out=False, **kw) is None
        torch.cuda.synchronize()
        assert torch.equal(packed2, packed)


@pytest.mark.parametrize("m,n,k", [(32, 151936, 896), (5, 2048, 896), (17, 32000, 1024), (64, 151936, 896)])
def test_stream_linear_greedy_candidates(m, n, k):
    """LM head + arg-max candidates in one launch: reducing the candidate records gives exactly the arg-max (lowest index on
    ties) of the bf16 logits the same kernel writes, in both grid forms (one tile per workgroup / multi-tile workgroups)."""
    from nanovllm_hip import ops
    g = torch.Generator().manual_seed(n + k + m)
    x = torch.randn(m, k, generator=g).bfloat16().cuda()
    w = (torch.randn(n, k, generator=g) * 0.05).bfloat16().cuda()
    w[n - 7] = w[11]                                           # duplicate rows: exact ties between columns 11 and n-7
    w[n // 2] = w[3]
    groups = ops.linear_candidate_groups(n, k)
    assert groups > 0
    cv = torch.full((groups, 64), float("nan"), dtype=torch.float32, device="cuda")
    ci = torch.full((groups, 64), -1, dtype=torch.int32, device="cuda")
    logits = ops.fused_linear(x, w, norm_folded=True, norm_eps=1e-6, candidates=(cv, ci))
    torch.cuda.synchronize()
    ref = torch.stack([ops.argmax_rows(logits)]).squeeze(0).cpu()
    vals = cv[:, :m].cpu()
    idxs = ci[:, :m].cpu().long()
    for r in range(m):
        best = vals[:, r].max()
        cand = idxs[:, r][vals[:, r] == best].min()
        assert int(cand) == int(ref[r]) and float(best) == float(logits[r, cand])
    # candidates-only launch (no logits) + the advance kernel on a fake session state
    cv2, ci2 = torch.zeros_like(cv), torch.zeros_like(ci)
    assert ops.fused_linear(ops.pack_rows(x), w, x_packed_rows=m, norm_folded=True, norm_eps=1e-6, candidates=(cv2, ci2), want_out=False) is None
    bs = 256
    ctx = torch.randint(1, 500, (m,), generator=g).int()
    ctx[0] = 0                                                 # a padding row: untouched
    bt = torch.arange(m * 2, dtype=torch.int32).view(m, 2).cuda()
    ids = torch.zeros(m, dtype=torch.int64, device="cuda")
    pos = ctx.long().cuda()
    ctxd, slots = ctx.cuda(), torch.full((m,), -1, dtype=torch.int32, device="cuda")
    log = torch.zeros(4, m, dtype=torch.int64, device="cuda")
    steps = torch.zeros(m, dtype=torch.int64, device="cuda")
    ops.greedy_advance_candidates(cv2, ci2, groups, m, ids, pos, ctxd, slots, bt, bs, log, steps)
    torch.cuda.synchronize()
    live = ctx > 0
    assert torch.equal(ids.cpu()[live], ref[live]) and torch.equal(log[0].cpu()[live], ref[live])
    assert torch.equal(ctxd.cpu()[live], ctx[live] + 1) and int(ctxd[0]) == 0 and int(ids[0]) == 0
    exp_slot = bt.cpu()[torch.arange(m), (ctx // bs).long()] * bs + ctx % bs
    assert torch.equal(slots.cpu()[live], exp_slot[live].int())
    # the same launch with the next step's embedding lookup: identical bookkeeping, hidden rows = embed[token] (the padding row
    # keeps the token it had), row-major and in fragment order
    hid = 64 if n > 100000 else 96                             # small tables: [n, hid]
    emb = torch.randn(n, hid, generator=g).bfloat16().cuda()
    ids2 = torch.full((m,), 5, dtype=torch.int64, device="cuda")
    pos2, ctx2, slots2 = ctx.long().cuda(), ctx.cuda(), torch.full((m,), -1, dtype=torch.int32, device="cuda")
    log2, steps2 = torch.zeros(4, m, dtype=torch.int64, device="cuda"), torch.zeros(m, dtype=torch.int64, device="cuda")
    hout = torch.full((m, hid), float("nan"), dtype=torch.bfloat16, device="cuda")
    hpack = torch.zeros(((m + 15) // 16) * 16 * hid, dtype=torch.bfloat16, device="cuda")
    ops.greedy_advance_candidates(cv2, ci2, groups, m, ids2, pos2, ctx2, slots2, bt, bs, log2, steps2, embed=(emb, hout, hpack))
    torch.cuda.synchronize()
    assert torch.equal(ids2.cpu()[live], ids.cpu()[live]) and int(ids2[0]) == 5 and torch.equal(ctx2, ctxd) and torch.equal(slots2, slots)
    assert torch.equal(log2, log) and torch.equal(steps2, steps) and torch.equal(pos2.cpu()[live], pos.cpu()[live])
    assert torch.equal(hout, emb[ids2])
    assert torch.equal(hpack, ops.pack_rows(hout))


@pytest.mark.parametrize("name", ["Qwen2-0.5B", "Qwen3-0.6B"])
def test_fused_decode_layer_matches_plain_layer(name):
    """One decode step of a 2-layer model through the fused layer (norm-folded streaming GEMMs, packed activations, for Qwen3
    the separate q/k-norm+RoPE+store launch) against the plain layer-by-layer body: same logits up to bf16 rounding of the
    intermediate activations, same greedy tokens."""
    from nanovllm_hip.engine.llm_engine import LLMEngine
    from nanovllm_hip.engine.model_runner import build_decode_meta
    from nanovllm_hip.engine.sequence import Sequence
    from nanovllm_hip.models import qwen
    from nanovllm_hip.models.qwen import model_config
    from nanovllm_hip.utils.context import reset_context, set_context
    cfg = model_config(name, num_hidden_layers=2, vocab_size=4096)
    g = torch.Generator().manual_seed(7)
    prompts = [torch.randint(0, 4096, (n,), generator=g).tolist() for n in (300, 17, 256, 5, 129)]
    logits = {}
    for fused in (True, False):
        qwen.FUSED_DECODE = fused
        try:
            eng = LLMEngine(cfg, num_kvcache_blocks=16, enforce_eager=True, seed=3)
            seqs = [Sequence(p, max_tokens=4) for p in prompts]
            eng.prefill(seqs, reserve_tokens=4)
            r = eng.runner
            m = build_decode_meta(seqs, r.block_size)
            with torch.inference_mode():
                set_context(False, slot_mapping=r._dev(m["slot_mapping"]), context_lens=r._dev(m["context_lens"]), block_tables=r._dev(m["block_tables"]))
                hidden = r.model(r._dev(m["input_ids"]), r._dev(m["positions"]))
                logits[fused] = r.model.compute_logits(hidden).float().cpu()
                reset_context()
        finally:
            qwen.FUSED_DECODE = True
    a, b = logits[True], logits[False]
    scale = b.abs().max()
    assert (a - b).abs().max() <= 0.03 * scale
    assert torch.equal(a.argmax(-1), b.argmax(-1))


@pytest.mark.parametrize("m,hidden", [(32, 896), (5, 3584), (64, 1024)])
def test_residual_add_pack(m, hidden):
    """nvh_residual_add_pack: residual += y with one bf16 rounding (bit-exact vs torch on the same fp32 sum) and the updated
    rows in fragment order (the step after a tensor-parallel all-reduce in the fused decode layer)."""
    from nanovllm_hip import ops
    g = torch.Generator().manual_seed(m + hidden)
    buf = torch.randn(m, hidden + 64, generator=g).bfloat16().cuda()
    res = buf[:, :hidden]                                        # strided rows
    y = torch.randn(m, hidden, generator=g).bfloat16().cuda()
    exp = (res.float() + y.float()).bfloat16()
    packed = torch.zeros(((m + 15) // 16) * 16 * hidden, dtype=torch.bfloat16, device="cuda")
    ops.residual_add_pack(res, y, packed)
    torch.cuda.synchronize()
    assert torch.equal(res, exp)
    assert torch.equal(ops.unpack_rows(packed, m, hidden), exp)
    assert torch.equal(buf[:, hidden:], buf[:, hidden:])          # the padding columns were not touched (no NaN)
