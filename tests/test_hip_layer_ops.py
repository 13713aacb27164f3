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


def test_engine_greedy_tokens_graph_equals_eager():
    """End to end through the engine: prefill + 12 decode steps of a 2-layer Qwen2-shaped model; the device-resident
    HIP-graph session must produce exactly the tokens of the eager, host-metadata path (model_runner.py:278-303)."""
    from nanovllm_hip.engine.llm_engine import LLMEngine
    from nanovllm_hip.models.qwen import model_config
    cfg = model_config("Qwen2-0.5B", num_hidden_layers=2, vocab_size=2048)
    g = torch.Generator().manual_seed(0)
    prompts = [torch.randint(0, 2048, (n,), generator=g).tolist() for n in (300, 17, 256, 5)]
    outs = []
    for eager in (True, False):
        eng = LLMEngine(cfg, num_kvcache_blocks=16, enforce_eager=eager, seed=1)
        outs.append(eng.generate(prompts, max_tokens=13))
    assert outs[0] == outs[1]
    assert all(len(o) == 13 for o in outs[0])


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


@pytest.mark.parametrize("m,inter,k", [(32, 4864, 896), (5, 608, 896), (64, 1536, 1024)])
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
