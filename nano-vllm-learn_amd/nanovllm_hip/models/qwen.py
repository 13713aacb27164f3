"""Qwen2 / Qwen3 decoder bodies on PyTorch-ROCm, just enough to drive the attention path end to end.

Only the attention op is this package's product; everything else here (GEMMs through hipBLASLt, norms,
RoPE, SwiGLU) is plumbing that mirrors the reference model so the attention module is called exactly the
way nanovllm/models/qwen3.py calls it:
  * backend dispatch by string (qwen3.py:44-56) through nanovllm_hip.resolve_attention
  * positional construction `Attention(num_heads, head_dim, scale, num_kv_heads, **kw)` (qwen3.py:89-95)
  * qkv_proj -> split -> (per-head q/k RMSNorm, Qwen3 only) -> RoPE -> attn(q, k, v) -> o_proj (qwen3.py:99-119);
    v reaches the attention module as a strided view of the fused qkv output
  * tensor parallel = head split (qwen3.py:30-39) with an all-reduce after o_proj and down_proj
    (layers/linear.py:185-190); embedding and LM head are kept replicated (a simplification: every rank
    computes the same greedy token, so no gather is needed).
Weights are random (N(0, 0.02), seed fixed): the container has no checkpoints and no network.
The reference hard-codes Qwen3 (model_runner.py:37); Qwen2 shapes (BASELINE.json) need bias on qkv and no
q/k norm, which `qk_norm=False, qkv_bias=True` selects.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.distributed as dist
import torch.nn.functional as F
from torch import nn

from ..config import resolve_attention


@dataclass
class ModelConfig:
    name: str
    num_hidden_layers: int
    hidden_size: int
    num_attention_heads: int
    num_key_value_heads: int
    head_dim: int
    intermediate_size: int
    vocab_size: int
    tie_word_embeddings: bool
    qkv_bias: bool
    qk_norm: bool
    rope_theta: float = 1000000.0
    rms_norm_eps: float = 1e-6
    max_position_embeddings: int = 32768
    attn_backend: str = "hip"
    kvcache_block_size: int = 256


# public HF config.json values
SHAPES = {
    "Qwen2-0.5B": dict(num_hidden_layers=24, hidden_size=896, num_attention_heads=14, num_key_value_heads=2, head_dim=64,
                       intermediate_size=4864, vocab_size=151936, tie_word_embeddings=True, qkv_bias=True, qk_norm=False),
    "Qwen2-7B": dict(num_hidden_layers=28, hidden_size=3584, num_attention_heads=28, num_key_value_heads=4, head_dim=128,
                     intermediate_size=18944, vocab_size=152064, tie_word_embeddings=False, qkv_bias=True, qk_norm=False),
    "Qwen3-0.6B": dict(num_hidden_layers=28, hidden_size=1024, num_attention_heads=16, num_key_value_heads=8, head_dim=128,
                       intermediate_size=3072, vocab_size=151936, tie_word_embeddings=True, qkv_bias=False, qk_norm=True),
}


def model_config(name: str, **overrides) -> ModelConfig:
    return ModelConfig(name=name, **{**SHAPES[name], **overrides})


def _tp():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def tp_all_reduce(x):
    """The all-reduce of RowParallelLinear.forward (layers/linear.py:185-190), in place: the one-shot IPC kernel when the process
    chose it at start-up and the message fits its staging buffers (decode sizes), RCCL through torch.distributed otherwise."""
    from ..distributed import tensor_parallel_comm
    comm = tensor_parallel_comm()
    if comm is not None and comm.fits(x):
        return comm.all_reduce(x)
    dist.all_reduce(x)
    return x


def tp_all_reduce_residual_add_pack(y, residual, packed):
    """all-reduce(y) -> residual += sum -> fragment-packed copy: ONE launch on the one-shot path, RCCL + nvh_residual_add_pack
    otherwise (same rounding points: the sum is rounded to bf16 once, then added to the residual stream in fp32)."""
    from .. import ops
    from ..distributed import tensor_parallel_comm
    comm = tensor_parallel_comm()
    if comm is not None and comm.fits(y):
        return comm.all_reduce_residual_add(y, residual, packed)
    dist.all_reduce(y)
    return ops.residual_add_pack(residual, y, packed)


def tp_partition(num_heads, num_kv_heads, tp, rank):
    """Head ranges of one tensor-parallel rank: (q_start, q_count, kv_start, kv_count).

    The reference splits both head counts evenly and asserts divisibility (models/qwen3.py:30-36).  When that holds
    this is exactly its split.  When tp exceeds what the kv heads allow (Qwen2-0.5B: 14/2 heads at tp=4/8, Qwen2-7B:
    28/4 at tp=8) every kv head is REPLICATED on tp/KVH ranks and the G = H/KVH query heads that
    share it are divided among those ranks as evenly as possible (sizes differ by at most one).  Each q head still
    attends exactly its own kv head and o_proj is row-sliced by the same q heads, so the summed result is unchanged;
    it departs from the reference only in lifting its divisibility assert."""
    if num_heads % tp == 0 and num_kv_heads % tp == 0:
        return rank * (num_heads // tp), num_heads // tp, rank * (num_kv_heads // tp), num_kv_heads // tp
    assert tp % num_kv_heads == 0, f"tp={tp} must be a multiple of num_kv_heads={num_kv_heads} when it does not divide it"
    per_kv = tp // num_kv_heads                      # ranks sharing one kv head
    group = num_heads // num_kv_heads
    assert group >= per_kv, f"group of {group} q heads cannot be split over {per_kv} ranks"
    kv, sub = rank // per_kv, rank % per_kv
    base, extra = divmod(group, per_kv)
    return kv * group + sub * base + min(sub, extra), base + (sub < extra), kv, 1


class RMSNorm(nn.Module):
    """RMSNorm with the reference's optional fused residual add (layers/layernorm.py:43-50): forward(x) -> y, or
    forward(x, residual) -> (y, x + residual).  On the GPU it is one HIP launch (nvh_add_rmsnorm); the torch form below
    has the same rounding points and serves CPU-side tests of the model plumbing."""

    def __init__(self, size, eps):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(size))

    def forward(self, x, residual=None):
        if x.is_cuda and x.dtype == torch.bfloat16 and x.shape[-1] % 8 == 0 and x.shape[-1] <= 8192:
            from .. import ops
            if residual is None:
                return ops.add_rmsnorm(x, self.weight, self.eps)
            return ops.add_rmsnorm(x, self.weight, self.eps, residual), residual        # residual updated in place
        x32 = x.float()
        if residual is not None:
            x32 = x32 + residual.float()
            residual = x32.to(x.dtype)
        y = (x32 * torch.rsqrt(x32.pow(2).mean(dim=-1, keepdim=True) + self.eps)).to(x.dtype) * self.weight
        return y if residual is None else (y, residual)


def linear(x, weight, bias=None):
    """F.linear; decode-sized batches on the GPU take the weight-streaming HIP kernel (ops.linear_small_m), anything
    else (prefill, CPU tests) goes to the vendor GEMM."""
    if x.is_cuda and x.dtype == torch.bfloat16 and x.numel() // x.shape[-1] <= 64 and x.shape[-1] % 64 == 0 and weight.shape[0] % 16 == 0:
        from .. import ops
        return ops.linear_small_m(x, weight, bias)
    return F.linear(x, weight, bias)


def silu_and_mul(gate_up):
    """SiluAndMul (layers/activation.py:11-14); one HIP launch on the GPU."""
    if gate_up.is_cuda and gate_up.dtype == torch.bfloat16 and gate_up.shape[-1] % 16 == 0:
        from .. import ops
        return ops.silu_mul(gate_up)
    gate, up = gate_up.chunk(2, dim=-1)
    return F.silu(gate) * up


_COS_SIN = {}


def cos_sin_table(head_dim, max_position, base, device):
    """fp32 [max_position, head_dim] = cos | sin, one copy per device shared by all layers
    (the reference's cos_sin_cache, layers/rotary_embedding.py:29-36; kept fp32 regardless of the model dtype)."""
    key = (head_dim, max_position, float(base), str(device))
    if key not in _COS_SIN:
        inv_freq = 1.0 / (base ** (torch.arange(0, head_dim, 2, dtype=torch.float) / head_dim))
        freqs = torch.outer(torch.arange(max_position, dtype=torch.float), inv_freq)
        _COS_SIN[key] = torch.cat((freqs.cos(), freqs.sin()), dim=-1).to(device)
    return _COS_SIN[key]


class RotaryEmbedding(nn.Module):
    """Neox-style RoPE with a precomputed cos/sin table, fp32 math (layers/rotary_embedding.py:6-55)."""

    def __init__(self, head_dim, max_position, base):
        super().__init__()
        self.head_dim, self.max_position, self.base = head_dim, max_position, base

    def table(self, device):
        return cos_sin_table(self.head_dim, self.max_position, self.base, device)

    def _rot(self, x, cos, sin):
        x1, x2 = torch.chunk(x.float(), 2, dim=-1)
        return torch.cat((x1 * cos - x2 * sin, x2 * cos + x1 * sin), dim=-1).to(x.dtype)

    def forward(self, positions, q, k):
        cos, sin = self.table(q.device)[positions].unsqueeze(-2).chunk(2, dim=-1)
        n = positions.shape[0]
        q = self._rot(q.view(n, -1, self.head_dim), cos, sin).view(q.shape)
        k = self._rot(k.view(n, -1, self.head_dim), cos, sin).view(k.shape)
        return q, k


class QwenAttention(nn.Module):
    def __init__(self, cfg: ModelConfig):
        super().__init__()
        rank, tp = _tp()
        _, self.num_heads, _, self.num_kv_heads = tp_partition(cfg.num_attention_heads, cfg.num_key_value_heads, tp, rank)
        self.head_dim = cfg.head_dim
        self.q_size = self.num_heads * self.head_dim
        self.kv_size = self.num_kv_heads * self.head_dim
        self.scaling = self.head_dim ** -0.5
        attn_cls, attn_kwargs = resolve_attention(cfg.attn_backend, cfg.kvcache_block_size)       # qwen3.py:44-56
        self.qkv_proj = nn.Linear(cfg.hidden_size, self.q_size + 2 * self.kv_size, bias=cfg.qkv_bias)
        self.o_proj = nn.Linear(self.q_size, cfg.hidden_size, bias=False)
        self.rotary_emb = RotaryEmbedding(self.head_dim, cfg.max_position_embeddings, cfg.rope_theta)
        self.attn = attn_cls(self.num_heads, self.head_dim, self.scaling, self.num_kv_heads, **attn_kwargs)   # qwen3.py:89-95
        self.qk_norm = cfg.qk_norm
        if cfg.qk_norm:
            self.q_norm = RMSNorm(self.head_dim, cfg.rms_norm_eps)
            self.k_norm = RMSNorm(self.head_dim, cfg.rms_norm_eps)

    def forward(self, positions, hidden_states):
        qkv = linear(hidden_states, self.qkv_proj.weight, self.qkv_proj.bias)
        if hasattr(self.attn, "rope_store_attend"):              # hip backend: norm -> RoPE -> store fused into one launch
            o = self.attn.rope_store_attend(qkv, positions, self.rotary_emb.table(qkv.device),
                                            self.q_norm.weight if self.qk_norm else None,
                                            self.k_norm.weight if self.qk_norm else None, self.q_norm.eps if self.qk_norm else 1e-6)
            out = linear(o, self.o_proj.weight)
            if _tp()[1] > 1:
                tp_all_reduce(out)
            return out
        q, k, v = qkv.split([self.q_size, self.kv_size, self.kv_size], dim=-1)
        if self.qk_norm:
            q = self.q_norm(q.reshape(-1, self.num_heads, self.head_dim).contiguous()).view(-1, self.q_size)
            k = self.k_norm(k.reshape(-1, self.num_kv_heads, self.head_dim).contiguous()).view(-1, self.kv_size)
        q, k = self.rotary_emb(positions, q, k)
        o = self.attn(q, k, v)                                   # qwen3.py:117 — the hot path
        out = linear(o, self.o_proj.weight)
        if _tp()[1] > 1:
            tp_all_reduce(out)                                   # layers/linear.py:188-189 (one-shot over xGMI, or RCCL)
        return out


class QwenMLP(nn.Module):
    def __init__(self, cfg: ModelConfig):
        super().__init__()
        _, tp = _tp()
        assert cfg.intermediate_size % tp == 0
        self.inter = cfg.intermediate_size // tp
        # a shard that is not a multiple of 64 columns (Qwen2-0.5B at tp = 8: 608) is padded with zero weights to the next one:
        # SiLU(0) * 0 = 0 meets zero down_proj columns, the result is unchanged, and the streaming GEMMs (whole 128-byte weight
        # lines per K piece) and with them the fused decode layer apply to every rank count
        inter = self.inter if tp == 1 else (self.inter + 63) // 64 * 64
        self.gate_up_proj = nn.Linear(cfg.hidden_size, 2 * inter, bias=False)
        self.down_proj = nn.Linear(inter, cfg.hidden_size, bias=False)

    def forward(self, x):
        w = self.gate_up_proj.weight
        if x.is_cuda and x.dtype == torch.bfloat16 and x.numel() // x.shape[-1] <= 64 and x.shape[-1] % 64 == 0 and w.shape[0] % 32 == 0:
            from .. import ops
            act = ops.linear_small_m(x, w, silu_mul=True)         # gate_up projection + SiLU*mul in one launch
        else:
            act = silu_and_mul(F.linear(x, w))
        out = linear(act, self.down_proj.weight)
        if _tp()[1] > 1:
            tp_all_reduce(out)
        return out


class QwenDecoderLayer(nn.Module):
    def __init__(self, cfg: ModelConfig):
        super().__init__()
        self.self_attn = QwenAttention(cfg)
        self.mlp = QwenMLP(cfg)
        self.input_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps)
        self.post_attention_layernorm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps)

    def forward(self, positions, hidden_states, residual):
        # the reference's residual flow (models/qwen3.py:179-193): every norm also performs the pending residual add
        if residual is None:
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
        else:
            hidden_states, residual = self.input_layernorm(hidden_states, residual)
        hidden_states = self.self_attn(positions, hidden_states)
        hidden_states, residual = self.post_attention_layernorm(hidden_states, residual)
        return self.mlp(hidden_states), residual


FUSED_DECODE = True          # tests flip this to compare the fused decode layer with the plain one
KV_AHEAD = None              # A/B runs only (bench.py --kv-ahead): "o_k+gu_v" / "o_v+gu_k" / "o_k" — the o_proj / gate_up launches of layer i touch layer i + 1's whole
                             # K / V cache regions (plain pointer prefetch: only meaningful when the cache holds little more than the live blocks, as in bench.py)
PREFETCH_WEIGHTS = True      # qkv / down launches pull the NEXT small projection's weights into the caches on their idle CUs (A/B runs flip it)
QKV_ATTEND_MODE = "two_launches"   # how nvh_qkv_rope_attend runs the front of a decode layer (ops.QKV_ATTEND_MODES; A/B runs change it)
KV_PREFETCH_PASSES = 1             # mode "two_launches_kv_prefetch": first passes of every attention workgroup the qkv launch touches


class PackedResidual:
    """What the fused decode path hands from forward() to compute_logits() / greedy_candidates(): the UN-normalised residual
    stream (row-major) and the same rows in fragment order.  The final RMSNorm rides in the LM head's launch, so there is no
    normalised hidden state to return; returning this object (instead of stashing the packed buffer on the module) keeps a
    forward whose logits are never asked for from leaking into the next call."""
    __slots__ = ("residual", "packed")

    def __init__(self, residual, packed):
        self.residual, self.packed = residual, packed

    @property
    def shape(self):
        return self.residual.shape

    @property
    def device(self):
        return self.residual.device


def _fused_decode_ok(cfg, x):
    """The 5-launch decoder layer (6 with Qwen3's q/k-norm) applies to single-GPU decode batches of at most 64 rows."""
    from ..utils.context import get_context
    ctx = get_context()
    return (FUSED_DECODE and x.is_cuda and x.dtype == torch.bfloat16 and x.shape[0] <= 64 and not ctx.is_prefill
            and ctx.context_lens is not None and ctx.slot_mapping is not None and cfg.attn_backend == "hip")


class QwenForCausalLM(nn.Module):
    def __init__(self, cfg: ModelConfig):
        super().__init__()
        self.cfg = cfg
        self.embed_tokens = nn.Embedding(cfg.vocab_size, cfg.hidden_size)
        self.layers = nn.ModuleList([QwenDecoderLayer(cfg) for _ in range(cfg.num_hidden_layers)])
        self.norm = RMSNorm(cfg.hidden_size, cfg.rms_norm_eps)
        if not cfg.tie_word_embeddings:
            self.lm_head = nn.Linear(cfg.hidden_size, cfg.vocab_size, bias=False)

    def forward(self, input_ids, positions, embedded=None):
        """embedded = (hidden [m, hidden], packed) — fused decode path only: the token embeddings already looked up by the
        previous step's nvh_greedy_advance_candidates_embed, row-major (it becomes the residual stream, updated in place) and in
        fragment order (what layer 0's projection reads); input_ids is then not touched."""
        if embedded is not None:
            assert _fused_decode_ok(self.cfg, embedded[0]) and self._fused_shapes_ok()
            return self._forward_decode_fused(embedded[0], positions, packed0=embedded[1])
        h, residual = self.embed_tokens(input_ids), None
        if _fused_decode_ok(self.cfg, h) and self._fused_shapes_ok():
            return self._forward_decode_fused(h, positions)
        for layer in self.layers:
            h, residual = layer(positions, h, residual)
        h, _ = self.norm(h, residual)
        return h

    def _fused_shapes_ok(self):
        """Every contraction length of this rank's projections must be a multiple of 64 (one 128-byte line per weight row per
        DMA piece of the streaming GEMM); QwenMLP pads its shard to that (Qwen2-0.5B at tp=8: 608 -> 640 columns)."""
        a, mlp = self.layers[0].self_attn, self.layers[0].mlp
        return self.cfg.hidden_size % 64 == 0 and a.q_size % 64 == 0 and mlp.down_proj.weight.shape[1] % 64 == 0

    def _folded_weights(self):
        """RMSNorm weights multiplied into the projections that consume the normalised activations (w := w * diag(g)),
        built once: RMSNorm(x) @ W^T == (x @ (g*W)^T) * rsqrt(mean(x^2) + eps), which nvh_linear_small_m_ex evaluates with the
        row scale in its epilogue — no separate norm launch and no second pass over x."""
        if getattr(self, "_folded", None) is None:
            with torch.no_grad():
                def fold(w, g):
                    return (w.float() * g.float().unsqueeze(0)).to(torch.bfloat16).contiguous()
                head = self.embed_tokens.weight if self.cfg.tie_word_embeddings else self.lm_head.weight
                self._folded = dict(
                    qkv=[fold(l.self_attn.qkv_proj.weight, l.input_layernorm.weight) for l in self.layers],
                    gate_up=[fold(l.mlp.gate_up_proj.weight, l.post_attention_layernorm.weight) for l in self.layers],
                    head=fold(head, self.norm.weight))
        return self._folded

    def _decode_buffers(self, m, device):
        """Per batch-size scratch of the fused decode path, allocated once (before graph capture) and reused: the residual
        stream and the MLP activation in MFMA-fragment order (what the streaming GEMM reads at full address rate), and the
        zero-filled split-K workspace (tickets return to zero after every launch)."""
        from .. import ops
        key = (m, str(device))
        bufs = self.__dict__.setdefault("_decode_bufs", {})
        if key not in bufs:
            cfg, rows = self.cfg, ((m + 15) // 16) * 16
            inter = self.layers[0].mlp.down_proj.weight.shape[1]
            head = self.embed_tokens.weight if cfg.tie_word_embeddings else self.lm_head.weight
            need = max(ops.linear_workspace_bytes(m, head.shape[0], cfg.hidden_size, "none"),
                       ops.linear_workspace_bytes(m, cfg.hidden_size, cfg.hidden_size, "residual_add"),
                       ops.linear_workspace_bytes(m, cfg.hidden_size, inter, "residual_add"),
                       ops.linear_workspace_bytes(m, self.layers[0].self_attn.qkv_proj.weight.shape[0], cfg.hidden_size, "rope_store"),
                       ops.linear_workspace_bytes(m, 2 * inter, cfg.hidden_size, "silu_mul"), 16)
            bufs[key] = dict(resid_p=torch.zeros(rows * cfg.hidden_size, dtype=torch.bfloat16, device=device),
                             act_p=torch.zeros(rows * inter, dtype=torch.bfloat16, device=device),
                             attn_p=torch.zeros(rows * self.layers[0].self_attn.o_proj.weight.shape[1], dtype=torch.bfloat16, device=device),
                             y=torch.zeros((m, cfg.hidden_size), dtype=torch.bfloat16, device=device),     # TP: partial sums to all-reduce
                             ws=torch.zeros(need, dtype=torch.uint8, device=device))
        return bufs[key]

    def _forward_decode_fused(self, residual, positions, packed0=None):
        """Decode step with 6 launches per layer (qkv, attention split + combine, o_proj, gate_up, down): the norms ride in the
        projections (folded weights + epilogue row scale), residual adds / SiLU*mul / RoPE+store are GEMM epilogues.
        `residual` is the running residual stream (the embedding output, updated in place); every GEMM that updates it also
        writes it in fragment order for the next GEMM (csrc/linear_stream.hip), and the MLP activation exists only in that
        order.  The final norm belongs to the LM head's launch, so this returns the un-normalised stream as a PackedResidual."""
        from .. import ops
        from ..utils.context import get_context
        ctx = get_context()
        fw = self._folded_weights()
        m = residual.shape[0]
        b = self._decode_buffers(m, residual.device)
        resid_p, act_p, attn_p, ws, ybuf = b["resid_p"], b["act_p"], b["attn_p"], b["ws"], b["y"]
        tp = _tp()[1]
        # weight prefetch (nvh_linear_desc.prefetch): the qkv launch runs 36 workgroups and the o_proj launch 56 on 256 CUs, each
        # pulling 28-57 KB of weights whose first-byte latency is most of the launch.  The launch right before each of them (down of the
        # previous layer, qkv) lends its idle CUs to read those weights first: -3.3 % on the decode step.  The same for the two large
        # projections does not pay (gate_up's 17 MB in the o_proj launch: +1.3 %; down's in the gate_up launch: 0), nor does a
        # prefetch two launches ahead
        nxt = list(self.layers)[1:] + [None]
        for i, layer in enumerate(self.layers):
            a, mlp = layer.self_attn, layer.mlp
            pf_in_qkv = a.o_proj.weight if PREFETCH_WEIGHTS else None
            pf_in_dn = fw["qkv"][i + 1] if PREFETCH_WEIGHTS and nxt[i] is not None else None
            pf_in_o = pf_in_gu = None
            if KV_AHEAD and nxt[i] is not None:
                nk, nv = nxt[i].self_attn.attn.k_cache, nxt[i].self_attn.attn.v_cache
                if nk.numel():
                    pf_in_o, pf_in_gu = {"o_k+gu_v": (nk, nv), "o_v+gu_k": (nv, nk), "o_k": (nk, None), "gu_k": (None, nk)}[KV_AHEAD]
            if i == 0:                                                    # layer 0 reads the embedding rows as they are, or packed
                x, xrows = (residual, None) if packed0 is None else (packed0, m)
            else:
                x, xrows = resid_p, m
            if a.qk_norm:
                # Qwen3: the per-head q/k RMSNorm needs a whole head in one workgroup, so it cannot ride in the GEMM epilogue:
                # plain projection, then (q/k-norm -> RoPE -> KV store) as the one nvh_rope_store launch (qwen3.py:108-116)
                qkv = ops.fused_linear(x, fw["qkv"][i], x_packed_rows=xrows, bias=a.qkv_proj.bias, norm_folded=True,
                                       norm_eps=layer.input_layernorm.eps, epilogue="none", workspace=ws, prefetch=pf_in_qkv)
                ops.rope_store(qkv, positions, a.rotary_emb.table(residual.device), a.num_heads, a.num_kv_heads, a.head_dim,
                               a.attn.k_cache, a.attn.v_cache, ctx.slot_mapping, a.q_norm.weight, a.k_norm.weight, a.q_norm.eps)
                q = qkv[:, :a.q_size]
            elif xrows is not None:
                # qkv projection (+ folded norm, bias, RoPE, K/V store) AND the decode attention on its q rows: one launch whose attention
                # workgroups stream K/V from the first microsecond and wait only for q and the newest cache row (csrc/qkv_attend.hip)
                a.attn.qkv_rope_store_attend(x, xrows, fw["qkv"][i], a.qkv_proj.bias, layer.input_layernorm.eps, positions,
                                             a.rotary_emb.table(residual.device), out_packed=attn_p, prefetch=pf_in_qkv,
                                             mode=QKV_ATTEND_MODE, linear_workspace=ws,
                                             kv_prefetch_passes=KV_PREFETCH_PASSES if QKV_ATTEND_MODE == "two_launches_kv_prefetch" else 0)
            else:
                q = ops.fused_linear(x, fw["qkv"][i], x_packed_rows=xrows, bias=a.qkv_proj.bias, norm_folded=True,
                                     norm_eps=layer.input_layernorm.eps, epilogue="rope_store", workspace=ws, prefetch=pf_in_qkv,
                                     rope=dict(positions=positions, cos_sin=a.rotary_emb.table(residual.device), k_cache=a.attn.k_cache,
                                               v_cache=a.attn.v_cache, slot_mapping=ctx.slot_mapping, num_heads=a.num_heads,
                                               num_kv_heads=a.num_kv_heads, head_dim=a.head_dim))
            if a.qk_norm or xrows is None:
                a.attn.decode_attend(q, out_packed=attn_p)
            if tp == 1:
                ops.fused_linear(attn_p, a.o_proj.weight, x_packed_rows=m, epilogue="residual_add", out=residual, out_packed=resid_p,
                                 workspace=ws, prefetch=pf_in_o)
            else:
                # tensor parallel: this rank's heads give a partial sum (bf16, as RowParallelLinear, layers/linear.py:185-190);
                # one-shot all-reduce over IPC-mapped peer buffers with the residual add + fragment-packed copy in the SAME launch
                # (nvh_allreduce_oneshot), or RCCL followed by one small launch when the one-shot path is not set up
                ops.fused_linear(attn_p, a.o_proj.weight, x_packed_rows=m, epilogue="none", out=ybuf, workspace=ws)
                tp_all_reduce_residual_add_pack(ybuf, residual, resid_p)
            ops.fused_linear(resid_p, fw["gate_up"][i], x_packed_rows=m, norm_folded=True, norm_eps=layer.post_attention_layernorm.eps,
                             epilogue="silu_mul", out_packed=act_p, want_out=False, workspace=ws, prefetch=pf_in_gu)
            if tp == 1:
                ops.fused_linear(act_p, mlp.down_proj.weight, x_packed_rows=m, epilogue="residual_add", out=residual, out_packed=resid_p,
                                 workspace=ws, prefetch=pf_in_dn)
            else:
                ops.fused_linear(act_p, mlp.down_proj.weight, x_packed_rows=m, epilogue="none", out=ybuf, workspace=ws)
                tp_all_reduce_residual_add_pack(ybuf, residual, resid_p)
        return PackedResidual(residual, resid_p)

    def compute_logits(self, hidden_states):
        w = self.embed_tokens.weight if self.cfg.tie_word_embeddings else self.lm_head.weight
        if isinstance(hidden_states, PackedResidual):                 # fused decode path: final RMSNorm in the LM-head launch
            from .. import ops
            m = hidden_states.shape[0]
            return ops.fused_linear(hidden_states.packed, self._folded_weights()["head"], x_packed_rows=m, norm_folded=True, norm_eps=self.norm.eps,
                                    workspace=self._decode_buffers(m, hidden_states.device)["ws"])
        return linear(hidden_states, w)

    def fused_embedding_ok(self, m, device):
        """DecodeSession asks: may the next step's embedding lookup ride in the arg-max launch?  (single rank: the reference's
        vocab-parallel embedding needs an all-reduce otherwise; and the fused decode layer must apply)"""
        from .. import ops
        return (FUSED_DECODE and _tp()[1] == 1 and self.cfg.attn_backend == "hip" and device.type == "cuda" and m <= 64
                and self.embed_tokens.weight.dtype == torch.bfloat16 and self.cfg.hidden_size % 32 == 0 and self._fused_shapes_ok()
                and ops.linear_candidate_groups(self.cfg.vocab_size, self.cfg.hidden_size) > 0)   # the arg-max must be the candidates launch

    def greedy_candidates(self, hidden_states):
        """Fused decode path only: LM head + per-workgroup arg-max candidates in one launch, logits never written.  Returns
        (val [groups, stride] float32, idx int32, groups) for ops.greedy_advance_candidates, or None when this path does not
        apply (then use compute_logits)."""
        if not isinstance(hidden_states, PackedResidual):
            return None
        packed = hidden_states.packed
        from .. import ops
        head = self._folded_weights()["head"]
        n, k = head.shape
        groups = ops.linear_candidate_groups(n, k)
        if groups <= 0:
            return None
        m = hidden_states.shape[0]
        b = self._decode_buffers(m, hidden_states.device)
        if "cand" not in b:
            b["cand"] = (torch.empty((groups, 64), dtype=torch.float32, device=hidden_states.device),
                         torch.empty((groups, 64), dtype=torch.int32, device=hidden_states.device))
        ops.fused_linear(packed, head, x_packed_rows=m, norm_folded=True, norm_eps=self.norm.eps, candidates=b["cand"], want_out=False)
        return b["cand"][0], b["cand"][1], groups

    @torch.no_grad()
    def init_random(self, seed=0):
        """Deterministic N(0, 0.02) weights; every rank draws the full tensors and keeps its shard so that
        tp=N equals tp=1 on the same seed."""
        gen = torch.Generator(device="cpu").manual_seed(seed)
        rank, tp = _tp()
        cfg = self.cfg

        def draw(*shape):
            return torch.randn(*shape, generator=gen) * 0.02

        q0, qn, kv0, kvn = tp_partition(cfg.num_attention_heads, cfg.num_key_value_heads, tp, rank)

        def shard_qkv(full):                                       # rows of this rank's q heads | kv heads | kv heads
            wq, wk, wv = full.split([hq, hkv, hkv], dim=0)
            return torch.cat([wq[q0 * d:(q0 + qn) * d], wk[kv0 * d:(kv0 + kvn) * d], wv[kv0 * d:(kv0 + kvn) * d]], dim=0)

        self.embed_tokens.weight.copy_(draw(cfg.vocab_size, cfg.hidden_size))
        if not cfg.tie_word_embeddings:
            self.lm_head.weight.copy_(draw(cfg.vocab_size, cfg.hidden_size))
        hq, hkv, d = cfg.num_attention_heads * cfg.head_dim, cfg.num_key_value_heads * cfg.head_dim, cfg.head_dim
        for layer in self.layers:
            a, m = layer.self_attn, layer.mlp
            a.qkv_proj.weight.copy_(shard_qkv(draw(hq + 2 * hkv, cfg.hidden_size)))
            if cfg.qkv_bias:
                a.qkv_proj.bias.copy_(shard_qkv(draw(hq + 2 * hkv)))
            a.o_proj.weight.copy_(draw(cfg.hidden_size, hq)[:, q0 * d:(q0 + qn) * d])
            pad = m.down_proj.weight.shape[1] - m.inter                   # zero rows / columns of a padded shard (QwenMLP)
            gate_up = [F.pad(p.chunk(tp, dim=0)[rank], (0, 0, 0, pad)) for p in draw(2 * cfg.intermediate_size, cfg.hidden_size).chunk(2, dim=0)]
            m.gate_up_proj.weight.copy_(torch.cat(gate_up, dim=0))
            m.down_proj.weight.copy_(F.pad(draw(cfg.hidden_size, cfg.intermediate_size).chunk(tp, dim=1)[rank], (0, pad)))
        return self
