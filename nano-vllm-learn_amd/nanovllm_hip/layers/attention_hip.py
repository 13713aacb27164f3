"""`--attn-backend hip`: the MI355X attention module, a drop-in beside the reference's flash / sdpa /
triton `Attention` classes (nanovllm/layers/attention.py:58-103, attention_sdpa.py:208-269,
attention_triton.py:386-464).

Same constructor `(num_heads, head_dim, scale, num_kv_heads, **kw)`, same `k_cache` / `v_cache`
attributes (the runner binds cache views by `hasattr`, engine/model_runner.py:148-157), same
`forward(q, k, v) -> o` on flattened `[N, H*D]` / `[N, KVH*D]` tensors, same global Context.
All arithmetic happens in libnvh_attn.so; there is no eager / CPU fallback.
"""
import torch
from torch import nn

from .. import ops
from ..utils.context import get_context


class Attention(nn.Module):

    def __init__(self, num_heads, head_dim, scale, num_kv_heads, block_size: int = 256, fused_decode: bool = True,
                 prefill_pv_fp16=None):
        super().__init__()
        self.num_heads = num_heads
        self.head_dim = head_dim
        self.scale = scale                      # softmax scale, D**-0.5 (models/qwen3.py:40); used as given
        self.num_kv_heads = num_kv_heads
        self.block_size = block_size
        self.fused_decode = fused_decode        # one C-ABI call for store + attend on the decode step
        # prefill P V on the fp16 matrix pipe (ops.flash_attn_varlen_func pv_fp16): None = the library's rule (bf16 output, packed rows, >= 512 keys per sequence or config 5's short batches;
        # range-guarded, falls back by itself), False = never (P as bf16 hi + lo, 6e-6 instead of <= 2^-12 * max|v|), True = at every length
        self.prefill_pv_fp16 = prefill_pv_fp16
        self.k_cache = self.v_cache = torch.tensor([])

    def rope_store_attend(self, qkv, positions, cos_sin, q_norm_weight=None, k_norm_weight=None, eps=1e-6):
        """Widened entry: takes the fused projection output, applies (q/k norm ->) RoPE and the
        cache store in ONE launch (nvh_rope_store), then attends without a second store.  Same result as
        q_norm/k_norm + rotary_emb + forward(q, k, v) of the reference call sequence (qwen3.py:104-117)."""
        context = get_context()
        h, kvh, d = self.num_heads, self.num_kv_heads, self.head_dim
        ops.rope_store(qkv, positions, cos_sin, h, kvh, d, self.k_cache, self.v_cache, context.slot_mapping,
                       q_norm_weight, k_norm_weight, eps)
        q, k, v = qkv.split([h * d, kvh * d, kvh * d], dim=-1)
        return self._attend(q.view(-1, h, d), k.view(-1, kvh, d), v.view(-1, kvh, d), context, store=False)

    def decode_attend(self, q, out_packed=None):
        """Decode attention for rows whose K/V are ALREADY in the cache (stored by the fused qkv launch): no store here.
        out_packed: optional flat buffer that also receives the output in fragment order for the output projection."""
        context = get_context()
        o = ops.flash_attn_with_kvcache(q.view(-1, 1, self.num_heads, self.head_dim), self.k_cache, self.v_cache,
                                        cache_seqlens=context.context_lens, block_table=context.block_tables,
                                        softmax_scale=self.scale, causal=True, out_packed=out_packed)
        return o.view(-1, self.num_heads * self.head_dim)

    def qkv_rope_store_attend(self, x_packed, rows, qkv_weight, qkv_bias, norm_eps, positions, cos_sin, out_packed=None, prefetch=None,
                              mode="auto", linear_workspace=None, kv_prefetch_passes=0):
        """The whole front of a decode layer as ONE launch (nvh_qkv_rope_attend; SURVEY.md section 8f-2 taken to its end): the fused qkv
        projection of the fragment-packed residual rows (RMSNorm folded into `qkv_weight`), bias, RoPE, the K/V store and the
        decode attention on the result — the reference's qkv_proj -> rotary_emb -> self.attn (models/qwen3.py:104-117).  Shapes the
        one-launch kernel does not serve run as the two launches, same results.  Returns the attention output [rows, H*D]."""
        context = get_context()
        rope = dict(positions=positions, cos_sin=cos_sin, k_cache=self.k_cache, v_cache=self.v_cache, slot_mapping=context.slot_mapping,
                    num_heads=self.num_heads, num_kv_heads=self.num_kv_heads, head_dim=self.head_dim)
        _, o, _ = ops.qkv_rope_attend(x_packed, qkv_weight, rope=rope, context_lens=context.context_lens, block_tables=context.block_tables,
                                      bias=qkv_bias, norm_eps=norm_eps, norm_folded=True, x_packed_rows=rows, attn_out_packed=out_packed,
                                      softmax_scale=self.scale, prefetch=prefetch, linear_workspace=linear_workspace,
                                      mode=mode, spin_limit=kv_prefetch_passes)
        return o.view(-1, self.num_heads * self.head_dim)

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor):
        q = q.view(-1, self.num_heads, self.head_dim)
        k = k.view(-1, self.num_kv_heads, self.head_dim)
        v = v.view(-1, self.num_kv_heads, self.head_dim)
        return self._attend(q, k, v, get_context(), store=True)

    def _attend(self, q, k, v, context, store):
        k_cache, v_cache = self.k_cache, self.v_cache
        have_cache = k_cache.numel() > 0 and v_cache.numel() > 0       # warmup prefill runs before allocation
        store = store and have_cache and context.slot_mapping is not None   # attention.py:84, attention_sdpa.py:242

        if context.is_prefill:
            if store:
                ops.store_kvcache(k, v, k_cache, v_cache, context.slot_mapping)
            if context.block_tables is not None:                        # prefix-cache hit: read K/V from the cache
                k, v = k_cache, v_cache
            o = ops.flash_attn_varlen_func(q, k, v,
                                           max_seqlen_q=context.max_seqlen_q, cu_seqlens_q=context.cu_seqlens_q,
                                           max_seqlen_k=context.max_seqlen_k, cu_seqlens_k=context.cu_seqlens_k,
                                           softmax_scale=self.scale, causal=True, block_table=context.block_tables,
                                           pv_fp16=self.prefill_pv_fp16 if context.block_tables is None else False)
        else:
            if not have_cache:
                raise RuntimeError("decode needs an allocated KV cache (k_cache/v_cache not bound)")
            if store and self.fused_decode:
                o = ops.decode_step(q, k, v, k_cache, v_cache, context.slot_mapping, context.context_lens,
                                    context.block_tables, softmax_scale=self.scale)
            else:
                if store:
                    ops.store_kvcache(k, v, k_cache, v_cache, context.slot_mapping)
                o = ops.flash_attn_with_kvcache(q.unsqueeze(1), k_cache, v_cache, cache_seqlens=context.context_lens,
                                                block_table=context.block_tables, softmax_scale=self.scale, causal=True)
        return o.view(-1, self.num_heads * self.head_dim)
