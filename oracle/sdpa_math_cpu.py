"""CPU port of the reference's sdpa.math decode path, in torch ops.  TEST INFRASTRUCTURE / cpu_baseline ONLY.

Same algorithm, same order of operations as nanovllm/layers/attention_sdpa.py:122-182
(flash_attn_with_kvcache): build flat token indices from the block table (:149-157), gather dense
[B, max_blocks*block_size, KVH, D] K and V (:163-164), zero the invalid tail (:165-167), boolean mask
(:168), softmax(QK^T/sqrt(D))V with GQA sharing.  The SDPA call (:171-180, math backend) is written out as
its definition; tests/test_oracle_golden.py pins it to the reference's outputs.  Only bench.py's
`cpu_baseline` leg and tests may import this file; the product never does.
"""
import math

import torch


def flash_attn_with_kvcache_cpu(q, k_cache, v_cache, cache_seqlens, block_table):
    """q [B, 1, H, D]; caches [NB, bs, KVH, D]; cache_seqlens int32 [B]; block_table int32 [B, max_blocks]."""
    b, one, h, d = q.shape
    assert one == 1
    block_size = k_cache.size(1)
    kvh = k_cache.size(2)
    block_table = block_table.to(torch.long)
    max_seq_len = block_table.size(1) * block_size
    positions = torch.arange(max_seq_len, dtype=torch.long)
    block_ids = block_table[:, positions // block_size].clamp(min=0)             # [B, S]
    flat_idx = block_ids * block_size + positions % block_size
    batch_k = k_cache.view(-1, kvh, d)[flat_idx]                                  # [B, S, KVH, D]
    batch_v = v_cache.view(-1, kvh, d)[flat_idx]
    kv_valid = positions.unsqueeze(0) < cache_seqlens.unsqueeze(1)                # [B, S]
    batch_k = batch_k * kv_valid.unsqueeze(-1).unsqueeze(-1)
    batch_v = batch_v * kv_valid.unsqueeze(-1).unsqueeze(-1)
    g = h // kvh
    qh = q.transpose(1, 2)                                                        # [B, H, 1, D]
    kh = batch_k.transpose(1, 2).repeat_interleave(g, dim=1)                      # enable_gqa
    vh = batch_v.transpose(1, 2).repeat_interleave(g, dim=1)
    scores = (qh @ kh.transpose(-1, -2)) / math.sqrt(d)                           # [B, H, 1, S]
    scores = scores.masked_fill(~kv_valid[:, None, None, :], float("-inf"))
    probs = torch.softmax(scores, dim=-1)
    probs = torch.nan_to_num(probs, nan=0.0)                                      # all-masked rows (ctx == 0) -> zeros
    return (probs @ vh).transpose(1, 2)                                           # [B, 1, H, D]


def flash_attn_varlen_func_cpu(q, k, v, cu_seqlens_q, cu_seqlens_k):
    """CPU port of the reference's sdpa.math prefill (attention_sdpa.py:65-119): a Python loop over the sequences of the packed
    batch (:83-92), each one a causal softmax(QK^T/sqrt(D))V with GQA sharing (the SDPA call :101-110, math backend, written
    out as its definition: scores materialised [H, Sq, Sk], top-left causal mask as `is_causal=True` builds it).
    q [Tq, H, D]; k, v [Tk, KVH, D]; cu_seqlens int32 [B+1].  tests/test_oracle_golden.py pins it to the reference's outputs."""
    h, d = q.shape[1], q.shape[2]
    g = h // k.shape[1]
    outs = []
    for i in range(cu_seqlens_q.numel() - 1):
        qi = q[cu_seqlens_q[i]:cu_seqlens_q[i + 1]].transpose(0, 1)               # [H, Sq, D]
        ki = k[cu_seqlens_k[i]:cu_seqlens_k[i + 1]].transpose(0, 1).repeat_interleave(g, dim=0)
        vi = v[cu_seqlens_k[i]:cu_seqlens_k[i + 1]].transpose(0, 1).repeat_interleave(g, dim=0)
        sq, sk = qi.shape[1], ki.shape[1]
        scores = (qi @ ki.transpose(-1, -2)) / math.sqrt(d)
        mask = torch.ones(sq, sk, dtype=torch.bool).tril()                        # is_causal: top-left aligned
        scores = scores.masked_fill(~mask, float("-inf"))
        outs.append((torch.softmax(scores, dim=-1) @ vi).transpose(0, 1))         # [Sq, H, D]
    return torch.cat(outs, dim=0)
