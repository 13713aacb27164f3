#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REFERENCE's own sdpa.math functions on CPU.

Run once, in the build container, where /root/reference exists:

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py

TEST INFRASTRUCTURE ONLY.  The reference never travels to the GPU box; only the small
fixtures written here (inputs + the reference's outputs) do.  Nothing in tests/,
bench.py or the product imports this script.

What is pinned by the reference itself (executed here, fp32 on bf16-valued inputs):
  decode_*.npz   nanovllm.layers.attention_sdpa.flash_attn_with_kvcache   (:122-182)
  prefill_*.npz  nanovllm.layers.attention_sdpa.flash_attn_varlen_func    (:65-119)
  meta_*.npz     nanovllm.engine.model_runner.ModelRunner.prepare_decode /
                 prepare_prefill (:160-269) driven with nanovllm.engine.sequence.Sequence
                 objects; the two device hops (pin_memory=True, .cuda()) are made
                 no-ops for the call because this container has no GPU.
What is NOT executable here (Triton launch needs a GPU): store_kvcache.  store_*.npz
holds inputs plus the result of the literal one-line semantic the kernel states
(attention.py:37-41: cache.view(-1, D)[slot] = key[i]) evaluated with torch indexing,
with slot < 0 rows skipped (attention_triton.py:29-31).
"""
import os
import sys
from unittest import mock

import numpy as np
import torch

REF = os.environ.get("NVH_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

from torch.nn.attention import SDPBackend  # noqa: E402
from nanovllm.layers import attention_sdpa as ref_sdpa  # noqa: E402
from nanovllm.utils.context import get_context  # noqa: E402


def bf16_bits(t: torch.Tensor) -> np.ndarray:
    return t.to(torch.bfloat16).contiguous().view(torch.int16).numpy().view(np.uint16)


def randn_bf16(gen, *shape, scale=1.0):
    """bf16-valued float32 tensor."""
    return (torch.randn(*shape, generator=gen) * scale).to(torch.bfloat16).to(torch.float32)


def gen_decode(name, seed, H, KVH, D, ctxs, block_size=256, pad=-1, width=None, spare=2):
    gen = torch.Generator().manual_seed(seed)
    B = len(ctxs)
    need = [(c + block_size - 1) // block_size for c in ctxs]
    NB = sum(need) + spare
    perm = torch.randperm(NB, generator=gen).tolist()           # shuffled, non-contiguous ids
    width = width or max(max(need), 1)
    bt = torch.full((B, width), pad, dtype=torch.int32)
    it = iter(perm)
    for b, n in enumerate(need):
        for j in range(n):
            bt[b, j] = next(it)
    kc = randn_bf16(gen, NB, block_size, KVH, D)
    vc = randn_bf16(gen, NB, block_size, KVH, D)
    q = randn_bf16(gen, B, 1, H, D)
    cl = torch.tensor(ctxs, dtype=torch.int32)
    out = ref_sdpa.flash_attn_with_kvcache(q, kc, vc, cache_seqlens=cl, block_table=bt,
                                           causal=False, sdpa_backend=SDPBackend.MATH)
    np.savez_compressed(os.path.join(OUT, name), q=bf16_bits(q[:, 0]), k_cache=bf16_bits(kc),
                        v_cache=bf16_bits(vc), context_lens=cl.numpy(), block_tables=bt.numpy(),
                        expected=out[:, 0].float().numpy(), shape=np.array([H, KVH, D, block_size]))
    print(name, "max|o|", float(out.abs().max()))


def gen_prefill(name, seed, H, KVH, D, lens):
    gen = torch.Generator().manual_seed(seed)
    T = sum(lens)
    q = randn_bf16(gen, T, H, D)
    k = randn_bf16(gen, T, KVH, D)
    v = randn_bf16(gen, T, KVH, D)
    cu = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32)
    out = ref_sdpa.flash_attn_varlen_func(q, k, v, max_seqlen_q=max(lens), cu_seqlens_q=cu,
                                          max_seqlen_k=max(lens), cu_seqlens_k=cu, causal=True,
                                          sdpa_backend=SDPBackend.MATH)
    np.savez_compressed(os.path.join(OUT, name), q=bf16_bits(q), k=bf16_bits(k), v=bf16_bits(v),
                        cu_seqlens=cu.numpy(), expected=out.float().numpy(),
                        shape=np.array([H, KVH, D]))
    print(name, "max|o|", float(out.abs().max()))


def gen_store(name, seed, KVH, D, n_tokens, NB, block_size=256):
    gen = torch.Generator().manual_seed(seed)
    H = 7 * KVH
    qkv = randn_bf16(gen, n_tokens, (H + 2 * KVH) * D)           # fused projection output, qwen3.py:104-106
    k = qkv[:, H * D:(H + KVH) * D].view(n_tokens, KVH, D)       # strided views: row stride (H+2KVH)*D
    v = qkv[:, (H + KVH) * D:].view(n_tokens, KVH, D)
    kc = randn_bf16(gen, NB, block_size, KVH, D)
    vc = randn_bf16(gen, NB, block_size, KVH, D)
    # slots: a run crossing a block boundary, scattered singles, and some -1 (skipped)
    run_start = block_size * 2 - 5
    slots = list(range(run_start, run_start + n_tokens // 2))
    rest = torch.randperm(NB * block_size, generator=gen).tolist()
    taken = set(slots)
    for s in rest:
        if len(slots) == n_tokens:
            break
        if s not in taken:
            slots.append(s)
            taken.add(s)
    slots = torch.tensor(slots, dtype=torch.int32)
    slots[1::7] = -1
    kc_exp, vc_exp = kc.clone(), vc.clone()
    live = slots >= 0
    kc_exp.view(-1, KVH * D)[slots[live].long()] = k[live].reshape(-1, KVH * D)   # attention.py:37-41
    vc_exp.view(-1, KVH * D)[slots[live].long()] = v[live].reshape(-1, KVH * D)
    np.savez_compressed(os.path.join(OUT, name), qkv=bf16_bits(qkv), k_cache=bf16_bits(kc), v_cache=bf16_bits(vc),
                        slot_mapping=slots.numpy(), k_cache_expected=bf16_bits(kc_exp),
                        v_cache_expected=bf16_bits(vc_exp), shape=np.array([H, KVH, D, block_size]))
    print(name, "tokens", n_tokens, "skipped", int((~live).sum()))


def gen_meta(name):
    """Drive the reference's prepare_decode / prepare_prefill (model_runner.py:171-269)."""
    from nanovllm.engine.sequence import Sequence
    from nanovllm.engine import model_runner as mr

    real_tensor = torch.tensor

    def cpu_tensor(*a, **kw):
        kw.pop("pin_memory", None)                  # no GPU here: pinned memory unavailable
        return real_tensor(*a, **kw)

    class Shell:                                     # the two attributes the methods read from self
        block_size = 256
        prepare_block_tables = mr.ModelRunner.prepare_block_tables

    def mkseq(n_tokens, table, cached=0):
        s = Sequence(list(range(n_tokens)))
        s.block_table = list(table)
        s.num_cached_tokens = cached
        return s

    dec_seqs = [mkseq(1025, [7, 3, 11, 2, 9]), mkseq(256, [5]), mkseq(257, [4, 8]), mkseq(700, [0, 12, 6]), mkseq(1, [10])]
    pre_seqs = [mkseq(300, [3, 9]), mkseq(1, [4]), mkseq(256, [1]), mkseq(513, [2, 5, 7])]
    pfx_seqs = [mkseq(600, [3, 9, 8], cached=512), mkseq(40, [4]), mkseq(300, [6, 1], cached=256)]
    out = {}
    with mock.patch.object(torch, "tensor", cpu_tensor), mock.patch.object(torch.Tensor, "cuda", lambda self, *a, **k: self):
        sh = Shell()
        ids, pos = mr.ModelRunner.prepare_decode(sh, dec_seqs)
        c = get_context()
        out.update(dec_input_ids=ids.numpy(), dec_tokens=[len(s) for s in dec_seqs], dec_positions=pos.numpy(), dec_slot_mapping=c.slot_mapping.numpy(),
                   dec_context_lens=c.context_lens.numpy(), dec_block_tables=c.block_tables.numpy())
        for tag, seqs in (("pre", pre_seqs), ("pfx", pfx_seqs)):
            ids, pos = mr.ModelRunner.prepare_prefill(sh, seqs)
            c = get_context()
            out.update({f"{tag}_input_ids": ids.numpy(), f"{tag}_tokens": [len(s) for s in seqs], f"{tag}_cached": [s.num_cached_tokens for s in seqs],
                        f"{tag}_positions": pos.numpy(), f"{tag}_cu_seqlens_q": c.cu_seqlens_q.numpy(),
                        f"{tag}_cu_seqlens_k": c.cu_seqlens_k.numpy(),
                        f"{tag}_max_seqlen": np.array([c.max_seqlen_q, c.max_seqlen_k]),
                        f"{tag}_slot_mapping": c.slot_mapping.numpy(),
                        f"{tag}_block_tables": (c.block_tables.numpy() if c.block_tables is not None else np.zeros((0, 0), np.int32))})
        for tag, seqs in (("dec", dec_seqs), ("pre", pre_seqs), ("pfx", pfx_seqs)):
            w = max(len(s.block_table) for s in seqs)
            out[f"{tag}_tables_in"] = np.array([s.block_table + [-9] * (w - len(s.block_table)) for s in seqs], dtype=np.int32)
        # a decode TRAJECTORY: the reference's own between-steps bookkeeping (scheduler.py:60-110: may_append before the step,
        # append_token after it; block_manager.py:62-159) driven for 300 steps across block boundaries, prepare_decode's
        # tensors recorded at every step.  Token ids come from a seeded generator (the model is not part of this path).
        from nanovllm.engine.block_manager import BlockManager
        bm = BlockManager(64, 256)
        gen = torch.Generator().manual_seed(77)
        traj_lens = [255, 256, 1, 511, 700, 130]
        traj_seqs = []
        for n in traj_lens:
            s = Sequence(torch.randint(0, 10000, (n,), generator=gen).tolist())
            bm.allocate(s)
            traj_seqs.append(s)
        steps = 300
        toks = torch.randint(0, 10000, (steps + 1, len(traj_seqs)), generator=gen)
        for s, t in zip(traj_seqs, toks[0].tolist()):            # the token prefill sampled (postprocess of the prefill step)
            s.append_token(t)
        rec = {k: [] for k in ("input_ids", "positions", "slot_mapping", "context_lens")}
        for i in range(steps):
            for s in traj_seqs:
                assert bm.can_append(s)
                bm.may_append(s)                                  # scheduler.schedule, decode branch
            ids, pos = mr.ModelRunner.prepare_decode(sh, traj_seqs)
            c = get_context()
            rec["input_ids"].append(ids.numpy()); rec["positions"].append(pos.numpy())
            rec["slot_mapping"].append(c.slot_mapping.numpy()); rec["context_lens"].append(c.context_lens.numpy())
            for s, t in zip(traj_seqs, toks[i + 1].tolist()):    # scheduler.postprocess
                s.append_token(t)
        w = max(len(s.block_table) for s in traj_seqs)
        out.update(traj_prompt_lens=np.array(traj_lens), traj_tokens=toks.numpy(),
                   traj_final_tables=np.array([s.block_table + [-1] * (w - len(s.block_table)) for s in traj_seqs], dtype=np.int32),
                   **{f"traj_{k}": np.stack(v) for k, v in rec.items()})
    np.savez_compressed(os.path.join(OUT, name), **out)
    print(name, "keys", len(out))


def gen_rope(name, seed, heads, D, n_tokens, base=1000000.0, max_position=4096):
    """nanovllm.layers.rotary_embedding: RotaryEmbedding.__init__ builds cos_sin_cache (:29-36, plain code) and
    apply_rotary_emb (:6-16, plain function) rotates; forward itself is @torch.compile'd and is not called."""
    from nanovllm.layers.rotary_embedding import RotaryEmbedding, apply_rotary_emb
    gen = torch.Generator().manual_seed(seed)
    rope = RotaryEmbedding(D, D, max_position, base)
    x = randn_bf16(gen, n_tokens, heads, D).to(torch.bfloat16)
    pos = torch.randint(0, max_position, (n_tokens,), generator=gen)
    cos, sin = rope.cos_sin_cache[pos].chunk(2, dim=-1)
    y = apply_rotary_emb(x, cos, sin)
    np.savez_compressed(os.path.join(OUT, name), x=bf16_bits(x), positions=pos.numpy(), expected=bf16_bits(y),
                        cos_sin=rope.cos_sin_cache.float().numpy()[:64], shape=np.array([heads, D, max_position]), base=np.array([base]))
    print(name, "tokens", n_tokens)


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    if sys.argv[1:] == ["meta"]:                     # only the runner-metadata fixture (leaves the other files' bytes alone)
        gen_meta("meta_runner.npz")
        return
    # (1) decode: three shape sets, edge context lengths, -1 and 0 padding, a ctx==0 row
    gen_decode("decode_q2_0p5b.npz", 1, 14, 2, 64, [700, 257, 0, 256], pad=-1)
    gen_decode("decode_q2_0p5b_graphpad.npz", 2, 14, 2, 64, [255, 1, 17, 0, 300], pad=0, width=16)
    gen_decode("decode_q2_7b_tp4.npz", 3, 7, 1, 128, [700, 17, 256, 1], pad=-1)
    gen_decode("decode_g2_d128.npz", 4, 4, 2, 128, [300, 255, 1], pad=-1)
    gen_decode("decode_single.npz", 5, 14, 2, 64, [513], pad=-1)
    # (2) prefill: packed varlen batches incl. length-1 sequences and non-multiples of every tile size
    gen_prefill("prefill_q2_0p5b.npz", 11, 14, 2, 64, [1, 5, 7, 64, 129, 300])
    gen_prefill("prefill_g2_d128.npz", 12, 4, 2, 128, [3, 130, 257])
    gen_prefill("prefill_q2_7b_tp4.npz", 13, 7, 1, 128, [33, 1, 96])
    # (3) store
    gen_store("store_d64.npz", 21, 2, 64, 61, 4)
    gen_store("store_d128.npz", 22, 1, 128, 40, 3)
    # (3b) RoPE (the step before attention)
    gen_rope("rope_d64.npz", 31, 16, 64, 37)
    gen_rope("rope_d128.npz", 32, 5, 128, 19)
    # (4) runner metadata
    gen_meta("meta_runner.npz")


if __name__ == "__main__":
    main()
