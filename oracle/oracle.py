"""CPU oracle for the nano-vllm paged-attention hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import it.
The shipped path (``nano-vllm-learn_amd/nanovllm_hip``) never imports anything
under ``oracle/`` and raises if the HIP library is missing.

It restates, in plain numpy with explicit loops, what the reference computes on the
attention path.  Every function cites the reference lines it follows
(paths relative to the reference checkout):

  store_kvcache        nanovllm/layers/attention.py:19-55  (+ slot<0 skip of
                       nanovllm/layers/attention_triton.py:29-31)
  paged_decode         nanovllm/layers/attention_sdpa.py:122-182
  prefill_varlen       nanovllm/layers/attention_sdpa.py:65-119
  paged_prefill        flash-attn semantics of the call at
                       nanovllm/layers/attention.py:90-96 (PARITY UNPINNED: the
                       reference's own sdpa/triton backends get this mode wrong,
                       SURVEY.md App. B3; nothing in the reference pins it)
  prepare_decode /     nanovllm/engine/model_runner.py:160-269
  prepare_prefill

Pinning: ``oracle/gen_golden.py`` imported the reference's ``sdpa.math`` functions
on CPU in the build container and wrote ``tests/golden/*.npz``;
``tests/test_oracle_golden.py`` checks this restatement against those vectors
(fp32, max abs diff <= 2e-6).  The reference has no tests or golden vectors of its
own.

All arithmetic is float64 internally (inputs are bf16- or fp32-valued), outputs are
returned as float32.  Sizes: pure loops, intended for cases that finish in seconds.
"""
from __future__ import annotations

import numpy as np

# (rope_neox / rms_norm_heads below restate the step just before attention: SURVEY.md section 8f row 2)
__all__ = [
    "rope_cos_sin", "rope_neox", "rms_norm_heads",
    "bf16_bits_to_f32", "f32_to_bf16_bits", "round_to_bf16",
    "store_kvcache", "paged_decode", "prefill_varlen", "paged_prefill",
    "prepare_block_tables", "prepare_decode", "prepare_prefill", "SeqState",
]


# --------------------------------------------------------------------------- bf16 helpers
def bf16_bits_to_f32(bits: np.ndarray) -> np.ndarray:
    """uint16 bf16 bit patterns -> float32 values (exact)."""
    bits = np.asarray(bits, dtype=np.uint16)
    return (bits.astype(np.uint32) << 16).view(np.float32)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    """float32 -> bf16 bit patterns, round-to-nearest-even (finite inputs)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    rounded = (u + 0x7FFF + ((u >> 16) & 1)) >> 16
    return rounded.astype(np.uint16)


def round_to_bf16(x: np.ndarray) -> np.ndarray:
    """Round float32 values to the nearest bf16-representable float32."""
    return bf16_bits_to_f32(f32_to_bf16_bits(x))


# --------------------------------------------------------------------------- a1: store
def store_kvcache(key, value, k_cache, v_cache, slot_mapping):
    """In-place scatter of new K/V rows into the paged cache.

    Reference: attention.py:19-41 — ``cache.view(-1, KVH*D)[slot[i], :] = key[i]`` for
    each token i; the Triton variant (attention_triton.py:29-31) returns early when
    ``slot == -1``.  The HIP path skips every ``slot < 0``.

    key/value: [N, KVH, D]; caches: [NB, block_size, KVH, D]; slot_mapping: [N] ints.
    """
    n, kvh, d = key.shape
    assert value.shape == key.shape
    assert k_cache.shape[2:] == (kvh, d) and v_cache.shape == k_cache.shape
    assert len(slot_mapping) == n
    kflat = k_cache.reshape(-1, kvh, d)
    vflat = v_cache.reshape(-1, kvh, d)
    for i in range(n):
        slot = int(slot_mapping[i])
        if slot < 0:
            continue
        kflat[slot] = key[i]
        vflat[slot] = value[i]


# --------------------------------------------------------------------------- softmax core
def _attend(q, k, v, scale, n_valid_per_row):
    """q [Sq, D], k/v [Sk, D] float64; row i attends k[0:n_valid_per_row[i]]."""
    sq, d = q.shape
    out = np.zeros((sq, v.shape[1]), dtype=np.float64)
    s = (q @ k.T) * scale                       # [Sq, Sk]
    for i in range(sq):
        n = int(n_valid_per_row[i])
        if n <= 0:
            continue                            # no visible key -> zeros (oracle: masked softmax of an
            #                                     all-zero gathered K/V, attention_sdpa.py:166-167)
        row = s[i, :n]
        m = row.max()
        p = np.exp(row - m)
        out[i] = (p / p.sum()) @ v[:n]
    return out


# --------------------------------------------------------------------------- a2: decode
def paged_decode(q, k_cache, v_cache, context_lens, block_tables, scale=None):
    """Single-query attention over a paged KV cache.

    Reference: attention_sdpa.py:122-182.  Token t of sequence b lives at
    ``cache[block_tables[b, t // bs], t % bs, h // (H // KVH), :]`` (:155-164); only
    ``t < context_lens[b]`` is visible (:165, :168); ``scale=None`` = D**-0.5 (:179).
    A row with ``context_lens[b] == 0`` yields zeros.  block_table entries at or past
    ``ceil(ctx/bs)`` are never dereferenced here (the reference clamps them to 0 and
    masks, which is the same thing).

    q: [B, H, D]; caches [NB, bs, KVH, D]; returns float32 [B, H, D].
    """
    q = np.asarray(q, dtype=np.float64)
    b_, h, d = q.shape
    nb, bs, kvh, d2 = k_cache.shape
    assert d2 == d and h % kvh == 0
    g = h // kvh
    if scale is None:
        scale = float(d) ** -0.5
    out = np.zeros((b_, h, d), dtype=np.float64)
    for b in range(b_):
        ctx = int(context_lens[b])
        if ctx <= 0:
            continue
        nblk = (ctx + bs - 1) // bs
        ids = [int(block_tables[b, j]) for j in range(nblk)]
        assert all(0 <= i < nb for i in ids), "live block id out of range"
        kseq = np.concatenate([np.asarray(k_cache[i], dtype=np.float64) for i in ids], axis=0)[:ctx]
        vseq = np.concatenate([np.asarray(v_cache[i], dtype=np.float64) for i in ids], axis=0)[:ctx]
        for hh in range(h):
            kh = hh // g
            out[b, hh] = _attend(q[b, hh:hh + 1], kseq[:, kh], vseq[:, kh], scale, [ctx])[0]
    return out.astype(np.float32)


# --------------------------------------------------------------------------- a3: prefill
def prefill_varlen(q, k, v, cu_seqlens_q, cu_seqlens_k, scale=None, causal=True):
    """Variable-length causal attention over packed sequences.

    Reference: attention_sdpa.py:65-119 — per sequence i, slice q/k/v by the cumulative
    lengths (:84-93) and run SDPA with ``is_causal`` and GQA (:105-112).  torch's
    ``is_causal`` mask is top-left aligned (``tril``), which equals flash-attn's
    bottom-right alignment only when seqlen_q == seqlen_k — the only case the reference
    can reach through this function without a block_table.  This restatement uses the
    bottom-right form (row i of Sq sees keys ``0 .. i + (Sk - Sq)``), identical when
    the lengths are equal.

    q [Tq, H, D], k/v [Tk, KVH, D]; returns float32 [Tq, H, D].
    """
    q = np.asarray(q, dtype=np.float64)
    k = np.asarray(k, dtype=np.float64)
    v = np.asarray(v, dtype=np.float64)
    tq, h, d = q.shape
    kvh = k.shape[1]
    g = h // kvh
    if scale is None:
        scale = float(d) ** -0.5
    out = np.zeros((tq, h, d), dtype=np.float64)
    nseq = len(cu_seqlens_q) - 1
    for i in range(nseq):
        q0, q1 = int(cu_seqlens_q[i]), int(cu_seqlens_q[i + 1])
        k0, k1 = int(cu_seqlens_k[i]), int(cu_seqlens_k[i + 1])
        sq, sk = q1 - q0, k1 - k0
        if sq == 0:
            continue
        if causal:
            nvalid = [min(sk, r + 1 + (sk - sq)) for r in range(sq)]
        else:
            nvalid = [sk] * sq
        for hh in range(h):
            kh = hh // g
            out[q0:q1, hh] = _attend(q[q0:q1, hh], k[k0:k1, kh], v[k0:k1, kh], scale, nvalid)
    return out.astype(np.float32)


# --------------------------------------------------------------------------- a4: paged prefill
def paged_prefill(q, k_cache, v_cache, cu_seqlens_q, cu_seqlens_k, block_tables, scale=None):
    """Prefill whose K/V come from the paged cache (prefix-cache hit).

    Call site: attention.py:90-96 with ``block_table`` set (triggered by
    model_runner.py:225-232).  Sequence i has ``Sk = cu_k[i+1]-cu_k[i]`` cached+new keys
    located through ``block_tables[i]`` and ``Sq <= Sk`` new queries; causal mask is
    bottom-right aligned (flash-attn).  PARITY UNPINNED by the reference.
    """
    q = np.asarray(q, dtype=np.float64)
    tq, h, d = q.shape
    nb, bs, kvh, _ = k_cache.shape
    g = h // kvh
    if scale is None:
        scale = float(d) ** -0.5
    out = np.zeros((tq, h, d), dtype=np.float64)
    for i in range(len(cu_seqlens_q) - 1):
        q0, q1 = int(cu_seqlens_q[i]), int(cu_seqlens_q[i + 1])
        sk = int(cu_seqlens_k[i + 1]) - int(cu_seqlens_k[i])
        sq = q1 - q0
        if sq == 0:
            continue
        nblk = (sk + bs - 1) // bs
        ids = [int(block_tables[i, j]) for j in range(nblk)]
        kseq = np.concatenate([np.asarray(k_cache[j], dtype=np.float64) for j in ids], axis=0)[:sk]
        vseq = np.concatenate([np.asarray(v_cache[j], dtype=np.float64) for j in ids], axis=0)[:sk]
        nvalid = [min(sk, r + 1 + (sk - sq)) for r in range(sq)]
        for hh in range(h):
            kh = hh // g
            out[q0:q1, hh] = _attend(q[q0:q1, hh], kseq[:, kh], vseq[:, kh], scale, nvalid)
    return out.astype(np.float32)


# --------------------------------------------------------------------------- a7: metadata producers
class SeqState:
    """The fields of the reference ``Sequence`` that the metadata producers read
    (engine/sequence.py:14-69): token count, cached-token count, block table."""

    def __init__(self, num_tokens, block_table, num_cached_tokens=0, block_size=256, last_token=0):
        self.num_tokens = int(num_tokens)
        self.block_table = list(block_table)
        self.num_cached_tokens = int(num_cached_tokens)
        self.block_size = int(block_size)
        self.last_token = int(last_token)

    def __len__(self):
        return self.num_tokens

    @property
    def num_blocks(self):                      # sequence.py:58-60
        return (self.num_tokens + self.block_size - 1) // self.block_size

    @property
    def num_cached_blocks(self):               # sequence.py:53-55
        return self.num_cached_tokens // self.block_size

    @property
    def last_block_num_tokens(self):           # sequence.py:62-65
        return self.num_tokens - (self.num_blocks - 1) * self.block_size


def prepare_block_tables(seqs):
    """model_runner.py:160-169: right-pad each block table with -1 to the widest."""
    width = max(len(s.block_table) for s in seqs)
    return np.array([s.block_table + [-1] * (width - len(s.block_table)) for s in seqs], dtype=np.int32)


def prepare_decode(seqs):
    """model_runner.py:244-269 -> (positions, slot_mapping, context_lens, block_tables)."""
    positions, slots, ctx = [], [], []
    for s in seqs:
        positions.append(len(s))                                        # :251
        ctx.append(len(s))                                              # :252
        slots.append(s.block_table[-1] * s.block_size + s.last_block_num_tokens - 1)   # :254-258
    return (np.array(positions, dtype=np.int64), np.array(slots, dtype=np.int32),
            np.array(ctx, dtype=np.int32), prepare_block_tables(seqs))


def prepare_prefill(seqs):
    """model_runner.py:171-242 -> dict with positions, cu_seqlens_q/k, max_seqlen_q/k,
    slot_mapping, block_tables (None unless some sequence has cached tokens)."""
    positions, slots = [], []
    cu_q, cu_k = [0], [0]
    max_q = max_k = 0
    for s in seqs:
        seqlen = len(s)
        positions.extend(range(s.num_cached_tokens, seqlen))            # :187
        sq, sk = seqlen - s.num_cached_tokens, seqlen                   # :189-191
        cu_q.append(cu_q[-1] + sq)
        cu_k.append(cu_k[-1] + sk)
        max_q, max_k = max(max_q, sq), max(max_k, sk)
        if not s.block_table:                                           # :209-210
            continue
        for i in range(s.num_cached_blocks, s.num_blocks):              # :212-221
            start = s.block_table[i] * s.block_size
            end = start + (s.block_size if i != s.num_blocks - 1 else s.last_block_num_tokens)
            slots.extend(range(start, end))
    block_tables = prepare_block_tables(seqs) if cu_k[-1] > cu_q[-1] else None   # :225-232
    return dict(positions=np.array(positions, dtype=np.int64),
                cu_seqlens_q=np.array(cu_q, dtype=np.int32), cu_seqlens_k=np.array(cu_k, dtype=np.int32),
                max_seqlen_q=max_q, max_seqlen_k=max_k,
                slot_mapping=np.array(slots, dtype=np.int32), block_tables=block_tables)


# --------------------------------------------------------------------------- f2: (q/k norm ->) RoPE, the step before attention
def rope_cos_sin(head_dim, max_position, base):
    """fp32 table [max_position, head_dim] = cos | sin  (layers/rotary_embedding.py:29-36)."""
    inv_freq = (1.0 / (np.float32(base) ** (np.arange(0, head_dim, 2, dtype=np.float32) / np.float32(head_dim)))).astype(np.float32)
    freqs = np.outer(np.arange(max_position, dtype=np.float32), inv_freq).astype(np.float32)
    return np.concatenate([np.cos(freqs), np.sin(freqs)], axis=-1).astype(np.float32)


def rope_neox(x, positions, cos_sin):
    """apply_rotary_emb (layers/rotary_embedding.py:6-16): x [N, heads, D] (bf16-valued); fp32 products and sums
    rounded separately, result rounded to bf16 (returned as bf16-valued float32)."""
    x = np.asarray(x, dtype=np.float32)
    d = x.shape[-1]
    cs = cos_sin[np.asarray(positions)]
    cos, sin = cs[:, None, : d // 2], cs[:, None, d // 2:]
    x1, x2 = x[..., : d // 2], x[..., d // 2:]
    y1 = (x1 * cos).astype(np.float32) - (x2 * sin).astype(np.float32)
    y2 = (x2 * cos).astype(np.float32) + (x1 * sin).astype(np.float32)
    return round_to_bf16(np.concatenate([y1, y2], axis=-1).astype(np.float32))


def rms_norm_heads(x, weight, eps):
    """RMSNorm.rms_forward (layers/layernorm.py:17-27) over the last dim: fp32 normalise, round to bf16, multiply
    by the bf16 weight in bf16."""
    x = np.asarray(x, dtype=np.float32)
    var = np.mean(x.astype(np.float64) ** 2, axis=-1, keepdims=True)
    normed = round_to_bf16((x * (1.0 / np.sqrt(var + eps))).astype(np.float32))
    return round_to_bf16((normed * np.asarray(weight, dtype=np.float32)).astype(np.float32))
