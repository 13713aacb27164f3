"""Pin the CPU oracle (oracle/oracle.py) to the reference's own outputs.

The golden vectors in tests/golden/ were produced by oracle/gen_golden.py, which ran the
reference's sdpa.math functions (attention_sdpa.py:65-182) and metadata producers
(model_runner.py:160-269) on CPU in the build container.  fp32 bar: 2e-6 abs.
"""
import numpy as np
import pytest

from oracle import oracle as O

DECODE = ["decode_q2_0p5b.npz", "decode_q2_0p5b_graphpad.npz", "decode_q2_7b_tp4.npz",
          "decode_g2_d128.npz", "decode_single.npz"]
PREFILL = ["prefill_q2_0p5b.npz", "prefill_g2_d128.npz", "prefill_q2_7b_tp4.npz"]
STORE = ["store_d64.npz", "store_d128.npz"]
TOL = 2e-6


@pytest.mark.parametrize("name", DECODE)
def test_decode_matches_reference(golden, name):
    g = golden(name)
    out = O.paged_decode(O.bf16_bits_to_f32(g["q"]), O.bf16_bits_to_f32(g["k_cache"]),
                         O.bf16_bits_to_f32(g["v_cache"]), g["context_lens"], g["block_tables"])
    assert out.shape == g["expected"].shape
    assert np.abs(out - g["expected"]).max() <= TOL
    zero_rows = np.where(g["context_lens"] == 0)[0]
    for b in zero_rows:                                       # ctx == 0 -> zeros, as the reference returns
        assert not g["expected"][b].any() and not out[b].any()


@pytest.mark.parametrize("name", PREFILL)
def test_prefill_matches_reference(golden, name):
    g = golden(name)
    out = O.prefill_varlen(O.bf16_bits_to_f32(g["q"]), O.bf16_bits_to_f32(g["k"]), O.bf16_bits_to_f32(g["v"]),
                           g["cu_seqlens"], g["cu_seqlens"])
    assert np.abs(out - g["expected"]).max() <= TOL


@pytest.mark.parametrize("name", PREFILL)
def test_paged_prefill_equals_varlen_without_prefix(golden, name):
    """a4 restatement reduces to a3 when nothing is cached (Sq == Sk): scatter K/V into a paged
    cache with the oracle's own store, then read it back through block tables."""
    g = golden(name)
    q, k, v = (O.bf16_bits_to_f32(g[n]) for n in ("q", "k", "v"))
    cu = g["cu_seqlens"]
    bs = 256
    lens = np.diff(cu)
    need = [(int(n) + bs - 1) // bs for n in lens]
    nb = sum(need) + 1
    rng = np.random.default_rng(0)
    ids = rng.permutation(nb)[:sum(need)].tolist()
    bt = np.full((len(lens), max(need)), -1, np.int32)
    slots, it = [], iter(ids)
    for i, n in enumerate(lens):
        for j in range(need[i]):
            bt[i, j] = next(it)
        slots += [bt[i, t // bs] * bs + t % bs for t in range(int(n))]
    kc = np.zeros((nb, bs) + k.shape[1:], np.float32)
    vc = np.zeros_like(kc)
    O.store_kvcache(k, v, kc, vc, slots)
    out = O.paged_prefill(q, kc, vc, cu, cu, bt)
    assert np.abs(out - g["expected"]).max() <= TOL


@pytest.mark.parametrize("name", STORE)
def test_store_matches_statement(golden, name):
    g = golden(name)
    h, kvh, d, bs = (int(x) for x in g["shape"])
    qkv = g["qkv"]
    n = qkv.shape[0]
    k = qkv[:, h * d:(h + kvh) * d].reshape(n, kvh, d)
    v = qkv[:, (h + kvh) * d:].reshape(n, kvh, d)
    kc, vc = g["k_cache"].copy(), g["v_cache"].copy()
    O.store_kvcache(k, v, kc, vc, g["slot_mapping"])
    assert np.array_equal(kc, g["k_cache_expected"]) and np.array_equal(vc, g["v_cache_expected"])
    assert (g["slot_mapping"] < 0).any()


def _seqs(tokens, tables, cached=None):
    cached = cached if cached is not None else [0] * len(tokens)
    return [O.SeqState(int(n), [int(x) for x in row if x != -9], int(c)) for n, row, c in zip(tokens, tables, cached)]


def test_metadata_matches_reference(golden):
    g = golden("meta_runner.npz")
    pos, slots, ctx, bt = O.prepare_decode(_seqs(g["dec_tokens"], g["dec_tables_in"]))
    assert np.array_equal(pos, g["dec_positions"]) and np.array_equal(slots, g["dec_slot_mapping"])
    assert np.array_equal(ctx, g["dec_context_lens"]) and np.array_equal(bt, g["dec_block_tables"])
    assert slots.dtype == np.int32 and ctx.dtype == np.int32 and bt.dtype == np.int32
    for tag in ("pre", "pfx"):
        m = O.prepare_prefill(_seqs(g[f"{tag}_tokens"], g[f"{tag}_tables_in"], g[f"{tag}_cached"]))
        assert np.array_equal(m["positions"], g[f"{tag}_positions"])
        assert np.array_equal(m["cu_seqlens_q"], g[f"{tag}_cu_seqlens_q"])
        assert np.array_equal(m["cu_seqlens_k"], g[f"{tag}_cu_seqlens_k"])
        assert [m["max_seqlen_q"], m["max_seqlen_k"]] == g[f"{tag}_max_seqlen"].tolist()
        assert np.array_equal(m["slot_mapping"], g[f"{tag}_slot_mapping"])
        if g[f"{tag}_block_tables"].size:
            assert np.array_equal(m["block_tables"], g[f"{tag}_block_tables"])
        else:
            assert m["block_tables"] is None
    assert g["pfx_block_tables"].size and not g["pre_block_tables"].size


def test_bf16_roundtrip_helpers():
    x = np.array([0.0, 1.0, -1.5, 3.14159, 1e-3, 65504.0, 1.00390625], np.float32)
    r = O.round_to_bf16(x)
    assert np.array_equal(O.round_to_bf16(r), r)
    assert np.all(np.abs(r - x) <= np.abs(x) * 2.0 ** -8)
    assert O.f32_to_bf16_bits(np.array([1.00390625], np.float32))[0] == 0x3F80   # tie -> even


@pytest.mark.parametrize("name", DECODE)
def test_sdpa_math_cpu_port_matches_reference(golden, name):
    """The torch port timed as `cpu_baseline` (oracle/sdpa_math_cpu.py) equals the reference's output."""
    import torch
    from oracle.sdpa_math_cpu import flash_attn_with_kvcache_cpu
    g = golden(name)
    q = torch.from_numpy(O.bf16_bits_to_f32(g["q"])).unsqueeze(1)
    out = flash_attn_with_kvcache_cpu(q, torch.from_numpy(O.bf16_bits_to_f32(g["k_cache"])),
                                      torch.from_numpy(O.bf16_bits_to_f32(g["v_cache"])),
                                      torch.from_numpy(g["context_lens"]), torch.from_numpy(g["block_tables"]))
    assert np.abs(out[:, 0].numpy() - g["expected"]).max() <= TOL


@pytest.mark.parametrize("name", PREFILL)
def test_sdpa_math_cpu_prefill_port_matches_reference(golden, name):
    """The torch port of the reference's prefill timed in `cpu_baseline` (config 1 sample) equals the reference's output."""
    import torch
    from oracle.sdpa_math_cpu import flash_attn_varlen_func_cpu
    g = golden(name)
    cu = torch.from_numpy(g["cu_seqlens"])
    out = flash_attn_varlen_func_cpu(torch.from_numpy(O.bf16_bits_to_f32(g["q"])), torch.from_numpy(O.bf16_bits_to_f32(g["k"])),
                                     torch.from_numpy(O.bf16_bits_to_f32(g["v"])), cu, cu)
    assert np.abs(out.numpy() - g["expected"]).max() <= TOL


@pytest.mark.parametrize("name", ["rope_d64.npz", "rope_d128.npz"])
def test_rope_restatement_matches_reference(golden, name):
    """oracle.rope_neox / rope_cos_sin against the reference's apply_rotary_emb + cos_sin_cache (bit-exact bf16)."""
    g = golden(name)
    heads, d, max_pos = (int(x) for x in g["shape"])
    import torch
    base = float(g["base"][0])
    # the reference's table, built with the same torch ops as rotary_embedding.py:29-36
    inv_freq = 1.0 / (base ** (torch.arange(0, d, 2, dtype=torch.float) / d))
    freqs = torch.einsum("i,j -> ij", torch.arange(max_pos, dtype=torch.float), inv_freq)
    table = torch.cat((freqs.cos(), freqs.sin()), dim=-1).numpy()
    assert np.array_equal(table[:64], g["cos_sin"])
    got = O.rope_neox(O.bf16_bits_to_f32(g["x"]), g["positions"], table)
    assert np.array_equal(got, O.bf16_bits_to_f32(g["expected"]))                 # bit-exact
    # the pure-numpy table agrees to fp32 argument-reduction accuracy (pos * inv_freq in fp32 at positions up to 4096)
    assert np.abs(O.rope_cos_sin(d, max_pos, base) - table).max() <= 5e-4
