"""Pin the PRODUCT's metadata producers (nanovllm_hip/engine/model_runner.py build_* + Sequence + BlockManager) to the
reference's own outputs: tests/golden/meta_runner.npz holds what nanovllm.engine.model_runner.prepare_decode /
prepare_prefill (:160-269) returned in the build container, plus a 300-step decode trajectory driven through the reference's
Sequence.append_token / BlockManager.may_append (scheduler.py:60-110, block_manager.py:62-159).  Integer work: bit-exact,
dtypes included.  (The oracle's copy of the same producers is pinned in test_oracle_golden.py.)"""
import numpy as np
import pytest
import torch

from nanovllm_hip.engine.block_manager import BlockManager
from nanovllm_hip.engine.model_runner import build_block_tables, build_decode_meta, build_prefill_meta
from nanovllm_hip.engine.sequence import Sequence
from oracle import oracle as O


def _mkseq(n_tokens, table_row, cached=0):
    s = Sequence(list(range(int(n_tokens))))            # the generator used token ids 0..n-1 (oracle/gen_golden.py gen_meta)
    s.block_table = [int(x) for x in table_row if x != -9]
    s.num_cached_tokens = int(cached)
    return s


def _same(t, ref):
    assert isinstance(t, torch.Tensor)
    a = t.numpy()
    assert a.dtype == ref.dtype, (a.dtype, ref.dtype)
    assert a.shape == ref.shape and np.array_equal(a, ref)


def test_decode_meta_equals_reference(golden):
    g = golden("meta_runner.npz")
    seqs = [_mkseq(n, row) for n, row in zip(g["dec_tokens"], g["dec_tables_in"])]
    m = build_decode_meta(seqs)
    _same(m["input_ids"], g["dec_input_ids"])
    _same(m["positions"], g["dec_positions"])
    _same(m["slot_mapping"], g["dec_slot_mapping"])
    _same(m["context_lens"], g["dec_context_lens"])
    _same(m["block_tables"], g["dec_block_tables"])
    assert (g["dec_block_tables"] == -1).any()            # the -1 right padding is exercised
    _same(build_block_tables(seqs), g["dec_block_tables"])


@pytest.mark.parametrize("tag", ["pre", "pfx"])
def test_prefill_meta_equals_reference(golden, tag):
    g = golden("meta_runner.npz")
    seqs = [_mkseq(n, row, c) for n, row, c in zip(g[f"{tag}_tokens"], g[f"{tag}_tables_in"], g[f"{tag}_cached"])]
    m = build_prefill_meta(seqs)
    _same(m["input_ids"], g[f"{tag}_input_ids"])
    _same(m["positions"], g[f"{tag}_positions"])
    _same(m["cu_seqlens_q"], g[f"{tag}_cu_seqlens_q"])
    _same(m["cu_seqlens_k"], g[f"{tag}_cu_seqlens_k"])
    assert [m["max_seqlen_q"], m["max_seqlen_k"]] == g[f"{tag}_max_seqlen"].tolist()
    _same(m["slot_mapping"], g[f"{tag}_slot_mapping"])
    if tag == "pfx":                                      # num_cached_tokens > 0 somewhere: block tables are handed over
        assert any(s.num_cached_tokens for s in seqs)
        _same(m["block_tables"], g["pfx_block_tables"])
    else:
        assert g["pre_block_tables"].size == 0 and m["block_tables"] is None


def test_prefill_meta_of_warmup_sequences_has_no_slots():
    """Sequences without a block table (the start-up warmup, model_runner.py:107-121,209-210): no slot is produced."""
    m = build_prefill_meta([Sequence([0] * 40), Sequence([0] * 7)])
    assert m["slot_mapping"].numel() == 0 and m["cu_seqlens_q"].tolist() == [0, 40, 47] and m["block_tables"] is None


def _trajectory(golden):
    g = golden("meta_runner.npz")
    gen = torch.Generator().manual_seed(77)                # same draws as gen_meta: prompts first, then the step tokens
    prompts = [torch.randint(0, 10000, (int(n),), generator=gen).tolist() for n in g["traj_prompt_lens"]]
    toks = torch.randint(0, 10000, g["traj_tokens"].shape, generator=gen)
    assert np.array_equal(toks.numpy(), g["traj_tokens"])
    return g, prompts, g["traj_tokens"]


def test_decode_trajectory_equals_reference(golden):
    """Product Sequence + BlockManager.allocate / may_append + build_decode_meta, step by step over 300 steps (every
    sequence crosses at least one block boundary), against the reference's own trajectory."""
    g, prompts, toks = _trajectory(golden)
    bm = BlockManager(64, 256)
    seqs = []
    for p in prompts:
        s = Sequence(p)
        bm.allocate(s)
        seqs.append(s)
    for s, t in zip(seqs, toks[0]):
        s.append_token(int(t))
    for i in range(g["traj_input_ids"].shape[0]):
        for s in seqs:
            assert bm.can_append(s)
            bm.may_append(s)
        m = build_decode_meta(seqs)
        _same(m["input_ids"], g["traj_input_ids"][i])
        _same(m["positions"], g["traj_positions"][i])
        _same(m["slot_mapping"], g["traj_slot_mapping"][i])
        _same(m["context_lens"], g["traj_context_lens"][i])
        for s, t in zip(seqs, toks[i + 1]):
            s.append_token(int(t))
    _same(build_block_tables(seqs), g["traj_final_tables"])


def test_decode_meta_with_reserved_blocks_equals_reference(golden):
    """The graph-replayed session books every block up front (BlockManager.allocate(reserve_tokens=...)): the slot must then
    come from the block of the LAST TOKEN, not from block_table[-1] — same integers as the reference's growing tables."""
    g, prompts, toks = _trajectory(golden)
    final = g["traj_final_tables"]
    seqs = []
    for p, row in zip(prompts, final):
        s = Sequence(p)
        s.block_table = [int(x) for x in row if x >= 0]
        seqs.append(s)
    for s, t in zip(seqs, toks[0]):
        s.append_token(int(t))
    for i in (0, 1, 2, 150, 299):
        while len(seqs[0]) < int(g["traj_context_lens"][i][0]):
            k = len(seqs[0]) - int(g["traj_context_lens"][0][0]) + 1
            for s, t in zip(seqs, toks[k]):
                s.append_token(int(t))
        m = build_decode_meta(seqs)
        _same(m["slot_mapping"], g["traj_slot_mapping"][i])
        _same(m["context_lens"], g["traj_context_lens"][i])
        _same(m["positions"], g["traj_positions"][i])
        _same(m["input_ids"], g["traj_input_ids"][i])


def test_oracle_trajectory_equals_reference(golden):
    """The oracle's prepare_decode (the checker the GPU advance test leans on) on the same trajectory."""
    g, prompts, toks = _trajectory(golden)
    final = g["traj_final_tables"]
    for i in (0, 1, 255, 256, 299):
        lens = g["traj_context_lens"][i]
        states = [O.SeqState(int(n), [int(x) for x in row if x >= 0][: (int(n) + 255) // 256]) for n, row in zip(lens, final)]
        pos, slots, ctx, _ = O.prepare_decode(states)
        assert np.array_equal(pos, g["traj_positions"][i]) and np.array_equal(slots, g["traj_slot_mapping"][i])
        assert np.array_equal(ctx, lens)


def test_block_size_mismatch_is_refused():
    """Sequence.block_size is the reference's class constant (sequence.py:15); the producers refuse another block size
    instead of computing slots from the wrong block."""
    s = _mkseq(300, [3, 9])
    with pytest.raises(AssertionError):
        build_decode_meta([s], block_size=128)
    with pytest.raises(AssertionError):
        build_prefill_meta([s], block_size=128)
