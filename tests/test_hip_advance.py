"""GPU: the device-side between-steps bookkeeping (nvh_greedy_advance / _candidates / _candidates_embed: arg-max, append the
token, and the NEXT step's prepare_decode integers on the device) against the REFERENCE's own trajectory.

tests/golden/meta_runner.npz traj_* holds what nanovllm.engine.model_runner.prepare_decode (:244-269) returned at each of 300
decode steps while the reference's Sequence.append_token / BlockManager.may_append (scheduler.py:60-110) advanced six sequences
across block boundaries.  The device version works on STATIC block tables (all blocks booked up front, padding 0), so it must
land on the same slot / context / position integers without ever growing a table.  Integer work: bit-exact, every step."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

VOCAB = 10000


def _state(g, dev):
    final = torch.from_numpy(np.where(g["traj_final_tables"] < 0, 0, g["traj_final_tables"]).astype(np.int32)).to(dev)   # graph-style padding
    n = final.shape[0]
    st = dict(block_tables=final,
              input_ids=torch.from_numpy(g["traj_input_ids"][0]).to(dev), positions=torch.from_numpy(g["traj_positions"][0]).to(dev),
              context_lens=torch.from_numpy(g["traj_context_lens"][0]).to(dev), slot_mapping=torch.from_numpy(g["traj_slot_mapping"][0]).to(dev),
              tokens=torch.zeros(g["traj_input_ids"].shape[0] + 1, n, dtype=torch.int64, device=dev), row_steps=torch.zeros(n, dtype=torch.int64, device=dev))
    return st, n


def _check_step(g, st, i, what):
    for key in ("input_ids", "positions", "context_lens", "slot_mapping"):
        got = st[key].cpu().numpy()
        exp = g[f"traj_{key}"][i]
        assert got.dtype == exp.dtype and np.array_equal(got, exp), f"{what}: {key} differs from the reference at step {i}: {got} vs {exp}"


def test_initial_state_comes_from_the_product_producer(golden):
    """The session's starting tensors are build_decode_meta's (pinned to the reference on CPU in test_producers_cpu.py)."""
    from nanovllm_hip.engine.model_runner import build_decode_meta
    from nanovllm_hip.engine.sequence import Sequence
    g = golden("meta_runner.npz")
    gen = torch.Generator().manual_seed(77)
    seqs = []
    for n, row in zip(g["traj_prompt_lens"], g["traj_final_tables"]):
        s = Sequence(torch.randint(0, 10000, (int(n),), generator=gen).tolist())
        s.block_table = [int(x) for x in row if x >= 0]
        seqs.append(s)
    for s, t in zip(seqs, g["traj_tokens"][0]):
        s.append_token(int(t))
    m = build_decode_meta(seqs)
    for key in ("input_ids", "positions", "context_lens", "slot_mapping"):
        assert np.array_equal(m[key].numpy(), g[f"traj_{key}"][0])


@pytest.mark.parametrize("form", ["logits", "candidates", "candidates_embed"])
def test_device_advance_follows_reference_trajectory(golden, form):
    from nanovllm_hip import ops
    g = golden("meta_runner.npz")
    dev = torch.device("cuda")
    st, n = _state(g, dev)
    steps = g["traj_input_ids"].shape[0]
    toks = g["traj_tokens"]
    hidden, groups = 64, 5
    if form == "logits":
        logits = torch.zeros(n, VOCAB, dtype=torch.bfloat16, device=dev)
    else:
        cand_val = torch.zeros(groups, 64, dtype=torch.float32, device=dev)
        cand_idx = torch.zeros(groups, 64, dtype=torch.int32, device=dev)
        gen = torch.Generator().manual_seed(5)
        embed_w = torch.randn(VOCAB, hidden, generator=gen).bfloat16().to(dev)
        hid = torch.zeros(n, hidden, dtype=torch.bfloat16, device=dev)
        hid_p = torch.zeros(16 * hidden, dtype=torch.bfloat16, device=dev)
    rows = torch.arange(n, device=dev)
    for i in range(steps - 1):
        _check_step(g, st, i, form)                              # what step i's attention call would be fed
        nxt = torch.from_numpy(toks[i + 1]).to(dev)              # the token "sampled" at step i (postprocess appends it)
        if form == "logits":
            logits.zero_()
            logits[rows, nxt] = 1.0
            ops.greedy_advance(logits, st["input_ids"], st["positions"], st["context_lens"], st["slot_mapping"], st["block_tables"], 256,
                               st["tokens"], st["row_steps"])
        else:
            # the winner sits in a different candidate group per row; losers carry smaller values and other columns
            cand_val.fill_(-1.0)
            cand_idx.copy_(torch.randint(0, VOCAB, (groups, 64), generator=gen).int())
            grp = (rows + i) % groups
            cand_val[grp, rows] = 2.0
            cand_idx[grp, rows] = nxt.int()
            embed = (embed_w, hid, hid_p) if form == "candidates_embed" else None
            ops.greedy_advance_candidates(cand_val, cand_idx, groups, n, st["input_ids"], st["positions"], st["context_lens"], st["slot_mapping"],
                                          st["block_tables"], 256, st["tokens"], st["row_steps"], embed=embed)
            if embed is not None and i % 37 == 0:                # the next step's embedding rows, row-major and fragment-packed
                assert torch.equal(hid, embed_w[nxt])
                assert torch.equal(ops.unpack_rows(hid_p, n, hidden), embed_w[nxt])
    _check_step(g, st, steps - 1, form)
    torch.cuda.synchronize()
    assert torch.equal(st["tokens"][: steps - 1].cpu(), torch.from_numpy(toks[1:steps]))   # the generated-token log
    assert (st["row_steps"] == steps - 1).all()


def test_padding_rows_are_left_alone():
    """Graph padding rows (context 0, slot -1) are not advanced and log nothing."""
    from nanovllm_hip import ops
    dev = torch.device("cuda")
    logits = torch.zeros(3, 512, dtype=torch.bfloat16, device=dev)
    logits[:, 7] = 1.0
    ids = torch.tensor([5, 6, 9], device=dev)
    pos = torch.tensor([300, 0, 12], device=dev)
    ctx = torch.tensor([300, 0, 12], dtype=torch.int32, device=dev)
    slots = torch.tensor([2 * 256 + 43, -1, 256 + 11], dtype=torch.int32, device=dev)
    bt = torch.tensor([[1, 2], [0, 0], [1, 0]], dtype=torch.int32, device=dev)
    log = torch.full((2, 3), -7, dtype=torch.int64, device=dev)
    rs = torch.zeros(3, dtype=torch.int64, device=dev)
    ops.greedy_advance(logits, ids, pos, ctx, slots, bt, 256, log, rs)
    torch.cuda.synchronize()
    assert ids.tolist() == [7, 6, 7] and pos.tolist() == [301, 0, 13] and ctx.tolist() == [301, 0, 13]
    assert slots.tolist() == [2 * 256 + 44, -1, 256 + 12] and rs.tolist() == [1, 0, 1] and log[0].tolist() == [7, -7, 7]


@pytest.mark.parametrize("n", [8, 1000, 151936, 4099])
def test_argmax_edge_rows_follow_torch(n):
    """a row of -inf or NaN logits must still give a VALID index (torch.argmax: NaN counts as the maximum, ties ->
    lowest index), never the kernel's internal sentinel."""
    from nanovllm_hip import ops
    dev = torch.device("cuda")
    stride = (n + 7) // 8 * 8
    buf = torch.randn(6, stride, device=dev).bfloat16()
    x = buf[:, :n]
    x[0] = float("-inf")
    x[1] = float("nan")
    x[2, n // 2] = float("nan")
    x[3] = 0.25
    x[4, : n - 1] = float("-inf")                                # the only finite value is the last column
    x[5, n - 1] = float("nan")
    x[5, 0] = float("inf")
    got = ops.argmax_rows(x)
    ref = torch.argmax(x.float(), dim=-1)
    torch.cuda.synchronize()
    assert got.tolist() == ref.tolist() and int(got.min()) >= 0 and int(got.max()) < n


def test_greedy_advance_survives_degenerate_logits_and_candidates():
    """-inf / NaN rows through greedy_advance and greedy_advance_candidates_embed: the token written back and the embedding row
    read are valid (the INT_MAX sentinel used to reach `embed + INT_MAX * hidden`)."""
    from nanovllm_hip import ops
    dev = torch.device("cuda")
    vocab, hidden, rows = 512, 64, 4
    logits = torch.zeros(rows, vocab, dtype=torch.bfloat16, device=dev)
    logits[0] = float("-inf")
    logits[1] = float("nan")
    logits[2, 100] = float("nan")
    logits[3, 9] = 3.0

    def fresh():
        return (torch.zeros(rows, dtype=torch.int64, device=dev), torch.full((rows,), 10, dtype=torch.int64, device=dev),
                torch.full((rows,), 10, dtype=torch.int32, device=dev), torch.full((rows,), 9, dtype=torch.int32, device=dev),
                torch.zeros(rows, 1, dtype=torch.int32, device=dev), torch.zeros(2, rows, dtype=torch.int64, device=dev),
                torch.zeros(rows, dtype=torch.int64, device=dev))

    ids, pos, ctx, slots, bt, log, rs = fresh()
    ops.greedy_advance(logits, ids, pos, ctx, slots, bt, 256, log, rs)
    torch.cuda.synchronize()
    assert ids.tolist() == torch.argmax(logits.float(), dim=-1).tolist() == [0, 0, 100, 9]

    groups = 3
    cv = torch.full((groups, 64), float("-inf"), dtype=torch.float32, device=dev)
    ci = torch.full((groups, 64), 0x7fffffff, dtype=torch.int32, device=dev)       # what a workgroup that saw only -inf / nothing would hold
    ci[:, 0] = torch.tensor([300, 40, 77], dtype=torch.int32, device=dev)          # row 0: all -inf, valid columns -> lowest column
    cv[1, 1] = float("nan"); ci[1, 1] = 123                                        # row 1: NaN wins
    cv[2, 2] = 1.0; ci[2, 2] = 200; cv[0, 2] = 1.0; ci[0, 2] = 150                 # row 2: tie -> lowest column
    embed_w = torch.randn(vocab, hidden, device=dev).bfloat16()                    # row 3: nothing but sentinels -> clamped into the table
    hid = torch.zeros(rows, hidden, dtype=torch.bfloat16, device=dev)
    ids, pos, ctx, slots, bt, log, rs = fresh()
    ops.greedy_advance_candidates(cv, ci, groups, rows, ids, pos, ctx, slots, bt, 256, log, rs, embed=(embed_w, hid, None))
    torch.cuda.synchronize()
    assert ids.tolist()[:3] == [40, 123, 150] and 0 <= ids[3].item() < vocab
    assert torch.equal(hid, embed_w[ids])
