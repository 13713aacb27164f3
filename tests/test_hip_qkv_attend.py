"""nvh_qkv_rope_attend: the qkv projection (+ folded norm, bias, RoPE, K/V store) and the decode attention on its output in ONE
launch (csrc/qkv_attend.hip) against (a) the same call as two launches — q and the cache rows must be the SAME BITS, and so must the
attention output once the stand-alone call runs the same formulation (128-token passes: NVH_DECODE_CHUNKED_P128; nvh_paged_decode
picks 256-token passes for some chunk counts, which only changes the order fp32 partials are summed in) —
and (b) the CPU oracle (attention_sdpa.py:122-182 restated in oracle/oracle.py) on the cache state the projection leaves.
What the reference does here: models/qwen3.py:104-117 with layers/attention.py:84-86 and :99-101 inside."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(B, H, KVH, K, ctxs, seed, width=None, bs=256, slots_minus_one=()):
    """A decode step's inputs: packed residual rows, folded qkv weights, bias, positions, block tables with shuffled block ids,
    caches pre-filled for tokens < ctx - 1 (garbage elsewhere, NaN bit patterns included: nothing past the live range may be read)."""
    from nanovllm_hip import ops
    from nanovllm_hip.models.qwen import cos_sin_table
    D = 64
    g = torch.Generator().manual_seed(seed)
    ctxs = list(ctxs)
    assert len(ctxs) == B
    need = [(c + bs - 1) // bs for c in ctxs]
    width = width or max(1, max(need))
    nb = sum(need) + 2
    ids = torch.randperm(nb, generator=g).tolist()
    bt = torch.full((B, width), -1, dtype=torch.int32)
    it = iter(ids)
    for i in range(B):
        for j in range(need[i]):
            bt[i, j] = next(it)
    kc = torch.randn(nb, bs, KVH, D, generator=g).bfloat16()
    vc = torch.randn(nb, bs, KVH, D, generator=g).bfloat16()
    # rows at and past ctx - 1 hold a NaN bit pattern: reading one of them as data would poison the output
    nan = torch.tensor(float("nan")).bfloat16()
    slots = torch.full((B,), -1, dtype=torch.int32)
    for i, c in enumerate(ctxs):
        if c == 0:
            continue
        t = c - 1
        blk, off = int(bt[i, t // bs]), t % bs
        kc[blk, off:] = nan
        vc[blk, off:] = nan
        for j in range(t // bs + 1, need[i]):
            kc[int(bt[i, j])] = nan
            vc[int(bt[i, j])] = nan
        if i not in slots_minus_one:
            slots[i] = blk * bs + off
    n = (H + 2 * KVH) * D
    x = torch.randn(B, K, generator=g).bfloat16()
    w = (torch.randn(n, K, generator=g) * 0.05).bfloat16()
    bias = torch.randn(n, generator=g).bfloat16()
    pos = torch.tensor([max(c - 1, 0) for c in ctxs], dtype=torch.int64)
    dev = "cuda"
    return dict(B=B, H=H, KVH=KVH, D=D, K=K, bs=bs, ctxs=ctxs, x=x.to(dev), xp=ops.pack_rows(x.to(dev)), w=w.to(dev), bias=bias.to(dev),
                pos=pos.to(dev), slots=slots.to(dev), bt=bt.to(dev), cl=torch.tensor(ctxs, dtype=torch.int32).to(dev), kc=kc.to(dev), vc=vc.to(dev),
                table=cos_sin_table(D, 8192, 1e6, dev))


def _run(c, mode, kc, vc, ws=None, **kw):
    from nanovllm_hip import ops
    rope = dict(positions=c["pos"], cos_sin=c["table"], k_cache=kc, v_cache=vc, slot_mapping=c["slots"], num_heads=c["H"],
                num_kv_heads=c["KVH"], head_dim=c["D"])
    packed = torch.zeros(((c["B"] + 15) // 16) * 16 * c["H"] * c["D"], dtype=torch.bfloat16, device="cuda")
    q, o, fused = ops.qkv_rope_attend(c["xp"], c["w"], rope=rope, context_lens=c["cl"], block_tables=c["bt"], bias=c["bias"], norm_eps=1e-6,
                                      x_packed_rows=c["B"], attn_out_packed=packed, mode=mode, workspace=ws, **kw)
    return q, o, packed, fused


def _bits(t):
    return t.view(torch.int16) if t.dtype == torch.bfloat16 else t


def _check_against_two_launches(c, ws=None):
    from nanovllm_hip import ops
    kc1, vc1, kc2, vc2 = c["kc"].clone(), c["vc"].clone(), c["kc"].clone(), c["vc"].clone()
    q1, o1, p1, f1 = _run(c, "two_launches", kc1, vc1, ws)
    q2, o2, p2, f2 = _run(c, "one_launch", kc2, vc2, ws)
    torch.cuda.synchronize()
    assert not f1 and f2
    assert ops.qkv_rope_attend_status(ws, "cuda") == 0
    assert torch.equal(_bits(q1), _bits(q2)), "q rows differ"
    assert torch.equal(_bits(kc1), _bits(kc2)) and torch.equal(_bits(vc1), _bits(vc2)), "cache rows differ"
    live = [i for i, n in enumerate(c["ctxs"]) if n > 0]
    assert not torch.isnan(o2[live].float()).any(), "a row past the live range (NaN-filled) was read"
    H, D = c["H"], c["D"]
    ref = ops.flash_attn_with_kvcache(q1.view(-1, H, D), kc1, vc1, c["cl"], c["bt"], variant="chunked_p128")
    torch.cuda.synchronize()
    assert torch.equal(_bits(ref), _bits(o2)), "attention output differs from the stand-alone kernel of the same formulation"
    assert torch.equal(_bits(ops.unpack_rows(p2, c["B"], H * D)), _bits(o2.view(c["B"], H * D))), "fragment-packed attention output differs"
    # the default stand-alone call may sum its partials in another order (256-token passes): bf16 rounding noise at most
    d = (o1.float() - o2.float()).abs()
    assert (d[live] <= 2.0 ** -7 * o1.float().abs()[live] + 1e-3).all()
    return q2, o2, kc2, vc2


@pytest.mark.parametrize("B,H,KVH,K,ctxs", [
    (32, 14, 2, 896, [1025 + 31 * i for i in range(32)]),                     # BASELINE config 2 shapes, contexts 1025 .. 1986
    (32, 14, 2, 896, [2048] * 32),
    (7, 14, 2, 896, [1, 2, 16, 17, 255, 256, 257]),                           # the newest token first / last in its tile, block edges
    (5, 14, 2, 896, [0, 129, 0, 513, 1]),                                     # graph-padding rows (context 0, slot -1)
    (33, 16, 8, 1024, [300 + 7 * i for i in range(33)]),                      # 8 kv heads, 3 row tiles, k = 1024 (16 pieces)
    (64, 7, 1, 512, [2049 + 32 * i for i in range(64)]),                      # config-4-like group (7 q heads on 1 kv head), config 3 contexts
    (1, 14, 2, 896, [4096]),                                                  # one sequence over 128 chunks
    (16, 14, 2, 896, [113, 128, 129, 144, 145, 2047, 2033, 1, 15, 31, 32, 33, 1024, 1040, 1041, 3000]),
])
def test_one_launch_equals_two_launches_and_oracle(B, H, KVH, K, ctxs):
    from oracle import oracle as O
    c = _case(B, H, KVH, K, ctxs, seed=B * 131 + K, width=16 if max(ctxs) <= 4096 else None)
    q, o, kc, vc = _check_against_two_launches(c)
    # against the oracle, on the cache state the projection left (bf16 output: 1e-3 abs + one bf16 ulp, as every decode parity test)
    live = [i for i, n in enumerate(ctxs) if n > 0]
    kcn, vcn = np.nan_to_num(kc.float().cpu().numpy()), np.nan_to_num(vc.float().cpu().numpy())
    exp = O.paged_decode(q.float().cpu().view(B, H, 64).numpy(), kcn, vcn, np.array(ctxs, np.int32), c["bt"].cpu().numpy())
    got = o.float().cpu().numpy()
    err = np.abs(got - exp)
    assert (err[live] <= 1e-3 + 2.0 ** -8 * np.abs(exp[live])).all(), f"max err {err[live].max()}"
    for i, n in enumerate(ctxs):
        if n == 0:
            assert (got[i] == 0).all()


def test_slot_minus_one_rows_store_nothing():
    """A row whose slot is -1 (eager padding) stores no K / V row; its attention then reads whatever the cache holds at ctx - 1
    in both forms alike (the reference never runs such a row with a live context; the two forms must still agree)."""
    c = _case(6, 14, 2, 896, [40, 300, 17, 1, 700, 256], seed=5, slots_minus_one=(1, 4))
    # make the un-stored rows finite so that the comparison is meaningful
    c["kc"] = torch.nan_to_num(c["kc"].float(), nan=0.25).bfloat16()
    c["vc"] = torch.nan_to_num(c["vc"].float(), nan=-0.5).bfloat16()
    _check_against_two_launches(c)


def test_graph_replay_with_moving_contexts():
    """One captured one-launch call replayed while the contexts advance (metadata updated in place between replays, as a decode
    session does): every replay equals the eager two-launch call on a copy of the same state; the hand-off counters return to zero."""
    from nanovllm_hip import ops
    B, H, KVH, K, bs = 32, 14, 2, 896, 256
    ctx0 = [1 + 37 * i for i in range(B)]
    c = _case(B, H, KVH, K, [n + 70 for n in ctx0], seed=9, width=8)
    c["kc"] = torch.nan_to_num(c["kc"].float(), nan=0.125).bfloat16()
    c["vc"] = torch.nan_to_num(c["vc"].float(), nan=0.375).bfloat16()
    ws = torch.zeros(ops.decode_workspace_bytes(B, H, 64, 8, bs), dtype=torch.uint8, device="cuda")
    kcA, vcA, kcB, vcB = c["kc"].clone(), c["vc"].clone(), c["kc"].clone(), c["vc"].clone()

    def set_step(s):
        ctxs = [n + s for n in ctx0]
        c["cl"].copy_(torch.tensor(ctxs, dtype=torch.int32))
        c["pos"].copy_(torch.tensor([n - 1 for n in ctxs]))
        bt = c["bt"].cpu()
        c["slots"].copy_(torch.tensor([int(bt[i, (n - 1) // bs]) * bs + (n - 1) % bs for i, n in enumerate(ctxs)], dtype=torch.int32))
        c["xp"].copy_(ops.pack_rows(torch.randn(B, K, generator=torch.Generator().manual_seed(100 + s)).bfloat16().cuda()))

    set_step(0)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        _run(c, "one_launch", kcA.clone(), vcA.clone(), ws)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        qg, og, pg, fg = _run(c, "one_launch", kcA, vcA, ws)
    assert fg
    for s in range(70):
        set_step(s)
        graph.replay()
        q1, o1, p1, _ = _run(c, "two_launches", kcB, vcB)
        ref = ops.flash_attn_with_kvcache(q1.view(B, H, 64), kcB, vcB, c["cl"], c["bt"], variant="chunked_p128")
        torch.cuda.synchronize()
        assert torch.equal(_bits(qg), _bits(q1)) and torch.equal(_bits(og), _bits(ref)), f"step {s}"
        assert torch.equal(_bits(ops.unpack_rows(pg, B, H * 64)), _bits(og.view(B, H * 64))), f"step {s}"
    assert torch.equal(_bits(kcA), _bits(kcB)) and torch.equal(_bits(vcA), _bits(vcB))
    hdr = ws[65536:65536 + 8192].view(torch.int32)
    assert int(hdr.abs().sum()) == 0, "ready / done counters or the status word were left non-zero"


def test_unsupported_shapes_run_as_two_launches():
    """head_dim 128 (and k > 1024) is not served by the one-launch kernel: mode auto falls back, mode one_launch refuses."""
    from nanovllm_hip import ops
    from nanovllm_hip.models.qwen import cos_sin_table
    B, H, KVH, D, K, bs = 4, 7, 1, 128, 1024, 256
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, K, generator=g).bfloat16().cuda()
    w = (torch.randn((H + 2 * KVH) * D, K, generator=g) * 0.05).bfloat16().cuda()
    kc = torch.randn(6, bs, KVH, D, generator=g).bfloat16().cuda()
    vc = torch.randn(6, bs, KVH, D, generator=g).bfloat16().cuda()
    bt = torch.tensor([[0, 1], [2, -1], [3, -1], [4, 5]], dtype=torch.int32).cuda()
    ctxs = [300, 7, 256, 257]
    cl = torch.tensor(ctxs, dtype=torch.int32).cuda()
    pos = torch.tensor([n - 1 for n in ctxs]).cuda()
    slots = torch.tensor([int(bt[i, (n - 1) // bs]) * bs + (n - 1) % bs for i, n in enumerate(ctxs)], dtype=torch.int32).cuda()
    rope = dict(positions=pos, cos_sin=cos_sin_table(D, 4096, 1e6, "cuda"), k_cache=kc, v_cache=vc, slot_mapping=slots, num_heads=H, num_kv_heads=KVH, head_dim=D)
    q, o, fused = ops.qkv_rope_attend(ops.pack_rows(x), w, rope=rope, context_lens=cl, block_tables=bt, x_packed_rows=B)
    torch.cuda.synchronize()
    assert not fused
    ref = ops.flash_attn_with_kvcache(q.view(B, H, D), kc, vc, cl, bt)
    assert torch.equal(_bits(ref), _bits(o))
    with pytest.raises(RuntimeError, match="one launch"):
        ops.qkv_rope_attend(ops.pack_rows(x), w, rope=rope, context_lens=cl, block_tables=bt, x_packed_rows=B, mode="one_launch")


def test_a_wait_that_runs_out_gives_nan_and_a_status_not_a_hang():
    """Consumers told to wait for a producer that does not exist give up after `spin_limit` polls: NaN rows and the status word,
    never a plausible number; the counters are back at zero, and after clearing the status the next call is clean."""
    from nanovllm_hip import ops
    c = _case(8, 14, 2, 896, [100, 0, 1300, 17, 256, 257, 1, 900], seed=21)
    ws = torch.zeros(ops.decode_workspace_bytes(8, 14, 64, c["bt"].shape[1], 256), dtype=torch.uint8, device="cuda")
    kc, vc = c["kc"].clone(), c["vc"].clone()
    q, o, p, fused = _run(c, "one_launch", kc, vc, ws, spin_limit=50, missing_producers=1)
    torch.cuda.synchronize()
    assert fused and ops.qkv_rope_attend_status(ws) != 0
    live = [i for i, n in enumerate(c["ctxs"]) if n > 0]
    assert torch.isnan(o[live].float()).all(), "a timed-out row must be NaN"
    assert (o[1].float() == 0).all()
    hdr = ws[65536:65536 + 8192].view(torch.int32)
    assert int(hdr[:2048 // 4 * 2].abs().sum()) == 0, "ready / done counters must be zero after the launch"
    ws[65536 + 4096:65536 + 4100].zero_()
    _check_against_two_launches(c, ws)


def test_kv_prefetch_form_changes_nothing():
    """Mode 3 (the projection launch's idle CUs touch the attention launch's first K/V images) is a pure hint: same bits as mode 1."""
    c = _case(32, 14, 2, 896, [1025 + 31 * i for i in range(32)], seed=77, width=16)
    kc1, vc1, kc2, vc2 = c["kc"].clone(), c["vc"].clone(), c["kc"].clone(), c["vc"].clone()
    q1, o1, p1, f1 = _run(c, "two_launches", kc1, vc1)
    q2, o2, p2, f2 = _run(c, "two_launches_kv_prefetch", kc2, vc2, spin_limit=2)
    torch.cuda.synchronize()
    assert not f1 and not f2
    assert torch.equal(_bits(q1), _bits(q2)) and torch.equal(_bits(o1), _bits(o2)) and torch.equal(_bits(p1), _bits(p2))
    assert torch.equal(_bits(kc1), _bits(kc2)) and torch.equal(_bits(vc1), _bits(vc2))
