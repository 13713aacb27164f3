"""GPU: the one-shot all-reduce over IPC-mapped peer buffers (nvh_allreduce_oneshot; replaces the dist.all_reduce of
nanovllm/layers/linear.py:185-190 at decode sizes) rehearsed with 2 and 4 PROCESSES that share cuda:0.

A one-GPU box cannot exercise xGMI, but it does exercise everything else: hipIpc export / open across processes, the flag
protocol between kernels of different processes, the epoch and the double-buffered staging over many calls (eager and replayed
from a HIP graph), rank-order summation (every rank must produce the same bits) and the fused residual-add + fragment-pack
epilogue.  Expected values are exact: every rank regenerates all ranks' inputs from seeds and sums them in fp32 in rank order."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIDDEN, MAX_ROWS = 896, 64


def _inputs(it, world, rows, hidden):
    xs = []
    for r in range(world):
        g = torch.Generator().manual_seed(100003 * it + r)
        xs.append(torch.randn(rows, hidden, generator=g).bfloat16())
    return xs


def _expected_sum(xs):
    acc = torch.zeros_like(xs[0], dtype=torch.float32)
    for x in xs:                                              # rank order, fp32, one rounding: what the kernel does
        acc = acc + x.float()
    return acc.bfloat16()


def _worker(rank, world, port, ret):
    for p in (ROOT, os.path.join(ROOT, "nano-vllm-learn_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)                                  # every rank on the one GPU of the box
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nanovllm_hip import ops
        from nanovllm_hip.distributed import OneShotAllReduce
        comm = OneShotAllReduce(MAX_ROWS, HIDDEN)
        dev = torch.device("cuda:0")
        checked = 0
        # ---- eager calls: changing row counts (different workgroup counts per call), plain and fused epilogues alternating
        for it in range(40):
            rows = (1, 17, 32, 64)[it % 4]
            xs = _inputs(it, world, rows, HIDDEN)
            exp = _expected_sum(xs)
            dist.barrier()                                    # keep the ranks within a launch of each other (bounded in-kernel spin)
            if it % 2 == 0:
                y = xs[rank].to(dev)
                comm.all_reduce(y)
                torch.cuda.synchronize()
                assert torch.equal(y.cpu(), exp), f"rank {rank} call {it}: all_reduce differs from the rank-order fp32 sum"
            else:
                g = torch.Generator().manual_seed(7 + it)
                res = torch.randn(rows, HIDDEN, generator=g).bfloat16()       # the same residual stream on every rank
                res_d, y = res.to(dev), xs[rank].to(dev)
                packed = torch.zeros(((rows + 15) // 16) * 16 * HIDDEN, dtype=torch.bfloat16, device=dev)
                comm.all_reduce_residual_add(y, res_d, packed)
                torch.cuda.synchronize()
                want = (res.float() + exp.float()).bfloat16()
                assert torch.equal(res_d.cpu(), want), f"rank {rank} call {it}: fused residual add differs"
                assert torch.equal(ops.unpack_rows(packed, rows, HIDDEN).cpu(), want), f"rank {rank} call {it}: packed copy differs"
                assert torch.equal(y.cpu(), xs[rank])          # the partial itself is left alone
            checked += 1
        # ---- the two all-reduces of a decoder layer captured in ONE HIP graph and replayed with new inputs every time
        rows = 32
        y1 = torch.zeros(rows, HIDDEN, dtype=torch.bfloat16, device=dev)
        y2 = torch.zeros_like(y1)
        res = torch.zeros_like(y1)
        dist.barrier()
        comm.all_reduce_residual_add(y1, res)                 # warm-up on the capture-free path (epoch 41, 42 on every rank)
        comm.all_reduce_residual_add(y2, res)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        dist.barrier()
        with torch.cuda.graph(graph):
            comm.all_reduce_residual_add(y1, res)
            comm.all_reduce_residual_add(y2, res)
        for it in range(100, 130):
            a, b = _inputs(it, world, rows, HIDDEN), _inputs(it + 1000, world, rows, HIDDEN)
            y1.copy_(a[rank].to(dev))
            y2.copy_(b[rank].to(dev))
            res.zero_()
            torch.cuda.synchronize()
            dist.barrier()
            graph.replay()
            torch.cuda.synchronize()
            want = (_expected_sum(a).float()).bfloat16()
            want = (want.float() + _expected_sum(b).float()).bfloat16()
            assert torch.equal(res.cpu(), want), f"rank {rank} replay {it}: graph-replayed all-reduces differ"
            checked += 1
        assert comm.failed_epoch() == 0
        dist.barrier()
        comm.close()
        ret[rank] = checked
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_oneshot_allreduce_between_processes_sharing_one_gpu(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29900 + (os.getpid() + 13 * world) % 300
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert sorted(ret.keys()) == list(range(world)) and all(v == 70 for v in ret.values())


def test_allreduce_argument_errors():
    from nanovllm_hip import _lib
    lib = _lib.load()
    assert lib.nvh_allreduce_stage_bytes(64, 896) == 2 * 64 * 896 * 2 and lib.nvh_allreduce_stage_bytes(64, 900) == 0
    assert lib.nvh_allreduce_flag_bytes(8) == 8 * 32 * 4
    x = torch.zeros(4, 896, dtype=torch.bfloat16, device="cuda")
    tab = torch.zeros(2, dtype=torch.int64, device="cuda")
    st = torch.zeros(16, dtype=torch.int32, device="cuda")
    args = (x.data_ptr(), x.data_ptr(), None, tab.data_ptr(), tab.data_ptr(), st.data_ptr())
    assert lib.nvh_allreduce_oneshot(*args, 1, 0, 4, 896, 896, 896, 1 << 20, 0, 0, None) == -2          # world of one
    assert lib.nvh_allreduce_oneshot(*args, 2, 0, 4, 896, 896, 896, 16, 0, 0, None) == -4               # staging too small
    assert lib.nvh_allreduce_oneshot(*args, 2, 0, 4, 896, 896, 896, 1 << 20, 7, 0, None) == -2          # unknown epilogue
    assert lib.nvh_allreduce_oneshot(*args, 2, 0, 0, 896, 896, 896, 1 << 20, 0, 0, None) == 0           # zero rows: nothing to do


def _engine_worker(rank, world, port, ret):
    for p in (ROOT, os.path.join(ROOT, "nano-vllm-learn_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nanovllm_hip.distributed import tensor_parallel_comm
        from nanovllm_hip.engine.llm_engine import LLMEngine
        from nanovllm_hip.models.qwen import model_config
        cfg = model_config("Qwen2-0.5B", num_hidden_layers=2, vocab_size=2048)
        g = torch.Generator().manual_seed(0)
        prompts = [torch.randint(0, 2048, (n,), generator=g).tolist() for n in (300, 17, 256, 5)]
        outs = {}
        for mode in ("oneshot", "rccl"):
            os.environ["NVH_ALLREDUCE"] = mode
            eng = LLMEngine(cfg, num_kvcache_blocks=16, enforce_eager=False, seed=1)
            assert (tensor_parallel_comm() is not None) == (mode == "oneshot")
            if mode == "oneshot":                              # the step holds only this library's launches: it must be a graph
                sess_probe = eng.runner.comm is not None
                assert sess_probe
            outs[mode] = eng.generate(prompts, max_tokens=24)
            dist.barrier()
            if eng.runner.comm is not None:
                assert eng.runner.comm.failed_epoch() == 0
        gathered = [None] * world
        dist.all_gather_object(gathered, outs)
        ret[rank] = (outs["oneshot"] == outs["rccl"], all(x == gathered[0] for x in gathered), len(outs["oneshot"][0]))
    finally:
        dist.destroy_process_group()


def test_tensor_parallel_engine_oneshot_graph_equals_collective_eager():
    """Two ranks (head split 7+7 / 1+1 of a 2-layer Qwen2-shaped model) sharing cuda:0: the decode step with the one-shot
    all-reduce INSIDE the captured HIP graph must give exactly the tokens of eager steps that all-reduce through
    torch.distributed (gloo here, RCCL on a real node): with two ranks both sum two bf16 partials in fp32 and round once.
    Every rank must hold the same tokens."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29900 + (os.getpid() + 101) % 300
    mp.spawn(_engine_worker, args=(world, port, ret), nprocs=world, join=True)
    assert sorted(ret.keys()) == [0, 1]
    for same_paths, same_ranks, n in ret.values():
        assert same_paths and same_ranks and n == 24


def _timeout_worker(rank, world, port, ret):
    for p in (ROOT, os.path.join(ROOT, "nano-vllm-learn_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nanovllm_hip import distributed as D
        comm = D.init_tensor_parallel_comm(MAX_ROWS, HIDDEN)
        assert comm is not None and D.last_measurement is None           # the choice is the self-test's alone; nothing was timed
        comm.set_spin_limit(200)                                          # ~ a millisecond instead of seconds
        dev = torch.device("cuda:0")
        x = torch.ones(8, HIDDEN, dtype=torch.bfloat16, device=dev)
        dist.barrier()
        if rank == 0:
            # three calls nobody answers: each gives up, writes NaN and marks the failure; its flags at the peer now read 3 calls ahead
            for _ in range(3):
                y = x.clone()
                comm.all_reduce(y)
                torch.cuda.synchronize()
                assert torch.isnan(y.float()).all(), "a timed-out all-reduce must not return plausible numbers"
            assert comm.failed_epoch() != 0
        dist.barrier()
        if rank == 1:
            # the LATE rank: its peer's flag satisfies the wait at once, but is two calls ahead — the slot it would read has been
            # reused; it must fail as well instead of summing a later call's bytes
            y = x.clone()
            comm.all_reduce(y)
            torch.cuda.synchronize()
            assert torch.isnan(y.float()).all(), "the late rank summed a reused slot"
            assert comm.failed_epoch() != 0
        dist.barrier()
        raised = False
        try:
            D.raise_if_failed()                                           # what the engine calls at its token readback: EVERY rank raises
        except RuntimeError as e:
            raised = "one-shot all-reduce failed" in str(e)
        D.close_tensor_parallel_comm()
        ret[rank] = raised
    finally:
        dist.destroy_process_group()


def test_peer_timeout_surfaces_as_an_error_on_every_rank():
    """a rank whose peer never arrives writes NaN and marks state[2]; the late peer finds the flag two calls ahead
    and fails too; the host-side check at a sync point raises on every rank instead of letting arg-max pick tokens from NaN rows."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29900 + (os.getpid() + 57) % 300
    mp.spawn(_timeout_worker, args=(world, port, ret), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


def _fallback_worker(rank, world, port, ret):
    for p in (ROOT, os.path.join(ROOT, "nano-vllm-learn_amd")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from nanovllm_hip import distributed as D
        from nanovllm_hip.engine.llm_engine import LLMEngine
        from nanovllm_hip.models.qwen import model_config
        if rank == 1:                                                     # IPC export fails on ONE rank only
            def boom(self, ptr):
                raise RuntimeError("injected: hipIpcGetMemHandle refused")
            D.OneShotAllReduce._export = boom
        cfg = model_config("Qwen2-0.5B", num_hidden_layers=2, vocab_size=2048)
        g = torch.Generator().manual_seed(0)
        prompts = [torch.randint(0, 2048, (n,), generator=g).tolist() for n in (300, 17, 256, 5)]
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            eng = LLMEngine(cfg, num_kvcache_blocks=16, enforce_eager=False, seed=1)
        fell_back = eng.runner.comm is None and D.tensor_parallel_comm() is None and "injected" in D.last_choice
        toks = eng.generate(prompts, max_tokens=12)
        gathered = [None] * world
        dist.all_gather_object(gathered, toks)
        ret[rank] = (fell_back, all(t == gathered[0] for t in gathered), len(toks[0]))
    finally:
        dist.destroy_process_group()


def test_ipc_failure_on_one_rank_makes_every_rank_fall_back():
    """The fallback DECISION path (torch.distributed instead of the one-shot kernel) when the IPC set-up fails on one rank: injected on
    rank 1 of a two-process rehearsal on cuda:0; both ranks must take the collective path, say why, and produce the same tokens."""
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29900 + (os.getpid() + 211) % 300
    mp.spawn(_fallback_worker, args=(world, port, ret), nprocs=world, join=True)
    assert sorted(ret.keys()) == [0, 1]
    for fell_back, same, n in ret.values():
        assert fell_back and same and n == 12
