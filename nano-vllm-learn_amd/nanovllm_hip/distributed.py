"""Tensor-parallel all-reduce of the row-parallel projections: where the reference calls `dist.all_reduce(y)` in
RowParallelLinear.forward (nanovllm/layers/linear.py:185-190; NCCL), this package offers two interchangeable paths, chosen ONCE at
start-up, identically on every rank:

  * `OneShotAllReduce` — the MI355X-native path for decode-sized messages: every rank maps every peer's staging buffer and flag
    table through hipIpc handles and the HIP kernel behind `nvh_allreduce_oneshot` (csrc/allreduce_oneshot.hip) reads the p-1
    peers over p-1 distinct xGMI links at once, sums in rank order and applies the residual add that follows (layernorm.py:35-36)
    in the same launch.  Capture-safe: it can sit inside the HIP graph of a decode step.
  * RCCL through torch.distributed (`backend="nccl"`), the fallback: used when the one-shot set-up is not available on some rank
    (no IPC between the processes, a message larger than the staging buffers, a world of one).

The handle exchange uses the process group that already exists (all_gather_object: a host-side, one-time exchange of 64-byte
handles); the data path never touches torch.distributed.
"""
from __future__ import annotations

import ctypes

import torch
import torch.distributed as dist

from . import _lib
from ._lib import AR_EPI_NONE, AR_EPI_RESIDUAL_ADD, IPC_HANDLE_BYTES, NVH_BF16


class OneShotAllReduce:
    """All-reduce (sum) of [rows <= max_rows, hidden] bf16 tensors over the ranks of `group`, one process per GPU."""

    def __init__(self, max_rows: int, hidden: int, group=None, device=None):
        assert dist.is_initialized(), "needs an initialised process group (one process per GPU)"
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.max_rows, self.hidden = max_rows, hidden
        lib = _lib.load()
        self._lib = lib
        self.stage_bytes = int(lib.nvh_allreduce_stage_bytes(max_rows, hidden))
        flag_bytes = int(lib.nvh_allreduce_flag_bytes(self.world))
        assert self.stage_bytes > 0 and flag_bytes > 0, "hidden must be a positive multiple of 8"
        self._own, self._opened = [], []
        # Two phases, each closed by a collective that EVERY rank reaches whether its own part worked or not: a rank whose HIP
        # call fails must not leave the others blocked in the handle exchange.
        stage = flags = state = None
        handles, err = None, None
        with torch.cuda.device(self.device):
            try:
                stage, flags, state = (self._alloc(n) for n in (self.stage_bytes, flag_bytes, 64))
                handles = (self._export(stage), self._export(flags))
            except Exception as e:
                err = f"rank {self.rank}: {type(e).__name__}: {e}"
            gathered = [None] * self.world
            dist.all_gather_object(gathered, (err, handles, self._placement()), group=group)   # host-side, once: 2 x 64 bytes per rank
            errs = [g[0] for g in gathered if g[0]]
            stage_ptrs, flag_ptrs = [], []
            if not errs:
                try:
                    self._check_peers([g[2] for g in gathered])                # a kernel touching an unreachable peer would fault
                    for r, (_, (hs, hf), _) in enumerate(gathered):
                        stage_ptrs.append(stage if r == self.rank else self._open(hs))
                        flag_ptrs.append(flags if r == self.rank else self._open(hf))
                except Exception as e:
                    err = f"rank {self.rank}: {type(e).__name__}: {e}"
            gathered2 = [None] * self.world
            dist.all_gather_object(gathered2, err, group=group)
            errs = [g for g in gathered2 if g]
            if errs:
                self.close()
                raise RuntimeError("one-shot all-reduce set-up failed: " + "; ".join(errs))
        self._state = state
        self._stage_tab = torch.tensor(stage_ptrs, dtype=torch.int64, device=self.device)     # void* const [world]
        self._flag_tab = torch.tensor(flag_ptrs, dtype=torch.int64, device=self.device)
        torch.cuda.synchronize(self.device)
        dist.barrier(group=group)                                               # nobody launches before every table is up

    # ---- set-up helpers (host-synchronous HIP calls behind the C ABI)
    def _placement(self):
        import os
        import socket
        vis = tuple(os.environ.get(k, "") for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"))
        return (socket.gethostname(), vis, self.device.index if self.device.index is not None else torch.cuda.current_device())

    def _check_peers(self, placements):
        """Refuse (cleanly, before any kernel runs) a group the peer mappings cannot serve: ranks on another host, or a peer GPU
        this device has no peer access to.  Where the ranks see different device lists the ordinals cannot be compared and the
        check is left to hipIpcOpenMemHandle and the start-up self-test."""
        host, vis, mine = placements[self.rank]
        for r, (h, v, theirs) in enumerate(placements):
            if r == self.rank:
                continue
            if h != host:
                raise RuntimeError(f"rank {r} runs on host {h}, this rank on {host}: IPC mappings need one node")
            if v == vis and theirs != mine and not torch.cuda.can_device_access_peer(mine, theirs):
                raise RuntimeError(f"device {mine} has no peer access to device {theirs} (rank {r})")

    def _alloc(self, nbytes):
        p = ctypes.c_void_p()
        _lib.check(self._lib.nvh_comm_alloc(ctypes.byref(p), nbytes), "nvh_comm_alloc")
        self._own.append(p.value)
        return p.value

    def _export(self, ptr):
        buf = (ctypes.c_ubyte * IPC_HANDLE_BYTES)()
        _lib.check(self._lib.nvh_comm_ipc_export(ptr, buf), "nvh_comm_ipc_export")
        return bytes(buf)

    def _open(self, handle):
        p = ctypes.c_void_p()
        _lib.check(self._lib.nvh_comm_ipc_open(handle, ctypes.byref(p)), "nvh_comm_ipc_open")
        self._opened.append(p.value)
        return p.value

    def close(self):
        """Unmap the peers' buffers and free this rank's (after every rank has stopped calling)."""
        if torch.cuda.is_available():
            torch.cuda.synchronize(self.device)
        for p in self._opened:
            self._lib.nvh_comm_ipc_close(p)
        for p in self._own:
            self._lib.nvh_comm_free(p)
        self._opened, self._own = [], []

    # ---- data path
    def fits(self, x) -> bool:
        return (x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 2 and x.shape[1] == self.hidden and x.shape[0] <= self.max_rows
                and x.stride(1) == 1 and x.stride(0) % 8 == 0)

    def _launch(self, out, x, packed, epilogue):
        rc = self._lib.nvh_allreduce_oneshot(out.data_ptr(), x.data_ptr(), packed.data_ptr() if packed is not None else None,
                                             self._stage_tab.data_ptr(), self._flag_tab.data_ptr(), self._state, self.world, self.rank,
                                             x.shape[0], self.hidden, x.stride(0), out.stride(0), self.stage_bytes, epilogue, NVH_BF16,
                                             torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "nvh_allreduce_oneshot")

    def all_reduce(self, x):
        """In place: x <- sum over ranks (the call of linear.py:188-189)."""
        assert self.fits(x)
        self._launch(x, x, None, AR_EPI_NONE)
        return x

    def all_reduce_residual_add(self, y, residual, packed=None):
        """residual <- bf16(residual + bf16(sum over ranks of y)), and the updated rows in fragment order into `packed` (flat bf16
        buffer of ceil(rows / 16) * 16 * hidden elements) if given: all-reduce + the add of layernorm.py:35-36 in one launch."""
        assert self.fits(y) and residual.shape == y.shape and residual.dtype == torch.bfloat16 and residual.stride(1) == 1
        if packed is not None:
            assert packed.dtype == torch.bfloat16 and packed.is_contiguous() and packed.numel() >= ((y.shape[0] + 15) // 16) * 16 * self.hidden
        self._launch(residual, y, packed, AR_EPI_RESIDUAL_ADD)
        return residual

    def set_spin_limit(self, polls: int):
        """Tests: replace the kernel's poll limit (state[3]; 0 restores the default of ~seconds)."""
        import numpy as np
        v = torch.from_numpy(np.array([polls], dtype=np.uint32).view(np.int32)).to(self.device)
        hip = ctypes.CDLL("libamdhip64.so.7")                                   # already mapped by torch: same runtime instance
        torch.cuda.synchronize(self.device)
        rc = hip.hipMemcpy(ctypes.c_void_p(self._state + 12), ctypes.c_void_p(v.data_ptr()), ctypes.c_size_t(4), ctypes.c_int(3))   # device to device
        assert rc == 0, rc
        torch.cuda.synchronize(self.device)

    def failed_epoch(self) -> int:
        """0, or the call number at which a peer failed to show up (the kernel then wrote NaN rows instead of hanging)."""
        calls, failed = ctypes.c_uint32(0), ctypes.c_uint32(0)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.nvh_allreduce_status(self._state, ctypes.byref(calls), ctypes.byref(failed)), "nvh_allreduce_status")
        return int(failed.value)


_comm = None          # process-wide choice, made once by init_tensor_parallel_comm


def _self_test(comm, group) -> str | None:
    """Run the one-shot kernel on data every rank can predict (rank r contributes a constant pattern scaled by r + 1) at three
    sizes and both epilogues; returns None or what went wrong.  Cheap (six launches) and it turns a mis-set-up IPC mapping or
    an unexpected visibility problem on real hardware into a clean fall-back instead of wrong tokens."""
    world, rank, dev, hid = comm.world, comm.rank, comm.device, comm.hidden
    for rows in (1, 17, comm.max_rows):
        base = (torch.arange(rows * hid, dtype=torch.float32) % 61 - 30).view(rows, hid) / 16     # exactly representable in bf16
        mine = (base * (rank + 1)).to(torch.bfloat16).to(dev)
        acc = torch.zeros_like(base)
        for r in range(world):
            acc = acc + (base * (r + 1)).to(torch.bfloat16).float()
        want = acc.to(torch.bfloat16)
        y = mine.clone()
        comm.all_reduce(y)
        res = torch.ones(rows, hid, dtype=torch.bfloat16, device=dev)
        comm.all_reduce_residual_add(mine, res)
        torch.cuda.synchronize(dev)
        if not torch.equal(y.cpu(), want):
            return f"rank {rank}: all_reduce of {rows} rows differs from the rank-order sum"
        if not torch.equal(res.cpu(), (1.0 + want.float()).to(torch.bfloat16)):
            return f"rank {rank}: fused residual add of {rows} rows differs"
    if comm.failed_epoch():
        return f"rank {rank}: a peer did not show up at call {comm.failed_epoch()}"
    return None


def _time_us(fn, dev, group, iters=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize(dev)
    dist.barrier(group=group)
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    end.record()
    torch.cuda.synchronize(dev)
    return start.elapsed_time(end) * 1e3 / iters


def raise_if_failed(group=None):
    """Call wherever the host synchronises anyway (token readback, end of a generation): if the one-shot all-reduce of ANY rank marked a
    failed call (a peer that never showed up, or one found two calls ahead), EVERY rank raises — the affected rows are NaN and the
    tokens chosen from them are garbage; nothing downstream should see them.  No-op without the one-shot path."""
    comm = _comm
    if comm is None:
        return
    failed = comm.failed_epoch()
    t = torch.tensor([failed], dtype=torch.int64, device=comm.device if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    worst = int(t.item())
    if worst:
        raise RuntimeError(f"one-shot all-reduce failed at call {worst} on some rank (this rank: {failed or 'ok'}): a peer did not arrive in time; "
                           "the residual stream of that step is NaN and its tokens are invalid")


def close_tensor_parallel_comm():
    """Unmap / free the process-wide one-shot communicator (every rank, after the last call)."""
    global _comm
    if _comm is not None:
        _comm.close()
        _comm = None


def init_tensor_parallel_comm(max_rows: int, hidden: int, group=None, prefer_oneshot: bool = True, measure: bool | None = None):
    """Choose the all-reduce path for this process group, the same on every rank and the same on every run: one-shot over IPC if EVERY
    rank managed to set it up AND passed the self-test, RCCL otherwise.  NVH_ALLREDUCE=rccl forces the fallback, =oneshot insists (an error
    if the set-up fails).  The two paths round differently (one-shot: fp32 sum in rank order, one rounding; the RCCL ring rounds at every
    hop), so the choice must not depend on a timing: the comparison with RCCL (`measure`, default only with NVH_ALLREDUCE_MEASURE=1) is
    LOGGED in `last_choice` / `last_measurement`, never acted on.  Returns the OneShotAllReduce or None (= use dist.all_reduce)."""
    import os
    global _comm, last_choice, last_measurement
    close_tensor_parallel_comm()                                                # a second engine in one process: drop the first one's mappings
    want = os.environ.get("NVH_ALLREDUCE", "").lower()
    if measure is None:
        measure = os.environ.get("NVH_ALLREDUCE_MEASURE") == "1"
    if want == "rccl":
        prefer_oneshot = False
    last_measurement = None
    _comm, last_choice = None, "torch.distributed (single rank, no GPU, or one-shot not requested)"
    if not (dist.is_initialized() and dist.get_world_size(group) > 1 and prefer_oneshot and torch.cuda.is_available()):
        return None
    try:
        comm = OneShotAllReduce(max_rows, hidden, group=group)                # raises on EVERY rank if it failed on any (agreed inside)
    except RuntimeError as e:                                                    # e.g. IPC not available between these processes
        if want == "oneshot":
            raise
        import warnings
        warnings.warn(f"{e}; falling back to RCCL (dist.all_reduce)")
        last_choice = f"torch.distributed: {e}"
        return None
    problem = None
    try:
        problem = _self_test(comm, group)
    except Exception as e:
        problem = f"rank {comm.rank}: self-test raised {type(e).__name__}: {e}"
    problems = [None] * comm.world
    dist.all_gather_object(problems, problem, group=group)
    problems = [p for p in problems if p]
    note = "one-shot over IPC-mapped peer buffers (self-test passed)"
    if not problems and measure and dist.get_backend(group) == "nccl":
        # measure, don't guess: the decode message of this model through both paths, on this node, now
        x = torch.zeros(min(32, max_rows), hidden, dtype=torch.bfloat16, device=comm.device)
        t_one = _time_us(lambda: comm.all_reduce(x), comm.device, group)
        t_rccl = _time_us(lambda: dist.all_reduce(x, group=group), comm.device, group)
        t = torch.tensor([t_one, t_rccl], dtype=torch.float64, device=comm.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)                  # the slowest rank's view, identical everywhere
        t_one, t_rccl = float(t[0]), float(t[1])
        last_measurement = {"oneshot_us": round(t_one, 2), "rccl_us": round(t_rccl, 2), "rows": int(x.shape[0]), "hidden": int(hidden)}
        note += f"; measured (logged only): one-shot {t_one:.1f} us vs RCCL {t_rccl:.1f} us per [{x.shape[0]}, {hidden}] bf16 all-reduce, eager launches, max over ranks"
    if problems:
        if want == "oneshot":
            comm.close()
            raise RuntimeError("NVH_ALLREDUCE=oneshot but the one-shot path is not usable: " + "; ".join(problems))
        import warnings
        warnings.warn("one-shot all-reduce not used: " + "; ".join(problems))
        comm.close()
        last_choice = "torch.distributed: " + "; ".join(problems)
        return None
    _comm, last_choice = comm, note
    return _comm


last_choice = "not initialised"
last_measurement = None       # {"oneshot_us", "rccl_us", "rows", "hidden"} when init_tensor_parallel_comm timed both paths (logged, not acted on)


def tensor_parallel_comm():
    return _comm
