"""Device boundary of the engine: KV-cache allocation and binding, the metadata producers that feed the
attention path, eager execution, and a device-resident HIP-graph decode session.

Mirrors nanovllm/engine/model_runner.py for the parts that produce attention inputs:
  allocate_kv_cache      :124-157  one tensor [2, L, num_blocks, block_size, KVH/tp, D]; views bound to every
                                   module that has k_cache / v_cache attributes
  build_block_tables     :160-169  right-pad with -1
  build_prefill_meta     :171-242
  build_decode_meta      :244-269  (positions = len(seq): the reference's own convention)
  run                    :278-313
  DecodeSession          :281-303,316-370 re-thought for MI355X: instead of rebuilding five host tensors and
                                   copying them in before each replay, the decode metadata lives on the device
                                   and is advanced inside the captured graph (context_lens += 1, slot from the
                                   block table), so a decode step is one graph replay and zero host work.
                                   Padding rows carry slot -1 / context 0.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from .. import ops
from ..models.qwen import ModelConfig, QwenForCausalLM, tp_partition
from ..utils.context import reset_context, set_context
from .sequence import Sequence


# ----------------------------------------------------------------------------- metadata producers (host, CPU tensors)
def greedy_tokens(logits):
    """Temperature-0 sampling (sampler.py with bench_my's settings): one HIP launch on the GPU."""
    if logits.is_cuda and logits.dtype == torch.bfloat16 and logits.stride(-1) == 1 and logits.stride(0) % 8 == 0:
        return ops.argmax_rows(logits)
    return logits.argmax(dim=-1)


def build_block_tables(seqs, width=None, pad=-1):
    width = width or max(len(s.block_table) for s in seqs)
    rows = [s.block_table + [pad] * (width - len(s.block_table)) for s in seqs]
    return torch.tensor(rows, dtype=torch.int32)


def _check_block_size(seqs, block_size):
    """Sequence.block_size is a class constant in the reference (engine/sequence.py:15) and feeds num_blocks /
    last_block_num_tokens; a runner with another kvcache_block_size would compute slots from the wrong block."""
    for s in seqs:
        assert s.block_size == block_size, f"Sequence.block_size {s.block_size} != runner block size {block_size}"


def build_prefill_meta(seqs, block_size=256):
    _check_block_size(seqs, block_size)
    input_ids, positions, slot_mapping = [], [], []
    cu_q, cu_k = [0], [0]
    max_q = max_k = 0
    for seq in seqs:
        seqlen = len(seq)
        input_ids.extend(seq[seq.num_cached_tokens:])
        positions.extend(range(seq.num_cached_tokens, seqlen))
        seqlen_q, seqlen_k = seqlen - seq.num_cached_tokens, seqlen
        cu_q.append(cu_q[-1] + seqlen_q)
        cu_k.append(cu_k[-1] + seqlen_k)
        max_q, max_k = max(max_q, seqlen_q), max(max_k, seqlen_k)
        if not seq.block_table:                      # warmup: no cache allocated yet
            continue
        for i in range(seq.num_cached_blocks, seq.num_blocks):
            start = seq.block_table[i] * block_size
            end = start + (block_size if i != seq.num_blocks - 1 else seq.last_block_num_tokens)
            slot_mapping.extend(range(start, end))
    block_tables = build_block_tables(seqs) if cu_k[-1] > cu_q[-1] else None      # prefix-cache hit somewhere
    return dict(input_ids=torch.tensor(input_ids, dtype=torch.int64), positions=torch.tensor(positions, dtype=torch.int64),
                cu_seqlens_q=torch.tensor(cu_q, dtype=torch.int32), cu_seqlens_k=torch.tensor(cu_k, dtype=torch.int32),
                max_seqlen_q=max_q, max_seqlen_k=max_k, slot_mapping=torch.tensor(slot_mapping, dtype=torch.int32),
                block_tables=block_tables)


def build_decode_meta(seqs, block_size=256):
    _check_block_size(seqs, block_size)
    input_ids = [s.last_token for s in seqs]
    positions = [len(s) for s in seqs]
    context_lens = [len(s) for s in seqs]
    slot_mapping = [s.block_table[s.num_blocks - 1] * block_size + s.last_block_num_tokens - 1 for s in seqs]
    return dict(input_ids=torch.tensor(input_ids, dtype=torch.int64), positions=torch.tensor(positions, dtype=torch.int64),
                slot_mapping=torch.tensor(slot_mapping, dtype=torch.int32), context_lens=torch.tensor(context_lens, dtype=torch.int32),
                block_tables=build_block_tables(seqs))


# ----------------------------------------------------------------------------- runner
class ModelRunner:
    def __init__(self, cfg: ModelConfig, num_kvcache_blocks: int, device=None, max_model_len=4096, seed=0):
        self.cfg = cfg
        self.block_size = cfg.kvcache_block_size
        assert self.block_size == Sequence.block_size, (
            f"kvcache_block_size {self.block_size} != Sequence.block_size {Sequence.block_size} (fixed at 256 as in the reference, "
            "engine/sequence.py:15, config.py:20,31)")
        self.max_model_len = max_model_len
        self.rank = dist.get_rank() if dist.is_initialized() else 0
        self.world_size = dist.get_world_size() if dist.is_initialized() else 1
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self.model = QwenForCausalLM(cfg).init_random(seed).to(device=self.device, dtype=torch.bfloat16)
        self.model.eval()
        self.num_kvcache_blocks = num_kvcache_blocks
        self.allocate_kv_cache()
        # tensor parallel: choose the all-reduce path once, the same on every rank (one-shot over IPC-mapped peer buffers for the
        # decode-sized messages, RCCL otherwise / as the fallback); NVH_ALLREDUCE=rccl keeps everything on RCCL
        self.comm = None
        if self.world_size > 1 and self.device.type == "cuda":
            from ..distributed import init_tensor_parallel_comm
            self.comm = init_tensor_parallel_comm(64, cfg.hidden_size, prefer_oneshot=os.environ.get("NVH_ALLREDUCE", "oneshot") != "rccl")

    def allocate_kv_cache(self):
        cfg = self.cfg
        kvh = tp_partition(cfg.num_attention_heads, cfg.num_key_value_heads, self.world_size, self.rank)[3]
        self.kv_cache = torch.zeros(2, cfg.num_hidden_layers, self.num_kvcache_blocks, self.block_size, kvh, cfg.head_dim,
                                    dtype=torch.bfloat16, device=self.device)
        layer_id = 0
        for module in self.model.modules():
            if hasattr(module, "k_cache") and hasattr(module, "v_cache"):         # duck-typed, model_runner.py:151
                module.k_cache = self.kv_cache[0, layer_id]
                module.v_cache = self.kv_cache[1, layer_id]
                layer_id += 1
        assert layer_id == cfg.num_hidden_layers

    def _dev(self, t):
        return t.to(self.device, non_blocking=True) if t is not None else None

    def warmup_model(self, max_num_batched_tokens=16384, max_num_seqs=512):
        """One maximum-size prefill of dummy sequences without cache slots, as the reference does at start-up
        (engine/model_runner.py:107-121): loads every kernel, lets the GEMM library pick its solutions and sizes the
        allocator's pools, so that the first real prefill is not a cold start."""
        num_seqs = max(1, min(max_num_batched_tokens // self.max_model_len, max_num_seqs))
        self.run([Sequence([0] * self.max_model_len) for _ in range(num_seqs)], True)
        torch.cuda.synchronize(self.device)

    @torch.inference_mode()
    def run(self, seqs, is_prefill):
        """One eager forward (prefill, or a decode step with host-built metadata); returns greedy token ids."""
        m = build_prefill_meta(seqs, self.block_size) if is_prefill else build_decode_meta(seqs, self.block_size)
        if is_prefill:
            slots = m["slot_mapping"] if m["slot_mapping"].numel() else None          # warmup: nothing to store
            set_context(True, self._dev(m["cu_seqlens_q"]), self._dev(m["cu_seqlens_k"]), m["max_seqlen_q"], m["max_seqlen_k"],
                        self._dev(slots), None, self._dev(m["block_tables"]))
        else:
            set_context(False, slot_mapping=self._dev(m["slot_mapping"]), context_lens=self._dev(m["context_lens"]),
                        block_tables=self._dev(m["block_tables"]))
        hidden = self.model(self._dev(m["input_ids"]), self._dev(m["positions"]))
        if is_prefill:
            last = self._dev(m["cu_seqlens_q"])[1:].long() - 1                     # last token of every sequence (embed_head.py:62-65)
            hidden = hidden[last]
        tokens = greedy_tokens(self.model.compute_logits(hidden))
        reset_context()
        out = tokens.tolist()                                                         # (the host synchronises here anyway)
        self.raise_if_device_failed()
        return out

    def raise_if_device_failed(self):
        """At a point where the host has synchronised: a time-out marked by one of the bounded device-side waits (the one-shot all-reduce of
        any rank; the one-launch qkv + attention form) must surface as an error on every rank, not as garbage tokens."""
        if self.comm is not None:
            from ..distributed import raise_if_failed
            raise_if_failed()
        from ..models import qwen
        if qwen.QKV_ATTEND_MODE in ("one_launch", "auto") and self.device.type == "cuda" and ops.qkv_rope_attend_status(device=self.device):
            raise RuntimeError("nvh_qkv_rope_attend (one launch): a consumer's wait for the projection's producers ran out; the step's rows are NaN")

    def close(self):
        """Release the tensor-parallel communicator's IPC mappings (every rank, after the last step)."""
        if self.comm is not None:
            from ..distributed import close_tensor_parallel_comm
            close_tensor_parallel_comm()
            self.comm = None

    def decode_session(self, seqs, max_new_tokens, use_graph=True):
        return DecodeSession(self, seqs, max_new_tokens, use_graph)


def _force_end_capture(stream):
    """Best effort: hipStreamEndCapture on `stream` through the HIP runtime torch already loaded.  Measured on ROCm 7.2 with
    an uncapturable (gloo) collective: the runtime answers 908 and the stream stays invalidated, so this does not rescue that
    case — which is why DecodeSession does not attempt capture with non-RCCL backends in the first place."""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so.7")                 # already mapped by torch: same runtime instance
        g = ctypes.c_void_p()
        hip.hipStreamEndCapture(ctypes.c_void_p(stream.cuda_stream), ctypes.byref(g))
        if g.value:
            hip.hipGraphDestroy(g)
        hip.hipGetLastError()
    except (OSError, AttributeError):
        pass
    try:
        torch.cuda.synchronize()
    except Exception:
        pass


class DecodeSession:
    """Batched greedy decode of `seqs` for a fixed number of steps with device-resident metadata.

    All blocks the sequences will need are already in their block tables (BlockManager.allocate(reserve_tokens=...)),
    so block tables are static; per step the device advances input_ids / positions / context_lens / slot_mapping
    itself.  The step is captured once into a HIP graph (torch.cuda.CUDAGraph on ROCm) and replayed."""

    # steps captured in the second graph: step(n) replays it while n allows and the one-step graph for the rest.  Between two replays the
    # GPU idles for several microseconds (graph launch: 8.8 us under rocprofv3 between the last kernel of one replay and the first of the
    # next, against ~1.3 us between kernels inside one), which a graph of MULTI steps pays once per MULTI tokens
    MULTI = 4

    def __init__(self, runner: ModelRunner, seqs, max_new_tokens, use_graph=True):
        self.runner, self.seqs, self.use_graph = runner, seqs, use_graph
        dev, bs = runner.device, runner.block_size
        b = len(seqs)
        self.batch = b
        need_tokens = max(len(s) for s in seqs) + max_new_tokens
        width = (need_tokens + bs - 1) // bs
        for s in seqs:
            assert len(s.block_table) * bs >= len(s) + max_new_tokens, "reserve blocks for the whole generation first"
        m = build_decode_meta(seqs, bs)
        self.block_tables = build_block_tables(seqs, width=max(width, max(len(s.block_table) for s in seqs)), pad=0).to(dev)
        self.input_ids = m["input_ids"].to(dev)
        self.positions = m["positions"].to(dev)
        self.context_lens = m["context_lens"].to(dev)
        self.slot_mapping = m["slot_mapping"].to(dev)
        self.tokens = torch.zeros(max_new_tokens + 1, b, dtype=torch.int64, device=dev)
        self.step_idx = torch.zeros((), dtype=torch.int64, device=dev)
        self.row_steps = torch.zeros(b, dtype=torch.int64, device=dev)          # tokens generated per row (device-side log index)
        self.max_new_tokens = max_new_tokens
        self.steps_done = 0
        cfg = runner.cfg
        h = tp_partition(cfg.num_attention_heads, cfg.num_key_value_heads, runner.world_size, runner.rank)[1]
        ops.reserve_workspace(dev, ops.decode_workspace_bytes(b, h, cfg.head_dim, self.block_tables.shape[1], bs))
        # the next step's embedding lookup rides in the arg-max launch when the model allows it: the session then owns the
        # embedded rows (row-major = the residual stream of the step, and in fragment order for layer 0's projection)
        self.fuse_embed = bool(getattr(runner.model, "fused_embedding_ok", lambda *_: False)(b, torch.device(dev)))
        if self.fuse_embed:
            hid = cfg.hidden_size
            self.hidden_in = torch.zeros(b, hid, dtype=torch.bfloat16, device=dev)
            self.hidden_in_p = torch.zeros(((b + 15) // 16) * 16 * hid, dtype=torch.bfloat16, device=dev)
            self._refresh_embedding()
        self.graph = None
        self.graph_multi = None
        # Graph or eager is decided BEFORE any capture attempt.  Capturable: a single rank; several ranks whose step contains only
        # this library's own launches (the one-shot all-reduce) or RCCL collectives.  Anything else (e.g. a gloo rehearsal without
        # the one-shot path) runs eager steps.  A capture that fails all the same is FATAL: ROCm leaves the stream invalidated
        # (hipStreamEndCapture answers 908, every later call on it fails), and with a collective inside the other ranks would be
        # left on poisoned streams too — the process reports why and stops; a supervisor may start a fresh one.
        if use_graph and runner.world_size > 1 and runner.comm is None and dist.get_backend() != "nccl":
            use_graph = False
        if use_graph:
            try:
                self._capture()
            except Exception as e:
                raise RuntimeError(
                    f"HIP-graph capture of the decode step failed ({type(e).__name__}: {e}).  The capturing stream is invalidated and "
                    "cannot be used again in this process: restart with enforce_eager=True (bench.py --eager) or fix the uncapturable "
                    "call.  Not continuing on a poisoned stream.") from e

    def _advance(self, next_tokens):
        """Device-side postprocess + prepare_decode for the following step."""
        bs = self.runner.block_size
        self.tokens.index_copy_(0, self.step_idx.view(1), next_tokens.view(1, -1))    # capturable (no host read of step_idx)
        self.step_idx += 1
        self.input_ids.copy_(next_tokens)
        self.positions += 1
        self.context_lens += 1
        last = (self.context_lens - 1).long()
        blk = torch.gather(self.block_tables, 1, (last // bs).unsqueeze(1)).squeeze(1)
        self.slot_mapping.copy_((blk.long() * bs + last % bs).int())

    @torch.inference_mode()
    def _refresh_embedding(self):
        """(Re)build the embedded rows from input_ids: at the start and whenever the metadata is set from outside."""
        self.hidden_in.copy_(self.runner.model.embed_tokens(self.input_ids))
        packed = ops.pack_rows(self.hidden_in)
        self.hidden_in_p[: packed.numel()].copy_(packed)

    def _step(self):
        set_context(False, slot_mapping=self.slot_mapping, context_lens=self.context_lens, block_tables=self.block_tables)
        model = self.runner.model
        if self.fuse_embed:
            hidden = model(self.input_ids, self.positions, embedded=(self.hidden_in, self.hidden_in_p))
        else:
            hidden = model(self.input_ids, self.positions)
        cand = model.greedy_candidates(hidden)                 # LM head + arg-max candidates in one launch (fused decode path)
        if cand is not None:
            embed = (model.embed_tokens.weight, self.hidden_in, self.hidden_in_p) if self.fuse_embed else None
            ops.greedy_advance_candidates(cand[0], cand[1], cand[2], self.batch, self.input_ids, self.positions, self.context_lens,
                                          self.slot_mapping, self.block_tables, self.runner.block_size, self.tokens, self.row_steps, embed=embed)
            reset_context()
            return
        assert not self.fuse_embed, "the fused embedding needs the candidates path"
        logits = self.runner.model.compute_logits(hidden)
        if logits.is_cuda and logits.dtype == torch.bfloat16 and logits.stride(0) % 8 == 0:
            # sampling + postprocess + next step's prepare_decode in ONE launch (nvh_greedy_advance)
            ops.greedy_advance(logits, self.input_ids, self.positions, self.context_lens, self.slot_mapping, self.block_tables,
                               self.runner.block_size, self.tokens, self.row_steps)
        else:
            self._advance(greedy_tokens(logits))
        reset_context()

    def _live(self):
        return (self.input_ids, self.positions, self.context_lens, self.slot_mapping, self.tokens, self.step_idx, self.row_steps)

    def _restore(self, saved):
        for t, s in zip(self._live(), saved):
            t.copy_(s)
        if self.fuse_embed:
            self._refresh_embedding()

    @torch.inference_mode()
    def _capture(self):
        # capture must not disturb the live state: snapshot, warm up + capture, restore
        self._capture_saved = saved = [t.clone() for t in self._live()]
        stream = torch.cuda.Stream(device=self.runner.device)
        stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(stream):
            self._step()                                      # warm-up on the side stream (allocator, hipBLASLt heuristics)
        torch.cuda.current_stream().wait_stream(stream)
        torch.cuda.synchronize()
        self.step_idx.zero_()
        self.row_steps.zero_()
        graph = torch.cuda.CUDAGraph()
        cap_stream = torch.cuda.Stream(device=self.runner.device)
        try:
            with torch.cuda.graph(graph, stream=cap_stream):
                self._step()
        except Exception:
            # a failed capture can leave the stream in the invalidated-capture state, in which every later call fails
            # ("operation failed due to a previous error during capture"): end it by hand, drop the sticky error, then re-raise
            _force_end_capture(cap_stream)
            raise
        self.graph = graph
        self.graph_multi = None
        if self.MULTI > 1 and self.max_new_tokens >= self.MULTI:
            gm = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(gm, stream=cap_stream):
                    for _ in range(self.MULTI):               # the same launches on the same buffers: the device-resident metadata carries the state
                        self._step()
            except Exception:
                _force_end_capture(cap_stream)
                raise
            self.graph_multi = gm
        self._restore(saved)
        # the KV rows the warm-up/capture steps wrote lie beyond the live context and are overwritten by real steps

    @torch.inference_mode()
    def step(self, n=1):
        assert self.steps_done + n <= self.max_new_tokens
        left = n
        while left > 0:
            if self.graph is not None and getattr(self, "graph_multi", None) is not None and left >= self.MULTI:
                self.graph_multi.replay()
                left -= self.MULTI
                continue
            if self.graph is not None:
                self.graph.replay()
            else:
                self._step()
            left -= 1
        self.steps_done += n

    def rewind(self, seqs_state):
        """Reset the device metadata to a saved state (benchmark use: time the same context window repeatedly)."""
        for t, s in zip((self.input_ids, self.positions, self.context_lens, self.slot_mapping, self.step_idx, self.row_steps), seqs_state):
            t.copy_(s)
        if self.fuse_embed:
            self._refresh_embedding()
        self.steps_done = int(self.row_steps.max().item())

    def state(self):
        return [t.clone() for t in (self.input_ids, self.positions, self.context_lens, self.slot_mapping, self.step_idx, self.row_steps)]

    def finish(self):
        """Copy generated tokens back into the host sequences (one device->host sync for the whole generation)."""
        toks = self.tokens[: self.steps_done].cpu().tolist()
        self.runner.raise_if_device_failed()                                          # (behind the sync of the readback)
        for row in toks:
            for s, t in zip(self.seqs, row):
                s.append_token(int(t))
        return toks
