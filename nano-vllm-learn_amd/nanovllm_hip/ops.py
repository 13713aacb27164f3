"""Tensor-level wrappers over the C ABI (include/nvh_attn.h).

Same function names and argument meaning as the helpers the reference's attention modules call
(store_kvcache: nanovllm/layers/attention.py:44-55; flash_attn_with_kvcache / flash_attn_varlen_func:
call sites attention.py:93-101, oracle bodies nanovllm/layers/attention_sdpa.py:65-182), so parity
tests read like the reference.  Everything runs on torch's CURRENT stream and is graph-capture safe:
no host synchronisation, no host reads of device data.  There is no fallback: tensors must live on
a GPU and the HIP library must load.
"""
from __future__ import annotations

import ctypes

import torch

from . import _lib
from ._lib import NVH_BF16, NVH_F32

_workspaces: dict[tuple, torch.Tensor] = {}
_retired_workspaces: list[torch.Tensor] = []


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _out_code(dtype):
    if dtype == torch.bfloat16:
        return NVH_BF16
    if dtype == torch.float32:
        return NVH_F32
    raise TypeError(f"output dtype {dtype} unsupported (bfloat16 or float32)")


def _require_gpu_bf16(**tensors):
    for name, t in tensors.items():
        if not t.is_cuda:
            raise RuntimeError(f"{name} must be a GPU tensor: the hip attention backend has no CPU path")
        if t.dtype != torch.bfloat16:
            raise TypeError(f"{name} must be bfloat16 (the reference's runtime dtype), got {t.dtype}")


def _require_i32(**tensors):
    for name, t in tensors.items():
        if not t.is_cuda or t.dtype != torch.int32:
            raise TypeError(f"{name} must be an int32 GPU tensor, got {t.dtype} on {t.device}")


# --------------------------------------------------------------------------------------- store
def store_kvcache(key: torch.Tensor, value: torch.Tensor, k_cache: torch.Tensor, v_cache: torch.Tensor,
                  slot_mapping: torch.Tensor) -> None:
    """cache.view(-1, KVH*D)[slot_mapping[i]] = key[i] / value[i]; rows with slot < 0 are skipped.

    key/value [N, KVH, D] with contiguous inner dims and any row stride (attention.py:49-55)."""
    n, kvh, hd = key.shape
    d = kvh * hd
    _require_gpu_bf16(key=key, value=value, k_cache=k_cache, v_cache=v_cache)
    _require_i32(slot_mapping=slot_mapping)
    assert value.shape == key.shape
    assert key.stride(-1) == 1 and value.stride(-1) == 1                 # attention.py:51
    assert key.stride(1) == hd and value.stride(1) == hd                 # attention.py:52
    assert k_cache.stride(1) == d and v_cache.stride(1) == d             # attention.py:53
    assert k_cache.is_contiguous() and v_cache.is_contiguous()
    assert slot_mapping.numel() == n and slot_mapping.is_contiguous()    # attention.py:54
    rc = _lib.load().nvh_store_kvcache(key.data_ptr(), value.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(),
                                       slot_mapping.data_ptr(), n, kvh, hd, key.stride(0), value.stride(0),
                                       NVH_BF16, _stream())
    _lib.check(rc, "nvh_store_kvcache")


# --------------------------------------------------------------------------------------- decode
def decode_workspace_bytes(batch, num_heads, head_dim, max_blocks, block_size) -> int:
    return int(_lib.load().nvh_paged_decode_workspace(batch, num_heads, head_dim, max_blocks, block_size))


def reserve_workspace(device, nbytes: int, kind: str = "chunked") -> torch.Tensor:
    """Grow (never shrink) the per-device decode scratch.  Call before graph capture with the largest shape; all layers
    share it (they run back to back on ONE stream — the reference runs one process per GPU and one stream per process;
    callers that launch decode attention on several streams concurrently must pass their own `workspace=` per stream).
    ZERO-FILLED once (the contract of nvh_attn.h): the buffer starts with a fixed header of arrival tickets (one uint32 per
    (sequence, kv head) pair, each on a 128-byte line of its own) and the counters of the fused qkv + attention launch; the chunk
    records behind it are plain fp32 rows published with write-through stores and need no initial value.  Every launch returns the
    tickets and counters it used to zero, so one buffer serves all layers, shapes and graph replays; ONE launch at a time per
    workspace; after a launch that was aborted mid-flight (a device fault, a killed process) zero-fill it again.  The split +
    combine variants (tests, A/B) keep their partials in a buffer of their own (`kind`)."""
    idx = torch.device(device).index
    idx = torch.cuda.current_device() if idx is None else idx
    idx = (idx, kind)
    ws = _workspaces.get(idx)
    if ws is None or ws.numel() < nbytes:
        if ws is not None and torch.cuda.is_current_stream_capturing():
            return torch.zeros(nbytes, dtype=torch.uint8, device=device)   # graph-pool memory, not cached
        if ws is not None:
            _retired_workspaces.append(ws)                 # a captured graph may still point at it: never freed
        ws = torch.zeros(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        if not torch.cuda.is_current_stream_capturing():
            _workspaces[idx] = ws
    return ws


def _decode_common(q, k_cache, v_cache, cache_seqlens, block_table, softmax_scale, out, out_dtype, workspace=None, kind="chunked"):
    squeeze = q.dim() == 4
    if squeeze:
        assert q.shape[1] == 1, "Decode stage should have seq_len=1"      # attention_sdpa.py:133
        q3 = q[:, 0]
    else:
        q3 = q
    b, h, hd = q3.shape
    nb, bs, kvh, hd2 = k_cache.shape
    assert hd2 == hd and v_cache.shape == k_cache.shape
    _require_gpu_bf16(q=q, k_cache=k_cache, v_cache=v_cache)
    _require_i32(cache_seqlens=cache_seqlens, block_table=block_table)
    assert q3.stride(-1) == 1 and q3.stride(1) == hd
    assert k_cache.is_contiguous() and v_cache.is_contiguous()
    assert block_table.dim() == 2 and block_table.shape[0] == b and block_table.stride(1) == 1
    assert cache_seqlens.numel() == b and cache_seqlens.is_contiguous()
    if softmax_scale is None:
        softmax_scale = hd ** -0.5
    out_dtype = out_dtype or torch.bfloat16
    if out is None:
        out = torch.empty((b, h, hd), dtype=out_dtype, device=q.device)
    else:
        assert out.shape == (b, h, hd) and out.is_contiguous() and out.dtype == out_dtype
    max_blocks = block_table.shape[1]
    need = decode_workspace_bytes(b, h, hd, max_blocks, bs)
    if workspace is not None:
        assert workspace.is_cuda and workspace.dtype == torch.uint8 and workspace.is_contiguous() and workspace.numel() >= need
        ws = workspace
    else:
        ws = reserve_workspace(q.device, need, kind)
    return squeeze, q3, b, h, hd, kvh, bs, max_blocks, float(softmax_scale), out, ws


def flash_attn_with_kvcache(q, k_cache, v_cache, cache_seqlens, block_table, softmax_scale=None, causal=True,
                            out=None, out_dtype=None, out_packed=None, variant=None, waves=0, chunks=0, workspace=None):
    """Decode attention, drop-in for the call at attention.py:99-101.

    q [B, 1, H, D] (or [B, H, D]); caches [NB, bs, KVH, D]; cache_seqlens int32 [B] (0 -> zero row);
    block_table int32 [B, max_blocks].  `causal` is accepted for signature parity; with one query per
    sequence it has no effect.  Returns [B, 1, H, D] (or [B, H, D]).  out_packed: optional flat bf16 buffer of
    ceil(B/16)*16*H*D elements that also receives the result in MFMA-fragment order (pack_rows layout) for fused_linear.
    variant ("chunked" | "split_mfma" | "split_valu"), waves, chunks: tests / A-B only (nvh_paged_decode_variant); the module
    path never passes them.  workspace: a caller-owned ZERO-FILLED uint8 buffer of decode_workspace_bytes() (one per stream
    that may run decode attention concurrently); default: the per-device buffer of reserve_workspace()."""
    squeeze, q3, b, h, hd, kvh, bs, max_blocks, scale, out, ws = _decode_common(
        q, k_cache, v_cache, cache_seqlens, block_table, softmax_scale, out, out_dtype, workspace,
        "split" if variant in ("split_mfma", "split_valu") else "chunked")
    tail = (q3.data_ptr(), k_cache.data_ptr(), v_cache.data_ptr(), block_table.data_ptr(), cache_seqlens.data_ptr(), b, h, kvh, hd, bs,
            max_blocks, q3.stride(0), block_table.stride(0), scale, NVH_BF16, _out_code(out.dtype), ws.data_ptr(), ws.numel(), _stream())
    if variant is not None or waves or chunks:
        assert out_packed is None
        rc = _lib.load().nvh_paged_decode_variant(_lib.DECODE_VARIANTS[variant or "chunked"], int(waves), int(chunks), out.data_ptr(), *tail)
    elif out_packed is not None:
        _require_gpu_bf16(out_packed=out_packed)
        assert out_packed.is_contiguous() and out_packed.numel() >= ((b + 15) // 16) * 16 * h * hd
        rc = _lib.load().nvh_paged_decode_packed(out.data_ptr(), out_packed.data_ptr(), *tail)
    else:
        rc = _lib.load().nvh_paged_decode(out.data_ptr(), *tail)
    _lib.check(rc, "nvh_paged_decode")
    return out.unsqueeze(1) if squeeze else out


def decode_step(q, k_new, v_new, k_cache, v_cache, slot_mapping, cache_seqlens, block_table, softmax_scale=None,
                out=None, out_dtype=None):
    """store_kvcache(k_new, v_new, ...) followed by flash_attn_with_kvcache(q, ...) as one C-ABI call
    (the two calls at attention.py:84-86 and :99-101)."""
    squeeze, q3, b, h, hd, kvh, bs, max_blocks, scale, out, ws = _decode_common(
        q, k_cache, v_cache, cache_seqlens, block_table, softmax_scale, out, out_dtype)
    _require_gpu_bf16(k_new=k_new, v_new=v_new)
    _require_i32(slot_mapping=slot_mapping)
    assert k_new.shape == (b, kvh, hd) and v_new.shape == (b, kvh, hd)
    assert k_new.stride(-1) == 1 and v_new.stride(-1) == 1 and k_new.stride(1) == hd and v_new.stride(1) == hd
    assert slot_mapping.numel() == b and slot_mapping.is_contiguous()
    rc = _lib.load().nvh_decode_step(out.data_ptr(), q3.data_ptr(), k_new.data_ptr(), v_new.data_ptr(),
                                     k_cache.data_ptr(), v_cache.data_ptr(), slot_mapping.data_ptr(),
                                     block_table.data_ptr(), cache_seqlens.data_ptr(), b, h, kvh, hd, bs, max_blocks,
                                     q3.stride(0), k_new.stride(0), v_new.stride(0), block_table.stride(0), scale,
                                     NVH_BF16, _out_code(out.dtype), ws.data_ptr(), ws.numel(), _stream())
    _lib.check(rc, "nvh_decode_step")
    return out.unsqueeze(1) if squeeze else out


# --------------------------------------------------------------------------------------- prefill
PV16_MIN_KEYS = 512       # fp16 P V (nvh_prefill_varlen_pv16) by default from this many keys per sequence on (40.5 vs 42.0 us at 32 x 512; below, the conversion launch costs what it saves)


def flash_attn_varlen_func(q, k, v, max_seqlen_q, cu_seqlens_q, max_seqlen_k, cu_seqlens_k, softmax_scale=None,
                           causal=True, block_table=None, out_dtype=None, kernel=None, short_waves=0, pv_fp16=None):
    """Packed varlen causal attention, drop-in for the call at attention.py:93-96.

    q [Tq, H, D]; without block_table k/v are [Tk, KVH, D] (any row stride); with block_table they are
    the paged caches [NB, bs, KVH, D] and sequence i reads its keys through block_table[i].
    kernel ("auto" | "tiled" | "short" | "tiled_f16v"), short_waves: tests / A-B only (nvh_prefill_varlen_variant).
    pv_fp16: P V on the fp16 matrix pipe (nvh_prefill_varlen_pv16: V converted to fp16 rows in a scratch buffer with a range guard, P rounded to
    11 bits; a V value that does not fit fp16 makes the kernel itself fall back to the exact form, no host read).  None (default) = where it pays and
    is invisible: bf16 output (whose own rounding, 2^-9 |o|, is 8x coarser than P's), no block_table, max_seqlen_k >= PV16_MIN_KEYS or a batch the
    short-sequence kernel takes (head_dim 64, <= 128 keys, >= 128 (sequence, kv head) pairs: it converts V inside the kernel); True = always
    (no block_table); False = never (P as bf16 hi + lo, 6e-6).  Error of the fp16 form <= 2^-12 * max|v|: 4.5e-4 on the reference goldens (the
    reference's flash backend rounds P to a single bf16, 8 bits).  DESIGN.md section 12.2."""
    if kernel is not None or short_waves or block_table is not None:
        assert not pv_fp16 or kernel == "tiled_f16v", "pv_fp16 goes with the default kernel choice and packed k / v rows"
        pv_fp16 = False
    elif pv_fp16 is None:
        # ... or on the short-sequence kernel's shapes (it converts its resident V images itself: no extra launch; the library's rule, prefill_mfma.hip launch_short)
        short = (q.shape[2] == 64 and 64 < int(max_seqlen_k) <= 128 and int(max_seqlen_q) <= int(max_seqlen_k)
                 and (cu_seqlens_q.numel() - 1) * k.shape[1] >= 128)
        pv_fp16 = (out_dtype in (None, torch.bfloat16)) and (int(max_seqlen_k) >= PV16_MIN_KEYS or short)
    if not causal:
        raise NotImplementedError("the reference only ever calls this with causal=True (attention.py:96)")
    tq, h, hd = q.shape
    if kernel == "tiled_f16v":                                  # measurement variant: v already converted to fp16 by the caller
        assert v.dtype == torch.float16 and v.is_cuda and block_table is None
        _require_gpu_bf16(q=q, k=k)
    else:
        _require_gpu_bf16(q=q, k=k, v=v)
    _require_i32(cu_seqlens_q=cu_seqlens_q, cu_seqlens_k=cu_seqlens_k)
    assert q.stride(-1) == 1 and q.stride(1) == hd
    batch = cu_seqlens_q.numel() - 1
    assert cu_seqlens_k.numel() == batch + 1
    if softmax_scale is None:
        softmax_scale = hd ** -0.5
    out = torch.empty((tq, h, hd), dtype=out_dtype or torch.bfloat16, device=q.device)
    if block_table is not None:
        _require_i32(block_table=block_table)
        nb, bs, kvh, hd2 = k.shape
        assert hd2 == hd and v.shape == k.shape and k.is_contiguous() and v.is_contiguous()
        assert block_table.shape[0] == batch and block_table.stride(1) == 1
        max_blocks, bt_stride, bt_ptr = block_table.shape[1], block_table.stride(0), block_table.data_ptr()
        k_stride = v_stride = kvh * hd
    else:
        tk, kvh, hd2 = k.shape
        assert hd2 == hd and v.shape == k.shape
        assert k.stride(-1) == 1 and v.stride(-1) == 1 and k.stride(1) == hd and v.stride(1) == hd
        bs, max_blocks, bt_stride, bt_ptr = 0, 0, 0, None
        k_stride, v_stride = k.stride(0), v.stride(0)
    tail = (out.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), cu_seqlens_q.data_ptr(), cu_seqlens_k.data_ptr(), bt_ptr, batch,
            int(max_seqlen_q), int(max_seqlen_k), h, kvh, hd, bs, max_blocks, q.stride(0), k_stride, v_stride, bt_stride,
            float(softmax_scale), NVH_BF16, _out_code(out.dtype), _stream())
    if pv_fp16:
        lib = _lib.load()
        if int(max_seqlen_k) <= 128 and not lib.nvh_prefill_pv16_uses_scratch(batch, int(max_seqlen_q), int(max_seqlen_k), kvh, hd):
            sc_ptr, sc_bytes = None, 0                           # the short-sequence kernel converts V in LDS: one launch, no scratch
        else:
            scratch = torch.empty(lib.nvh_prefill_pv16_scratch_bytes(tk, kvh, hd), dtype=torch.uint8, device=q.device)
            sc_ptr, sc_bytes = scratch.data_ptr(), scratch.numel()
        rc = lib.nvh_prefill_varlen_pv16(out.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), cu_seqlens_q.data_ptr(), cu_seqlens_k.data_ptr(), batch,
                                         int(max_seqlen_q), int(max_seqlen_k), tk, h, kvh, hd, q.stride(0), k_stride, v_stride, float(softmax_scale),
                                         NVH_BF16, _out_code(out.dtype), sc_ptr, sc_bytes, _stream())
        _lib.check(rc, "nvh_prefill_varlen_pv16")
        return out
    if kernel is not None or short_waves:
        rc = _lib.load().nvh_prefill_varlen_variant(_lib.PREFILL_KERNELS[kernel or "auto"], int(short_waves), *tail)
    else:
        rc = _lib.load().nvh_prefill_varlen(*tail)
    _lib.check(rc, "nvh_prefill_varlen")
    return out


# --------------------------------------------------------------------------------------- rope + store (next row f2)
def rope_store(qkv, positions, cos_sin, num_heads, num_kv_heads, head_dim, k_cache=None, v_cache=None, slot_mapping=None,
               q_norm_weight=None, k_norm_weight=None, eps=1e-6):
    """In-place (optional per-head RMSNorm ->) neox RoPE on the q and k parts of the fused projection output
    `qkv` [N, (H+2KVH)*D], then k/v rows -> paged cache at slot_mapping (skipped when no cache is bound or slot < 0).
    Equivalent to q_norm/k_norm + rotary_emb + store_kvcache of the reference (qwen3.py:108-116, attention.py:84-86)."""
    _require_gpu_bf16(qkv=qkv)
    n = qkv.shape[0]
    assert qkv.dim() == 2 and qkv.stride(1) == 1 and qkv.shape[1] == (num_heads + 2 * num_kv_heads) * head_dim
    assert positions.dtype == torch.int64 and positions.is_cuda and positions.numel() == n and positions.is_contiguous()
    assert cos_sin.dtype == torch.float32 and cos_sin.is_cuda and cos_sin.is_contiguous() and cos_sin.shape[1] == head_dim
    have_cache = k_cache is not None and k_cache.numel() > 0 and slot_mapping is not None
    if have_cache:
        _require_gpu_bf16(k_cache=k_cache, v_cache=v_cache)
        _require_i32(slot_mapping=slot_mapping)
        assert k_cache.is_contiguous() and v_cache.is_contiguous() and k_cache.stride(1) == num_kv_heads * head_dim
        assert slot_mapping.numel() == n and slot_mapping.is_contiguous()
    if q_norm_weight is not None:
        _require_gpu_bf16(q_norm_weight=q_norm_weight, k_norm_weight=k_norm_weight)
    rc = _lib.load().nvh_rope_store(qkv.data_ptr(), positions.data_ptr(), cos_sin.data_ptr(),
                                    q_norm_weight.data_ptr() if q_norm_weight is not None else None,
                                    k_norm_weight.data_ptr() if k_norm_weight is not None else None, float(eps),
                                    k_cache.data_ptr() if have_cache else None, v_cache.data_ptr() if have_cache else None,
                                    slot_mapping.data_ptr() if have_cache else None,
                                    n, num_heads, num_kv_heads, head_dim, qkv.stride(0), NVH_BF16, _stream())
    _lib.check(rc, "nvh_rope_store")


# --------------------------------------------------------------------------------------- row-wise ops around the attention block
def add_rmsnorm(x, weight, eps, residual=None):
    """RMSNorm.forward(x[, residual]) of the reference (layernorm.py:43-50) in one launch.  `residual`, when given, is
    updated IN PLACE to x + residual (bf16) and the norm is taken of the fp32 sum.  Returns the normalised tensor."""
    _require_gpu_bf16(x=x, weight=weight)
    hidden = x.shape[-1]
    x2 = x.view(-1, hidden)
    assert x2.stride(1) == 1 and weight.numel() == hidden and weight.is_contiguous()
    out = torch.empty((x2.shape[0], hidden), dtype=torch.bfloat16, device=x.device)
    r2 = None
    if residual is not None:
        _require_gpu_bf16(residual=residual)
        r2 = residual.view(-1, hidden)
        assert r2.shape == x2.shape and r2.stride(1) == 1
    rc = _lib.load().nvh_add_rmsnorm(out.data_ptr(), x2.data_ptr(), r2.data_ptr() if r2 is not None else None, weight.data_ptr(), float(eps),
                                     x2.shape[0], hidden, x2.stride(0), out.stride(0), r2.stride(0) if r2 is not None else 0, NVH_BF16, _stream())
    _lib.check(rc, "nvh_add_rmsnorm")
    return out.view(x.shape)


def residual_add_pack(residual, y, packed=None):
    """residual += y in place (bf16, one rounding) and, optionally, the updated rows in fragment order into `packed`
    (nvh_residual_add_pack): the step after a tensor-parallel all-reduce in the fused decode layer."""
    _require_gpu_bf16(residual=residual, y=y)
    assert residual.dim() == 2 and residual.shape == y.shape and residual.stride(1) == 1 and y.stride(1) == 1
    m, hidden = residual.shape
    if packed is not None:
        _require_gpu_bf16(packed=packed)
        assert packed.is_contiguous() and packed.numel() >= ((m + 15) // 16) * 16 * hidden
    rc = _lib.load().nvh_residual_add_pack(residual.data_ptr(), y.data_ptr(), packed.data_ptr() if packed is not None else None, m, hidden,
                                           residual.stride(0), y.stride(0), NVH_BF16, _stream())
    _lib.check(rc, "nvh_residual_add_pack")
    return residual


def silu_mul(gate_up):
    """SiluAndMul.forward (activation.py:11-14): silu(gate_up[..., :I]) * gate_up[..., I:]."""
    _require_gpu_bf16(gate_up=gate_up)
    inter = gate_up.shape[-1] // 2
    g2 = gate_up.view(-1, 2 * inter)
    assert g2.stride(1) == 1
    out = torch.empty((g2.shape[0], inter), dtype=torch.bfloat16, device=gate_up.device)
    rc = _lib.load().nvh_silu_mul(out.data_ptr(), g2.data_ptr(), g2.shape[0], inter, g2.stride(0), out.stride(0), NVH_BF16, _stream())
    _lib.check(rc, "nvh_silu_mul")
    return out.view(*gate_up.shape[:-1], inter)


LINEAR_SMALL_M_MAX = 64


def linear_small_m(x, weight, bias=None, silu_mul=False):
    """F.linear(x, weight, bias) for at most 64 rows, streaming `weight` [N, K] once; with silu_mul=True `weight` is the
    merged gate_up projection and the result is SiLU(x gate^T) * (x up^T) (projection + activation.py:11-14 in one launch)."""
    _require_gpu_bf16(x=x, weight=weight)
    k = x.shape[-1]
    x2 = x.view(-1, k)
    n = weight.shape[0]
    assert x2.shape[0] <= LINEAR_SMALL_M_MAX and x2.stride(1) == 1 and weight.is_contiguous() and weight.shape[1] == k
    inter = n // 2 if silu_mul else 0
    out = torch.empty((x2.shape[0], inter if silu_mul else n), dtype=torch.bfloat16, device=x.device)
    if bias is not None:
        _require_gpu_bf16(bias=bias)
        assert bias.numel() == n and bias.is_contiguous()
    rc = _lib.load().nvh_linear_small_m(out.data_ptr(), x2.data_ptr(), weight.data_ptr(), bias.data_ptr() if bias is not None else None,
                                        x2.shape[0], n, k, inter, x2.stride(0), out.stride(0), NVH_BF16, _stream())
    _lib.check(rc, "nvh_linear_small_m")
    return out.view(*x.shape[:-1], out.shape[1])


def pack_rows(x, rows=None):
    """[M, C] bf16 -> the MFMA-fragment order the streaming GEMM reads at full address rate (nvh_pack_index):
    [ceil(M/16)][C/32][lane = 16 * (col/8 % 4) + row % 16][8], rows padded with zeros.  Torch-side helper (tests, first layer)."""
    m, c = x.shape
    assert c % 32 == 0
    mt = (max(m, rows or m) + 15) // 16
    xp = torch.zeros((mt * 16, c), dtype=x.dtype, device=x.device)
    xp[:m] = x
    return xp.view(mt, 16, c // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(-1)


def unpack_rows(xp, m, c):
    """Inverse of pack_rows: flat fragment-order buffer -> [m, c]."""
    mt = xp.numel() // (16 * c)
    return xp.view(mt, c // 32, 4, 16, 8).permute(0, 3, 1, 2, 4).reshape(mt * 16, c)[:m]


def linear_workspace_bytes(m, n, k, epilogue="none") -> int:
    return int(_lib.load().nvh_linear_small_m_workspace(m, n, k, _EPI_CODES[epilogue]))


_EPI_CODES = {"none": _lib.EPI_NONE, "silu_mul": _lib.EPI_SILU_MUL, "residual_add": _lib.EPI_RESIDUAL_ADD, "rope_store": _lib.EPI_ROPE_STORE}


def fused_linear(x, weight, *, bias=None, norm_weight=None, norm_eps=1e-6, norm_folded=False, epilogue="none", out=None, rope=None,
                 x_packed_rows=None, out_packed=None, workspace=None, want_out=True, candidates=None, prefetch=None):
    """nvh_linear_small_m_ex: x [M<=64, K] . weight[N, K]^T with an optional RMSNorm prologue and one of the epilogues
    "none" (+bias) | "silu_mul" | "residual_add" (out = the residual stream, updated in place) | "rope_store"
    (rope = dict(positions, cos_sin, k_cache, v_cache, slot_mapping, num_heads, num_kv_heads, head_dim); returns q [M, H*D]).
    Streaming form: x_packed_rows = M when `x` is a flat fragment-order buffer (pack_rows); out_packed = a flat bf16 buffer
    that receives the result in fragment order as well; workspace = a ZERO-FILLED uint8 buffer of linear_workspace_bytes()
    (needed for K > 1024); want_out=False skips the row-major output when out_packed is given ("none" / "silu_mul");
    candidates = (val float32 [groups, stride], idx int32 [groups, stride]) with groups = linear_candidate_groups(n, k):
    per workgroup and row the best bf16 output and its column ("none" only) — with want_out=False the outputs themselves are
    never written (LM head + greedy arg-max in one pass; finish with greedy_advance_candidates).
    prefetch = a contiguous CUDA tensor the NEXT launch will stream (the following projection's weights): the CUs this launch
    leaves idle read it into the caches first (a hint; results do not depend on it)."""
    d, out = _linear_desc(x, weight, bias=bias, norm_weight=norm_weight, norm_eps=norm_eps, norm_folded=norm_folded, epilogue=epilogue, out=out,
                          rope=rope, x_packed_rows=x_packed_rows, out_packed=out_packed, workspace=workspace, want_out=want_out,
                          candidates=candidates, prefetch=prefetch)
    rc = _lib.load().nvh_linear_small_m_ex(ctypes.byref(d), NVH_BF16, _stream())
    _lib.check(rc, "nvh_linear_small_m_ex")
    return out


def _linear_desc(x, weight, *, bias=None, norm_weight=None, norm_eps=1e-6, norm_folded=False, epilogue="none", out=None, rope=None,
                 x_packed_rows=None, out_packed=None, workspace=None, want_out=True, candidates=None, prefetch=None):
    """Build the nvh_linear_desc of fused_linear's arguments; returns (descriptor, out)."""
    _require_gpu_bf16(x=x, weight=weight)
    n, k = weight.shape
    if x_packed_rows is not None:
        m = int(x_packed_rows)
        assert x.dim() == 1 and x.numel() >= ((m + 15) // 16) * 16 * k and x.is_contiguous()
    else:
        m = x.shape[0]
        assert x.shape[1] == k and x.stride(1) == 1
    assert weight.is_contiguous() and m <= LINEAR_SMALL_M_MAX
    d = _lib.LinearDesc()
    d.x, d.w, d.m, d.n, d.k = x.data_ptr(), weight.data_ptr(), m, n, k
    d.x_row_stride = k if x_packed_rows is not None else x.stride(0)
    d.x_packed = 1 if x_packed_rows is not None else 0
    d.bias = bias.data_ptr() if bias is not None else None
    if norm_folded:                                   # `weight` already carries the norm weight: w * diag(g)
        assert norm_weight is None
        d.norm_folded, d.norm_eps = 1, float(norm_eps)
    if norm_weight is not None:
        _require_gpu_bf16(norm_weight=norm_weight)
        assert norm_weight.numel() == k
        d.norm_weight, d.norm_eps = norm_weight.data_ptr(), float(norm_eps)
    if epilogue == "none":
        d.epilogue, cols = _lib.EPI_NONE, n
    elif epilogue == "silu_mul":
        d.epilogue, d.silu_inter, cols = _lib.EPI_SILU_MUL, n // 2, n // 2
    elif epilogue == "residual_add":
        assert out is not None and out.shape == (m, n) and out.dtype == torch.bfloat16 and out.stride(1) == 1
        d.epilogue, cols = _lib.EPI_RESIDUAL_ADD, n
    elif epilogue == "rope_store":
        h, kvh, hd = rope["num_heads"], rope["num_kv_heads"], rope["head_dim"]
        _require_i32(slot_mapping=rope["slot_mapping"])
        assert rope["positions"].dtype == torch.int64 and rope["cos_sin"].dtype == torch.float32 and rope["cos_sin"].is_contiguous()
        assert rope["k_cache"].is_contiguous() and rope["v_cache"].is_contiguous()
        d.epilogue, cols = _lib.EPI_ROPE_STORE, h * hd
        d.positions, d.cos_sin = rope["positions"].data_ptr(), rope["cos_sin"].data_ptr()
        d.k_cache, d.v_cache, d.slot_mapping = rope["k_cache"].data_ptr(), rope["v_cache"].data_ptr(), rope["slot_mapping"].data_ptr()
        d.h, d.kvh, d.hd = h, kvh, hd
    else:
        raise ValueError(epilogue)
    if out_packed is not None:
        _require_gpu_bf16(out_packed=out_packed)
        assert out_packed.is_contiguous() and out_packed.numel() >= ((m + 15) // 16) * 16 * cols
        d.out_packed = out_packed.data_ptr()
    if workspace is not None:
        assert workspace.is_cuda and workspace.dtype == torch.uint8 and workspace.is_contiguous()
        d.workspace, d.workspace_bytes = workspace.data_ptr(), workspace.numel()
    if candidates is not None:
        cv, ci = candidates
        assert epilogue == "none" and cv.dtype == torch.float32 and ci.dtype == torch.int32 and cv.is_cuda and ci.is_cuda
        assert cv.shape == ci.shape and cv.dim() == 2 and cv.is_contiguous() and ci.is_contiguous()
        assert cv.shape[0] >= linear_candidate_groups(n, k) > 0 and cv.shape[1] >= m
        d.candidate_val, d.candidate_idx, d.candidate_stride = cv.data_ptr(), ci.data_ptr(), cv.stride(0)
    if prefetch is not None:
        assert prefetch.is_cuda and prefetch.is_contiguous()
        d.prefetch, d.prefetch_bytes = prefetch.data_ptr(), prefetch.numel() * prefetch.element_size()
    if out is None and (want_out or (out_packed is None and candidates is None)):
        out = torch.empty((m, cols), dtype=torch.bfloat16, device=x.device)
    if out is not None:
        d.out, d.out_row_stride = out.data_ptr(), out.stride(0)
    return d, out


QKV_ATTEND_MODES = {"auto": 0, "two_launches": 1, "one_launch": 2, "two_launches_kv_prefetch": 3}


def qkv_rope_attend(x, weight, *, rope, context_lens, block_tables, bias=None, norm_eps=1e-6, norm_folded=True, x_packed_rows=None,
                    q_out=None, attn_out=None, attn_out_packed=None, softmax_scale=None, workspace=None, linear_workspace=None, prefetch=None,
                    mode="auto", spin_limit=0, missing_producers=0):
    """nvh_qkv_rope_attend: the front of a decode layer in ONE launch — fused_linear(..., epilogue="rope_store") followed by
    flash_attn_with_kvcache on its q rows and the caches it has just extended (models/qwen3.py:104-117 with
    layers/attention.py:84-86 and :99-101 inside).  Shapes the one-launch kernel does not serve (head_dim 128, k > 1024,
    row-major x, ...) run as the two launches with the same results.  Returns (q [M, H*D], attn_out [M, H, D], one_launch: bool).
    workspace = the decode workspace (default: the per-device one); linear_workspace = fused_linear's (split-K, k > 1024 only).
    mode / spin_limit / missing_producers: tests and A/B only (nvh_qkv_rope_attend_variant)."""
    d, q = _linear_desc(x, weight, bias=bias, norm_eps=norm_eps, norm_folded=norm_folded, epilogue="rope_store", out=q_out, rope=rope,
                        x_packed_rows=x_packed_rows, workspace=linear_workspace, prefetch=prefetch)
    h, kvh, hd = rope["num_heads"], rope["num_kv_heads"], rope["head_dim"]
    m = d.m
    k_cache = rope["k_cache"]
    bs = k_cache.shape[1]
    _require_i32(context_lens=context_lens, block_tables=block_tables)
    assert block_tables.dim() == 2 and block_tables.shape[0] == m and block_tables.stride(1) == 1
    assert context_lens.numel() == m and context_lens.is_contiguous()
    if attn_out is None:
        attn_out = torch.empty((m, h, hd), dtype=torch.bfloat16, device=q.device)
    else:
        assert attn_out.shape == (m, h, hd) and attn_out.is_contiguous() and attn_out.dtype == torch.bfloat16
    if attn_out_packed is not None:
        _require_gpu_bf16(attn_out_packed=attn_out_packed)
        assert attn_out_packed.is_contiguous() and attn_out_packed.numel() >= ((m + 15) // 16) * 16 * h * hd
    max_blocks = block_tables.shape[1]
    need = decode_workspace_bytes(m, h, hd, max_blocks, bs)
    if workspace is not None:
        assert workspace.is_cuda and workspace.dtype == torch.uint8 and workspace.is_contiguous() and workspace.numel() >= need
        ws = workspace
    else:
        ws = reserve_workspace(q.device, need)
    scale = float(hd ** -0.5 if softmax_scale is None else softmax_scale)
    fused = ctypes.c_int(0)
    rc = _lib.load().nvh_qkv_rope_attend_variant(
        QKV_ATTEND_MODES[mode], int(spin_limit), int(missing_producers), ctypes.byref(fused), ctypes.byref(d), attn_out.data_ptr(),
        attn_out_packed.data_ptr() if attn_out_packed is not None else None, block_tables.data_ptr(), context_lens.data_ptr(), bs, max_blocks,
        block_tables.stride(0), scale, NVH_BF16, ws.data_ptr(), ws.numel(), _stream())
    _lib.check(rc, "nvh_qkv_rope_attend")
    return q, attn_out, bool(fused.value)


def qkv_rope_attend_status(workspace=None, device=None) -> int:
    """Host-synchronous: non-zero after a one-launch call whose consumers gave up waiting for a producer (its rows are NaN)."""
    ws = workspace if workspace is not None else reserve_workspace(device if device is not None else torch.device("cuda", torch.cuda.current_device()), 1)
    v = ctypes.c_uint32(0)
    _lib.check(_lib.load().nvh_qkv_rope_attend_status(ws.data_ptr(), ctypes.byref(v)), "nvh_qkv_rope_attend_status")
    return int(v.value)


def linear_candidate_groups(n, k) -> int:
    return int(_lib.load().nvh_linear_small_m_candidate_groups(n, k))


def greedy_advance_candidates(cand_val, cand_idx, groups, n_rows, input_ids, positions, context_lens, slot_mapping, block_tables, block_size,
                              tokens_log, row_steps, embed=None):
    """greedy_advance on the candidate records of fused_linear(candidates=...) instead of logits.
    embed = (weight [vocab, hidden], hidden_out [n_rows, hidden], hidden_packed or None): the same launch also looks up the
    chosen tokens' embedding rows for the next step (row-major and, optionally, in fragment order)."""
    _require_i32(context_lens=context_lens, slot_mapping=slot_mapping, block_tables=block_tables)
    for t in (input_ids, positions, tokens_log, row_steps):
        assert t.dtype == torch.int64 and t.is_cuda
    if embed is not None:
        w, hid, packed = embed
        _require_gpu_bf16(embed_weight=w, hidden_out=hid)
        assert w.is_contiguous() and hid.shape == (n_rows, w.shape[1]) and hid.stride(1) == 1
        assert packed is None or (packed.dtype == torch.bfloat16 and packed.is_cuda and packed.numel() >= ((n_rows + 15) // 16) * 16 * w.shape[1])
        rc = _lib.load().nvh_greedy_advance_candidates_embed(
            cand_val.data_ptr(), cand_idx.data_ptr(), groups, cand_val.stride(0), n_rows, input_ids.data_ptr(), positions.data_ptr(),
            context_lens.data_ptr(), slot_mapping.data_ptr(), block_tables.data_ptr(), block_tables.stride(0), block_size,
            tokens_log.data_ptr(), tokens_log.stride(0), row_steps.data_ptr(), w.data_ptr(), w.shape[0], w.shape[1], hid.data_ptr(), hid.stride(0),
            None if packed is None else packed.data_ptr(), NVH_BF16, _stream())
        _lib.check(rc, "nvh_greedy_advance_candidates_embed")
        return
    rc = _lib.load().nvh_greedy_advance_candidates(cand_val.data_ptr(), cand_idx.data_ptr(), groups, cand_val.stride(0), n_rows,
                                                   input_ids.data_ptr(), positions.data_ptr(), context_lens.data_ptr(), slot_mapping.data_ptr(),
                                                   block_tables.data_ptr(), block_tables.stride(0), block_size, tokens_log.data_ptr(),
                                                   tokens_log.stride(0), row_steps.data_ptr(), _stream())
    _lib.check(rc, "nvh_greedy_advance_candidates")


def argmax_rows(logits):
    """Greedy token per row: argmax over the last dim of bf16 logits [M, N] (ties -> lowest index), int64."""
    _require_gpu_bf16(logits=logits)
    assert logits.dim() == 2 and logits.stride(1) == 1
    out = torch.empty(logits.shape[0], dtype=torch.int64, device=logits.device)
    rc = _lib.load().nvh_argmax_rows(out.data_ptr(), logits.data_ptr(), logits.shape[0], logits.shape[1], logits.stride(0), NVH_BF16, _stream())
    _lib.check(rc, "nvh_argmax_rows")
    return out


def greedy_advance(logits, input_ids, positions, context_lens, slot_mapping, block_tables, block_size, tokens_log, row_steps):
    """argmax_rows fused with the between-steps bookkeeping of a greedy decode session (nvh_greedy_advance): appends the
    token to `tokens_log[row_steps[r], r]` and rewrites next step's input_ids / positions / context_lens / slot_mapping in place."""
    _require_gpu_bf16(logits=logits)
    _require_i32(context_lens=context_lens, slot_mapping=slot_mapping, block_tables=block_tables)
    m = logits.shape[0]
    assert logits.dim() == 2 and logits.stride(1) == 1 and block_tables.stride(1) == 1
    for t in (input_ids, positions, tokens_log, row_steps):
        assert t.dtype == torch.int64 and t.is_cuda
    assert tokens_log.dim() == 2 and tokens_log.shape[1] >= m and tokens_log.stride(1) == 1
    rc = _lib.load().nvh_greedy_advance(logits.data_ptr(), m, logits.shape[1], logits.stride(0), input_ids.data_ptr(), positions.data_ptr(),
                                        context_lens.data_ptr(), slot_mapping.data_ptr(), block_tables.data_ptr(), block_tables.stride(0),
                                        block_size, tokens_log.data_ptr(), tokens_log.stride(0), row_steps.data_ptr(), NVH_BF16, _stream())
    _lib.check(rc, "nvh_greedy_advance")
