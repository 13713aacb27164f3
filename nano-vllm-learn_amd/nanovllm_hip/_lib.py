"""ctypes binding of libnvh_attn.so (C ABI: include/nvh_attn.h).

The library is the product; there is no CPU or PyTorch fallback.  If it is missing or fails to
load, every op raises — loudly — instead of computing something else.
"""
import ctypes
import os

import torch  # noqa: F401  (imported first on purpose: libnvh_attn.so binds to the HIP runtime torch loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NVH_LIB_PATH") or os.path.join(_HERE, "lib", "libnvh_attn.so")   # override: A/B of two builds

NVH_BF16 = 0
NVH_F32 = 1
IPC_HANDLE_BYTES = 64                                                         # NVH_COMM_IPC_HANDLE_BYTES
AR_EPI_NONE, AR_EPI_RESIDUAL_ADD = 0, 1

class LinearDesc(ctypes.Structure):
    """nvh_linear_desc (include/nvh_attn.h)."""
    _fields_ = [("out", ctypes.c_void_p), ("x", ctypes.c_void_p), ("w", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("m", ctypes.c_int32), ("n", ctypes.c_int32), ("k", ctypes.c_int32), ("silu_inter", ctypes.c_int32),
                ("x_row_stride", ctypes.c_int64), ("out_row_stride", ctypes.c_int64),
                ("norm_weight", ctypes.c_void_p), ("norm_eps", ctypes.c_float), ("epilogue", ctypes.c_int32),
                ("positions", ctypes.c_void_p), ("cos_sin", ctypes.c_void_p), ("k_cache", ctypes.c_void_p), ("v_cache", ctypes.c_void_p),
                ("slot_mapping", ctypes.c_void_p), ("h", ctypes.c_int32), ("kvh", ctypes.c_int32), ("hd", ctypes.c_int32),
                ("norm_folded", ctypes.c_int32), ("x_packed", ctypes.c_int32), ("out_packed", ctypes.c_void_p),
                ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_size_t),
                ("candidate_val", ctypes.c_void_p), ("candidate_idx", ctypes.c_void_p), ("candidate_stride", ctypes.c_int64),
                ("prefetch", ctypes.c_void_p), ("prefetch_bytes", ctypes.c_size_t)]


EPI_NONE, EPI_SILU_MUL, EPI_RESIDUAL_ADD, EPI_ROPE_STORE = 0, 1, 2, 3
PREFILL_KERNELS = {"auto": 0, "tiled": 1, "short": 2, "tiled_f16v": 3}                         # NVH_PREFILL_* (nvh_prefill_varlen_variant)
DECODE_VARIANTS = {"chunked": 0, "split_mfma": 1, "split_valu": 2, "chunked_p128": 3, "chunked_p256": 4, "chunked_p64": 5}      # NVH_DECODE_* (nvh_paged_decode_variant)

_c_i32p = ctypes.c_void_p
_SIGS = {
    "nvh_version": (ctypes.c_int, []),
    "nvh_last_error": (ctypes.c_char_p, []),
    "nvh_store_kvcache": (ctypes.c_int, [ctypes.c_void_p] * 4 + [_c_i32p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "nvh_paged_decode_workspace": (ctypes.c_size_t, [ctypes.c_int] * 5),
    "nvh_paged_decode": (ctypes.c_int, [ctypes.c_void_p] * 4 + [_c_i32p, _c_i32p] + [ctypes.c_int] * 6 +
                         [ctypes.c_int64, ctypes.c_int64, ctypes.c_float, ctypes.c_int, ctypes.c_int,
                          ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "nvh_paged_decode_variant": (ctypes.c_int, [ctypes.c_int] * 3 + [ctypes.c_void_p] * 4 + [_c_i32p, _c_i32p] + [ctypes.c_int] * 6 +
                                 [ctypes.c_int64, ctypes.c_int64, ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "nvh_paged_decode_packed": (ctypes.c_int, [ctypes.c_void_p] * 5 + [_c_i32p, _c_i32p] + [ctypes.c_int] * 6 +
                                [ctypes.c_int64, ctypes.c_int64, ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                 ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "nvh_decode_step": (ctypes.c_int, [ctypes.c_void_p] * 6 + [_c_i32p] * 3 + [ctypes.c_int] * 6 +
                        [ctypes.c_int64] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "nvh_prefill_varlen": (ctypes.c_int, [ctypes.c_void_p] * 4 + [_c_i32p] * 3 + [ctypes.c_int] * 8 +
                           [ctypes.c_int64] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "nvh_prefill_varlen_variant": (ctypes.c_int, [ctypes.c_int] * 2 + [ctypes.c_void_p] * 4 + [_c_i32p] * 3 + [ctypes.c_int] * 8 +
                                   [ctypes.c_int64] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "nvh_prefill_pv16_scratch_bytes": (ctypes.c_size_t, [ctypes.c_int] * 3),
    "nvh_prefill_pv16_uses_scratch": (ctypes.c_int, [ctypes.c_int] * 5),
    "nvh_prefill_varlen_pv16": (ctypes.c_int, [ctypes.c_void_p] * 4 + [_c_i32p] * 2 + [ctypes.c_int] * 7 + [ctypes.c_int64] * 3 +
                                [ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "nvh_bf16_rows_to_f16": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]),
    "nvh_rope_store": (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_float] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 +
                       [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "nvh_add_rmsnorm": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_int] + [ctypes.c_int64] * 3 +
                        [ctypes.c_int, ctypes.c_void_p]),
    "nvh_residual_add_pack": (ctypes.c_int, [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "nvh_silu_mul": (ctypes.c_int, [ctypes.c_void_p] * 2 + [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "nvh_linear_small_m": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "nvh_argmax_rows": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "nvh_greedy_advance": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int64] + [ctypes.c_void_p] * 5 +
                           [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]),
    "nvh_linear_small_m_ex": (ctypes.c_int, [ctypes.POINTER(LinearDesc), ctypes.c_int, ctypes.c_void_p]),
    "nvh_linear_small_m_workspace": (ctypes.c_size_t, [ctypes.c_int] * 4),
    "nvh_qkv_rope_attend": (ctypes.c_int, [ctypes.POINTER(LinearDesc), ctypes.c_void_p, ctypes.c_void_p, _c_i32p, _c_i32p, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int64, ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "nvh_qkv_rope_attend_variant": (ctypes.c_int, [ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(LinearDesc),
                                                   ctypes.c_void_p, ctypes.c_void_p, _c_i32p, _c_i32p, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                                   ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "nvh_qkv_rope_attend_status": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32)]),
    "nvh_pack_index": (ctypes.c_int64, [ctypes.c_int] * 3),
    "nvh_linear_small_m_candidate_groups": (ctypes.c_int, [ctypes.c_int] * 2),
    "nvh_comm_alloc": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]),
    "nvh_comm_free": (ctypes.c_int, [ctypes.c_void_p]),
    "nvh_comm_ipc_export": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
    "nvh_comm_ipc_open": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p)]),
    "nvh_comm_ipc_close": (ctypes.c_int, [ctypes.c_void_p]),
    "nvh_allreduce_stage_bytes": (ctypes.c_size_t, [ctypes.c_int, ctypes.c_int]),
    "nvh_allreduce_flag_bytes": (ctypes.c_size_t, [ctypes.c_int]),
    "nvh_allreduce_status": (ctypes.c_int, [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint32)]),
    "nvh_allreduce_oneshot": (ctypes.c_int, [ctypes.c_void_p] * 6 + [ctypes.c_int] * 4 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_size_t,
                                                                        ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "nvh_greedy_advance_candidates": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int] +
                                      [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                                                                ctypes.c_void_p, ctypes.c_void_p]),
    "nvh_greedy_advance_candidates_embed": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int] +
                                            [ctypes.c_void_p] * 5 + [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64,
                                                                      ctypes.c_void_p] +
                                            [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_int,
                                             ctypes.c_void_p]),
}
EXPORTS = tuple(_SIGS)

_lib = None


class NvhLibraryError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises NvhLibraryError if the .so is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NvhLibraryError(
            f"{LIB_PATH} not found: build it with `python nano-vllm-learn_amd/build.py` "
            "(the hip attention backend has no fallback path)")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise NvhLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in _SIGS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NvhLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    """Turn a non-zero return code into the exception the reference's callers expect."""
    if rc == 0:
        return
    msg = load().nvh_last_error().decode(errors="replace")
    text = f"{what} failed (code {rc}): {msg}"
    if "out of memory" in msg.lower():
        raise torch.cuda.OutOfMemoryError(text)        # bench_my.py:13-17 maps this to "OOM"
    raise RuntimeError(text)
