"""ctypes binding of libnvh_attn.so (C ABI: include/nvh_attn.h).

The library is the product; there is no CPU or PyTorch fallback.  If it is missing or fails to
load, every op raises — loudly — instead of computing something else.
"""
import ctypes
import os

import torch  # noqa: F401  (imported first on purpose: libnvh_attn.so binds to the HIP runtime torch loaded)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libnvh_attn.so")

NVH_BF16 = 0
NVH_F32 = 1

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
    "nvh_decode_step": (ctypes.c_int, [ctypes.c_void_p] * 6 + [_c_i32p] * 3 + [ctypes.c_int] * 6 +
                        [ctypes.c_int64] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_int,
                                                ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]),
    "nvh_prefill_varlen": (ctypes.c_int, [ctypes.c_void_p] * 4 + [_c_i32p] * 3 + [ctypes.c_int] * 8 +
                           [ctypes.c_int64] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
    "nvh_rope_store": (ctypes.c_int, [ctypes.c_void_p] * 5 + [ctypes.c_float] + [ctypes.c_void_p] * 3 + [ctypes.c_int] * 4 +
                       [ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "nvh_add_rmsnorm": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_int] + [ctypes.c_int64] * 3 +
                        [ctypes.c_int, ctypes.c_void_p]),
    "nvh_silu_mul": (ctypes.c_int, [ctypes.c_void_p] * 2 + [ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
    "nvh_linear_small_m": (ctypes.c_int, [ctypes.c_void_p] * 4 + [ctypes.c_int] * 4 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_void_p]),
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
