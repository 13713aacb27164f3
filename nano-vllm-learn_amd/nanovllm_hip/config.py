"""Backend registration, mirroring nanovllm/config.py:6 and the dispatch at
nanovllm/models/qwen3.py:44-56.  `hip` must not start with "sdpa" or the reference's runner would
force eager mode (engine/model_runner.py:24-26); it is graph-captured like `flash`."""

VALID_ATTN_BACKENDS = ("flash", "sdpa", "sdpa.math", "triton", "hip")


def resolve_attention(attn_backend: str, block_size: int = 256):
    """Return (Attention class, ctor kwargs) for a backend flag, the way qwen3.py:44-56 does.
    Only `hip` is provided by this package; the other names belong to the reference."""
    if attn_backend not in VALID_ATTN_BACKENDS:
        raise ValueError(f"Unknown attention backend: {attn_backend}")
    if attn_backend == "hip":
        from .layers.attention_hip import Attention
        return Attention, {"block_size": block_size}
    raise ValueError(f"attention backend {attn_backend!r} lives in the reference package, not in nanovllm_hip")
