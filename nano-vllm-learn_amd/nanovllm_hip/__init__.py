"""nanovllm_hip — MI355X (gfx950) paged-attention backend for nano-vllm (`--attn-backend hip`).

Host-side mirror of the reference's attention interface over the C-ABI library libnvh_attn.so."""
from .config import VALID_ATTN_BACKENDS, resolve_attention
from .utils.context import Context, get_context, reset_context, set_context

__all__ = ["VALID_ATTN_BACKENDS", "resolve_attention", "Context", "get_context", "set_context", "reset_context"]
