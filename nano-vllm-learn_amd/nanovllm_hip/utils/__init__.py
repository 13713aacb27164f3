from .context import Context, get_context, reset_context, set_context

__all__ = ["Context", "get_context", "set_context", "reset_context"]
