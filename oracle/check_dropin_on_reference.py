#!/usr/bin/env python3
"""Prove the drop-in on the REFERENCE's own caller, in the build container only (where /root/reference exists).

    PYTHONDONTWRITEBYTECODE=1 python oracle/check_dropin_on_reference.py

TEST INFRASTRUCTURE ONLY — never shipped to the GPU box, never imported by the product; no forward pass, no GPU.
What it does, with the reference's code executed as it lies under /root/reference:
  1. the three edits of INTEGRATION.md section 2 are applied IN MEMORY to the text of nanovllm/config.py and
     nanovllm/models/qwen3.py (nothing is written anywhere): "hip" joins VALID_ATTN_BACKENDS (config.py:6) and the dispatch of
     Qwen3Attention.__init__ (qwen3.py:44-56) gets the `elif attn_backend == "hip"` branch returning this package's Attention;
  2. under a world-size-1 gloo group the patched module's Qwen3Attention and Qwen3ForCausalLM-style stack of decoder layers are
     CONSTRUCTED with attn_backend="hip" — i.e. the reference calls `Attention(num_heads, head_dim, scale, num_kv_heads,
     **{"block_size": block_size})` positionally (qwen3.py:89-95) on our class;
  3. the reference's own cache-binding loop — the text of ModelRunner.allocate_kv_cache from `layer_id = 0` to its end
     (engine/model_runner.py:146-157: `hasattr(module, "k_cache") and hasattr(module, "v_cache")`) — is executed over that module tree with a
     dummy cache tensor, and every hip Attention module must come out bound to its own layer's views;
  4. an unknown backend still raises ValueError, and the name does not start with "sdpa" (model_runner.py:24-26 would force eager).
Exit code 0 and a one-line summary on success.
"""
import inspect
import os
import sys
import textwrap
import types

REF = os.environ.get("NVH_REFERENCE", "/root/reference")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

HIP_BRANCH = '''        elif attn_backend == "hip":
            from nanovllm_hip.layers.attention_hip import Attention
            attn_kwargs = {"block_size": block_size}
'''


def patched_module(name, path, edit):
    """Execute the reference file at `path` as module `name` with `edit` applied to its text in memory."""
    src = edit(open(path).read())
    mod = types.ModuleType(name)
    mod.__file__ = path
    sys.modules[name] = mod
    exec(compile(src, path, "exec"), mod.__dict__)
    return mod


def main():
    if not os.path.isdir(os.path.join(REF, "nanovllm")):
        print(f"[dropin] {REF} not present: nothing to check here")
        return 0
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    sys.path.insert(0, os.path.join(ROOT, "nano-vllm-learn_amd"))
    import torch
    import torch.distributed as dist

    # --- 1. INTEGRATION.md section 2, applied to the reference's text in memory
    def edit_config(src):
        old = 'VALID_ATTN_BACKENDS = ("flash", "sdpa", "sdpa.math", "triton")'
        assert old in src, "config.py:6 changed: update INTEGRATION.md"
        return src.replace(old, 'VALID_ATTN_BACKENDS = ("flash", "sdpa", "sdpa.math", "triton", "hip")')

    def edit_qwen3(src):
        anchor = "        else:\n            raise ValueError(f\"Unknown attention backend: {attn_backend}\")"
        assert src.count(anchor) == 1, "qwen3.py:44-56 changed: update INTEGRATION.md"
        return src.replace(anchor, HIP_BRANCH + anchor)

    import nanovllm  # noqa: F401  (the package itself, unpatched)
    cfg_mod = patched_module("nanovllm.config", os.path.join(REF, "nanovllm", "config.py"), edit_config)
    assert "hip" in cfg_mod.VALID_ATTN_BACKENDS and not "hip".startswith("sdpa")
    qwen3 = patched_module("nanovllm.models.qwen3", os.path.join(REF, "nanovllm", "models", "qwen3.py"), edit_qwen3)

    # --- 2. construct the reference's modules with the hip backend (world size 1, gloo, CPU; no weights, no forward)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29571")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        from nanovllm_hip.layers.attention_hip import Attention as HipAttention
        layers, hidden, heads, kv_heads, head_dim, block = 3, 1024, 16, 8, 128, 256          # Qwen3-0.6B shapes, three layers
        stack = torch.nn.ModuleList([qwen3.Qwen3Attention(hidden_size=hidden, num_heads=heads, num_kv_heads=kv_heads, head_dim=head_dim,
                                                          attn_backend="hip", block_size=block) for _ in range(layers)])
        for m in stack:
            a = m.attn
            assert type(a) is HipAttention, type(a)
            assert (a.num_heads, a.head_dim, a.num_kv_heads, a.block_size) == (heads, head_dim, kv_heads, block)
            assert abs(a.scale - head_dim ** -0.5) < 1e-12                                    # qwen3.py:40: used as given
            assert hasattr(a, "k_cache") and hasattr(a, "v_cache") and a.k_cache.numel() == 0 # attention.py:72
        try:
            qwen3.Qwen3Attention(hidden_size=hidden, num_heads=heads, num_kv_heads=kv_heads, head_dim=head_dim, attn_backend="nope")
            raise AssertionError("an unknown backend must still raise")
        except ValueError:
            pass

        # --- 3. the reference's own binding loop (engine/model_runner.py:146-157), executed on this module tree
        from nanovllm.engine import model_runner as ref_runner
        src = inspect.getsource(ref_runner.ModelRunner.allocate_kv_cache)
        start = src.index("layer_id = 0")
        loop = textwrap.dedent(" " * (len(src[:start]) - len(src[:start].rstrip(" "))) + src[start:])
        assert 'hasattr(module, "k_cache") and hasattr(module, "v_cache")' in loop, "model_runner.py:148-157 changed"
        runner = types.SimpleNamespace(model=stack, kv_cache=torch.zeros(2, layers, 4, block, kv_heads, head_dim))
        exec(compile(loop, "model_runner.py:allocate_kv_cache[binding loop]", "exec"), {"self": runner})
        for i, m in enumerate(stack):
            assert m.attn.k_cache.data_ptr() == runner.kv_cache[0, i].data_ptr() and m.attn.k_cache.shape == (4, block, kv_heads, head_dim)
            assert m.attn.v_cache.data_ptr() == runner.kv_cache[1, i].data_ptr()
        bound = sum(1 for mod in stack.modules() if hasattr(mod, "k_cache") and hasattr(mod, "v_cache"))
        assert bound == layers, f"{bound} modules took a cache view, expected {layers}"
    finally:
        dist.destroy_process_group()
    print(f"[dropin] ok: reference Qwen3Attention built {layers} hip Attention modules (ctor kwargs accepted), "
          f"the reference's binding loop bound {bound} k_cache/v_cache views, unknown backends still raise")
    return 0


if __name__ == "__main__":
    sys.exit(main())
