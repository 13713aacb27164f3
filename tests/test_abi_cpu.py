"""CPU checks of the drop-in boundary: the C-ABI library loads, exports every symbol include/nvh_attn.h
declares, validates arguments without touching a GPU, and the host-side mirror keeps the reference's
interface (names, constructor, attributes, Context fields).  No compute calls here."""
import ctypes
import inspect
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from nanovllm_hip import _lib
    return _lib.load()


def test_header_symbols_all_exported(lib):
    from nanovllm_hip import _lib
    header = open(os.path.join(ROOT, "include", "nvh_attn.h")).read()
    declared = set(re.findall(r"\b(nvh_[a-z_0-9]+)\s*\(", header))
    assert declared, "no declarations parsed"
    assert declared == set(_lib.EXPORTS), f"binding/header mismatch: {declared ^ set(_lib.EXPORTS)}"
    for name in declared:
        assert hasattr(lib, name), f"libnvh_attn.so does not export {name}"


def test_version_matches_header(lib):
    header = open(os.path.join(ROOT, "include", "nvh_attn.h")).read()
    assert lib.nvh_version() == int(re.search(r"#define NVH_VERSION (\d+)", header).group(1))


def test_workspace_is_pure_function_of_static_shapes(lib):
    a = lib.nvh_paged_decode_workspace(32, 14, 64, 16, 256)
    assert a == lib.nvh_paged_decode_workspace(32, 14, 64, 16, 256) and a > 0
    hdr = 65536 + 8192                                                       # fixed header (tickets + the fused launch's counters), then the partial records
    assert lib.nvh_paged_decode_workspace(64, 14, 64, 16, 256) - hdr == 2 * (a - hdr)
    assert lib.nvh_paged_decode_workspace(32, 14, 96, 16, 256) == 0          # unsupported head_dim
    # partial = (D + 2) floats per (row, head, split)
    assert (a - hdr) % ((64 + 2) * 4 * 32 * 14) == 0 and (a - hdr) // ((64 + 2) * 4 * 32 * 14) == 2 * 16


def test_argument_validation_without_gpu(lib):
    buf = (ctypes.c_char * 4096)()
    p = ctypes.addressof(buf)
    p = (p + 15) & ~15
    # null pointers
    assert lib.nvh_store_kvcache(None, None, None, None, None, 4, 2, 64, 128, 128, 0, None) == -5
    assert b"null" in lib.nvh_last_error()
    # zero work is a no-op that needs no pointers
    assert lib.nvh_store_kvcache(None, None, None, None, None, 0, 2, 64, 128, 128, 0, None) == 0
    assert lib.nvh_paged_decode(None, None, None, None, None, None, 0, 14, 2, 64, 256, 4, 896, 4, 0.125, 0, 0, None, 0, None) == 0
    # dtype / shape / stride / alignment / workspace checks fire before any launch
    assert lib.nvh_store_kvcache(p, p, p, p, p, 4, 2, 64, 128, 128, 7, None) == -1
    assert lib.nvh_store_kvcache(p, p, p, p, p, 4, 2, 64, 100, 128, 0, None) == -3
    assert lib.nvh_store_kvcache(p + 2, p, p, p, p, 4, 2, 64, 128, 128, 0, None) == -6
    assert lib.nvh_paged_decode(p, p, p, p, p, p, 1, 14, 2, 96, 256, 4, 1344, 4, 0.1, 0, 0, p, 1 << 20, None) == -2
    assert lib.nvh_paged_decode(p, p, p, p, p, p, 1, 14, 3, 64, 256, 4, 896, 4, 0.1, 0, 0, p, 1 << 20, None) == -2
    assert lib.nvh_paged_decode(p, p, p, p, p, p, 1, 14, 2, 64, 100, 4, 896, 4, 0.1, 0, 0, p, 1 << 20, None) == -2
    assert lib.nvh_paged_decode(p, p, p, p, p, p, 1, 14, 2, 64, 256, 4, 896, 4, 0.1, 0, 0, p, 16, None) == -4
    assert b"workspace" in lib.nvh_last_error()
    assert lib.nvh_prefill_varlen(p, p, p, p, p, p, None, 2, 16, 16, 14, 2, 64, 0, 0, 890, 128, 128, 0, 0.1, 0, 0, None) == -3
    # arg-max + advance + next step's embedding lookup: argument errors before any launch
    adv = (p, p, 8, 64, 4, p, p, p, p, p, 4, 256, p, 4, p)
    assert lib.nvh_greedy_advance_candidates_embed(*adv, None, 1000, 896, p, 896, p, 0, None) == -5          # no embedding table
    assert lib.nvh_greedy_advance_candidates_embed(*adv, p, 1000, 896, None, 896, p, 0, None) == -5          # no output rows
    assert lib.nvh_greedy_advance_candidates_embed(*adv, p, 1000, 900, p, 900, None, 0, None) == -2          # hidden % 32
    assert lib.nvh_greedy_advance_candidates_embed(*adv, p, 1000, 896, p, 800, None, 0, None) == -2          # row stride < hidden
    assert lib.nvh_greedy_advance_candidates_embed(*adv, p, 0, 896, p, 896, None, 0, None) == -2             # vocab
    assert lib.nvh_greedy_advance_candidates_embed(*adv, p + 2, 1000, 896, p, 896, None, 0, None) == -6
    assert lib.nvh_greedy_advance_candidates_embed(*adv, p, 1000, 896, p, 896, None, 3, None) == -1
    assert lib.nvh_greedy_advance_candidates_embed(p, p, 8, 64, 0, p, p, p, p, p, 4, 256, p, 4, p, None, 0, 0, None, 0, None, 0, None) == 0   # zero rows
    # the variant entry points validate their selectors
    dec = (p, p, p, p, p, p, 1, 14, 2, 64, 256, 4, 896, 4, 0.1, 0, 0, p, 1 << 20, None)
    assert lib.nvh_paged_decode_variant(6, 0, 0, *dec) == -2 and lib.nvh_paged_decode_variant(0, 5, 0, *dec) == -2
    assert lib.nvh_paged_decode_variant(2, 0, 0, p, p, p, p, p, p, 1, 16, 1, 64, 256, 4, 1024, 4, 0.1, 0, 0, p, 1 << 20, None) == -2   # VALU form: G <= 8
    pre = (p, p, p, p, p, p, None, 2, 16, 16, 14, 2, 64, 0, 0, 896, 128, 128, 0, 0.1, 0, 0, None)
    assert lib.nvh_prefill_varlen_variant(4, 0, *pre) == -2 and lib.nvh_prefill_varlen_variant(2, 4, *pre) == -2
    paged = (p, p, p, p, p, p, p, 2, 16, 16, 14, 2, 64, 256, 4, 896, 128, 128, 4, 0.1, 0, 0, None)
    assert lib.nvh_prefill_varlen_variant(3, 0, *paged) == -2                       # the fp16-V measurement variant never reads a paged cache
    # the fused front of a decode layer and the fp16 row conversion: argument errors before any launch
    assert lib.nvh_qkv_rope_attend(None, p, None, p, p, 256, 4, 4, 0.1, 0, p, 1 << 20, None) == -5
    assert lib.nvh_qkv_rope_attend_status(None, None) == -5
    assert lib.nvh_bf16_rows_to_f16(p, p, 4, 100, 128, 128, None) == -2 and lib.nvh_bf16_rows_to_f16(p, p, 4, 128, 100, 128, None) == -3
    assert lib.nvh_bf16_rows_to_f16(None, None, 0, 128, 128, 128, None) == 0
    # prefill with P V on the fp16 pipe: scratch size is a pure function of the shapes; argument errors before any stream operation
    assert lib.nvh_prefill_pv16_scratch_bytes(100, 2, 64) == 256 + 100 * 2 * 64 * 2 and lib.nvh_prefill_pv16_scratch_bytes(0, 2, 64) == 0
    pv = (p, p, p, p, p, p, 2, 1024, 1024, 2048, 14, 2, 64, 896, 128, 128, 0.125, 0, 0)
    assert lib.nvh_prefill_varlen_pv16(*pv, None, 0, None) == -5                      # no scratch
    assert lib.nvh_prefill_pv16_uses_scratch(2, 1024, 1024, 2, 64) == 1 and lib.nvh_prefill_pv16_uses_scratch(128, 128, 128, 2, 64) == 0
    assert lib.nvh_prefill_pv16_uses_scratch(16, 128, 128, 2, 64) == 1               # few sequences: the tiled kernel
    assert lib.nvh_prefill_varlen_pv16(*pv, p, 4096, None) == -4                      # scratch too small
    assert b"scratch" in lib.nvh_last_error()
    assert lib.nvh_prefill_varlen_pv16(*pv[:9], 0, *pv[10:], p, 1 << 20, None) == -4  # total_k missing
    assert lib.nvh_prefill_varlen_pv16(p, p, p, p, p, p, 0, 0, 0, 0, 14, 2, 64, 896, 128, 128, 0.125, 0, 0, None, 0, None) == 0   # empty batch


def test_ops_refuse_cpu_tensors():
    """No CPU fallback: the product path must fail loudly off-GPU."""
    from nanovllm_hip import ops
    q = torch.zeros(1, 14, 64, dtype=torch.bfloat16)
    kc = torch.zeros(1, 256, 2, 64, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.flash_attn_with_kvcache(q, kc, kc, torch.ones(1, dtype=torch.int32), torch.zeros(1, 1, dtype=torch.int32))
    with pytest.raises(RuntimeError, match="no CPU path"):
        ops.store_kvcache(q[:, :2], q[:, :2], kc, kc, torch.zeros(1, dtype=torch.int32))


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from nanovllm_hip import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libnvh_attn.so"))
    with pytest.raises(_lib.NvhLibraryError, match="no fallback"):
        _lib.load()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "nano-vllm-learn_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f"{f} imports the oracle"


def test_attention_module_mirrors_reference_interface():
    from nanovllm_hip import VALID_ATTN_BACKENDS, resolve_attention
    from nanovllm_hip.layers.attention_hip import Attention
    assert "hip" in VALID_ATTN_BACKENDS and not "hip".startswith("sdpa")
    for ref_name in ("flash", "sdpa", "sdpa.math", "triton"):             # config.py:6
        assert ref_name in VALID_ATTN_BACKENDS
    cls, kw = resolve_attention("hip", block_size=256)
    assert cls is Attention and kw == {"block_size": 256}
    with pytest.raises(ValueError):
        resolve_attention("nope")
    params = list(inspect.signature(Attention.__init__).parameters)
    assert params[1:5] == ["num_heads", "head_dim", "scale", "num_kv_heads"]     # attention.py:60-66, positional use qwen3.py:89-95
    m = Attention(14, 64, 0.125, 2, **kw)
    assert m.k_cache.numel() == 0 and m.v_cache.numel() == 0              # attention.py:72, duck-typed by model_runner.py:151
    assert list(inspect.signature(m.forward).parameters) == ["q", "k", "v"]


def test_context_fields_match_reference():
    from nanovllm_hip import Context, get_context, reset_context, set_context
    assert [f for f in Context.__dataclass_fields__] == ["is_prefill", "cu_seqlens_q", "cu_seqlens_k", "max_seqlen_q",
                                                         "max_seqlen_k", "slot_mapping", "context_lens", "block_tables"]
    set_context(False, slot_mapping=1, context_lens=2, block_tables=3)
    c = get_context()
    assert (c.is_prefill, c.slot_mapping, c.context_lens, c.block_tables) == (False, 1, 2, 3)
    reset_context()
    assert get_context() == Context()


def test_pack_index_matches_the_torch_packer():
    """nvh_pack_index (host-callable) is the layout ops.pack_rows / unpack_rows produce."""
    import torch
    from nanovllm_hip import _lib, ops
    lib = _lib.load()
    m, c = 37, 96
    x = torch.arange(m * c, dtype=torch.float32).view(m, c).bfloat16()
    xp = ops.pack_rows(x)
    assert xp.numel() == 48 * c
    for row, col in [(0, 0), (1, 0), (0, 8), (15, 31), (16, 0), (36, 95), (17, 40)]:
        assert xp[lib.nvh_pack_index(row, col, c)] == x[row, col]
    assert torch.equal(ops.unpack_rows(xp, m, c), x)
    assert lib.nvh_linear_small_m_workspace(32, 896, 896, 2) == 0           # K <= 1024: one workgroup per tile, no hand-off
    assert lib.nvh_linear_small_m_workspace(32, 896, 4864, 2) >= 56 * 4 + 56 * 5 * (2 * 256 + 32) * 4
