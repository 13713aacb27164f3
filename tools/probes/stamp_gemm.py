#!/usr/bin/env python3
"""Diagnostic: per-wave phase timeline of the decode-sized GEMM (linear_small_m) at the four per-layer shapes of Qwen2.5-0.5B
(s_memrealtime stamps, 10 ns ticks; separate -DNVH_STAMPS library, as stamp_decode.py)."""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools", "probes"))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-learn_amd"))
import stamp_decode
if not os.path.exists(stamp_decode.OUT) or "--build" in sys.argv:
    stamp_decode.build()
from nanovllm_hip import _lib, ops
_lib.LIB_PATH = stamp_decode.OUT                   # the diagnostic build, not the shipped library
lib = _lib.load()
lib.nvh_debug_set_stamps.argtypes = [ctypes.c_void_p]
M, HID, INTER, H, KVH, D = 32, 896, 4864, 14, 2, 64
dev = "cuda"
torch.manual_seed(0)
x = torch.randn(M, HID, device=dev, dtype=torch.bfloat16)
act = torch.randn(M, INTER, device=dev, dtype=torch.bfloat16)
res = torch.randn(M, HID, device=dev, dtype=torch.bfloat16)
# rotate over > 256 MB of weights per shape so every call streams from HBM (the Infinity Cache holds 256 MB);
# "warm" = the same weights every call (what a prefetch of the next GEMM's weights would give)
def copies(n, k):
    cnt = int(320e6 / (n * k * 2)) + 1
    return [torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(cnt)]
w_gu, w_dn, w_o = copies(2 * INTER, HID), copies(HID, INTER), copies(HID, H * D)
stamps = torch.zeros(1024 * 64, dtype=torch.int64, device=dev)
lib.nvh_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
names = ["entry", "W DMA issued", "x loads issued", "all landed", "MFMAs done", "reduced / last arriver summed", "end"]
ws = torch.zeros(4 << 20, dtype=torch.uint8, device=dev)

def report(tag, nwg=None, waves=8):
    """Workgroups and waves are taken from the stamps themselves (a slot that was never written stays zero), so the report
    follows whatever grid the launcher chose (wide tiles, balanced splits, 4 or 8 waves)."""
    torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(-1, 8, 8)[:, :, :7].astype(np.float64) * 0.01
    st = st[st[:, 0, 0] > 0]                                 # workgroups that ran
    waves = int((st[0, :, 0] > 0).sum())
    st = st[:, :waves, :]
    live = st[..., 6] > 0                                    # workgroups that were not the last arriver stop after stamp 4
    t0 = st[..., 0].min()
    print(f"{tag}: {st.shape[0]} workgroups x {waves} waves; span (first entry -> last end) {st[..., 6].max() - t0:.2f} us")
    for k, n in enumerate(names):
        sel = live if k >= 5 else np.ones_like(live)
        v = (st[..., k] - t0)[sel]
        d = ""
        if k:
            dd = (st[..., k] - st[..., k - 1])[sel]
            d = f"   delta: med {np.median(dd):5.2f} p90 {np.percentile(dd, 90):5.2f} max {dd.max():5.2f}"
        print(f"   {k} {n:<30} min {v.min():6.2f} med {np.median(v):6.2f} p90 {np.percentile(v, 90):6.2f} max {v.max():6.2f}{d}")

def run(tag, nwg, fn, ws_list):
    for warm in (False, True):
        n = 2 * len(ws_list)
        for l in range(n - 1):
            fn(ws_list[0 if warm else l % len(ws_list)])
        torch.cuda.synchronize()
        stamps.zero_()                                         # the timeline of ONE call (the last arriver differs per call)
        fn(ws_list[0 if warm else (n - 1) % len(ws_list)])
        report(tag + (" [warm: same weights every call]" if warm else " [cold: weights from HBM]"), nwg)

from nanovllm_hip.models.qwen import cos_sin_table
w_qkv = copies((H + 2 * KVH) * D, HID)
b_qkv = torch.randn((H + 2 * KVH) * D, device=dev, dtype=torch.bfloat16)
cs = cos_sin_table(D, 4096, 1e6, dev)
pos = torch.randint(0, 4096, (M,), device=dev)
slots = torch.randperm(8 * 256, device=dev)[:M].int()
kvc = torch.zeros(2, 8, 256, KVH, D, dtype=torch.bfloat16, device=dev)
rope = dict(positions=pos, cos_sin=cs, k_cache=kvc[0], v_cache=kvc[1], slot_mapping=slots, num_heads=H, num_kv_heads=KVH, head_dim=D)
xp = ops.pack_rows(x)
actp = ops.pack_rows(act)
run("gate_up  same, packed x", INTER // 16, lambda w: ops.fused_linear(xp, w, x_packed_rows=M, norm_folded=True, norm_eps=1e-6, epilogue="silu_mul"), w_gu)
run("down     same, packed x", 5 * HID // 16, lambda w: ops.fused_linear(actp, w, x_packed_rows=M, epilogue="residual_add", out=res, workspace=ws), w_dn)
run("qkv      K=896 N=1152 folded-norm + bias + RoPE + KV store, packed x", (H + 2 * KVH) * D // 32, lambda w: ops.fused_linear(xp, w, x_packed_rows=M, bias=b_qkv, norm_folded=True, norm_eps=1e-6, epilogue="rope_store", rope=rope), w_qkv)
run("o_proj   same, packed x", HID // 16, lambda w: ops.fused_linear(xp, w, x_packed_rows=M, epilogue="residual_add", out=res), w_o)
