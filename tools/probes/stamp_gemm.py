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
L = 6                                         # rotate weights so every call streams from HBM, not L2/MALL of its own previous call
w_gu = [torch.randn(2 * INTER, HID, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(L)]
w_dn = [torch.randn(HID, INTER, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(L)]
w_o = [torch.randn(HID, H * D, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(L)]
stamps = torch.zeros(1024 * 64, dtype=torch.int64, device=dev)
lib.nvh_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
names = ["entry", "kernargs arrived", "x loads issued", "W DMA issued", "first group landed", "K loop done", "after barrier", "end"]
order = [0, 6, 7, 1, 2, 3, 4, 5]

def report(tag, nwg, waves):
    torch.cuda.synchronize()
    st = stamps.cpu().numpy()[: nwg * 64].reshape(nwg, 8, 8)[:, :waves, :][:, :, order].astype(np.float64) * 0.01
    t0 = st[..., 0].min()
    print(f"{tag}: {nwg} workgroups x {waves} waves; span (first entry -> last end) {st[..., 7].max() - t0:.2f} us")
    for k, n in enumerate(names):
        v = st[..., k] - t0
        d = "" if k == 0 else f"   delta: med {np.median(st[..., k] - st[..., k - 1]):5.2f} p90 {np.percentile(st[..., k] - st[..., k - 1], 90):5.2f} max {(st[..., k] - st[..., k - 1]).max():5.2f}"
        print(f"   {k} {n:<20} min {v.min():6.2f} med {np.median(v):6.2f} p90 {np.percentile(v, 90):6.2f} max {v.max():6.2f}{d}")

for rep in range(3):
    for l in range(L):
        ops.fused_linear(x, w_gu[l], norm_folded=True, norm_eps=1e-6, epilogue="silu_mul")
report("gate_up  K=896 N=9728 silu (folded norm)", INTER // 16, 4)
for rep in range(3):
    for l in range(L):
        ops.fused_linear(act, w_dn[l], epilogue="residual_add", out=res)
report("down     K=4864 N=896 residual", HID // 16, 8)
for rep in range(3):
    for l in range(L):
        ops.fused_linear(x, w_o[l], epilogue="residual_add", out=res)
report("o_proj   K=896 N=896 residual", HID // 16, 4)
