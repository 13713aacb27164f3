#!/usr/bin/env python3
"""Probe: what does a decode GEMM gain when ANOTHER kernel has just read its weights (a prefetch into the Infinity Cache / some XCD's L2)?
Cycles over > 256 MB of weight copies per shape; per copy either [gemm] (cold), [touch, gemm] (a torch reduction over the copy first: every CU reads
a share, so lines land in the memory-side cache and in arbitrary XCDs' L2s), or [gemm, gemm] (the second one is the L2-warm bound).
Run under rocprofv3 --kernel-trace --stats and compare the average duration of linear_stream_kernel between the modes (one mode per process):
    python tools/probes/prefetch_probe.py {cold|touch|twice} {gate_up|down|qkv|o_proj}"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-learn_amd"))
from nanovllm_hip import ops
mode, shape = sys.argv[1], sys.argv[2]
M, HID, INTER, H, KVH, D = 32, 896, 4864, 14, 2, 64
dev = "cuda"
torch.manual_seed(0)
x = torch.randn(M, HID, device=dev, dtype=torch.bfloat16)
act = torch.randn(M, INTER, device=dev, dtype=torch.bfloat16)
res = torch.randn(M, HID, device=dev, dtype=torch.bfloat16)
ws = torch.zeros(4 << 20, dtype=torch.uint8, device=dev)
xp, actp = ops.pack_rows(x), ops.pack_rows(act)
def copies(n, k):
    return [torch.randn(n, k, device=dev, dtype=torch.bfloat16) * 0.02 for _ in range(int(320e6 / (n * k * 2)) + 1)]
if shape == "gate_up":
    W = copies(2 * INTER, HID); fn = lambda w: ops.fused_linear(xp, w, x_packed_rows=M, norm_folded=True, norm_eps=1e-6, epilogue="silu_mul")
elif shape == "down":
    W = copies(HID, INTER); fn = lambda w: ops.fused_linear(actp, w, x_packed_rows=M, epilogue="residual_add", out=res, workspace=ws)
elif shape == "o_proj":
    W = copies(HID, H * D); fn = lambda w: ops.fused_linear(xp, w, x_packed_rows=M, epilogue="residual_add", out=res)
else:
    raise SystemExit("shape")
sink = torch.zeros(1, device=dev)
for rep in range(6):
    for w in W:
        if mode == "touch":
            sink += w.view(torch.int32).view(-1)[::1].sum()          # reads every byte of the copy (a reduction over all CUs)
        fn(w)
        if mode == "twice":
            fn(w)
torch.cuda.synchronize()
print("done", mode, shape, len(W))
