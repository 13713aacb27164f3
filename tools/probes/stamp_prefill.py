#!/usr/bin/env python3
"""Diagnostic: per-tile timeline of the prefill kernel (wave 0 of each workgroup, s_memrealtime, 10 ns ticks)."""
import ctypes, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tools", "probes"))
import stamp_decode
OUT = stamp_decode.OUT
if not os.path.exists(OUT) or "--build" in sys.argv:
    stamp_decode.build()
lib = ctypes.CDLL(OUT)
def _arg(name, default):
    return int(sys.argv[sys.argv.index(name) + 1]) if name in sys.argv else default
B, S, H, KVH, D = _arg("--batch", 16), _arg("--seq", 1024), 14, 2, 64
T = B * S
qkv = torch.randn(T, (H + 2 * KVH) * D, device="cuda", dtype=torch.bfloat16)
q, k, v = qkv[:, :H * D], qkv[:, H * D:(H + KVH) * D], qkv[:, (H + KVH) * D:]
cu = torch.arange(0, T + 1, S, dtype=torch.int32, device="cuda")
out = torch.empty(T, H * D, device="cuda", dtype=torch.bfloat16)
nq = S // 64
stamps = torch.zeros(B * H * nq * 32, dtype=torch.int64, device="cuda")
lib.nvh_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
lib.nvh_prefill_varlen.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int] * 8 + [ctypes.c_int64] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
for _ in range(3):
    rc = lib.nvh_prefill_varlen(out.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), cu.data_ptr(), cu.data_ptr(), None, B, S, S, H, KVH, D, 0, 0,
                                qkv.stride(0), qkv.stride(0), qkv.stride(0), 0, D ** -0.5, 0, 0, None)
    assert rc == 0
torch.cuda.synchronize()
st = stamps.cpu().numpy().reshape(B, H, nq, 32).astype(np.float64) * 0.01
t0 = st[..., 0].min()
print(f"kernel span {st[..., 31].max() - t0:.1f} us; workgroup start: med {np.median(st[..., 0] - t0):.1f} max {(st[..., 0] - t0).max():.1f}")
print(f"Q load wait: med {np.median(st[..., 1] - st[..., 0]):.2f} us")
for qt in sorted({0, min(3, nq - 1), nq - 1}):
    w = st[:, :, qt]
    n = qt + 1
    dma = [np.median(w[..., 2 + 3 * i] - (w[..., 1] if i == 0 else w[..., 4 + 3 * (i - 1)])) for i in range(min(n, 9))]
    bar = [np.median(w[..., 3 + 3 * i] - w[..., 2 + 3 * i]) for i in range(min(n, 9))]
    comp = [np.median(w[..., 4 + 3 * i] - w[..., 3 + 3 * i]) for i in range(min(n, 9))]
    print(f"q-tile {qt:2d}: life med {np.median(w[..., 31] - w[..., 0]):6.2f} us; per tile: stage+wait {np.round(dma, 2).tolist()} barrier {np.round(bar, 2).tolist()} compute {np.round(comp, 2).tolist()}")
life = st[..., 31] - st[..., 0]
span = st[..., 31].max() - t0
print(f"mean resident workgroups per CU: {life.sum() / span / 256:.2f}  (sum of lives {life.sum():.0f} us over span {span:.1f} us)")
tiles = sum(qt + 1 for qt in range(nq)) * B * H
print(f"CU-time per workgroup-tile: {span * 256 / tiles * 1e3:.0f} ns  ({tiles} tiles)")
# census: workgroups alive at sampled times, and by q-tile
starts, ends = st[..., 0].ravel() - t0, st[..., 31].ravel() - t0
for t in np.linspace(0, span, 13)[1:-1]:
    alive = ((starts <= t) & (ends > t)).sum()
    print(f"  t = {t:6.1f} us: {alive:5d} workgroups alive ({alive / 256:.2f} per CU), {int((starts > t).sum()):5d} not yet started")
