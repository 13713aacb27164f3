#!/usr/bin/env python3
"""Diagnostic: per-WAVE timeline of the short-sequence prefill kernel (BASELINE config 5: 128 sequences x 128 tokens), s_memrealtime, 10 ns ticks.
Slots per wave: 0 start, 1 DMA + first Q issued, 2 own loads landed, 3 barrier passed, 4 + 2r tiles of round r done, 5 + 2r stores of round r issued, 15 end."""
import ctypes, os, sys
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
B, S, H, KVH, D, NW = _arg("--batch", 128), _arg("--seq", 128), 14, 2, 64, _arg("--waves", 16)
T = B * S
bufs = [torch.randn(T, (H + 2 * KVH) * D, device="cuda", dtype=torch.bfloat16) for _ in range(6)]      # cold inputs: cycle over > 256 MiB? (6 x 37.7 MB: partly)
cu = torch.arange(0, T + 1, S, dtype=torch.int32, device="cuda")
out = torch.empty(T, H * D, device="cuda", dtype=torch.bfloat16)
stamps = torch.zeros(B * KVH * NW * 16, dtype=torch.int64, device="cuda")
lib.nvh_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
lib.nvh_prefill_varlen_variant.argtypes = [ctypes.c_int] * 2 + [ctypes.c_void_p] * 7 + [ctypes.c_int] * 8 + [ctypes.c_int64] * 4 + [ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
for qkv in bufs:
    q, k, v = qkv[:, :H * D], qkv[:, H * D:(H + KVH) * D], qkv[:, (H + KVH) * D:]
    rc = lib.nvh_prefill_varlen_variant(2, NW, out.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), cu.data_ptr(), cu.data_ptr(), None, B, S, S, H, KVH, D, 0, 0,
                                        qkv.stride(0), qkv.stride(0), qkv.stride(0), 0, D ** -0.5, 0, 0, None)
    assert rc == 0
torch.cuda.synchronize()
st = stamps.cpu().numpy().reshape(B * KVH, NW, 16).astype(np.float64) * 0.01
t0 = st[..., 0].min()
end = st[..., 15].max() - t0
print(f"kernel span (first wave start -> last wave end) {end:.2f} us; wave start med {np.median(st[..., 0] - t0):.2f} max {(st[..., 0] - t0).max():.2f}")
def med(a): return f"med {np.median(a):5.2f}  p90 {np.percentile(a, 90):5.2f}  max {a.max():5.2f}"
print("issue of DMA + first Q      :", med(st[..., 1] - st[..., 0]))
print("own loads landed (wait)     :", med(st[..., 2] - st[..., 1]))
print("barrier                     :", med(st[..., 3] - st[..., 2]))
print("since kernel start at barrier passed:", med(st[..., 3] - t0))
prev = st[..., 3]
for r in range(4):
    done, stored = st[..., 4 + 2 * r], st[..., 5 + 2 * r]
    ok = done > 0
    if not ok.any(): break
    print(f"round {r}: waves {int(ok.sum()):5d}  tiles {med((done - prev)[ok])}   finalise+store+next-Q move {med((stored - done)[ok])}")
    prev = np.where(ok, stored, prev)
print("wave end since kernel start :", med(st[..., 15] - t0))
wg_end = st[..., 15].max(axis=1) - t0
print("workgroup end               :", med(wg_end))
