#!/usr/bin/env python3
"""Diagnostic: per-wave timeline of ONE nvh_qkv_rope_attend launch (s_memrealtime stamps, 10 ns ticks), producers and consumers.
Builds a SEPARATE library with -DNVH_STAMPS (tools/probes/libnvh_attn_stamps.so); the shipped library never contains stamp code.
Usage: python tools/probes/stamp_qkv_attend.py [--batch 32 --ctx 1536]"""
import argparse, ctypes, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "nano-vllm-learn_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "probes", "libnvh_attn_stamps.so")
sys.path.insert(0, os.path.join(ROOT, "nano-vllm-learn_amd"))


def build(extra=()):
    srcs = [os.path.join(CSRC, f) for f in ("api.hip", "store_kvcache.hip", "paged_decode.hip", "prefill_mfma.hip", "rope_store.hip", "layer_ops.hip",
                                             "skinny_gemm.hip", "linear_stream.hip", "allreduce_oneshot.hip", "qkv_attend.hip")]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DNVH_STAMPS", *extra, "-mllvm", "-amdgpu-mfma-vgpr-form",
                    "-mllvm", "-amdgpu-kernarg-preload-count=14", "-Wno-unused-command-line-argument", *srcs, "-o", OUT], check=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32); ap.add_argument("--ctx", type=int, default=1536)
    ap.add_argument("--build-only", action="store_true"); ap.add_argument("--extra", nargs="*", default=[])
    args = ap.parse_args()
    if args.build_only:
        build([e.lstrip("=") for e in args.extra]); return
    os.environ["NVH_LIB_PATH"] = OUT
    from nanovllm_hip import _lib, ops
    from nanovllm_hip.models.qwen import cos_sin_table
    lib = _lib.load()
    B, H, KVH, D, K, bs = args.batch, 14, 2, 64, 896, 256
    nblk = (args.ctx + bs - 1) // bs
    nb = B * nblk + 1
    layers = 6
    g = torch.Generator().manual_seed(0)
    caches = [torch.randn(2, nb, bs, KVH, D, device="cuda", dtype=torch.bfloat16) for _ in range(layers)]
    ws_ = [(torch.randn((H + 2 * KVH) * D, K, generator=g) * 0.05).bfloat16().cuda() for _ in range(layers)]
    bias = torch.randn((H + 2 * KVH) * D, generator=g).bfloat16().cuda()
    bt = torch.randperm(nb - 1, generator=g)[: B * nblk].view(B, nblk).int().cuda()
    cl = torch.full((B,), args.ctx, dtype=torch.int32, device="cuda")
    pos = torch.full((B,), args.ctx - 1, dtype=torch.int64, device="cuda")
    slots = (bt[:, (args.ctx - 1) // bs].long() * bs + (args.ctx - 1) % bs).int()
    xp = ops.pack_rows(torch.randn(B, K, generator=g).bfloat16().cuda())
    table = cos_sin_table(D, 8192, 1e6, "cuda")
    nsplit = (nblk * bs + 255) // 256
    tiles = (H + 2 * KVH) * 2
    n_cons = B * KVH * nsplit * 8 * 8
    stamps = torch.zeros(n_cons + tiles * 8 * 8, dtype=torch.int64, device="cuda")
    lib.nvh_debug_set_stamps.argtypes = [ctypes.c_void_p]
    lib.nvh_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))

    def call(l):
        rope = dict(positions=pos, cos_sin=table, k_cache=caches[l][0], v_cache=caches[l][1], slot_mapping=slots, num_heads=H, num_kv_heads=KVH, head_dim=D)
        _, _, fused = ops.qkv_rope_attend(xp, ws_[l], rope=rope, context_lens=cl, block_tables=bt, bias=bias, x_packed_rows=B, mode="one_launch")
        assert fused
    for rep in range(3):
        for l in range(layers): call(l)
    torch.cuda.synchronize()
    stamps.zero_()
    call(0)
    torch.cuda.synchronize()
    raw = stamps.cpu().numpy().astype(np.float64) * 0.01
    cons = raw[:n_cons].reshape(-1, 8, 8)
    prod = raw[n_cons:].reshape(tiles, 8, 8)
    cons = cons[cons[:, 0, 0] > 0]
    t0 = min(cons[:, :, 0][cons[:, :, 0] > 0].min(), prod[:, :, 0][prod[:, :, 0] > 0].min())

    def table_of(st, names):
        for k, n in enumerate(names):
            ok = st[:, :, k] > 0
            if not ok.any():
                continue
            v = (st[:, :, k] - t0)[ok]
            print(f"  {k} {n:<44} min {v.min():6.2f} med {np.median(v):6.2f} p90 {np.percentile(v, 90):6.2f} max {v.max():6.2f}")
    print(f"B={B} ctx={args.ctx}: producers {tiles} workgroups, consumers {cons.shape[0]} live workgroups; us since the first wave's start")
    print(" producers (weight tile of the qkv projection):")
    table_of(prod, ["start", "all loads issued", "W, x, bias, position landed", "MFMAs done", "reduced + RoPE, tile in LDS", "stores issued",
                    "stores acknowledged + barrier", "ready counter added"])
    print(" consumers (attention chunk):")
    table_of(cons, ["start", "scalars (ctx, block ids)", "two passes issued, later ones touched", "hand-off seen (barrier)", "q landed, first QK^T + softmax done",
                    "first V landed", "all passes done", "merged + written (last arriver)"])
    end = max(cons[:, :, 7].max(), cons[:, :, 6].max())
    print(f" kernel span (first start -> last stamp) {end - t0:.2f} us")


if __name__ == "__main__":
    main()
