#!/usr/bin/env python3
"""Diagnostic: per-wave phase timeline of the decode split kernel (s_memrealtime stamps, 10 ns ticks).
Builds a SEPARATE library with -DNVH_STAMPS (tools/probes/libnvh_attn_stamps.so); the shipped library
never contains stamp code.  Usage: python tools/probes/stamp_decode.py [--batch 32 --ctx 1536]"""
import argparse, ctypes, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "nano-vllm-learn_amd", "csrc")
OUT = os.path.join(ROOT, "tools", "probes", "libnvh_attn_stamps.so")

def build(tail=False):
    srcs = [os.path.join(CSRC, f) for f in ("api.hip", "store_kvcache.hip", "paged_decode.hip", "prefill_mfma.hip", "rope_store.hip", "layer_ops.hip", "skinny_gemm.hip", "linear_stream.hip", "allreduce_oneshot.hip", "qkv_attend.hip")]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DNVH_STAMPS", *(["-DNVH_STAMPS_TAIL"] if tail else []), "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form",
                    "-mllvm", "-amdgpu-kernarg-preload-count=14",
                    "-Wno-unused-command-line-argument", *srcs, "-o", OUT], check=True)

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32); ap.add_argument("--ctx", type=int, default=1536)
    ap.add_argument("--heads", type=int, default=14); ap.add_argument("--kv-heads", type=int, default=2)
    ap.add_argument("--head-dim", type=int, default=64); ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--tail", action="store_true", help="slots 1..5 follow the hand-off tail instead of the first pass (always rebuilds)")
    args = ap.parse_args()
    if args.build_only or args.tail or not os.path.exists(OUT):
        build(args.tail)
        if args.build_only: return
    lib = ctypes.CDLL(OUT)
    b, h, kvh, d, bs = args.batch, args.heads, args.kv_heads, args.head_dim, 256
    nblk = (args.ctx + bs - 1) // bs
    nb = b * nblk + 1
    layers = 8
    caches = [torch.randn(2, nb, bs, kvh, d, device="cuda", dtype=torch.bfloat16) for _ in range(layers)]
    perm = torch.randperm(nb - 1)[: b * nblk].view(b, nblk).int().cuda()
    cl = torch.full((b,), args.ctx, dtype=torch.int32, device="cuda")
    q = torch.randn(b, h, d, device="cuda", dtype=torch.bfloat16)
    out = torch.empty_like(q)
    nsplit = (nblk * bs + 255) // 256
    lib.nvh_paged_decode_workspace.restype = ctypes.c_size_t
    lib.nvh_paged_decode_workspace.argtypes = [ctypes.c_int] * 5
    ws = torch.zeros(lib.nvh_paged_decode_workspace(b, h, d, nblk, bs), dtype=torch.uint8, device="cuda")   # ticket header + chunk records
    waves = 8                                                         # waves per workgroup of the chunked kernel (both head dims)
    stamps = torch.zeros(b * kvh * nsplit * waves * 8, dtype=torch.int64, device="cuda")
    lib.nvh_debug_set_stamps(ctypes.c_void_p(stamps.data_ptr()))
    lib.nvh_paged_decode.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int] * 6 + [ctypes.c_int64, ctypes.c_int64, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    def call(l):
        rc = lib.nvh_paged_decode(out.data_ptr(), q.data_ptr(), caches[l][0].data_ptr(), caches[l][1].data_ptr(), perm.data_ptr(), cl.data_ptr(),
                                  b, h, kvh, d, bs, nblk, h * d, nblk, d ** -0.5, 0, 0, ws.data_ptr(), ws.numel(), None)
        assert rc == 0, rc
    for rep in range(3):
        for l in range(layers): call(l)
    torch.cuda.synchronize()
    stamps.zero_()                                                             # the timeline of ONE call (the last arriver differs per call)
    call(0)
    torch.cuda.synchronize()
    st = stamps.cpu().numpy().reshape(-1, waves, 8).astype(np.float64) * 0.01     # us
    st = st[st[:, 0, 1] > 0]                                                   # live workgroups only
    t0 = st[:, :, 0].min()
    names = ["start", "waves merged in LDS", "record stores issued", "stores acknowledged + barrier", "ticket returned + barrier", "records read + merged (last arriver)",
             "all passes done", "written (last arriver)"] if args.tail else ["start", "scalars+branch", "first loads issued", "first K landed", "first QK+softmax done", "first V landed", "all passes done",
             "merged + written (last arriver)"]
    print(f"live workgroups {st.shape[0]}, kernel span {st[:, :, 7].max() - t0:.2f} us (first wave start -> last wave end)")
    order = [0, 6, 1, 2, 3, 4, 5, 7] if args.tail else list(range(8))
    for pos, k in enumerate(order):
        n = names[k]
        ok = st[:, :, k] > 0
        v = (st[:, :, k] - t0)[ok]
        line = f"  {k} {n:<32} abs: min {v.min():6.2f} med {np.median(v):6.2f} p90 {np.percentile(v, 90):6.2f} max {v.max():6.2f}"
        if pos:
            kp = order[pos - 1]
            both = ok & (st[:, :, kp] > 0)
            d = (st[:, :, k] - st[:, :, kp])[both]
            line += f"   delta vs prev: med {np.median(d):5.2f} max {d.max():5.2f}"
        print(line)
    if args.tail:                                                              # per-workgroup critical path (the barrier hides wave skew)
        done = st[:, :, 6].max(axis=1); first_done = np.where(st[:, :, 6] > 0, st[:, :, 6], np.inf).min(axis=1)
        merged = st[:, :, 1].max(axis=1)
        print(f"  per workgroup: wave skew at the end of the passes med {np.median(done - first_done):.2f} max {(done - first_done).max():.2f}; "
              f"last wave done -> merged med {np.median(merged - done):.2f} max {(merged - done).max():.2f}")
        last = st[:, 0, 5] > 0
        seq = [6, 1, 2, 3, 4, 5, 7]
        tl = [np.median(st[last][:, :, k].max(axis=1) - t0) for k in seq]
        print("  last arrivers (median abs, max over waves):", " ".join(f"{names[k].split(' (')[0]}={v:.2f}" for k, v in zip(seq, tl)))
        starts = st[:, :, 0].min(axis=1) - t0
        print(f"  workgroup start: med {np.median(starts):.2f} p90 {np.percentile(starts, 90):.2f} max {starts.max():.2f}; passes per workgroup differ (ctx {args.ctx})")
if __name__ == "__main__":
    main()
