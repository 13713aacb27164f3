#!/usr/bin/env python3
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nano-vllm-learn_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from nanovllm_hip import ops
from oracle import oracle as O
from test_hip_parity import _decode_case, dev_i32
for (G, D, ctx) in [(7, 128, 16), (7, 128, 17), (7, 128, 33), (7, 128, 128), (7, 128, 129), (7, 128, 700), (1, 128, 700), (8, 128, 700), (16, 128, 700), (7, 64, 700)]:
    q, kc, vc, ctxs, bt = _decode_case(5, 1, G, 1, D, ctx, ctx)
    exp = O.paged_decode(q.float().numpy(), kc.float().numpy(), vc.float().numpy(), ctxs, bt)
    qd, kd, vd, cl, btd = q.cuda(), kc.cuda(), vc.cuda(), dev_i32(ctxs), dev_i32(bt)
    for name, kw in [("w8c1", dict(variant="chunked", waves=8, chunks=1)), ("w4c1", dict(variant="chunked", waves=4, chunks=1))]:
        outs = [ops.flash_attn_with_kvcache(qd, kd, vd, cl, btd, out_dtype=torch.float32, **kw).clone() for _ in range(30)]
        torch.cuda.synchronize()
        ndiff = sum(1 for o in outs[1:] if not torch.equal(o, outs[0]))
        err = max(float(np.abs(o.cpu().numpy() - exp).max()) for o in outs)
        extra = ""
        if ndiff:
            k = [i for i, o in enumerate(outs) if not torch.equal(o, outs[0])][0]
            d = (outs[k] - outs[0]).abs()
            extra = f" max|diff| {d.max().item():.2e} at {np.unravel_index(int(d.argmax()), d.shape)}; heads differing {sorted(set((d > 0).nonzero()[:, 1].tolist()))}; dims differing {len(set((d > 0).nonzero()[:, 2].tolist()))}"
        print(f"G{G} D{D} ctx{ctx} {name}: {ndiff}/29 differ; err {err:.2e}{extra}")
