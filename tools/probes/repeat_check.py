#!/usr/bin/env python3
"""Repeatability / correctness probe for one decode geometry: runs the call N times per variant and reports how many
distinct results appear and the worst error against the oracle."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nano-vllm-learn_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from nanovllm_hip import ops
from oracle import oracle as O
from test_hip_parity import _decode_case, dev_i32

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rng = np.random.default_rng(1000 + seed)
D = int(rng.choice([64, 128])); KVH = int(rng.choice([1, 2, 3, 4, 8])); G = int(rng.choice([1, 2, 4, 7, 8])); B = int(rng.choice([1, 2, 3, 5, 9, 17, 33]))
H = KVH * G
hi = int(rng.choice([65, 257, 600, 1300])); lo = int(rng.choice([1, hi // 2]))
width = None if rng.random() < 0.5 else (hi + 255) // 256 + int(rng.integers(0, 3))
pad = int(rng.choice([-1, 0]))
q, kc, vc, ctxs, bt = _decode_case(2000 + seed, B, H, KVH, D, lo, hi, width, pad)
print("geometry", dict(D=D, KVH=KVH, G=G, B=B, ctxs=ctxs.tolist(), width=bt.shape[1]))
exp = O.paged_decode(q.float().numpy(), kc.float().numpy(), vc.float().numpy(), ctxs, bt)
qd, kd, vd, cl, btd = q.cuda(), kc.cuda(), vc.cuda(), dev_i32(ctxs), dev_i32(bt)
for name, kw in [("w4", dict(variant="chunked", waves=4)), ("w8", dict(variant="chunked", waves=8)), ("w8c1", dict(variant="chunked", waves=8, chunks=1)),
                 ("w8c2", dict(variant="chunked", waves=8, chunks=2)), ("w4c2", dict(variant="chunked", waves=4, chunks=2))]:
    outs = [ops.flash_attn_with_kvcache(qd, kd, vd, cl, btd, out_dtype=torch.float32, **kw).clone() for _ in range(50)]
    torch.cuda.synchronize()
    distinct = sum(1 for o in outs[1:] if not torch.equal(o, outs[0]))
    errs = [float(np.abs(o.cpu().numpy() - exp).max()) for o in outs]
    bad = (outs[0] != outs[[i for i, o in enumerate(outs) if not torch.equal(o, outs[0])][0]]).nonzero()[:5].tolist() if distinct else []
    print(f"{name}: {distinct}/49 differ from the first; err vs oracle min {min(errs):.2e} max {max(errs):.2e}; first differing coords {bad}")
