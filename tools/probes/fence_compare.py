#!/usr/bin/env python3
"""Results of the fenced cross-check build of the decode kernel (argv[1]) against the shipped library, bit for bit, on geometries
that exercise the chunk hand-off (each library in its own process: a process binds one library)."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CHILD = r'''
import os, sys
import numpy as np, torch
ROOT = sys.argv[1]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nano-vllm-learn_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from nanovllm_hip import ops
from test_hip_parity import _decode_case, dev_i32
outs = {}
for i, (B, H, KVH, D, lo, hi) in enumerate([(32, 14, 2, 64, 1025, 2048), (1, 7, 1, 64, 4096, 4096), (3, 14, 2, 128, 650, 1300), (5, 7, 1, 128, 1, 900), (8, 16, 8, 128, 200, 700), (64, 14, 2, 64, 2049, 4096)]):
    q, kc, vc, ctxs, bt = _decode_case(900 + i, B, H, KVH, D, lo, hi, 16, 0)
    for rep in range(3):
        o = ops.flash_attn_with_kvcache(q.cuda(), kc.cuda(), vc.cuda(), dev_i32(ctxs), dev_i32(bt), out_dtype=torch.float32)
    torch.cuda.synchronize()
    outs[f"case{i}"] = o.cpu().numpy()
np.savez(sys.argv[2], **outs)
'''
res = []
for lib in (None, sys.argv[1]):
    env = dict(os.environ)
    if lib:
        env["NVH_LIB_PATH"] = lib
    else:
        env.pop("NVH_LIB_PATH", None)
    f = tempfile.mktemp(suffix=".npz")
    subprocess.run([sys.executable, "-c", CHILD, ROOT, f], check=True, env=env)
    res.append(dict(np.load(f)))
for k in sorted(res[0]):
    same = np.array_equal(res[0][k], res[1][k])
    print(f"{k}: shape {res[0][k].shape} fenced == shipped bit for bit: {same}")
    assert same
print("fenced and fence-free hand-offs agree bit for bit")
