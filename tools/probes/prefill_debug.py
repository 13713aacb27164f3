import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nano-vllm-learn_amd"))
from nanovllm_hip import ops
from oracle import oracle as O
H, KVH, D = 2, 1, 64
for lens in ([5], [16], [17], [33], [40, 3]):
    T = sum(lens)
    gen = torch.Generator().manual_seed(1)
    q = torch.randn(T, H, D, generator=gen).bfloat16(); k = torch.randn(T, KVH, D, generator=gen).bfloat16(); v = torch.randn(T, KVH, D, generator=gen).bfloat16()
    cu = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    exp = O.prefill_varlen(q.float().numpy(), k.float().numpy(), v.float().numpy(), cu, cu)
    got = ops.flash_attn_varlen_func(q.cuda(), k.cuda(), v.cuda(), max(lens), torch.from_numpy(cu).cuda(), max(lens), torch.from_numpy(cu).cuda(), out_dtype=torch.float32).cpu().numpy()
    err = np.abs(got - exp)
    print("lens", lens, "max err", err.max())
    print("  per-row max err:", np.round(err.max(axis=(1, 2)), 3).tolist())
    if err.max() > 1e-2:
        r = int(err.max(axis=(1, 2)).argmax())
        print("  worst row", r, "per-dim err head0:", np.round(err[r, 0], 2).tolist())
