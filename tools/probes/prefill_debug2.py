import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nano-vllm-learn_amd"))
from nanovllm_hip import ops
from oracle import oracle as O
np.set_printoptions(linewidth=200, precision=3, suppress=True)
H, KVH, D = 1, 1, 64
T = 8
cu = torch.tensor([0, T], dtype=torch.int32).cuda()
def run(q, k, v):
    return ops.flash_attn_varlen_func(q.bfloat16().cuda(), k.bfloat16().cuda(), v.bfloat16().cuda(), T, cu, T, cu, out_dtype=torch.float32).cpu().numpy()
g = torch.Generator().manual_seed(0)
q = torch.randn(T, H, D, generator=g); k = torch.randn(T, KVH, D, generator=g)
vt = torch.arange(T).float().view(T, 1, 1).expand(T, 1, D).contiguous()      # V[t,:] = t
print("K=0, V[t]=t -> expect row r = mean(0..r):", [round(r / 2, 2) for r in range(T)])
print(run(q, torch.zeros_like(k), vt)[:, 0, :4])
vd = torch.arange(D).float().view(1, 1, D).expand(T, 1, D).contiguous()      # V[:,d] = d
print("K=0, V[:,d]=d -> expect every row = [0,1,2,...]")
print(run(q, torch.zeros_like(k), vd)[:3, 0, :20])
print("random K, V[t]=t vs oracle")
exp = O.prefill_varlen(q.bfloat16().float().numpy(), k.bfloat16().float().numpy(), vt.numpy(), [0, T], [0, T])
print(np.stack([run(q, k, vt)[:, 0, 0], exp[:, 0, 0]]))
# scores probe: one-hot V picks out softmax weights: V[t, d] = (t == d)
ve = torch.eye(T, D).view(T, 1, D)
exp = O.prefill_varlen(q.bfloat16().float().numpy(), k.bfloat16().float().numpy(), ve.numpy(), [0, T], [0, T])
print("softmax weights got:"); print(run(q, k, ve)[:, 0, :T]); print("expected:"); print(exp[:, 0, :T])
