import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nano-vllm-learn_amd"))
from nanovllm_hip import ops
from nanovllm_hip.models.qwen import cos_sin_table
from oracle import oracle as O
H, KVH, D = 14, 2, 64
gen = torch.Generator().manual_seed(H * D)
n = 45
table = cos_sin_table(D, 4096, 1e6, "cuda")
qkv = torch.randn(n, (H + 2 * KVH) * D, generator=gen).bfloat16()
pos = torch.randint(0, 4096, (n,), generator=gen)
x = qkv.float().numpy()
q = x[:, :H * D].reshape(n, H, D)
q_exp = O.rope_neox(q, pos.numpy(), table.cpu().numpy())
qkv_d = qkv.cuda()
ops.rope_store(qkv_d, pos.cuda(), table, H, KVH, D)
torch.cuda.synchronize()
got = qkv_d.float().cpu().numpy()[:, :H * D].reshape(n, H, D)
bad = np.argwhere(got != q_exp)
print("mismatches", len(bad), "of", got.size)
for t, h, e in bad[:8]:
    i = e % (D // 2)
    cs = table[pos[t]].cpu().numpy()
    x1, x2 = q[t, h, i], q[t, h, i + D // 2]
    print(t, h, e, "got", got[t, h, e], "exp", q_exp[t, h, e], "x1", x1, "x2", x2, "cos", cs[i], "sin", cs[D // 2 + i],
          "f32:", np.float32(x1 * cs[i]) - np.float32(x2 * cs[D // 2 + i]) if e < D // 2 else np.float32(x2 * cs[i]) + np.float32(x1 * cs[D // 2 + i]))
