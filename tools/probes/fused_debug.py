"""Where do the one-launch and two-launch forms of nvh_qkv_rope_attend differ?  (development probe)"""
import sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "nano-vllm-learn_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_hip_qkv_attend as T

def bits(t): return t.view(torch.int16)
for ctxs in ([1, 2, 16, 17, 255, 256, 257], [300], [300, 300], [1025 + 31 * i for i in range(32)], [113, 128, 129, 144, 145, 2047, 2033, 1, 15, 31, 32, 33, 1024, 1040, 1041, 3000]):
    B = len(ctxs)
    c = T._case(B, 14, 2, 896, ctxs, seed=B * 131 + 896, width=16)
    outs = {"two_launches": [], "one_launch": []}
    for rep in range(4):
        for mode in outs:
            kc, vc = c["kc"].clone(), c["vc"].clone()
            q, o, p, f = T._run(c, mode, kc, vc)
            torch.cuda.synchronize()
            outs[mode].append(o.clone())
    for mode, os_ in outs.items():
        print(len(ctxs), mode, "differs from its first run in", [int((bits(os_[0]) != bits(x)).sum()) for x in os_[1:]], "elements")
    print("   one vs two:", int((bits(outs["one_launch"][0]) != bits(outs["two_launches"][0])).sum()), "elements")
