#!/usr/bin/env python3
"""Turn two rocprofv3 counter-collection CSVs (one pass with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE, both of
`tools/microbench.py decode ...`) into the per-launch HBM traffic record bench.py attaches as roofline.traffic.
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts half the bytes of wide coalesced reads -> doubled;
WRITE_SIZE is exact; both are in KB.
usage: pmc_traffic.py FETCH.csv WRITE.csv OUT.json --batch 32 --ctx 1536 --heads 14 --kv-heads 2 --head-dim 64"""
import argparse, csv, json, re, statistics, sys

ap = argparse.ArgumentParser()
ap.add_argument("fetch"); ap.add_argument("write"); ap.add_argument("out")
ap.add_argument("--batch", type=int, default=32); ap.add_argument("--ctx", type=int, default=1536)
ap.add_argument("--heads", type=int, default=14); ap.add_argument("--kv-heads", type=int, default=2)
ap.add_argument("--head-dim", type=int, default=64); ap.add_argument("--block-size", type=int, default=256)
ap.add_argument("--append", action="store_true", help="OUT.json holds {'records': [...]}: add this record to it (one per profiled context)")
a = ap.parse_args()

def per_kernel(path, counter):
    vals = {}
    for r in csv.DictReader(open(path)):
        m = re.search(r"paged_decode_\w+", r["Kernel_Name"])
        if m and r["Counter_Name"] == counter:
            vals.setdefault(m.group(0), []).append(float(r["Counter_Value"]))
    return {k: statistics.mean(v[len(v) // 4:]) for k, v in vals.items()}       # skip the cold first quarter

f, w = per_kernel(a.fetch, "FETCH_SIZE"), per_kernel(a.write, "WRITE_SIZE")
kernels, total = {}, 0
for k in sorted(set(f) | set(w)):
    b = int(round((2 * f.get(k, 0.0) + w.get(k, 0.0)) * 1024))
    kernels[k] = {"FETCH_SIZE_KB_mean": round(f.get(k, 0.0), 1), "WRITE_SIZE_KB_mean": round(w.get(k, 0.0), 1), "hbm_bytes_per_launch": b}
    total += b
nblk = (a.ctx + a.block_size - 1) // a.block_size
alg = a.batch * (2 * a.ctx * a.kv_heads * a.head_dim * 2 + 2 * a.heads * a.head_dim * 2 + 4 * nblk + 4)
rec = ({"command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace --output-format csv -- python3 tools/microbench.py decode "
                      f"--batch {a.batch} --ctx {a.ctx} --iters 3 --warmup 1 (one counter per pass)",
           "workload": {"batch": a.batch, "ctx": a.ctx, "heads": a.heads, "kv_heads": a.kv_heads, "head_dim": a.head_dim},
           "correction": "gfx950: FETCH_SIZE counts half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE exact; unit KB",
           "kernels": kernels, "hbm_bytes_per_attention_call": total, "algorithmic_bytes_per_attention_call": alg})
if a.append:
    import os
    doc = json.load(open(a.out)) if os.path.exists(a.out) else {"records": []}
    doc["records"] = [r for r in doc["records"] if r["workload"] != rec["workload"]] + [rec]
    json.dump(doc, open(a.out, "w"), indent=1)
else:
    json.dump(rec, open(a.out, "w"), indent=1)
print(json.dumps({"kernels": kernels, "hbm_bytes_per_attention_call": total, "algorithmic": alg}))
