#!/bin/bash
# inter-kernel gaps of the replayed decode step from a rocprofv3 kernel trace (start/end timestamps): inside a graph vs between two replays
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/gap
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gap/t -- python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline > gpurun_out/gap/log.txt 2>&1 || { tail -5 gpurun_out/gap/log.txt; exit 1; }
f=$(ls gpurun_out/gap/t/*/*kernel_trace.csv | head -1)
python3 - "$f" <<'PY'
import csv, sys, statistics as st
rows = sorted(({"n": r["Kernel_Name"], "s": int(r["Start_Timestamp"]), "e": int(r["End_Timestamp"])} for r in csv.DictReader(open(sys.argv[1]))), key=lambda r: r["s"])
# decode region: find argmax_candidates kernels (one per step)
idx = [i for i, r in enumerate(rows) if "argmax_candidates" in r["n"]]
idx = idx[-30:]
inter, intra, step = [], [], []
for a, b in zip(idx[:-1], idx[1:]):
    seg = rows[a:b + 1]
    inter.append((seg[1]["s"] - seg[0]["e"]) / 1e3)           # argmax end -> first kernel of the next step
    for x, y in zip(seg[1:-1], seg[2:]):
        intra.append((y["s"] - x["e"]) / 1e3)
    step.append((seg[-1]["e"] - seg[0]["e"]) / 1e3)
print("steps", len(step), "step us med", st.median(step))
print("gap between replays us: med", st.median(inter), "min", min(inter), "max", max(inter))
print("gap inside a replay us: med", st.median(intra), "mean", sum(intra) / len(intra), "n per step", len(intra) / len(step))
dur = {}
for a, b in zip(idx[:-1], idx[1:]):
    for r in rows[a + 1:b + 1]:
        k = r["n"][:70]; dur.setdefault(k, []).append((r["e"] - r["s"]) / 1e3)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1])):
    print(f"  {sum(v)/len(step):8.1f} us/step  avg {sum(v)/len(v):6.2f}  x{len(v)/len(step):5.1f}  {k}")
PY
rm -rf gpurun_out/gap/t
