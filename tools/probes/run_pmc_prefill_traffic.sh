#!/bin/bash
# L2 -> memory-side traffic of the prefill attention call from the PMC counters (one rocprofv3 --pmc pass per counter), against its algorithmic bytes
# (q in, o out, k and v in: tokens x (2H + 2KVH) x D x 2).  gfx950: FETCH_SIZE doubled (MI355X_MICROARCH.md, HBM section); unit KB.
R=${R:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
run() {  # $1 label, $2 algorithmic bytes, rest = microbench args
  label=$1; alg=$2; shift 2
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmcp_$c -- python3 tools/microbench.py "$@" --iters 3 --warmup 1 > $O/pmcp_$c.log 2>&1 || { tail -3 $O/pmcp_$c.log; exit 1; }
    f=$(ls $O/pmcp_$c/*/*counter_collection.csv | head -1); cp $f $O/pmcp_$c.csv; rm -rf $O/pmcp_$c
  done
  python3 - "$label" "$alg" $O/pmcp_FETCH_SIZE.csv $O/pmcp_WRITE_SIZE.csv <<'PY'
import csv, statistics, sys, re
label, alg = sys.argv[1], int(sys.argv[2])
def per_kernel(path, counter):
    vals = {}
    for r in csv.DictReader(open(path)):
        m = re.search(r"(prefill_\w+|bf16_rows_to_f16\w*)", r["Kernel_Name"])
        if m and r["Counter_Name"] == counter:
            vals.setdefault(m.group(1), []).append(float(r["Counter_Value"]))
    return {k: statistics.mean(v[len(v) // 4:]) for k, v in vals.items()}
f, w = per_kernel(sys.argv[3], "FETCH_SIZE"), per_kernel(sys.argv[4], "WRITE_SIZE")
tot = 0
print(f"# {label}: algorithmic {alg / 1e6:.1f} MB")
for k in sorted(set(f) | set(w)):
    b = (2 * f.get(k, 0) + w.get(k, 0)) * 1024
    tot += b
    print(f"  {k:28s} fetch {2 * f.get(k, 0) * 1024 / 1e6:8.2f} MB  write {w.get(k, 0) * 1024 / 1e6:8.2f} MB")
print(f"  total {tot / 1e6:.2f} MB = {tot / alg:.3f} x algorithmic")
PY
}
{
run "S=1024 x 16, 14/2/64, P as bf16 hi + lo" $((16384 * 32 * 64 * 2)) prefill --batch 16 --seq 1024 --pv exact
run "S=1024 x 16, 14/2/64, fp16 P V (conversion + attention)" $((16384 * 32 * 64 * 2)) prefill --batch 16 --seq 1024 --pv fp16
run "S=4096 x 4, 14/2/64, fp16 P V" $((16384 * 32 * 64 * 2)) prefill --batch 4 --seq 4096 --pv fp16
run "S=4096 x 4, 16/8/128, fp16 P V" $((16384 * 48 * 128 * 2)) prefill --batch 4 --seq 4096 --heads 16 --kv-heads 8 --head-dim 128 --pv fp16
run "config 5 half batch: S=128 x 128, 14/2/64 (short-sequence kernel, default form)" $((16384 * 32 * 64 * 2)) prefill --batch 128 --seq 128
run "S=256 x 64, 14/2/64 (tiled kernel, exact)" $((16384 * 32 * 64 * 2)) prefill --batch 64 --seq 256
} | tee $O/${R}_pmc_prefill_traffic.txt
