#!/bin/bash
# Infinity-Cache experiment: the o_proj / gate_up launches of layer i touch layer i + 1's K / V regions (bench.py --kv-ahead), decode step over the full window
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
export NVH_LIB_PATH=tools/probes/ab/pf64.so
for r in 1 2; do for m in none o_k gu_k o_k+gu_v o_v+gu_k; do
  flag=""; [ $m != none ] && flag="--kv-ahead $m"
  timeout -k 10 300 python3 bench.py --steps 1024 --warmup 16 --no-cpu-baseline --no-sweep --no-full-window $flag > $O/kvahead_$m.log 2>$O/kvahead_$m.err || { tail -3 $O/kvahead_$m.err; exit 1; }
  tail -1 $O/kvahead_$m.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', d['value'], d['ms_per_step'], 'attn us', d['roofline']['us_per_launch'])"
done; done | tee $O/r03_kv_ahead_ab.txt
