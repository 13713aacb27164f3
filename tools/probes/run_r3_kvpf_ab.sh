#!/bin/bash
# decode step: qkv + attention as two launches, with / without the K/V first-image prefetch on the qkv launch's idle CUs, and as one launch
set -o pipefail
O=gpurun_out/r3_fused; mkdir -p $O
line() { timeout -k 10 300 python3 bench.py --steps ${STEPS:-200} --warmup 5 --no-cpu-baseline "$@" 2>>${NVH_PROBE_ERR:-gpurun_out/probe_stderr.log} | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; }
for round in 1 2 3; do
  echo "two_launches:            $(line --qkv-attend two_launches)"
  echo "kv_prefetch 1 pass:      $(line --qkv-attend two_launches_kv_prefetch --kv-prefetch-passes 1)"
  echo "kv_prefetch 2 passes:    $(line --qkv-attend two_launches_kv_prefetch --kv-prefetch-passes 2)"
  echo "kv_prefetch 8 passes:    $(line --qkv-attend two_launches_kv_prefetch --kv-prefetch-passes 8)"
  echo "one_launch:              $(line --qkv-attend one_launch)"
done | tee $O/kvpf_ab.txt
