#!/bin/bash
# decode step with the one-launch front under different consumer start delays (A/B libraries in tools/probes/ab/), and the two-launch form
set -o pipefail
O=gpurun_out/r3_fused; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_hip_qkv_attend.py -x -q 2>&1 | tail -2
for round in 1 2; do
  unset NVH_LIB_PATH
  echo "two launches: $(timeout -k 10 300 python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline --two-launches 2>>${NVH_PROBE_ERR:-gpurun_out/probe_stderr.log} | tail -1 | cut -c95-160)"
  echo "one launch, delay 0: $(timeout -k 10 300 python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline 2>>${NVH_PROBE_ERR:-gpurun_out/probe_stderr.log} | tail -1 | cut -c95-160)"
  for v in $(ls tools/probes/ab/delay*.so); do
    export NVH_LIB_PATH=$GRAFT_REPO_ROOT/$v
    echo "$v: $(timeout -k 10 300 python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline 2>>${NVH_PROBE_ERR:-gpurun_out/probe_stderr.log} | tail -1 | cut -c95-160)"
  done
done | tee $O/delay_ab.txt
