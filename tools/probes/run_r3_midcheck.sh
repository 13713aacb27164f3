#!/bin/bash
set -o pipefail
O=gpurun_out/r3_mid; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_allreduce.py tests/test_hip_qkv_attend.py -x -q > $O/pytest_ar.log 2>&1; rc=$?; echo "allreduce+fused tests rc=$rc"; tail -5 $O/pytest_ar.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench rc=$?"; tail -3 $O/bench_steps20.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_mid/bench_steps20.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','step_floor','full_window','attention_sweep'): print(k, d[k])
print(d['roofline']['us_per_launch'], d['prefill']['s1024']['achieved_TFLOPs'], d['prefill']['config5_half']['us_per_launch'])
PY
