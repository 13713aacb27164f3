#!/bin/bash
# the one-launch qkv + attention call: its own tests first, then every decode / layer / model test around it
set -o pipefail
O=gpurun_out/r3_fused
mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_hip_qkv_attend.py -x -q > $O/pytest_fused.log 2>&1; rc=$?; echo "fused tests rc=$rc"; tail -15 $O/pytest_fused.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_all.log 2>&1; rc=$?; echo "all gpu tests rc=$rc"; tail -5 $O/pytest_all.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3_fused/bench_steps20.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step')}, d['roofline']['us_per_launch'])
PY
