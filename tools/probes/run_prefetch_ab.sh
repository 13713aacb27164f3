#!/bin/bash
# weight prefetch of the next small projection on the idle CUs of the qkv / down launches: tests, then same-box A/B of the decode step
set -o pipefail
mkdir -p gpurun_out/pfab
timeout -k 10 900 python -m pytest tests/test_hip_layer_ops.py tests/test_hip_model_vs_oracle.py tests/test_hip_advance.py tests/test_hip_allreduce.py -m gpu -x -q > gpurun_out/pfab/pytest.log 2>&1
rc=$?
tail -5 gpurun_out/pfab/pytest.log
[ $rc -ne 0 ] && exit $rc
rm -f gpurun_out/pfab/bench.log
for rep in 1 2 3; do
for flag in "" "--no-prefetch"; do
  echo "# ${flag:-prefetch} rep $rep" >> gpurun_out/pfab/bench.log
  timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline $flag 2>>${NVH_PROBE_ERR:-gpurun_out/probe_stderr.log} | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], d['roofline']['us_per_launch'])" >> gpurun_out/pfab/bench.log || exit 1
done
done
paste - - < gpurun_out/pfab/bench.log
