#!/bin/bash
# decode steps per replayed graph: engine / model tests, then same-box A/B
set -o pipefail
mkdir -p gpurun_out/multi
timeout -k 10 900 python -m pytest tests/test_hip_model_vs_oracle.py tests/test_hip_advance.py tests/test_hip_layer_ops.py -m gpu -x -q > gpurun_out/multi/pytest.log 2>&1
rc=$?
tail -5 gpurun_out/multi/pytest.log
[ $rc -ne 0 ] && exit $rc
rm -f gpurun_out/multi/bench.log
for rep in 1 2 3; do
for n in 1 4 8 16; do
  echo "# graph-steps $n rep $rep" >> gpurun_out/multi/bench.log
  timeout -k 10 200 python bench.py --steps 208 --warmup 16 --no-cpu-baseline --graph-steps $n 2>>${NVH_PROBE_ERR:-gpurun_out/probe_stderr.log} | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> gpurun_out/multi/bench.log || exit 1
done
done
paste - - < gpurun_out/multi/bench.log
