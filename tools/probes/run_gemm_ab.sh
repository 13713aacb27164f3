#!/bin/bash
# streaming GEMM change: layer / model / engine tests, then same-box A/B of the decode step against $BASE
set -o pipefail
mkdir -p gpurun_out/gab
timeout -k 10 900 python -m pytest tests/test_hip_layer_ops.py tests/test_hip_model_vs_oracle.py tests/test_hip_advance.py tests/test_hip_rope_store.py -m gpu -x -q > gpurun_out/gab/pytest.log 2>&1
rc=$?
tail -4 gpurun_out/gab/pytest.log
[ $rc -ne 0 ] && exit $rc
rm -f gpurun_out/gab/bench.log
for rep in 1 2 3; do
for lib in "" ${BASE:-tools/probes/ab/base.so}; do
  export NVH_LIB_PATH=$lib; [ -z "$lib" ] && unset NVH_LIB_PATH
  echo "# lib ${lib:-new} rep $rep" >> gpurun_out/gab/bench.log
  timeout -k 10 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline 2>>${NVH_PROBE_ERR:-gpurun_out/probe_stderr.log} | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])" >> gpurun_out/gab/bench.log || exit 1
done
done
paste - - < gpurun_out/gab/bench.log
