#!/bin/bash
# prefill attention: parity tests, then the S sweep with the software-pipelined tile loop and without it
set -o pipefail
O=gpurun_out/r3_prefill; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_hip_model_vs_oracle.py -x -q -k "prefill or model or chunked_prefill or short" > $O/pytest.log 2>&1; rc=$?; echo "prefill tests rc=$rc"; tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
{
for v in tiled tiled_unpipelined; do
  echo "# variant $v, Qwen2-0.5B heads 14/2/64"
  for s in 256 512 1024 2048 4096; do timeout -k 10 100 python3 tools/microbench.py prefill --batch $((16384 / s)) --seq $s --variant $v 2>>${NVH_PROBE_ERR:-gpurun_out/probe_stderr.log}; done
  echo "# variant $v, 16/8/128"
  for s in 1024 4096; do timeout -k 10 100 python3 tools/microbench.py prefill --batch $((16384 / s)) --seq $s --heads 16 --kv-heads 8 --head-dim 128 --variant $v 2>>${NVH_PROBE_ERR:-gpurun_out/probe_stderr.log}; done
done
} | tee $O/sweep.txt
