#!/bin/bash
# prefill iteration: parity tests of the prefill kernels, then the S sweep
set -o pipefail
mkdir -p gpurun_out/r2p
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_hip_model_vs_oracle.py -m gpu -x -q -k "prefill or config5 or attention_module or model or decode_golden or randomised" > gpurun_out/r2p/pytest.log 2>&1
rc=$?
tail -6 gpurun_out/r2p/pytest.log
[ $rc -ne 0 ] && exit $rc
rm -f gpurun_out/r2p/micro.log
for s in 128 256 512 1024 2048 4096; do timeout -k 10 100 python3 tools/microbench.py prefill --batch $((16384 / s > 256 ? 256 : 16384 / s)) --seq $s >> gpurun_out/r2p/micro.log 2>&1 || exit 1; done
timeout -k 10 100 python3 tools/microbench.py prefill --batch 4 --seq 4096 --heads 16 --kv-heads 8 --head-dim 128 >> gpurun_out/r2p/micro.log 2>&1 || exit 1
timeout -k 10 100 python3 tools/microbench.py prefill --batch 128 --seq 128 --variant tiled >> gpurun_out/r2p/micro.log 2>&1 || exit 1
for ctx in 1025 1536 2048; do timeout -k 10 100 python3 tools/microbench.py decode --graph --width 16 --ctx $ctx >> gpurun_out/r2p/micro.log 2>&1 || exit 1; done
grep us_per_call gpurun_out/r2p/micro.log | cut -c1-200
