#!/bin/bash
# decode-kernel iteration: decode parity tests, then the microbench points VERDICT names
set -o pipefail
mkdir -p gpurun_out/r2b
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_hip_model_vs_oracle.py tests/test_hip_layer_ops.py -m gpu -x -q -k "decode or handoff or config3 or config4 or attention_module or model or engine or fused_decode" > gpurun_out/r2b/pytest.log 2>&1
rc=$?
tail -12 gpurun_out/r2b/pytest.log
[ $rc -ne 0 ] && exit $rc
rm -f gpurun_out/r2b/micro.log
for ctx in 1025 1536 2048; do
  timeout -k 10 120 python tools/microbench.py decode --batch 32 --ctx $ctx --graph --width 16 >> gpurun_out/r2b/micro.log 2>&1 || exit 1
done
timeout -k 10 120 python tools/microbench.py decode --batch 32 --ctx 1536 --graph --heads 7 --kv-heads 1 --head-dim 128 >> gpurun_out/r2b/micro.log 2>&1 || exit 1
timeout -k 10 120 python tools/microbench.py decode --batch 64 --ctx 3072 --graph --width 16 >> gpurun_out/r2b/micro.log 2>&1 || exit 1
timeout -k 10 120 python tools/microbench.py decode --batch 32 --ctx 1536 --graph --heads 16 --kv-heads 8 --head-dim 128 >> gpurun_out/r2b/micro.log 2>&1 || exit 1
grep us_per_call gpurun_out/r2b/micro.log
