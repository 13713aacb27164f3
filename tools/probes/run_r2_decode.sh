#!/bin/bash
# decode-kernel iteration: decode parity tests, then the microbench points VERDICT names
set -o pipefail
mkdir -p gpurun_out/r2b
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_hip_model_vs_oracle.py tests/test_hip_layer_ops.py -m gpu -x -q -k "decode or handoff or config3 or config4 or attention_module or model or engine or fused_decode" > gpurun_out/r2b/pytest.log 2>&1
rc=$?
tail -12 gpurun_out/r2b/pytest.log
[ $rc -ne 0 ] && exit $rc
rm -f gpurun_out/r2b/micro.log
mb() { timeout -k 10 120 python tools/microbench.py decode --graph "$@" >> gpurun_out/r2b/micro.log 2>&1 || exit 1; }
for ctx in 1025 1536 2048; do mb --batch 32 --ctx $ctx --width 16; done
mb --batch 64 --ctx 3072 --width 16
for w in 4 8; do
  echo "# waves $w" >> gpurun_out/r2b/micro.log
  mb --batch 32 --ctx 1536 --heads 7 --kv-heads 1 --head-dim 128 --waves $w
  mb --batch 32 --ctx 1536 --heads 16 --kv-heads 8 --head-dim 128 --waves $w
  mb --batch 32 --ctx 1536 --heads 28 --kv-heads 4 --head-dim 128 --waves $w
done
grep "us_per_call\|# waves" gpurun_out/r2b/micro.log | cut -c1-200
