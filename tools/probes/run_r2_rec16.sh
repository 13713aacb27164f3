#!/bin/bash
# 16-byte record accesses in the decode hand-off: parity, then same-box A/B against the build before the change (${BASE:-tools/probes/ab/base.so})
set -o pipefail
mkdir -p gpurun_out/rec16
timeout -k 10 700 python -m pytest tests/test_hip_parity.py tests/test_hip_model_vs_oracle.py tests/test_hip_layer_ops.py -m gpu -x -q -k "decode or handoff or config3 or config4 or attention_module or model or engine or fused_decode" > gpurun_out/rec16/pytest.log 2>&1
rc=$?
tail -5 gpurun_out/rec16/pytest.log
[ $rc -ne 0 ] && exit $rc
rm -f gpurun_out/rec16/micro.log
mb() { timeout -k 10 120 python tools/microbench.py decode --graph "$@" >> gpurun_out/rec16/micro.log 2>&1 || exit 1; }
for rep in 1 2; do
for lib in "" "${BASE:-tools/probes/ab/base.so}"; do
  export NVH_LIB_PATH=$lib
  [ -z "$lib" ] && unset NVH_LIB_PATH
  echo "# lib ${lib:-new} rep $rep" >> gpurun_out/rec16/micro.log
  for ctx in 1034 1536 2048; do mb --batch 32 --ctx $ctx --width 16; done
  mb --batch 64 --ctx 3072 --width 16
  mb --batch 32 --ctx 1536 --heads 7 --kv-heads 1 --head-dim 128
  mb --batch 32 --ctx 1536 --heads 16 --kv-heads 8 --head-dim 128
done
done
grep "us_per_call\|# lib" gpurun_out/rec16/micro.log | cut -c1-160
