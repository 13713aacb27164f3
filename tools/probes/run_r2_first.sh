#!/bin/bash
# round 2, first GPU call: the whole GPU suite, then the decode microbench baseline (before any kernel change)
set -o pipefail
mkdir -p gpurun_out/r2a
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1
rc=$?
tail -15 gpurun_out/r2a/pytest.log
[ $rc -ne 0 ] && exit $rc
for ctx in 1025 1536 2048; do
  timeout -k 10 120 python tools/microbench.py decode --batch 32 --ctx $ctx --graph --width 16 >> gpurun_out/r2a/micro.log 2>&1 || exit 1
done
timeout -k 10 120 python tools/microbench.py decode --batch 32 --ctx 1536 --graph --heads 7 --kv-heads 1 --head-dim 128 >> gpurun_out/r2a/micro.log 2>&1 || exit 1
timeout -k 10 120 python tools/microbench.py decode --batch 64 --ctx 3072 --graph --width 16 >> gpurun_out/r2a/micro.log 2>&1 || exit 1
cat gpurun_out/r2a/micro.log
