#!/bin/bash
set -o pipefail
O=gpurun_out/r3_prefill; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -k "prefill" > $O/pytest_f16v.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/pytest_f16v.log
[ $rc -eq 0 ] || exit $rc
for round in 1 2; do
for v in tiled tiled_f16v; do
  for s in 512 1024 2048 4096; do echo "$v 14/2/64 $(timeout -k 10 100 python3 tools/microbench.py prefill --batch $((16384 / s)) --seq $s --variant $v 2>>$O/err.log | cut -c28-140)"; done
  for s in 1024 4096; do echo "$v 16/8/128 $(timeout -k 10 100 python3 tools/microbench.py prefill --batch $((16384 / s)) --seq $s --heads 16 --kv-heads 8 --head-dim 128 --variant $v 2>>$O/err.log | cut -c28-140)"; done
done
done | tee $O/f16v_ab.txt
