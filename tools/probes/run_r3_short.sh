#!/bin/bash
set -o pipefail
O=gpurun_out/r3_short; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_hip_parity.py tests/test_hip_model_vs_oracle.py -x -q -k "prefill or model or short" > $O/pytest.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit $rc
{
for v in short short_hilo; do for w in 16 8; do
  echo "$v waves $w S=128 B=128: $(timeout -k 10 100 python3 tools/microbench.py prefill --batch 128 --seq 128 --variant $v --waves $w 2>>$O/err.log | cut -c40-140)"
  echo "$v waves $w S=64 B=256: $(timeout -k 10 100 python3 tools/microbench.py prefill --batch 256 --seq 64 --variant $v --waves $w 2>>$O/err.log | cut -c40-140)"
done; done
echo "short 7/1/128 S=128 B=256: $(timeout -k 10 100 python3 tools/microbench.py prefill --batch 256 --seq 128 --heads 7 --kv-heads 1 --head-dim 128 --variant short 2>>$O/err.log | cut -c40-140)"
echo "short_hilo 7/1/128 S=128 B=256: $(timeout -k 10 100 python3 tools/microbench.py prefill --batch 256 --seq 128 --heads 7 --kv-heads 1 --head-dim 128 --variant short_hilo 2>>$O/err.log | cut -c40-140)"
} | tee $O/short_f16.txt
