#!/bin/bash
# short-sequence prefill kernel (config 5): parity tests, then the call at S = 128 / 64 / 96, 16 and 8 waves; BASE=<saved build> adds the same lines for it
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -k "prefill or config5" > $O/pytest_short.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/pytest_short.log
[ $rc -eq 0 ] || exit 1
{
for lib in ${BASE:-} ""; do
  [ -n "$lib" ] && export NVH_LIB_PATH=$lib || unset NVH_LIB_PATH
  echo "# library: ${lib:-this build}"
  for r in 1 2; do
    for s in 128 64 96; do for pv in exact auto; do timeout -k 10 100 python3 tools/microbench.py prefill --graph --batch 128 --seq $s --pv $pv 2>>$O/probe_stderr.log || exit 1; done; done
    for pv in exact auto; do timeout -k 10 100 python3 tools/microbench.py prefill --batch 128 --seq 128 --pv $pv 2>>$O/probe_stderr.log || exit 1; done
    timeout -k 10 100 python3 tools/microbench.py prefill --batch 128 --seq 128 --variant short --waves 8 2>>$O/probe_stderr.log || exit 1
    timeout -k 10 100 python3 tools/microbench.py prefill --batch 128 --seq 128 --heads 16 --kv-heads 8 --head-dim 128 2>>$O/probe_stderr.log || exit 1
  done
done
} | grep -v amdgpu.ids | tee $O/r03_prefill_short_ab.txt
