#!/bin/bash
# causal mask applied only in the 16-key groups that reach past the sub-tile's first row (both prefill kernels): parity tests, then same-box A/B against BASE
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -k "prefill or config5 or module" > $O/pytest_mask.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/pytest_mask.log
[ $rc -eq 0 ] || exit 1
{
for r in 1 2; do for lib in ${BASE:-} ""; do
  [ -n "$lib" ] && export NVH_LIB_PATH=$lib || unset NVH_LIB_PATH
  echo "# library: ${lib:-this build}"
  timeout -k 10 100 python3 tools/microbench.py prefill --graph --batch 128 --seq 128 2>>$O/probe_stderr.log || exit 1
  for s in 256 512 1024 2048; do timeout -k 10 100 python3 tools/microbench.py prefill --batch $((16384 / s)) --seq $s 2>>$O/probe_stderr.log || exit 1; done
  timeout -k 10 100 python3 tools/microbench.py prefill --batch 16 --seq 1024 --pv exact 2>>$O/probe_stderr.log || exit 1
  timeout -k 10 100 python3 tools/microbench.py prefill --batch 16 --seq 1024 --heads 16 --kv-heads 8 --head-dim 128 2>>$O/probe_stderr.log || exit 1
done; done
} | grep -v amdgpu.ids | tee $O/r03_prefill_mask_ab.txt
