#!/bin/bash
# prefill workgroup order on an XCD: all of its (sequence, kv head) pairs per q-tile rank (shipped) against one pair after the other (BASE-less A/B: NVH_LIB_PATH variant)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
{
for r in 1 2; do for lib in "" tools/probes/ab/pairmajor.so; do
  [ -n "$lib" ] && export NVH_LIB_PATH=$lib || unset NVH_LIB_PATH
  echo "# library: ${lib:-this build}"
  for s in 1024 4096; do timeout -k 10 100 python3 tools/microbench.py prefill --batch $((16384 / s)) --seq $s --heads 16 --kv-heads 8 --head-dim 128 2>>$O/probe_stderr.log || exit 1; done
  timeout -k 10 100 python3 tools/microbench.py prefill --batch 2 --seq 8192 --heads 16 --kv-heads 8 --head-dim 128 2>>$O/probe_stderr.log || exit 1
  timeout -k 10 100 python3 tools/microbench.py prefill --batch 4 --seq 4096 --heads 28 --kv-heads 4 --head-dim 128 2>>$O/probe_stderr.log || exit 1
  for s in 1024 4096; do timeout -k 10 100 python3 tools/microbench.py prefill --batch $((16384 / s)) --seq $s 2>>$O/probe_stderr.log || exit 1; done
done; done
} | grep -v amdgpu.ids | tee $O/r03_prefill_pairmajor_ab.txt
