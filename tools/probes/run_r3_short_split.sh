#!/bin/bash
# short-sequence prefill kernel: 2 / 3 / 4 workgroups of 8 waves per (sequence, kv head) pair, with and without a register cap (variant builds), config 5's batch
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
{
for r in 1 2; do for lib in "" tools/probes/ab/short_s2.so tools/probes/ab/short_s3.so tools/probes/ab/short_s3w5.so tools/probes/ab/short_s3w6.so tools/probes/ab/short_s4w6.so; do
  [ -n "$lib" ] && export NVH_LIB_PATH=$lib || unset NVH_LIB_PATH
  echo "# library: ${lib:-this build}"
  for pv in exact auto; do timeout -k 10 100 python3 tools/microbench.py prefill --graph --batch 128 --seq 128 --pv $pv 2>>$O/probe_stderr.log || exit 1; done
done; done
} | grep -v amdgpu.ids | tee $O/r03_prefill_short_split_ab.txt
