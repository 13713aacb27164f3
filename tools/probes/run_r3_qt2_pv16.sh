#!/bin/bash
# fp16 P V form: from which length two query sub-tiles per wave pay (variant builds -DNVH_PREFILL_QT2_FROM_PV16=1024 / 512 against the shipped 2048)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
{
for r in 1 2; do for lib in "" tools/probes/ab/qt2_1024.so tools/probes/ab/qt2_512.so; do
  [ -n "$lib" ] && export NVH_LIB_PATH=$lib || unset NVH_LIB_PATH
  echo "# library: ${lib:-this build}"
  for s in 512 1024 1536; do timeout -k 10 100 python3 tools/microbench.py prefill --batch $((16384 / s)) --seq $s --pv fp16 2>>$O/probe_stderr.log || exit 1; done
done; done
} | grep -v amdgpu.ids | tee $O/r03_prefill_qt2_pv16_ab.txt
