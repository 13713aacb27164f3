#!/bin/bash
# usage: run_r3_stamps.sh "<extra hipcc flags>" [ctx ...]
set -o pipefail
mkdir -p gpurun_out/r3_fused
extra="$1"; shift
python3 tools/probes/stamp_qkv_attend.py --build-only --extra $extra 2>&1 | tail -3
for ctx in "$@"; do timeout -k 10 120 python3 tools/probes/stamp_qkv_attend.py --ctx $ctx 2>&1 | grep -v amdgpu.ids; done | tee -a gpurun_out/r3_fused/stamps.txt
