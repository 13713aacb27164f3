#!/bin/bash
# stamped timelines of the decode attention call at 7/1/128 against 14/2/64 (same bytes), pass phases and hand-off tail
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
{
for mode in "" "--tail"; do
  for shape in "--heads 14 --kv-heads 2 --head-dim 64" "--heads 7 --kv-heads 1 --head-dim 128"; do
    echo "== stamp_decode.py $mode $shape"
    timeout -k 10 300 python3 tools/probes/stamp_decode.py $mode $shape --ctx 1536 2>>$O/probe_stderr.log || exit 1
  done
done
} | grep -v amdgpu.ids | tee $O/r03_decode_d128_stamps.txt
