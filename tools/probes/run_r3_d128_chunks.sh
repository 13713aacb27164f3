#!/bin/bash
# 7/1/128 (Qwen2-7B per rank at tp=4), B=32: waves x chunks x pass size through the variant entry (is half the chip with twice the bytes in flight per CU and a 4-way hand-off faster?)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
{
for r in 1 2; do
  echo "# default"; timeout -k 10 100 python3 tools/microbench.py decode --graph --width 16 --ctx 1536 --heads 7 --kv-heads 1 --head-dim 128 2>>$O/probe_stderr.log || exit 1
  for v in chunked_p64 chunked_p128; do for w in 4 8; do for c in 4 6 8; do
    echo "# $v waves $w chunks $c"
    timeout -k 10 100 python3 tools/microbench.py decode --graph --width 16 --ctx 1536 --heads 7 --kv-heads 1 --head-dim 128 --variant $v --waves $w --chunks $c 2>>$O/probe_stderr.log || echo "(not available)"
  done; done; done
done
} | grep -v amdgpu.ids | tee $O/r03_decode_d128_chunks_ab.txt
