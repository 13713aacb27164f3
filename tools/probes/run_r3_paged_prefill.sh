#!/bin/bash
# prefix-cached / chunked prefill (K / V through the paged cache, nvh_prefill_varlen with a block table) against the packed-rows call of the same shape (exact form both)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
{
for s in 1024 4096; do
  echo "# $((16384 / s)) x $s tokens, 14/2/64: packed rows (exact), then the same through the cache, then a chunk of the last 512 / 1024 tokens against the cached prefix"
  timeout -k 10 100 python3 tools/microbench.py prefill --batch $((16384 / s)) --seq $s --pv exact 2>>$O/probe_stderr.log || exit 1
  timeout -k 10 100 python3 tools/microbench.py prefill --batch $((16384 / s)) --seq $s --paged 2>>$O/probe_stderr.log || exit 1
  timeout -k 10 100 python3 tools/microbench.py prefill --batch $((16384 / s)) --seq $s --paged --q-len 512 2>>$O/probe_stderr.log || exit 1
done
echo "# 4 x 4096 tokens, 16/8/128"
timeout -k 10 100 python3 tools/microbench.py prefill --batch 4 --seq 4096 --heads 16 --kv-heads 8 --head-dim 128 --pv exact 2>>$O/probe_stderr.log || exit 1
timeout -k 10 100 python3 tools/microbench.py prefill --batch 4 --seq 4096 --heads 16 --kv-heads 8 --head-dim 128 --paged 2>>$O/probe_stderr.log || exit 1
timeout -k 10 100 python3 tools/microbench.py prefill --batch 4 --seq 4096 --heads 16 --kv-heads 8 --head-dim 128 --paged --q-len 1024 2>>$O/probe_stderr.log || exit 1
} | grep -v amdgpu.ids | tee $O/r03_prefill_paged.txt
