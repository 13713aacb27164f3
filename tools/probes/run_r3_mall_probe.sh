#!/bin/bash
# does the decode attention call run faster when its K/V is resident in the 256 MiB Infinity Cache?  graph of 24 calls over 24 distinct caches (24 x 25 MB: HBM)
# against the same graph cycling over few caches (1, 2, 4, 8 x 25 MB: re-read from the Infinity Cache)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
{
for r in 1 2; do for ctx in 1034 1536 2048; do for l in 24 8 4 2 1; do
  timeout -k 10 100 python3 tools/microbench.py decode --graph --width 16 --ctx $ctx --layers 24 --caches $l --iters 40 2>>$O/probe_stderr.log | sed "s/^{/{\"caches\": $l, /" || exit 1
done; done; done
} | grep -v amdgpu.ids | tee $O/r03_decode_mall_probe.txt
