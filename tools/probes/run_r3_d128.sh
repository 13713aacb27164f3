#!/bin/bash
set -o pipefail
O=gpurun_out/r3_d128; mkdir -p $O
{
for shape in "7 1" "28 4" "16 8"; do
  set -- $shape
  for v in chunked_p128 chunked_p64; do
    for ctx in 1034 1536 2048 3072; do
      echo "$v $(timeout -k 10 100 python3 tools/microbench.py decode --graph --width 16 --heads $1 --kv-heads $2 --head-dim 128 --ctx $ctx --variant $v 2>>${NVH_PROBE_ERR:-gpurun_out/probe_stderr.log} | cut -c40-200)"
    done
  done
done
for b in 8 16 64; do for v in chunked_p128 chunked_p64; do echo "B=$b $v $(timeout -k 10 100 python3 tools/microbench.py decode --graph --width 16 --batch $b --heads 7 --kv-heads 1 --head-dim 128 --ctx 1536 --variant $v 2>>${NVH_PROBE_ERR:-gpurun_out/probe_stderr.log} | cut -c40-200)"; done; done
} | tee $O/p64.txt
