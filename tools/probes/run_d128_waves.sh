#!/bin/bash
# D = 128: 4 against 8 waves per workgroup after the hand-off changes
set -o pipefail
mkdir -p gpurun_out/d128
rm -f gpurun_out/d128/micro.log
mb() { echo "# $*" >> gpurun_out/d128/micro.log; timeout -k 10 120 python tools/microbench.py decode --graph --width 16 "$@" >> gpurun_out/d128/micro.log 2>&1 || exit 1; }
for rep in 1 2; do
for w in 4 8; do
for ctx in 1034 1536 2048; do
mb --heads 7 --kv-heads 1 --head-dim 128 --ctx $ctx --waves $w
done
mb --heads 28 --kv-heads 4 --head-dim 128 --ctx 1536 --waves $w
mb --heads 16 --kv-heads 8 --head-dim 128 --ctx 1536 --waves $w
mb --heads 14 --kv-heads 2 --head-dim 64 --ctx 1034 --waves $w
mb --heads 14 --kv-heads 2 --head-dim 64 --ctx 1536 --waves $w
mb --batch 64 --heads 7 --kv-heads 1 --head-dim 128 --ctx 3072 --waves $w
done
done
grep "us_per_call\|^#" gpurun_out/d128/micro.log | cut -c1-150
