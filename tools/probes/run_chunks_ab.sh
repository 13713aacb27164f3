#!/bin/bash
# chunk-count heuristic at batch sizes whose nearest-rounded chunk count overshoots the CU count
set -o pipefail
mkdir -p gpurun_out/chunks
rm -f gpurun_out/chunks/micro.log
mb() { echo "# $*" >> gpurun_out/chunks/micro.log; timeout -k 10 120 python tools/microbench.py decode --graph --width 16 "$@" >> gpurun_out/chunks/micro.log 2>&1 || exit 1; }
for ctx in 1100 1536; do
mb --batch 48 --ctx $ctx --chunks 2
mb --batch 48 --ctx $ctx --chunks 3
mb --batch 28 --ctx $ctx --chunks 4
mb --batch 28 --ctx $ctx --chunks 5
mb --batch 44 --ctx $ctx --chunks 2
mb --batch 44 --ctx $ctx --chunks 3
mb --batch 56 --ctx $ctx --chunks 2
mb --batch 56 --ctx $ctx --chunks 3
mb --batch 36 --ctx $ctx --chunks 3
mb --batch 36 --ctx $ctx --chunks 4
done
grep "us_per_call\|^#" gpurun_out/chunks/micro.log | cut -c1-150
