#!/bin/bash
# kernel durations of the guarded fp16 prefill call against the fp16 kernel alone (rocprofv3 --kernel-trace --stats), S = 1024 / 4096
set -o pipefail
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/pv16prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for s in 1024 4096; do
  for mode in "--pv fp16" "--variant tiled_f16v" "--pv exact"; do
    tag=$(echo "$mode" | tr -d ' -')_s$s
    timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$tag -- python3 $GRAFT_REPO_ROOT/tools/microbench.py prefill --batch $((16384 / s)) --seq $s $mode > $O/$tag.log 2>&1 || { tail -5 $O/$tag.log; exit 1; }
    echo "== $tag"; grep -h "us_per_call" $O/$tag.log | cut -c1-160
    f=$(find $O/$tag -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/summarize_rocprof.py $f 4 </dev/null
  done
done
