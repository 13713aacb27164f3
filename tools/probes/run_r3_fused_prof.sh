#!/bin/bash
# kernel durations of the decode step with the qkv + attention front as one launch and as two (rocprofv3 --kernel-trace --stats)
set -o pipefail
O=gpurun_out/r3_fused
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in one two; do
  flag=""; [ $v = two ] && flag="--two-launches"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline $flag > $O/bench_prof_$v.log 2>&1 || { tail -5 $O/bench_prof_$v.log; exit 1; }
  f=$(ls $O/prof_$v/*/*kernel_stats.csv | head -1); python3 tools/summarize_rocprof.py $f 10 > $O/kernel_stats_$v.txt; rm -rf $O/prof_$v
  echo "== $v"; cut -c1-150 $O/kernel_stats_$v.txt; tail -1 $O/bench_prof_$v.log | cut -c1-200
done
for v in one two one two; do
  flag=""; [ $v = two ] && flag="--two-launches"
  timeout -k 10 300 python3 bench.py --steps 200 --warmup 5 --no-cpu-baseline $flag 2>>${NVH_PROBE_ERR:-gpurun_out/probe_stderr.log} | tail -1 | cut -c1-190
done
