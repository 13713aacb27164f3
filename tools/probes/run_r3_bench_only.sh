#!/bin/bash
# the default bench line only (no profiler), into gpurun_out/r03_bench_default_line.json
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 bench.py > gpurun_out/bench_default.log 2>gpurun_out/bench_default.err || { tail -5 gpurun_out/bench_default.err; exit 1; }
tail -1 gpurun_out/bench_default.log > gpurun_out/r03_bench_default_line.json
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_bench_default_line.json').read())
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['us_per_launch'])
for k,v in d['prefill'].items(): print(k, {kk:v[kk] for kk in ('us_per_launch','us_per_launch_inputs_in_infinity_cache','achieved_TFLOPs','achieved_GBps','frac')})
PY
