#!/bin/bash
# the driver's bench invocation (steps 20) and the default one, plus PMC traffic at the contexts they run
set -o pipefail
mkdir -p gpurun_out/r2c
bash tools/probes/run_pmc_traffic.sh || exit 1
cp gpurun_out/r02_pmc_decode_traffic.json profiles/ 2>>${NVH_PROBE_ERR:-gpurun_out/probe_stderr.log}
timeout -k 10 500 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r2c/bench_steps20.json 2> gpurun_out/r2c/bench_steps20.err || { tail -20 gpurun_out/r2c/bench_steps20.err; exit 1; }
tail -c 4000 gpurun_out/r2c/bench_steps20.json
