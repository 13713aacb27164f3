#!/bin/bash
# round-3 artifacts: PMC traffic of the decode attention call, the driver's bench command and the default one, rocprofv3 kernel stats of the default
# run, the attention microbench sweeps, SQ counters of the attention kernels.  Everything lands in gpurun_out/ (copied into profiles/ afterwards).
set -o pipefail
export R=r03
cd $GRAFT_REPO_ROOT
bash tools/probes/run_pmc_traffic.sh || { echo "pmc traffic failed"; exit 1; }
cp gpurun_out/r03_pmc_decode_traffic.json profiles/ 2>/dev/null
echo "[progress] pmc traffic done"
timeout -k 10 500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bench_steps20.log 2>gpurun_out/bench_steps20.err || { tail -5 gpurun_out/bench_steps20.err; exit 1; }
tail -1 gpurun_out/bench_steps20.log > gpurun_out/r03_bench_steps20_line.json
echo "[progress] steps20 bench done"
bash tools/probes/run_artifacts.sh || exit 1
echo "[progress] default bench + kernel stats + microbench done"
bash tools/probes/run_pmc_final.sh > gpurun_out/r03_pmc_sq_counters_attention.txt 2>gpurun_out/pmc_final.err || { tail -5 gpurun_out/pmc_final.err; exit 1; }
head -25 gpurun_out/r03_pmc_sq_counters_attention.txt
