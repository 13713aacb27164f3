# Round artifacts: default bench line, rocprofv3 kernel stats of the same command, decode / prefill sweeps.
R=${R:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 500 python3 bench.py > gpurun_out/bench_default.log 2>&1; tail -1 gpurun_out/bench_default.log > gpurun_out/${R}_bench_default_line.json
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_def -- python3 bench.py --no-cpu-baseline > gpurun_out/bench_prof.log 2>&1
f=$(ls gpurun_out/prof_def/*/*kernel_stats.csv | head -1); cp $f gpurun_out/${R}_bench_default_kernel_stats.csv; python3 tools/summarize_rocprof.py $f 24 > gpurun_out/${R}_bench_default_kernel_stats.txt; rm -rf gpurun_out/prof_def
{
echo "# decode attention call (nvh_paged_decode), HIP-graph of 24 calls, block-table width 16 (graph-replay shape)"
for c in 1025 1536 2048; do timeout -k 10 100 python3 tools/microbench.py decode --graph --width 16 --ctx $c; done
echo "# config 3: B=64"
for c in 2049 3072 4096; do timeout -k 10 100 python3 tools/microbench.py decode --graph --width 16 --batch 64 --ctx $c; done
echo "# config 4 per-rank shape (Qwen2-7B tp=4: 7/1/128): the default (four waves, 64-token passes since round 3), then --waves 4 (four waves, 128-token passes: the round-2 default)"
timeout -k 10 100 python3 tools/microbench.py decode --graph --width 16 --heads 7 --kv-heads 1 --head-dim 128 --ctx 1536
timeout -k 10 100 python3 tools/microbench.py decode --graph --width 16 --heads 7 --kv-heads 1 --head-dim 128 --ctx 1536 --waves 4
echo "# Qwen2-7B tp=1 head shape (28/4/128)"
timeout -k 10 100 python3 tools/microbench.py decode --graph --width 16 --heads 28 --kv-heads 4 --head-dim 128 --ctx 1536
echo "# Qwen3-0.6B head shape (16/8/128): 8 kv heads -> 8x the K/V bytes per layer"
for c in 1025 1536 2048; do timeout -k 10 100 python3 tools/microbench.py decode --graph --width 16 --heads 16 --kv-heads 8 --head-dim 128 --ctx $c; done
echo "# the older decode formulations at config 2 (ctx 1536): split MFMA + combine, VALU + wavefront reductions + combine (two launches each)"
timeout -k 10 100 python3 tools/microbench.py decode --graph --width 16 --ctx 1536 --variant split_mfma
timeout -k 10 100 python3 tools/microbench.py decode --graph --width 16 --ctx 1536 --variant split_valu
echo "# prefill S sweep (config 5 family), Qwen2-0.5B heads"
for s in 128 256 512 1024 2048 4096; do timeout -k 10 100 python3 tools/microbench.py prefill --batch $((16384 / s > 256 ? 256 : 16384 / s)) --seq $s; done
echo "# prefill, Qwen3-0.6B heads (16/8/128)"
timeout -k 10 100 python3 tools/microbench.py prefill --batch 4 --seq 4096 --heads 16 --kv-heads 8 --head-dim 128
} > gpurun_out/${R}_attention_microbench.txt 2>&1
cat gpurun_out/${R}_bench_default_kernel_stats.txt | head -14 | cut -c1-140; cat gpurun_out/${R}_attention_microbench.txt | cut -c1-230
