cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=qwen2_7b_shapes
timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --model Qwen2-7B --steps 64 --warmup 4 --no-cpu-baseline > gpurun_out/bench_$tag.log 2>&1
grep '^{"metric"' gpurun_out/bench_$tag.log | tail -1 > gpurun_out/r01_bench_${tag}_line.json
f=$(ls gpurun_out/prof_$tag/*/*kernel_stats.csv | head -1); python3 tools/summarize_rocprof.py $f 12 > gpurun_out/r01_bench_${tag}_kernel_stats.txt; rm -rf gpurun_out/prof_$tag
python3 -c "import json; d=json.load(open('gpurun_out/r01_bench_${tag}_line.json')); print('$tag', d['value'], d['ms_per_step'], d['roofline']['frac'], d['decode_step_roofline']['frac'])"
