# Bench lines (+ rocprofv3 kernel stats) for configurations other than the headline one; not bench lines of the contract.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # $1 = tag, rest = bench args
  tag=$1; shift
  timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py "$@" --no-cpu-baseline > gpurun_out/bench_$tag.log 2>&1
  grep '^{"metric"' gpurun_out/bench_$tag.log | tail -1 > gpurun_out/${R:-r02}_bench_${tag}_line.json
  f=$(ls gpurun_out/prof_$tag/*/*kernel_stats.csv | head -1); python3 tools/summarize_rocprof.py $f 12 > gpurun_out/${R:-r02}_bench_${tag}_kernel_stats.txt; rm -rf gpurun_out/prof_$tag
  python3 -c "import json; d=json.load(open('gpurun_out/${R:-r02}_bench_${tag}_line.json')); print('$tag', d['value'], d['ms_per_step'], d['roofline']['frac'], d['decode_step_roofline']['frac'])"
}
run config3_bs64_in2048 --batch 64 --input-len 2048 --steps 256 --warmup 8
run qwen3_0p6b --model Qwen3-0.6B --steps 256 --warmup 8
run qwen2_7b_shapes --model Qwen2-7B --steps 64 --warmup 4
