#!/bin/bash
# round-3 starting point on today's box: GPU tests, the driver's bench command, microbench points of the verdict
set -o pipefail
O=gpurun_out/r3_base
mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee $O/pytest.rc
tail -3 $O/pytest.log
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench rc=$?"
tail -c 1500 $O/bench_steps20.json
for ctx in 1034 1536 2048; do timeout -k 10 200 python tools/microbench.py decode --batch 32 --ctx $ctx --graph --width 16 2>>$O/mb.err | tee -a $O/mb.txt; done
timeout -k 10 200 python tools/microbench.py decode --batch 32 --ctx 1536 --heads 7 --kv-heads 1 --head-dim 128 --graph --width 16 2>>$O/mb.err | tee -a $O/mb.txt
timeout -k 10 200 python tools/microbench.py prefill --batch 16 --seq 1024 2>>$O/mb.err | tee -a $O/mb.txt
timeout -k 10 200 python tools/microbench.py prefill --batch 4 --seq 4096 2>>$O/mb.err | tee -a $O/mb.txt
timeout -k 10 200 python tools/microbench.py prefill --batch 4 --seq 4096 --heads 16 --kv-heads 8 --head-dim 128 2>>$O/mb.err | tee -a $O/mb.txt
timeout -k 10 200 python tools/microbench.py prefill --batch 128 --seq 128 2>>$O/mb.err | tee -a $O/mb.txt
