#!/bin/bash
# final build: every GPU test, the smoke entry, and the N = 2 / 4 code path of bench.py rehearsed over gloo on the one GPU (not a measurement)
set -o pipefail
O=gpurun_out/r3_final; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; rc=$?; echo "gpu tests rc=$rc"; tail -3 $O/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
for n in 2 4; do
  NVH_BENCH_REHEARSE=gloo timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus $n --steps 32 --warmup 4 > $O/rehearse_$n.log 2> $O/rehearse_$n.err; echo "rehearsal N=$n rc=$?"
  tail -1 $O/rehearse_$n.log | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['n_gpus'], d['value'], d['config']['collective_detail'], d['full_window'])" || tail -5 $O/rehearse_$n.err
done
