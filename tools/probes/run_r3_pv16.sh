#!/bin/bash
# nvh_prefill_varlen_pv16 (fp16 P V behind a range guard, the default from 1024 keys on): prefill parity tests, then the S sweep in both forms on one box.
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -k "prefill or module" > $O/pytest_pv16.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 $O/pytest_pv16.log
[ $rc -eq 0 ] || exit 1
{
echo "# prefill S sweep, Qwen2-0.5B heads: --pv exact (bf16 hi + lo) / --pv fp16 (guarded fp16 form, conversion inside the timed call), two rounds"
for r in 1 2; do for s in 256 512 1024 2048 4096; do for pv in exact fp16; do
  timeout -k 10 100 python3 tools/microbench.py prefill --batch $((16384 / s)) --seq $s --pv $pv 2>>$O/probe_stderr.log || exit 1
done; done; done
echo "# the fp16 kernel alone (--variant tiled_f16v: v converted outside the timed loop, no flags): what conversion + flags cost in the lines above"
for s in 1024 2048 4096; do timeout -k 10 100 python3 tools/microbench.py prefill --batch $((16384 / s)) --seq $s --variant tiled_f16v 2>>$O/probe_stderr.log || exit 1; done
echo "# Qwen3-0.6B heads (16/8/128)"
for s in 1024 4096; do for pv in exact fp16; do
  timeout -k 10 100 python3 tools/microbench.py prefill --batch $((16384 / s)) --seq $s --heads 16 --kv-heads 8 --head-dim 128 --pv $pv 2>>$O/probe_stderr.log || exit 1
done; done
echo "# Qwen2-7B heads at tp=1 (28/4/128)"
for pv in exact fp16; do timeout -k 10 100 python3 tools/microbench.py prefill --batch 4 --seq 4096 --heads 28 --kv-heads 4 --head-dim 128 --pv $pv 2>>$O/probe_stderr.log || exit 1; done
} | grep -v amdgpu.ids | tee $O/r03_prefill_pv16_sweep.txt
