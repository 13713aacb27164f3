#!/bin/bash
# prefill S sweep of the final build (XCD-aware order, Q loads behind the first DMA), default form and the opt-in fp16 P V form
O=gpurun_out; 
{
echo "# prefill S sweep, FINAL build of round 3 (XCD-aware workgroup order, Q loads behind the first tile's DMA), Qwen2-0.5B heads; tools/microbench.py (one input, re-used)"
for s in 128 256 512 1024 2048 4096; do timeout -k 10 100 python3 tools/microbench.py prefill --batch $((16384 / s > 256 ? 256 : 16384 / s)) --seq $s 2>>$O/probe_stderr.log; done
echo "# Qwen3-0.6B heads (16/8/128)"
timeout -k 10 100 python3 tools/microbench.py prefill --batch 4 --seq 4096 --heads 16 --kv-heads 8 --head-dim 128 2>>$O/probe_stderr.log
echo "# the opt-in fp16 P V form (--variant tiled_f16v; v converted outside the timed loop), same shapes"
for s in 1024 2048 4096; do timeout -k 10 100 python3 tools/microbench.py prefill --batch $((16384 / s)) --seq $s --variant tiled_f16v 2>>$O/probe_stderr.log; done
timeout -k 10 100 python3 tools/microbench.py prefill --batch 4 --seq 4096 --heads 16 --kv-heads 8 --head-dim 128 --variant tiled_f16v 2>>$O/probe_stderr.log
} | tee $O/r03_prefill_sweep_final.txt
