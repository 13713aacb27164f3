#!/bin/bash
# usage: run_r3_pf_ab.sh <variant .so>: prefill legs (cold / warm) of the default library against a variant build, two alternating rounds
set -o pipefail
O=gpurun_out/r3_prefill; mkdir -p $O
cat > /tmp/pf.py <<'PY'
import sys, os, json, torch
sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "nano-vllm-learn_amd")); sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
from nanovllm_hip.models.qwen import model_config
cfg = model_config("Qwen2-0.5B")
for b, s in ((64, 256), (32, 512), (16, 1024), (8, 2048), (4, 4096)):
    r = bench.prefill_leg(cfg, 1, b, s, buffers=8, iters=4)
    print(b, s, "cold", r["us_per_launch"], r["achieved_TFLOPs"], "warm", r["us_per_launch_inputs_in_infinity_cache"])
PY
for round in 1 2; do
  for lib in default $1; do
    if [ $lib = default ]; then unset NVH_LIB_PATH; else export NVH_LIB_PATH=$GRAFT_REPO_ROOT/$lib; fi
    echo "== $lib"; timeout -k 10 300 python3 /tmp/pf.py 2>>$O/err.log
  done
done | tee $O/ab_$(basename $1 .so).txt
