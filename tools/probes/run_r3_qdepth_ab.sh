#!/bin/bash
# short-sequence prefill kernel, Q two tasks ahead: prefill tests, then the bench's prefill legs (cold / Infinity-Cache-warm) with BASE and this build
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_hip_parity.py -x -q -k "prefill or config5" > $O/pytest_qd.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 $O/pytest_qd.log
[ $rc -eq 0 ] || exit 1
cat > /tmp/legs.py <<'PY'
import sys, json
sys.path.insert(0, '.'); sys.path.insert(0, 'nano-vllm-learn_amd')
import torch, bench
from nanovllm_hip.models.qwen import model_config
cfg = model_config("Qwen2-0.5B")
for b, s in ((128, 128), (128, 96), (256, 64)):
    r = bench.prefill_leg(cfg, 1, b, s)
    print(json.dumps({"batch": b, "seq": s, "cold_us": r["us_per_launch"], "warm_us": r["us_per_launch_inputs_in_infinity_cache"]}))
PY
for r in 1 2; do for lib in ${BASE:-} ""; do
  [ -n "$lib" ] && export NVH_LIB_PATH=$lib || unset NVH_LIB_PATH
  echo "# library: ${lib:-this build}"
  timeout -k 10 200 python3 /tmp/legs.py 2>>$O/probe_stderr.log || exit 1
done; done | grep -v amdgpu.ids | tee $O/r03_prefill_short_qdepth_ab.txt
