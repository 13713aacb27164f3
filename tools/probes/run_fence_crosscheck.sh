#!/bin/bash
# Cross-check of the fence-free chunk hand-off: build the library with -DNVH_HANDOFF_FENCES (plain stores, agent-scope release fence,
# ticket, agent-scope acquire fence, plain loads: the HIP memory model's textbook form) and run the decode parity tests against it,
# then compare its results with the shipped library's BIT FOR BIT on a set of geometries, and time both.
set -o pipefail
cd $GRAFT_REPO_ROOT
python nano-vllm-learn_amd/build.py --variant /tmp/libnvh_fences.so --extra =paged_decode.hip:-DNVH_HANDOFF_FENCES > /tmp/fence_build.log 2>&1 || { tail -5 /tmp/fence_build.log; exit 1; }
echo "== parity tests against the fenced build"
NVH_LIB_PATH=/tmp/libnvh_fences.so timeout -k 10 600 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "decode or handoff or config3 or config4" 2>&1 | tail -3 || exit 1
echo "== bitwise comparison, fenced vs shipped"
timeout -k 10 300 python tools/probes/fence_compare.py /tmp/libnvh_fences.so || exit 1
echo "== timing (us per call, ctx 1536): shipped, then fenced"
timeout -k 10 100 python tools/microbench.py decode --graph --width 16 --ctx 1536 2>&1 | grep us_per_call | cut -c1-160
NVH_LIB_PATH=/tmp/libnvh_fences.so timeout -k 10 100 python tools/microbench.py decode --graph --width 16 --ctx 1536 2>&1 | grep us_per_call | cut -c1-160
