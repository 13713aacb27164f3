#!/bin/bash
# per-wave timeline of the short-sequence prefill kernel at config 5's half batch (diagnostic -DNVH_STAMPS build)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
timeout -k 10 600 python3 tools/probes/stamp_decode.py --build-only 2>>$O/probe_stderr.log || exit 1
{
for w in 16 8; do echo "== $w waves"; timeout -k 10 120 python3 tools/probes/stamp_prefill_short.py --waves $w 2>>$O/probe_stderr.log || exit 1; done
} | tee $O/r03_prefill_short_stamps.txt
