#!/bin/bash
O=gpurun_out/r3_d128; mkdir -p $O
NVH_LIB_PATH=$GRAFT_REPO_ROOT/tools/probes/ab/pinloads.so timeout -k 10 600 python -m pytest tests/test_hip_parity.py -x -q -k "decode or handoff" 2>&1 | tail -2
for round in 1 2; do
for lib in default tools/probes/ab/pinloads.so; do
  if [ $lib = default ]; then unset NVH_LIB_PATH; else export NVH_LIB_PATH=$GRAFT_REPO_ROOT/$lib; fi
  echo "== $lib"
  for ctx in 1034 1536 2048; do echo "7/1/128 $(timeout -k 10 100 python3 tools/microbench.py decode --graph --width 16 --heads 7 --kv-heads 1 --head-dim 128 --ctx $ctx 2>>$O/err.log | cut -c50-150)"; done
  for b in 8 16; do echo "14/2/64 B=$b $(timeout -k 10 100 python3 tools/microbench.py decode --graph --width 16 --batch $b --ctx 1536 2>>$O/err.log | cut -c50-150)"; done
  echo "14/2/64 B=32 $(timeout -k 10 100 python3 tools/microbench.py decode --graph --width 16 --ctx 1536 2>>$O/err.log | cut -c50-150)"
done
done | tee $O/pin_ab.txt
