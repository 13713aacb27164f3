# prefill attention: default library vs tools/probes/ab/*.so, S sweep
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for v in default $(ls tools/probes/ab/*.so 2>/dev/null); do
  if [ $v = default ]; then unset NVH_LIB_PATH; else export NVH_LIB_PATH=$GRAFT_REPO_ROOT/$v; fi
  echo "== $v"
  for s in 512 1024 4096; do timeout -k 10 100 python3 tools/microbench.py prefill --batch $((16384 / s)) --seq $s 2>/dev/null | grep TFLOPs | cut -c1-140; done
  timeout -k 10 100 python3 tools/microbench.py prefill --batch 4 --seq 4096 --heads 16 --kv-heads 8 --head-dim 128 2>/dev/null | grep TFLOPs | cut -c1-140
done
done
