cd $GRAFT_REPO_ROOT
for v in default tools/probes/ab/libnvh_qt2.so; do
  if [ $v = default ]; then unset NVH_LIB_PATH; else export NVH_LIB_PATH=$GRAFT_REPO_ROOT/$v; fi
  for s in 128 1024 4096; do echo "$v S=$s: $(timeout -k 10 60 python tools/microbench.py prefill --batch $((16384 / s)) --seq $s 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['us_per_call'], d['TFLOPs'])")"; done
  echo "$v 16/8/128 S=4096: $(timeout -k 10 60 python tools/microbench.py prefill --batch 4 --seq 4096 --heads 16 --kv-heads 8 --head-dim 128 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['us_per_call'], d['TFLOPs'])")"
done
unset NVH_LIB_PATH; timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "prefill or varlen" > /dev/null 2>&1; echo "default prefill tests rc=$?"
export NVH_LIB_PATH=$GRAFT_REPO_ROOT/tools/probes/ab/libnvh_qt2.so; timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "prefill or varlen" > gpurun_out/pfv.log 2>&1; echo "vgpr-form prefill tests rc=$?"; tail -1 gpurun_out/pfv.log
