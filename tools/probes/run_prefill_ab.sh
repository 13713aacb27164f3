#!/bin/bash
# same-box A/B of the prefill kernels: shipped library vs a variant built with EXTRA flags (default: -DNVH_PREFILL_NO_GROUP_SKIP)
cd $GRAFT_REPO_ROOT
EXTRA=${EXTRA:-=prefill_mfma.hip:-DNVH_CVT_ASM}
python nano-vllm-learn_amd/build.py --variant /tmp/libnvh_ab.so --extra $EXTRA > /tmp/ab_build.log 2>&1 || { tail -5 /tmp/ab_build.log; exit 1; }
for rep in 1 2; do
for s in 128 256 512 1024 2048 4096; do
  b=$((16384 / s > 256 ? 256 : 16384 / s))
  a=$(timeout -k 10 100 python3 tools/microbench.py prefill --batch $b --seq $s 2>>${NVH_PROBE_ERR:-gpurun_out/probe_stderr.log} | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['us_per_call'])")
  v=$(NVH_LIB_PATH=/tmp/libnvh_ab.so timeout -k 10 100 python3 tools/microbench.py prefill --batch $b --seq $s 2>>${NVH_PROBE_ERR:-gpurun_out/probe_stderr.log} | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['us_per_call'])")
  echo "S=$s shipped $a us   variant $v us"
done
a=$(timeout -k 10 100 python3 tools/microbench.py prefill --batch 4 --seq 4096 --heads 16 --kv-heads 8 --head-dim 128 2>>${NVH_PROBE_ERR:-gpurun_out/probe_stderr.log} | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['us_per_call'])")
v=$(NVH_LIB_PATH=/tmp/libnvh_ab.so timeout -k 10 100 python3 tools/microbench.py prefill --batch 4 --seq 4096 --heads 16 --kv-heads 8 --head-dim 128 2>>${NVH_PROBE_ERR:-gpurun_out/probe_stderr.log} | python3 -c "import sys,json; print(json.loads(sys.stdin.readline())['us_per_call'])")
echo "16/8/128 S=4096 shipped $a us   variant $v us"
done
