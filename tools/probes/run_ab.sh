# A/B of library variants on one box: default vs each tools/probes/ab/*.so (bench, 256 steps, two rounds)
cd $GRAFT_REPO_ROOT
for round in 1 2; do
  for v in default $(ls tools/probes/ab/*.so); do
    if [ $v = default ]; then unset NVH_LIB_PATH; else export NVH_LIB_PATH=$GRAFT_REPO_ROOT/$v; fi
    echo "$v: $(timeout -k 10 300 python bench.py --steps 256 --no-cpu-baseline 2>>${NVH_PROBE_ERR:-gpurun_out/probe_stderr.log} | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline']['us_per_launch'])")"
  done
done
