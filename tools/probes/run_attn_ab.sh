# decode attention call: default library vs tools/probes/ab/*.so at three contexts, two rounds
cd $GRAFT_REPO_ROOT
for round in 1 2; do
for v in default $(ls tools/probes/ab/*.so 2>>${NVH_PROBE_ERR:-gpurun_out/probe_stderr.log}); do
  if [ $v = default ]; then unset NVH_LIB_PATH; else export NVH_LIB_PATH=$GRAFT_REPO_ROOT/$v; fi
  echo "== $v"
  for ctx in 1025 1280 1536 1792 2048; do timeout -k 10 200 python tools/microbench.py decode --batch 32 --ctx $ctx --graph --width 16 2>&1 | grep us_per_call | cut -c40-130; done
done
done
