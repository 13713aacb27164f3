# SQ counters of the decode GEMM kernels inside the real decode step (bench, 48 steps, eager launches so that every launch is a
# dispatch the profiler sees); one rocprofv3 --pmc pass per counter set
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
sets=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU")
i=0
for set in "${sets[@]}"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcg_$i -- python3 bench.py --steps 48 --warmup 2 --eager --no-cpu-baseline > gpurun_out/pmcg_$i.log 2>&1
  f=$(ls gpurun_out/pmcg_$i/*/*counter_collection.csv 2>>${NVH_PROBE_ERR:-gpurun_out/probe_stderr.log} | head -1)
  if [ -n "$f" ]; then
    for k in "linear_stream_kernel<2, 2, 0, true, false, 2, false, 8>" "linear_stream_kernel<2, 2, 0, true, false, 3, false, 8>" "linear_stream_kernel<2, 1, 2, true, false, 2, true, 8>" "linear_stream_kernel<2, 3, 2, true, false, 2, false, 8>" "paged_decode_chunked_kernel"; do
      echo "# $k"; python3 tools/pmc_kernel.py $f "$k"
    done
  else tail -3 gpurun_out/pmcg_$i.log; fi
  rm -rf gpurun_out/pmcg_$i
done
