# SQ counters of the prefill kernel (default build and any variant in tools/probes/ab), S=1024, B=16
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in default $(ls tools/probes/ab/*.so 2>/dev/null); do
  if [ $v = default ]; then unset NVH_LIB_PATH; else export NVH_LIB_PATH=$GRAFT_REPO_ROOT/$v; fi
  echo "=== $v"
  i=0
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM SQ_INSTS_SMEM"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcp_$i -- python3 tools/microbench.py prefill --batch 16 --seq 1024 --iters 3 --warmup 1 > gpurun_out/pmcp_$i.log 2>&1
    f=$(ls gpurun_out/pmcp_$i/*/*counter_collection.csv 2>/dev/null | head -1)
    if [ -n "$f" ]; then python3 tools/pmc_kernel.py $f prefill_varlen; else tail -3 gpurun_out/pmcp_$i.log; fi
    rm -rf gpurun_out/pmcp_$i
  done
done
