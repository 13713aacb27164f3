cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcd_$i -- python3 tools/microbench.py decode --batch 32 --ctx 1536 --width 16 --iters 3 --warmup 1 > gpurun_out/pmcd_$i.log 2>&1
  f=$(ls gpurun_out/pmcd_$i/*/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then python3 tools/pmc_kernel.py $f paged_decode_chunked; else tail -3 gpurun_out/pmcd_$i.log; fi
  rm -rf gpurun_out/pmcd_$i
done
