# SQ counters of the short-sequence prefill kernel at config 5's shape (B=128 per scheduler batch, S=128)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
sets=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY")
echo "# prefill_short_kernel<64,2,16>  B=128 S=128 H/KVH/D=14/2/64 (sums over the chip; SQ_*_CYCLES / ACTIVE / WAIT in quad-cycles, MFMA_BUSY in cycles)"
i=0
for set in "${sets[@]}"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcs_$i -- python3 tools/microbench.py prefill --batch 128 --seq 128 --iters 3 --warmup 1 > gpurun_out/pmcs_$i.log 2>&1
  f=$(ls gpurun_out/pmcs_$i/*/*counter_collection.csv 2>>${NVH_PROBE_ERR:-gpurun_out/probe_stderr.log} | head -1)
  if [ -n "$f" ]; then python3 tools/pmc_kernel.py $f prefill_short; else tail -3 gpurun_out/pmcs_$i.log; fi
  rm -rf gpurun_out/pmcs_$i
done
