# SQ counters of the shipped prefill and decode attention kernels (one rocprofv3 --pmc pass per counter set)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
sets=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM")
run() {  # $1 = kernel substring, rest = program args
  k=$1; shift; i=0
  for set in "${sets[@]}"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d gpurun_out/pmcf_$i -- python3 tools/microbench.py "$@" --iters 3 --warmup 1 > gpurun_out/pmcf_$i.log 2>&1
    f=$(ls gpurun_out/pmcf_$i/*/*counter_collection.csv 2>>${NVH_PROBE_ERR:-gpurun_out/probe_stderr.log} | head -1)
    if [ -n "$f" ]; then python3 tools/pmc_kernel.py $f $k; else tail -3 gpurun_out/pmcf_$i.log; fi
    rm -rf gpurun_out/pmcf_$i
  done
}
echo "# prefill_varlen_kernel<64,false,1,0> (P as bf16 hi + lo: --pv exact)  B=16 S=1024 H/KVH/D=14/2/64 (sums over the chip; SQ_*_CYCLES / ACTIVE / WAIT in quad-cycles, MFMA_BUSY in cycles)"
run prefill_varlen prefill --batch 16 --seq 1024 --pv exact
echo "# prefill_varlen_kernel<64,false,2,2> (fp16 P V behind the range guard, two sub-tiles per wave: the default call)  B=16 S=1024 H/KVH/D=14/2/64"
run prefill_varlen prefill --batch 16 --seq 1024 --pv fp16
echo "# prefill_varlen_kernel<128,false,2,2> (fp16 P V, the default call) B=4 S=4096 H/KVH/D=16/8/128"
run prefill_varlen prefill --batch 4 --seq 4096 --heads 16 --kv-heads 8 --head-dim 128
echo "# paged_decode_chunked_kernel<64, 8, 128>  B=32 ctx=1536 width 16 (eager launches; 128-token passes: four chunks per pair)"
run paged_decode_chunked decode --batch 32 --ctx 1536 --width 16
echo "# paged_decode_chunked_kernel<128, 4, 64>  B=32 ctx=1536 H/KVH/D=7/1/128 (eager launches; four waves, 16-token wave tiles, 64-token passes)"
run paged_decode_chunked decode --batch 32 --ctx 1536 --heads 7 --kv-heads 1 --head-dim 128
echo "# prefill_short_kernel<64,2,16> (P as bf16 hi + lo: --pv exact)  B=128 S=128 H/KVH/D=14/2/64 (BASELINE config 5, one scheduler batch)"
run prefill_short prefill --batch 128 --seq 128 --pv exact
echo "# prefill_short_kernel<64,2,16> (fp16 P V, V converted inside the kernel: the default call)  B=128 S=128 H/KVH/D=14/2/64"
run prefill_short prefill --batch 128 --seq 128 --pv fp16
