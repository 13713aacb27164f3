# HBM traffic of one decode attention call from the PMC counters (one rocprofv3 --pmc pass per counter), for roofline.traffic
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmct_$c -- python3 tools/microbench.py decode --batch 32 --ctx 1536 --iters 3 --warmup 1 > gpurun_out/pmct_$c.log 2>&1
  f=$(ls gpurun_out/pmct_$c/*/*counter_collection.csv | head -1); cp $f gpurun_out/r01_pmc_${c,,}_decode_b32_ctx1536.csv; rm -rf gpurun_out/pmct_$c
done
python3 tools/pmc_traffic.py gpurun_out/r01_pmc_fetch_size_decode_b32_ctx1536.csv gpurun_out/r01_pmc_write_size_decode_b32_ctx1536.csv gpurun_out/r01_pmc_decode_traffic.json
