# HBM traffic of one decode attention call from the PMC counters (one rocprofv3 --pmc pass per counter and context), for roofline.traffic
# contexts: 1034 (mean context of the driver's --steps 20 run), 1536 (the default --steps 1024 window), 2048
R=${R:-r03}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -f gpurun_out/${R}_pmc_decode_traffic.json
for ctx in 1034 1536 2048; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d gpurun_out/pmct_$c -- python3 tools/microbench.py decode --batch 32 --ctx $ctx --iters 3 --warmup 1 > gpurun_out/pmct_$c.log 2>&1 || exit 1
    f=$(ls gpurun_out/pmct_$c/*/*counter_collection.csv | head -1); cp $f gpurun_out/${R}_pmc_${c,,}_decode_b32_ctx$ctx.csv; rm -rf gpurun_out/pmct_$c
  done
  python3 tools/pmc_traffic.py gpurun_out/${R}_pmc_fetch_size_decode_b32_ctx$ctx.csv gpurun_out/${R}_pmc_write_size_decode_b32_ctx$ctx.csv gpurun_out/${R}_pmc_decode_traffic.json --ctx $ctx --append || exit 1
done
