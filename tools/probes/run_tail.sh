# tail timeline of the chunked decode kernel + chunk-count sweep of the attention call
cd $GRAFT_REPO_ROOT
timeout -k 10 300 python tools/probes/stamp_decode.py --tail 2>&1 | grep -v amdgpu.ids
for c in 0 5 6 8; do
  echo "NVH_DECODE_CHUNKS=$c"
  NVH_DECODE_CHUNKS=$c timeout -k 10 200 python tools/microbench.py decode --batch 32 --ctx 1536 --graph 2>&1 | grep us_per_call
  NVH_DECODE_CHUNKS=$c timeout -k 10 200 python tools/microbench.py decode --batch 32 --ctx 1025 --graph 2>&1 | grep us_per_call
  NVH_DECODE_CHUNKS=$c timeout -k 10 200 python tools/microbench.py decode --batch 32 --ctx 2048 --graph 2>&1 | grep us_per_call
done
