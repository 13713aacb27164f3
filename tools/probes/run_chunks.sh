cd $GRAFT_REPO_ROOT
for c in 0 3 2; do
  echo "NVH_DECODE_CHUNKS=$c"
  for ctx in 1025 1536 2048; do NVH_DECODE_CHUNKS=$c timeout -k 10 200 python tools/microbench.py decode --batch 32 --ctx $ctx --graph --width 16 2>&1 | grep us_per_call | cut -c1-150; done
done
