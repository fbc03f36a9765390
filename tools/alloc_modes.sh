R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2 3; do
  for m in "" "2097152,0" "2097152,4096" "2097152,73728" "1073741824,0"; do
    SLAB=$m SEED=$rep timeout -k 10 300 python3 profiles/alloc_probe.py 4 2>/dev/null | grep fwd_ms
  done
done
