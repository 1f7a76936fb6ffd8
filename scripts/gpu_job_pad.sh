R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2; do
for PAD in 0 64 128 192 32; do
  for SH in "131072 768 3072" "131072 2304 768"; do
    PAD=$PAD STAMPS=1 timeout 120 python scripts/gemm_micro.py $SH 2 5 2>&1 | tail -3 | grep -v "real time" | cut -c1-200
  done
done
done
