# per-phase shader-clock stamps of the persistent GEMM on the in-model shapes (prologue / main loop / epilogue per tile)
cd $GRAFT_REPO_ROOT
for spec in "131072 768 768 RES=1" "131072 768 768 RES=1 STREAMT=1" "131072 768 3072 RES=1" "131072 768 3072 RES=1 STREAMT=1" "81920 768 768 RES=1 BIAS=1" "81920 768 768 RES=1 BIAS=1 STREAMT=1" "131072 2304 768 BF16OUT=1" "131072 3072 768 BF16OUT=1 ACT=1" "81920 3072 768 BF16OUT=1 BIAS=1 ACT=3"; do
  set -- $spec; M=$1; N=$2; K=$3; shift 3
  env STAMPS=1 "$@" python scripts/gemm_micro.py $M $N $K 0 5 2>&1 | grep -v amdgpu.ids | sed "s/^/[$*] /"
done
