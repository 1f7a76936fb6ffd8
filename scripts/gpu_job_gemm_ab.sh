R=$GRAFT_REPO_ROOT
cd $R
for rep in 1 2; do
for CFG in "0 0" "1 0" "0 1" "1 1"; do
set -- $CFG
export SPREAD=$1 PRIO=$2
for SH in "131072 2304 768" "131072 768 3072" "81920 3072 768" "131072 768 768"; do
python scripts/gemm_micro.py $SH 2 5 2>&1 | tail -1
done
done
done
SPREAD=0 PRIO=0 python scripts/gemm_micro.py 131072 2304 768 1 5 | tail -1
