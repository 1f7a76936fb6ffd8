cd $GRAFT_REPO_ROOT
for qg in 1 2; do
  QG=$qg timeout 300 python scripts/attn_micro.py 256 12 512 64 10 2>&1 | tail -1
done
QG=1 timeout 300 python scripts/attn_micro.py 256 12 1024 64 5 2>&1 | tail -1
QG=1 CHECK=1 timeout 300 python scripts/attn_micro.py 16 12 500 64 1 2>&1 | tail -2 | head -1
