cd $GRAFT_REPO_ROOT
for g in 0 1; do
  timeout 300 python scripts/warm_steps.py warm 20 256 graphs=$g 2>&1 | tail -1
done
for g in 0 1; do
  timeout 300 python scripts/warm_steps.py inc 16 256 graphs=$g 2>&1 | tail -1
done
