cd $GRAFT_REPO_ROOT
echo "strided heads [B,L,24,32]:"; LQ=8 MODE=1 timeout 300 python scripts/attn_micro.py 256 24 512 32 20 2>&1 | tail -1
echo "contiguous per (b,h) (H=1, B=6144):"; LQ=8 MODE=1 timeout 300 python scripts/attn_micro.py 6144 1 512 32 20 2>&1 | tail -1
echo "two heads per row (H=2, B=3072):"; LQ=8 MODE=1 timeout 300 python scripts/attn_micro.py 3072 2 512 32 20 2>&1 | tail -1
echo "four heads per row (H=4, B=1536):"; LQ=8 MODE=1 timeout 300 python scripts/attn_micro.py 1536 4 512 32 20 2>&1 | tail -1
