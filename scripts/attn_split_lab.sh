cd /root/repo
for rep in 1 2 3; do for t in base splitp; do echo "== $t"; VIMA_HIP_LIB=build_ablate/libvima_hip_$t.so MODE=1 LQ=8 CHECK=1 timeout 120 python scripts/attn_micro.py 256 24 512 32 30 2>&1 | tail -2; done; done
VIMA_HIP_LIB=build_ablate/libvima_hip_splitp.so MODE=1 LQ=8 CHECK=1 timeout 120 python scripts/attn_micro.py 16 24 1500 32 3 2>&1 | tail -2
VIMA_HIP_LIB=build_ablate/libvima_hip_splitp.so MODE=1 LQ=8 CHECK=1 timeout 120 python scripts/attn_micro.py 16 12 300 64 3 2>&1 | tail -2
