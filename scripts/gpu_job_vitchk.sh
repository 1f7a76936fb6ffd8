cd $GRAFT_REPO_ROOT
VIMA_VIT_ATTN_LDS=0 python scripts/vit_lds_check.py old_p0_big 0 16384 2>&1 | tail -1
VIMA_VIT_ATTN_LDS=1 python scripts/vit_lds_check.py lds_p0_big 0 16384 2>&1 | tail -1
VIMA_VIT_ATTN_LDS=1 python scripts/vit_lds_check.py lds_p1_big 1 16384 2>&1 | tail -1
VIMA_VIT_ATTN_LDS=1 python scripts/vit_lds_check.py lds_p0_c7 0 7 2>&1 | tail -1
VIMA_VIT_ATTN_LDS=1 python scripts/vit_lds_check.py lds_p0_c8 0 8 2>&1 | tail -1
python - <<'PY'
import torch
L=lambda t: torch.load(f'gpurun_out/vitchk_{t}.pt')
ref=L('old_p0_big')
for t in ('lds_p0_big','lds_p1_big','lds_p0_c7','lds_p0_c8'):
    x=L(t)
    for k in ('ptok','otok'):
        d=(x[k]-ref[k]).abs()
        print(t,k,'equal' if torch.equal(x[k],ref[k]) else f'DIFF max {d.max().item():.3e} n {(d>0).sum().item()} of {d.numel()} rows {sorted(set((d>0).nonzero()[:,0].tolist()))[:8]} {sorted(set((d>0).nonzero()[:,1].tolist()))[:12]}')
PY
rm -f gpurun_out/vitchk_*.pt
