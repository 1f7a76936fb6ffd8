# what does the vendor GEMM look like on our shapes? kernel names (Tensile encodes the tile config), LDS, registers, grid
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/blp -o blp -- python $R/scripts/blaslt_ref.py > /tmp/blp.log 2>&1
DB=$(find /tmp/blp -name "*.db" | head -1)
python - "$DB" > $R/gpurun_out/blaslt_kernels.txt <<'PY'
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for r in c.execute("select name, count(*), avg(end-start)/1e3, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(workgroup_x), max(grid_x), max(grid_y), max(grid_z) from kernels group by name order by 3 desc"):
    print(r)
PY
cat $R/gpurun_out/blaslt_kernels.txt
