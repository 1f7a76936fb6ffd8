# round 3: persistent vs ping-pong GEMM on random data (no python): square calibration shapes + in-model shapes, with phase stamps
cd $GRAFT_REPO_ROOT
L=scripts/micro/gemm_lab
export STAMPS=1
timeout 120 $L 4096 4096 4096 1 0 7
timeout 120 $L 8192 8192 8192 1 0 5
timeout 120 $L 131072 2304 768 1 0 5
timeout 120 $L 131072 3072 768 1 1 5
timeout 120 $L 131072 768 3072 4 0 5
timeout 120 $L 131072 768 768 4 0 5
timeout 120 $L 81920 3072 768 1 3 5
