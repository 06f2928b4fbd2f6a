set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s8; mkdir -p $O
python tools/diag_bf16x3_locate.py > $O/locate2.txt 2>&1
