#!/bin/bash
# Step time under a restricted host-core budget (what a rank gets when N ranks share one host): bench.py --host-cores K
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/${1:-hostb}; mkdir -p $R
NP=$(nproc)
{
echo "# tools/host_budget.sh: bench.py --host-cores K (sched_setaffinity in-process before any GPU call), training step B=64, 1 GPU; host has $NP cores"
echo "# K = 0: unrestricted; nproc/8: the share of one of 8 ranks on this host"
for K in 0 $((NP/8)) 8 4 0 $((NP/8)) 8 4; do
  python bench.py --steps 50 --host-cores $K --no-cpu-baseline --no-op-rooflines 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('host-cores %3d  ms/step %.3f  regions %s  %s' % ($K, d['ms_per_step'], d['extra']['ms_per_step_regions'], d['extra']['host_cores']))"
done
} | tee $R/host_budget.txt
