set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s10; mkdir -p $O
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/h1 /tmp/h2 /tmp/h3
rocprofv3 --kernel-trace --output-format csv -d /tmp/h1 -- python3 $GRAFT_REPO_ROOT/tools/pm_probe.py > $GRAFT_REPO_ROOT/$O/pm_probe.out 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/h2 -- python3 $GRAFT_REPO_ROOT/tools/pm_probe.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/h3 -- python3 $GRAFT_REPO_ROOT/tools/pm_probe.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/pm_table.py $(find /tmp/h1 -name "*kernel_trace.csv" | head -1) $(find /tmp/h2 -name "*counter_collection.csv" | head -1) $(find /tmp/h3 -name "*counter_collection.csv" | head -1) $O/pm_probe.out > $O/pm_hbm_table.md 2>&1
head -3 $(find /tmp/h2 -name "*counter_collection.csv" | head -1) > $O/pmc_csv_head.txt
tail -3 $O/pm_probe.out
