cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pl -- python3 $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-op-rooflines > /dev/null 2>&1)
python3 - <<'PY'
import csv,glob
f=glob.glob("/tmp/pl/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "loss_sample" in r["Name"] or "knn_kernel" in r["Name"]: print(r["Name"][:40], r["Calls"], "avg us %.1f"%(float(r["AverageNs"])/1e3))
PY
timeout 900 python -m pytest tests/test_gpu_loss.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
