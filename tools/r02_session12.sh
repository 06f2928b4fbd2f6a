# round 2, GPU session 12: numbers for the docs -- full-size parity printouts, op-level HBM PMC passes, bench
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s12; mkdir -p $O
python -m pytest tests/test_gpu_model.py -m gpu -q -s -k "full_size or two_rank" > $O/pytest_fullsize.log 2>&1; echo "rc $?" >> $O/pytest_fullsize.log
python bench.py > $O/bench_train.json 2> $O/bench_train.err
export TMPDIR=/tmp
cd /tmp; rm -rf /tmp/o1 /tmp/o2 /tmp/o3
rocprofv3 --kernel-trace --output-format csv -d /tmp/o1 -- python3 $GRAFT_REPO_ROOT/tools/group_probe.py > $GRAFT_REPO_ROOT/$O/group_probe.out 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/o2 -- python3 $GRAFT_REPO_ROOT/tools/group_probe.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/o3 -- python3 $GRAFT_REPO_ROOT/tools/group_probe.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py $(find /tmp/o2 -name "*counter_collection.csv" | head -1) ball_query group_points inverse > $O/op_fetch.txt 2>&1
python tools/pmc_summary.py $(find /tmp/o3 -name "*counter_collection.csv" | head -1) ball_query group_points inverse > $O/op_write.txt 2>&1
python - > $O/op_durations.txt <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/o1/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name'].split('(')[0]
    if any(k in n for k in ('ball_query', 'group_points', 'inverse')): d[(n, r['Grid_Size'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(d.items()): print(k[0], 'grid', k[1], 'launches', len(v), 'mean us %.1f' % (sum(v) / len(v) / 1e3))
PY
