# round 2, GPU session 9: full GPU suite, final bench lines, rocprof kernel stats, PMC passes (GEMM + point-major HBM kernels)
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s9; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --durations=10 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python bench.py --gemm-table $O/gemm_shapes_train.md > $O/bench_train.json 2> $O/bench_train.err
python bench.py --mode fwd --gemm-table $O/gemm_shapes_fwd.md > $O/bench_fwd.json 2>/dev/null
python bench.py --model cmflow_t --no-cpu-baseline --no-op-rooflines > $O/bench_cmflow_t.json 2>/dev/null
python bench.py --model raflow --no-cpu-baseline --no-op-rooflines > $O/bench_raflow.json 2>/dev/null
python bench.py --force-allreduce --no-cpu-baseline --no-op-rooflines > $O/bench_allreduce.json 2>/dev/null
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/p1 /tmp/p2 /tmp/p3 /tmp/g1 /tmp/g2 /tmp/g3 /tmp/h1 /tmp/h2 /tmp/h3
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-rooflines > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-rooflines --mode fwd > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-rooflines --serial > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv
cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $O/fwd_kernel_stats.csv
cp $(find /tmp/p3 -name "*kernel_stats.csv" | head -1) $O/train_serial_kernel_stats.csv
python tools/trace_overlap.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) loss_sample_kernel 5 11 > $O/train_overlap.txt 2>&1
cd /tmp
# GEMM PMC passes (tools/gemm_probe.py: 524288 x 256 x 512 fwd / dX / dW)
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES --output-format csv -d /tmp/g1 -- python3 $GRAFT_REPO_ROOT/tools/gemm_probe.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/g2 -- python3 $GRAFT_REPO_ROOT/tools/gemm_probe.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/g3 -- python3 $GRAFT_REPO_ROOT/tools/gemm_probe.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
for g in g1 g2 g3; do python tools/pmc_summary.py $(find /tmp/$g -name "*counter_collection.csv" | head -1) gemm_kernel > $O/gemm_pmc_$g.txt 2>&1; done
python - > $O/gemm_pmc_durations.txt <<'PY'
import csv, glob, collections
f = glob.glob('/tmp/g1/**/*kernel_trace.csv', recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'gemm_kernel' in r['Kernel_Name']: d[r['Kernel_Name'].split('(')[0]].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in d.items(): print(k, 'launches', len(v), 'mean us %.1f' % (sum(v) / len(v) / 1e3))
PY
# point-major HBM kernels
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/h1 -- python3 $GRAFT_REPO_ROOT/tools/pm_probe.py > $GRAFT_REPO_ROOT/$O/pm_probe.out 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/h2 -- python3 $GRAFT_REPO_ROOT/tools/pm_probe.py > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/h3 -- python3 $GRAFT_REPO_ROOT/tools/pm_probe.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/pm_table.py $(find /tmp/h1 -name "*kernel_trace.csv" | head -1) $(find /tmp/h2 -name "*counter_collection.csv" | head -1) $(find /tmp/h3 -name "*counter_collection.csv" | head -1) $O/pm_probe.out > $O/pm_hbm_table.md 2>&1
head -3 $(find /tmp/h2 -name "*counter_collection.csv" | head -1) > $O/pmc_csv_head.txt
ls -la $O; tail -5 $O/pytest.log
