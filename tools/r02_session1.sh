# round 2, GPU session 1: full GPU test suite, bench lines, GEMM timeline diagnostics
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s1; mkdir -p $O
nproc > $O/nproc.txt; free -g >> $O/nproc.txt
timeout 1500 python -m pytest tests -m gpu -q -x --durations=15 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python bench.py > $O/bench_train.json 2> $O/bench_train.err
python bench.py --force-allreduce --no-cpu-baseline --no-op-rooflines > $O/bench_train_allreduce.json 2> $O/bench_train_allreduce.err
python bench.py --model cmflow_t --no-cpu-baseline --no-op-rooflines > $O/bench_cmflow_t.json 2> $O/bench_cmflow_t.err
for w in fwd dx plain; do python tools/gemm_timeline.py $w > $O/timeline_$w.txt 2>&1; done
python tools/gemm_timeline.py fwd 131072 512 512 > $O/timeline_fwd_131072.txt 2>&1
CMF_GEMM_DIAG_RT=8 python tools/gemm_timeline.py fwd > $O/timeline_fwd_noepi.txt 2>&1
CMF_GEMM_DIAG_RT=8 python tools/gemm_timeline.py dx > $O/timeline_dx_noepi.txt 2>&1
tail -5 $O/pytest.log; cat $O/bench_train.json
