# round 2, GPU session 2: new GEMM epilogue (kind-specialised, straight-line), ref path via oracle modules over HIP ops
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s2; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x > $O/pytest_gemm.log 2>&1; echo "rc $?" >> $O/pytest_gemm.log
for w in fwd dx plain dw; do python tools/gemm_timeline.py $w > $O/timeline_$w.txt 2>&1; done
python tools/gemm_timeline.py fwd 131072 512 512 > $O/timeline_fwd_131072.txt 2>&1
python tools/gemm_timeline.py dx 131072 512 512 > $O/timeline_dx_131072.txt 2>&1
python tools/gemm_variants.py > $O/gemm_variants.txt 2>&1
python bench.py --no-cpu-baseline --no-op-rooflines > $O/bench_train.json 2> $O/bench_train.err
python bench.py --mode fwd --no-cpu-baseline --no-op-rooflines > $O/bench_fwd.json 2> $O/bench_fwd.err
python bench.py --force-allreduce --no-cpu-baseline --no-op-rooflines > $O/bench_train_allreduce.json 2> $O/bench_train_allreduce.err; echo "rc $?" >> $O/bench_train_allreduce.err
timeout 1500 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_gemm.py > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python tools/aten_census.py > $O/aten_census.txt 2>&1
tail -5 $O/pytest_gemm.log $O/pytest.log; cat $O/bench_train.json | cut -c1-400
