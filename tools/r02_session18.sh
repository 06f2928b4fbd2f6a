set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s18; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k thin_bwd > $O/pytest_thin.txt 2>&1; tail -15 $O/pytest_thin.txt
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_modules.py tests/test_gpu_raflow.py -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
B="python bench.py --no-cpu-baseline --no-op-rooflines --steps 20 --warmup 5"
CMF_THIN_FUSED=0 $B > $O/bench_f0.json 2> $O/bench_f0.err
CMF_THIN_FUSED=1 $B > $O/bench_f1.json 2> $O/bench_f1.err
CMF_THIN_FUSED=0 $B > $O/bench_f0b.json 2> $O/bench_f0b.err
CMF_THIN_FUSED=1 $B > $O/bench_f1b.json 2> $O/bench_f1b.err
cat $O/bench_f*.json
