set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s13; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_extension_surface.py -m gpu -q -x > $O/pytest_ops.log 2>&1; echo "rc $?" >> $O/pytest_ops.log
python bench.py --no-cpu-baseline --steps 5 --warmup 2 > $O/bench.json 2> $O/bench.err
