set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s7; mkdir -p $O
python tools/diag_bf16x3_batch.py > $O/diag.txt 2>&1
timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_raflow.py -m gpu -q --durations=5 > $O/pytest_model.log 2>&1; echo "pytest rc $?" >> $O/pytest_model.log
