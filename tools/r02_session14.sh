set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s14; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python bench.py > $O/bench_train.json 2> $O/bench_train.err
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
