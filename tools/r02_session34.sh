cd $GRAFT_REPO_ROOT
O=gpurun_out/s34; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-op-rooflines --steps 20 --warmup 5"
for r in 1 2 3; do
for w in 0 1; do
CMF_THIN_WIDE=$w timeout 300 $B > $O/bench.json 2> $O/bench.err
python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('wide $w', d['ms_per_step'], d['value'])"
done; done
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_modules.py tests/test_gpu_raflow.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
