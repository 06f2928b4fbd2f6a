cd $GRAFT_REPO_ROOT
O=gpurun_out/s27; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-op-rooflines --steps 20 --warmup 5"
for r in 1 2 3; do
CMF_STACK_KERNELS=0 $B > $O/bench_k0_$r.json 2> $O/bench.err
CMF_STACK_KERNELS=1 $B > $O/bench_k1_$r.json 2> $O/bench.err
done
for f in $O/bench_k*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'])"; done
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_modules.py tests/test_gpu_raflow.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
