cd $GRAFT_REPO_ROOT
O=gpurun_out/s28; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "pooled or thin_bwd" 2>&1 | grep -E "passed|failed|Error|error|assert" | head
B="python bench.py --no-cpu-baseline --no-op-rooflines --steps 20 --warmup 5"
for r in 1 2 3; do
CMF_THIN_FUSED=0 $B > $O/bench_f0_$r.json 2> $O/bench.err
CMF_THIN_FUSED=1 $B > $O/bench_f1_$r.json 2> $O/bench.err
done
for f in $O/bench_f*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'])"; done
timeout 1500 python -m pytest tests/test_gpu_model.py tests/test_gpu_modules.py tests/test_gpu_raflow.py -x -q -m gpu 2>&1 | grep -E "passed|failed"
