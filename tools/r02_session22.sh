set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s22; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-op-rooflines --steps 20 --warmup 5"
$B > $O/bench_a.json 2> $O/bench_a.err
$B > $O/bench_b.json 2> $O/bench_b.err
for f in $O/bench_*.json; do python -c "
import json,sys
d=json.loads(open('$f').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'])"; done
timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
