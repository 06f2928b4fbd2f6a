cd $GRAFT_REPO_ROOT
O=gpurun_out/s33; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-op-rooflines --steps 20 --warmup 5"
for r in 1 2; do
timeout 300 $B > $O/bench.json 2> $O/bench.err; python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('q default', d['ms_per_step'], d['value'])"
for q in 2 3 4; do
GPU_MAX_HW_QUEUES=$q timeout 300 $B > $O/bench.json 2> $O/bench.err
python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('q $q', d['ms_per_step'], d['value'])"
done; done
