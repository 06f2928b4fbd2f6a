cd $GRAFT_REPO_ROOT
O=gpurun_out/s33; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-op-rooflines --steps 20 --warmup 5"
for r in 1 2 3; do
for deal in "0,1,2,0|1,2,0,1" "0,1,1,2|1,2,2,0" "0,1,1,2|0,1,1,2" "0,1,1,2|2,0,0,1" "1,0,0,2|2,1,1,0" "0,0,0,1|1,1,1,2" "0,0,1,2|2,2,0,1"; do
CMF_SCALE_SLOTS="$deal" timeout 300 $B > $O/bench.json 2> $O/bench.err
python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('deal $deal', d['ms_per_step'], d['value'])"
done; done
