# Scheduling A/B runs of the training step (one gpurun call): which encoder scale shares which side stream
# (CMF_SCALE_SLOTS), number of side streams, hardware queues.  Three runs per setting; prints ms per step.
cd $GRAFT_REPO_ROOT
O=gpurun_out/s33; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-op-rooflines --steps 20 --warmup 5"
run() { timeout 300 $B > $O/bench.json 2> $O/bench.err; python -c "
import json
d=json.loads(open('$O/bench.json').read().strip().splitlines()[-1]); print('$1', d['ms_per_step'], d['value'])"; }
for r in 1 2 3; do
for deal in "0,1,1,2|1,2,2,0" "0,1,2,0|1,2,0,1" "0,0,1,2|1,1,2,0" "0,1,2,2|1,2,0,0" "2,1,0,0|0,2,1,1" "0,1,2,1|1,2,0,2" "0,1,1,2|0,1,1,2" "0,0,0,1|1,1,1,2"; do
CMF_SCALE_SLOTS="$deal" run "deal $deal"
done
CMF_SIDE_STREAMS=2 CMF_SCALE_SLOTS="0,1,1,0|1,0,0,1" run "two side streams"
for q in 2 3 4; do GPU_MAX_HW_QUEUES=$q run "GPU_MAX_HW_QUEUES=$q"; done
done
