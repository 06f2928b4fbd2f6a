#!/bin/bash
# A/B/A/B of the lock-step first encoder (CMF_BODY_BATCH) in one call: training steps, and the first encoder alone
for b in 0 1 0 1; do
  CMF_BODY_BATCH=$b python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-op-rooflines 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train CMF_BODY_BATCH=$b', d['ms_per_step'], d['extra']['ms_per_step_regions'])"
done
for b in 0 1; do
  echo "== CMF_BODY_BATCH=$b: first encoder alone (both clouds, forward + backward, isolated kernel durations)"
  CMF_BODY_BATCH=$b ENC1_SERIAL=1 CMF_CHAIN_TRAIN=0 python tools/enc1_profile.py 2>&1 | grep -v "Warning\|_warn_once\|amdgpu.ids" | head -16 | cut -c1-150
done
