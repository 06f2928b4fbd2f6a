#!/bin/bash
# A/B of the narrow forward layers' full-tile body (CMF_THIN_GENERAL=1: the general body) in one call
for g in 1 0 1 0; do
  CMF_THIN_GENERAL=$g python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-op-rooflines 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train general=$g', d['ms_per_step'], d['extra']['ms_per_step_regions'])"
done
for g in 1 0 1 0; do
  CMF_THIN_GENERAL=$g python bench.py --mode fwd --steps 100 --warmup 10 --no-cpu-baseline --no-op-rooflines 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fwd general=$g', d['ms_per_step'], d['extra']['ms_per_step_regions'])"
done
