#!/bin/bash
# A/B of the 128 x 256 tiles of the gathering forward GEMM (CMF_GEMM_WIDE) in one call: the probe at the four scales, inference and training steps
mkdir -p gpurun_out
for w in 0 1 0 1; do echo "== CMF_GEMM_WIDE=$w"; CMF_GEMM_WIDE=$w python tools/gather_probe.py 2>&1 | tail -4; done
for w in 0 1 0 1; do
  CMF_GEMM_WIDE=$w python bench.py --mode fwd --steps 100 --warmup 10 --no-cpu-baseline --no-op-rooflines 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('fwd wide=$w', d['ms_per_step'], d['extra']['ms_per_step_regions'])"
done
for w in 0 1 0 1; do
  CMF_GEMM_WIDE=$w python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-op-rooflines 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('train wide=$w', d['ms_per_step'], d['extra']['ms_per_step_regions'], d['roofline_isolated']['frac'])"
done
