#!/bin/bash
# A/B of the 256 x 128 tile config (CMF_GEMM_TALL) on the plain shapes + the GEMM test file
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/${1:-tall}; mkdir -p $R
for t in 1 0 1 0; do echo "== CMF_GEMM_TALL=$t"; CMF_GEMM_TALL=$t python tools/gemm_vendor_compare.py 2>&1 | grep -v "amdgpu.ids\|Warning"; done | tee $R/tall_ab.txt
timeout 1500 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu 2>&1 | tail -8 | tee $R/gemm_tests.txt
