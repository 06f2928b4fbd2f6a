#!/bin/bash
# Which kernels does the vendor fp32 GEMM behind torch.mm run on the model's plain shapes (macro-tile, waves, registers, LDS),
# at what clock and MFMA utilisation -- next to cmf_gemm on the same shapes.  Study target only.
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/${1:-vendor}; mkdir -p $R
rm -rf /tmp/v1 /tmp/v2
(cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/v1 -- python3 $GRAFT_REPO_ROOT/tools/gemm_vendor_compare.py > $R/compare.txt 2>/dev/null)
python tools/kernel_census.py $(find /tmp/v1 -name "*kernel_trace.csv" | head -1) 100 > $R/census_trace.txt
(cd /tmp && rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/v2 -- python3 $GRAFT_REPO_ROOT/tools/gemm_vendor_compare.py > /dev/null 2>&1)
python tools/kernel_census.py $(find /tmp/v2 -name "*counter_collection.csv" | head -1) 100 > $R/census_pmc.txt
cat $R/compare.txt $R/census_trace.txt $R/census_pmc.txt
