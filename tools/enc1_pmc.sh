#!/bin/bash
# clock / MFMA busy / wait shares of the first encoder's kernels (serial chains), chain training on and off
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for c in 1 0; do
  rm -rf /tmp/e1
  (cd /tmp && ENC1_SERIAL=1 CMF_CHAIN_TRAIN=$c rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_WAVES --output-format csv -d /tmp/e1 -- python3 $GRAFT_REPO_ROOT/tools/enc1_profile.py > /dev/null 2>&1)
  echo "== CMF_CHAIN_TRAIN=$c"
  python3 tools/kernel_census.py $(find /tmp/e1 -name "*counter_collection.csv" | head -1) 15 | grep -A2 "chain_kernel\|thin_fwd_kernel<4\|thin_bwd_layer_kernel<" | grep -v "^--"
done
