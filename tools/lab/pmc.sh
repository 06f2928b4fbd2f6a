#!/bin/bash
# clock + MFMA busy of lab variants: tools/lab/pmc.sh "<LAB_ONLY filter>" "<flags list>" <shape...>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
ONLY="$1"; FL="$2"; shift 2
for f in $FL; do
  rm -rf /tmp/lp
  (cd /tmp && LAB_ONLY="$ONLY" LAB_FLAGS=$f rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/lp -- $GRAFT_REPO_ROOT/tools/lab/gemm_lab "$@" > /dev/null 2>&1)
  echo "== FLAGS $f"
  python3 tools/kernel_census.py $(find /tmp/lp -name "*counter_collection.csv" | head -1) 100 | grep -v "fill_kernel\|naive" | sed 's/(Args)//'
done
