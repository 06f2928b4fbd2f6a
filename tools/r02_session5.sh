# round 2, GPU session 5: stream / hardware-queue A/B for the encoder chains
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s5; mkdir -p $O
run() { tag=$1; shift; env "$@" python bench.py --no-cpu-baseline --no-op-rooflines > $O/$tag.json 2>/dev/null; }
for i in 1 2; do
run base_$i A=1
run s4q5_$i CMF_SIDE_STREAMS=4 GPU_MAX_HW_QUEUES=5 CMF_SCALE_SLOTS="0,1,2,3|3,2,1,0"
run s8q8_$i CMF_SIDE_STREAMS=8 GPU_MAX_HW_QUEUES=8 CMF_SCALE_SLOTS="0,1,2,3|4,5,6,7"
run s8q9_$i CMF_SIDE_STREAMS=8 GPU_MAX_HW_QUEUES=9 CMF_SCALE_SLOTS="0,1,2,3|4,5,6,7"
run s8q4_$i CMF_SIDE_STREAMS=8 CMF_SCALE_SLOTS="0,1,2,3|4,5,6,7"
run s6q7_$i CMF_SIDE_STREAMS=6 GPU_MAX_HW_QUEUES=7 CMF_SCALE_SLOTS="0,1,2,3|4,5,3,2"
done
python tools/enc1_profile.py > $O/enc1_base.txt 2>&1
CMF_SIDE_STREAMS=8 GPU_MAX_HW_QUEUES=9 CMF_SCALE_SLOTS="0,1,2,3|4,5,6,7" python tools/enc1_profile.py > $O/enc1_s8q9.txt 2>&1
CMF_GEMM_MODE=bf16x3 python bench.py --no-cpu-baseline --no-op-rooflines > $O/bf16x3.json 2>/dev/null
CMF_GEMM_MODE=bf16x3 python tools/gemm_variants.py > $O/variants_bf16x3.txt 2>&1
