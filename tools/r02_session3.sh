# round 2, GPU session 3: GEMM A/B -- 256x128 tiles, backward kinds at 3 waves/SIMD, epilogue priority
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s3; mkdir -p $O
python tools/gemm_variants.py > $O/variants_default.txt 2>&1
CMF_GEMM_TILE256=1 python tools/gemm_variants.py > $O/variants_tile256.txt 2>&1
CMF_LIB=$PWD/tools/diag/libcmflow_w3.so python tools/gemm_variants.py > $O/variants_w3.txt 2>&1
CMF_GEMM_DIAG_RT=16 python tools/gemm_variants.py > $O/variants_prio.txt 2>&1
CMF_GEMM_TILE256=1 timeout 600 python -m pytest tests/test_gpu_gemm.py -q -x > $O/pytest_gemm_tile256.log 2>&1; echo "rc $?" >> $O/pytest_gemm_tile256.log
for i in 1 2; do
python bench.py --no-cpu-baseline --no-op-rooflines > $O/bench_default_$i.json 2>/dev/null
CMF_GEMM_TILE256=1 python bench.py --no-cpu-baseline --no-op-rooflines > $O/bench_tile256_$i.json 2>/dev/null
CMF_LIB=$PWD/tools/diag/libcmflow_w3.so python bench.py --no-cpu-baseline --no-op-rooflines > $O/bench_w3_$i.json 2>/dev/null
done
CMF_GEMM_TILE256=1 python tools/gemm_timeline.py fwd > $O/timeline_fwd_tile256.txt 2>&1
python -X faulthandler bench.py --force-allreduce --no-cpu-baseline --no-op-rooflines --steps 3 --warmup 1 > $O/allreduce.out 2> $O/allreduce.err; echo "rc $?" >> $O/allreduce.err
