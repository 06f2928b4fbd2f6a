# round 2, GPU session 6: kinds 2/3 at 3 waves/SIMD, q kinds, runtime GEMM mode; full GPU suite under both modes
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s6; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x --durations=10 > $O/pytest.log 2>&1; echo "pytest rc $?" >> $O/pytest.log
python tools/gemm_variants.py > $O/variants_fp32.txt 2>&1
CMF_GEMM_MODE=bf16x3 python tools/gemm_variants.py > $O/variants_bf16x3.txt 2>&1
for i in 1 2; do
python bench.py --no-cpu-baseline --no-op-rooflines > $O/bench_fp32_$i.json 2>/dev/null
CMF_GEMM_MODE=bf16x3 python bench.py --no-cpu-baseline --no-op-rooflines > $O/bench_bf16x3_$i.json 2>/dev/null
done
python bench.py --no-cpu-baseline > $O/bench_full.json 2> $O/bench_full.err
tail -5 $O/pytest.log
