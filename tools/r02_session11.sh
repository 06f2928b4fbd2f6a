set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s11; mkdir -p $O
python tools/gemm_variants.py fwd > /dev/null 2>&1   # warm the clocks
for w in fwd dx plain; do python tools/gemm_timeline.py $w > $O/timeline_$w.txt 2>&1; done
python bench.py --gemm-table $O/gemm_shapes_train.md > $O/bench_train.json 2> $O/bench_train.err
