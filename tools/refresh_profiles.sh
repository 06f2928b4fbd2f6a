set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r
python bench.py > gpurun_out/r/bench_train.json 2> gpurun_out/r/bench_train.err
python bench.py --mode fwd > gpurun_out/r/bench_fwd.json 2> gpurun_out/r/bench_fwd.err
python bench.py --model cmflow_t --no-cpu-baseline > gpurun_out/r/bench_cmflow_t.json 2>/dev/null
python bench.py --model raflow --no-cpu-baseline > gpurun_out/r/bench_raflow.json 2>/dev/null
CMF_GEMM_MODE=bf16x3 python bench.py --no-cpu-baseline > gpurun_out/r/bench_bf16x3.json 2>/dev/null
export TMPDIR=/tmp
rm -rf /tmp/p1 /tmp/p2 /tmp/p3
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2>&1)
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) gpurun_out/r/train_kernel_stats.csv
python tools/trace_overlap.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) loss_sample_kernel 5 11 > gpurun_out/r/train_overlap.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --mode fwd > /dev/null 2>&1)
cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) gpurun_out/r/fwd_kernel_stats.csv
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --serial > /dev/null 2>&1)
cp $(find /tmp/p3 -name "*kernel_stats.csv" | head -1) gpurun_out/r/train_serial_kernel_stats.csv
ls -la gpurun_out/r
