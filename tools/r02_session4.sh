# round 2, GPU session 4: profiles of the current build (kernel trace, overlap, launch census, host time), all-reduce A/B
set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/s4; mkdir -p $O
python bench.py --gemm-table $O/gemm_shapes_train.md > $O/bench_train.json 2> $O/bench_train.err
python bench.py --mode fwd --no-cpu-baseline --no-op-rooflines --gemm-table $O/gemm_shapes_fwd.md > $O/bench_fwd.json 2>/dev/null
for i in 1 2; do
python bench.py --no-cpu-baseline --no-op-rooflines > $O/ab_plain_$i.json 2>/dev/null
python bench.py --no-cpu-baseline --no-op-rooflines --force-allreduce > $O/ab_allreduce_$i.json 2>/dev/null
GPU_MAX_HW_QUEUES=5 python bench.py --no-cpu-baseline --no-op-rooflines --force-allreduce > $O/ab_allreduce_q5_$i.json 2>/dev/null
GPU_MAX_HW_QUEUES=5 python bench.py --no-cpu-baseline --no-op-rooflines > $O/ab_plain_q5_$i.json 2>/dev/null
done
python bench.py --model cmflow_t --no-cpu-baseline --no-op-rooflines > $O/bench_cmflow_t.json 2>/dev/null
python tools/host_profile.py > $O/host_profile.txt 2>&1
export TMPDIR=/tmp
rm -rf /tmp/p1 /tmp/p3
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-rooflines > /dev/null 2>&1)
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/train_kernel_stats.csv
python tools/trace_overlap.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) loss_sample_kernel 5 11 > $O/train_overlap.txt 2>&1
python tools/trace_census.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) > $O/train_census.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-rooflines --serial > /dev/null 2>&1)
cp $(find /tmp/p3 -name "*kernel_stats.csv" | head -1) $O/train_serial_kernel_stats.csv
ls -la $O
