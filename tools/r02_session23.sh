set -x
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s23; mkdir -p $O
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-rooflines --serial > $O/bench_serial.json 2> $O/bench_serial.err)
cp $(find /tmp/p3 -name "*kernel_stats.csv" | head -1) $O/serial_kernel_stats.csv
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-rooflines > $O/bench_conc.json 2> $O/bench_conc.err)
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $O/conc_kernel_stats.csv
cp $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) $O/conc_kernel_trace.csv
ls -la $O
