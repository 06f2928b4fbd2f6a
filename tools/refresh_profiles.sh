# One gpurun call that regenerates the bench / kernel-stats / overlap / probe files under profiles/ (raw output in gpurun_out/r).
set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r; mkdir -p $R
python bench.py --gemm-table $R/gemm_shapes_train.md > $R/bench_train.json 2> $R/bench_train.err
python bench.py --mode fwd --gemm-table $R/gemm_shapes_fwd.md > $R/bench_fwd.json 2> $R/bench_fwd.err
python bench.py --model cmflow_t --no-cpu-baseline --no-op-rooflines > $R/bench_cmflow_t.json 2>/dev/null
python bench.py --model raflow --no-cpu-baseline --no-op-rooflines > $R/bench_raflow.json 2>/dev/null
python bench.py --force-allreduce --no-cpu-baseline --no-op-rooflines > $R/bench_forced_allreduce.json 2>/dev/null
export TMPDIR=/tmp
rm -rf /tmp/p1 /tmp/p2 /tmp/p3
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-rooflines > /dev/null 2>&1)
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $R/train_kernel_stats.csv
python tools/trace_overlap.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) loss_sample_kernel 5 11 > $R/train_overlap.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-rooflines --mode fwd > /dev/null 2>&1)
cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $R/fwd_kernel_stats.csv
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-rooflines --serial > /dev/null 2>&1)
cp $(find /tmp/p3 -name "*kernel_stats.csv" | head -1) $R/train_serial_kernel_stats.csv
python tools/enc1_profile.py 2>&1 | grep -v "Warning\|_warn_once\|amdgpu.ids" > $R/enc1_profile.txt
python tools/thin_bwd_probe.py 2>&1 | grep -v amdgpu.ids > $R/thin_bwd_probe.txt
python tools/thin_wide_probe.py 2>&1 | grep -v amdgpu.ids > $R/thin_wide_probe.txt
python tools/ball_query_probe.py 2>&1 | grep -v amdgpu.ids > $R/ball_query_probe.txt
python tools/gemm_vendor_compare.py 2>&1 | grep -v amdgpu.ids > $R/gemm_vendor_compare.txt
python tools/host_time_probe.py 20 2>&1 | grep -v amdgpu.ids > $R/host_time_probe.txt
python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids > $R/phase_probe.txt
python tools/finalize_probe.py 2>&1 | grep -v amdgpu.ids > $R/finalize_probe.txt
bash tools/session.sh r op_pmc > /dev/null 2>&1
ls -la $R
