# One gpurun call that regenerates the bench / kernel-stats / overlap / probe files under profiles/ (raw output in gpurun_out/r).
set -x
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT/gpurun_out/r; mkdir -p $R
export TMPDIR=/tmp
# PMC constants first (bench.py reports them only beside matching kernel sources): the in-step GEMM traffic and the drop-in op table
bash tools/session.sh r pmc_gemm > /dev/null 2>&1
cp $R/gemm_traffic.json profiles/r06_gemm_traffic_instep.json
bash tools/session.sh r op_pmc > /dev/null 2>&1
cp $R/op_hbm_pmc.json profiles/r06_op_hbm_pmc.json
python bench.py --gemm-table $R/gemm_shapes_train.md > $R/bench_train.json 2> $R/bench_train.err
python bench.py --mode fwd --gemm-table $R/gemm_shapes_fwd.md > $R/bench_fwd.json 2> $R/bench_fwd.err
python bench.py --model cmflow_t --no-cpu-baseline --no-op-rooflines > $R/bench_cmflow_t.json 2>/dev/null
python bench.py --model raflow --no-cpu-baseline --no-op-rooflines > $R/bench_raflow.json 2>/dev/null
python bench.py --force-allreduce --no-cpu-baseline --no-op-rooflines > $R/bench_forced_allreduce.json 2>/dev/null
rm -rf /tmp/p1 /tmp/p2 /tmp/p3
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-rooflines --no-config2 > /dev/null 2>&1)
cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $R/train_kernel_stats.csv
python tools/trace_overlap.py $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) loss_sample_kernel 5 11 > $R/train_overlap.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-rooflines --mode fwd > /dev/null 2>&1)
cp $(find /tmp/p2 -name "*kernel_stats.csv" | head -1) $R/fwd_kernel_stats.csv
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-rooflines --no-config2 --serial > /dev/null 2>&1)
cp $(find /tmp/p3 -name "*kernel_stats.csv" | head -1) $R/train_serial_kernel_stats.csv
for c in 0 1; do echo "== CMF_CHAIN_TRAIN=$c (ENC1_SERIAL=1: isolated kernel durations)"; ENC1_SERIAL=1 CMF_CHAIN_TRAIN=$c python tools/enc1_profile.py 2>&1 | grep -v "Warning\|_warn_once\|amdgpu.ids" | head -22; done > $R/enc1_profile.txt
python tools/gemm_vendor_compare.py 2>&1 | grep -v amdgpu.ids > $R/gemm_vendor_compare.txt
python tools/host_time_probe.py 20 2>&1 | grep -v amdgpu.ids > $R/host_time_probe.txt
python tools/phase_probe.py 2>&1 | grep -v amdgpu.ids > $R/phase_probe.txt
python tools/gather_probe.py 2>&1 | grep rows > $R/gather_probe.txt
python tools/aten_census.py 2>&1 | grep -v "amdgpu.ids\|_warn_once" > $R/aten_census.txt
for s in 1 0; do echo "== FC_SERIAL=$s"; FC_SERIAL=$s python tools/fc_profile.py 2>&1 | grep -v "Warning\|_warn_once\|amdgpu.ids" | head -40; done > $R/fc_profile.txt
python tools/peak_mem_probe.py 2>&1 | grep -v "amdgpu.ids" > $R/peak_mem_now.txt
timeout 300 tools/lab/gemm_lab 131072x512x256 131072x512x512 131072x512x1024 524288x256x512 16384x2048x1024 65536x256x512 > $R/gemm_lab.txt 2>&1
LAB_FLAGS=1 timeout 300 tools/lab/gemm_lab 131072x512x256 131072x512x512 131072x512x1024 > $R/gemm_lab_nostore.txt 2>&1
timeout 200 tools/lab/mfma_peak > $R/mfma_peak.txt 2>&1
ls -la $R
