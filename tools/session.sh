#!/bin/bash
# One gpurun session: tools/session.sh <outdir> <step> [<step> ...] runs the named steps below in order, output under
# gpurun_out/<outdir>/.
# (replaces the one-off tools/r02_session*.sh scripts of round 2)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
NAME=$1; shift
R=$GRAFT_REPO_ROOT/gpurun_out/$NAME; mkdir -p $R
filter() { grep -v "amdgpu.ids\|_warn_once\|Warning:" ; }

gemm_diag() {
    for d in 0 8 9 12 1; do CMF_GEMM_DIAG_RT=$d python tools/gemm_diag.py 2>&1 | filter; done > $R/gemm_diag.txt
    CMF_GEMM_DIAG_RT=8 python tools/gemm_timeline.py dxq 524288 256 512 2>&1 | filter > $R/timeline_dxq_noepi.txt
    python tools/gemm_timeline.py dxq 524288 256 512 2>&1 | filter > $R/timeline_dxq.txt
}
pgemm() {              # persistent GEMM: parity against the tiled kernel, then A/B rates on the model's data-gradient shapes
    timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "persistent" 2>&1 | tail -15 > $R/pgemm_tests.txt; cat $R/pgemm_tests.txt
    timeout 600 python tools/pgemm_bench.py 2>&1 | filter > $R/pgemm_bench.txt; cat $R/pgemm_bench.txt
}
pexp() {               # experiment builds of the persistent kernel (tools/diag/libcmflow_<name>.so, VARIANTS="...") against the product, one shape
    for v in product $VARIANTS; do
        L=$GRAFT_REPO_ROOT/tools/diag/libcmflow_$v.so; [ $v = product ] && L=
        echo "== $v"; CMF_LIB=$L timeout 300 python tools/pgemm_bench.py ${PSHAPES:-524288x512x256} 2>&1 | filter
    done | tee $R/pexp.txt
}
prace() { for a in "65536x512x256 2 1 60" "65536x512x256 1 1 60" "65536x512x256 1 0 60" "131072x512x256 2 0 60"; do echo "== $a"; timeout 300 python tools/pgemm_race.py $a 2>&1 | filter | tail -12; done | tee $R/prace.txt; }
pclock() {             # effective clock + MFMA busy of the persistent kernel (product and experiment builds): GRBM_GUI_ACTIVE / duration
    for v in product $VARIANTS; do
        L=$GRAFT_REPO_ROOT/tools/diag/libcmflow_$v.so; [ $v = product ] && L=
        rm -rf /tmp/pc_$v
        (cd /tmp && CMF_LIB=$L rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY --output-format csv -d /tmp/pc_$v -- python3 $GRAFT_REPO_ROOT/tools/pgemm_bench.py 524288x512x256 > /dev/null 2>&1)
        echo "== $v"; python3 tools/pclock.py $(find /tmp/pc_$v -name "*kernel_trace.csv" | head -1) $(find /tmp/pc_$v -name "*counter_collection.csv" | head -1)
    done | tee $R/pclock.txt
}
ptimeline() { for a in "524288x512x256 1 1" "524288x512x256 2 0" "65536x512x256 1 1" "131072x512x512 2 1"; do echo "== $a"; timeout 300 python tools/pgemm_timeline.py $a 2>&1 | filter; done | tee $R/ptimeline.txt; }
pgrid() { for g in 256 512 256 512; do echo "== grid $g"; PGRID=$g timeout 300 python tools/pgemm_bench.py 524288x512x256 131072x512x512 2>&1 | filter; done | tee $R/pgrid.txt; }
op_pmc() {              # the drop-in calls of the bench line's roofline_hbm rows: durations + FETCH_SIZE / WRITE_SIZE (separate passes) -> table
    rm -rf /tmp/o1 /tmp/o2 /tmp/o3
    (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/o1 -- python3 $GRAFT_REPO_ROOT/tools/op_probe.py > $R/op_probe.out 2>/dev/null)
    (cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/o2 -- python3 $GRAFT_REPO_ROOT/tools/op_probe.py > /dev/null 2>&1)
    (cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/o3 -- python3 $GRAFT_REPO_ROOT/tools/op_probe.py > /dev/null 2>&1)
    python tools/op_table.py $(find /tmp/o1 -name "*kernel_trace.csv" | head -1) $(find /tmp/o2 -name "*counter_collection.csv" | head -1) \
        $(find /tmp/o3 -name "*counter_collection.csv" | head -1) $R/op_probe.out $R/op_hbm_pmc.json > $R/op_hbm_pmc.md; cat $R/op_hbm_pmc.md
    python tools/op_kernels.py $(find /tmp/o1 -name "*kernel_trace.csv" | head -1) > $R/op_kernels.txt; cat $R/op_kernels.txt
}
gemm_tests() { python -m pytest tests/test_gpu_gemm.py -x -q -m gpu 2>&1 | tail -8 > $R/gemm_tests.txt; }
model_tests() { python -m pytest tests/test_gpu_model.py -q -m gpu -s -k "full_size or two_rank" 2>&1 | tail -30 > $R/model.txt; }
bench3() { for i in 1 2 3; do python bench.py --no-cpu-baseline --no-op-rooflines 2>/dev/null; done > $R/bench3.json; cat $R/bench3.json | python -c "import sys,json; [print(json.loads(l)['ms_per_step'], json.loads(l)['roofline']['frac'], json.loads(l)['roofline_isolated']['frac']) for l in sys.stdin]"; }
ops_tests() { python -m pytest tests/test_gpu_ops.py tests/test_gpu_extension_surface.py tests/test_gpu_modules.py -q -m gpu 2>&1 | tail -15 > $R/ops_tests.txt; }
epi_diag() {           # which part of the backward epilogue costs: 16 no C stores, 32 no Z loads, 48 neither, 8 no epilogue
    for d in 0 16 32 48 8; do CMF_GEMM_DIAG_RT=$d python tools/gemm_diag.py 2>&1 | filter | grep "dX BN"; done > $R/epi_diag.txt
}
model_full() { python -m pytest tests/test_gpu_model.py -q -m gpu -s -k "full_size_train or full_size_cmflow_t" 2>&1 | grep -v "^  \|^$" | tail -60 > $R/model_full.txt; }
opbench() { python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>$R/opbench.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); [print(r) for r in d['roofline_hbm']['rows']]" > $R/opbench.txt; tail -3 $R/opbench.err; }
trace() {              # kernel trace of 10 timed steps (csv copied back for tools/trace_overlap.py / offline analysis) + aten attribution
    rm -rf /tmp/p1
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-rooflines --no-config2 > /dev/null 2>&1)
    cp $(find /tmp/p1 -name "*kernel_stats.csv" | head -1) $R/train_kernel_stats.csv
    cp $(find /tmp/p1 -name "*kernel_trace.csv" | head -1) $R/train_kernel_trace.csv
    python tools/trace_overlap.py $R/train_kernel_trace.csv loss_sample_kernel 5 11 > $R/train_overlap.txt 2>&1
    python tools/aten_sources.py 2>&1 | filter > $R/aten_sources.txt
    python tools/phase_probe.py 2>&1 | filter > $R/phase_probe.txt
}
dp_tests() { python -m pytest tests/test_gpu_model.py -q -m gpu -s -k "two_rank or rccl or bench_two" 2>&1 | tail -12 > $R/dp_tests.txt; }
allreduce_ab() {       # world-1 RCCL all-reduce inside every step: none / overlapped segments / one bucket after backward
    python bench.py --no-cpu-baseline --no-op-rooflines 2>/dev/null > $R/ar_none.json
    CMF_OVERLAP_ALLREDUCE=1 python bench.py --no-cpu-baseline --no-op-rooflines --force-allreduce 2>/dev/null > $R/ar_overlap.json
    python bench.py --no-cpu-baseline --no-op-rooflines --force-allreduce 2>/dev/null > $R/ar_single.json
    for f in none overlap single; do python -c "import json; print('$f', json.loads(open('$R/ar_$f.json').read())['ms_per_step'])"; done
}
tail_ab() { for i in 1 2; do python bench.py --no-cpu-baseline --no-op-rooflines 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('batched tails', d['ms_per_step'], d['roofline_isolated']['frac'])";
    CMF_TAIL_BATCH=0 python bench.py --no-cpu-baseline --no-op-rooflines 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('per-block tails', d['ms_per_step'])"; done > $R/tail_ab.txt; cat $R/tail_ab.txt; }
model_quick() { python -m pytest tests/test_gpu_model.py tests/test_gpu_stress.py tests/test_gpu_raflow.py -x -q -m gpu -k "not full_size and not two_rank and not bench_two and not dense" 2>&1 | grep -E "passed|failed|Error|assert" | tail -8 > $R/model_quick.txt; cat $R/model_quick.txt; }
mlp_test() { python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "mlp_chain or small_m" 2>&1 | grep -E "passed|failed|Error|assert" | tail -8 > $R/mlp_test.txt; cat $R/mlp_test.txt; }
phases() { python tools/phase_probe.py 2>&1 | filter > $R/phase_probe.txt; cat $R/phase_probe.txt; }
serial_ab() { for i in 1 2; do python bench.py --no-cpu-baseline --no-op-rooflines 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('streams', d['ms_per_step'])";
    python bench.py --no-cpu-baseline --no-op-rooflines --serial 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('serial', d['ms_per_step'])";
    CMF_SIDE_STREAMS=2 python bench.py --no-cpu-baseline --no-op-rooflines 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('2 side streams', d['ms_per_step'])"; done > $R/serial_ab.txt; cat $R/serial_ab.txt; }
pmc_gemm() {           # HBM traffic of cmf_gemm inside the training step: FETCH_SIZE and WRITE_SIZE in separate passes (TCC slot limit)
    for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf /tmp/pm_$c
        (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pm_$c -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-op-rooflines --no-config2 > /dev/null 2>&1)
        cp $(find /tmp/pm_$c -name "*counter_collection.csv" | head -1) $R/pmc_$c.csv
    done
    python tools/gemm_traffic.py $R/pmc_FETCH_SIZE.csv $R/pmc_WRITE_SIZE.csv > $R/gemm_traffic.json; cat $R/gemm_traffic.json
    rm -f $R/pmc_FETCH_SIZE.csv $R/pmc_WRITE_SIZE.csv
}
serial_profile() {     # the product's launches with every chain on one stream: bench line + per-kernel stats under rocprofv3
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-op-rooflines --gemm-table $R/gemm_shapes_train.md > $R/bench_short.json 2>/dev/null
    python -c "import json; d=json.load(open('$R/bench_short.json')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline_isolated']['frac'], d['roofline_isolated']['launches'])"
    rm -rf /tmp/p3
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-op-rooflines --no-config2 --serial > $R/bench_serial.json 2>/dev/null)
    cp $(find /tmp/p3 -name "*kernel_stats.csv" | head -1) $R/train_serial_kernel_stats.csv
    python -c "import json; d=json.load(open('$R/bench_serial.json')); print('serial', d['ms_per_step'], d['roofline']['frac'])"
}
pm_hbm() {             # HBM-bound kernels of the product path: durations + FETCH_SIZE / WRITE_SIZE in separate passes -> markdown table
    rm -rf /tmp/q1 /tmp/q2 /tmp/q3
    (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/q1 -- python3 $GRAFT_REPO_ROOT/tools/pm_probe.py > $R/pm_probe.out 2>/dev/null)
    (cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/q2 -- python3 $GRAFT_REPO_ROOT/tools/pm_probe.py > /dev/null 2>&1)
    (cd /tmp && rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/q3 -- python3 $GRAFT_REPO_ROOT/tools/pm_probe.py > /dev/null 2>&1)
    python tools/pm_table.py $(find /tmp/q1 -name "*kernel_trace.csv" | head -1) $(find /tmp/q2 -name "*counter_collection.csv" | head -1) \
        $(find /tmp/q3 -name "*counter_collection.csv" | head -1) $R/pm_probe.out > $R/pm_hbm_kernels.md; cat $R/pm_hbm_kernels.md
}
bnb_ab() {             # BN backward of the 512 -> 256 layer inside the weight-gradient GEMM (default) against the stand-alone pass
    python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "fused_bn_backward or block or setconv or set_conv" 2>&1 | grep -E "passed|failed|Error|assert" | tail -8
    for v in 0 1 0 1; do CMF_BNB_FUSED=$v python bench.py --steps 40 --no-cpu-baseline --no-op-rooflines 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused $v', d['ms_per_step'], d['roofline']['frac'], d['roofline_isolated']['frac'])"; done | tee $R/bnb_ab.txt
}
ar_trace() {            # kernel + HIP API trace of the step's tail with and without the forced world-1 all-reduce (tools/step_tail_probe.py)
    for v in base forced; do
        rm -rf /tmp/ar_$v
        F=""; [ $v = forced ] && F="--force-allreduce"
        (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/ar_$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-op-rooflines $F > $R/ar_bench_$v.json 2>/dev/null)
        python tools/step_tail_probe.py $(find /tmp/ar_$v -name "*kernel_trace.csv" | head -1) 10 > $R/ar_tail_$v.txt 2>&1
    done
    for v in base forced base forced; do F=""; [ $v = forced ] && F="--force-allreduce"
        python bench.py --no-cpu-baseline --no-op-rooflines $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['extra']['ms_per_step_regions'])"; done | tee $R/ar_ab.txt
}
ar_diag() {             # where the forced world-1 all-reduce's step time goes: process group only / per-kernel totals base vs forced
    for v in base pgonly forced base pgonly forced; do F=""; E=0; [ $v = forced ] && F="--force-allreduce"; [ $v = pgonly ] && E=1
        CMF_BENCH_PG_ONLY=$E python bench.py --no-cpu-baseline --no-op-rooflines $F 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['extra']['ms_per_step_regions'])"; done | tee $R/ar_diag.txt
    for v in base forced; do
        rm -rf /tmp/ard_$v
        F=""; [ $v = forced ] && F="--force-allreduce"
        (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ard_$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-op-rooflines $F > /dev/null 2>&1)
        cp $(find /tmp/ard_$v -name "*kernel_stats.csv" | head -1) $R/ard_stats_$v.csv
    done
}
knobs() {               # hardware-queue count and side-stream count re-measured on the current step (round 1-2 settled them at a 27 ms step)
    one() { python bench.py --steps 60 --no-cpu-baseline --no-op-rooflines 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], d['extra']['ms_per_step_regions'])"; }
    for r in 1 2; do
        one "default"
        for q in 2 3 6 8; do GPU_MAX_HW_QUEUES=$q one "GPU_MAX_HW_QUEUES=$q"; done
        for s in 1 2 4; do CMF_SIDE_STREAMS=$s one "CMF_SIDE_STREAMS=$s"; done
        CMF_STREAM_PROBE=0 one "CMF_STREAM_PROBE=0"
    done | tee $R/knobs.txt
}
peak_mem() {            # peak device memory of the step: this tree, CMF_TRAIN_GATHER=0 (materialised sizes), and the pre-change worktree if present
    python tools/peak_mem_probe.py 2>&1 | filter | tee $R/peak_mem.txt
    CMF_TRAIN_GATHER=0 python tools/peak_mem_probe.py 2>&1 | filter | sed 's/^/CMF_TRAIN_GATHER=0 /' | tee -a $R/peak_mem.txt
    [ -d tools/diag/old ] && python tools/peak_mem_probe.py tools/diag/old 2>&1 | filter | tee -a $R/peak_mem.txt
}
bnbg_ab() {             # BN backward inside the gathering weight-gradient GEMM (CMF_BNB_GATHER=1) against the stand-alone pass
    python -m pytest tests/test_gpu_gemm.py -x -q -m gpu -k "fused_bn_backward" 2>&1 | grep -E "passed|failed|Error|assert" | tail -5 | tee $R/bnbg_tests.txt
    for v in 1 0 1 0; do CMF_BNB_GATHER=$v python bench.py --steps 60 --no-cpu-baseline --no-op-rooflines 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('CMF_BNB_GATHER=$v', d['ms_per_step'], d['extra']['ms_per_step_regions'], d['roofline_isolated']['frac'])"; done | tee $R/bnbg_ab.txt
}
fc() { for s in 1 0; do echo "== FC_SERIAL=$s"; FC_SERIAL=$s python tools/fc_profile.py 2>&1 | filter | head -40; done > $R/fc_profile.txt; head -12 $R/fc_profile.txt; }
dxsum_diag() {          # summed data gradient: product build, then the diagnostics build with 32 no per-point loads / 64 no row walk / 96 both / 8 no epilogue
    echo "== product"; python tools/dxsum_probe.py 2>&1 | filter | grep -v "storage order" | tee $R/dxsum_product.txt
    for d in 0 32 64 96 8; do echo "== diag build, CMF_GEMM_DIAG_RT=$d"; CMF_LIB=$GRAFT_REPO_ROOT/tools/diag/libcmflow_x.so CMF_GEMM_DIAG_RT=$d python tools/dxsum_probe.py 2>&1 | filter | grep -v "storage order"; done | tee $R/dxsum_diag.txt
}
allreduce_abab() {      # world-1 RCCL all-reduce inside every step, two rounds: none / one bucket after backward / three segments overlapped with backward
    for r in 1 2; do
        python bench.py --steps 60 --no-cpu-baseline --no-op-rooflines --no-config2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('none   ', d['ms_per_step'], d['extra']['ms_per_step_regions'])"
        python bench.py --steps 60 --no-cpu-baseline --no-op-rooflines --no-config2 --force-allreduce 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('single ', d['ms_per_step'], d['extra']['ms_per_step_regions'], d['allreduce_ms'])"
        CMF_OVERLAP_ALLREDUCE=1 python bench.py --steps 60 --no-cpu-baseline --no-op-rooflines --no-config2 --force-allreduce 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('overlap', d['ms_per_step'], d['extra']['ms_per_step_regions'])"
    done | tee $R/allreduce_abab.txt
}
dense_train() { python -m pytest tests/test_gpu_model.py -x -q -m gpu -s -k "dense_cloud" 2>&1 | tail -25 > $R/dense_train.txt; cat $R/dense_train.txt; }
adam_test() { python -m pytest tests/test_gpu_model.py -x -q -m gpu -k "flat_adam" 2>&1 | tail -8 > $R/adam_test.txt; cat $R/adam_test.txt; }
suite() { python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $R/suite.txt; }
bench() { python bench.py > $R/bench.json 2> $R/bench.err; tail -3 $R/bench.err; cat $R/bench.json; }

for step in "$@"; do echo "== $step"; $step; done
ls -la $R
