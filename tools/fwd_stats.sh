#!/bin/bash
# per-kernel stats of the inference forward (serial: isolated durations) under rocprofv3
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT/gpurun_out/${1:-fwdstats}; mkdir -p $R
rm -rf /tmp/f1
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/f1 -- python3 $GRAFT_REPO_ROOT/bench.py --mode fwd --steps 20 --warmup 3 --no-cpu-baseline --no-op-rooflines --serial > $R/bench_fwd_serial.json 2>/dev/null)
cp $(find /tmp/f1 -name "*kernel_stats.csv" | head -1) $R/fwd_serial_kernel_stats.csv
python3 - <<PY
import csv
rows = list(csv.DictReader(open("$R/fwd_serial_kernel_stats.csv")))
n = 26.0   # 20 timed + 3 warm-up + 3 isolated-pass steps ... normalise by the loss-free forward count below
calls = {r["Name"]: int(r["Calls"]) for r in rows}
steps = max(1, min(v for k, v in calls.items() if "knn_kernel" in k) // 2)
tot = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
print("forwards: %d, kernel time per forward %.3f ms, launches per forward %.0f" % (steps, tot, sum(calls.values()) / steps))
for r in rows[:32]:
    print("%6.3f ms  n %5.1f  avg %7.1f us  %s" % (float(r["TotalDurationNs"]) / steps / 1e6, int(r["Calls"]) / steps, float(r["AverageNs"]) / 1e3, r["Name"][:100]))
PY
