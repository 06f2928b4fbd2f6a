"""Aggregate a rocprofv3 kernel_trace.csv by (kernel, grid, block): launches, mean/min duration.  Diagnostic."""
import csv, sys, collections
rows = csv.DictReader(open(sys.argv[1]))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
agg = collections.defaultdict(list)
for r in rows:
    key = (r["Kernel_Name"][:64], r["Grid_Size_X"] + "x" + r["Grid_Size_Y"] + "x" + r["Grid_Size_Z"], r["Workgroup_Size_X"])
    agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
out = sorted(agg.items(), key=lambda kv: -sum(kv[1]))
for (name, grid, wg), d in out[:int(sys.argv[3]) if len(sys.argv) > 3 else 60]:
    print("%-64s grid=%-16s wg=%-4s n/step=%6.1f mean_us=%8.1f min_us=%8.1f ms/step=%7.3f" %
          (name, grid, wg, len(d) / steps, sum(d) / len(d), min(d), sum(d) / 1e3 / steps))
