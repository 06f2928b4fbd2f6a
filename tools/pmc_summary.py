"""Per-kernel mean of a rocprofv3 --pmc counter from counter_collection.csv (kernels matching a substring)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
want = sys.argv[2:]
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0][:60]
    if any(w in name for w in want):
        agg[(name, r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
for (name, grid, ctr), v in sorted(agg.items()):
    print("%-62s grid=%-9s %-12s launches=%2d mean=%.4g" % (name, grid, ctr, len(v), sum(v) / len(v)))
