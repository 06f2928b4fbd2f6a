"""Merge rocprofv3 outputs of tools/pm_probe.py into a markdown table: kernel, mean duration, algorithmic bytes, achieved
GB/s, FETCH_SIZE x 2 (gfx950 correction) and WRITE_SIZE per launch.  Launches are matched by name and order (3 per entry).
    python tools/pm_table.py <kernel_trace.csv> <fetch counter_collection.csv> <write counter_collection.csv> <probe stdout>"""
import csv, json, sys, collections
trace, fetch, write, probe = sys.argv[1:5]
seq = json.loads([l for l in open(probe) if l.startswith("SEQ ")][0][4:])
def by_name(path, value):
    d = collections.defaultdict(list)
    rows = list(csv.DictReader(open(path)))
    key = "Start_Timestamp" if "Start_Timestamp" in rows[0] else "Dispatch_Id"
    rows.sort(key=lambda r: int(r[key]))
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].split("<")[0]
        d[name[5:] if name.startswith("void ") else name].append(value(r))
    return d
D = by_name(trace, lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
F = by_name(fetch, lambda r: float(r["Counter_Value"]))
W = by_name(write, lambda r: float(r["Counter_Value"]))
print("| kernel | mean us | algorithmic MB | achieved GB/s | of 8 TB/s | FETCH_SIZE x2 MB | WRITE_SIZE MB | counted / algorithmic |")
print("|---|---|---|---|---|---|---|---|")
pos = collections.Counter()
for name, nbytes in seq:
    i = pos[name]; pos[name] += 3
    d, f, w = D[name][i:i + 3], F[name][i:i + 3], W[name][i:i + 3]
    us = sum(d) / len(d) / 1e3
    fmb = 2 * sum(f) / len(f) * 1024 / 1e6          # FETCH_SIZE / WRITE_SIZE are reported in KB; FETCH counts 64 B per 128-B request
    wmb = sum(w) / len(w) * 1024 / 1e6
    print("| %s | %.1f | %.1f | %.0f | %.3f | %.1f | %.1f | %.2f |" % (name + (" (C=%d)" % (nbytes // (12 * 524288)) if "bn_bwd" in name else ""),
          us, nbytes / 1e6, nbytes / us / 1e3, nbytes / us / 1e3 / 8000, fmb, wmb, (fmb + wmb) / (nbytes / 1e6)))
