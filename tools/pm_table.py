"""Merge rocprofv3 outputs of tools/pm_probe.py into a markdown table: kernel, launches, mean duration, algorithmic bytes,
achieved GB/s, FETCH_SIZE x 2 (gfx950 correction) and WRITE_SIZE per launch.
    python tools/pm_table.py <kernel_trace.csv> <fetch counter_collection.csv> <write counter_collection.csv> <probe stdout>"""
import csv, json, sys, collections
trace, fetch, write, probe = sys.argv[1:5]
alg = json.loads([l for l in open(probe) if l.startswith("ALG ")][0][4:])
dur = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    dur[(r["Kernel_Name"].split("(")[0], r.get("Grid_Size") or r.get("Grid_Size_X"))].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
def counters(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        d[(r["Kernel_Name"].split("(")[0], r["Grid_Size"])].append(float(r["Counter_Value"]))
    return d
F, W = counters(fetch), counters(write)
print("| kernel | grid | launches | mean us | algorithmic MB | achieved GB/s | of 8 TB/s | FETCH_SIZE x2 MB | WRITE_SIZE MB | traffic / algorithmic |")
print("|---|---|---|---|---|---|---|---|---|---|")
for key, nbytes in alg.items():
    name, _, tag = key.partition("@")
    cands = [(k, v) for k, v in dur.items() if k[0] == name]
    if not cands:
        continue
    # the probe's launch of this kernel with the largest grid (or, with a tag, ordered by duration: C = 64 < C = 256)
    cands.sort(key=lambda kv: sum(kv[1]) / len(kv[1]))
    (k, v) = cands[-1] if tag in ("", "256") else cands[-2] if len(cands) > 1 else cands[-1]
    us = sum(v) / len(v) / 1e3
    f = F.get(k); w = W.get(k)
    fmb = 2 * sum(f) / len(f) * 1024 / 1e6 if f else float("nan")      # FETCH_SIZE / WRITE_SIZE are reported in KB
    wmb = sum(w) / len(w) * 1024 / 1e6 if w else float("nan")
    print("| %s | %s | %d | %.1f | %.1f | %.0f | %.3f | %.1f | %.1f | %.2f |" % (key, k[1], len(v), us, nbytes / 1e6, nbytes / us / 1e3,
          nbytes / us / 1e3 / 8000, fmb, wmb, (fmb + wmb) / (nbytes / 1e6)))
