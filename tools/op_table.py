"""Merge three rocprofv3 passes over tools/op_probe.py (kernel trace, --pmc FETCH_SIZE, --pmc WRITE_SIZE) into a markdown
table per (op, shape): kernels of one call, duration, algorithmic bytes, achieved GB/s against 8 TB/s, counted
HBM bytes (FETCH_SIZE x 2: gfx950 counts 64 B per 128-B request; both counters are reported in KB) and counted / algorithmic.
    python tools/op_table.py <kernel_trace.csv> <fetch counter_collection.csv> <write counter_collection.csv> <probe stdout> [out.json]"""
import csv, json, sys, collections
import os as _os, sys as _sys
_sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))))
from cmflow_amd import _lib
trace, fetch, write, probe = sys.argv[1:5]
json_out = sys.argv[5] if len(sys.argv) > 5 else None
records = []
phases = json.loads([l for l in open(probe) if l.startswith("PHASES ")][0][7:])


def segments(path, value):
    rows = list(csv.DictReader(open(path)))
    key = "Start_Timestamp" if "Start_Timestamp" in rows[0] else "Dispatch_Id"
    rows.sort(key=lambda r: int(r[key]))
    segs, cur = [], None
    for r in rows:
        name = r["Kernel_Name"]
        if "sigmoid" in name:
            cur = []
        elif "cos" in name and "elementwise" in name:
            if cur is not None:
                segs.append(cur)
            cur = None
        elif cur is not None:
            short = name.split("(")[0]
            cur.append((short[5:] if short.startswith("void ") else short, value(r)))
    return segs


D = segments(trace, lambda r: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
F = segments(fetch, lambda r: float(r["Counter_Value"]))
W = segments(write, lambda r: float(r["Counter_Value"]))
assert len(D) == len(F) == len(W) == len(phases), (len(D), len(F), len(W), len(phases))
print("| call | shape (B,N,K,C) | kernels of one call | us | algorithmic MB | GB/s | of 8 TB/s | FETCH_SIZE x2 MB | WRITE_SIZE MB | counted / algorithmic |")
print("|---|---|---|---|---|---|---|---|---|---|")
for ph, d, f, w in zip(phases, D, F, W):
    reps = ph["reps"]
    names = collections.Counter(n for n, _ in d)
    kern = ", ".join("%s x%d" % (n.split("<")[0], c // reps) if c // reps > 1 else n.split("<")[0] for n, c in names.items())
    us = sum(v for _, v in d) / reps / 1e3
    fmb = 2 * sum(v for _, v in f) / reps * 1024 / 1e6
    wmb = sum(v for _, v in w) / reps * 1024 / 1e6
    mb = ph["bytes"] / 1e6
    print("| %s | %s | %s | %.1f | %.1f | %.0f | %.3f | %.1f | %.1f | %.2f |" % (
        ph["op"], tuple(ph["shape_BNKC"]), kern, us, mb, ph["bytes"] / us / 1e3, ph["bytes"] / us / 1e3 / 8000, fmb, wmb, (fmb + wmb) / mb))
    records.append({"op": ph["op"], "shape_BNKC": ph["shape_BNKC"], "us": round(us, 1), "algorithmic_bytes": ph["bytes"],
                    "fetch_bytes_x2": round(fmb * 1e6), "write_bytes": round(wmb * 1e6), "traffic_bytes": round((fmb + wmb) * 1e6)})
if json_out:
    json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over tools/op_probe.py; FETCH_SIZE x 2 "
                       "(gfx950 counts 64 B per 128-B request); per call (mean of 3)", "source_id": _lib.source_id(), "rows": records}, open(json_out, "w"), indent=1)
