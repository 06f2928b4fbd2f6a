"""Per-kernel mean durations of every phase of tools/op_probe.py from one rocprofv3 --kernel-trace pass:
    python tools/op_kernels.py <kernel_trace.csv>"""
import csv, sys, collections
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
seg, segs = None, []
for r in rows:
    n = r["Kernel_Name"]
    if "sigmoid" in n:
        seg = []
    elif "cos" in n and "elementwise" in n:
        if seg is not None:
            segs.append(seg)
        seg = None
    elif seg is not None:
        grid = r.get("Grid_Size", r.get("Grid_Size_X", "?")); wg = r.get("Workgroup_Size", r.get("Workgroup_Size_X", "?"))
        seg.append((n.split("(")[0][:56], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, grid, wg))
for s in segs:
    d = collections.OrderedDict()
    for n, t, g, w in s:
        d.setdefault((n, g, w), []).append(t)
    print(" | ".join("%s grid %s wg %s: %.1f us" % (k[0], k[1], k[2], sum(v) / len(v)) for k, v in d.items()))
