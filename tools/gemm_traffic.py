"""HBM traffic of the tiled cmf_gemm launches inside a training step, from two rocprofv3 PMC passes over bench.py
(tools/session.sh pmc_gemm: --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs, as MI355X_MICROARCH.md prescribes).
FETCH_SIZE is doubled (gfx950 counts 64 B per 128-B request); both counters are reported in KB.  Dispatches of
gemm_kernel<128,128,...> and of the persistent pgemm_kernel<...> with >= 256 workgroups (the launches bench.py brackets: >= 1 GFLOP).  Prints one JSON object."""
import csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmflow_amd import _lib


def per_launch(path, counter):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        name = r.get("Kernel_Name", "")
        if ("gemm_kernel<128, 128" not in name and "gemm_kernel<256, 128" not in name and "pgemm_kernel<" not in name) or r.get("Counter_Name") != counter:
            continue
        wgs = int(r["Grid_Size"]) // max(1, int(r["Workgroup_Size"]))
        if wgs < 256:
            continue
        tot += float(r["Counter_Value"])
        n += 1
    return tot, n


f, nf = per_launch(sys.argv[1], "FETCH_SIZE")
w, nw = per_launch(sys.argv[2], "WRITE_SIZE")
out = {"kernel": "gemm_kernel<128,128,...> / <256,128,...> and pgemm_kernel<...> dispatches with >= 256 workgroups, bench.py --steps 3 --warmup 2 (all steps of the run)",
       "launches": nf, "fetch_bytes_per_launch": round(2.0 * f * 1024 / max(nf, 1)), "write_bytes_per_launch": round(w * 1024 / max(nw, 1)),
       "note": "FETCH_SIZE x 2 (gfx950: 64 B counted per 128-B request), WRITE_SIZE as reported; KB -> bytes"}
out["source_id"] = _lib.source_id()
out["traffic_bytes_per_launch"] = out["fetch_bytes_per_launch"] + out["write_bytes_per_launch"]
print(json.dumps(out))
