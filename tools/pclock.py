"""Per-kernel effective clock and MFMA utilisation from one rocprofv3 run (--kernel-trace + --pmc GRBM_GUI_ACTIVE SQ_*):
clock = GRBM_GUI_ACTIVE / duration; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x GRBM_GUI_ACTIVE)."""
import csv, sys, collections
trace, pmc = sys.argv[1], sys.argv[2]
dur = {}
for r in csv.DictReader(open(trace)):
    dur[r["Dispatch_Id"]] = (r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
cnt = collections.defaultdict(dict)
for r in csv.DictReader(open(pmc)):
    cnt[r["Dispatch_Id"]][r["Counter_Name"]] = float(r["Counter_Value"])
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for d, (name, ns) in dur.items():
    if "gemm_kernel" not in name or d not in cnt:
        continue
    key = name.split("(")[0][:60]
    a = agg[key]
    a["n"] += 1; a["ns"] += ns
    for k, v in cnt[d].items():
        a[k] += v
for key, a in sorted(agg.items()):
    if a["n"] < 4:
        continue
    gui = a.get("GRBM_GUI_ACTIVE", 0.0)
    print("%-62s n=%3d  %7.1f us  clock %.3f GHz  mfma busy %.3f  wave-cycles: wait_any %.3f wait_inst %.3f" % (
        key, a["n"], a["ns"] / a["n"] / 1e3, gui / a["ns"] if a["ns"] else 0,
        a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (4 * 256 * gui) if gui else 0,
        a.get("SQ_WAIT_ANY", 0) / a["SQ_WAVE_CYCLES"] if a.get("SQ_WAVE_CYCLES") else 0,
        a.get("SQ_WAIT_INST_ANY", 0) / a["SQ_WAVE_CYCLES"] if a.get("SQ_WAVE_CYCLES") else 0))
