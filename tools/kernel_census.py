"""Per-kernel census of rocprofv3 output: launch geometry + register / LDS footprint and mean duration from a --kernel-trace csv;
effective clock, MFMA utilisation and wait shares from a --pmc counter_collection csv (GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES
SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY; GRBM_GUI_ACTIVE is summed over the 8 XCDs).  Kernels shorter than min_us are skipped.

    python tools/kernel_census.py <csv> [min_us]
"""
import csv, sys, collections
path = sys.argv[1]
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
rows = csv.DictReader(open(path))
agg = collections.OrderedDict()
is_pmc = "Counter_Name" in rows.fieldnames
seen = set()
for r in rows:
    ns = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    if is_pmc:
        key = (r["Kernel_Name"], r["Workgroup_Size"], r["Grid_Size"], r["LDS_Block_Size"], r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"])
    else:
        key = (r["Kernel_Name"], r["Workgroup_Size_X"], str(int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])), r["LDS_Block_Size"],
               r["VGPR_Count"], r["Accum_VGPR_Count"], r["SGPR_Count"], r["Scratch_Size"])
    a = agg.setdefault(key, collections.defaultdict(float))
    if r["Dispatch_Id"] not in seen:
        seen.add(r["Dispatch_Id"])
        a["n"] += 1; a["ns"] += ns
    if is_pmc:
        a[r["Counter_Name"]] += float(r["Counter_Value"])
for key, a in agg.items():
    us = a["ns"] / a["n"] / 1e3
    if us < min_us:
        continue
    print(key[0][:200])
    line = "    n=%d  %.1f us  wg %s  grid %s (%d workgroups)  LDS %s  VGPR %s  AGPR %s  SGPR %s  scratch %s" % (
        a["n"], us, key[1], key[2], int(key[2]) // max(1, int(key[1])), key[3], key[4], key[5], key[6], key[7])
    gui = a.get("GRBM_GUI_ACTIVE", 0.0)
    if gui:
        line += "\n    clock %.3f GHz" % (gui / 8.0 / a["ns"])
        if a.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            line += "  mfma busy %.3f" % (a["SQ_VALU_MFMA_BUSY_CYCLES"] / (4 * 256 * gui / 8.0))
        if a.get("SQ_WAVE_CYCLES"):
            line += "  wait_any %.3f wait_inst %.3f" % (a.get("SQ_WAIT_ANY", 0) / a["SQ_WAVE_CYCLES"], a.get("SQ_WAIT_INST_ANY", 0) / a["SQ_WAVE_CYCLES"])
    print(line)
