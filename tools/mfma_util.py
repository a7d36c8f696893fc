"""MFMA utilisation per kernel from `rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE`.
SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of every SIMD's matrix pipe (MI355X_MICROARCH.md: 32 per
v_mfma_f32_32x32x16_bf16; checked: K3's 79 GFLOP = 2.4 M such MFMAs, x 640/602 column padding -> 82 M, counter
87 M); GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (2.52 M "cycles" for a 139 us kernel = 8 x 139 us x
2.26 GHz).  utilisation = busy / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs).   usage: mfma_util.py <counter_collection.csv>"""
import csv, json, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "gsage::" in r["Kernel_Name"]:
        name = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[(name, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = []
for (name, grid), c in acc.items():
    if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
        continue
    busy = sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(c["SQ_VALU_MFMA_BUSY_CYCLES"])
    act = sum(c["GRBM_GUI_ACTIVE"]) / len(c["GRBM_GUI_ACTIVE"])
    if busy > 0:
        out.append({"kernel": name, "grid_threads": grid, "launches": len(c["GRBM_GUI_ACTIVE"]),
                    "mfma_busy_cycles": busy, "gui_active_cycles": act, "mfma_util": busy / (act / 8.0 * 1024.0)})
out.sort(key=lambda r: -r["mfma_busy_cycles"])
print(json.dumps(out, indent=1))
