"""Per-kernel averages of whatever counters a `rocprofv3 --kernel-trace --pmc ...` pass collected.
usage: pmc_any.py <counter_collection.csv> [kernel substring]"""
import csv, json, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")
    if "gsage::" in name and (len(sys.argv) < 3 or sys.argv[2] in name):
        acc[(name, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = []
for (name, grid), ctrs in acc.items():
    rec = {"kernel": name, "grid_threads": grid, "launches": max(len(v) for v in ctrs.values())}
    rec.update({c: sum(v) / len(v) for c, v in sorted(ctrs.items())})
    out.append(rec)
out.sort(key=lambda r: -r.get("SQ_WAVE_CYCLES", r.get("GRBM_GUI_ACTIVE", 0)))
print(json.dumps(out, indent=1))
