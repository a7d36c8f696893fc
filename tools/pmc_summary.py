"""Summarise `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs (separate passes: the
two counters do not fit the TCC slots together): HBM bytes per launch, per kernel.

Raw counters are KB.  Calibration (MI355X_MICROARCH.md, HBM section): the calibration pass
(tools/pmc_calib.py under the same counter) copies a known number of bytes with the step's access shape;
factor = known bytes / raw bytes of its k_gather_mean launches (FETCH_SIZE: 2.0 on gfx950 -- the guide's
"reports exactly half"; WRITE_SIZE: measured here).  Without a calibration file the guide's x2 is used
for FETCH_SIZE and WRITE_SIZE is reported raw with "calibrated": false.

usage: pmc_summary.py <counter_collection.csv of the bench run> [<counter_collection.csv of the calib run>]"""
import csv, json, sys
from collections import defaultdict

CAL_BYTES = 400_000 * 640 * 2


def load(path):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "gsage::" in r["Kernel_Name"] and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            name = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[(r["Counter_Name"], name, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return acc


run = load(sys.argv[1])
factor, calibrated = {"FETCH_SIZE": 2.0, "WRITE_SIZE": 1.0}, {"FETCH_SIZE": "guide (x2)", "WRITE_SIZE": False}
if len(sys.argv) > 2:
    for (ctr, name, grid), v in load(sys.argv[2]).items():
        if name.startswith("gsage::k_gather_mean<"):
            raw = sum(v) / len(v) * 1024
            factor[ctr] = CAL_BYTES / raw
            calibrated[ctr] = "measured: %d known bytes / %.0f raw" % (CAL_BYTES, raw)
out = []
for (ctr, name, grid), v in sorted(run.items(), key=lambda kv: -sum(kv[1])):
    raw = sum(v) / len(v)
    key = "hbm_read_bytes_per_launch" if ctr == "FETCH_SIZE" else "hbm_write_bytes_per_launch"
    out.append({"kernel": name, "grid_threads": grid, "launches": len(v), "counter": ctr, "raw_kb_avg": raw,
                "factor": factor[ctr], "calibration": calibrated[ctr], key: raw * 1024 * factor[ctr]})
print(json.dumps(out, indent=1))
