#!/usr/bin/env python
"""Copies the summaries of the last `tools/gpu_round.sh tests bench bench20 prof pmc` call from
gpurun_out/ (scratch) into profiles/ (tracked) under this round's names.
usage: python tools/refresh_profiles.py [r02]"""
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT, PROF = os.path.join(ROOT, "gpurun_out"), os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"


def json_line(path):
    for line in open(path):
        if line.startswith("{"):
            return json.loads(line)
    raise SystemExit("no JSON line in " + path)


for src, dst in (("bench.log", "bench.json"), ("bench20.log", "bench_20steps.json")):
    p = os.path.join(OUT, src)
    if os.path.exists(p):
        json.dump(json_line(p), open(os.path.join(PROF, "%s_%s" % (tag, dst)), "w"), indent=1)

stats = glob.glob(os.path.join(OUT, "prof", "**", "*kernel_stats*.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(PROF, tag + "_kernel_stats.csv"))

per = {}
for c, key in (("FETCH_SIZE", "fetch_size"), ("WRITE_SIZE", "write_size")):
    p = os.path.join(OUT, "pmc_%s.json" % c)
    if os.path.exists(p):
        per[c] = json.load(open(p))
        shutil.copy(p, os.path.join(PROF, "%s_pmc_%s.json" % (tag, key)))
if len(per) == 2:
    rd = {r["kernel"]: r for r in per["FETCH_SIZE"]}
    wr = {r["kernel"]: r for r in per["WRITE_SIZE"]}
    k = [n for n in rd if "k_gather_multi_adam" in n][0]
    # per-step totals: every kernel's bytes x its launches per launch of the dominant kernel
    n_dom = rd[k]["launches"]
    step_r = sum(r["hbm_read_bytes_per_launch"] * r["launches"] for r in rd.values() if "gsage::" in r["kernel"]) / n_dom
    step_w = sum(r["hbm_write_bytes_per_launch"] * r["launches"] for r in wr.values() if "gsage::" in r["kernel"]) / n_dom
    json.dump({
        "kernel": k,
        "workload": "bench.py default (Reddit shape, B=512, fan-out 25/10)",
        "hbm_read_bytes_per_launch": rd[k]["hbm_read_bytes_per_launch"],
        "hbm_write_bytes_per_launch": wr[k]["hbm_write_bytes_per_launch"],
        "fetch_factor": rd[k]["factor"], "fetch_calibration": rd[k]["calibration"],
        "write_factor": wr[k]["factor"], "write_calibration": wr[k]["calibration"],
        "launches_sampled": n_dom,
        "step_hbm_read_bytes": step_r, "step_hbm_write_bytes": step_w,
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) over python "
                  "bench.py --steps 20 --warmup 5; tools/pmc_summary.py with the calibration pass tools/pmc_calib.py; "
                  "profiles/%s_pmc_fetch_size.json, %s_pmc_write_size.json" % (tag, tag),
    }, open(os.path.join(PROF, tag + "_pmc_gather_launch.json"), "w"), indent=1)

p = os.path.join(OUT, "parity_errors.jsonl")
if os.path.exists(p):
    shutil.copy(p, os.path.join(PROF, tag + "_parity_errors.jsonl"))
print("profiles/%s_* refreshed" % tag)
