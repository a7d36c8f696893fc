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
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"


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

# per-configuration PMC passes (tools/gpu_round.sh pmcx): the launch each `extra` roofline names
PICK = [("reddit", "reddit_gather", "k_gather_multi_adam"), ("reddit", "reddit_seed_level", "k_mean_tail_"),
        ("max_pool", "maxpool_k3", "k_pool_mlp_packed"), ("attention", "attention_k4", "k_attn_fused_fwd"),
        ("attention", "attention_k4_bwd", "k_attn_fused_bwd"),
        ("papers", "papers_gather", "k_gather_multi_adam"), ("papers", "papers_seed_level", "k_mean_tail_"),
        ("pokec", "pokec_k4", "k_attn_fused_fwd"), ("pokec", "pokec_k4_bwd", "k_attn_fused_bwd")]
launches = {}
for cfg, key, kname in PICK:
    rec = {}
    for c, field in (("FETCH_SIZE", "hbm_read_bytes_per_launch"), ("WRITE_SIZE", "hbm_write_bytes_per_launch")):
        pth = os.path.join(OUT, "pmcx_%s_%s.json" % (cfg, c))
        if not os.path.exists(pth):
            continue
        rows = [r for r in json.load(open(pth)) if kname in r["kernel"]]
        if not rows:
            continue
        r = max(rows, key=lambda r: r["grid_threads"])          # the launch over the last hop / the step's gather
        rec[field] = r[field]
        rec.setdefault("kernel", r["kernel"])
        rec.setdefault("grid_threads", r["grid_threads"])
        rec[c.lower() + "_factor"], rec[c.lower() + "_calibration"] = r["factor"], r["calibration"]
        rec["launches_sampled"] = r["launches"]
        # whole step: every gsage kernel's bytes x its launches, per launch of the named kernel (= per step)
        allr = [q for q in json.load(open(pth)) if "gsage::" in q["kernel"] and "k_gather_mean<" not in q["kernel"]]
        rec["step_" + field.replace("_per_launch", "")] = sum(q[field] * q["launches"] for q in allr) / r["launches"]
        shutil.copy(pth, os.path.join(PROF, "%s_pmc_%s_%s.json" % (tag, cfg, c.lower())))
    if rec:
        rec["source"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes, one calibration copy "
                         "each) over the %s configuration; tools/gpu_round.sh pmcx, tools/pmc_summary.py" % cfg)
        launches[key] = rec
if launches:
    json.dump(launches, open(os.path.join(PROF, tag + "_pmc_launches.json"), "w"), indent=1)

for cfg in ("max_pool", "attention", "papers", "pokec"):
    pth = os.path.join(OUT, "profx_%s_kernel_stats.csv" % cfg)
    if os.path.exists(pth):
        shutil.copy(pth, os.path.join(PROF, "%s_%s_kernel_stats.csv" % (tag, cfg)))

p = os.path.join(OUT, "parity_errors.jsonl")
if os.path.exists(p):
    shutil.copy(p, os.path.join(PROF, tag + "_parity_errors.jsonl"))
print("profiles/%s_* refreshed" % tag)
