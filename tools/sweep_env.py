"""Headline bench under several values of one environment switch: ms/step and the per-launch times of the chain.
    python tools/sweep_env.py NAME v1 v2 ... [-- extra bench.py arguments]"""
import json, os, subprocess, sys
args = sys.argv[1:]
extra = []
if "--" in args:
    i = args.index("--"); args, extra = args[:i], args[i + 1:]
name, values = args[0], args[1:]
for v in values:
    env = dict(os.environ); env[name] = v
    out = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline", "--extra", ""] + extra,
                         env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(name, v, "FAILED", out.stderr[-400:]); continue
    d = json.loads(line[-1]); r = d.get("roofline", {})
    print("%s=%s  %.4f ms/step  %.3f M/s  seed-level %.1f  k5 %.1f  k5b %.1f  update %.1f us" % (
        name, v, d["ms_per_step"], d["value"] / 1e6, r.get("avg_launch_us", 0), r.get("k5_launch", {}).get("avg_launch_us", 0),
        r.get("k5b_launch", {}).get("avg_launch_us", 0), r.get("gather_launch", {}).get("avg_launch_us", 0)), flush=True)
