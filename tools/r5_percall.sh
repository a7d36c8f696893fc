#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out; mkdir -p $OUT
for f in 0 1; do
  GSAGE_FOLD_FINALIZE=$f timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --extra '' --min-time 0.4 --per-step-copy > $OUT/pc_$f.log 2>&1
  python - $f $OUT/pc_$f.log <<'PY'
import sys, json
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
print("per-call fold=%s ms/step %.4f launches %.1f" % (sys.argv[1], d["ms_per_step"], d["config"]["kernel_launches_per_step"]))
PY
done
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_pc -o r --output-format csv -- env GSAGE_FOLD_FINALIZE=1 python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --extra '' --min-time 0 --per-step-copy > /tmp/prof_pc.log 2>&1
f=$(find /tmp/prof_pc -name "*kernel_stats*.csv" | head -1); head -12 "$f" | cut -c1-140
cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/prof_pc0 -o r --output-format csv -- env GSAGE_FOLD_FINALIZE=0 python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 10 --no-cpu-baseline --extra '' --min-time 0 --per-step-copy > /tmp/prof_pc0.log 2>&1
f=$(find /tmp/prof_pc0 -name "*kernel_stats*.csv" | head -1); head -12 "$f" | cut -c1-140
