#!/bin/bash
# step time of the default bench workload against the CU share of the gather stream (engine split mode)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
for g in 0 64 80 96 112 128; do
  echo -n "gather_cus=$g: "; GSAGE_GATHER_CUS=$g python bench.py --steps 200 --warmup 20 --no-cpu-baseline --extra "" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('%.4f ms/step  %.2f M seeds/s  step frac %.3f | gather launch %.1f us (%.2f TB/s)' % (d['ms_per_step'], d['value']/1e6, r['step']['frac'], r['avg_launch_us'], r['achieved']/1e3))"
done
