#!/bin/bash
# headline step vs the sampler role's seeds per workgroup and the side roles' position in the gather launch's grid
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.ensure_built()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
: > $OUT/spw_sweep.txt
for spw in 1 2 4 8; do for pos in 0.0 0.05; do
  GSAGE_HOPS_SPW=$spw GSAGE_SIDE_ROLE_POS=$pos timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --extra "" --min-time 0.4 > $OUT/spw_one.log 2>&1
  r=$(grep '^{' $OUT/spw_one.log | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('%.4f ms/step, gather launch %.1f us, seed level %.1f us' % (d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['seed_level_launch']['avg_launch_us']))
except Exception as e:
    print('failed', e)")
  echo "hops_spw=$spw side_pos=$pos : $r" | tee -a $OUT/spw_sweep.txt
done; done
