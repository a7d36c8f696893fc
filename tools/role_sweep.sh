#!/bin/bash
# In-step sweeps at config 2: where the optimizer / sampler roles sit in the gather launch's grid, and the seed-level
# launch's share of the last hop's gather.  usage (GPU box): bash tools/role_sweep.sh [pos] [frac]
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
run() { timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --extra "" 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms/step, gather launch %.1f us, seed level %.1f us' % (d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['seed_level_launch']['avg_launch_us']))"; }
WHAT="${*:-pos frac}"
for w in $WHAT; do
  case $w in
    pos) for p in 0.0 0.25 0.5 0.75 1.0; do echo -n "side roles at $p: "; GSAGE_SIDE_ROLE_POS=$p run; done ;;
    frac) for f in 0.25 0.3 0.35 0.4 0.45 0.5; do echo -n "tail gather frac $f: "; GSAGE_TAIL_GATHER_FRAC=$f run; done ;;
  esac
done
