#!/bin/bash
# ms/step of the headline bench for the side-section gather (GSAGE_SIDE_GATHER_FRAC x GSAGE_SIDE_JOIN x GSAGE_TAIL_GATHER_FRAC)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
run() {
  python bench.py --steps 100 --warmup 20 --extra "" --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads([l for l in sys.stdin.read().splitlines() if l.startswith('{')][-1])
r = d['roofline']
print('%-46s %.4f ms/step  gather %.1f us  tail %.1f us' % (sys.argv[1], d['ms_per_step'], r['avg_launch_us'], r.get('seed_level_launch', {}).get('avg_launch_us', 0)))" "$1"
}
# cfg = side fraction : side start : side join : seed-level launch's share : projection launch's share
for cfg in ${@:-0:tail:tail:0.4:0 0:tail:tail:0.4:0.1 0:tail:tail:0.4:0.2 0:tail:tail:0.4:0.3 0:tail:tail:0.4:0.4 0:tail:tail:0.3:0.3 0:tail:tail:0.45:0.25}; do
  IFS=: read side at join tail k5 <<< "$cfg"
  GSAGE_K5_GATHER_FRAC=$k5 GSAGE_SIDE_GATHER_FRAC=$side GSAGE_SIDE_AT=$at GSAGE_SIDE_JOIN=$join GSAGE_TAIL_GATHER_FRAC=$tail run "side=$side at=$at join=$join tailfrac=$tail k5frac=$k5"
done
