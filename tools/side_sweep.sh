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
for cfg in ${@:-0:tail:tail:0.4 0.1:k5:k5:0.4 0.2:k5:k5:0.4 0.3:k5:k5:0.4 0.2:k5:tail:0.4 0.1:tail:tail:0.4 0.2:k5:fin:0.4}; do
  IFS=: read side at join tail <<< "$cfg"
  GSAGE_SIDE_GATHER_FRAC=$side GSAGE_SIDE_AT=$at GSAGE_SIDE_JOIN=$join GSAGE_TAIL_GATHER_FRAC=$tail run "side=$side at=$at join=$join tailfrac=$tail"
done
