#!/bin/bash
# seed-level gather-role share sweep at the papers shape (3 layers, fan-out 15/10/5, 256-byte rows)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
for f in 0 0.1 0.2 0.4 0.7; do
  echo -n "papers frac=$f: "; GSAGE_TAIL_GATHER_FRAC=$f python tools/bench_configs.py papers --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms/step  %.3f of roofline' % (d['ms_per_step'], d['frac_of_hbm_gather_roofline']))"
done
