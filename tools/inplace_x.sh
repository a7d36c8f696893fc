#!/bin/bash
# config 2 with the level-0 x rows copied by the gather launch (GSAGE_MEAN_INPLACE_X=0) against read in place by K5 / K5b
# (default); the engine tests once more on the copy path.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
run() { timeout 200 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --extra "" 2>/dev/null | grep "^{" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms/step, gather launch %.1f us' % (d['ms_per_step'], d['roofline']['avg_launch_us']))"; }
echo -n "x rows copied:   "; GSAGE_MEAN_INPLACE_X=0 run
echo -n "x rows in place: "; run
echo -n "x rows copied:   "; GSAGE_MEAN_INPLACE_X=0 run
echo -n "x rows in place: "; run
GSAGE_MEAN_INPLACE_X=0 timeout 300 python -m pytest tests/test_gpu_engine.py tests/test_gpu_engine_golden.py -m gpu -q -x --timeout 200 -p no:cacheprovider 2>&1 | grep -E "passed|failed" | tail -2
timeout 100 rocprofv3 --kernel-trace --stats -d /tmp/px -o r --output-format csv -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline --extra '' > /dev/null 2>&1; head -7 $(find /tmp/px -name "*kernel_stats*.csv" | head -1) | cut -c1-60,150-260
