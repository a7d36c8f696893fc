#!/bin/bash
# One gpurun call: gpu tests + smoke + bench (+ optional rocprofv3).  Everything lands in gpurun_out/.
# usage: tools/gpu_round.sh [tests] [smoke] [bench] [prof] [pmc]
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
WHAT="${*:-tests smoke bench}"
python -c "import __graft_entry__ as g; g.ensure_built()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
rocminfo 2>/dev/null | grep -m2 -E "gfx|Compute Unit" ; nproc
for w in $WHAT; do
  case $w in
    tests)
      timeout 1500 python -m pytest tests -m gpu -q --tb=short --timeout 600 -p no:cacheprovider > $OUT/tests.log 2>&1
      echo "== tests rc=$?"; tail -60 $OUT/tests.log ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
      echo "== smoke rc=$?"; tail -5 $OUT/smoke.log ;;
    bench)
      timeout 600 python bench.py --steps 50 --warmup 10 --no-graph --no-cpu-baseline > $OUT/bench_eager.log 2>&1
      echo "== bench eager rc=$?"; tail -3 $OUT/bench_eager.log
      timeout 900 python bench.py > $OUT/bench.log 2>&1
      echo "== bench rc=$?"; tail -3 $OUT/bench.log ;;
    prof)
      rm -rf $OUT/prof
      timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r --output-format csv -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $OUT/prof.log 2>&1
      echo "== prof rc=$?"; tail -3 $OUT/prof.log
      find $OUT/prof -name "*kernel_stats*.csv" | head -3
      f=$(find $OUT/prof -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && head -25 "$f"
      # keep the merged-back payload small
      find $OUT/prof -name "*kernel_trace*.csv" -size +20M -delete ;;
    pmc)
      rm -rf $OUT/pmc
      timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc -o f --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/pmc.log 2>&1
      echo "== pmc rc=$?"; tail -3 $OUT/pmc.log
      find $OUT/pmc -name "*.csv" | head; find $OUT/pmc -name "*kernel_trace*.csv" -size +20M -delete ;;
  esac
done
