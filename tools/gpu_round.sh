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
      rm -f $OUT/parity_errors.jsonl
      # one pytest process per file: a GPU fault (which aborts the process) costs that file's summary only
      : > $OUT/tests.log; rc_all=0
      for f in ${GSAGE_TEST_FILES:-tests/test_*.py}; do
        GSAGE_PARITY_LOG=$PWD/$OUT/parity_errors.jsonl timeout 1700 python -X faulthandler -m pytest $f -m gpu -q --tb=short --timeout 900 -p no:cacheprovider ${GSAGE_PYTEST_ARGS} >> $OUT/tests.log 2>&1
        rc=$?; [ $rc -ne 0 ] && [ $rc -ne 5 ] && { rc_all=$rc; echo "!! $f rc=$rc" >> $OUT/tests.log; }
      done
      echo "== tests rc=$rc_all"; grep -E "passed|failed|^FAILED|^ERROR|!! |Error|error:|assert " $OUT/tests.log | grep -v "^  File" | cut -c1-260 | tail -70 ;;
    smoke)
      timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1
      echo "== smoke rc=$?"; tail -5 $OUT/smoke.log ;;
    bench)
      timeout 900 python bench.py > $OUT/bench.log 2>&1
      echo "== bench rc=$?"; tail -3 $OUT/bench.log ;;
    pool|attn)
      a=max_pool; [ $w = attn ] && a=attention
      timeout 600 python bench.py --aggregator $a --steps 50 --warmup 10 --no-cpu-baseline --extra '' > $OUT/bench_$w.log 2>&1
      echo "== bench $a rc=$?"; tail -1 $OUT/bench_$w.log | cut -c1-900 ;;
    sq)
      # where the waves of one configuration wait: SQ counters (one pass), then L2 hit / miss (second pass)
      a=${GSAGE_SQ_AGG:-max_pool}
      rocprofv3 -L > $OUT/pmc_list.txt 2>&1
      i=0
      for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
        i=$((i+1)); rm -rf $OUT/sq_${a}_$i
        timeout 600 rocprofv3 --kernel-trace --pmc $set -d $OUT/sq_${a}_$i -o f --output-format csv -- python bench.py --aggregator $a --steps 20 --warmup 5 --no-cpu-baseline --extra '' --min-time 0 > $OUT/sq_${a}_$i.log 2>&1
        echo "== sq $a pass $i rc=$?"
        c=$(find $OUT/sq_${a}_$i -name "*counter_collection.csv" | head -1)
        [ -n "$c" ] && python tools/pmc_any.py "$c" > $OUT/sq_${a}_$i.json
        find $OUT/sq_${a}_$i -name "*.csv" -size +2M -delete
      done ;;
    profx)
      # kernel-trace stats of one `extra` configuration: GSAGE_PROFX=attention|max_pool|pokec|papers
      c=${GSAGE_PROFX:-attention}
      case $c in
        attention|max_pool) cmd="python bench.py --aggregator $c --steps 50 --warmup 10 --no-cpu-baseline --extra '' --min-time 0" ;;
        *) cmd="python tools/bench_configs.py $c --steps 50" ;;
      esac
      rm -rf $OUT/profx_$c
      eval timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/profx_$c -o r --output-format csv -- $cmd > $OUT/profx_$c.log 2>&1
      echo "== profx $c rc=$?"
      f=$(find $OUT/profx_$c -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/profx_${c}_kernel_stats.csv && head -30 "$f" | cut -c1-150
      find $OUT/profx_$c -name "*kernel_trace*.csv" -size +20M -delete ;;
    gemmref)
      timeout 600 python tools/kbench.py gemmref > $OUT/gemmref.log 2>&1
      echo "== gemmref rc=$?"; grep gemmref $OUT/gemmref.log ;;
    benchx)
      # bench.py with caller-chosen arguments: GSAGE_BENCH_ARGS="--extra ddp_1rank,pokec --no-cpu-baseline"
      timeout 900 python bench.py ${GSAGE_BENCH_ARGS} > $OUT/benchx.log 2>&1
      echo "== benchx rc=$?"; tail -1 $OUT/benchx.log | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('ms_per_step', d['ms_per_step'])
for k, v in d.get('extra', {}).items():
    print(k, {a: (b if not isinstance(b, dict) else {x: y for x, y in b.items() if x in ('ms_per_step', 'error')}) for a, b in v.items() if a in ('ms_per_step', 'overlapped', 'inline', 'error', 'cli_seeds_per_s', 'wall_s')})
" ;;
    bench20)
      timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench20.log 2>&1
      echo "== bench20 rc=$?"; tail -2 $OUT/bench20.log ;;
    prof)
      rm -rf $OUT/prof
      timeout 900 rocprofv3 --kernel-trace --stats -d $OUT/prof -o r --output-format csv -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline --extra '' > $OUT/prof.log 2>&1
      echo "== prof rc=$?"; tail -3 $OUT/prof.log
      find $OUT/prof -name "*kernel_stats*.csv" | head -3
      f=$(find $OUT/prof -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && head -25 "$f"
      # keep the merged-back payload small
      find $OUT/prof -name "*kernel_trace*.csv" -size +20M -delete ;;
    pmcx)
      # per-configuration PMC passes for the `extra` rooflines: FETCH_SIZE / WRITE_SIZE separately, one calibration
      # run per counter; tools/refresh_profiles.py turns the per-kernel summaries into profiles/rNN_pmc_launches.json
      for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf $OUT/pmcx_cal_$c
        timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmcx_cal_$c -o f --output-format csv -- python tools/pmc_calib.py > $OUT/pmcx_cal_$c.log 2>&1
        b=$(find $OUT/pmcx_cal_$c -name "*counter_collection.csv" | head -1)
        for cfg in reddit max_pool attention papers pokec; do
          case $cfg in
            reddit) cmd="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --extra '' --min-time 0" ;;
            max_pool|attention) cmd="python bench.py --aggregator $cfg --steps 20 --warmup 5 --no-cpu-baseline --extra '' --min-time 0" ;;
            *) cmd="python tools/bench_configs.py $cfg --steps 20" ;;
          esac
          rm -rf $OUT/pmcx_${cfg}_$c
          eval timeout 900 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmcx_${cfg}_$c -o f --output-format csv -- $cmd > $OUT/pmcx_${cfg}_$c.log 2>&1
          echo "== pmcx $cfg $c rc=$?"
          a=$(find $OUT/pmcx_${cfg}_$c -name "*counter_collection.csv" | head -1)
          [ -n "$a" ] && python tools/pmc_summary.py "$a" $b > $OUT/pmcx_${cfg}_$c.json
          find $OUT/pmcx_${cfg}_$c -name "*.csv" -size +2M -delete
        done
        find $OUT/pmcx_cal_$c -name "*.csv" -size +2M -delete
      done ;;
    pmc)
      # FETCH_SIZE and WRITE_SIZE in separate passes (TCC slots), each with its calibration run
      for c in FETCH_SIZE WRITE_SIZE; do
        rm -rf $OUT/pmc_$c $OUT/pmc_cal_$c
        timeout 900 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_$c -o f --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --extra '' --min-time 0 > $OUT/pmc_$c.log 2>&1
        echo "== pmc $c rc=$?"; tail -2 $OUT/pmc_$c.log
        timeout 300 rocprofv3 --kernel-trace --pmc $c -d $OUT/pmc_cal_$c -o f --output-format csv -- python tools/pmc_calib.py > $OUT/pmc_cal_$c.log 2>&1
        echo "== pmc calib $c rc=$?"; tail -1 $OUT/pmc_cal_$c.log
        a=$(find $OUT/pmc_$c -name "*counter_collection.csv" | head -1); b=$(find $OUT/pmc_cal_$c -name "*counter_collection.csv" | head -1)
        [ -n "$a" ] && python tools/pmc_summary.py "$a" $b > $OUT/pmc_$c.json && head -40 $OUT/pmc_$c.json
        find $OUT/pmc_$c $OUT/pmc_cal_$c -name "*.csv" -size +8M -delete
      done ;;
  esac
done
