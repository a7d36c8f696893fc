#!/bin/bash
# Round-end extras in one gpurun call: the default bench line again (so that profiles/ holds a line that read the
# refreshed PMC file), kernel stats of the secondary configurations, the 1-rank RCCL group.  Everything under timeouts.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out
mkdir -p $OUT
timeout 400 python bench.py > $OUT/bench.log 2>&1; echo "== bench rc=$?"
timeout 200 env GSAGE_FORCE_DDP=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --extra "" 2>/dev/null | grep "^{" > $OUT/bench_ddp1.json; echo "== ddp1 rc=$?"
for cfg in pokec papers; do
  rm -rf $OUT/prof_$cfg
  timeout 250 rocprofv3 --kernel-trace --stats -d $OUT/prof_$cfg -o r --output-format csv -- python tools/bench_configs.py $cfg > $OUT/prof_$cfg.log 2>&1
  echo "== prof $cfg rc=$?"; grep "^{" $OUT/prof_$cfg.log | cut -c1-200
  find $OUT/prof_$cfg -name "*kernel_trace*" -size +10M -delete
done
for agg in attention max_pool; do
  rm -rf $OUT/prof_x_$agg
  timeout 250 rocprofv3 --kernel-trace --stats -d $OUT/prof_x_$agg -o r --output-format csv -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --extra $agg --min-time 0 > $OUT/prof_x_$agg.log 2>&1
  echo "== prof $agg rc=$?"
  find $OUT/prof_x_$agg -name "*kernel_trace*" -size +10M -delete
done
