#!/bin/bash
# The data-parallel step with a 1-rank RCCL group (GSAGE_FORCE_DDP=1): ms/step for every order the engine can issue
# it in, and a kernel timeline of the overlapped one.  usage (GPU box): bash tools/ddp_sweep.sh [trace]
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0 GSAGE_FORCE_DDP=1 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 MASTER_ADDR=127.0.0.1
OUT=gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.ensure_built()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
: > $OUT/ddp_sweep.txt
port=29600
for ov in 1 0; do for k1 in 1 0; do for nil in 1 0; do for one in 1 0; do
  [ $ov = 0 ] && [ $k1 = 0 ] && continue          # (K1 placement only matters around the side section)
  port=$((port+1))
  MASTER_PORT=$port GSAGE_DDP_OVERLAP=$ov GSAGE_DDP_K1_EARLY=$k1 GSAGE_DDP_NORM_IN_LAUNCH=$nil GSAGE_DDP_ONE_LIST=$one \
      timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --extra "" --min-time 0.3 > $OUT/ddp_one.log 2>&1
  r=$(grep '^{' $OUT/ddp_one.log | tail -1 | python -c "
import sys, json
try:
    d = json.loads(sys.stdin.read()); print('%.4f ms/step, %s launches' % (d['ms_per_step'], d['config']['kernel_launches_per_step']))
except Exception as e:
    print('failed', e)")
  echo "overlap=$ov k1_early=$k1 norm_in_launch=$nil one_list=$one : $r" | tee -a $OUT/ddp_sweep.txt
done; done; done; done
if [ "$1" = trace ]; then
  for ov in 0 1; do
    rm -rf $OUT/ddp_trace
    MASTER_PORT=2970$ov GSAGE_DDP_OVERLAP=$ov timeout 600 rocprofv3 --kernel-trace -d $OUT/ddp_trace -o t --output-format csv -- \
        python bench.py --steps 50 --warmup 10 --no-cpu-baseline --extra "" --min-time 0 > $OUT/ddp_trace.log 2>&1
    f=$(find $OUT/ddp_trace -name "*kernel_trace.csv" | head -1)
    [ -n "$f" ] && python tools/timeline.py "$f" 60 k_finalize_grads > $OUT/ddp_timeline_overlap$ov.txt && cat $OUT/ddp_timeline_overlap$ov.txt
    rm -rf $OUT/ddp_trace
  done
fi
