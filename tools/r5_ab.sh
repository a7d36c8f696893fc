#!/bin/bash
# round 5, GPU call 1: new-kernel tests, then the headline step A/B (seed level VALU / MFMA, finalisation launch / folded)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
OUT=gpurun_out; mkdir -p $OUT
python -c "import __graft_entry__ as g; g.ensure_built()" > $OUT/build.log 2>&1 || { tail -20 $OUT/build.log; exit 1; }
if [ "${R5_TESTS:-1}" = "1" ]; then
  timeout 900 python -m pytest tests/test_gpu_kernels.py -k "seed_level" tests/test_gpu_round5.py -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/r5_tests.log 2>&1
  echo "== new tests rc=$?"; tail -15 $OUT/r5_tests.log | cut -c1-300
fi
b() {  # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --steps 100 --warmup 20 --no-cpu-baseline --extra '' --min-time 0.4 > $OUT/ab_$name.log 2>&1
  python - "$name" $OUT/ab_$name.log <<'PY'
import sys, json
name, path = sys.argv[1:3]
try:
    d = json.loads([l for l in open(path) if l.startswith("{")][-1])
    r = d["roofline"]
    print("%-22s ms/step %.4f  seeds/s %.3fM  launches %.1f  gather %.1f us  seed %.1f us  k5 %.1f  k5b %.1f" % (
        name, d["ms_per_step"], d["value"] / 1e6, d["config"]["kernel_launches_per_step"],
        r.get("gather_launch", r)["avg_launch_us"],
        r.get("seed_level_launch", r).get("avg_launch_us", -1), r.get("k5_launch", {}).get("avg_launch_us", -1),
        r.get("k5b_launch", {}).get("avg_launch_us", -1)))
except Exception as e:
    print(name, "FAILED", e); print(open(path).read()[-1500:])
PY
}
for cfg in ${R5_CFGS:-base mfma mfma_fold f15 f30 f40}; do
  case $cfg in
    base)      b base GSAGE_TAIL_MFMA=0 GSAGE_FOLD_FINALIZE=0 ;;
    mfma)      b mfma GSAGE_TAIL_MFMA=1 GSAGE_FOLD_FINALIZE=0 ;;
    mfma_fold) b mfma_fold GSAGE_TAIL_MFMA=1 GSAGE_FOLD_FINALIZE=1 ;;
    f*)        b mfma_fold_$cfg GSAGE_TAIL_MFMA=1 GSAGE_FOLD_FINALIZE=1 GSAGE_TAIL_GATHER_FRAC=0.${cfg#f} ;;
    s*)        b stop5_$cfg GSAGE_TAIL_STOP=5 GSAGE_TAIL_GATHER_FRAC=0.${cfg#s} ;;
    t*)        b stop3_$cfg GSAGE_TAIL_STOP=3 GSAGE_TAIL_GATHER_FRAC=0.${cfg#t} ;;
    p*)        b spw4_$cfg GSAGE_HOPS_SPW=4 GSAGE_TAIL_GATHER_FRAC=0.${cfg#p} ;;
    w*)        b gather_wgs_$cfg GSAGE_TAIL_GATHER_WGS=${cfg#w} GSAGE_TAIL_GATHER_FRAC=1.5 ;;
    z0)        b finalize_launch GSAGE_FOLD_FINALIZE=0 ;;
    z1)        b folded_wide GSAGE_FOLD_FINALIZE=1 ;;
    a0)        b adam_descs_global GSAGE_ADAM_STAGE=0 ;;
    a1)        b adam_descs_staged GSAGE_ADAM_STAGE=1 ;;
    v0)        b gather_narrow GSAGE_GATHER_WIDE=0 ;;
    v1)        b gather_wide GSAGE_GATHER_WIDE=1 ;;
    q0)        b k1_in_gather_launch GSAGE_K1_IN_TAIL=0 ;;
    q*)        b k1_in_tail_$cfg GSAGE_K1_IN_TAIL=1 GSAGE_TAIL_SMP_WGS=${cfg#q} ;;
    x0)        b x_rows_copied GSAGE_MEAN_INPLACE_X=0 ;;
    x1)        b x_rows_in_place GSAGE_MEAN_INPLACE_X=1 ;;
    k0)        b k1_in_gather GSAGE_K1_IN_K5=0 ;;
    k1)        b k1_in_k5 GSAGE_K1_IN_K5=1 ;;
    g*)        b mfma_fold_$cfg GSAGE_TAIL_MFMA=1 GSAGE_FOLD_FINALIZE=1 GSAGE_TAIL_GATHER_FRAC=0.${cfg#g} ;;
    n*)        b mfma_$cfg GSAGE_TAIL_MFMA=1 GSAGE_FOLD_FINALIZE=0 GSAGE_TAIL_GATHER_FRAC=0.${cfg#n} ;;
  esac
done
