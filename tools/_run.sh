cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_engine.py tests/test_gpu_kernels.py -m gpu -x -q -p no:cacheprovider -k "pool" 2>&1 | tail -12
for a in max_pool mean_pool; do python bench.py --no-cpu-baseline --aggregator $a --steps 30 --warmup 5 2>&1 | grep -E "metric|rror" | cut -c1-200; done
