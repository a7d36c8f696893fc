cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests/test_gpu_engine.py -m gpu -x -q -p no:cacheprovider -k "pool" 2>&1 | tail -3
python bench.py --no-cpu-baseline --aggregator max_pool --steps 30 --warmup 5 2>&1 | grep -E "metric|rror" | cut -c1-200
