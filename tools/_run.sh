cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_dist.py tests/test_gpu_engine.py -m gpu -x -q -p no:cacheprovider -k "two_rank or data_parallel" 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
GSAGE_FORCE_DDP=1 timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep metric | cut -c1-200
GSAGE_FORCE_DDP=1 timeout 300 python bench.py --no-cpu-baseline 2>&1 | grep metric | cut -c1-200
