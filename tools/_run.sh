cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "pool or model" 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
python bench.py --no-cpu-baseline --aggregator max_pool --steps 30 --warmup 5 2>&1 | grep metric | cut -c1-200
