cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
for a in max_pool mean_pool attention; do python bench.py --no-cpu-baseline --aggregator $a --steps 30 --warmup 5 2>&1 | grep metric | cut -c1-200; done
