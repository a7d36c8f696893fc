cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
python bench.py --no-cpu-baseline 2>&1 | grep -E "metric|rror" | cut -c1-200
python bench.py --no-cpu-baseline 2>&1 | grep -E "metric|rror" | cut -c1-200
