cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
python bench.py --no-cpu-baseline 2>&1 | grep metric | cut -c1-200
GSAGE_FORCE_DDP=1 python bench.py --no-cpu-baseline 2>&1 | grep metric | cut -c1-200
