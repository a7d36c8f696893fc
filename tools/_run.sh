cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python tools/_dbg.py 2>&1 | grep -E "bad|expected"
python tools/kbench.py wgrad 2>&1 | grep wgrad
python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | grep -E "passed|failed|Error|error|assert" | tail -8
python bench.py --no-cpu-baseline 2>&1 | grep metric | cut -c1-200
